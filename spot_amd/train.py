"""Training driver around the hot path: the counterpart of the reference's `main` (train_e2e_spatial.py:1034-1236) for
the parts that touch the model -- optimizer, warm-up + cosine schedule, epochs, checkpoints, resume -- with the reference's
flag names, file names and file contents, so a `save_dir` written here is read by the reference and the other way round.

At >10^4 frames/s a GPU is best fed from pre-decoded clips; the loader's random augmentation (dataset/frame.py:78-140) runs on
the GPU under `--augment` (spot_amd/augment.py).  `frame_dir` holds shards `train_*.npz` / `val_*.npz` with

    frame uint8 [n, clip_len, 3, H, W]   decoded RGB frames (the stem normalises them on the GPU)
    label int64 [n, clip_len]            0 = background
    xy    float32 [n, clip_len, 2]       event location in [0, 1]

or is the literal string `synthetic[:n_clips]`, or -- the reference's own layout -- holds JPEG frames `<video>/000123.jpg` with the
label files `data/<dataset>/{train,val}.json` (spot_amd/dataset.py: the reference's frame reader and clip sampler; frames of
another size than crop_dim need `--augment`).  `--criterion map` (best epoch by validation mAP, from `--start_val_epoch`
on) additionally needs whole validation videos: `val.json` in the reference's label format (data/<dataset>/val.json) and
`val_videos/<video>.npy`, uint8 [num_frames, 3, H, W], which are memory-mapped; `synthetic` makes two random videos.

    python -m spot_amd.train DATASET FRAME_DIR -m rny008_gsm -t gru --clip_len 64 --batch_size 16 --predict_location -s DIR

`--fused` trains through `E2EModel.train_step` (loss, backward and AdamW in HIP, no host synchronisation inside an epoch)
instead of the reference-style `epoch()` loop; it is implied when the script runs under torchrun:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m spot_amd.train ... (same flags)

Then `--batch_size` is per GPU (16 clips each, as the reference's nn.DataParallel run at 8x the batch), rank r takes every
world-th batch, gradients are all-reduced over RCCL (spot_amd/dp.py), rank 0 writes the files.  The optimizer file keeps
torch.optim.AdamW's format in both modes, so a run can be resumed in the other one, or by the reference.
"""
from __future__ import annotations

import argparse
import random
import glob
import json
import os
import re

import numpy as np
import torch
from torch.optim.lr_scheduler import ChainedScheduler, CosineAnnealingLR, LinearLR

FEATURE_ARCHS = ["rny008", "rny008_tsm", "rny008_gsm"]


def get_args(argv=None):
    """The reference's flags (train_e2e_spatial.py:45-177) that reach the model or the training loop."""
    p = argparse.ArgumentParser()
    p.add_argument("dataset", type=str)
    p.add_argument("frame_dir", type=str, help="directory of pre-decoded clip shards, or synthetic[:n_clips]")
    p.add_argument("--modality", type=str, choices=["rgb"], default="rgb")
    p.add_argument("-m", "--feature_arch", type=str, required=True, choices=FEATURE_ARCHS)
    p.add_argument("-t", "--temporal_arch", type=str, default="gru")
    p.add_argument("-p", "--pred_loc_arch", type=str, default="mlp")
    p.add_argument("--clip_len", type=int, default=100)
    p.add_argument("--crop_dim", type=int, default=224)
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("-ag", "--acc_grad_iter", type=int, default=1)
    p.add_argument("--warm_up_epochs", type=int, default=3)
    p.add_argument("--num_epochs", type=int, default=50)
    p.add_argument("-lr", "--learning_rate", type=float, default=0.001)
    p.add_argument("-s", "--save_dir", type=str, required=True)
    p.add_argument("--resume", action="store_true")
    p.add_argument("--predict_location", action="store_true")
    p.add_argument("--criterion", choices=["map", "loss"], default="map")      # the reference's default (train_e2e_spatial.py:136)
    p.add_argument("--start_val_epoch", type=int, default=30)
    p.add_argument("--time_backward", action="store_true")
    p.add_argument("--use_channel_attention", action="store_true")
    p.add_argument("--num_classes", type=int, default=None, help="event classes (without background) when data/<dataset>/class.txt is absent")
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--fused", action="store_true", help="HIP loss + backward + AdamW (implied under torchrun)")
    p.add_argument("--mixup", action="store_true", help="the reference loader's mixup (dataset/frame.py:338-353; train_e2e_spatial.py:141); "
                   "soft labels: not with --fused or --predict_location")
    p.add_argument("--dilate_len", type=int, default=0, help="label dilation when training (JPEG frame directories)")
    p.add_argument("--fg_upsample", type=float, default=None, help="sample around events with probability 1 - fg_upsample (JPEG frame directories)")
    p.add_argument("-j", "--num_workers", type=int, default=None, help="base number of decoding workers (JPEG frame directories)")
    p.add_argument("--debug_only", action="store_true", help="500 frames per epoch instead of 500,000 (train_e2e_spatial.py:1035-1037)")
    p.add_argument("--augment", action="store_true",
                   help="the reference loader's random augmentation (dataset/frame.py:78-140) on the GPU; shards may then hold "
                        "frames of any size (validation clips are resized to crop_dim)")
    return p.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------- data
class ClipShards:
    """Batches in the format the reference's loaders hand to `E2EModel.epoch` (train_e2e_spatial.py:537-547): a dict with
    `frame`, `label`, `xy`.  Frames stay uint8 until the stem kernel."""

    def __init__(self, files, batch_size, rank=0, world=1):
        self.files, self.batch_size, self.rank, self.world = sorted(files), batch_size, rank, world
        self._n = []
        for f in self.files:
            with np.load(f) as z:
                self._n.append(int(z["label"].shape[0]))

    def __len__(self):
        return sum(n // self.batch_size for n in self._n) // self.world      # every rank the same number of steps

    def __iter__(self):
        b, todo = 0, len(self) * self.world
        for f in self.files:
            with np.load(f) as z:
                frame, label, xy = z["frame"], z["label"], z["xy"]
            for i in range(0, label.shape[0] - self.batch_size + 1, self.batch_size):
                b += 1
                if b > todo:
                    return
                if (b - 1) % self.world != self.rank:
                    continue
                s = slice(i, i + self.batch_size)
                yield {"frame": torch.from_numpy(frame[s]), "label": torch.from_numpy(label[s]).long(),
                       "xy": torch.from_numpy(xy[s]).float()}


class SyntheticClips:
    """Random uint8 clips with the foreground rate of the reference's dataset (6137 events in 251110 frames)."""

    def __init__(self, n_clips, batch_size, clip_len, dim, num_classes, seed):
        self.n, self.bs, self.T, self.dim, self.nc, self.seed = n_clips, batch_size, clip_len, dim, num_classes, seed

    def __len__(self):
        return self.n // self.bs

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(len(self)):
            fg = torch.rand(self.bs, self.T, generator=g) < 0.0244
            yield {"frame": torch.randint(0, 256, (self.bs, self.T, 3, self.dim, self.dim), generator=g, dtype=torch.uint8),
                   "label": fg.long() * torch.randint(1, self.nc, (self.bs, self.T), generator=g),
                   "xy": torch.rand(self.bs, self.T, 2, generator=g)}


class AugmentedClips:
    """A loader's batches through the reference's transform stack (dataset/frame.py:78-140: random geometry and colour for
    training clips, Resize + Normalize for evaluation clips), whole batches at a time on the GPU: uint8 frames go up, the
    normalised fp32 tensor the reference's loader would have produced comes back, `xy` moves with the pixels."""

    def __init__(self, loader, device, crop_dim, is_eval, seed):
        from .augment import ClipAugmenter
        self.loader, self.device = loader, device
        self.aug = ClipAugmenter(crop_dim, is_eval=is_eval, seed=seed)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for batch in self.loader:
            frame, xy = self.aug(batch["frame"].to(self.device, non_blocking=True), batch["xy"])
            yield {**batch, "frame": frame, "xy": xy}


class MixupClips:
    """Mixup of the reference's training loader (dataset/frame.py:338-353) on whole batches: every clip is blended with a second
    clip (here: its neighbour in the batch, where the reference draws another clip from the dataset) with a weight
    l ~ Beta(0.2, 0.2); `label` becomes the [T, num_classes] distribution (num_classes counts the background class) l * onehot(a) + (1 - l) * onehot(b), `frame` the
    same blend of the two normalised clips, `contains_event` the larger of the two, `xy` stays the first clip's.
    As in the reference the soft labels only go through the classification loss: its location loss masks with `label != 0`
    on hard labels (train_e2e_spatial.py:582), so mixup and --predict_location do not combine there either."""

    MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)      # dataset/frame.py:124 on frames scaled by :29

    def __init__(self, loader, num_classes, seed):
        self.loader, self.nc, self.rng = loader, num_classes, random.Random(seed)

    def __len__(self):
        return len(self.loader)

    def normalised(self, frame):
        if frame.dtype != torch.uint8:
            return frame.float()
        shape = (1, 1, -1, 1, 1)
        mean = torch.tensor(self.MEAN[:frame.shape[2]], device=frame.device).view(shape)
        std = torch.tensor(self.STD[:frame.shape[2]], device=frame.device).view(shape)
        return (frame.float() / 255 - mean) / std

    def __iter__(self):
        for batch in self.loader:
            frame, label = self.normalised(batch["frame"]), batch["label"]
            B, T = label.shape
            lam = torch.tensor([self.rng.betavariate(0.2, 0.2) for _ in range(B)], dtype=torch.float32)
            other = torch.roll(torch.arange(B), 1)
            dist = torch.zeros(B, T, self.nc)
            dist.scatter_(2, label.unsqueeze(-1), lam.view(B, 1, 1).expand(B, T, 1))
            dist.scatter_add_(2, label[other].unsqueeze(-1), (1.0 - lam).view(B, 1, 1).expand(B, T, 1))
            lf = lam.to(frame.device).view(B, 1, 1, 1, 1)
            out = {**batch, "frame": lf * frame + (1.0 - lf) * frame[other.to(frame.device)], "label": dist}
            ce = batch.get("contains_event", (label.sum(1) > 0).long())
            out["contains_event"] = torch.maximum(ce, ce[other])
            yield out


def get_loaders(args, num_classes, rank=0, world=1):
    tr, va = _get_loaders(args, num_classes, rank, world)
    if getattr(args, "augment", False):
        tr = AugmentedClips(tr, args.device, args.crop_dim, False, 1000 + rank)
        va = AugmentedClips(va, args.device, args.crop_dim, True, 0)
    if getattr(args, "mixup", False):
        tr = MixupClips(tr, num_classes, 2000 + rank)
    return tr, va


def _get_loaders(args, num_classes, rank=0, world=1):
    bs = args.batch_size // args.acc_grad_iter
    if args.frame_dir.startswith("synthetic"):
        n = int(args.frame_dir.split(":")[1]) if ":" in args.frame_dir else 4 * args.batch_size
        mk = lambda seed, k: SyntheticClips(k, bs, args.clip_len, args.crop_dim, num_classes, seed)
        return mk(2 * rank, n), mk(2 * rank + 1, max(bs, n // 4))
    tr, va = (glob.glob(os.path.join(args.frame_dir, f"{s}_*.npz")) for s in ("train", "val"))
    if not tr or not va:
        if os.path.exists(os.path.join("data", args.dataset, "train.json")):
            return _jpeg_loaders(args, bs, rank, world)
        raise FileNotFoundError(f"no train_*.npz / val_*.npz shards in {args.frame_dir} and no data/{args.dataset}/train.json "
                                "for a JPEG frame directory")
    return ClipShards(tr, bs, rank, world), ClipShards(va, bs, rank, world)


EPOCH_NUM_FRAMES, BASE_NUM_WORKERS = 500000, 4          # train_e2e_spatial.py:32-34


def _jpeg_loaders(args, bs, rank, world):
    """The reference's datasets and loaders over a JPEG frame directory (train_e2e_spatial.py:880-932, 1072-1093): 500,000
    frames of random clips per training epoch, a quarter of that for validation, 2 x / 1 x the base number of workers."""
    from .dataset import ActionSpotDataset, JpegClips
    classes = load_classes(args)
    frames_per_epoch = 500 if getattr(args, "debug_only", False) else EPOCH_NUM_FRAMES
    dataset_len = max(bs * world, frames_per_epoch // args.clip_len)
    kw = {"crop_dim": args.crop_dim, "dilate_len": getattr(args, "dilate_len", 0), "dataset": args.dataset}
    if getattr(args, "fg_upsample", None) is not None:
        assert args.fg_upsample > 0
        kw["fg_upsample"] = args.fg_upsample
    base = BASE_NUM_WORKERS if getattr(args, "num_workers", None) is None else args.num_workers
    label = lambda split: os.path.join("data", args.dataset, split + ".json")
    tr = ActionSpotDataset(classes, label("train"), args.frame_dir, args.modality, args.clip_len, dataset_len, is_eval=False, **kw)
    va = ActionSpotDataset(classes, label("val"), args.frame_dir, args.modality, args.clip_len, max(bs * world, dataset_len // 4),
                           is_eval=True, **kw)
    tr.print_info()
    va.print_info()
    return (JpegClips(tr, bs, min(os.cpu_count(), 2 * base), rank, world), JpegClips(va, bs, min(os.cpu_count(), base), rank, world))


def get_val_videos(args, classes):
    """The whole-video validation set of `--criterion map` (reference train_e2e_spatial.py:888-904: clips of clip_len, no
    overlap)."""
    from .evaluate import VideoClipDataset
    if args.frame_dir.startswith("synthetic"):
        g = torch.Generator().manual_seed(2)
        names, labels, frames = sorted(classes), [], {}
        for v, n in (("synthetic0", 3 * args.clip_len - 7), ("synthetic1", 2 * args.clip_len)):
            ev = [{"frame": int(f), "label": names[int(f) % len(names)], "xy": torch.rand(2, generator=g).tolist()}
                  for f in sorted(torch.randperm(n, generator=g)[: max(2, n // 40)].tolist())]
            labels.append({"video": v, "num_frames": n, "fps": 25, "events": ev})
            frames[v] = torch.randint(0, 256, (n, 3, args.crop_dim, args.crop_dim), generator=g, dtype=torch.uint8)
    elif os.path.exists(os.path.join(args.frame_dir, "val.json")):
        with open(os.path.join(args.frame_dir, "val.json")) as fp:
            labels = json.load(fp)
        frames = {l["video"]: np.load(os.path.join(args.frame_dir, "val_videos", l["video"] + ".npy"), mmap_mode="r") for l in labels}
    else:
        # a JPEG frame directory: the reference's ActionSpotVideoDataset (dataset/frame.py:364-437), frames decoded clip by clip
        from .dataset import FrameReader, JpegVideo
        with open(os.path.join("data", args.dataset, "val.json")) as fp:
            labels = json.load(fp)
        if getattr(args, "debug_only", False):
            labels = labels[:2]
        reader = FrameReader(args.frame_dir, args.modality)
        frames = {l["video"]: JpegVideo(reader, l["video"], l["num_frames"]) for l in labels}
    return VideoClipDataset(classes, labels, frames, args.clip_len, overlap_len=0)


def load_classes(args):
    """reference util/dataset.py load_classes: one class name per line -> index from 1 (0 is background)."""
    f = os.path.join("data", args.dataset, "class.txt")
    if os.path.exists(f):
        with open(f) as fp:
            names = [l.strip() for l in fp if l.strip()]
        return {n: i + 1 for i, n in enumerate(names)}
    if args.num_classes is None:
        raise FileNotFoundError(f"{f} not found and --num_classes not given")
    return {f"class{i + 1}": i + 1 for i in range(args.num_classes)}


# ------------------------------------------------------------------------------------------------- schedule, save, resume
def get_lr_scheduler(args, optimizer, num_steps_per_epoch):
    """reference train_e2e_spatial.py:1014-1031: linear warm-up from 1 % chained with cosine annealing, stepped per iteration."""
    cosine_epochs = args.num_epochs - args.warm_up_epochs
    print("Using Linear Warmup ({}) + Cosine Annealing LR ({})".format(args.warm_up_epochs, cosine_epochs))
    return args.num_epochs, ChainedScheduler([
        LinearLR(optimizer, start_factor=0.01, end_factor=1.0, total_iters=args.warm_up_epochs * num_steps_per_epoch),
        CosineAnnealingLR(optimizer, num_steps_per_epoch * cosine_epochs)])


def get_last_epoch(save_dir):
    """reference train_e2e_spatial.py:830-838"""
    last = -1
    for name in os.listdir(save_dir):
        if name.startswith("optim_"):
            last = max(last, int(os.path.splitext(name)[0].split("optim_")[1]))
    return last


def get_best_epoch_and_history(save_dir, criterion):
    """reference train_e2e_spatial.py:841-849"""
    with open(os.path.join(save_dir, "loss.json")) as fp:
        data = json.load(fp)
    if criterion == "map":
        best = max(data, key=lambda x: x["val_mAP"]); key = "val_mAP"
    else:
        best = min(data, key=lambda x: x["val"]); key = "val"
    return data, best["epoch"], best[key]


def load_from_save(args, model, optimizer, scaler, lr_scheduler):
    """reference train_e2e_spatial.py:936-981: the newest checkpoint without its `_fc_out` tensors (a changed class list
    changes their shape) loaded non-strictly; optimizer/scaler/scheduler state only on --resume and only when nothing was
    filtered; a state that does not fit resets the optimizer with a warning instead of failing."""
    assert args.save_dir is not None
    epoch = get_last_epoch(args.save_dir)
    print("Loading from epoch {}".format(epoch))
    checkpoint = torch.load(os.path.join(args.save_dir, "checkpoint_{:03d}.pt".format(epoch)), map_location="cpu", weights_only=True)
    filtered = {k: v for k, v in checkpoint.items() if "_fc_out" not in k}
    filter_applied = len(filtered) < len(checkpoint)
    if filter_applied:
        print(f"Loading checkpoint with strict=False (filtered {len(checkpoint) - len(filtered)} keys containing '_fc_out')")
    else:
        print("Loading checkpoint with strict=False")
    model.load(filtered, strict=False)
    if args.resume:
        if filter_applied:
            print("Skipping optimizer/scheduler load due to model architecture change (keys filtered).")
        else:
            try:
                opt_data = torch.load(os.path.join(args.save_dir, "optim_{:03d}.pt".format(epoch)), map_location="cpu", weights_only=True)
                optimizer.load_state_dict(opt_data["optimizer_state_dict"])
                if scaler is not None:
                    scaler.load_state_dict(opt_data["scaler_state_dict"])
                lr_scheduler.load_state_dict(opt_data["lr_state_dict"])
            except Exception as e:       # noqa: BLE001 -- the reference catches everything here
                print(f"Warning: Failed to load optimizer/scheduler state (likely due to model architecture change). Resetting optimizer. Error: {e}")
    losses, best_epoch, best_criterion = get_best_epoch_and_history(args.save_dir, args.criterion)
    return epoch, losses, best_epoch, best_criterion


def store_config(file_path, args, num_epochs, classes):
    """reference train_e2e_spatial.py:984-1011 (the keys inference reads back: num_classes, archs, clip_len, crop_dim)."""
    config = {"dataset": args.dataset, "num_classes": len(classes), "modality": args.modality, "feature_arch": args.feature_arch,
              "temporal_arch": args.temporal_arch, "clip_len": args.clip_len, "batch_size": args.batch_size, "crop_dim": args.crop_dim,
              "num_epochs": num_epochs, "warm_up_epochs": args.warm_up_epochs, "learning_rate": args.learning_rate,
              "predict_location": args.predict_location, "pred_loc_arch": args.pred_loc_arch, "time_backward": args.time_backward, "use_channel_attention": args.use_channel_attention,
              "acc_grad_iter": args.acc_grad_iter, "criterion": args.criterion, "start_val_epoch": args.start_val_epoch}
    with open(file_path, "w") as fp:
        json.dump(config, fp, indent=2, sort_keys=True)


def clear_files(dir_name, re_str):
    for name in os.listdir(dir_name):
        if re.match(re_str, name):
            os.remove(os.path.join(dir_name, name))


# ------------------------------------------------------------------------------------------- fused step <-> torch AdamW
def export_adamw_state(model, optimizer):
    """The engine's flat AdamW moments as torch.optim.AdamW per-parameter state (what `optim_NNN.pt` holds)."""
    eng = model._model.engine
    if eng.adam_m is None:
        return
    for p in model._model.parameters():
        if id(p) in eng._goff:
            o, n, shape = eng._goff[id(p)]
            optimizer.state[p] = {"step": torch.tensor(float(eng.step_count)), "exp_avg": eng.adam_m[o:o + n].view(shape).clone(),
                                  "exp_avg_sq": eng.adam_v[o:o + n].view(shape).clone()}


def import_adamw_state(model, optimizer):
    """The reverse: moments and step count of a loaded torch AdamW state into the engine's flat buffers."""
    eng = model._model.engine
    steps = [int(st["step"]) for st in optimizer.state.values() if "step" in st]
    if not steps:
        return
    eng.adam_m, eng.adam_v = torch.zeros_like(eng.flat), torch.zeros_like(eng.flat)
    eng.step_count = max(steps)
    for p, st in optimizer.state.items():
        if id(p) in eng._goff and "exp_avg" in st:
            o, n, shape = eng._goff[id(p)]
            eng.adam_m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            eng.adam_v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))


def fused_epoch(trainer, loader, lr_scheduler, device, with_xy=True):
    """One training epoch through the fused step; the loss is read back once, at the end.  Returns the reference's dict."""
    import warnings
    total = torch.zeros(3, device=device)
    for batch in loader:
        trainer.lr = lr_scheduler.get_last_lr()[0]
        out = trainer.step(batch["frame"].to(device, non_blocking=True), batch["label"].to(device),
                           batch["xy"].to(device) if with_xy else None)
        total += out[:3]
        with warnings.catch_warnings():           # the torch optimizer only carries the schedule here; it never steps
            warnings.simplefilter("ignore", UserWarning)
            lr_scheduler.step()
    s, c, l = (total / max(1, len(loader))).tolist()
    return {"sum": s, "cls": c, "loc": l, "contrast": 0.0}


def scaler_state(scaler):
    """State dict for the optimizer file.  The reference resumes with optimizer, scaler and scheduler state inside one
    `try` (train_e2e_spatial.py:966-976) and its GradScaler is enabled: an empty dict (what a disabled scaler returns; bf16
    needs no loss scaling) would raise there after the optimizer has loaded and before the scheduler does.  So a disabled
    scaler is written as a fresh enabled one's state, which any GradScaler loads."""
    sd = scaler.state_dict() if scaler is not None else {}
    if not sd:
        sd = {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000, "_growth_tracker": 0}
    return sd


# ---------------------------------------------------------------------------------------------------------------- main
def main(args):
    from .e2e import E2EModel
    assert args.batch_size % args.acc_grad_iter == 0
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ       # under torchrun, whatever the number of ranks
    fused = args.fused or launched
    dist = None
    if launched:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        args.device = f"cuda:{local}"
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if fused and args.acc_grad_iter != 1:
        raise NotImplementedError("--fused takes the whole per-GPU batch in one step (acc_grad_iter 1)")
    if getattr(args, "mixup", False) and (fused or args.predict_location):
        raise NotImplementedError("--mixup gives soft labels: the fused step and the location loss take class indices "
                                  "(the reference's location loss fails on them too, train_e2e_spatial.py:582-591)")
    classes = load_classes(args)
    model = E2EModel(len(classes) + 1, args.feature_arch, args.temporal_arch, clip_len=args.clip_len, modality=args.modality,
                     device=args.device, predict_location=args.predict_location, pred_loc_arch=args.pred_loc_arch,
                     time_backward=args.time_backward, use_channel_attention=args.use_channel_attention, verbose=rank == 0)
    train_loader, val_loader = get_loaders(args, len(classes) + 1, rank, world)
    optimizer, scaler = model.get_optimizer({"lr": args.learning_rate})
    num_steps_per_epoch = len(train_loader) // args.acc_grad_iter
    num_epochs, lr_scheduler = get_lr_scheduler(args, optimizer, num_steps_per_epoch)
    val_videos = get_val_videos(args, classes) if args.criterion == "map" else None
    losses, best_epoch, best_criterion = [], None, (-0.1 if args.criterion == "map" else float("inf"))
    epoch = 0
    if args.resume:
        epoch, losses, best_epoch, best_criterion = load_from_save(args, model, optimizer, scaler, lr_scheduler)
        epoch += 1
    trainer = None
    if fused:
        from .dp import DataParallelTrainer
        trainer = DataParallelTrainer(model, dist, lr=args.learning_rate)       # broadcasts rank 0's weights
        import_adamw_state(model, optimizer)
    if rank == 0:
        os.makedirs(args.save_dir, exist_ok=True)
    for epoch in range(epoch, num_epochs):
        if fused:
            train_loss = fused_epoch(trainer, train_loader, lr_scheduler, args.device, args.predict_location)
        else:
            train_loss = model.epoch(train_loader, optimizer, scaler, lr_scheduler=lr_scheduler, acc_grad_iter=args.acc_grad_iter)
        if dist is not None:
            # BatchNorm running statistics are rank-local during training; validation, evaluation and the checkpoint all
            # use rank 0's, so that the criterion describes the weights that are saved
            for b in model._model.buffers():
                if b.dtype.is_floating_point:
                    dist.broadcast(b, src=0)
        val_loss = model.epoch(val_loader, acc_grad_iter=args.acc_grad_iter)
        if dist is not None:
            # train: every rank's loss is already divided by the GLOBAL denominators (dp.loss_denominators), i.e. it is
            # this rank's part of the global-batch loss -> sum; validation: per-rank means over equal shards -> average
            t = torch.tensor([train_loss[k] for k in ("sum", "cls", "loc")] + [val_loss[k] / world for k in ("sum", "cls", "loc")],
                             device=args.device, dtype=torch.float64)
            dist.all_reduce(t)
            t = t.tolist()
            train_loss.update(zip(("sum", "cls", "loc"), t[:3])); val_loss.update(zip(("sum", "cls", "loc"), t[3:]))
        if rank == 0:
            print("Epoch {}: train cls {:.5f} loc {:.5f} sum {:.5f} | val cls {:.5f} loc {:.5f} sum {:.5f}".format(
                epoch, train_loss["cls"], train_loss["loc"], train_loss["sum"], val_loss["cls"], val_loss["loc"], val_loss["sum"]))
        val_mAP = 0
        if args.criterion == "loss":
            if val_loss["sum"] < best_criterion:
                best_criterion, best_epoch = val_loss["sum"], epoch
                if rank == 0:
                    print("New best epoch!")
        elif epoch >= args.start_val_epoch:                  # reference train_e2e_spatial.py:1158-1178
            from .evaluate import evaluate
            val_mAP = float(evaluate(model, val_videos, "VAL", classes, os.path.join(args.save_dir, "pred-val.{}".format(epoch)),
                                     save_scores=False, predict_location=args.predict_location, dist=dist, device=args.device))
            if val_mAP > best_criterion:
                best_criterion, best_epoch = val_mAP, epoch
                if rank == 0:
                    print("New best epoch!")
        losses.append({"epoch": epoch, "train": train_loss["sum"], "val": val_loss["sum"], "val_mAP": val_mAP})
        if rank == 0:
            with open(os.path.join(args.save_dir, "loss.json"), "w") as fp:
                json.dump(losses, fp, indent=2, sort_keys=True)
            torch.save(model.state_dict(), os.path.join(args.save_dir, "checkpoint_{:03d}.pt".format(epoch)))
            clear_files(args.save_dir, r"optim_\d+\.pt")
            if fused:
                export_adamw_state(model, optimizer)
            torch.save({"optimizer_state_dict": optimizer.state_dict(),
                        "scaler_state_dict": scaler_state(scaler),
                        "lr_state_dict": lr_scheduler.state_dict()}, os.path.join(args.save_dir, "optim_{:03d}.pt".format(epoch)))
            store_config(os.path.join(args.save_dir, "config.json"), args, num_epochs, classes)
        if dist is not None:
            dist.barrier()
    if rank == 0:
        print("Best epoch: {}\n".format(best_epoch))
    if best_epoch is not None:
        model.load(torch.load(os.path.join(args.save_dir, "checkpoint_{:03d}.pt".format(best_epoch)), map_location="cpu", weights_only=True))
    if dist is not None:
        dist.destroy_process_group()
    return model, losses, best_epoch


if __name__ == "__main__":
    main(get_args())
