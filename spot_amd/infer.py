"""Long-video sliding-window inference (the reference's inference_on_mp4.py path, :204-277) on pre-decoded frames.

Video decoding (cv2) and rendering are outside the hot path; the input here is a float tensor [n_frames, 3, H, W] already
resized and normalised (what the reference's v2.Resize/ToDtype/Normalize pipeline yields, inference_on_mp4.py:520-526).
Windows have size = stride = clip_len (:531-532); the last window is zero-padded; windows are independent (the gated
shift never crosses a clip, the GRU restarts per clip), so with N GPUs rank r takes every N-th batch of windows and the
per-frame scores are gathered on the host: replicas only, no collective on the data path.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .util.postproc import ScoreAccumulator, events_from_scores, filter_events_by_score, non_max_suppression_events


def num_windows(n_frames, window, stride):
    """reference inference_on_mp4.py:174"""
    return max(1, math.ceil((n_frames - window) / stride + 1))


def window_batches(n_frames, clip_len, batch_size, rank=0, world=1):
    """Start frames of the windows this rank evaluates, grouped in batches (batch b goes to rank b % world)."""
    starts = [i * clip_len for i in range(num_windows(n_frames, clip_len, clip_len))]
    batches = [starts[i:i + batch_size] for i in range(0, len(starts), batch_size)]
    return [b for i, b in enumerate(batches) if i % world == rank]


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def pad_last_window(clip, clip_len, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """A window that runs past the end of the video is filled up with black frames.  The reference pads with zero frames
    BEFORE its Resize/ToDtype/Normalize transform (VideoFrameSlidingDataset, inference_on_mp4.py:523-524), so a padded frame of
    an already normalised float clip holds (0 - mean) / std per channel; a uint8 clip is padded with zero bytes (the stem
    kernel normalises them).  Padded frames reach real ones through the gated shift and the bidirectional GRU."""
    n = clip_len - clip.shape[0]
    if n <= 0:
        return clip
    pad = torch.zeros((n,) + tuple(clip.shape[1:]), dtype=clip.dtype, device=clip.device)
    if clip.dtype.is_floating_point:
        c = clip.shape[1]
        m = torch.tensor(mean[:c], dtype=torch.float32, device=clip.device)
        sd = torch.tensor(std[:c], dtype=torch.float32, device=clip.device)
        pad = pad + (-m / sd).to(clip.dtype).view(1, c, 1, 1)
    return torch.cat([clip, pad], 0)


def run_inference(model, frames, classes, clip_len, batch_size=16, postprocess=True, rank=0, world=1, video="video",
                  gather=None):
    """model: E2EModel; frames [n_frames, C, H, W]: float tensor (normalised; host or device), uint8 tensor (decoded), or a
    numpy array / memory map of either.  Returns the reference's
    [{'video', 'events'}] list (after score filter 0.15 and NMS 3 when postprocess)."""
    n_frames = frames.shape[0]
    acc = ScoreAccumulator(n_frames, len(classes) + 1)
    full_batch_seen = False
    pipelined = hasattr(model, "predict_async")
    pending = None

    def take(starts, scores, loc):
        if torch.is_tensor(scores):
            scores = scores.cpu().numpy()
            loc = None if loc is None else loc.cpu().numpy()
        for i, s in enumerate(starts):
            acc.add(s, scores[i], None if loc is None else loc[i])

    for starts in window_batches(n_frames, clip_len, batch_size, rank, world):
        clips = []
        for s in starts:
            clip = frames[s:s + clip_len]
            if not torch.is_tensor(clip):          # a numpy array or read-only memory map: copy the window
                clip = torch.from_numpy(np.array(clip))
            if clip.shape[0] < clip_len:
                clip = pad_last_window(clip, clip_len, getattr(model, "input_mean", IMAGENET_MEAN), getattr(model, "input_std", IMAGENET_STD))
            clips.append(clip)
        if full_batch_seen and len(clips) < batch_size:
            # a ragged last batch would make the engine lay out its activations for another batch size (and back again for
            # the next video): the last window is repeated instead and the repeats' outputs are dropped
            clips += [clips[-1]] * (batch_size - len(clips))
        full_batch_seen = full_batch_seen or len(clips) == batch_size
        batch = torch.stack(clips, 0)
        if pipelined:
            # this batch is enqueued before the previous one's results are read back: the host's launches for it overlap the
            # GPU's work on the previous batch instead of following the read-back's wait
            cur = model.predict_async(batch)
            if pending is not None:
                take(*pending)
            pending = (starts, cur[0], cur[1])
        else:
            out = model.predict(batch)
            take(starts, out[1], out[2] if len(out) > 2 else None)
    if pending is not None:
        take(*pending)
    if gather is not None:                      # multi-rank: disjoint frame ranges, sum on the host
        acc.scores, acc.support, acc.locations = gather(acc.scores, acc.support, acc.locations)
    scores, locations = acc.finish()
    pred_events = [{"video": video, "events": events_from_scores(scores, locations, classes, min_score=0.15 if postprocess else None)}]
    if postprocess:
        pred_events = filter_events_by_score(pred_events, fg_threshold=0.15)
        pred_events = non_max_suppression_events(pred_events, 3)
    return pred_events, scores


def main(argv=None):
    """Counterpart of the reference's `python inference_on_mp4.py --video_path V --checkpoint_path C --save_dir D
    --batch_size 16 --clip_len 64` (:495-552) for a video that is already decoded and resized: `--frames` is a .npy file
    [n_frames, 3, H, W], uint8 (normalised on the GPU) or float32 (already normalised).  The architecture comes from the
    `config.json` beside the checkpoint when there is one (what the training driver writes), else from the flags, whose
    defaults are the reference's (`get_args`, :27-48).  Writes D/predictions.json like the reference."""
    import argparse
    import json
    import os
    p = argparse.ArgumentParser()
    p.add_argument("--frames", type=str, required=True)
    p.add_argument("--checkpoint_path", type=str, required=True)
    p.add_argument("--save_dir", type=str, default="exp/demo1")
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--clip_len", type=int, default=64)
    p.add_argument("--feature_arch", type=str, default="rny008_gsm")
    p.add_argument("--temporal_arch", type=str, default="gru")
    p.add_argument("--classes", type=str, default="block dig net pass receive score serve set spike")
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--no_postprocess", action="store_true")
    args = p.parse_args(argv)
    from .e2e import E2EModel
    classes = args.classes.split()
    kw = {"feature_arch": args.feature_arch, "temporal_arch": args.temporal_arch, "predict_location": True}
    cfg_path = os.path.join(os.path.dirname(os.path.abspath(args.checkpoint_path)), "config.json")
    if os.path.exists(cfg_path):
        with open(cfg_path) as fp:
            cfg = json.load(fp)
        kw = {k: cfg.get(k, v) for k, v in kw.items()}
        for k in ("time_backward", "use_channel_attention"):
            kw[k] = bool(cfg.get(k, False))
        if cfg.get("num_classes") and cfg["num_classes"] != len(classes):
            classes = [f"class{i + 1}" for i in range(cfg["num_classes"])]
    model = E2EModel(len(classes) + 1, kw.pop("feature_arch"), kw.pop("temporal_arch"), clip_len=args.clip_len, modality="rgb",
                     device=args.device, verbose=False, **kw)
    model.load(torch.load(os.path.abspath(args.checkpoint_path), map_location="cpu", weights_only=True))
    frames = np.load(args.frames, mmap_mode="r")
    pred_events, _ = run_inference(model, frames, classes, args.clip_len, args.batch_size, postprocess=not args.no_postprocess,
                                   video=os.path.basename(args.frames))
    os.makedirs(args.save_dir, exist_ok=True)
    with open(os.path.join(args.save_dir, "predictions.json"), "w") as fp:
        json.dump(pred_events, fp)
    print("{} events -> {}".format(len(pred_events[0]["events"]), os.path.join(args.save_dir, "predictions.json")))
    return pred_events


if __name__ == "__main__":
    main()
