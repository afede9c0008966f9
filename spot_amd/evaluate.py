"""Whole-video evaluation: overlapping clips -> accumulated per-frame scores -> events -> error, F1, temporal and spatial mAP.

Follows the reference's evaluate() (train_e2e_spatial.py:656-827) and the clip layout of its ActionSpotVideoDataset
(dataset/frame.py:364-492).  Frames come pre-decoded as uint8 [N, 3, H, W] at the network's resolution (decoding and
resizing are outside this path); the engine normalises them on the GPU, so a zero byte frame used as padding becomes the
same (0 - mean) / std image the reference's pad-then-normalise yields.
"""
from __future__ import annotations

import copy
import gzip
import json
import os

import numpy as np
import torch

from .util.eval import process_frame_predictions, process_frame_predictions_with_location
from .util.score import compute_mAPs, compute_mAPs_with_locations

INFERENCE_BATCH_SIZE = 4          # train_e2e_spatial.py:38
DEFAULT_PAD_LEN = 5               # dataset/frame.py:69


class VideoClipDataset:
    """labels: the reference's label JSON (list of {"video", "num_frames", "fps", "events": [{"frame", "label", "xy"}]});
    frames: video -> uint8 tensor/array [num_frames, 3, H, W].  Clips start every (clip_len - overlap_len) * stride frames
    from -pad_len * stride; frames before 0 or past the end are zero (dataset/frame.py:403-437, :34-65)."""

    augment = False

    def __init__(self, classes, labels, frames, clip_len, overlap_len=0, stride=1, pad_len=DEFAULT_PAD_LEN, skip_partial_end=True):
        self._labels, self._class_dict, self._frames = labels, classes, frames
        self._video_idxs = {x["video"]: i for i, x in enumerate(labels)}
        self._clip_len, self._stride = clip_len, stride
        self._clips = []
        for l in labels:
            stop = max(0, l["num_frames"] - (overlap_len * stride) * int(skip_partial_end))
            starts = list(range(-pad_len * stride, stop, (clip_len - overlap_len) * stride))
            assert starts, l
            self._clips += [(l["video"], s) for s in starts]

    def __len__(self):
        return len(self._clips)

    def __getitem__(self, idx):
        video, start = self._clips[idx]
        src, st = self._frames[video], self._stride
        k_lo = min(self._clip_len, -(-max(0, -start) // st))
        k_hi = max(k_lo, min(self._clip_len, -(-(src.shape[0] - start) // st)))
        block = src[start + k_lo * st: start + k_hi * st: st]
        block = block if torch.is_tensor(block) else torch.from_numpy(np.array(block))     # a copy: the source may be a read-only map
        out = torch.zeros((self._clip_len,) + tuple(src.shape[1:]), dtype=block.dtype)
        out[k_lo:k_hi] = block
        return {"video": video, "start": start // self._stride, "frame": out}

    def batches(self, batch_size, rank=0, world=1):
        """Batches of clips in dataset order; batch b belongs to rank b % world (clips are independent)."""
        for b, lo in enumerate(range(0, len(self), batch_size)):
            if b % world == rank:
                items = [self[i] for i in range(lo, min(lo + batch_size, len(self)))]
                yield {"video": [x["video"] for x in items], "start": [x["start"] for x in items],
                       "frame": torch.stack([x["frame"] for x in items]), "index": list(range(lo, lo + len(items)))}

    def get_labels(self, video, with_locations=False):
        meta = self._labels[self._video_idxs[video]]
        n = -(-meta["num_frames"] // self._stride)
        labels, locations = np.zeros(n, np.int32), np.zeros((n, 2), np.float32)
        for e in meta["events"]:
            if e["frame"] < meta["num_frames"]:
                labels[e["frame"] // self._stride] = self._class_dict[e["label"]]
                locations[e["frame"] // self._stride] = e.get("xy", [0, 0])
            else:
                print("Warning: {} >= {} is past the end {}".format(e["frame"], meta["num_frames"], meta["video"]))
        return (labels, locations) if with_locations else labels

    @property
    def videos(self):
        return sorted((v["video"], v["num_frames"] // self._stride, v["fps"] / self._stride) for v in self._labels)

    @property
    def labels(self):
        if self._stride == 1:
            return self._labels
        out = copy.deepcopy(self._labels)
        for x in out:
            x["fps"] /= self._stride
            x["num_frames"] //= self._stride
            for e in x["events"]:
                e["frame"] //= self._stride
        return out


def accumulate(model, dataset, num_classes, predict_location, batch_size=INFERENCE_BATCH_SIZE, rank=0, world=1, writer=None):
    """video -> (score sums, support, last location) over this rank's clips (train_e2e_spatial.py:668-738).  `writer`
    (video -> int array, -1 = nobody) receives the dataset index of the clip that wrote each frame's location last."""
    pred = {v: (np.zeros((n, num_classes + 1), np.float32), np.zeros(n, np.int32), np.zeros((n, 2), np.float32))
            for v, n, _ in dataset.videos}
    for clip in dataset.batches(batch_size, rank, world):
        out = model.predict(clip["frame"])
        b_scores = out[1]
        b_loc = out[2].reshape(b_scores.shape[0], -1, 2) if predict_location else None
        for i, video in enumerate(clip["video"]):
            scores, support, locations = pred[video]
            s, start = b_scores[i], clip["start"][i]
            loc = b_loc[i] if predict_location else None
            if start < 0:
                s, loc, start = s[-start:], (loc[-start:] if predict_location else None), 0
            end = min(start + s.shape[0], scores.shape[0])
            scores[start:end] += s[: end - start]
            support[start:end] += 1
            if predict_location:
                locations[start:end] = loc[: end - start]
            if writer is not None:
                writer[video][start:end] = clip["index"][i]
    return pred


def accumulate_distributed(model, dataset, num_classes, predict_location, dist, batch_size=INFERENCE_BATCH_SIZE, device="cpu"):
    """Every rank evaluates every world-th batch of clips (replicas only: clips are independent), then the accumulators are
    summed over the ranks.  Where clips overlap, a frame's location is the one written by the clip latest in dataset order,
    as in the reference's single loop -- whichever rank ran it."""
    rank, world = dist.get_rank(), dist.get_world_size()
    writer = {v: np.full(n, -1, np.int64) for v, n, _ in dataset.videos}
    pred = accumulate(model, dataset, num_classes, predict_location, batch_size, rank, world, writer)
    out = {}
    for v in sorted(pred):
        scores, support, locations = (torch.from_numpy(a).to(device) for a in pred[v])
        w = torch.from_numpy(writer[v]).to(device)
        last = w.clone()
        dist.all_reduce(last, op=dist.ReduceOp.MAX)
        locations = locations * ((w == last) & (w >= 0)).unsqueeze(1)       # each frame survives on exactly one rank
        for t in (scores, support, locations):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out[v] = (scores.cpu().numpy(), support.cpu().numpy(), locations.cpu().numpy())
    return out


def evaluate(model, dataset, split, classes, save_pred=None, calc_stats=True, save_scores=True, predict_location=False,
             px_scale=224, batch_size=INFERENCE_BATCH_SIZE, dist=None, device="cpu", quiet=False):
    """Returns the harmonic mean of the temporal mAP (tolerances 1, 2, 4) and the spatial mAP when predict_location, else
    the mean temporal mAP over tolerances 0..5.  With `dist` (an initialised torch.distributed: "nccl" with device="cuda",
    or "gloo") the clips are dealt over the ranks and every rank returns the same number; rank 0 writes the files."""
    if dataset.augment:
        raise NotImplementedError("flip / multi-crop test-time augmentation")
    if dist is not None:
        pred = accumulate_distributed(model, dataset, len(classes), predict_location, dist, batch_size, device)
        quiet = quiet or dist.get_rank() != 0
        save_pred = save_pred if dist.get_rank() == 0 else None
    else:
        pred = accumulate(model, dataset, len(classes), predict_location, batch_size)
    proc = process_frame_predictions_with_location if predict_location else process_frame_predictions
    err, f1, pred_events, pred_events_high_recall, pred_scores = proc(
        dataset, classes, pred if predict_location else {v: p[:2] for v, p in pred.items()})
    avg_mAP = None
    if calc_stats:
        if not quiet:
            print("=== Results on {} (FG_THRES=0.25) ===".format(split))
            print("Error (frame-level): {:0.2f}\n".format(err.get() * 100))
            print("%-22s%9s%7s%7s%7s" % ("Exact frame", "F1", "TP", "FP", "FN"))
            for name in ["any"] + sorted(classes):
                k = classes[name] if name != "any" else None
                print("%-22s%9.2f%7d%7d%7d" % ((name, f1.get(k) * 100) + f1.tp_fp_fn(k)))
            print()
        if predict_location:
            mAPs_t, mAPs_p = compute_mAPs_with_locations(dataset.labels, pred_events_high_recall, px_scale=px_scale,
                                                         fg_threshold=0.25, quiet=quiet)
            t, s = np.mean(mAPs_t[1:]), np.mean(mAPs_p)
            avg_mAP = 2 * t * s / (t + s + 1e-6)
            if not quiet:
                print("Harmonic mean (temporal and spatial mAPs): {:0.2%}".format(avg_mAP))
        else:
            avg_mAP = np.mean(compute_mAPs(dataset.labels, pred_events_high_recall, quiet=quiet)[0])
    if save_pred is not None:
        os.makedirs(os.path.dirname(save_pred) or ".", exist_ok=True)
        with open(save_pred + ".json", "w") as fp:
            json.dump(pred_events, fp)
        with gzip.open(save_pred + ".recall.json.gz", "wt", encoding="ascii") as fp:
            json.dump(pred_events_high_recall, fp)
        if save_scores:
            with gzip.open(save_pred + ".score.json.gz", "wt", encoding="ascii") as fp:
                json.dump(pred_scores, fp)
    return avg_mAP
