"""The reference's frame reader and clip sampler (dataset/frame.py:20-65, 156-333) in front of the hot path: JPEG frame
directories `frame_dir/<video>/000123.jpg` and the label JSON `[{"video", "num_frames", "fps", "events": [{"frame", "label",
"xy"}]}]` become batches of DECODED uint8 clips.  Everything after decoding that the reference does on the host -- scaling
by 1/255, the random transform stack, Normalize -- runs on the GPU here (spot_amd/augment.py, the stem kernel), so a
sample's `frame` is uint8 [T, C, H, W] and `xy` is the label file's location in [0, 1] of the source frame.

Decoding is PIL's (libjpeg), the reference's torchvision.io.read_image (libjpeg as well); torchvision is not available where
this was written, so "same pixels" is by construction of the two libraries' default decoder settings (slow integer IDCT,
fancy upsampling), not by a test against it: parity of the decoded pixels is unpinned.  The sampler draws from Python's
`random` in the reference's order (foreground coin, video / event choice, start offset, per-frame stride jitter), so a seeded
run picks the reference's clips.
"""
from __future__ import annotations

import json
import os
import random

import numpy as np
import torch

DEFAULT_PAD_LEN = 5          # dataset/frame.py:69: empty frames before the start and after the end of a video
LOCATION_DATASETS = ("kovo", "imrcvolley", "vnl")      # datasets whose events carry "xy" (dataset/frame.py:260-264)


class FrameReader:
    """dataset/frame.py:20-65.  read_frame: decoded uint8 [C, H, W] (the reference divides by 255 here; the GPU does that);
    a missing or unreadable file raises RuntimeError like torchvision.io.read_image."""

    IMG_NAME = "{:06d}.jpg"

    def __init__(self, frame_dir, modality):
        self._frame_dir = frame_dir
        self._is_flow = modality == "flow"

    def read_frame(self, frame_path):
        from PIL import Image
        try:
            with Image.open(frame_path) as im:
                arr = np.asarray(im.convert("RGB"))
        except (OSError, ValueError) as err:
            raise RuntimeError(f"cannot read {frame_path}: {err}") from err
        img = torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous()
        if self._is_flow:
            img = img[1:, :, :]          # GB channels contain data
        return img

    def load_frames(self, video_name, start, end, pad=False, stride=1, randomize=False):
        """Frames start, start + stride, .. < end of a video as uint8 [T', C, H, W].  Frames before 0 are zero frames in
        front (always); frames that cannot be read are counted and, with `pad`, appended as zero frames at the END (the
        reference's behaviour: a hole in the middle shifts the later frames forward)."""
        ret = []
        n_pad_start = 0
        n_pad_end = 0
        for frame_num in range(start, end, stride):
            if randomize and stride > 1:
                frame_num += random.randint(0, stride - 1)
            if frame_num < 0:
                n_pad_start += 1
                continue
            frame_path = os.path.join(self._frame_dir, video_name, FrameReader.IMG_NAME.format(frame_num))
            try:
                ret.append(self.read_frame(frame_path))
            except RuntimeError:
                n_pad_end += 1
        ret = torch.stack(ret, dim=0)          # (an all-padding clip has nothing to stack, as in the reference)
        if n_pad_start > 0 or (pad and n_pad_end > 0):
            ret = torch.nn.functional.pad(ret, (0, 0, 0, 0, 0, 0, n_pad_start, n_pad_end if pad else 0))
        return ret


def load_json(path):
    with open(path) as fp:
        return json.load(fp)


class ActionSpotDataset(torch.utils.data.Dataset):
    """dataset/frame.py:156-333: `dataset_len` random clips per epoch, videos drawn in proportion to their length (or, with
    `fg_upsample`, a clip around a random event with probability 1 - fg_upsample), labels dilated by `dilate_len` frames.
    An item: {"video", "start", "end", "frame" uint8 [T, C, H, W], "contains_event", "label" int64 [T], "xy" float32 [T, 2] |
    None}.  The reference's per-item mixup branch reads an attribute it never sets (`_gpu_transform`, :345); mixup lives in
    the batch wrapper spot_amd.train.MixupClips instead and `mixup=True` is refused here."""

    def __init__(self, classes, label_file, frame_dir, modality, clip_len, dataset_len, is_eval=True, crop_dim=None, stride=1,
                 same_transform=True, dilate_len=0, mixup=False, pad_len=DEFAULT_PAD_LEN, fg_upsample=-1, dataset="finediving"):
        if mixup:
            raise NotImplementedError("mixup is applied to whole batches: wrap the loader in spot_amd.train.MixupClips")
        self._dataset = dataset
        self._src_file = label_file
        self._labels = load_json(label_file) if isinstance(label_file, (str, os.PathLike)) else label_file
        self._class_dict = classes
        self._video_idxs = {x["video"]: i for i, x in enumerate(self._labels)}
        num_frames = [v["num_frames"] for v in self._labels]
        self._weights_by_length = np.array(num_frames) / np.sum(num_frames)
        assert clip_len > 0 and stride > 0 and dataset_len > 0 and pad_len >= 0
        self._clip_len, self._stride, self._dataset_len, self._pad_len = clip_len, stride, dataset_len, pad_len
        self._is_eval = is_eval
        self._dilate_len = dilate_len
        self._fg_upsample = fg_upsample
        if self._fg_upsample > 0:
            self._flat_labels = [(i, e["frame"]) for i, x in enumerate(self._labels) for e in x["events"] if e["frame"] < x["num_frames"]]
        self._crop_dim = crop_dim
        self._frame_reader = FrameReader(frame_dir, modality)

    def _sample_uniform(self):
        video_meta = random.choices(self._labels, weights=self._weights_by_length)[0]
        video_len = video_meta["num_frames"]
        base_idx = -self._pad_len * self._stride + random.randint(
            0, max(0, video_len - 1 + (2 * self._pad_len - self._clip_len) * self._stride))
        return video_meta, base_idx

    def _sample_foreground(self):
        video_idx, frame_idx = random.choices(self._flat_labels)[0]
        video_meta = self._labels[video_idx]
        video_len = video_meta["num_frames"]
        lower_bound = max(-self._pad_len * self._stride, frame_idx - self._clip_len * self._stride + 1)
        upper_bound = min(video_len - 1 + (self._pad_len - self._clip_len) * self._stride, frame_idx)
        base_idx = random.randint(lower_bound, upper_bound) if upper_bound > lower_bound else lower_bound
        assert base_idx <= frame_idx
        assert base_idx + self._clip_len > frame_idx
        return video_meta, base_idx

    def clip_labels(self, video_meta, base_idx):
        """Per-frame class indices and event locations of the clip that starts at base_idx (dataset/frame.py:257-285)."""
        labels = np.zeros(self._clip_len, np.int64)
        event_xys = np.zeros((self._clip_len, 2), np.float32) if any(k in self._dataset for k in LOCATION_DATASETS) else None
        for event in video_meta["events"]:
            label_idx = (event["frame"] - base_idx) // self._stride
            if -self._dilate_len <= label_idx < self._clip_len + self._dilate_len:
                label = self._class_dict[event["label"]]
                for i in range(max(0, label_idx - self._dilate_len), min(self._clip_len, label_idx + self._dilate_len + 1)):
                    labels[i] = label
                    if event_xys is not None:
                        event_xys[i] = event.get("xy", [0, 0])
        return labels, event_xys

    def _get_one(self):
        if self._fg_upsample > 0 and random.random() >= self._fg_upsample:
            video_meta, base_idx = self._sample_foreground()
        else:
            video_meta, base_idx = self._sample_uniform()
        labels, event_xys = self.clip_labels(video_meta, base_idx)
        frames = self._frame_reader.load_frames(video_meta["video"], base_idx, base_idx + self._clip_len * self._stride, pad=True,
                                                stride=self._stride, randomize=not self._is_eval)
        return {"video": video_meta["video"], "start": base_idx, "end": base_idx + self._clip_len * self._stride, "frame": frames,
                "contains_event": int(np.sum(labels) > 0), "label": labels, "xy": event_xys}

    def __getitem__(self, unused):
        return self._get_one()

    def __len__(self):
        return self._dataset_len

    def print_info(self):
        num_frames = sum(x["num_frames"] for x in self._labels)
        num_events = sum(len(x["events"]) for x in self._labels)
        print("{} : {} videos, {} frames, {:0.5f}% non-bg".format(self._src_file, len(self._labels), num_frames, num_events / num_frames * 100))


class JpegVideo:
    """A video's frame directory with the indexing spot_amd.evaluate.VideoClipDataset asks of a decoded array: `.shape`
    [num_frames, C, H, W] and `video[a:b:s]` -> uint8 frames (unreadable frames become zero frames at the end of the slice,
    FrameReader.load_frames' rule).  Frames are decoded when a clip is cut, never the whole video."""

    def __init__(self, reader, video, num_frames):
        self._reader, self._video = reader, video
        first = reader.read_frame(os.path.join(reader._frame_dir, video, FrameReader.IMG_NAME.format(0)))
        self.shape = (num_frames,) + tuple(first.shape)
        self.dtype = first.dtype

    def __getitem__(self, sl):
        start, stop, step = sl.indices(self.shape[0])
        if stop <= start:
            return torch.zeros((0,) + self.shape[1:], dtype=self.dtype)
        return self._reader.load_frames(self._video, start, stop, pad=True, stride=step)


def _collate(items):
    out = {"frame": torch.stack([x["frame"] for x in items]),
           "label": torch.from_numpy(np.stack([x["label"] for x in items])),
           "contains_event": torch.tensor([x["contains_event"] for x in items]),
           "video": [x["video"] for x in items], "start": torch.tensor([x["start"] for x in items])}
    if items[0]["xy"] is not None:
        out["xy"] = torch.from_numpy(np.stack([x["xy"] for x in items]))
    return out


class JpegClips:
    """The training / validation loader of the reference's main (train_e2e_spatial.py:1072-1093) over an ActionSpotDataset:
    batches {"frame" uint8 [B, T, C, H, W], "label" [B, T], "xy" [B, T, 2], "contains_event" [B]}; worker w of epoch e seeds
    Python's `random` with w + 100 e (the reference's worker_init_fn; with no workers the loading process itself is seeded
    so); rank r of `world` ranks offsets the seed by 1000 r so that ranks draw different clips."""

    def __init__(self, dataset, batch_size, num_workers=0, rank=0, world=1):
        self.dataset, self.batch_size, self.num_workers, self.rank, self.world = dataset, batch_size, num_workers, rank, world
        self.epoch = 0

    def __len__(self):
        return len(self.dataset) // self.batch_size // self.world

    def __iter__(self):
        base = self.epoch * 100 + self.rank * 1000
        self.epoch += 1
        n = len(self) * self.batch_size
        if self.num_workers == 0:
            random.seed(base)
            for lo in range(0, n, self.batch_size):
                yield _collate([self.dataset[i] for i in range(lo, lo + self.batch_size)])
            return
        loader = torch.utils.data.DataLoader(torch.utils.data.Subset(self.dataset, range(n)), shuffle=False, batch_size=self.batch_size,
                                             num_workers=self.num_workers, collate_fn=_collate, drop_last=True,
                                             worker_init_fn=lambda wid: random.seed(wid + base))
        yield from loader
