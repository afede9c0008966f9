"""Sliding-window score accumulation and event peak-picking (host side, numpy).

Same arithmetic, in the same order, as the reference's run_inference (inference_on_mp4.py:204-270) and its
post-processing helpers filter_events_by_score / non_max_suppression_events (util/score.py:246-300), so that identical
per-frame scores give bit-identical event lists.
"""
from __future__ import annotations

import numpy as np


class ScoreAccumulator:
    """scores[start:end] += window scores; support += 1; locations overwritten (inference_on_mp4.py:209-237)."""

    def __init__(self, n_frames: int, num_classes: int):
        self.scores = np.zeros((n_frames, num_classes), np.float32)
        self.support = np.zeros(n_frames, np.int32)
        self.locations = np.zeros((n_frames, 2), np.float32)

    def add(self, start: int, pred_scores: np.ndarray, pred_loc: np.ndarray | None = None):
        end = min(start + pred_scores.shape[0], len(self.support))
        self.scores[start:end, :] += pred_scores[: end - start, :]
        self.support[start:end] += 1
        if pred_loc is not None:
            self.locations[start:end, :] = pred_loc.reshape(-1, 2)[: end - start, :]

    def finish(self):
        assert np.min(self.support) > 0, self.support.tolist()
        self.scores /= self.support[:, None]
        return self.scores, self.locations


def events_from_scores(scores: np.ndarray, locations: np.ndarray, classes, min_score=None):
    """argmax != background -> event (inference_on_mp4.py:247-262).  min_score: leave out events the score filter that follows
    (filter_events_by_score with the same threshold) would drop anyway, so that no dict is built for them."""
    pred = np.argmax(scores, axis=1)
    idx = np.nonzero(pred)[0]                       # only the foreground frames are visited (a few per cent of a video)
    cls = pred[idx]
    if min_score is not None:
        keep = scores[idx, cls].astype(np.float64) >= min_score      # (in double, as the filter compares the Python floats)
        idx, cls = idx[keep], cls[keep]
    return [{"label": classes[c - 1], "frame": i, "score": s, "xy": xy}
            for i, c, s, xy in zip(idx.tolist(), cls.tolist(), scores[idx, cls].tolist(), locations[idx].tolist())]


def filter_events_by_score(data, fg_threshold):
    """util/score.py:246-258"""
    out = []
    for video in data:
        out.append({"video": video["video"], "events": [e for e in video["events"] if e["score"] >= fg_threshold],
                    "fps": video.get("fps", -1)})
    return out


def non_max_suppression_events(data, tol_t):
    """util/score.py:268-300: per label, walk events by frame; within tol_t of the group's current event keep the best."""
    for video in data:
        by_label = {}
        for e in video["events"]:
            by_label.setdefault(e["label"], []).append(e)
        kept = []
        for events in by_label.values():
            events.sort(key=lambda x: x["frame"])
            i = 0
            while i < len(events):
                cur = events[i]
                j = i + 1
                while j < len(events) and events[j]["frame"] - cur["frame"] <= tol_t:
                    if events[j]["score"] > cur["score"]:
                        cur = events[j]
                    j += 1
                kept.append(cur)
                i = j
        video["events"] = kept
    return data
