"""Frame-level statistics and event extraction from accumulated per-frame scores (host side, numpy).

Restates the reference's util/eval.py with the same arithmetic in the same order: frame error rate (:6-20), foreground F1
with per-class and class-agnostic counts (:23-58), the conversion of accumulated window scores into event lists
(:61-173) and the windowed peak picking used by its stand-alone evaluator (:176-202).
"""
from __future__ import annotations

import copy

import numpy as np


class ErrorStat:
    """Fraction of frames whose arg-max class differs from the label (util/eval.py:6-20)."""

    def __init__(self):
        self._total = 0
        self._err = 0

    def update(self, true, pred):
        self._err += np.sum(true != pred)
        self._total += true.shape[0]

    def get(self):
        return self._err / self._total

    def get_acc(self):
        return 1.0 - self.get()


class ForegroundF1:
    """Exact-frame F1.  Key None counts "any foreground class"; a class key counts that class (util/eval.py:23-58):
    a wrong foreground prediction is a false positive of the predicted class and a false negative of the true one."""

    def __init__(self):
        self._tp, self._fp, self._fn = {}, {}, {}

    @staticmethod
    def _inc(d, k):
        d[k] = d.get(k, 0) + 1

    def update(self, true, pred):
        if pred != 0:
            self._inc(self._tp if true != 0 else self._fp, None)
            if pred == true:
                self._inc(self._tp, pred)
            else:
                self._inc(self._fp, pred)
                if true != 0:
                    self._inc(self._fn, true)
        elif true != 0:
            self._inc(self._fn, None)
            self._inc(self._fn, true)

    def tp_fp_fn(self, k):
        return self._tp.get(k, 0), self._fp.get(k, 0), self._fn.get(k, 0)

    def get(self, k):
        tp, fp, fn = self.tp_fp_fn(k)
        denom = tp + 0.5 * fp + 0.5 * fn
        if denom == 0:
            assert tp == 0
            denom = 1
        return tp / denom


def _events_of_video(scores, pred, classes_inv, locations, threshold):
    """argmax events and every (frame, class) pair scoring at least `threshold` (the high-recall list)."""
    events, recall = [], []
    for i in range(pred.shape[0]):
        xy = {} if locations is None else {"xy": locations[i].tolist()}
        if pred[i] != 0:
            events.append({"label": classes_inv[pred[i]], "frame": i, "score": scores[i, pred[i]].item(), **xy})
        for j in classes_inv:
            if scores[i, j] >= threshold:
                recall.append({"label": classes_inv[j], "frame": i, "score": scores[i, j].item(), **xy})
    return events, recall


def _process(dataset, classes, pred_dict, threshold, with_location):
    classes_inv = {v: k for k, v in classes.items()}
    fps = {video: f for video, _, f in dataset.videos}
    err, f1 = ErrorStat(), ForegroundF1()
    pred_events, pred_events_high_recall, pred_scores = [], [], {}
    for video, entry in sorted(pred_dict.items()):
        scores, support = entry[0], entry[1]
        locations = entry[2] if with_location else None
        label = dataset.get_labels(video, with_locations=True)[0] if with_location else dataset.get_labels(video)
        assert np.min(support) > 0, (video, support.tolist())
        scores /= support[:, None]                       # in place, as the reference does: the caller's arrays become means
        pred = np.argmax(scores, axis=1)
        err.update(label, pred)
        pred_scores[video] = scores.tolist()
        for i in range(pred.shape[0]):
            f1.update(label[i], pred[i])
        events, recall = _events_of_video(scores, pred, classes_inv, locations, threshold)
        pred_events.append({"video": video, "events": events, "fps": fps[video]})
        pred_events_high_recall.append({"video": video, "events": recall, "fps": fps[video]})
    return err, f1, pred_events, pred_events_high_recall, pred_scores


def process_frame_predictions(dataset, classes, pred_dict, high_recall_score_threshold=0.01):
    """pred_dict: video -> (score sums (N, C+1), support (N,)) (util/eval.py:61-113)."""
    return _process(dataset, classes, pred_dict, high_recall_score_threshold, False)


def process_frame_predictions_with_location(dataset, classes, pred_dict, high_recall_score_threshold=0.01):
    """pred_dict: video -> (score sums, support, locations (N, 2)); locations are the last window's, not averaged
    (util/eval.py:116-173)."""
    return _process(dataset, classes, pred_dict, high_recall_score_threshold, True)


def non_maximum_supression(pred, window):
    """Keep an event unless another of its label at a different frame within `window` frames scores strictly higher
    (util/eval.py:176-202).  Quadratic per label like the reference; returns copies with `num_events`."""
    out = []
    for video_pred in pred:
        by_label = {}
        for e in video_pred["events"]:
            by_label.setdefault(e["label"], []).append(e)
        events = []
        for group in by_label.values():
            for e1 in group:
                if not any(e1["frame"] != e2["frame"] and abs(e1["frame"] - e2["frame"]) <= window and e1["score"] < e2["score"]
                           for e2 in group):
                    events.append(e1)
        events.sort(key=lambda x: x["frame"])
        new = copy.deepcopy(video_pred)
        new["events"] = events
        new["num_events"] = len(events)
        out.append(new)
    return out
