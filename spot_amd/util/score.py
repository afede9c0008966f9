"""Event-spotting metrics on prediction / ground-truth event lists (host side, pure Python + numpy).

Same definitions and the same floating-point evaluation order as the reference's util/score.py, so an identical event list
gives identical numbers: temporal average precision per class at a frame tolerance (:37-94), spatial average precision
over all classes at a frame and a pixel tolerance (:97-179), their tables over tolerances (`compute_mAPs` :182-243,
`compute_mAPs_with_locations` :303-431).  Event lists have the reference's JSON shape:
    [{"video": str, "events": [{"label": str, "frame": int, "score": float, "xy": [x, y]}, ...]}, ...]
Plotting is left out; tables are printed with plain formatting.
"""
from __future__ import annotations

import copy
import sys

import numpy as np

from .postproc import filter_events_by_score, non_max_suppression_events


def _by_label_and_video(truth):
    """label -> video -> ground-truth entries in file order; an entry is a frame, or (frame, xy) when the event has a
    location (util/score.py:9-21)."""
    out = {}
    for v in truth:
        for e in v["events"]:
            entry = (e["frame"], e["xy"]) if "xy" in e else e["frame"]
            out.setdefault(e["label"], {}).setdefault(v["video"], []).append(entry)
    return out


def _ranked(pred, label=None):
    """(video, frame, score, xy) of the predictions of one label (all when None), best score first; the sort is stable, so
    ties keep file order (util/score.py:24-32)."""
    rows = [(v["video"], e["frame"], e["score"], e.get("xy")) for v in pred for e in v["events"] if label is None or e["label"] == label]
    rows.sort(key=lambda r: -r[2])
    return rows


def _frame_of(entry):
    return entry[0] if isinstance(entry, tuple) else entry


def _area_under_interpolated_precision(precisions, total):
    """Precision recorded at each true positive -> running maximum from the right -> sum / number of ground-truth events
    (the curve has one bin per ground-truth event; unreached bins count zero)."""
    best, interp = 0.0, []
    for p in reversed(precisions):
        best = p if p > best else best
        interp.append(best)
    interp.reverse()
    return sum(interp) / total


def _match_walk(ranked, truth_by_video, accept):
    """Walk the ranked predictions; each takes the closest not-yet-matched ground-truth event of its video (first one on
    ties) and is a true positive when accept(prediction, ground truth entry) holds.  Returns the precision at every true
    positive."""
    matched, precisions, last = set(), [], 1
    for rank, row in enumerate(ranked, 1):
        video, frame, score = row[0], row[1], row[2]
        assert score <= last, "predictions must be sorted by score"
        last = score
        best = None
        for entry in truth_by_video.get(video, ()):
            f = _frame_of(entry)
            if (video, f) in matched:
                continue
            if best is None or abs(frame - _frame_of(best)) > abs(frame - f):
                best = entry
        if best is not None and accept(row, best):
            matched.add((video, _frame_of(best)))
            precisions.append(len(matched) / rank)
    return precisions


def average_precision(ranked, truth_by_video, tolerance=0):
    """Temporal AP of one class (util/score.py:37-94)."""
    total = sum(len(v) for v in truth_by_video.values())
    hits = _match_walk(ranked, truth_by_video, lambda row, gt: abs(row[1] - _frame_of(gt)) <= tolerance)
    return _area_under_interpolated_precision(hits, total)


def average_precision_with_locations(ranked, truth_by_label, tolerance_t=0, tolerance_p=20):
    """Spatial AP over the events of all classes pooled per video: a hit needs the frame within tolerance_t and the
    truncated pixel distance within tolerance_p (util/score.py:97-179)."""
    pooled = {}
    for per_video in truth_by_label.values():
        for video, entries in per_video.items():
            pooled.setdefault(video, []).extend(entries)
    total = sum(len(v) for v in pooled.values())

    def accept(row, gt):
        return abs(row[1] - gt[0]) <= tolerance_t and int(np.linalg.norm(np.subtract(row[3], gt[1]))) <= tolerance_p
    return _area_under_interpolated_precision(_match_walk(ranked, pooled, accept), total)


def _mean(xs):
    return np.mean(xs)


def _print_table(title, columns, rows):
    print(title.ljust(22) + "".join(str(c).rjust(9) for c in columns))
    for name, vals in rows:
        print(str(name).ljust(22) + "".join(("%.2f" % (100 * v)).rjust(9) for v in vals))


def compute_mAPs(truth, pred, tolerances=None, quiet=False):
    """util/score.py:182-243 -> (mAP per tolerance, tolerances)."""
    tolerances = list(range(6)) if tolerances is None else list(tolerances)
    assert {v["video"] for v in truth} == {v["video"] for v in pred}, "Video set mismatch!"
    by_label = _by_label_and_video(truth)
    labels = sorted(by_label)
    per_tol = [[average_precision(_ranked(pred, lab), by_label[lab], tol) for lab in labels] for tol in tolerances]
    mAPs = [_mean(aps) for aps in per_tol]
    if not quiet:
        _print_table("AP @ tol", tolerances, [(lab, [aps[i] for aps in per_tol]) for i, lab in enumerate(labels)] + [("mAP", mAPs)])
        print("Avg mAP (across tolerances): {:0.2f}".format(_mean(mAPs) * 100))
        sys.stdout.flush()
    return mAPs, tolerances


def _scale_xy(data, px_scale):
    for v in data:
        for e in v["events"]:
            e["xy"] = [c * px_scale for c in e["xy"]]
    return data


def compute_mAPs_with_locations(truth, pred, tolerances_t=(0, 1, 2, 4), tolerances_p=(2, 4, 6), fg_threshold=0.25, px_scale=224,
                                quiet=False):
    """util/score.py:303-431: predictions are score-filtered, scaled to pixels and peak-picked within 3 frames first.
    Returns (temporal mAP per frame tolerance, spatial AP for every (frame tolerance, pixel tolerance) pair, row-major)."""
    truth, pred = copy.deepcopy(truth), copy.deepcopy(pred)
    pred = non_max_suppression_events(_scale_xy(filter_events_by_score(pred, fg_threshold), px_scale), 3)
    truth = _scale_xy(truth, px_scale)
    assert {v["video"] for v in truth} == {v["video"] for v in pred}, "Video set mismatch!"
    by_label = _by_label_and_video(truth)
    labels = sorted(by_label)
    temporal, spatial = [], []
    for tol_t in tolerances_t:
        temporal.append([average_precision(_ranked(pred, lab), by_label[lab], tol_t) for lab in labels])
        spatial.append([average_precision_with_locations(_ranked(pred), by_label, tol_t, tol_p) for tol_p in tolerances_p])
    mAPs_t = [_mean(aps) for aps in temporal]
    mAP_ps = [ap for row in spatial for ap in row]
    if not quiet:
        _print_table("Temporal AP @ tol", tolerances_t, [(lab, [aps[i] for aps in temporal]) for i, lab in enumerate(labels)] + [("mAP", mAPs_t)])
        print("Avg Class mAP (across tolerances): {:0.2f}".format(_mean(mAPs_t) * 100))
        _print_table("Spatial AP @ tol_p", tolerances_p, [("tol_t=%d" % t, row) for t, row in zip(tolerances_t, spatial)])
        print("Avg Spatial mAP (across tolerances): {:0.2f}".format(_mean(mAP_ps) * 100))
        sys.stdout.flush()
    return mAPs_t, mAP_ps
