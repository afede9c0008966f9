"""Score a prediction file against ground truth: temporal mAP tables before and after windowed peak picking.

Counterpart of the reference's stand-alone evaluator (eval.py:13-78), for the files the training driver and
`spot_amd.evaluate` write (`pred-<split>.<epoch>.recall.json.gz`, or a plain `.json`):

    python -m spot_amd.eval PRED_FILE_OR_MODEL_DIR -s val [-d DATASET | --truth TRUTH.json] [--nms_window 1] [-t 0 1 2 4]

A model directory is searched for the prediction file of the latest test epoch like the reference does (:27-55).  The truth
file is `data/<dataset>/<split>.json` unless `--truth` names it.
"""
from __future__ import annotations

import argparse
import gzip
import json
import os
import re

from .util.eval import non_maximum_supression
from .util.score import compute_mAPs


def get_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("pred_file", help="Path to predictions or model dirs")
    p.add_argument("-d", "--dataset", type=str)
    p.add_argument("-s", "--split", type=str, required=True, choices=["train", "test", "val"])
    p.add_argument("--nms_window", type=int, default=1)
    p.add_argument("-t", "--tolerances", type=int, nargs="+")
    p.add_argument("--truth", type=str, default=None, help="ground-truth label JSON (default data/<dataset>/<split>.json)")
    return p.parse_args(argv)


def load_json(path):
    if path.endswith(".gz"):
        with gzip.open(path, "rt", encoding="ascii") as fp:
            return json.load(fp)
    with open(path) as fp:
        return json.load(fp)


def get_pred_file(pred_dir, split):
    """The recall file of the highest epoch of `split` in a model directory (eval.py:27-38)."""
    regex = re.compile(r"pred-{}\.(\d+)\.recall\.json\.gz".format(split))
    found = [(os.path.join(pred_dir, name), int(m.group(1))) for name in os.listdir(pred_dir) for m in [regex.match(name)] if m]
    if not found:
        raise FileNotFoundError("No suitable prediction file!")
    return max(found, key=lambda x: x[1])


def main(dataset, pred_file, split, nms_window, tolerances, truth=None):
    if os.path.isdir(pred_file):
        if dataset is None and truth is None:
            dataset = load_json(os.path.join(pred_file, "config.json"))["dataset"]
            print("Inferred dataset:", dataset)
        if split != "test":
            _, epoch = get_pred_file(pred_file, "test")
            pred_file = os.path.join(pred_file, "pred-{}.{}.recall.json.gz".format(split, epoch))
        else:
            pred_file, _ = get_pred_file(pred_file, split)
        print("Evaluating on: {}".format(pred_file))
    else:
        assert dataset is not None or truth is not None, "Dataset is required!"
    pred = load_json(pred_file)
    truth_data = load_json(truth if truth is not None else os.path.join("data", dataset, "{}.json".format(split)))
    kwargs = {} if tolerances is None else {"tolerances": tolerances}
    print("\n=== Results on {} (w/o NMS) ===".format(split))
    no_nms_result = compute_mAPs(truth_data, pred, **kwargs)
    pred = non_maximum_supression(pred, nms_window)
    print("\n=== Results on {} (w/ NMS) ===".format(split))
    nms_result = compute_mAPs(truth_data, pred, **kwargs)
    return no_nms_result, nms_result


if __name__ == "__main__":
    main(**vars(get_args()))
