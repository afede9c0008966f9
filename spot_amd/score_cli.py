"""Re-score a saved prediction file: event mAP before and after windowed peak picking.

    python -m spot_amd.score_cli PRED.json[.gz] | MODEL_DIR  -s val  [--truth TRUTH.json | -d DATASET]  [--nms_window 1] [-t 0 1 2 4]

Reads what `spot_amd.train` / `spot_amd.evaluate` write (`pred-<split>.<epoch>.recall.json.gz`, the reference's schema), so it
covers what the reference's stand-alone scorer does (eval.py:13-78) on top of this package's own metric code
(`spot_amd.util.score`, `spot_amd.util.eval`); the reference's script reads the same files unchanged.  For a model directory the
file of the newest *test* epoch decides the epoch, as the reference does (eval.py:41-55).
"""
from __future__ import annotations

import argparse
import glob
import gzip
import json
import os
import re

from .util.eval import non_maximum_supression
from .util.score import compute_mAPs


def read_json(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as fp:
        return json.load(fp)


def newest_epoch(model_dir, split):
    """(path, epoch) of the recall file with the largest epoch number for `split`; FileNotFoundError if there is none."""
    best = None
    for path in glob.glob(os.path.join(model_dir, f"pred-{split}.*.recall.json.gz")):
        m = re.fullmatch(rf"pred-{split}\.(\d+)\.recall\.json\.gz", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[1]):
            best = (path, int(m.group(1)))
    if best is None:
        raise FileNotFoundError(f"no pred-{split}.<epoch>.recall.json.gz in {model_dir}")
    return best


def resolve(pred, split, dataset=None):
    """Prediction file and dataset name for a file or a model directory."""
    if not os.path.isdir(pred):
        return pred, dataset
    if dataset is None and os.path.exists(os.path.join(pred, "config.json")):
        dataset = read_json(os.path.join(pred, "config.json")).get("dataset")
    path, epoch = newest_epoch(pred, "test")
    if split != "test":
        path = os.path.join(pred, f"pred-{split}.{epoch}.recall.json.gz")
    return path, dataset


def score(pred, truth, nms_window=1, tolerances=None, quiet=False):
    """{'raw': compute_mAPs(...), 'nms': compute_mAPs(after peak picking within nms_window frames)}."""
    kw = {} if tolerances is None else {"tolerances": list(tolerances)}
    out = {}
    for name, p in (("raw", pred), ("nms", non_maximum_supression(pred, nms_window))):
        if not quiet:
            print(f"\n=== {name}: {'no peak picking' if name == 'raw' else f'peak picking, window {nms_window}'} ===")
        out[name] = compute_mAPs(truth, p, quiet=quiet, **kw)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("pred", help="prediction file or model directory")
    ap.add_argument("-s", "--split", required=True, choices=["train", "val", "test"])
    ap.add_argument("-d", "--dataset")
    ap.add_argument("--truth", help="ground-truth JSON (default data/<dataset>/<split>.json)")
    ap.add_argument("--nms_window", type=int, default=1)
    ap.add_argument("-t", "--tolerances", type=int, nargs="+")
    a = ap.parse_args(argv)
    path, dataset = resolve(a.pred, a.split, a.dataset)
    truth_path = a.truth or (os.path.join("data", dataset, f"{a.split}.json") if dataset else None)
    if truth_path is None:
        ap.error("give --truth or -d DATASET (or a model directory whose config.json names the dataset)")
    print("scoring", path, "against", truth_path)
    return score(read_json(path), read_json(truth_path), a.nms_window, a.tolerances)


if __name__ == "__main__":
    main()
