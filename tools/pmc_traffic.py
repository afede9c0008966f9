#!/usr/bin/env python3
"""HBM traffic per launch of every kernel from two rocprofv3 counter runs (one per counter: a pass holds either).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o r1 -- python3 bench.py --steps 1 --warmup 1 ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o r1 -- python3 bench.py --steps 1 --warmup 1 ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/r1_counter_collection.csv gpurun_out/pmc_write/r1_counter_collection.csv > profiles/pmc_traffic.json

Both counters are in KiB.  On gfx950 FETCH_SIZE counts a wide (16 bytes per lane) coalesced read at half its bytes, and
every kernel here reads that way, so reads = 2 * FETCH_SIZE; WRITE_SIZE is exact for 16-byte stores and float atomics.
"""
import collections
import csv
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from prof_summary import short  # noqa: E402


def agg(path):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        d[k][0] += float(r["Counter_Value"])
        d[k][1] += 1
    return d


def main():
    f, w = agg(sys.argv[1]), agg(sys.argv[2])
    out = {}
    for k in sorted(f):
        if k not in w or f[k][1] != w[k][1]:
            continue
        n = f[k][1]
        fetch, write = f[k][0] / n * 1024, w[k][0] / n * 1024
        out[k] = {"launches": n, "fetch_size_bytes_per_launch": fetch, "write_size_bytes_per_launch": write,
                  "hbm_bytes_per_launch": 2 * fetch + write}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
