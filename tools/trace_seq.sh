#!/bin/bash
# Kernel sequence of one bench step in launch order (debug aid, GPU box):  bash tools/trace_seq.sh [bench flags] -> gpurun_out/seq.txt
set -e -o pipefail
out=gpurun_out/seq
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -o q -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs "$@" > $out/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/seq/**/q_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
with open("gpurun_out/seq.txt", "w") as o:
    for r in rows[-900:]:
        o.write("%8.1f us  q%s  %s\n" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
PY
rm -rf $out
