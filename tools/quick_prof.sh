#!/bin/bash
# Quick per-kernel table of a training step (or of another bench configuration: pass its flags) on the GPU box:
#   gpurun -- 'bash tools/quick_prof.sh [bench flags]'     -> gpurun_out/quick/top.txt
set -e -o pipefail
out=gpurun_out/quick
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o q -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs "$@" > $out/prof.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/quick/prof/**/q_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
# steps traced: 2 warm-up + 5 timed + 3 single steps for the host issue time (training); the forward configurations trace 7
steps = max([int(r["Calls"]) for r in rows if "adamw" in r["Name"]] + [14]) / 2
with open("gpurun_out/quick/top.txt", "w") as o:
    o.write("total kernel ms per step (%d steps traced): %.3f\n" % (steps, tot / steps / 1e6))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        o.write("%8.3f ms/step %6.1f calls/step %9.1f us  %s\n" % (float(r["TotalDurationNs"]) / steps / 1e6, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
rm -f $out/prof/*/*_kernel_trace.csv $out/prof/*_kernel_trace.csv
tail -n 3 $out/prof.log | cut -c1-400
