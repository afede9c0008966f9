#!/usr/bin/env python3
"""Per-launch listing of the last training step in a rocprofv3 kernel trace (csv): offset in the step, duration, kernel.
    python tools/step_trace.py gpurun_out/x/prof_kernel_trace.csv [min_us]"""
import csv
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import short

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "cast_bf16_multi" in r["Kernel_Name"]]
step = rows[idx[-2]:idx[-1]]
t0 = int(step[0]["Start_Timestamp"])
tot = 0.0
agg = {}
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    k = short(r["Kernel_Name"])
    a = agg.setdefault(k, [0.0, 0])
    a[0] += d
    a[1] += 1
    if d > min_us:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:8.3f} {d:8.1f} {k[:56]:56s} grid={int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])} wg={r['Workgroup_Size_X']} "
              f"vgpr={r['VGPR_Count']}+{r['Accum_VGPR_Count']} lds={r['LDS_Block_Size']} scr={r['Scratch_Size']}")
print("total kernel ms", tot / 1e3, "span ms", (int(step[-1]["End_Timestamp"]) - t0) / 1e6, "launches", len(step))
print("--- by kernel (ms, launches)")
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{t / 1e3:7.3f} {n:4d}  {k}")
