#!/usr/bin/env python3
"""Rebuild the files under profiles/ from the rocprofv3 output a GPU run left under gpurun_out/
(prof_e: the step as benchmarked; prof_es: --serial; pmc_fetch_e / pmc_write_e: counter passes; r1_e_bench.json)."""
import json
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
d = json.loads(open("gpurun_out/r1_e_bench.json").read().strip().splitlines()[-1])
open("profiles/r1_e_bench.json", "w").write(json.dumps(d, indent=1) + "\n")
r = d["roofline"]
shutil.copy("gpurun_out/prof_e/r1e_kernel_stats.csv", "profiles/r1_e_kernel_stats.csv")
shutil.copy("gpurun_out/prof_es/r1es_kernel_stats.csv", "profiles/r1_e_serial_kernel_stats.csv")
with open("profiles/pmc_traffic.json", "w") as fp:
    subprocess.run([sys.executable, "tools/pmc_traffic.py", "gpurun_out/pmc_fetch_e/r1_counter_collection.csv",
                    "gpurun_out/pmc_write_e/r1_counter_collection.csv"], stdout=fp, check=True)
tr = json.load(open("profiles/pmc_traffic.json")).get(r["kernel"], {})
cmd = "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
note_s = (f"Command: `{cmd} --serial` (4 steps traced). This is the mode bench.py's roofline pass measures in (each kernel alone on "
          f"the GPU): its HIP-event average for the dominant kernel `{r['kernel']}` was {r['avg_launch_us']:.1f} us per launch (this "
          f"trace: the row below). HBM traffic of every kernel from counter passes of the same --serial command: "
          f"`profiles/pmc_traffic.json` ({r['kernel']}: 2 x FETCH_SIZE + WRITE_SIZE = {tr.get('hbm_bytes_per_launch', 0) / 1e6:.0f} MB "
          f"per launch against {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB algorithmic; "
          f"{tr.get('write_size_bytes_per_launch', 0) / 1e6:.0f} MB of it written).")
note_o = (f"Command: `{cmd}` (4 steps traced). bench.py un-profiled: {d['ms_per_step']:.1f} ms/step = {d['value']:.0f} frames/s "
          f"(`profiles/r1_e_bench.json`). Kernels of the two streams run concurrently here, so a kernel's duration includes the time "
          f"it shares the GPU with another one and the column sums to more than the step time; durations of kernels running alone "
          f"are in `profiles/r1_e_serial_kernel_stats.md`.")
for src, dst, title, note in (
        ("gpurun_out/prof_es/r1es_kernel_stats.csv", "profiles/r1_e_serial_kernel_stats.md",
         "Profile (e), serial launches: full-size training step, weight-gradient kernels in program order on the main stream "
         "(B=16, T=64, 224p, 1 MI355X)", note_s),
        ("gpurun_out/prof_e/r1e_kernel_stats.csv", "profiles/r1_e_kernel_stats.md",
         "Profile (e): full-size training step as benchmarked, weight-gradient kernels on a side stream beside the input-gradient "
         "chain (B=16, T=64, 224p, 1 MI355X)", note_o)):
    with open(dst, "w") as fp:
        subprocess.run([sys.executable, "tools/prof_summary.py", src, "--steps", "4", "--title", title, "--note", note], stdout=fp, check=True)
print(d["ms_per_step"], d["value"], r["kernel"], r["frac"], r["traffic"], r["avg_launch_us"])
