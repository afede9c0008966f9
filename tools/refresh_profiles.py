#!/usr/bin/env python3
"""Rebuild the round's files under profiles/ from the rocprofv3 output a GPU run left under gpurun_out/<tag>/ (tools/profile_round.sh):
prof (the step as benchmarked), prof_serial (--serial), pmc_fetch / pmc_write (counter passes of the --serial command), bench*.json.

    python tools/refresh_profiles.py r3
"""
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(root)
src = f"gpurun_out/{tag}"
STEPS = 7            # traced steps of the profiled command: 1 warm-up + 3 timed + 3 single steps for the host issue time


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


d = last_json(f"{src}/bench.json")
open(f"profiles/{tag}_bench.json", "w").write(json.dumps(d, indent=1) + "\n")
for cfg in ("train_tsm_mingru", "infer_fwd", "infer_long_video"):
    if os.path.exists(f"{src}/bench_{cfg}.json"):
        open(f"profiles/{tag}_bench_{cfg}.json", "w").write(json.dumps(last_json(f"{src}/bench_{cfg}.json"), indent=1) + "\n")
r = d["roofline"]
shutil.copy(f"{src}/prof/{tag}_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
shutil.copy(f"{src}/prof_serial/{tag}s_kernel_stats.csv", f"profiles/{tag}_serial_kernel_stats.csv")
with open("profiles/pmc_traffic.json", "w") as fp:
    subprocess.run([sys.executable, "tools/pmc_traffic.py", f"{src}/pmc_fetch/{tag}_counter_collection.csv",
                    f"{src}/pmc_write/{tag}_counter_collection.csv"], stdout=fp, check=True)
tr = json.load(open("profiles/pmc_traffic.json")).get(r["kernel"], {})
cmd = "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs"
note_s = (f"Command: `{cmd} --serial` ({STEPS} steps traced). This is the mode bench.py's roofline pass measures in (each kernel alone on "
          f"the GPU): its HIP-event average for the dominant kernel `{r['kernel']}` was {r['avg_launch_us']:.1f} us per launch (this "
          f"trace: the row below). HBM traffic of every kernel from counter passes of the same --serial command: "
          f"`profiles/pmc_traffic.json` ({r['kernel']}: 2 x FETCH_SIZE + WRITE_SIZE = {tr.get('hbm_bytes_per_launch', 0) / 1e6:.0f} MB "
          f"per launch against {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB algorithmic; "
          f"{tr.get('write_size_bytes_per_launch', 0) / 1e6:.0f} MB of it written).")
note_o = (f"Command: `{cmd}` ({STEPS} steps traced). bench.py un-profiled: {d['ms_per_step']:.1f} ms/step = {d['value']:.0f} frames/s "
          f"(`profiles/{tag}_bench.json`). Kernels of the two streams run concurrently here, so a kernel's duration includes the time "
          f"it shares the GPU with another one and the column sums to more than the step time; durations of kernels running alone "
          f"are in `profiles/{tag}_serial_kernel_stats.md`.")
for s, dst, title, note in (
        (f"{src}/prof_serial/{tag}s_kernel_stats.csv", f"profiles/{tag}_serial_kernel_stats.md",
         "Serial launches: full-size training step, weight-gradient kernels in program order on the main stream "
         "(B=16, T=64, 224p, 1 MI355X)", note_s),
        (f"{src}/prof/{tag}_kernel_stats.csv", f"profiles/{tag}_kernel_stats.md",
         "Full-size training step as benchmarked, weight-gradient kernels on a side stream beside the input-gradient "
         "chain (B=16, T=64, 224p, 1 MI355X)", note_o)):
    with open(dst, "w") as fp:
        subprocess.run([sys.executable, "tools/prof_summary.py", s, "--steps", str(STEPS), "--title", title, "--note", note], stdout=fp, check=True)
if os.path.exists(f"{src}/prof_infer/{tag}i_kernel_stats.csv"):
    di = last_json(f"{src}/bench_infer_fwd.json")
    with open(f"profiles/{tag}_infer_fwd_kernel_stats.md", "w") as fp:
        subprocess.run([sys.executable, "tools/prof_summary.py", f"{src}/prof_infer/{tag}i_kernel_stats.csv", "--steps", "7", "--title",
                        "Evaluation forward (predict) of 16 clips x 64 frames x 224p, 1 MI355X", "--note",
                        "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --config infer_fwd --steps 5 --warmup 2 --no-cpu-baseline "
                        f"--no-roofline` (7 forward passes traced; model construction contributes the copy / fill kernels at the bottom). "
                        f"bench.py un-profiled: {di['ms_per_step']:.2f} ms per pass = {di['value']:.0f} frames/s (`profiles/{tag}_bench_infer_fwd.json`)."],
                       stdout=fp, check=True)
print(d["ms_per_step"], d["value"], r["kernel"], r["frac"], r["traffic"], r["avg_launch_us"])
