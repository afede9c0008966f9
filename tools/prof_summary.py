#!/usr/bin/env python3
"""Turn a rocprofv3 `*_kernel_stats.csv` into the markdown table kept under profiles/.

    python tools/prof_summary.py gpurun_out/prof/x_kernel_stats.csv --steps 4 --title "..." --note "..." > profiles/NAME.md
"""
import argparse
import csv
import re


def short(name):
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:       # anonymous-namespace kernel with __bf16 parameters, which llvm-cxxfilt leaves mangled
        n = int(m.group(1))
        base, rest = name[m.end():m.end() + n], name[m.end() + n:]
        t = re.match(r"I((?:L[ib]\d+E)+)E", rest)
        if not t:
            return base
        args = [v if k == "i" else ("true" if v == "1" else "false") for k, v in re.findall(r"L([ib])(\d+)E", t.group(1))]
        return base + "<" + ", ".join(args) + ">"
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps", type=float, required=True, help="training steps covered by the trace (warm-up included)")
    ap.add_argument("--title", default="rocprofv3 --kernel-trace --stats")
    ap.add_argument("--note", default="")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {a.title}\n")
    if a.note:
        print(a.note + "\n")
    print(f"Sum of kernel time: {tot / a.steps / 1e6:.2f} ms/step over {a.steps:g} traced steps.\n")
    print("| ms/step | % | calls/step | avg us | kernel |\n|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[: a.top]:
        t = float(r["TotalDurationNs"])
        print(f"| {t / a.steps / 1e6:.2f} | {100 * t / tot:.2f} | {int(r['Calls']) / a.steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | `{short(r['Name'])}` |")


if __name__ == "__main__":
    main()
