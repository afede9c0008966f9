"""Per-kernel, per-launch-shape durations of the last traced training step of a rocprofv3 --kernel-trace CSV (debug aid):
    python tools/step_shapes.py gpurun_out/<dir>/<name>_kernel_trace.csv [rows]"""
import collections
import csv
import re
import sys


def short(n):
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", n)
    if m:
        k = int(m.group(1))
        rest = n[len(m.group(0)):]
        name, tail = rest[:k], rest[k:]
        t = re.match(r"I((?:L[ib]\d+E)+)E", tail)
        if t:
            name += "<" + ", ".join(re.findall(r"L[ib](\d+)E", t.group(1))) + ">"
        return name
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [short(r["Kernel_Name"]) for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("adamw")]
s, e = idx[-3] + 1, idx[-1] + 1
g = collections.OrderedDict()
for i in range(s, e):
    r = rows[i]
    key = (names[i], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["VGPR_Count"], r["LDS_Block_Size"])
    g.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
items = sorted(g.items(), key=lambda kv: -sum(kv[1]))
for (n, grid, vg, lds), v in items[:int(sys.argv[2]) if len(sys.argv) > 2 else 80]:
    print("%7.1f us  %3d x %7.1f  grid %6d vgpr %3s lds %6s  %s" % (sum(v), len(v), sum(v) / len(v), grid, vg, lds, n))
print("step: %d launches, %.2f ms of kernel time, %.2f ms wall" % (e - s, sum(sum(v) for v in g.values()) / 1e3,
      (int(rows[e - 1]["End_Timestamp"]) - int(rows[s]["Start_Timestamp"])) / 1e6))
