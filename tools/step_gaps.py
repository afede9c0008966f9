"""Every launch shape of one full-size training step with its algorithmic bytes, HIP-event duration, GB/s and the time it
would take at 5.5 TB/s, largest gap first (debug aid; run on the GPU box):  python tools/step_gaps.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from spot_amd import dp, engine, lib

model = bench.make_model("cuda:0")
x, label, xy = bench.make_batch("cuda:0", 0)
tr = dp.DataParallelTrainer(model, None, lr=1e-3)
for _ in range(2):
    tr.step(x, label, xy)
engine.OVERLAP = False
lib.TIMING = []
STEPS = 3
for _ in range(STEPS):
    tr.step(x, label, xy)
torch.cuda.synchronize()
recs, lib.TIMING = lib.TIMING, None
g = {}
for k, e0, e1, byt, fl in recs:
    a = g.setdefault((k, byt, fl), [0.0, 0])
    a[0] += e0.elapsed_time(e1) * 1e3
    a[1] += 1
rows = []
for (k, byt, fl), (us, n) in g.items():
    us /= STEPS
    n /= STEPS
    floor = (byt or 0) / 5.5e6 * n if byt else 0.0           # us at 5.5 TB/s
    mf = fl / 1.2e9 * n if fl else 0.0                      # us at 1.2 PFLOP/s
    floor = max(floor, mf)
    rows.append((us - floor, us, n, byt, fl, k, floor))
rows.sort(reverse=True)
tot = sum(r[1] for r in rows)
print("gap_us  total_us  n  avg_us  MB  GB/s  TF/s  kernel")
for gap, us, n, byt, fl, k, floor in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 70]:
    print("%7.0f %8.0f %4.0f %7.1f %8.1f %6.0f %5.0f  %s" % (gap, us, n, us / n, (byt or 0) / 1e6, (byt or 0) * n / us / 1e3 if byt else 0,
                                                          fl * n / us / 1e6 if fl else 0, k))
print("step kernel time %.2f ms; sum of floors %.2f ms" % (tot / 1e3, sum(r[6] for r in rows) / 1e3))
