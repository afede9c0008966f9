"""Which host-side operations of a training step launch device copy / fill kernels (torch.profiler, one step)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
from spot_amd import dp  # noqa: E402

dev = torch.device("cuda:0")
model = bench.make_model(dev)
x, label, xy = bench.make_batch(dev, 0)
trainer = dp.DataParallelTrainer(model, None, lr=1e-3)
for _ in range(3):
    trainer.step(x, label, xy)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.step(x, label, xy)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=60))
ev = [e for e in prof.events() if "copy" in e.name.lower() or "memcpy" in e.name.lower() or "memset" in e.name.lower()]
from collections import Counter
print(Counter(e.name for e in ev).most_common(10))
stacks = Counter()
for e in ev:
    if e.stack:
        stacks[" <- ".join(s for s in e.stack[:4])] += 1
for k, v in stacks.most_common(8):
    print(v, k)
