"""Run-to-run spread of the gradient of ONE training step at the benchmark's size (16 clips x 64 frames x 224 x 224), and the same
step twice in the library's deterministic mode (every float accumulation in a fixed order): cosine between runs per parameter
group, and whether the deterministic runs agree bit for bit.

    python tools/grad_noise.py [clips=16]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from spot_amd import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bench.B_PER_GPU = B
m = bench.make_model("cuda:0", seed=0)
m._model._pred_fine._dropout.p = 0.0
eng = m._model.engine
eng.drop_p = 0.0
x, label, xy = bench.make_batch("cuda:0", 0)
sd = {k: v.clone() for k, v in m._model.state_dict().items()}

def run():
    m._model.load_state_dict(sd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.train_step(x, label, xy, lr=0.0, optimizer_step=False)
    torch.cuda.synchronize()
    return eng.flat_grad.clone(), out[:3].clone(), time.perf_counter() - t0

def cos(a, b):
    a, b = a.double(), b.double()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))

def by_group(a, b):
    out = {}
    for name, (lo, hi) in eng.buckets.items():
        out[name] = round(cos(a[lo:hi], b[lo:hi]), 5)
    return out

run()
a, b = run(), run()
print("ordinary runs: gradient cosine %.5f, by bucket %s, loss %.6f / %.6f, %.1f ms" % (cos(a[0], b[0]), by_group(a[0], b[0]), a[1][0], b[1][0], 1e3 * b[2]))
worst = []
for name, p in m._model.named_parameters():
    o, n, _ = eng._goff[id(p)]
    if n >= 64:
        worst.append((cos(a[0][o:o + n], b[0][o:o + n]), name))
worst.sort()
print("lowest per-tensor cosines between two ordinary runs:", [(round(c, 4), n) for c, n in worst[:5]], " median %.5f" % worst[len(worst) // 2][0])
lib.set_deterministic(True)
try:
    c, d = run(), run()
    flag = int(lib.deterministic_ticket()[1])
finally:
    lib.set_deterministic(False)
same = torch.equal(c[0].view(torch.int32), d[0].view(torch.int32))
print("deterministic runs: bit-identical gradients %s (ticket flag %d), loss %.6f / %.6f, %.1f ms per step; cosine with an ordinary run %.5f" % (
    same, flag, c[1][0], d[1][0], 1e3 * d[2], cos(c[0], a[0])))
