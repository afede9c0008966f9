"""Does running a producer -> consumer pair on sub-batches (so the intermediate stays in the 256 MiB Infinity Cache) pay?
Stage-3 conv3 backward: BatchNorm-backward GEMM (writes dc3, dA3) followed by the weight gradient (reads dc3, c2) and the
squeeze-excite pooling pass (reads c2, dA3).  Inputs rotate over several buffer sets so that they come from HBM.
(debug aid; run on the GPU box):  python tools/bench_mall.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spot_amd import lib

BF = torch.bfloat16
NIMG, HW, C = 1024, 196, 320
M = NIMG * HW
NSET = 3
dZ = [torch.randn(M, C, device="cuda").to(BF) for _ in range(NSET)]
c3 = [torch.randn(M, C, device="cuda").to(BF) for _ in range(NSET)]
c2 = [torch.randn(M, C, device="cuda").to(BF) for _ in range(NSET)]
dA3 = [torch.empty(M, C, dtype=BF, device="cuda") for _ in range(NSET)]
dc3_full = [torch.empty(M, C, dtype=BF, device="cuda") for _ in range(NSET)]
W = torch.randn(C, C, device="cuda").to(BF)
coef = (torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1)
sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
gate = torch.rand(NIMG, C, device="cuda")
dW = torch.zeros(C, C, device="cuda")
S4 = torch.zeros(4 * NIMG * C, device="cuda")
dgate = torch.zeros(NIMG * C, device="cuda")
dpre2 = torch.zeros(NIMG * C, dtype=BF, device="cuda")
p = lib.ptr


def chain(i, nsub, pool=True):
    s = i % NSET
    rows = M // nsub
    img = NIMG // nsub
    for j in range(nsub):
        r0 = j * rows
        a, r, q, o = dZ[s][r0:r0 + rows], c3[s][r0:r0 + rows], c2[s][r0:r0 + rows], dA3[s][r0:r0 + rows]
        # the intermediate: one small buffer reused by every sub-batch when nsub > 1
        d = dc3_full[s][r0:r0 + rows] if nsub == 1 else dc3_full[0][:rows]
        lib.gemm(a, W, o, rows, C, C, bn_bwd=(r, coef, d))
        lib.wgrad(d, q, dW, rows, C, C, scale=sc, shift=sh, relu=True, gate=gate[j * img:(j + 1) * img], gate_group=HW)
        if pool:
            lib.call("stes_se_bwd_pool", p(q), p(sc), p(sh), p(o), p(S4[4 * j * img * C:]), p(dgate[j * img * C:]), p(gate[j * img:]),
                     p(dpre2[j * img * C:]), img, HW, C)


def run(name, fn, reps=12):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    print("%-40s %8.1f us" % (name, e0.elapsed_time(e1) * 1e3 / reps), flush=True)


for nsub in (1, 2, 4, 8, 16):
    run("gemm+wgrad+pool, %2d sub-batches" % nsub, lambda i: chain(i, nsub))
for nsub in (1, 4, 8):
    run("gemm+wgrad, %2d sub-batches" % nsub, lambda i: chain(i, nsub, pool=False))
