"""Stage 1's first 1x1 conv backward (12.8 M rows, 64 -> 32 channels): input-gradient GEMM with the BatchNorm-backward prologue
plus stes_wgrad, against the one-launch form (debug aid; run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spot_amd import lib

M, N, K, k0 = 16 * 64 * 112 * 112, 32, 64, 8
bf = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
D, R, W, Q, Q0 = bf(M, K), bf(M, K), bf(N, K), bf(M, N), bf(M, k0)
coef = [torch.randn(K, device="cuda") for _ in range(3)]
Cc, dc, dG = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.empty(M, K, dtype=torch.bfloat16, device="cuda"), torch.empty(M, k0, dtype=torch.bfloat16, device="cuda")
gw = torch.zeros(K, N, device="cuda")


def two():
    lib.gemm(D, W, Cc, M, N, K, bn_bwd=(R, coef, dc), C0=dG, ldc0=k0, n_split=k0)
    lib.wgrad(dc, Q, gw, M, K, N, Q0=Q0, ldq0=k0, k0=k0, perm_fold=k0)


def one():
    lib.gemm(D, W, Cc, M, N, K, bn_bwd=(R, coef, None), C0=dG, ldc0=k0, n_split=k0, wgrad_fused=(Q, gw, Q0, k0, k0, k0))


for name, fn in (("two launches", two), ("one launch", one), ("two launches", two), ("one launch", one)):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-14s %8.1f us" % (name, e0.elapsed_time(e1) * 100), flush=True)
