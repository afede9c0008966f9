"""Shapes of the full-size step that run on the tiled GEMM kernel (stage 4 and the heads), one by one (debug aid; run on
the GPU box):  python tools/bench_gemm_shapes.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spot_amd import lib

F = 16 * 64
SHAPES = [  # (what, M, N, K, calls per step)
    ("s4 conv3 fwd / dgrad 768x768", F * 49, 768, 768, 4), ("s4.b2 conv1 fwd / dgrad 768x768", F * 49, 768, 768, 2),
    ("s4.b1 conv1 dgrad 320<-768", F * 196, 320, 768, 1), ("s4 down dgrad 320<-768", F * 49, 320, 768, 1),
    ("gru gi 4608<-768", F, 4608, 768, 1), ("gru dgi->dfeat 768<-4608", F, 768, 4608, 1),
    ("loc 2304<-768", F, 2304, 768, 1), ("loc 768<-2304", F, 768, 2304, 2), ("loc dfeat 768<-2304", F, 768, 2304, 1),
    ("se s4 192<-768", F, 192, 768, 4), ("se s4 768<-192", F, 768, 192, 4),
]
torch.cuda.init()
tot = 0.0
for what, M, N, K, calls in SHAPES:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    Cc = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(3):
        lib.gemm(A, W, Cc, M, N, K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        lib.gemm(A, W, Cc, M, N, K)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    byt = 2.0 * (M * K + N * K + M * N)
    tot += us * calls
    print("%-34s M=%7d  %7.1f us  %6.0f GB/s  %6.1f TFLOP/s  x%d = %6.0f us" % (what, M, us, byt / us / 1e3, 2.0 * M * N * K / us / 1e6, calls, us * calls), flush=True)
print("total %.2f ms" % (tot / 1e3))
