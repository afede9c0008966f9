#!/usr/bin/env python3
"""Times stes_block_front_eval (conv1 + bn/act + grouped conv2 + squeeze of a bottleneck, evaluation mode) on the shapes of a
1,024-frame forward at 224 x 224 against the two kernels it replaces (stes_gemm, stes_conv3x3g_fwd_pool).
    gpurun -- 'python tools/bench_block_front.py'"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spot_amd import lib          # noqa: E402

BF = torch.bfloat16


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    N = 1024
    g = torch.Generator().manual_seed(0)
    for cin, cmid, wd, fold, stride in ((32, 64, 112, 8, 2), (64, 128, 56, 16, 2), (128, 128, 28, 32, 1), (128, 320, 28, 32, 2)):
        M = N * wd * wd
        Mo = M // (stride * stride)
        X = torch.randn(M, cin, generator=g).abs().to(BF).cuda()
        G = torch.randn(M, fold, generator=g).abs().to(BF).cuda()
        W1 = (torch.randn(cmid, cin, generator=g) / cin ** 0.5).to(BF).cuda()
        w2 = (torch.randn(cmid, 16, 3, 3, generator=g) / 12.0).cuda()
        wpf = torch.empty(cmid // 16 * 5 * 512, dtype=BF, device="cuda")
        wpd = torch.empty_like(wpf)
        lib.call("stes_pack_conv3x3g", lib.ptr(w2), lib.ptr(wpf), lib.ptr(wpd), cmid // 16, stride)
        bns = [((torch.rand(cmid, generator=g) + 0.5).cuda(), (torch.randn(cmid, generator=g) * 0.2).cuda()) for _ in range(2)]
        c1 = torch.empty(M, cmid, dtype=BF, device="cuda")
        c2 = torch.empty(Mo, cmid, dtype=BF, device="cuda")
        pool = torch.empty(N, cmid, dtype=BF, device="cuda")
        t_gemm = timed(lambda: lib.gemm(X, W1, c1, M, cmid, cin, A0=G, lda0=fold, k0=fold))
        t_conv = timed(lambda: lib.call("stes_conv3x3g_fwd_pool", lib.ptr(c1), lib.ptr(wpf), lib.ptr(c2), lib.ptr(bns[0][0]), lib.ptr(bns[0][1]),
                                        lib.ptr(bns[1][0]), lib.ptr(bns[1][1]), lib.ptr(pool), N, wd, wd, wd // stride, wd // stride, stride, cmid))
        t_front = timed(lambda: lib.block_front(X, G, fold, W1, bns[0], wpf, bns[1], c2, pool, N, wd, wd, cin, cmid, stride))
        byt = 2 * M * cin + 2 * Mo * cmid
        print("%3d -> %3d channels, %3d x %3d, stride %d: gemm %6.1f us + conv %6.1f us = %6.1f us; one kernel %6.1f us = %4.2f TB/s of %d MB"
              % (cin, cmid, wd, wd, stride, t_gemm, t_conv, t_gemm + t_conv, t_front, byt / t_front / 1e6, byt / 1e6))
        del X, G, c1, c2


if __name__ == "__main__":
    main()
