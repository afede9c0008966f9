"""Micro-benchmarks of single kernels at the shapes of the headline configuration (debug aid; run on the GPU box).

    python tools/bench_kernels.py gemm|wgrad|conv|bn|gsm|all
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import torch

from spot_amd import lib

NF = 1024  # frames
# (name, rows per frame, K, N) of the 1x1 convolutions
GEMMS = [("stem", 112 * 112, 32, 32), ("s1.c1", 112 * 112, 32, 64), ("s1.c3", 56 * 56, 64, 64), ("s2.b1.c1", 56 * 56, 64, 128),
         ("s2.c3", 28 * 28, 128, 128), ("s3.b1.c1", 28 * 28, 128, 320), ("s3.c3", 14 * 14, 320, 320), ("s4.b1.c1", 14 * 14, 320, 768),
         ("s4.c3", 7 * 7, 768, 768), ("gru.in", 1, 768, 4608), ("loc.1", 1, 768, 2304)]
CONVS = [("s1", 112, 64, 2), ("s2.b1", 56, 128, 2), ("s2", 28, 128, 1), ("s3.b1", 28, 320, 2), ("s3", 14, 320, 1), ("s4.b1", 14, 768, 2),
         ("s4", 7, 768, 1)]
GSMS = [("s1.b1", 112, 8), ("s2.b1", 56, 16), ("s2", 28, 32), ("s3", 14, 80), ("s4.b2", 7, 192)]


def timeit(fn, reps=int(os.environ.get('REPS', '5'))):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps   # us


def bf(*shape):
    return torch.randn(*shape, device="cuda").to(torch.bfloat16)


def flush():
    # push earlier tensors out of the 256 MB MALL / L2 so every timing starts cold-ish
    global _junk
    _junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda").zero_()


def bench_gemm():
    print("gemm: name M K N | plain us GB/s | +affine,relu,stats us GB/s")
    for name, rpf, K, N in GEMMS:
        M = NF * rpf
        A, W, Cc = bf(M, K), bf(N, K), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        sc, sh = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda")
        s1, s2 = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
        byt = 2 * M * (K + N) + 2 * N * K
        t0 = timeit(lambda: lib.gemm(A, W, Cc, M, N, K))
        t1 = timeit(lambda: lib.gemm(A, W, Cc, M, N, K, scale=sc, shift=sh, relu=True, stats=(s1, s2)))
        print(f"{name:10s} {M:9d} {K:4d} {N:5d} | {t0:8.1f} {byt / t0 / 1e3:7.0f} | {t1:8.1f} {byt / t1 / 1e3:7.0f}   {2 * M * N * K / t0 / 1e6:6.0f} TF", flush=True)


def bench_wgrad():
    print("wgrad: name M N1 N2 | us GB/s")
    for name, rpf, K, N in GEMMS:
        M = NF * rpf
        P, Q = bf(M, N), bf(M, K)
        dW = torch.zeros(N, K, device="cuda")
        sc, sh = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda")
        byt = 2 * M * (K + N)
        t0 = timeit(lambda: lib.wgrad(P, Q, dW, M, N, K))
        t1 = timeit(lambda: lib.wgrad(P, Q, dW, M, N, K, scale=sc, shift=sh, relu=True))
        print(f"{name:10s} {M:9d} {N:5d} {K:4d} | {t0:8.1f} {byt / t0 / 1e3:7.0f} | {t1:8.1f} {byt / t1 / 1e3:7.0f}  {2 * M * N * K / t0 / 1e6:6.0f} TF", flush=True)


def bench_conv():
    print("conv3x3g: name Hin C stride | fwd(+affine,stats) us GB/s | dgrad us GB/s | wgrad us GB/s")
    for name, Hi, Cc, stride in CONVS:
        Ho = Hi // stride
        X, Y = bf(NF * Hi * Hi, Cc), bf(NF * Ho * Ho, Cc)
        G = Cc // 16
        w = torch.randn(Cc, 16, 3, 3, device="cuda") * 0.1
        wf, wd = torch.empty(G * 5 * 64 * 8, dtype=torch.bfloat16, device="cuda"), torch.empty(G * 5 * 64 * 8, dtype=torch.bfloat16, device="cuda")
        lib.call("stes_pack_conv3x3g", lib.ptr(w), lib.ptr(wf), lib.ptr(wd), G, stride)
        sc, sh = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda")
        s1, s2 = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        dw = torch.zeros(Cc, 16, 3, 3, device="cuda")
        byt = 2 * Cc * NF * (Hi * Hi + Ho * Ho)
        tf = timeit(lambda: lib.call("stes_conv3x3g_fwd", lib.ptr(X), lib.ptr(wf), lib.ptr(Y), lib.ptr(sc), lib.ptr(sh), lib.ptr(s1), lib.ptr(s2),
                                     NF, Hi, Hi, Ho, Ho, stride, Cc))
        tp = timeit(lambda: lib.call("stes_conv3x3g_fwd", lib.ptr(X), lib.ptr(wf), lib.ptr(Y), None, None, None, None, NF, Hi, Hi, Ho, Ho, stride, Cc))
        ta = timeit(lambda: lib.call("stes_conv3x3g_fwd", lib.ptr(X), lib.ptr(wf), lib.ptr(Y), lib.ptr(sc), lib.ptr(sh), None, None, NF, Hi, Hi, Ho, Ho, stride, Cc))
        ts = timeit(lambda: lib.call("stes_conv3x3g_fwd", lib.ptr(X), lib.ptr(wf), lib.ptr(Y), None, None, lib.ptr(s1), lib.ptr(s2), NF, Hi, Hi, Ho, Ho, stride, Cc))
        td = timeit(lambda: lib.call("stes_conv3x3g_dgrad", lib.ptr(Y), lib.ptr(wd), lib.ptr(X), None, None, None, None, None, NF, Hi, Hi, Ho, Ho, stride, Cc))
        X2 = bf(NF * Hi * Hi, Cc)
        tm = timeit(lambda: lib.call("stes_conv3x3g_dgrad", lib.ptr(Y), lib.ptr(wd), lib.ptr(X), lib.ptr(X2), lib.ptr(sc), lib.ptr(sh), lib.ptr(s1), lib.ptr(s2),
                                     NF, Hi, Hi, Ho, Ho, stride, Cc))
        tw = timeit(lambda: lib.call("stes_conv3x3g_wgrad", lib.ptr(Y), lib.ptr(X), lib.ptr(sc), lib.ptr(sh), lib.ptr(dw), NF, Hi, Hi, Ho, Ho, stride, Cc))
        print(f"{name:6s} {Hi:4d} {Cc:4d} {stride} | {tf:8.1f} {byt / tf / 1e3:6.0f} (plain {tp:6.1f} affine {ta:6.1f} stats {ts:6.1f}) | {td:8.1f} {byt / td / 1e3:6.0f} masked {tm:6.1f} | {tw:8.1f} {byt / tw / 1e3:6.0f}", flush=True)


def bench_bn():
    print("bn backward: M C | reduce us GB/s | apply us GB/s | bn_act_residual us GB/s")
    for name, rpf, K, N in GEMMS[:9]:
        M = NF * rpf
        D, raw, out = bf(M, N), bf(M, N), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        sums = [torch.zeros(N, device="cuda") for _ in range(2)]
        coef = [torch.rand(N, device="cuda") for _ in range(3)]
        tr = timeit(lambda: lib.bn_bwd_reduce(D, raw, M, N, sums))
        ta = timeit(lambda: lib.bn_bwd_apply(D, raw, M, N, coef, out))
        print(f"{name:10s} {M:9d} {N:4d} | {tr:8.1f} {4 * M * N / tr / 1e3:6.0f} | {ta:8.1f} {6 * M * N / ta / 1e3:6.0f}", flush=True)


def bench_gsm():
    print("gsm: name H fold | gate | shift | bwd_gate | bwd_wgrad | dxbn | bwd_dx   (us)")
    B, T = 16, 64
    for name, H, fold in GSMS:
        M = B * T * H * H
        geom = lib.gsm_geom(B, T, H, H, fold)
        Xc, G, dG, dxbn, dXf = bf(M, fold), bf(M, fold), bf(M, fold), bf(M, fold), bf(M, fold)
        sc, sh = torch.rand(fold, device="cuda") + 0.5, torch.randn(fold, device="cuda")
        w3d = torch.randn(2, fold // 2, 27, device="cuda") * 0.1
        wt = torch.empty(w3d.numel(), dtype=torch.bfloat16, device="cuda")
        b3d = torch.zeros(2, device="cuda")
        lib.call("stes_gsm_prep", lib.ptr(w3d), lib.ptr(wt), fold)
        gate, dpre = torch.empty(M, 2, device="cuda"), torch.empty(M, 2, device="cuda")
        dw, db = torch.zeros_like(w3d), torch.zeros(2, device="cuda")
        co = [torch.rand(fold, device="cuda") for _ in range(3)]
        g = C.byref(geom)
        ts = [timeit(lambda: lib.call("stes_gsm_gate", lib.ptr(Xc), lib.ptr(sc), lib.ptr(sh), lib.ptr(wt), lib.ptr(b3d), lib.ptr(gate), g)),
              timeit(lambda: lib.call("stes_gsm_shift", lib.ptr(Xc), lib.ptr(gate), lib.ptr(G), g)),
              timeit(lambda: lib.call("stes_gsm_bwd_gate", lib.ptr(dG), lib.ptr(Xc), lib.ptr(gate), lib.ptr(dpre), g)),
              timeit(lambda: lib.call("stes_gsm_bwd_wgrad", lib.ptr(dpre), lib.ptr(Xc), lib.ptr(sc), lib.ptr(sh), lib.ptr(dw), lib.ptr(db), g)),
              timeit(lambda: lib.call("stes_gsm_dxbn", lib.ptr(dpre), lib.ptr(Xc), lib.ptr(sc), lib.ptr(sh), lib.ptr(w3d), lib.ptr(dxbn), g)),
              timeit(lambda: lib.call("stes_gsm_bwd_dx", lib.ptr(dG), lib.ptr(Xc), lib.ptr(gate), lib.ptr(dxbn), lib.ptr(co[0]), lib.ptr(co[1]),
                                      lib.ptr(co[2]), lib.ptr(dXf), g))]
        print(f"{name:6s} {H:4d} {fold:4d} slice {2 * M * fold / 1e6:7.1f} MB | " + " | ".join(f"{t:7.1f}" for t in ts), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.cuda.init()
    for k, f in (("gemm", bench_gemm), ("wgrad", bench_wgrad), ("conv", bench_conv), ("bn", bench_bn), ("gsm", bench_gsm)):
        if what in (k, "all"):
            f()
