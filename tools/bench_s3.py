"""Stage-3 shapes (200,704 rows x 320 channels) of the streaming GEMM, each operand prologue, cycling over several buffer
sets so that operands come from HBM and not the Infinity Cache (debug aid; run on the GPU box):  python tools/bench_s3.py [M N K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spot_amd import lib

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16 * 64 * 196, 320, 320)
BF = torch.bfloat16
NB = 4
A = [torch.randn(M, K, device="cuda").to(BF) for _ in range(NB)]
R = [torch.randn(M, K, device="cuda").to(BF) for _ in range(NB)]
Cc = [torch.empty(M, N, dtype=BF, device="cuda") for _ in range(NB)]
AO = [torch.empty(M, K, dtype=BF, device="cuda") for _ in range(NB)]
W = torch.randn(N, K, device="cuda").to(BF)
sc, sh = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda")
gate = torch.rand(M // 196, K, device="cuda")
st = torch.zeros(2, N, device="cuda")


def run(name, byt, fn, reps=24):
    for i in range(4):
        fn(i % NB)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % NB)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("%-44s %7.1f us  %6.0f GB/s" % (name, us, byt / us / 1e3), flush=True)


ONLY = os.environ.get("S3_ONLY")
_run = run
def run(name, byt, fn, reps=24):
    if ONLY is None or ONLY in name:
        _run(name, byt, fn, reps)


base = 2.0 * (M * K + M * N)
run("plain", base, lambda i: lib.gemm(A[i], W, Cc[i], M, N, K))
run("bn+relu prologue", base, lambda i: lib.gemm(A[i], W, Cc[i], M, N, K, scale=sc, shift=sh, relu=True))
run("bn+relu prologue, stats", base, lambda i: lib.gemm(A[i], W, Cc[i], M, N, K, scale=sc, shift=sh, relu=True, stats=(st[0], st[1])))
run("bn+relu+gate prologue, stats", base, lambda i: lib.gemm(A[i], W, Cc[i], M, N, K, scale=sc, shift=sh, relu=True, gate=gate, gate_group=196,
                                                             stats=(st[0], st[1])))
run("bn-backward prologue + a_out", base + 4.0 * M * K, lambda i: lib.gemm(A[i], W, Cc[i], M, N, K, bn_bwd=(R[i], (sc, sh, sh), AO[i])))
