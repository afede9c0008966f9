"""Weight-gradient shapes of the full-size step, one by one (debug aid; run on the GPU box):
    python tools/bench_wgrad_shapes.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spot_amd import lib

F = 16 * 64
SHAPES = [  # (what, M, N1 (dY cols), N2 (X cols), calls per step)
    ("s1 conv1/3/down 64x32..", F * 56 * 56, 64, 64, 2), ("s1 conv1 64x32", F * 112 * 112, 64, 32, 1),
    ("s2 128x128", F * 28 * 28, 128, 128, 5), ("s2.b1 conv1 128x64", F * 56 * 56, 128, 64, 1), ("s2 down 128x64", F * 28 * 28, 128, 64, 1),
    ("s3 320x320", F * 14 * 14, 320, 320, 15), ("s3.b1 conv1 320x128", F * 28 * 28, 320, 128, 1), ("s3 down 320x128", F * 14 * 14, 320, 128, 1),
    ("s4 768x768", F * 7 * 7, 768, 768, 3), ("s4.b1 conv1 768x320", F * 14 * 14, 768, 320, 1), ("s4 down 768x320", F * 7 * 7, 768, 320, 1),
    ("gru wih 4608x768", F, 4608, 768, 1), ("gru whh 2304x768", F, 2304, 768, 2), ("loc 2304x768", F, 2304, 768, 1), ("loc 768x2304", F, 768, 2304, 1),
    ("se s3 80x320", F, 80, 320, 8), ("se s3 320x80", F, 320, 80, 8),
]
# the in-step forms of the conv3 weight gradients: Q = relu(bn2(c2)) * squeeze-excite gate, (what, M, N1, N2, calls, rows per gate row)
PRO_SHAPES = [("s2 conv3 128x128 bn+relu+gate", F * 28 * 28, 128, 128, 3, 784), ("s4 conv3 768x768 bn+relu+gate", F * 7 * 7, 768, 768, 2, 49),
              ("s4.b1 conv1 768x320 plain", F * 14 * 14, 768, 320, 1, 0), ("s1 conv3 64x64 bn+relu+gate", F * 56 * 56, 64, 64, 1, 3136)]
torch.cuda.init()
tot = 0.0
for what, M, N1, N2, calls, grp in PRO_SHAPES if "--pro" in sys.argv else []:
    P = torch.randn(M, N1, device="cuda").to(torch.bfloat16)
    Q = torch.randn(M, N2, device="cuda").to(torch.bfloat16)
    dW = torch.zeros(N1, N2, device="cuda")
    sc, sh = torch.rand(N2, device="cuda") + 0.5, torch.randn(N2, device="cuda")
    kw = dict(scale=sc, shift=sh, relu=True, gate=torch.rand(M // grp, N2, device="cuda"), gate_group=grp) if grp else {}
    for _ in range(3):
        lib.wgrad(P, Q, dW, M, N1, N2, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.wgrad(P, Q, dW, M, N1, N2, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print("%-32s M=%8d  %7.1f us  %6.0f GB/s  %6.1f TFLOP/s  x%d" % (what, M, us, 2.0 * M * (N1 + N2) / us / 1e3, 2.0 * M * N1 * N2 / us / 1e6, calls), flush=True)
    del P, Q, dW
if "--pro" in sys.argv:
    sys.exit(0)
for what, M, N1, N2, calls in SHAPES:
    P = torch.randn(M, N1, device="cuda").to(torch.bfloat16)
    Q = torch.randn(M, N2, device="cuda").to(torch.bfloat16)
    dW = torch.zeros(N1, N2, device="cuda")
    kw = {"ws": torch.empty(lib.lib().stes_wgrad_ws_floats(), device="cuda")} if (N1, N2) == (320, 320) else {}
    for _ in range(3):
        lib.wgrad(P, Q, dW, M, N1, N2, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        lib.wgrad(P, Q, dW, M, N1, N2, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    byt = 2.0 * M * (N1 + N2) + 4.0 * N1 * N2
    fl = 2.0 * M * N1 * N2
    tot += us * calls
    print("%-26s M=%8d  %7.1f us  %6.0f GB/s  %6.1f TFLOP/s  x%d = %6.0f us" % (what, M, us, byt / us / 1e3, fl / us / 1e6, calls, us * calls), flush=True)
    del P, Q, dW
print("total %.2f ms" % (tot / 1e3))
