"""Gated-shift kernels at the shapes of the full-size step, fused entry points beside the kernel sequences they replace
(debug aid; run on the GPU box):  python tools/bench_gsm.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spot_amd import lib

BF = torch.bfloat16
p = lib.ptr


def run(name, fn, byt, reps=10):
    try:
        fn()
    except lib.StesError as err:
        print("  %-34s not built for this shape (%s)" % (name, str(err)[:60]))
        return
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("  %-34s %8.1f us  %6.0f GB/s" % (name, us, byt / us / 1e3), flush=True)


ONLY = os.environ.get("GSM_ONLY")
SHAPES = [(16, 64, 112, 112, 8, 1), (16, 64, 56, 56, 16, 1), (16, 64, 28, 28, 32, 3), (16, 64, 14, 14, 80, 8), (16, 64, 7, 7, 192, 1)]
for (B, T, H, W, fold, n_blocks) in SHAPES:
    M = B * T * H * W
    half = fold // 2
    print("B=%d T=%d %dx%d fold=%d (Xc %.1f MB), %d block(s) of the network" % (B, T, H, W, fold, M * fold * 2 / 1e6, n_blocks))
    Xc = torch.randn(M, fold, device="cuda").to(BF)
    sc, sh = torch.rand(fold, device="cuda") + 0.5, torch.randn(fold, device="cuda") * 0.3
    w3d = torch.randn(2, half, 27, device="cuda") / (27 * half) ** 0.5
    b3d = torch.randn(2, device="cuda") * 0.1
    wt = torch.empty(27 * fold, dtype=BF, device="cuda")
    lib.call("stes_gsm_prep", p(w3d), p(wt), fold)
    gate = torch.empty(M, 2, device="cuda")
    G = torch.empty(M, fold, dtype=BF, device="cuda")
    dpre = torch.randn(M, 2, device="cuda") * 0.1
    dxbn = torch.empty(M, fold, dtype=BF, device="cuda")
    dw, db = torch.zeros(2, half, 27, device="cuda"), torch.zeros(2, device="cuda")
    sd, sdx = torch.zeros(fold, device="cuda"), torch.zeros(fold, device="cuda")
    geom = lib.gsm_geom(B, T, H, W, fold)
    ws = torch.zeros(lib.lib().stes_gsm_bwd_conv_ws_floats(fold), device="cuda")
    unit = M * fold * 2.0

    def old_fwd():
        lib.call("stes_gsm_gate", p(Xc), p(sc), p(sh), p(wt), p(b3d), p(gate), C.byref(geom))
        lib.call("stes_gsm_shift", p(Xc), p(gate), p(G), C.byref(geom))

    def old_bwd():
        lib.call("stes_gsm_bwd_wgrad", p(dpre), p(Xc), p(sc), p(sh), p(dw), p(db), C.byref(geom))
        lib.call("stes_gsm_dxbn", p(dpre), p(Xc), p(sc), p(sh), p(w3d), p(dxbn), C.byref(geom))
        lib.bn_bwd_reduce(dxbn, Xc, M, fold, [sd, sdx])

    dG = torch.randn(M, fold, device="cuda").to(BF)
    dXf = torch.empty(M, fold, dtype=BF, device="cuda")
    al, be, de = torch.rand(fold, device="cuda"), torch.rand(fold, device="cuda"), torch.rand(fold, device="cuda")
    if not ONLY or ONLY == "each":
        run("gate", lambda: lib.call("stes_gsm_gate", p(Xc), p(sc), p(sh), p(wt), p(b3d), p(gate), C.byref(geom)), unit + 8.0 * M)
        run("shift", lambda: lib.call("stes_gsm_shift", p(Xc), p(gate), p(G), C.byref(geom)), 2 * unit + 8.0 * M)
        run("bwd_gate", lambda: lib.call("stes_gsm_bwd_gate", p(dG), p(Xc), p(gate), p(dpre), C.byref(geom)), 2 * unit + 16.0 * M)
        run("bwd_wgrad", lambda: lib.call("stes_gsm_bwd_wgrad", p(dpre), p(Xc), p(sc), p(sh), p(dw), p(db), C.byref(geom)), unit + 8.0 * M)
        run("dxbn", lambda: lib.call("stes_gsm_dxbn", p(dpre), p(Xc), p(sc), p(sh), p(w3d), p(dxbn), C.byref(geom)), 2 * unit + 8.0 * M)
        run("bn_bwd_reduce", lambda: lib.bn_bwd_reduce(dxbn, Xc, M, fold, [sd, sdx]), 2 * unit)
        run("bwd_dx", lambda: lib.call("stes_gsm_bwd_dx", p(dG), p(Xc), p(gate), p(dxbn), p(al), p(be), p(de), p(dXf), C.byref(geom)),
            4 * unit + 8.0 * M)
    if not ONLY or ONLY == "fwd":
        run("gate + shift", old_fwd, 2 * unit)
        run("gsm_forward", lambda: lib.call("stes_gsm_forward", p(Xc), p(sc), p(sh), p(wt), p(b3d), p(gate), p(G), C.byref(geom)), 2 * unit)
    if not ONLY or ONLY == "bwd":
        run("bwd_wgrad + dxbn + reduce", old_bwd, 2 * unit)
        run("gsm_bwd_conv", lambda: lib.call("stes_gsm_bwd_conv", p(dpre), p(Xc), p(sc), p(sh), p(w3d), p(dxbn), p(sd), p(sdx), p(dw), p(db),
                                             p(ws), C.byref(geom)), 2 * unit)
        run("gsm_bwd_conv without conv3D grads", lambda: lib.call("stes_gsm_bwd_conv", p(dpre), p(Xc), p(sc), p(sh), p(w3d), p(dxbn), p(sd), p(sdx), None,
                                                                  None, None, C.byref(geom)), 2 * unit)
        run("gsm_bwd_conv, no workspace", lambda: lib.call("stes_gsm_bwd_conv", p(dpre), p(Xc), p(sc), p(sh), p(w3d), p(dxbn), p(sd), p(sdx),
                                                           p(dw), p(db), None, C.byref(geom)), 2 * unit)
