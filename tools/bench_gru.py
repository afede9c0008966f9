"""GRU recurrence at the head's shape (B=16, T=64, H=768): one launch against a launch per step (debug aid; run on the GPU box)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spot_amd import lib

BF = torch.bfloat16
B, T, H = 16, 64, 768
M = B * T
gi = (torch.randn(M, 6 * H, device="cuda") * 0.5).to(BF)
whh = torch.randn(2, 3 * H, H, device="cuda") / H ** 0.5
whh_d, whhT_d = whh.to(BF), whh.transpose(1, 2).contiguous().to(BF)
bhh = torch.randn(2, 3 * H, device="cuda") * 0.1
dy = torch.randn(B, T, 2 * H, device="cuda")
L = lib.lib()
for one in (False, True):
    nf = L.stes_gru_ws_floats(B, H, 0) if one else 4 * B * H
    nb = L.stes_gru_ws_floats(B, H, 1) if one else 4 * B * H
    y = torch.zeros(B, T, 2 * H, dtype=BF, device="cuda")
    hstate = torch.zeros(nf, device="cuda"); dhs = torch.zeros(nb, device="cuda")
    saved = torch.zeros(2, T, B, 5, H, device="cuda"); hprev = torch.zeros(2, T, B, H, dtype=BF, device="cuda")
    dgi = torch.zeros(M, 6 * H, dtype=BF, device="cuda"); dgh = torch.zeros(2, T, B, 3 * H, dtype=BF, device="cuda")
    af = lib.gru_args(gi=gi, whh=whh_d, whhT=whhT_d, bhh=bhh, y=y, hstate=hstate, saved=saved, hprev_bf=hprev, B=B, T=T, H=H, ws_floats=nf if one else 0)
    ab = lib.gru_args(gi=gi, whh=whh_d, whhT=whhT_d, bhh=bhh, y=y, hstate=hstate, saved=saved, hprev_bf=hprev, dy=dy, ld_dy=2 * H,
                      dgi=dgi, dgh=dgh, dhstate=dhs, B=B, T=T, H=H, ws_floats=nb if one else 0)
    for name, fn, a in (("forward", L.stes_gru_forward, af), ("backward", L.stes_gru_backward, ab)):
        for _ in range(2):
            lib.check(fn(C.byref(a), lib.stream()), name)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.check(fn(C.byref(a), lib.stream()), name)
        e1.record()
        torch.cuda.synchronize()
        print("%-10s %-9s %8.1f us" % ("one launch" if one else "per step", name, e0.elapsed_time(e1) * 1e3 / 5), flush=True)
