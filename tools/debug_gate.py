"""Compare stes_gsm_gate against torch conv3d on the GPU (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import torch.nn.functional as F
from spot_amd import lib
B, T, H, W, fold = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (2, 3, 5, 8, 8)
half = fold // 2
torch.manual_seed(0)
M = B * T * H * W
Xc = torch.randn(M, fold, device="cuda").to(torch.bfloat16)
sc = torch.rand(fold, device="cuda") + 0.5; sh = torch.randn(fold, device="cuda") * 0.3
w3d = (torch.randn(2, half, 3, 3, 3, device="cuda") / (27 * half) ** 0.5).contiguous(); b3d = torch.randn(2, device="cuda") * 0.1
wt = torch.empty(27 * fold, dtype=torch.bfloat16, device="cuda")
lib.call("stes_gsm_prep", lib.ptr(w3d), lib.ptr(wt), fold)
gate = torch.full((M, 2), 9.0, device="cuda")
geom = lib.gsm_geom(B, T, H, W, fold)
lib.call("stes_gsm_gate", lib.ptr(Xc), lib.ptr(sc), lib.ptr(sh), lib.ptr(wt), lib.ptr(b3d), lib.ptr(gate), C.byref(geom))
torch.cuda.synchronize()
xb = torch.relu(Xc.float() * sc + sh).to(torch.bfloat16).float().view(B, T, H, W, fold).permute(0, 4, 1, 2, 3)
ref = torch.tanh(F.conv3d(xb, w3d.to(torch.bfloat16).float(), b3d, padding=1, groups=2)).permute(0, 2, 3, 4, 1).reshape(M, 2)
err = (gate - ref).abs().view(B, T, H, W, 2)
print("max err", float(err.max()), "mean", float(err.mean()))
print("per frame", err.amax((0, 2, 3, 4)).tolist())
print("per row", err.amax((0, 1, 3, 4)).tolist())
print("per col", err.amax((0, 1, 2, 4)).tolist())
print("per grp", err.amax((0, 1, 2, 3)).tolist())
print(gate.view(B, T, H, W, 2)[0, 0, 0, :, 0].tolist()); print(ref.view(B, T, H, W, 2)[0, 0, 0, :, 0].tolist())
