"""Every stes_gemm call of one training step that lands on the tiled kernels, with its shape and options (debug aid; run on the
GPU box):  python tools/list_tiled_gemms.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from spot_amd import dp, engine, lib
model = bench.make_model("cuda:0")
x, label, xy = bench.make_batch("cuda:0", 0)
tr = dp.DataParallelTrainer(model, None, lr=1e-3)
tr.step(x, label, xy)
orig = lib.lib().stes_gemm
import ctypes as C
seen = []
class W:
    def __call__(self, aref, stream):
        a = aref._obj
        k = lib._gemm_kernel(a)
        if k.startswith("gemm_kernel"):
            seen.append((k, a.M, a.N, a.K, "A0" if a.A0 else "", "aff" if a.a_scale else "", "gate" if a.a_gate else "", "map" if a.map_stride > 1 else "",
                         "bias" if a.bias else "", "act%d" % a.act, "f32" if a.c_f32 else "", "stats" if a.col_sum else "", "mask" if a.relu_mask else "",
                         "split" if a.n_split else "", "scale" if a.out_scale not in (0.0, 1.0) else ""))
        return orig(aref, stream)
lib.lib().stes_gemm = W()
tr.step(x, label, xy)
torch.cuda.synchronize()
for s in seen: print(" ".join(str(v) for v in s if v != ""))
