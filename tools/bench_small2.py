"""Times the squeeze-excite FC pair (stes_gemm_small2) on the shapes of the four stages, forward and backward form, alone on the
GPU.  STES_SMALL2_LDS=0 selects the form that reads its operands from memory in rounds."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spot_amd import lib  # noqa: E402

BF = torch.bfloat16


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if __name__ == "__main__":
    M = 1024
    print("stage C rd | forward us | backward us")
    for name, C_, rd in (("s1", 64, 8), ("s2", 128, 16), ("s2'", 128, 32), ("s3.b1", 320, 32), ("s3", 320, 80), ("s4.b1", 768, 80), ("s4", 768, 192)):
        A = torch.randn(M, C_, device="cuda").to(BF)
        W1 = (torch.randn(rd, C_, device="cuda") / C_ ** 0.5).to(BF)
        W2 = (torch.randn(C_, rd, device="cuda") / rd ** 0.5).to(BF)
        b1, b2 = torch.randn(rd, device="cuda"), torch.randn(C_, device="cuda")
        H = torch.empty(M, rd, dtype=BF, device="cuda")
        G = torch.empty(M, C_, dtype=BF, device="cuda")
        mask = torch.randn(M, rd, device="cuda").to(BF)
        D = torch.empty(M, C_, dtype=torch.float32, device="cuda")
        tf = timeit(lambda: lib.gemm2(A, W1, H, W2, G, M, C_, rd, C_, bias1=b1, act1=1, bias2=b2, act2=2))
        tb = timeit(lambda: lib.gemm2(A, W2.t().contiguous() if False else W1, H, W2, D, M, C_, rd, C_, relu_mask1=mask, ldmask1=rd, out_scale=0.25))
        print(f"{name:6s} {C_:4d} {rd:4d} | {tf:7.1f} | {tb:7.1f}", flush=True)
