"""One GEMM or weight-gradient shape in a loop, for counter passes (debug aid; run on the GPU box):
    python tools/bench_one_gemm.py gemm|wgrad M K N [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from spot_amd import lib

kind = sys.argv[1]
M, K, N = (int(v) for v in sys.argv[2:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
torch.cuda.init()
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
if kind == "gemm":
    W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    Cc = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(reps):
        lib.gemm(A, W, Cc, M, N, K)
else:
    P = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    dW = torch.zeros(N, K, device="cuda")
    for _ in range(reps):
        lib.wgrad(P, A, dW, M, N, K)
torch.cuda.synchronize()
