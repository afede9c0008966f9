"""Time stes_augment_u8 on the configuration's batch (16 clips x 64 frames of 398x224 decoded video -> 224x224):
    python tools/bench_augment.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spot_amd.augment import ClipAugmenter  # noqa: E402

B, T, H, W = 16, 64, 224, 398
x = torch.randint(0, 256, (B, T, 3, H, W), dtype=torch.uint8, device="cuda")
aug = ClipAugmenter(224, seed=0)
params, _ = aug.sample(B, (H, W))
for label, p in (("drawn parameters", params), ("every colour step on", None)):
    if p is None:
        p = params.copy()
        p[:, 10], p[:, 11] = 1, 0.1
        for c in (12, 14, 16):
            p[:, c], p[:, c + 1] = 1, 0.9
        from spot_amd.augment import gaussian_taps
        p[:, 18], p[:, 19:24] = 1, gaussian_taps(1.0)
    for u8 in (False, True):
        for _ in range(3):
            aug.apply(x, p, u8=u8)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            aug.apply(x, p, u8=u8)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out_b = B * T * 3 * 224 * 224 * (1 if u8 else 4)
        alg = x.numel() + out_b
        print("%-22s %s output: %.3f ms per batch of %d frames = %.0f frames/s, %.2f TB/s of source + output bytes" % (
            label, "uint8" if u8 else "fp32 ", ms, B * T, B * T / ms * 1e3, alg / ms / 1e9))
