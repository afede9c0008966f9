"""Analytic work model of the named configuration and the live roofline measurement of the dominant kernel."""
from __future__ import annotations

import torch

MFMA_PEAK_FLOPS = 2.5e15      # dense bf16, MI355X
HBM_PEAK_BPS = 8.0e12         # spec; ~6.3e12 achievable

STAGES = ((64, 1), (128, 3), (320, 8), (768, 2))   # regnety_008: (width, depth); group width 16; stem 32


def per_frame_work(res=224, hidden=768, num_classes=7, T=64):
    """MACs and algorithmic HBM bytes per frame (bf16 activations: every conv output written once and read once, the
    block input read by conv1/gated shift and again by the shortcut; backward = read saved activation + read dY + write dX
    per conv = 1.5x forward)."""
    h = res // 2
    macs = 27 * 32 * h * h
    conv_out = 32 * h * h
    block_in = 0
    cin = 32
    gsm_macs = 0
    for w, d in STAGES:
        for b in range(d):
            s = 2 if b == 0 else 1
            ho = h // s
            fold = -(-(cin // 4) // 4) * 4
            gsm_macs += 27 * fold * h * h
            macs += cin * w * h * h + 9 * 16 * w * ho * ho + w * w * ho * ho
            conv_out += w * h * h + 2 * w * ho * ho
            if b == 0:
                macs += cin * w * ho * ho
                conv_out += w * ho * ho
            rd = round(cin * 0.25)
            macs += 2 * w * rd
            block_in += cin * h * h
            cin, h = w, ho
    gru = 2 * (3 * hidden * cin + 3 * hidden * hidden)
    loc = cin * 3 * hidden + 3 * hidden * hidden + 2 * hidden
    fc = 2 * hidden * num_classes
    fwd_macs = macs + gsm_macs + gru + loc + fc
    fwd_bytes = 4 * 3 * res * res + 2 * (2 * conv_out + block_in)
    return {"fwd_macs": fwd_macs, "step_flops": 3 * 2 * fwd_macs, "fwd_bytes": fwd_bytes, "step_bytes": 2.5 * fwd_bytes}


def step_fractions(fps_per_gpu):
    w = per_frame_work()
    return {"flops_per_frame": w["step_flops"], "bytes_per_frame": w["step_bytes"],
            "mfma_frac": w["step_flops"] * fps_per_gpu / MFMA_PEAK_FLOPS,
            "hbm_frac": w["step_bytes"] * fps_per_gpu / HBM_PEAK_BPS}


def measure_dominant_kernel(trainer, batch, steps=3):
    """Roofline of the kernel with the largest share of the step (gemm_kernel<128>: the 1x1 convolutions, their input
    gradients and the head GEMMs, ~64 launches per step of different shapes).  Every launch of `steps` training steps is
    bracketed by HIP events on the launch stream; achieved = sum of the launches' algorithmic bytes (A read once, W read
    once, C written once, SE gate read once) / sum of their durations.  FLOP rate reported beside it."""
    from . import lib
    x, label, xy = batch
    lib.TIMING = []
    try:
        for _ in range(steps):
            trainer.step(x, label, xy)
        torch.cuda.synchronize()
        recs = lib.TIMING
    finally:
        lib.TIMING = None
    t = sum(e0.elapsed_time(e1) for e0, e1, _, _ in recs) * 1e-3
    byt = sum(r[2] for r in recs)
    fl = sum(r[3] for r in recs)
    n = len(recs)
    gbs = byt / t / 1e9
    return {"kernel": "gemm_kernel<128> (stes_gemm, 128-wide tile)", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_BPS / 1e9,
            "unit": "GB/s", "frac": gbs / (HBM_PEAK_BPS / 1e9), "traffic": None,
            "launches_per_step": n // steps, "avg_launch_us": 1e6 * t / n, "algorithmic_bytes_per_launch": byt / n,
            "tflops": fl / t / 1e12, "mfma_frac": fl / t / MFMA_PEAK_FLOPS}
