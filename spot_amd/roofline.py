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


def measure_dominant_kernel(run, steps=3):
    """Roofline of the kernel with the largest share of `run()` (one training step, or one forward).  Every launch of `steps`
    runs is bracketed by HIP events on the launch stream (spot_amd.lib.TIMING); launches are grouped by kernel, the kernel
    with the largest summed duration is the dominant one, and achieved = sum of its launches' algorithmic bytes (every
    operand read once, every result written once) / sum of their durations.  During this pass the weight-gradient kernels
    run on the main stream instead of beside it (engine.OVERLAP off), so that a launch's duration is that kernel's alone and
    not a share of a GPU it splits with another kernel.  `traffic` (HBM bytes per launch from the PMC counters) comes
    from the committed counter run of the training step, profiles/pmc_traffic.json, when it holds this kernel."""
    import json
    import os
    from . import engine, lib
    lib.TIMING = []
    overlap, engine.OVERLAP = engine.OVERLAP, False
    try:
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        recs = lib.TIMING
    finally:
        lib.TIMING = None
        engine.OVERLAP = overlap
    groups = {}
    for kernel, e0, e1, byt, fl in recs:
        g = groups.setdefault(kernel, [0.0, 0, 0, 0, True])
        g[0] += e0.elapsed_time(e1) * 1e-3
        g[1] += 1
        g[2] += byt or 0
        g[3] += fl
        g[4] = g[4] and byt is not None
    total = sum(g[0] for g in groups.values())
    kernel, (t, n, byt, fl, has_bytes) = max(groups.items(), key=lambda kv: kv[1][0])
    gbs = byt / t / 1e9 if has_bytes else None
    traffic = None
    pmc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        rec = json.load(open(pmc)).get(kernel)
        if rec:
            traffic = rec["hbm_bytes_per_launch"]
    out = {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s",
           "frac": gbs / (HBM_PEAK_BPS / 1e9) if gbs else None, "traffic": traffic,
           "launches_per_step": n // steps, "avg_launch_us": 1e6 * t / n, "algorithmic_bytes_per_launch": byt / n if has_bytes else None,
           "share_of_step": t / total,
           # all launches whose byte count is known: what the step moves per the kernels' own accounting, and at what rate
           "step_kernel_ms": 1e3 * total / steps,
           "step_algorithmic_gb": sum(g[2] for g in groups.values() if g[4]) / steps / 1e9,
           "step_ms_with_known_bytes": 1e3 * sum(g[0] for g in groups.values() if g[4]) / steps,
           "by_kernel_ms_per_step": {k: round(1e3 * v[0] / steps, 3) for k, v in sorted(groups.items(), key=lambda kv: -kv[1][0])[:8]},
           # algorithmic GB/s of the kernels whose byte count is known, largest first
           "by_kernel_gbps": {k: round(v[2] / v[0] / 1e9) for k, v in sorted(groups.items(), key=lambda kv: -kv[1][0])[:16] if v[4] and v[2]}}
    if fl:
        out["tflops"] = fl / t / 1e12
        out["mfma_frac"] = fl / t / MFMA_PEAK_FLOPS
    return out
