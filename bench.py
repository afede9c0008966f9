#!/usr/bin/env python3
"""Headline benchmark: frames/sec of one full training step (forward + backward + AdamW) of
rny008_gsm + BiGRU + location MLP, clip_len 64, 224x224, 16 clips per GPU, bf16 storage / fp32 accumulate.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU.  Data parallel: every rank steps its own 16 clips (weak scaling); the only exchange is the
gradient all-reduce (sum) of the flat fp32 gradient buffer over RCCL, plus two scalars (CE weight sum, event count) so
that the summed gradient is exactly the gradient of the global-batch loss.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU, CLIP_LEN, RES, NUM_CLASSES = 16, 64, 224, 7
# analytic per-frame work of the named configuration (recomputable: spot_amd/roofline.py)
HBM_PEAK_GBS = 8000.0


def make_model(device, seed=0):
    from spot_amd.e2e import E2EModel
    torch.manual_seed(seed)
    m = E2EModel(NUM_CLASSES, "rny008_gsm", "gru", clip_len=CLIP_LEN, modality="rgb", device=device, predict_location=True,
                 verbose=False)
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed)
        for n, p in m._model.named_parameters():
            if "conv3D.weight" in n:        # zero init makes the gate trivially 0 (reference model/shift.py:89-90)
                p.copy_((torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5).to(p.device))
            elif n.endswith("conv3.bn.weight"):
                p.fill_(0.25)               # timm zero-inits it; keep the residual branch live
    return m


def make_batch(device, rank):
    g = torch.Generator(device=device).manual_seed(1000 + rank)
    x = torch.randn(B_PER_GPU, CLIP_LEN, 3, RES, RES, device=device, generator=g)
    fg = torch.rand(B_PER_GPU, CLIP_LEN, device=device, generator=g) < 0.0244      # 6137 events / 251110 frames
    label = fg.long() * torch.randint(1, NUM_CLASSES, (B_PER_GPU, CLIP_LEN), device=device, generator=g)
    xy = torch.rand(B_PER_GPU, CLIP_LEN, 2, device=device, generator=g)
    return x, label, xy


def cpu_baseline():
    """The CPU oracle (validated against the reference's own modules) timed on this host: one optimisation step on
    ONE clip of 64 frames at 224x224 (a bounded sample of the same workload), all cores the process may use."""
    from oracle import stes_oracle as O
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    om = O.Impl(NUM_CLASSES, "rny008_gsm", "gru", clip_len=CLIP_LEN, predict_location=True)
    O.randomize_(om, seed=0)
    opt = O.make_adamw(om, 1e-3)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, CLIP_LEN, 3, RES, RES, generator=g)
    label = torch.randint(0, NUM_CLASSES, (1, CLIP_LEN), generator=g)
    xy = torch.rand(1, CLIP_LEN, 2, generator=g)
    for _ in range(2):                                       # warm-up
        O.train_step(om, opt, x, label, xy, NUM_CLASSES)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        O.train_step(om, opt, x, label, xy, NUM_CLASSES)
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[len(ts) // 2]
    return {"value": CLIP_LEN / t, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "1 clip x 64 frames x 224x224, fwd+bwd+AdamW, fp32 torch CPU oracle, median of 5 after 2 warm-ups"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--serial", action="store_true",
                    help="weight-gradient kernels in program order on the main stream (per-kernel profiles: durations are each kernel's own)")
    args = ap.parse_args()
    if args.serial:
        from spot_amd import engine
        engine.OVERLAP = False

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))

    from spot_amd import dp
    model = make_model(device, seed=0)                  # identical initial weights on every rank
    x, label, xy = make_batch(device, rank)
    trainer = dp.DataParallelTrainer(model, dist if world > 1 else None, lr=1e-3)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(x, label, xy)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.step(x, label, xy)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss = [float(v) for v in out[:3].cpu()]
    frames = world * B_PER_GPU * CLIP_LEN * args.steps
    ms = 1e3 * dt / args.steps
    rec = {
        "metric": "frames/sec fwd+bwd, rny008_gsm+BiGRU clip64 224p", "value": frames / dt, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "rny008_gsm + BiGRU(L=1) + loc MLP training step (fwd+bwd+AdamW), bf16 storage/fp32 accumulate",
                   "clips_per_gpu": B_PER_GPU, "clip_len": CLIP_LEN, "resolution": RES, "global_batch_clips": world * B_PER_GPU,
                   "parallelism": f"dp{world}", "num_classes": NUM_CLASSES},
        "final_loss": loss, "overlap_weight_gradients": not args.serial,
    }
    if rank == 0 and world == 1:
        from spot_amd import roofline
        rec["step_roofline"] = roofline.step_fractions(frames / dt)
        if not args.no_roofline:
            rec["roofline"] = roofline.measure_dominant_kernel(trainer, (x, label, xy))
        if not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
