#!/usr/bin/env python3
"""Headline benchmark: frames/sec of one full training step (forward + backward + AdamW) of
rny008_gsm + BiGRU + location MLP, clip_len 64, 224x224, 16 clips per GPU, bf16 storage / fp32 accumulate.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

    python bench.py --config {train_gsm_gru (default, the headline), train_tsm_mingru, infer_fwd, infer_long_video}

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


def make_model(device, seed=0, arch="rny008_gsm", tarch="gru"):
    from spot_amd.e2e import E2EModel
    torch.manual_seed(seed)
    m = E2EModel(NUM_CLASSES, arch, tarch, clip_len=CLIP_LEN, modality="rgb", device=device, predict_location=True,
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


def _median_time(fn, reps, warm):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def cpu_baseline():
    """The CPU oracle (validated against the reference's own modules) timed on this host, on bounded samples of the same
    workload.  The headline leg: one optimisation step on ONE clip of 64 frames at 224x224, all cores the process may
    use.  Further legs: the reference's own CPU-runnable case (forward of one 64-frame clip at 398x224, predict_location
    on) on all cores, and the training step on ONE thread (the reference scripts export OMP_NUM_THREADS=1, run.sh:17) on
    a 16-frame clip so that the leg stays within seconds."""
    from oracle import stes_oracle as O
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)

    def train_leg(clip_len, nthreads, reps, warm):
        torch.set_num_threads(nthreads)
        om = O.Impl(NUM_CLASSES, "rny008_gsm", "gru", clip_len=clip_len, predict_location=True)
        O.randomize_(om, seed=0)
        opt = O.make_adamw(om, 1e-3)
        x = torch.randn(1, clip_len, 3, RES, RES, generator=g)
        label = torch.randint(0, NUM_CLASSES, (1, clip_len), generator=g)
        xy = torch.rand(1, clip_len, 2, generator=g)
        return clip_len / _median_time(lambda: O.train_step(om, opt, x, label, xy, NUM_CLASSES), reps, warm)

    main_fps = train_leg(CLIP_LEN, threads, 5, 2)
    legs = []
    om = O.Impl(NUM_CLASSES, "rny008_gsm", "gru", clip_len=CLIP_LEN, predict_location=True)
    O.randomize_(om, seed=0)
    x1 = torch.randn(1, CLIP_LEN, 3, 224, 398, generator=g)
    torch.set_num_threads(threads)
    legs.append({"value": CLIP_LEN / _median_time(lambda: O.predict(om, x1), 5, 2), "unit": "frames/s", "cores": threads, "kind": "port",
                 "sample": "forward (predict), 1 clip x 64 frames x 398x224, predict_location on, fp32 torch CPU oracle, median of 5 after 2 warm-ups"})
    legs.append({"value": train_leg(16, 1, 3, 1), "unit": "frames/s", "cores": 1, "kind": "port",
                 "sample": "1 clip x 16 frames x 224x224 (clip_len 16), fwd+bwd+AdamW, ONE thread, fp32 torch CPU oracle, median of 3 after 1 warm-up"})
    torch.set_num_threads(threads)
    return {"value": main_fps, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "1 clip x 64 frames x 224x224, fwd+bwd+AdamW, fp32 torch CPU oracle, median of 5 after 2 warm-ups",
            "host_cores": os.cpu_count(), "legs": legs}


# ---------------------------------------------------------------- the other configurations of the benchmark family
def bench_infer(args, device, rank, world, dist, long_video):
    """Forward only.  infer_fwd: predict() on a resident batch of 16 clips x 64 frames at 224x224.  infer_long_video: the
    sliding-window path over a 10-minute 25 fps video (15,000 decoded uint8 frames resident in HBM = 235 windows of 64, the
    last one zero-padded; batches of 16 windows dealt round-robin to the ranks), per-frame scores summed, events
    picked, filtered and suppressed as the reference does (inference_on_mp4.py:204-277)."""
    from spot_amd import infer
    model = make_model(device, seed=0)
    classes = ["c%d" % i for i in range(1, NUM_CLASSES)]
    if long_video:
        # a randomly initialised classifier is never confident enough to pass the 0.15 score filter, so the event list would
        # be empty and the filter + suppression steps would have nothing to do: scale its weights as a trained model's are
        with torch.no_grad():
            model._model._pred_fine._fc_out._fc_out.weight.mul_(8.0)
        n_frames = 15000
        gen = torch.Generator(device=device).manual_seed(7)
        frames = torch.randint(0, 256, (n_frames, 3, RES, RES), device=device, generator=gen, dtype=torch.uint8)

        def gather(scores, support, locations):            # disjoint frame ranges: the sum over ranks is the gather
            if world == 1:
                return scores, support, locations
            ts = [torch.from_numpy(a).to(device) for a in (scores, support, locations)]
            for t in ts:
                dist.all_reduce(t)
            return tuple(t.cpu().numpy() for t in ts)

        def one():
            return infer.run_inference(model, frames, classes, CLIP_LEN, B_PER_GPU, rank=rank, world=world, gather=gather)
        unit_frames = n_frames
    else:
        x = make_batch(device, rank)[0]

        def one():
            return model.predict(x)
        unit_frames = world * B_PER_GPU * CLIP_LEN
    for _ in range(args.warmup):
        out = one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    from spot_amd import roofline
    w = roofline.per_frame_work()
    fps = unit_frames * args.steps / dt
    rec = {"metric": "frames/sec forward, rny008_gsm+BiGRU clip64 224p" + (", 10-minute video sliding window" if long_video else ""),
           "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if long_video else "weak",
           "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": ("sliding-window inference over 15,000 uint8 frames (235 windows of 64, batches of 16), events produced"
                                   if long_video else "predict() on 16 clips x 64 frames, eval BatchNorm, softmax/argmax/sigmoid included"),
                      "clip_len": CLIP_LEN, "resolution": RES, "parallelism": f"replicas{world}", "num_classes": NUM_CLASSES},
           "step_roofline": {"flops_per_frame": 2 * w["fwd_macs"], "bytes_per_frame": w["fwd_bytes"],
                             "mfma_frac": 2 * w["fwd_macs"] * fps / world / roofline.MFMA_PEAK_FLOPS,
                             "hbm_frac": w["fwd_bytes"] * fps / world / roofline.HBM_PEAK_BPS}}
    if long_video:
        rec["events"] = len(out[0][0]["events"])
    elif rank == 0 and world == 1 and not args.no_roofline:
        rec["roofline"] = roofline.measure_dominant_kernel(one)
    del model
    torch.cuda.empty_cache()
    return rec


def bench_train(args, device, rank, world, dist, tsm):
    """One optimisation step (forward + loss + backward + AdamW) on 16 clips x 64 frames per rank, timed over args.steps."""
    from spot_amd import dp
    model = make_model(device, seed=0, arch="rny008_tsm" if tsm else "rny008_gsm", tarch="mingru" if tsm else "gru")   # identical initial weights on every rank
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
    # time the host needs to issue one step's launches (ctypes calls into the library, no synchronisation), measured on an
    # idle GPU so that a full launch queue cannot hold the host back: while it stays below ms_per_step the GPU never waits
    host_ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        trainer.step(x, label, xy)
        host_ts.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
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
        "host_ms_per_step": 1e3 * sorted(host_ts)[1],
    }
    if tsm:
        rec["metric"] = "frames/sec fwd+bwd, rny008_tsm+minGRU clip64 224p"
        rec["config"]["workload"] = "rny008_tsm + minGRU(3 layers) + loc MLP training step (fwd+bwd+AdamW), bf16 storage/fp32 accumulate"
    if rank == 0 and world == 1 and not tsm:
        from spot_amd import roofline
        rec["step_roofline"] = roofline.step_fractions(frames / dt)
        if not args.no_roofline:
            rec["roofline"] = roofline.measure_dominant_kernel(lambda: trainer.step(x, label, xy))
    del trainer, model
    torch.cuda.empty_cache()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="train_gsm_gru", choices=["train_gsm_gru", "train_tsm_mingru", "infer_fwd", "infer_long_video"],
                    help="train_gsm_gru is the headline configuration; the others are the further configurations of the benchmark family")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="headline line only (the default line also times the other three configurations)")
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

    if args.config.startswith("infer"):
        rec = bench_infer(args, device, rank, world, dist, args.config == "infer_long_video")
        if rank == 0:
            print(json.dumps(rec), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    rec = bench_train(args, device, rank, world, dist, args.config == "train_tsm_mingru")
    if args.config == "train_gsm_gru" and rank == 0 and world == 1:
        if not args.no_other_configs:
            # the further configurations of the benchmark family, shorter runs of the same code paths as --config runs them
            import copy
            short = copy.copy(args)
            short.steps, short.warmup = 5, 2
            other = {"infer_fwd": bench_infer(short, device, rank, world, dist, False)}
            short.steps, short.warmup = 2, 1
            other["infer_long_video"] = bench_infer(short, device, rank, world, dist, True)
            short.steps, short.warmup, short.no_roofline = 5, 2, True
            other["train_tsm_mingru"] = bench_train(short, device, rank, world, dist, True)
            rec["other_configs"] = other
        if not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
