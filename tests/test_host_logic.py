"""CPU tests: host-side logic, the C-ABI surface, the data-parallel bucketing over gloo (world_size 2)."""
import ctypes
import json
import os
import random
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_exports_every_declared_symbol():
    from spot_amd import lib as L
    if not os.path.exists(L.LIB_PATH):
        from spot_amd.build import build
        build(verbose=False)
    dll = ctypes.CDLL(L.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "stes.h")).read()
    declared = set(re.findall(r"\b(stes_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(dll, name), f"{name} declared in include/stes.h but not exported"
    assert declared == set(L.EXPORTS), (declared ^ set(L.EXPORTS))
    dll.stes_version.restype = ctypes.c_int
    assert dll.stes_version() >= 1


def test_no_product_import_of_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "spot_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f


def test_engine_refuses_cpu():
    from spot_amd.e2e import E2EModel
    from spot_amd.lib import StesError
    m = E2EModel(7, "rny008_gsm", "gru", clip_len=4, modality="rgb", device="cpu", predict_location=True, verbose=False)
    with pytest.raises(StesError):
        m.predict(torch.zeros(1, 4, 3, 32, 32))


def test_unsupported_archs_raise_like_reference():
    from spot_amd.e2e import E2EModel
    with pytest.raises(NotImplementedError):
        E2EModel(7, "rn18_gsm", "gru", clip_len=4, modality="rgb", device="cpu", verbose=False)
    with pytest.raises(NotImplementedError):
        E2EModel(7, "rny008_gsm", "lstm", clip_len=4, modality="rgb", device="cpu", verbose=False)
    with pytest.raises(NotImplementedError):
        E2EModel(7, "rny008_gsm", "gru", clip_len=4, modality="rgb", device="cpu", predict_location=True, pred_loc_arch="conv", verbose=False)


def test_postproc_bit_exact_vs_reference_golden(golden):
    from spot_amd.util import postproc as P
    z = golden("postproc")
    classes = ["serve", "receive", "set", "attack", "block", "score"]
    ev = P.events_from_scores(z["scores"].copy(), z["locs"], classes)
    f = P.filter_events_by_score([{"video": "v", "events": ev}], 0.15)
    import copy
    s = P.non_max_suppression_events(copy.deepcopy(f), 3)
    lab = {c: i for i, c in enumerate(classes)}
    enc = lambda e: np.array([[x["frame"], lab[x["label"]]] for x in e], np.int64).reshape(-1, 2)
    assert np.array_equal(enc(f[0]["events"]), z["filtered"])
    assert np.array_equal(enc(s[0]["events"]), z["nms"])
    assert np.array_equal(np.array([e["score"] for e in s[0]["events"]]), z["nms_scores"])


def test_event_metrics_bit_exact_vs_reference_golden(golden):
    """Temporal mAP and spatial AP tables on three synthetic videos (jittered detections, tied-score false alarms) equal
    the reference's util/score.py outputs to the last bit."""
    import copy
    from spot_amd.util import postproc as P, score as S
    z = golden("scores")
    truth, pred = json.loads(str(z["truth"])), json.loads(str(z["pred"]))
    truth0, pred0 = copy.deepcopy(truth), copy.deepcopy(pred)
    nms = P.non_max_suppression_events(P.filter_events_by_score(copy.deepcopy(pred), 0.15), 3)
    mAPs, tol = S.compute_mAPs(truth, nms, tolerances=[0, 1, 2, 4], quiet=True)
    assert tol == list(z["tolerances"]) and np.array_equal(np.array(mAPs), z["mAPs"])
    mt, mp = S.compute_mAPs_with_locations(truth, pred, quiet=True)
    assert np.array_equal(np.array(mt), z["mAPs_t"]) and np.array_equal(np.array(mp), z["mAP_ps"])
    assert truth == truth0 and pred == pred0                       # inputs are not modified
    assert 0.02 < z["mAPs"][0] < z["mAPs"][-1] < 1 and 0 < z["mAP_ps"][0] < z["mAP_ps"][-1]   # a non-degenerate case
    with pytest.raises(AssertionError, match="Video set mismatch"):
        S.compute_mAPs(truth, pred[:2], quiet=True)
    # one class, hand-checked: detections at ranks 1 and 3 hit -> precisions 1, 2/3 -> AP = (1 + 2/3) / 3 ground-truth events
    t = {"v": [10, 20, 30]}
    r = [("v", 10, 0.9, None), ("v", 50, 0.8, None), ("v", 21, 0.7, None)]
    assert S.average_precision(r, t, tolerance=1) == (1 + 2 / 3) / 3
    assert S.average_precision(r, t, tolerance=0) == 1 / 3


def test_frame_statistics_and_event_lists_bit_exact_vs_reference_golden(golden):
    """Frame error, F1 counts, argmax and high-recall event lists, windowed peak picking: equal to the reference's
    util/eval.py outputs on two synthetic videos (supports of 1 and 2, labels on 15% of frames)."""
    from spot_amd.util import eval as E
    z = golden("eval")
    classes = {c: i + 1 for i, c in enumerate(["serve", "receive", "set", "attack", "block", "score"])}

    class DS:
        videos = [("a", 90, 25.0), ("b", 57, 25.0)]

        @staticmethod
        def get_labels(video, with_locations=False):
            lab = z["label_" + video]
            return (lab, np.zeros((len(lab), 2), np.float32)) if with_locations else lab

    pred = {v: (z["scores_" + v].copy(), z["support_" + v], z["loc_" + v]) for v in "ab"}
    err, f1, ev, ev_hr, sc = E.process_frame_predictions_with_location(DS, classes, pred)
    assert err.get() == float(z["err"]) and 0 < err.get() < 1
    keys = [None] + list(range(1, 7))
    assert np.array_equal(np.array([f1.tp_fp_fn(k) for k in keys]), z["counts"]) and z["counts"].min(0).max() > 0
    assert np.array_equal(np.array([f1.get(k) for k in keys]), z["f1"])
    assert ev == json.loads(str(z["events"])) and ev_hr == json.loads(str(z["events_hr"])) and len(ev_hr[0]["events"]) > len(ev[0]["events"]) > 10
    assert sc == json.loads(str(z["mean_scores"]))
    assert E.non_maximum_supression(ev_hr, 2) == json.loads(str(z["nms"]))
    _, _, ev2, _, _ = E.process_frame_predictions(DS, classes, {v: (z["scores_" + v].copy(), z["support_" + v]) for v in "ab"})
    assert ev2 == json.loads(str(z["events_noloc"])) and "xy" not in ev2[0]
    with pytest.raises(AssertionError):                      # a frame no clip covered
        E.process_frame_predictions(DS, classes, {"a": (z["scores_a"].copy(), np.zeros(90, np.int32))})


def test_video_clip_layout_and_evaluate_with_a_stub_model(tmp_path):
    """Clip starts, padding and stride follow the reference's video dataset; evaluate() recovers planted events."""
    from spot_amd import evaluate as EV
    classes = {"serve": 1, "set": 2}
    labels = [{"video": "v0", "num_frames": 50, "fps": 25, "events": [{"frame": 7, "label": "serve", "xy": [0.25, 0.5]},
                                                                      {"frame": 30, "label": "set", "xy": [0.75, 0.25]}]},
              {"video": "v1", "num_frames": 21, "fps": 30, "events": [{"frame": 20, "label": "set", "xy": [0.5, 0.5]}]}]
    frames = {}
    for l in labels:                                         # frame f of a video holds f + 1 in every byte (0 = padding)
        frames[l["video"]] = (torch.arange(1, l["num_frames"] + 1, dtype=torch.uint8).view(-1, 1, 1, 1).expand(-1, 3, 2, 2).contiguous())
    ds = EV.VideoClipDataset(classes, labels, frames, clip_len=16, overlap_len=8)
    starts = [s for v, s in ds._clips if v == "v0"]
    assert starts == list(range(-5, 42, 8)) and [s for v, s in ds._clips if v == "v1"] == [-5, 3, 11]
    c = ds[0]
    assert c["start"] == -5 and c["frame"][:5].eq(0).all() and c["frame"][5, 0, 0, 0] == 1 and c["frame"][15, 0, 0, 0] == 11
    last = ds[len(starts) - 1]
    assert last["start"] == 35 and last["frame"][14, 0, 0, 0] == 50 and last["frame"][15].eq(0).all()
    assert ds.videos == [("v0", 50, 25.0), ("v1", 21, 30.0)]
    lab, loc = ds.get_labels("v0", with_locations=True)
    assert lab[7] == 1 and lab[30] == 2 and lab.sum() == 3 and loc[30].tolist() == [0.75, 0.25]
    ds2 = EV.VideoClipDataset(classes, labels, frames, clip_len=8, stride=2)
    assert ds2[0]["start"] == -5 and ds2[1]["frame"][:, 0, 0, 0].tolist() == [7, 9, 11, 13, 15, 17, 19, 21]
    assert ds2.videos[0] == ("v0", 25, 12.5) and ds2.labels[0]["events"][0]["frame"] == 3 and labels[0]["events"][0]["frame"] == 7
    assert sorted(b["start"][0] for r in range(2) for b in ds.batches(4, r, 2)) == sorted(ds[i]["start"] for i in range(0, len(ds), 4))

    truth = {l["video"]: {e["frame"]: e for e in l["events"]} for l in labels}

    class Stub:                                              # reads the frame number back from the pixels
        def predict(self, clip):
            B, T = clip.shape[:2]
            scores, loc = np.zeros((B, T, 3), np.float32), np.zeros((B, T * 2), np.float32)
            scores[..., 0] = 1
            for b in range(B):
                vid = self.videos[b]
                for t in range(T):
                    e = truth[vid].get(int(clip[b, t, 0, 0, 0]) - 1)
                    if e is not None:
                        scores[b, t] = 0.1
                        scores[b, t, classes[e["label"]]] = 0.8
                        loc[b, 2 * t:2 * t + 2] = e["xy"]
            return scores.argmax(-1), scores, loc

    stub = Stub()
    orig = ds.batches

    def batches(*a, **k):
        for b in orig(*a, **k):
            stub.videos = b["video"]
            yield b
    ds.batches = batches
    out = EV.evaluate(stub, ds, "VAL", classes, save_pred=str(tmp_path / "pred"), predict_location=True, quiet=True)
    assert abs(out - 1.0) < 1e-5                           # all three events found at the exact frame and pixel
    ev = json.load(open(tmp_path / "pred.json"))
    assert [(e["frame"], e["label"]) for e in ev[0]["events"]] == [(7, "serve"), (30, "set")] and ev[1]["events"][0]["frame"] == 20
    assert ev[0]["events"][1]["xy"] == [0.75, 0.25] and os.path.exists(tmp_path / "pred.recall.json.gz")
    assert EV.evaluate(stub, ds, "VAL", classes, predict_location=False, quiet=True) == 1.0


def test_scorer_cli_rescores_the_prediction_files(tmp_path, golden, capsys):
    """python -m spot_amd.score_cli: a model directory with pred-test.N / pred-val.N recall files and a truth file; the numbers
    are those of compute_mAPs before and after the windowed peak picking, the epoch is the newest test epoch's."""
    import gzip
    from spot_amd import score_cli as SC
    from spot_amd.util import eval as E, score as S
    z = golden("scores")
    truth, pred = json.loads(str(z["truth"])), json.loads(str(z["pred"]))
    d = tmp_path / "model"
    os.makedirs(d)
    for name in ("pred-test.3", "pred-test.7", "pred-val.7"):
        with gzip.open(d / (name + ".recall.json.gz"), "wt", encoding="ascii") as fp:
            json.dump(pred if name != "pred-test.3" else [], fp)
    json.dump(truth, open(tmp_path / "val.json", "w"))
    assert SC.newest_epoch(str(d), "test")[1] == 7
    out = SC.main([str(d), "-s", "val", "--truth", str(tmp_path / "val.json"), "-t", "0", "1", "2", "--nms_window", "2"])
    assert "pred-val.7.recall.json.gz" in capsys.readouterr().out
    assert out["raw"][0] == S.compute_mAPs(truth, pred, [0, 1, 2], quiet=True)[0]
    assert out["nms"][0] == S.compute_mAPs(truth, E.non_maximum_supression(pred, 2), [0, 1, 2], quiet=True)[0]
    assert out["nms"][0] != out["raw"][0]
    with pytest.raises(FileNotFoundError):
        SC.newest_epoch(str(tmp_path), "test")


def test_sliding_window_accumulator_and_sharding():
    from spot_amd import infer
    from spot_amd.util.postproc import ScoreAccumulator
    assert infer.num_windows(15000, 64, 64) == 235            # 25 fps x 10 min: the long-video inference case
    assert infer.num_windows(10, 64, 64) == 1
    all_b = infer.window_batches(15000, 64, 16)
    assert len(all_b) == 15 and sum(len(b) for b in all_b) == 235 and all_b[-1][-1] == 234 * 64
    shards = [infer.window_batches(15000, 64, 16, r, 8) for r in range(8)]
    flat = sorted(s for sh in shards for b in sh for s in b)
    assert flat == [i * 64 for i in range(235)]
    rng = np.random.RandomState(0)
    acc = ScoreAccumulator(100, 7)
    a, b = rng.rand(64, 7).astype(np.float32), rng.rand(64, 7).astype(np.float32)
    acc.add(0, a, rng.rand(64, 2).astype(np.float32)); acc.add(64, b, rng.rand(64, 2).astype(np.float32))
    sc, loc = acc.finish()
    assert np.array_equal(sc[:64], a) and np.array_equal(sc[64:], b[:36])
    # the last window's black frames: zero bytes for decoded input, (0 - mean) / std for input that is already normalised
    # (the reference pads before its Normalize transform, inference_on_mp4.py:523-524)
    u8 = infer.pad_last_window(torch.full((3, 3, 2, 2), 7, dtype=torch.uint8), 5)
    assert u8.shape[0] == 5 and int(u8[3:].max()) == 0 and int(u8[:3].min()) == 7
    fl = infer.pad_last_window(torch.ones(3, 3, 2, 2), 5)
    want = torch.tensor([-0.485 / 0.229, -0.456 / 0.224, -0.406 / 0.225])
    assert fl.shape[0] == 5 and torch.allclose(fl[3:, :, 0, 0], want.expand(2, 3)) and torch.equal(fl[:3], torch.ones(3, 3, 2, 2))
    assert infer.pad_last_window(fl, 5) is fl


def test_roofline_accounting_matches_the_layer_table():
    from spot_amd import roofline
    w = roofline.per_frame_work()
    assert abs(w["fwd_macs"] / 1e6 - 816.8) < 0.5            # backbone 796.41 + gate conv 9.74 + GRU 7.08 + loc MLP 3.54 + fc 0.01 MMAC / frame
    assert abs(w["step_flops"] / 1e9 - 4.90) < 0.01
    assert 55e6 < w["step_bytes"] < 65e6                        # ~60 MB / frame fwd+bwd


def _spawn_gloo(worker, world=2, attempts=3):
    """Run `worker(rank, world, port, queue)` in `world` processes; a rendezvous that fails (the free port found here can
    be taken before the ranks bind it) is tried again on another port."""
    import socket
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for attempt in range(attempts):
        q = ctx.Queue()
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = []
        try:
            res = [q.get(timeout=120) for _ in range(world)]
        except Exception:                       # noqa: BLE001 -- a rank died before reporting
            pass
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        if len(res) == world and all(p.exitcode == 0 for p in procs):
            return res
    raise AssertionError("gloo ranks failed %d times" % attempts)


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spot_amd import dp
    torch.manual_seed(rank)
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    red = dp.BucketReducer(dist, flat)
    for lo, hi in ((700, 1000), (400, 700), (100, 400), (0, 100)):     # the engine's bucket order: heads first
        red(lo, hi)
    red.wait()
    label = torch.tensor([[0, 1, 0, 3], [0, 0, 2, 0]]) if rank == 0 else torch.tensor([[0, 0, 0, 0], [5, 0, 0, 0]])
    den = dp.loss_denominators(label, 5.0)
    dist.all_reduce(den)
    q.put((rank, flat.clone(), den.clone()))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_bucketed_allreduce_and_global_denominators_gloo():
    res = _spawn_gloo(_dp_worker)
    want = torch.arange(1000, dtype=torch.float32) * 3
    for rank, flat, den in res:
        assert torch.equal(flat, want)
        # 16 frames, 4 events: weights 12*1 + 4*5 = 32
        assert den.tolist() == [32.0, 4.0]


class _PixelModel:
    """Scores and locations that depend on the frame's pixel value AND on the clip it sits in (its first pixel), so that
    overlapping clips disagree about a frame, as a real model's do."""

    def predict(self, clip):
        v = clip[:, :, 0, 0, 0].float().numpy()                       # [B, T]
        ctx = v[:, :1] * 0.01
        scores = np.stack([np.cos(v + ctx) ** 2, np.sin(v + ctx) ** 2 * 0.7, np.sin(2 * v + ctx) ** 2 * 0.4], -1).astype(np.float32)
        loc = np.stack([(v * 0.37 + ctx) % 1.0, (v * 0.11 + 2 * ctx) % 1.0], -1).astype(np.float32).reshape(v.shape[0], -1)
        return scores.argmax(-1), scores, loc


def _eval_dataset():
    from spot_amd import evaluate as EV
    classes = {"serve": 1, "set": 2}
    labels = [{"video": "v0", "num_frames": 70, "fps": 25, "events": [{"frame": 9, "label": "serve", "xy": [0.2, 0.5]},
                                                                      {"frame": 40, "label": "set", "xy": [0.7, 0.2]}]},
              {"video": "v1", "num_frames": 33, "fps": 25, "events": [{"frame": 20, "label": "set", "xy": [0.5, 0.5]}]}]
    frames = {l["video"]: torch.arange(1, l["num_frames"] + 1, dtype=torch.uint8).view(-1, 1, 1, 1).expand(-1, 3, 2, 2).contiguous() for l in labels}
    return classes, EV.VideoClipDataset(classes, labels, frames, clip_len=16, overlap_len=8)


def _eval_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spot_amd import evaluate as EV
    classes, ds = _eval_dataset()
    pred = EV.accumulate_distributed(_PixelModel(), ds, 2, True, dist, batch_size=2)
    m = EV.evaluate(_PixelModel(), ds, "VAL", classes, predict_location=True, dist=dist, batch_size=2, quiet=True)
    q.put((rank, {v: [a.copy() for a in p] for v, p in pred.items()}, float(m)))
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_evaluation_equals_the_single_process_loop_gloo():
    """Clips dealt over two ranks, accumulators summed: the same scores, support and last-writer locations (clips overlap by
    half here) as the reference's one loop over the dataset, and the same metric on both ranks."""
    from spot_amd import evaluate as EV
    classes, ds = _eval_dataset()
    want = EV.accumulate(_PixelModel(), ds, 2, True, batch_size=2)
    want_m = float(EV.evaluate(_PixelModel(), ds, "VAL", classes, predict_location=True, batch_size=2, quiet=True))
    res = _spawn_gloo(_eval_worker)
    for rank, pred, m in res:
        assert m == want_m
        for v in want:
            assert np.array_equal(pred[v][1], want[v][1]) and want[v][1].max() == 2
            np.testing.assert_allclose(pred[v][0], want[v][0], rtol=1e-6)
            assert np.array_equal(pred[v][2], want[v][2])


def test_train_driver_schedule_and_resume_rules(tmp_path):
    """spot_amd.train mirrors the reference's schedule (train_e2e_spatial.py:1014-1031) and resume rules (:936-981):
    `_fc_out` tensors are dropped and the rest loaded non-strictly; optimizer state is restored only on --resume and only
    when nothing was dropped."""
    from spot_amd import train as TR
    args = TR.get_args(["ds", "synthetic:8", "-m", "rny008_gsm", "-s", str(tmp_path), "--num_epochs", "5", "--warm_up_epochs", "2",
                        "--num_classes", "6", "--resume", "--criterion", "loss"])
    w = torch.nn.Parameter(torch.zeros(3))
    w.grad = torch.ones(3)
    opt = torch.optim.AdamW([{"params": w, "lr": 1e-3}])
    n, sched = TR.get_lr_scheduler(args, opt, num_steps_per_epoch=10)
    lrs = []
    for _ in range(50):
        lrs.append(opt.param_groups[0]["lr"]); opt.step(); sched.step()
    # both factors act from step 0 (ChainedScheduler): 1 % start; the cosine, whose period is the 30 post-warm-up steps,
    # reaches its zero at step 30 and turns up again
    assert n == 5 and abs(lrs[0] - 1e-5) < 1e-12 and max(lrs) < 1e-3
    assert int(np.argmin(lrs[1:])) + 1 == 30 and lrs[30] < 1e-9 and lrs[31] > lrs[30]

    class Stub:
        def load(self, sd, strict=True):
            self.sd, self.strict = sd, strict
    sd = {"_features.stem.conv.weight": torch.ones(2), "_pred_fine._fc_out._fc_out.weight": torch.ones(7, 4), "_pred_fine._fc_out._fc_out.bias": torch.ones(7)}
    torch.save(sd, tmp_path / "checkpoint_003.pt")
    opt.step()
    state = {"optimizer_state_dict": opt.state_dict(), "scaler_state_dict": {}, "lr_state_dict": sched.state_dict()}
    torch.save(state, tmp_path / "optim_003.pt")
    with open(tmp_path / "loss.json", "w") as fp:
        json.dump([{"epoch": 2, "train": 1.0, "val": 0.7, "val_mAP": 0}, {"epoch": 3, "train": 0.9, "val": 0.8, "val_mAP": 0}], fp)
    assert TR.get_last_epoch(str(tmp_path)) == 3
    m = Stub()
    opt2 = torch.optim.AdamW([{"params": torch.nn.Parameter(torch.zeros(3)), "lr": 1e-3}])
    _, sched2 = TR.get_lr_scheduler(args, opt2, 10)
    epoch, losses, best_epoch, best = TR.load_from_save(args, m, opt2, None, sched2)
    assert (epoch, best_epoch, best, len(losses)) == (3, 2, 0.7, 2)
    assert m.strict is False and list(m.sd) == ["_features.stem.conv.weight"]
    assert len(opt2.state_dict()["state"]) == 0                                      # filtered => optimizer state not restored
    torch.save({k: v for k, v in sd.items() if "_fc_out" not in k}, tmp_path / "checkpoint_003.pt")
    TR.load_from_save(args, m, opt2, None, sched2)
    assert len(opt2.state_dict()["state"]) == 1 and sched2.state_dict()["_schedulers"][0]["last_epoch"] == 50


def test_optimizer_file_scaler_state_loads_into_an_enabled_grad_scaler():
    """The reference resumes optimizer, GradScaler and scheduler inside one `try` (train_e2e_spatial.py:966-976) with an
    ENABLED scaler, which refuses an empty state dict: the disabled scaler of the bf16 path is saved as a loadable state."""
    from spot_amd import train as TR
    sd = TR.scaler_state(torch.amp.GradScaler("cpu", enabled=False))
    enabled = torch.amp.GradScaler("cpu")
    with pytest.raises(RuntimeError):
        enabled.load_state_dict({})
    enabled.load_state_dict(sd)
    assert enabled.state_dict()["scale"] == 65536.0
    assert TR.scaler_state(enabled) == enabled.state_dict() and TR.scaler_state(None) == sd


def test_mixup_loader_follows_the_reference_recipe():
    """reference dataset/frame.py:338-353: label distribution l * onehot(a) + (1 - l) * onehot(b), frames blended with the same
    weight after normalisation, contains_event the larger of the two, xy the first clip's; l ~ Beta(0.2, 0.2) from `random`."""
    import random
    from spot_amd.train import MixupClips
    g = torch.Generator().manual_seed(3)
    B, T, nc = 3, 5, 6
    batch = {"frame": torch.randint(0, 256, (B, T, 3, 4, 4), generator=g, dtype=torch.uint8),
             "label": torch.tensor([[0, 2, 0, 0, 5], [0, 0, 0, 0, 0], [1, 1, 0, 3, 0]]), "xy": torch.rand(B, T, 2, generator=g)}
    out = next(iter(MixupClips([batch], nc + 1, seed=11)))        # (the model's class count: events + background)
    rng = random.Random(11)
    lam = [rng.betavariate(0.2, 0.2) for _ in range(B)]
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    norm = (batch["frame"].float() / 255 - mean) / std
    for i in range(B):
        j = (i - 1) % B                                           # the neighbour clip
        want = np.zeros((T, nc + 1))                              # the reference's own lines
        want[range(T), batch["label"][i].numpy()] = lam[i]
        want[range(T), batch["label"][j].numpy()] += 1.0 - lam[i]
        np.testing.assert_allclose(out["label"][i].numpy(), want, atol=1e-6)
        np.testing.assert_allclose(out["frame"][i].numpy(), (lam[i] * norm[i] + (1.0 - lam[i]) * norm[j]).numpy(), atol=1e-5)
    assert torch.allclose(out["label"].sum(-1), torch.ones(B, T), atol=1e-6)
    assert out["contains_event"].tolist() == [1, 1, 1] and torch.equal(out["xy"], batch["xy"])
    # the label layout E2EModel.epoch flattens for the soft-label cross entropy (train_e2e_spatial.py:546-550)
    assert out["label"].view(-1, out["label"].shape[-1]).shape == (B * T, nc + 1)


# ------------------------------------------------------------------ JPEG frame directories (reference dataset/frame.py:20-65, 156-333)
def _jpeg_tree(tmp_path, videos):
    """frame_dir/<video>/%06d.jpg with frame f of video v filled with the grey level 10 + 3 f + 40 v (flat images survive JPEG
    within a level or two), and the label list in the reference's JSON layout."""
    from PIL import Image
    labels = []
    for v, (name, n, events, missing) in enumerate(videos):
        d = tmp_path / "frames" / name
        d.mkdir(parents=True)
        for f in range(n):
            if f in missing:
                continue
            img = np.full((24, 32, 3), 10 + 3 * f + 40 * v, np.uint8)
            img[:, :, 1] //= 2                      # channels differ, so a dropped channel shows
            Image.fromarray(img).save(str(d / ("%06d.jpg" % f)), quality=95)
        labels.append({"video": name, "num_frames": n, "fps": 25,
                       "events": [{"frame": f, "label": l, "xy": [0.25, 0.5]} for f, l in events]})
    return str(tmp_path / "frames"), labels


def test_frame_reader_pads_like_the_reference(tmp_path):
    from spot_amd.dataset import FrameReader
    frame_dir, _ = _jpeg_tree(tmp_path, [("v0", 12, [], {7})])
    rd = FrameReader(frame_dir, "rgb")
    # which frame a decoded image is (JPEG moves a level by one or two; frames are three levels apart); -1: a zero frame
    level = lambda t: -1 if int(t.max()) == 0 else int(round((float(t[0].float().mean()) - 10) / 3))
    x = rd.load_frames("v0", -3, 5, pad=True)
    assert x.dtype == torch.uint8 and tuple(x.shape) == (8, 3, 24, 32)
    assert int(x[:3].max()) == 0 and [level(f) for f in x[3:]] == [0, 1, 2, 3, 4]                 # frames before 0: zero frames in front
    # an unreadable frame in the middle: the later frames move up, the hole is a zero frame at the END (and only with pad)
    y = rd.load_frames("v0", 5, 10, pad=True)
    assert [level(f) for f in y] == [5, 6, 8, 9, -1]
    assert tuple(rd.load_frames("v0", 5, 10, pad=False).shape) == (4, 3, 24, 32)
    assert tuple(rd.load_frames("v0", -2, 6, pad=False).shape) == (8, 3, 24, 32)                  # the front is padded regardless
    assert [level(f) for f in rd.load_frames("v0", 0, 12, pad=True, stride=4)] == [0, 4, 8]
    assert abs(float(x[3][1].float().mean()) - 5) <= 2 and abs(float(x[3][2].float().mean()) - 10) <= 2   # RGB order kept
    assert tuple(FrameReader(frame_dir, "flow").load_frames("v0", 0, 2).shape) == (2, 2, 24, 32)  # flow: the G and B planes
    with pytest.raises(RuntimeError):
        rd.read_frame(os.path.join(frame_dir, "v0", "000007.jpg"))
    random.seed(3)
    jit = [level(f) for f in rd.load_frames("v0", 0, 6, stride=3, randomize=True)]
    random.seed(3)
    assert jit == [0 + random.randint(0, 2), 3 + random.randint(0, 2)]                           # one draw per frame, in order


def test_clip_sampler_draws_in_the_reference_order_and_labels_clips(tmp_path):
    from spot_amd.dataset import ActionSpotDataset
    frame_dir, labels = _jpeg_tree(tmp_path, [("a", 40, [(4, "x"), (30, "y")], set()), ("b", 25, [(10, "y")], set())])
    classes = {"x": 1, "y": 2}
    ds = ActionSpotDataset(classes, labels, frame_dir, "rgb", 16, 50, is_eval=True, dilate_len=1, dataset="vnl_2.0")
    random.seed(11)
    got = [ds[0] for _ in range(6)]
    random.seed(11)
    for item in got:          # uniform sampling: one weighted video choice, then one start offset in [-pad, len - 1 + pad - clip_len]
        meta = random.choices(labels, weights=ds._weights_by_length)[0]
        start = -5 + random.randint(0, max(0, meta["num_frames"] - 1 + (2 * 5 - 16)))
        assert (item["video"], item["start"], item["end"]) == (meta["video"], start, start + 16)
        want = np.zeros(16, np.int64)
        xy = np.zeros((16, 2), np.float32)
        for e in meta["events"]:
            for i in range(e["frame"] - start - 1, e["frame"] - start + 2):        # dilated by one frame either side
                if 0 <= i < 16:
                    want[i] = classes[e["label"]]
                    xy[i] = (0.25, 0.5)
        assert np.array_equal(item["label"], want) and np.array_equal(item["xy"], xy)
        assert item["contains_event"] == int(want.sum() > 0)
        assert item["frame"].dtype == torch.uint8 and tuple(item["frame"].shape) == (16, 3, 24, 32)
        k = max(0, -start)                                                        # zero frames in front of the video's first
        assert int(item["frame"][:k].max() if k else 0) == 0
        v = 0 if meta["video"] == "a" else 1
        assert abs(float(item["frame"][k][0].float().mean()) - (10 + 3 * (start + k) + 40 * v)) <= 2
    # foreground sampling: (almost) every clip is drawn around an event and holds it
    fg = ActionSpotDataset(classes, labels, frame_dir, "rgb", 16, 50, fg_upsample=1e-9, dataset="vnl_2.0")
    random.seed(5)
    assert all(fg[0]["contains_event"] == 1 for _ in range(20))
    # datasets without locations hand out xy = None; mixup is the batch wrapper's job
    assert ActionSpotDataset(classes, labels, frame_dir, "rgb", 16, 50, dataset="finediving")[0]["xy"] is None
    with pytest.raises(NotImplementedError):
        ActionSpotDataset(classes, labels, frame_dir, "rgb", 16, 50, mixup=True)


def test_jpeg_loader_batches_and_video_view(tmp_path):
    from spot_amd.dataset import ActionSpotDataset, FrameReader, JpegClips, JpegVideo
    from spot_amd.evaluate import VideoClipDataset
    frame_dir, labels = _jpeg_tree(tmp_path, [("a", 40, [(4, "x"), (30, "y")], set()), ("b", 25, [(10, "y")], {20})])
    classes = {"x": 1, "y": 2}
    ds = ActionSpotDataset(classes, labels, frame_dir, "rgb", 8, 13, is_eval=False, dataset="vnl_2.0")
    ld = JpegClips(ds, 4)
    assert len(ld) == 3
    first = list(ld)
    assert [tuple(b["frame"].shape) for b in first] == [(4, 8, 3, 24, 32)] * 3
    assert first[0]["frame"].dtype == torch.uint8 and first[0]["label"].dtype == torch.int64 and tuple(first[0]["xy"].shape) == (4, 8, 2)
    ld.epoch = 0                                          # the same epoch again: the same clips; the next epoch: others
    again, nxt = list(ld), list(ld)
    assert all(torch.equal(a["start"], b["start"]) and a["video"] == b["video"] for a, b in zip(first, again))
    assert any(not torch.equal(a["start"], b["start"]) or a["video"] != b["video"] for a, b in zip(first, nxt))
    other = list(JpegClips(ds, 4, rank=1, world=2))
    assert len(other) == 1 and (other[0]["video"] != first[0]["video"] or not torch.equal(other[0]["start"], first[0]["start"]))
    two = list(JpegClips(ds, 4, num_workers=2))            # decoding workers: same batch format
    assert len(two) == 3 and tuple(two[0]["frame"].shape) == (4, 8, 3, 24, 32)
    # whole-video validation clips cut from the frame directory equal the clips cut from the decoded array
    rd = FrameReader(frame_dir, "rgb")
    lazy = {l["video"]: JpegVideo(rd, l["video"], l["num_frames"]) for l in labels}
    full = {l["video"]: rd.load_frames(l["video"], 0, l["num_frames"], pad=True) for l in labels}
    assert lazy["b"].shape == (25, 3, 24, 32) and tuple(full["b"].shape) == (25, 3, 24, 32)
    a = VideoClipDataset(classes, labels, lazy, 8)
    b = VideoClipDataset(classes, labels, full, 8)
    assert len(a) == len(b)
    for i in range(len(a)):
        x, y = a[i], b[i]
        assert (x["video"], x["start"]) == (y["video"], y["start"])
        if x["video"] == "a":                              # (video b has an unreadable frame: per-clip and whole-video reads put its zero frame in different places)
            assert torch.equal(x["frame"], y["frame"])


def test_training_driver_finds_a_jpeg_frame_directory(tmp_path, monkeypatch):
    from spot_amd import train
    frame_dir, labels = _jpeg_tree(tmp_path, [("a", 40, [(4, "x"), (30, "y")], set()), ("b", 25, [(10, "y")], set())])
    (tmp_path / "data" / "vnl_tiny").mkdir(parents=True)
    for split in ("train", "val"):
        (tmp_path / "data" / "vnl_tiny" / (split + ".json")).write_text(json.dumps(labels))
    (tmp_path / "data" / "vnl_tiny" / "class.txt").write_text("x\ny\n")
    monkeypatch.chdir(tmp_path)
    args = train.get_args(["vnl_tiny", frame_dir, "-m", "rny008_gsm", "-s", str(tmp_path / "out"), "--clip_len", "8", "--batch_size", "4",
                           "--debug_only", "-j", "0", "--dilate_len", "1", "--fg_upsample", "0.5", "--device", "cpu"])
    tr, va = train.get_loaders(args, 3)
    assert len(tr) == 500 // 8 // 4 and len(va) == max(4, 500 // 8 // 4) // 4
    batch = next(iter(tr))
    assert tuple(batch["frame"].shape) == (4, 8, 3, 24, 32) and batch["frame"].dtype == torch.uint8 and tuple(batch["xy"].shape) == (4, 8, 2)
    vids = train.get_val_videos(args, {"x": 1, "y": 2})
    assert [v for v, _, _ in vids.videos] == ["a", "b"] and tuple(vids[0]["frame"].shape) == (8, 3, 24, 32)
