"""Whole hot path on the GPU (through E2EModel -> engine -> C-ABI) against
 (1) the golden vectors produced by the reference's own modules (tests/golden/full_*.npz, fp32), and
 (2) the CPU oracle with bf16 storage rounding switched on (same rounding points as the kernels).
Stated tolerance (north star: "logits and (x,y) within a stated fp32/bf16 tolerance"): the path stores activations
in bf16 (8 significant bits, 2^-8 relative per rounding) through 14 bottleneck blocks; relative L2 error of the
features/logits/loc vs the fp32 reference must stay below 1e-2 in eval mode and 3e-2 in training mode (batch
statistics over 8 frames of a 2x2 map amplify rounding noise: the CPU oracle run in fp32 and with bf16 storage
rounding differs from ITSELF by 2e-2 on this case); gradients (relative L2 per tensor) below 8e-2 vs fp32 for tensors
that carry signal."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


# per-tensor gradient criterion of the 128x128 composite: for every tensor with signal whose fp32 gradient the CPU oracle
# itself reproduces within TIGHT_YARD under bf16 storage rounding (ReLU-mask flips decide the rest), cosine with and
# relative L2 distance from the fp32 gradient
TIGHT_YARD, TIGHT_COS, TIGHT_REL = 0.10, 0.95, 0.15


def rel(a, b):
    a, b = torch.as_tensor(a).double().flatten(), torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-12))


def build(arch="rny008_gsm", tarch="gru", clip_len=4, seed=3):
    from oracle import stes_oracle as O
    from spot_amd.e2e import E2EModel
    om = O.Impl(7, arch, tarch, clip_len=clip_len, predict_location=True)
    O.randomize_(om, seed=seed)
    m = E2EModel(7, arch, tarch, clip_len=clip_len, modality="rgb", device="cuda", predict_location=True, verbose=False)
    missing = m._model.load_state_dict(om.state_dict(), strict=True)     # state-dict keys are the compatibility surface
    return om, m


def test_state_dict_keys_match_oracle_and_timm_names():
    om, m = build()
    ks = set(m.state_dict().keys())
    assert ks == set(om.state_dict().keys())
    for k in ("_features.stem.conv.weight", "_features.s1.b1.conv1.gsm.conv3D.weight", "_features.s1.b1.conv1.net.conv.weight",
              "_features.s3.b4.se.fc1.bias", "_features.s2.b1.downsample.bn.running_var", "_pred_fine._gru.weight_hh_l0_reverse",
              "_pred_fine._fc_out._fc_out.bias", "_pred_loc.0.layers.1.weight", "_pred_loc.1.bias"):
        assert k in ks, k


def test_eval_forward_vs_reference_golden(golden):
    z = golden("full_rny008_gsm_gru")
    om, m = build()
    x = torch.from_numpy(z["x"])
    m._model.eval()
    with torch.no_grad():
        d = m._model(x.cuda())
    for k in ("cnn_feat", "im_feat", "loc_feat"):
        r = rel(d[k].cpu(), z["eval." + k])
        print(k, "rel L2 vs reference fp32:", r)
        assert r < 1e-2, (k, r)
    cls, scores, loc = m.predict(x)
    from oracle import stes_oracle as O
    ocls, oscores, oloc = O.predict(om, x)
    assert scores.shape == oscores.shape and loc.shape == oloc.shape
    np.testing.assert_allclose(scores, oscores, atol=2e-2)
    np.testing.assert_allclose(loc, oloc, atol=2e-2)


@pytest.mark.parametrize("arch,tarch,n_blocks", [("rny008_gsm", "gru", 7), ("rny008_tsm", "mingru", 7), ("rny008", "gru", 7)])
def test_eval_forward_224_fused_stage3_blocks_match_the_kernel_sequence(arch, tarch, n_blocks):
    """At 224 x 224 the seven identity bottlenecks of stage 3 (14 x 14 x 320) run as one kernel each in evaluation mode
    (stes_block_eval_s3), and conv1 + conv2 of the blocks of stages 1 and 2 as one kernel each (stes_block_front_eval), in every
    variant (without a gated shift there is no substituted slice).  Same model, same clip, with those paths on and
    off: the features agree to bf16 rounding noise (both paths round at the same points; fp32 summation order differs), far
    inside the 1e-2 stated against fp32."""
    from spot_amd import lib
    om, m = build(arch, tarch, clip_len=4)
    eng = m._model.engine
    x = torch.rand(2, 4, 3, 224, 224, generator=torch.Generator().manual_seed(5)).cuda()
    m._model.eval()
    out = {}
    for fused in (True, False):
        eng.block_fused = fused
        lib.TIMING = []
        try:
            with torch.no_grad():
                d = m._model(x)
            torch.cuda.synchronize()
            launches = sum(1 for rec in lib.TIMING if rec[0].startswith("block_eval_s3"))
            fronts = sum(1 for rec in lib.TIMING if rec[0].startswith("block_front"))
        finally:
            lib.TIMING = None
        assert launches == (n_blocks if fused else 0) and fronts == (5 if fused else 0), (launches, fronts)
        out[fused] = {k: d[k].float().cpu() for k in ("cnn_feat", "im_feat", "loc_feat")}
    eng.block_fused = True
    for k in out[True]:
        r = rel(out[True][k], out[False][k])
        print(k, "fused vs kernel sequence, rel L2:", r)
        # (either path is 5-6e-3 away from the fp32 reference on these tensors, test_eval_forward_vs_reference_golden: two bf16
        # paths that differ in summation order sit within that noise of each other; the location output, a small-norm vector
        # behind three more layers, shows the largest ratio)
        assert r < (1e-2 if k == "loc_feat" else 4e-3), (k, r)


@pytest.fixture
def deterministic_mode(request):
    """Switches the library's deterministic mode on for tests parametrised with det=True (and always off again)."""
    from spot_amd import lib
    on = bool(getattr(request, "param", False))
    if on:
        lib.set_deterministic(True)
    try:
        yield on
        if on:
            assert int(lib.deterministic_ticket()[1]) == 0, "a workgroup's wait for its turn ran out: the order was not kept"
    finally:
        if on:
            lib.set_deterministic(False)


@pytest.mark.parametrize("arch,tarch,tag,clip_len,deterministic_mode", [
    ("rny008_gsm", "gru", "full_rny008_gsm_gru", 4, False), ("rny008_tsm", "mingru", "full_rny008_tsm_mingru", 4, False),
    ("rny008_gsm", "gru", "full_rny008_gsm_gru_128", 8, False),
    # the same criteria with every float accumulation in a fixed order (one reproducible realisation instead of a draw)
    ("rny008_gsm", "gru", "full_rny008_gsm_gru_128", 8, True), ("rny008_tsm", "mingru", "full_rny008_tsm_mingru", 4, True)],
    indirect=["deterministic_mode"])
def test_train_forward_backward_vs_reference_and_oracle(golden, arch, tarch, tag, clip_len, deterministic_mode):
    """Training-mode outputs, loss and every parameter gradient against the reference's own fp32 run (golden) -- for the
    plain temporal shift this includes the autograd consequences of the reference's in-place shift (model/shift.py:55-78):
    unshifted shortcut gradient, ReLU mask taken from the shifted tensor.  The 128x128 case (16 frames, 4x4 final map) is
    the one where bf16 storage noise flips few ReLU masks per channel: it carries the tight per-tensor criterion."""
    from oracle import stes_oracle as O
    z = golden(tag)
    tight = clip_len == 8
    om, m = build(arch, tarch, clip_len=clip_len)
    x, label, xy = torch.from_numpy(z["x"]), torch.from_numpy(z["label"]), torch.from_numpy(z["xy"])
    impl = m._model
    impl.train()
    if tarch == "gru":
        impl._pred_fine._dropout.p = 0.0
        impl.engine.drop_p = 0.0
    d = impl(x.cuda())
    fwd_dev = {k: rel(d[k].detach().cpu(), z["train." + k]) for k in ("cnn_feat", "im_feat", "loc_feat")}
    # reference-style loss with torch ops on top of the engine outputs, backward through the autograd bridge
    loss, lc, ll = None, None, None
    import torch.nn.functional as F
    w = torch.tensor([1.0] + [5.0] * 6, device="cuda")
    lab = label.flatten().cuda()
    lc = F.cross_entropy(d["im_feat"].reshape(-1, 7), lab, weight=w)
    mask = (lab != 0).float()
    l1 = F.l1_loss(d["loc_feat"].reshape(-1, 2).sigmoid(), xy.reshape(-1, 2).cuda(), reduction="none").sum(-1)
    ll = (l1 * mask).sum() / (mask.sum() + 1e-6)
    np.testing.assert_allclose([lc.item(), ll.item()], z["loss"], rtol=3e-2)
    (lc + ll).backward()
    torch.cuda.synchronize()
    # oracle with the kernels' bf16 storage rounding
    O.set_storage_rounding(True)
    try:
        om.train()
        if tarch == "gru":
            om._pred_fine._dropout.p = 0.0
        od = om(x)
        oloss, _, _ = O.stes_loss(od["im_feat"], od["loc_feat"], label.flatten(), xy.reshape(-1, 2), 7)
        oloss.backward()
    finally:
        O.set_storage_rounding(False)
    for k in ("cnn_feat", "im_feat", "loc_feat"):
        # yardstick: what bf16 storage alone does to this output in the CPU oracle (training-mode BatchNorm over 8 frames
        # amplifies rounding noise; the location head's small outputs carry the most)
        yard = rel(od[k].detach(), z["train." + k])
        r = rel(d[k].detach().cpu(), od[k].detach())
        print(k, "train rel L2 vs reference fp32: %.4f (bf16 oracle: %.4f); vs bf16-rounded oracle: %.4f" % (fwd_dev[k], yard, r))
        assert fwd_dev[k] < max(3e-2, 2.0 * yard), (k, fwd_dev[k], yard)
        assert r < max(4e-2, 2.5 * yard), (k, r)
    # Gradients.  bf16 storage noise flips ReLU masks of near-zero pre-activations, and every flipped element changes
    # the masked gradient by its full size, so through ~45 ReLU layers the fp32 gradient cannot be matched tightly on a
    # batch of 8 frames: the CPU oracle run with bf16 storage rounding deviates from its own fp32 run by 20-40 % in
    # the backbone (2 % in the heads).  That self-deviation is the yardstick: per tensor, the HIP gradient must be as
    # close to the reference's fp32 gradient as the bf16-rounded oracle is (ratio of the two deviations: median < 1.3,
    # 95th percentile < 2, max < 3.5 over all tensors), with cosine > 0.75 for 95 % of the tensors and > 0.55 for all (the
    # tensors of the first block sit behind the whole chain of masks: about one run in ten one of them lands at 0.7).
    om32 = O.Impl(7, arch, tarch, clip_len=clip_len, predict_location=True)
    O.randomize_(om32, seed=3)
    om32.train()
    if tarch == "gru":
        om32._pred_fine._dropout.p = 0.0
    o32 = om32(x)
    l32, _, _ = O.stes_loss(o32["im_feat"], o32["loc_feat"], label.flatten(), xy.reshape(-1, 2), 7)
    l32.backward()
    g32 = dict((n, p.grad) for n, p in om32.named_parameters() if p.grad is not None)
    g16 = dict((n, p.grad) for n, p in om.named_parameters() if p.grad is not None)
    norms = dict(zip(z["grad_names"].tolist(), z["grad_norms"].tolist()))
    med = float(np.median(list(norms.values())))
    report, ratios, cosines = [], [], []
    for n, p in impl.named_parameters():
        if p.grad is None:
            continue
        ref = torch.from_numpy(z["g." + n]) if "g." + n in z.files else g32[n]     # reference's own fp32 gradient when stored
        # the stored reference norms pin the oracle's fp32 gradient for every tensor (fp32 summation order differs between
        # hosts and thread counts: 0.3 % was seen on a two-element bias gradient summed over 200,000 positions)
        assert abs(float(g32[n].double().norm()) - norms[n]) <= (1e-2 if tight else 2e-3) * norms[n] + 1e-7, n
        ours, yard = rel(p.grad.cpu(), ref), rel(g16[n], ref)
        a, b = p.grad.cpu().double().flatten(), ref.double().flatten()
        cosine = float(a @ b / (a.norm() * b.norm() + 1e-30))
        report.append((ours, yard, cosine, n))
        if norms[n] > 1e-2 * med and p.numel() >= 64:
            ratios.append(ours / max(yard, 2e-2))
            cosines.append(cosine)
            assert tight or cosine > 0.55, (n, cosine, ours, yard)
    ratios = np.array(ratios)
    if tight:
        sig = [r for r in report if norms[r[3]] > 1e-2 * med and dict(impl.named_parameters())[r[3]].numel() >= 64]
        quiet = [r for r in sig if r[1] < TIGHT_YARD]      # tensors whose fp32 gradient survives bf16 storage in the CPU oracle itself
        print("128x128 composite: %d tensors with signal, %d of them with a bf16-oracle deviation below %.2f" % (len(sig), len(quiet), TIGHT_YARD))
        print("  those: cosine min %.4f, rel L2 max %.4f;  all: cosine min %.4f p5 %.4f, rel L2 median %.4f" % (
            min(r[2] for r in quiet), max(r[0] for r in quiet), min(r[2] for r in sig), np.percentile([r[2] for r in sig], 5),
            np.median([r[0] for r in sig])))
        for r in quiet:
            assert r[2] > TIGHT_COS and r[0] < TIGHT_REL, r
        # everywhere else the CPU oracle under bf16 storage rounding is itself 20-50 % away from its fp32 run at this size too
        # (median 0.36: the random-init net stays sensitive to ReLU-mask flips however large the batch), so the criterion
        # there is the distribution of ours / yardstick below plus a cosine floor (the minimum over these tensors moves from
        # run to run with the order of the BatchNorm-sum atomics: 0.79, 0.82, 0.83, 0.85, 0.87 measured over five runs)
        assert len(quiet) >= 10 and min(r[2] for r in sig) > 0.7, sorted(sig, key=lambda r: r[2])[:4]
    assert np.percentile(cosines, 5) > 0.75, sorted(cosines)[:8]
    print("ours/yardstick: median %.2f  p95 %.2f  max %.2f over %d tensors" % (np.median(ratios), np.percentile(ratios, 95), ratios.max(), len(ratios)))
    # One noise realisation per run: the order of the fp32 atomics in the BatchNorm sums differs from run to run, that moves
    # pre-activations in the last bits, and on an 8-frame batch the ReLU masks that flip with them move backbone gradients by
    # 23-32 % between two runs of the SAME step -- measured identically with the weight-gradient kernels on the side stream
    # and in program order (tests/test_gpu_dp_world2.py), so it is not a stream hazard.  The criterion is therefore on the
    # distribution over the 250+ tensors.
    assert np.median(ratios) < 1.3 and np.percentile(ratios, 95) < 2.0 and ratios.max() < 3.5
    report.sort(reverse=True)
    print("largest gradient deviations (ours vs fp32, bf16-oracle vs fp32, cosine, name):")
    for r in report[:6]:
        print("   %.3f %.3f %.3f %s" % r)
    heads = [r for r in report if r[3].startswith("_pred_fine")]
    assert max(r[0] for r in heads) < 4e-2, heads
    # running statistics updated like torch BatchNorm
    for n, b in impl.named_buffers():
        if "buf." + n in z.files:
            np.testing.assert_allclose(b.cpu().numpy(), z["buf." + n], rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize("arch,tarch", [("rny008_gsm", "gru"), ("rny008_tsm", "mingru")])
def test_deterministic_mode_gives_bit_identical_steps(arch, tarch):
    """Run-to-run differences of a training step come from the order of fp32 atomics (BatchNorm sums, BatchNorm-backward
    sums, weight gradients) and from nothing else: with that order fixed (the library's deterministic mode) two runs of the
    same step give the same bits in every gradient, every BatchNorm buffer and the loss.  A race between kernels, a read of
    uninitialised scratch memory or a missing stream dependency would survive the mode and fail this test.  Without the
    mode the same two runs differ (which shows that the test can tell)."""
    from spot_amd import lib
    _, m = build(arch, tarch, seed=6)
    eng = m._model.engine
    if tarch == "gru":
        m._model._pred_fine._dropout.p = 0.0
        eng.drop_p = 0.0
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 4, 3, 64, 64, generator=g).cuda()
    label = torch.randint(0, 7, (2, 4), generator=g); label[0, 0] = 0; label[1, 1] = 3
    label, xy = label.cuda(), torch.rand(2, 4, 2, generator=g).cuda()
    sd = {k: v.clone() for k, v in m._model.state_dict().items()}

    def run():
        m._model.load_state_dict(sd)
        out = m.train_step(x, label, xy, lr=0.0, optimizer_step=False)
        torch.cuda.synchronize()
        bufs = torch.cat([b.detach().float().flatten() for _, b in m._model.named_buffers()])
        return eng.flat_grad.clone(), bufs.clone(), out[:3].clone()

    bits = lambda t: t.contiguous().view(torch.int32)
    lib.set_deterministic(True)
    try:
        a, b = run(), run()
        assert int(lib.deterministic_ticket()[1]) == 0
    finally:
        lib.set_deterministic(False)
    assert bool(torch.isfinite(a[0]).all()) and float(a[0].abs().max()) > 0
    for u, v, what in zip(a, b, ("gradients", "BatchNorm buffers", "loss")):
        same = torch.equal(bits(u), bits(v))
        if not same:
            bad = (bits(u) != bits(v)).nonzero().flatten()
            names = []
            if what == "gradients":
                for name, p in m._model.named_parameters():
                    o, n, _ = eng._goff[id(p)]
                    if bool(((bad >= o) & (bad < o + n)).any()):
                        names.append(name)
            raise AssertionError("%s differ between two deterministic runs in %d elements: %s" % (what, bad.numel(), names[:12]))
    # and the deterministic realisation is one of the ordinary ones: same loss, gradients as close as two ordinary runs are
    c, d = run(), run()
    cos = lambda p, q: float((p.double() * q.double()).sum() / (p.double().norm() * q.double().norm()))
    print("gradient cosine: ordinary/ordinary %.4f, deterministic/ordinary %.4f" % (cos(c[0], d[0]), cos(a[0], c[0])))
    assert not torch.equal(bits(c[0]), bits(d[0])), "two ordinary runs gave the same bits: the test cannot tell the modes apart"
    assert cos(a[0], c[0]) > min(0.9, cos(c[0], d[0]) - 0.05)
    assert abs(float(a[2][0] - c[2][0])) < 2e-2 * abs(float(c[2][0]))


def test_fused_train_step_matches_reference_step_semantics():
    """E2EModel.train_step (HIP loss + backward + AdamW) against the oracle's step (torch autograd + torch AdamW with the
    reference's per-tensor groups and lr/5 for _pred_loc)."""
    from oracle import stes_oracle as O
    om, m = build(seed=5)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 4, 3, 64, 64, generator=g)
    label = torch.randint(0, 7, (2, 4), generator=g); label[0, 0] = 0; label[1, 1] = 3
    xy = torch.rand(2, 4, 2, generator=g)
    m._model._pred_fine._dropout.p = 0.0
    m._model.engine.drop_p = 0.0
    om._pred_fine._dropout.p = 0.0
    opt = O.make_adamw(om, 1e-3)
    O.set_storage_rounding(True)
    try:
        ol = O.train_step(om, opt, x, label, xy, 7)
    finally:
        O.set_storage_rounding(False)
    out = m.train_step(x.cuda(), label.cuda(), xy.cuda(), lr=1e-3)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out[:3].cpu().numpy(), ol, rtol=2e-2)
    # after one AdamW step every weight moved by ~lr*sign(grad): compare the update direction on tensors with signal
    om0 = O.Impl(7, "rny008_gsm", "gru", clip_len=4, predict_location=True); O.randomize_(om0, seed=5)
    p0 = dict(om0.named_parameters())
    agree, total = 0, 0
    for n, p in m._model.named_parameters():
        du_ours = (p.detach().cpu() - p0[n].detach()).flatten()
        du_ref = (dict(om.named_parameters())[n].detach() - p0[n].detach()).flatten()
        sel = du_ref.abs() > 5e-4                     # well-determined updates (|m/sqrt(v)| ~ 1 at step 1)
        agree += int((torch.sign(du_ours[sel]) == torch.sign(du_ref[sel])).sum()); total += int(sel.sum())
    print("AdamW update sign agreement:", agree / max(total, 1), total)
    # the first AdamW update is lr*sign(g) elementwise; with the gradient noise quantified above (cosine ~0.92 in the
    # deepest layers) ~5 % of the signs differ between two bf16 runs
    assert agree / max(total, 1) > 0.9


def test_dp_trainer_over_rccl_world1_matches_plain_step():
    """The data-parallel trainer on a one-rank RCCL group: global loss denominators, bucketed asynchronous gradient
    all-reduce issued from inside backward (after the side stream's weight gradients), AdamW after the wait.  With one
    rank the sums are identities, so two steps must track the plain fused step."""
    import os
    import torch.distributed as dist
    from spot_amd import dp
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        _, m1 = build(seed=5)
        _, m2 = build(seed=5)
        for m in (m1, m2):
            m._model._pred_fine._dropout.p = 0.0
            m._model.engine.drop_p = 0.0
        g = torch.Generator().manual_seed(12)
        x = torch.randn(2, 4, 3, 64, 64, generator=g).cuda()
        label = torch.randint(0, 7, (2, 4), generator=g); label[0, 0] = 0; label[1, 1] = 3
        label, xy = label.cuda(), torch.rand(2, 4, 2, generator=g).cuda()
        trainer = dp.DataParallelTrainer(m2, dist, lr=1e-3)
        for _ in range(2):
            a = m1.train_step(x, label, xy, lr=1e-3).cpu()
            b = trainer.step(x, label, xy).cpu()
            np.testing.assert_allclose(a[:3].numpy(), b[:3].numpy(), rtol=3e-2)
            assert a[3:].tolist() == b[3:].tolist()             # class-weight sum and event count: exact
        g1, g2 = m1._model.engine.flat_grad.double(), m2._model.engine.flat_grad.double()
        cos = float((g1 * g2).sum() / (g1.norm() * g2.norm()))
        print("gradient cosine, plain vs data-parallel step:", cos)
        assert cos > 0.9
    finally:
        dist.destroy_process_group()


def test_side_stream_weight_gradients_match_program_order():
    """Weight-gradient kernels run on a side stream beside the input-gradient chain; with the overlap switched off they run
    in program order.  Both must give the same gradients (up to the run-to-run noise of the floating-point atomics on this
    8-frame batch): a missing stream dependency would show as stale or partial weight gradients."""
    from spot_amd import engine
    _, m = build(seed=6)
    m._model._pred_fine._dropout.p = 0.0
    m._model.engine.drop_p = 0.0
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 4, 3, 64, 64, generator=g).cuda()
    label = torch.randint(0, 7, (2, 4), generator=g); label[0, 0] = 0; label[1, 1] = 3
    label, xy = label.cuda(), torch.rand(2, 4, 2, generator=g).cuda()
    sd = {k: v.clone() for k, v in m._model.state_dict().items()}
    grads = {}
    try:
        for mode in (True, False, True):
            engine.OVERLAP = mode
            m._model.load_state_dict(sd)
            m.train_step(x, label, xy, lr=0.0, optimizer_step=False)
            torch.cuda.synchronize()
            gr = m._model.engine.flat_grad.double().clone()
            assert torch.isfinite(gr).all()
            grads.setdefault(mode, []).append(gr)
    finally:
        engine.OVERLAP = True
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()))
    noise = cos(grads[True][0], grads[True][1])          # two overlapped runs: the noise floor
    cross = cos(grads[True][0], grads[False][0])
    print("gradient cosine overlapped/overlapped %.4f, overlapped/program order %.4f" % (noise, cross))
    assert cross > min(0.9, noise - 0.05)
    # weight gradients specifically (the tensors the side stream writes): per-tensor norms agree
    eng = m._model.engine
    for name, p in m._model.named_parameters():
        if name.endswith("conv.weight") and "s3" in name:
            o, n, _ = eng._goff[id(p)]
            a, b = grads[True][0][o:o + n].norm(), grads[False][0][o:o + n].norm()
            assert abs(float(a - b)) < 0.5 * float(b) + 1e-9, name
