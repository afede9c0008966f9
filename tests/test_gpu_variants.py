"""Variants and harness on the GPU: plain-shift + minGRU (-m rny008_tsm -t mingru), 3-layer GRU, padded last window,
sliding-window inference with peak-picked events, and size-independent properties at the full benchmark size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).double().flatten(), torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _pair(arch, tarch, clip_len=4, seed=3):
    from oracle import stes_oracle as O
    from spot_amd.e2e import E2EModel
    om = O.Impl(7, arch, tarch, clip_len=clip_len, predict_location=True)
    O.randomize_(om, seed=seed)
    m = E2EModel(7, arch, tarch, clip_len=clip_len, modality="rgb", device="cuda", predict_location=True, verbose=False)
    m._model.load_state_dict(om.state_dict(), strict=True)
    return om, m


def test_tsm_mingru_eval_vs_reference_golden(golden):
    """rny008_tsm + mingru forward (reference TemporalShift in-place semantics: the shortcut sees the shifted tensor)."""
    z = golden("full_rny008_tsm_mingru")
    om, m = _pair("rny008_tsm", "mingru")
    m._model.eval()
    with torch.no_grad():
        d = m._model(torch.from_numpy(z["x"]).cuda())
    for k in ("cnn_feat", "im_feat", "loc_feat"):
        r = rel(d[k].cpu(), z["eval." + k])
        print(k, r)
        # the location MLP's small outputs carry 1.5 % noise already in the CPU oracle run with bf16 storage
        assert r < (2.5e-2 if k == "loc_feat" else 1e-2), (k, r)


def test_native_16x9_clip_forward_vs_oracle():
    """The reference's own inference shape: one clip of 64 frames at 398x224 (odd feature-map sizes 199, 100, 50, 25, 13
    down the stages), forward only, against the fp32 CPU oracle; time_backward flips the clip before the backbone and the
    predictions back (reference train_e2e_spatial.py:419-421,449-453)."""
    from oracle import stes_oracle as O
    om, m = _pair("rny008_gsm", "gru", clip_len=64, seed=11)
    x = torch.randn(1, 64, 3, 224, 398, generator=torch.Generator().manual_seed(12))
    torch.set_num_threads(16)
    om.eval()
    with torch.no_grad():
        od = om(x)
    m._model.eval()
    with torch.no_grad():
        d = m._model(x.cuda())
    for k in ("cnn_feat", "im_feat", "loc_feat"):
        r = rel(d[k].cpu(), od[k])
        print(k, r)
        assert r < (2.5e-2 if k == "loc_feat" else 1e-2), (k, r)


def test_eval_forward_on_small_maps_with_many_frames_vs_oracle():
    """64 x 64 crops, 8 clips of 64 frames: stage 3 has 4 x 4 maps, so a squeeze-excite gate row covers 16 rows of the last
    1x1 conv's operand, fewer than the streaming GEMM's block-tail form takes, while the row count (8,192) is one it would
    otherwise take: the forward must fall back to the GEMM + separate tail pass, not fail."""
    from oracle import stes_oracle as O
    om, m = _pair("rny008_gsm", "gru", clip_len=64, seed=21)
    x = torch.randn(8, 64, 3, 64, 64, generator=torch.Generator().manual_seed(22))
    torch.set_num_threads(16)
    om.eval()
    with torch.no_grad():
        od = om(x)
    m._model.eval()
    with torch.no_grad():
        d = m._model(x.cuda())
        d1 = m._model(x[:1].cuda())
    for k in ("cnn_feat", "im_feat", "loc_feat"):
        r = rel(d[k].cpu(), od[k])
        print(k, r)
        assert r < (2.5e-2 if k == "loc_feat" else 1e-2), (k, r)
    # clips are independent units
    assert rel(d1["im_feat"].cpu(), d["im_feat"][:1].cpu()) < 2e-3


@pytest.mark.parametrize("arch,tarch", [("rny008_gsm", "gru"), ("rny008_tsm", "mingru")])
def test_eval_forward_at_224_with_the_fused_blocks_vs_oracle(arch, tarch):
    """224 x 224 is where the evaluation forward runs its fused kernels (stes_block_front_eval for the first blocks of stages
    1-3 and the identity blocks of stage 2, stes_block_eval_s3 for the identity blocks of stage 3): the whole model against the
    CPU oracle in fp32, at the tolerance stated for the evaluation forward."""
    from spot_amd import lib
    om, m = _pair(arch, tarch, clip_len=4, seed=31)
    x = torch.randn(1, 4, 3, 224, 224, generator=torch.Generator().manual_seed(32))
    torch.set_num_threads(16)
    om.eval()
    with torch.no_grad():
        od = om(x)
    m._model.eval()
    lib.TIMING = []
    try:
        with torch.no_grad():
            d = m._model(x.cuda())
        torch.cuda.synchronize()
        fused = sorted(set(rec[0].split("<")[0] for rec in lib.TIMING if rec[0].startswith("block_")))
    finally:
        lib.TIMING = None
    assert fused == ["block_eval_s3_kernel", "block_front_kernel"], fused
    for k in ("cnn_feat", "im_feat", "loc_feat"):
        r = rel(d[k].cpu(), od[k])
        print(k, r)
        assert r < (2.5e-2 if k == "loc_feat" else 1e-2), (k, r)


def test_deeper_gru_eval_vs_oracle():
    from oracle import stes_oracle as O
    om, m = _pair("rny008_gsm", "deeper_gru")
    x = torch.randn(2, 4, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    cls, scores, loc = m.predict(x)
    ocls, oscores, oloc = O.predict(om, x)
    np.testing.assert_allclose(scores, oscores, atol=2e-2)
    np.testing.assert_allclose(loc, oloc, atol=2e-2)


def test_deeper_gru_training_gradients_of_the_head_vs_oracle():
    """-t deeper_gru (3 stacked bidirectional layers, reference train_e2e_spatial.py:379): one training forward/backward
    against the fp32 oracle -- loss, and the gradients of every GRU tensor (the recurrence through all three layers) and of
    the classifier.  Location targets are kept away from the L1 kink (see tools/gen_golden.py)."""
    from oracle import stes_oracle as O
    om, m = _pair("rny008_gsm", "deeper_gru", seed=8)
    for mod in (om, m._model):
        mod.train()
        mod._pred_fine._dropout.p = 0.0
    m._model.engine.drop_p = 0.0
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 4, 3, 64, 64, generator=g)
    label = torch.randint(0, 7, (2, 4), generator=g); label[0, 0] = 0; label[1, 1] = 3
    od = om(x)
    with torch.no_grad():
        s_ = od["loc_feat"].sigmoid()
        xy = (s_ + torch.where(torch.rand(2, 4, 2, generator=g) < 0.5, -0.2, 0.2)).clamp(0.0, 1.0)
    ol, _, _ = O.stes_loss(od["im_feat"], od["loc_feat"], label.flatten(), xy.reshape(-1, 2), 7)
    ol.backward()
    out = m.train_step(x.cuda(), label.cuda(), xy.cuda(), lr=0.0, optimizer_step=False).cpu()
    assert abs(float(out[0]) - float(ol.detach())) < 3e-2 * abs(float(ol.detach())), (out, ol)
    og = {n: p.grad for n, p in om.named_parameters() if p.grad is not None}
    worst = 0.0
    for n, p in m._model.named_parameters():
        if n.startswith("_pred_fine"):
            r = rel(p.grad.cpu(), og[n])
            worst = max(worst, r)
            assert r < 0.05, (n, r)          # bf16 features and weights feed the head: measured 0.015
    print("deeper_gru head gradients, worst rel. L2 vs fp32 oracle: %.3f" % worst)


def test_channel_attention_kernels_vs_reference_golden(golden):
    """The engine's kernel sequence for the clip-level channel gate on the reference's own vectors (ChannelAttention,
    model/modules.py:86-105, [3, 5, 64] features, reduction 8): output, input gradient, both weight gradients."""
    import ctypes as C
    from spot_amd import lib
    z = golden("channel_attention")
    dev, BF16 = "cuda", torch.bfloat16
    B, T, Fd, R = 3, 5, 64, 8
    t = lambda k, dt=torch.float32: torch.from_numpy(z[k]).to(dev).to(dt).contiguous()
    x, w1, w2, dy = t("x", BF16), t("w1", BF16), t("w2", BF16), t("dy")
    xq = x.float().cpu()
    p = lib.ptr
    pool, hid = torch.empty(B, Fd, dtype=BF16, device=dev), torch.empty(B, R, dtype=BF16, device=dev)
    gate, y = torch.empty(B, Fd, device=dev), torch.empty(B * T, Fd, dtype=BF16, device=dev)
    lib.call("stes_chan_pool", p(x), None, None, None, None, p(pool), B, T, Fd, 1.0 / T)
    lib.gemm(pool, w1, hid, B, R, Fd, act=1)
    lib.gemm(hid, w2, gate, B, Fd, R, act=2)
    lib.call("stes_clip_gate_fwd", p(x), p(gate), p(y), B, T, Fd)
    assert rel(y.cpu().view(B, T, Fd), z["y"]) < 1e-2
    dgate, dpool = torch.empty(B, Fd, device=dev), torch.empty(B, Fd, device=dev)
    dpre2, dpre1 = torch.empty(B, Fd, dtype=BF16, device=dev), torch.empty(B, R, dtype=BF16, device=dev)
    dw1, dw2, dx = torch.zeros(R, Fd, device=dev), torch.zeros(Fd, R, device=dev), torch.empty(B * T, Fd, device=dev)
    lib.call("stes_clip_gate_bwd", p(dy), None, p(x), None, None, p(dgate), None, B, T, Fd)
    np.testing.assert_allclose(dgate.cpu(), (torch.from_numpy(z["dy"]) * xq).sum(1), rtol=1e-5, atol=1e-5)
    lib.call("stes_sigmoid_bwd", p(dgate), p(gate), p(dpre2), B * Fd)
    lib.gemm(dpre2, w2.t().contiguous(), dpre1, B, R, Fd, relu_mask=hid, ldmask=R)
    lib.gemm(dpre1, w1.t().contiguous(), dpool, B, Fd, R, out_scale=1.0 / T)
    lib.wgrad(dpre2, hid, dw2, B, Fd, R)
    lib.wgrad(dpre1, pool, dw1, B, R, Fd)
    lib.call("stes_clip_gate_bwd", p(dy), None, None, p(gate), p(dpool), None, p(dx), B, T, Fd)
    torch.cuda.synchronize()
    for name, got, tol in (("dx", dx.view(B, T, Fd), 1e-2), ("dw1", dw1, 2e-2), ("dw2", dw2, 2e-2)):
        r = rel(got.cpu(), z[name])
        print(name, r)
        assert r < tol, (name, r)
    with pytest.raises(lib.StesError):            # neither the reduction's nor the apply pass's argument set
        lib.call("stes_clip_gate_bwd", p(dy), None, p(x), p(gate), p(dpool), p(dgate), p(dx), B, T, Fd)


def test_channel_attention_forward_and_gradients_vs_oracle():
    """--use_channel_attention (reference train_e2e_spatial.py:348-354,437-438): the per-clip channel gate between the
    backbone and the heads.  Eval predictions (also for a padded clip: the gate averages over the padded frames), then one
    training forward/backward against the fp32 oracle: loss, the gate MLP's two weight gradients, and the gradients on both
    sides of it (classifier, last backbone block)."""
    from oracle import stes_oracle as O
    from spot_amd.e2e import E2EModel
    om = O.Impl(7, "rny008_gsm", "gru", clip_len=4, predict_location=True, use_channel_attention=True)
    O.randomize_(om, seed=9)
    with torch.no_grad():                      # gates well away from 0.5 so that the attention matters
        om._channel_attention.fc[2].weight.mul_(3.0)
    m = E2EModel(7, "rny008_gsm", "gru", clip_len=4, modality="rgb", device="cuda", predict_location=True, verbose=False,
                 use_channel_attention=True)
    assert set(m._model.state_dict()) == set(om.state_dict())
    m._model.load_state_dict(om.state_dict(), strict=True)
    g = torch.Generator().manual_seed(41)
    for Tv in (4, 3):
        x = torch.randn(2, Tv, 3, 64, 64, generator=g)
        cls, scores, loc = m.predict(x)
        ocls, oscores, oloc = O.predict(om, x)
        np.testing.assert_allclose(scores, oscores, atol=2e-2)
        np.testing.assert_allclose(loc, oloc, atol=2e-2)
    om.eval(); m._model.eval()
    x = torch.randn(2, 4, 3, 64, 64, generator=g)
    with torch.no_grad():
        gate = torch.sigmoid(om._channel_attention.fc[2](torch.relu(om._channel_attention.fc[0](om._features(x.reshape(-1, 3, 64, 64)).reshape(2, 4, -1).mean(1)))))
    assert float(gate.std()) > 0.05                                  # not a constant scaling
    for mod in (om, m._model):
        mod.train()
        mod._pred_fine._dropout.p = 0.0
    m._model.engine.drop_p = 0.0
    x = torch.randn(2, 4, 3, 64, 64, generator=g)
    label = torch.randint(0, 7, (2, 4), generator=g); label[0, 0] = 0; label[1, 1] = 3
    od = om(x)
    with torch.no_grad():
        s_ = od["loc_feat"].sigmoid()
        xy = (s_ + torch.where(torch.rand(2, 4, 2, generator=g) < 0.5, -0.2, 0.2)).clamp(0.0, 1.0)
    ol, _, _ = O.stes_loss(od["im_feat"], od["loc_feat"], label.flatten(), xy.reshape(-1, 2), 7)
    ol.backward()
    og = {n: p.grad.clone() for n, p in om.named_parameters() if p.grad is not None}
    # yardstick: the same oracle with the kernels' bf16 storage rounding (see test_train_forward_backward_vs_reference_and_oracle)
    om.zero_grad()
    O.set_storage_rounding(True)
    try:
        od16 = om(x)
        l16, _, _ = O.stes_loss(od16["im_feat"], od16["loc_feat"], label.flatten(), xy.reshape(-1, 2), 7)
        l16.backward()
    finally:
        O.set_storage_rounding(False)
    g16 = {n: p.grad for n, p in om.named_parameters() if p.grad is not None}
    out = m.train_step(x.cuda(), label.cuda(), xy.cuda(), lr=0.0, optimizer_step=False).cpu()
    assert abs(float(out[0]) - float(ol.detach())) < 3e-2 * abs(float(ol.detach())), (out, ol)
    bad = []
    for n, p in m._model.named_parameters():
        if n.startswith("_channel_attention") or n.startswith("_pred_fine._fc_out") or n in (
                "_features.s4.b2.conv3.conv.weight", "_features.s4.b2.se.fc2.weight", "_pred_loc.0.layers.0.weight"):
            r, yard = rel(p.grad.cpu(), og[n]), rel(g16[n], og[n])
            print("%-40s ours vs fp32 %.3f   bf16-rounded oracle vs fp32 %.3f" % (n, r, yard))
            bad += [(n, r, yard)] if r > max(0.04, 2.5 * yard) else []
    assert not bad, bad


def test_short_clip_is_padded_then_unpadded_like_reference():
    """T < clip_len: frames are zero-padded for the backbone and the heads run on the true length
    (reference train_e2e_spatial.py:423-442)."""
    from oracle import stes_oracle as O
    om, m = _pair("rny008_gsm", "gru", clip_len=6)
    x = torch.randn(2, 4, 3, 64, 64, generator=torch.Generator().manual_seed(6))
    cls, scores, loc = m.predict(x)
    ocls, oscores, oloc = O.predict(om, x)
    assert scores.shape == (2, 4, 7) and loc.shape == (2, 4, 2)
    np.testing.assert_allclose(scores, oscores, atol=2e-2)
    np.testing.assert_allclose(loc, oloc, atol=2e-2)
    with pytest.raises(AssertionError):
        m.predict(torch.zeros(1, 7, 3, 64, 64))


def test_eval_forward_reuses_the_folded_batchnorm_until_something_changes():
    """Evaluation folds the running statistics into scale/shift once: a second forward launches no BatchNorm finalize and
    gives the same bits; an in-place change of a running statistic, a training step, or a loaded state dict is picked up."""
    from spot_amd import lib
    om, m = _pair("rny008_gsm", "gru", clip_len=4)
    x = torch.randn(2, 4, 3, 64, 64, generator=torch.Generator().manual_seed(8))

    def run():
        lib.TIMING = []
        try:
            _, scores, _ = m.predict(x)
            return scores, sum(t[0] == "bn_finalize_kernel" for t in lib.TIMING)
        finally:
            lib.TIMING = None
    s1, n1 = run()
    s2, n2 = run()
    assert n1 > 50 and n2 == 0 and np.array_equal(s1, s2)
    bn = m._model._features.stem.bn
    bn.running_var.mul_(9.0)                               # from outside, in place
    s3, n3 = run()
    assert n3 == n1 and not np.array_equal(s3, s1)
    bn.running_var.div_(9.0)
    s4, n4 = run()
    assert n4 == n1 and np.allclose(s4, s1, atol=1e-3)
    # a training step moves weights and statistics
    label = torch.zeros(2, 4, dtype=torch.long, device="cuda")
    m.train_step(x.cuda(), label, torch.rand(2, 4, 2, device="cuda"), 1e-3)
    s5, n5 = run()
    assert n5 == n1 and not np.array_equal(s5, s1)
    m._model.load_state_dict(om.state_dict(), strict=True)
    s6, n6 = run()
    assert n6 == n1 and np.array_equal(s6, s1)
    assert run()[1] == 0


def test_uint8_frames_equal_the_loaders_normalised_tensor():
    """Decoded uint8 clips at the boundary (`read_image`, reference dataset/frame.py:29) give the predictions of the fp32
    tensor the reference loader would have built from them (`/255`, Normalize :124), bit for bit in eval mode -- also for
    a short clip, whose zero frames are padding in normalised space (train_e2e_spatial.py:427-430) -- and train."""
    import torch.nn.functional as F
    _, m = _pair("rny008_gsm", "gru")
    g = torch.Generator().manual_seed(21)
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    m._model.eval()
    for Tv in (4, 3):
        u = torch.randint(0, 256, (2, Tv, 3, 64, 48), generator=g, dtype=torch.uint8)
        xf = (u.float() / 255 - mean[:, None, None]) / std[:, None, None]
        with torch.no_grad():
            a = {k: v.clone() for k, v in m._model(xf.cuda()).items()}
            b = m._model(u.cuda())
        for k in ("cnn_feat", "im_feat", "loc_feat"):
            assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), (Tv, k)
    cls_a, sc_a, loc_a = m.predict(xf)
    cls_b, sc_b, loc_b = m.predict(u)
    assert np.array_equal(cls_a, cls_b) and np.array_equal(sc_a, sc_b) and np.array_equal(loc_a, loc_b)
    # training step from uint8 clips: same loss as from the normalised tensor (the stem patches are identical; the rest of
    # a training step differs run to run only by the order of its floating-point atomics)
    u = torch.randint(0, 256, (2, 4, 3, 64, 48), generator=g, dtype=torch.uint8)
    xf = (u.float() / 255 - mean[:, None, None]) / std[:, None, None]
    label = torch.randint(0, 7, (2, 4), generator=g).cuda(); xy = torch.rand(2, 4, 2, generator=g).cuda()
    sd = {k: v.clone() for k, v in m._model.state_dict().items()}
    la = m.train_step(xf.cuda(), label, xy, lr=0.0).cpu()
    m._model.load_state_dict(sd)
    lb = m.train_step(u.cuda(), label, xy, lr=0.0).cpu()
    assert rel(lb[:3], la[:3]) < 2e-2, (la, lb)


def test_predict_async_returns_what_predict_returns_and_keeps_its_results():
    """predict_async (no read-back, used by the sliding-window loop to enqueue a batch before it reads the previous one): the
    same scores and locations as predict(), in tensors that a later forward does not overwrite."""
    om, m = _pair("rny008_gsm", "gru", clip_len=8, seed=3)
    g = torch.Generator().manual_seed(4)
    a = torch.randn(2, 8, 3, 64, 64, generator=g)
    b = torch.randn(2, 8, 3, 64, 64, generator=g)
    _, scores_a, loc_a = m.predict(a)
    sa, la = m.predict_async(a)
    sb, lb = m.predict_async(b)              # a second forward while the first results are still on the device
    torch.cuda.synchronize()
    assert np.array_equal(sa.cpu().numpy(), scores_a) and np.array_equal(la.cpu().numpy(), loc_a)
    _, scores_b, loc_b = m.predict(b)
    assert np.array_equal(sb.cpu().numpy(), scores_b) and np.array_equal(lb.cpu().numpy(), loc_b)
    assert np.array_equal(sa.cpu().numpy(), scores_a)          # still the first batch's


def test_sliding_window_events_match_oracle():
    """Long-video path: windows of clip_len frames, last one zero-padded, per-frame scores -> argmax events ->
    score filter 0.15 -> NMS(3).  Event frames and labels must be identical to the CPU oracle's on the same clip."""
    from oracle import stes_oracle as O
    from spot_amd import infer
    om, m = _pair("rny008_gsm", "gru", clip_len=8, seed=7)
    # make the classifier confident so that argmax margins exceed the bf16 noise (trained models have such margins)
    with torch.no_grad():
        om._pred_fine._fc_out._fc_out.weight.mul_(6.0)
    m._model.load_state_dict(om.state_dict())
    classes = ["c%d" % i for i in range(6)]
    frames = torch.randn(44, 3, 64, 64, generator=torch.Generator().manual_seed(8))     # 5 full windows + one of 4 frames
    ev, scores = infer.run_inference(m, frames, classes, clip_len=8, batch_size=4)
    # oracle through the same harness arithmetic
    wins = []
    for s in range(0, 44, 8):
        # the last window is filled up with black frames: (0 - mean) / std in an already normalised clip, because the reference
        # pads before its Normalize transform (inference_on_mp4.py:523-524)
        clip = infer.pad_last_window(frames[s:s + 8], 8)
        _, sc, lc = O.predict(om, clip)
        wins.append((s, sc[0], lc[0]))
    osc, oloc = O.accumulate_windows(44, 7, wins)
    oev = O.nms_events(O.filter_events_by_score(O.events_from_scores(osc, oloc, classes), 0.15), 3)
    margin = np.sort(osc, 1)[:, -1] - np.sort(osc, 1)[:, -2]
    print("min top-2 score margin in the oracle:", margin.min(), " max |score diff|:", np.abs(scores - osc).max())
    # the claim "identical event frames" is made where no argmax can flip: two scores that each move by at most the largest
    # score error e can only change order when they are no more than 2e apart
    err = np.abs(scores - osc).max()
    assert margin.min() > 2 * err
    got = [(e["frame"], e["label"]) for e in ev[0]["events"]]
    want = [(e["frame"], e["label"]) for e in oev]
    # ... and where non-maximum suppression is decisive by the same rule: an event may sit on another frame than the oracle's
    # only when both frames carry the label, lie within the suppression window and score within 2e of each other in the oracle
    only_got, only_want = sorted(set(got) - set(want)), sorted(set(want) - set(got))
    assert len(got) == len(want) and len(only_got) == len(only_want)
    for (fg, lg), (fw, lw) in zip(only_got, only_want):
        c = classes.index(lg) + 1
        assert lg == lw and abs(fg - fw) <= 3 and abs(osc[fg, c] - osc[fw, c]) <= 2 * err, ((fg, lg), (fw, lw), osc[fg, c], osc[fw, c], err)
    print("events on other frames than the oracle's (suppression ties within 2e):", list(zip(only_got, only_want)))
    # the two-GPU sharding of the windows gives the same result as one GPU (replicas only, host gather)
    parts = []
    for r in range(2):
        acc = infer.ScoreAccumulator(44, 7)
        for starts in infer.window_batches(44, 8, 2, r, 2):
            batch = torch.stack([infer.pad_last_window(frames[s:s + 8], 8) for s in starts])
            _, sc, lc = m.predict(batch)
            for i, s in enumerate(starts):
                acc.add(s, sc[i], lc[i])
        parts.append(acc)
    tot = infer.ScoreAccumulator(44, 7)
    tot.scores = parts[0].scores + parts[1].scores; tot.support = parts[0].support + parts[1].support
    assert np.array_equal(tot.support, np.ones(44, np.int32))
    np.testing.assert_allclose(tot.scores, scores, atol=1e-6)


def test_full_size_properties():
    """Benchmark size (16 clips x 64 frames x 224x224, 1024 frames): properties that need no oracle run.
    (1) conv3D = 0 -> the gated shift is the identity in source channel order, bit for bit;
    (2) training-mode BatchNorm: every normalised conv output has per-channel mean ~0 / variance ~1;
    (3) evaluation is deterministic (two runs bit-identical) and independent of the batch split (16 clips at once ==
        two halves), since nothing crosses a clip;
    (4) one fused optimisation step gives finite losses, changes the weights, and a second step lowers the loss."""
    from spot_amd.e2e import E2EModel
    torch.manual_seed(0)
    m = E2EModel(7, "rny008_gsm", "gru", clip_len=64, modality="rgb", device="cuda", predict_location=True, verbose=False)
    impl = m._model
    with torch.no_grad():
        for n, p in impl.named_parameters():
            if n.endswith("conv3.bn.weight"):
                p.fill_(0.25)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(16, 64, 3, 224, 224, device="cuda", generator=g)
    eng = impl.engine
    impl.eval()
    with torch.no_grad():
        a = impl(x)
        # (1) zero gate (the reference's init): G == Xc exactly, for the first and the last block
        for b in (eng.blocks[0], eng.blocks[-1]):
            assert torch.equal(b.G, b.Xc), b.name
            assert torch.equal(b.Xc.view(-1, b.fold), b.X.view(-1, b.cin)[:, :b.fold])
        a2 = impl(x)
        for k in a:
            assert torch.equal(a[k], a2[k]), k                    # (3) deterministic
        h1, h2 = impl(x[:8]), impl(x[8:])
        for k in a:
            assert torch.equal(torch.cat([h1[k], h2[k]]), a[k]), k    # (3) clips are independent
    # (2) batch statistics
    label = (torch.rand(16, 64, device="cuda", generator=g) < 0.03).long() * torch.randint(1, 7, (16, 64), device="cuda", generator=g)
    xy = torch.rand(16, 64, 2, device="cuda", generator=g)
    w0 = eng.flat.clone()
    out1 = m.train_step(x, label, xy, lr=1e-3).clone()
    for b in (eng.blocks[0], eng.blocks[5], eng.blocks[-1]):
        for raw, bn, cc in ((b.c1, b.bn1, b.cmid), (b.c3, b.bn3, b.cout)):
            rows = raw.numel() // cc        # random rows: a fixed stride aliases with the image width
            v = raw.view(-1, cc)[torch.randint(0, rows, (200000,), device="cuda", generator=g)].float() * bn.scale + bn.shift
            gam, bet = bn.mod.weight.detach(), bn.mod.bias.detach()
            zed = (v - bet) / gam
            assert float(zed.mean(0).abs().max()) < 0.05 and float((zed.var(0) - 1).abs().max()) < 0.1, b.name
    out2 = m.train_step(x, label, xy, lr=1e-3).clone()
    torch.cuda.synchronize()
    l1, l2 = out1[:3].cpu().numpy(), out2[:3].cpu().numpy()
    assert np.isfinite(l1).all() and np.isfinite(l2).all()
    assert not torch.equal(w0, eng.flat) and bool(torch.isfinite(eng.flat).all())
    assert l2[0] < l1[0], (l1, l2)                                   # (4)


def test_train_driver_runs_checkpoints_and_resumes(tmp_path):
    """spot_amd.train (counterpart of the reference's main, train_e2e_spatial.py:1034-1236) on synthetic uint8 clips: two
    epochs through E2EModel.epoch with a stock AdamW + chained schedule, the reference's files in save_dir, then --resume
    for a third epoch; a shard directory loads like the synthetic source."""
    import json
    import os
    from spot_amd import train as TR
    base = ["ds", "synthetic:8", "-m", "rny008_gsm", "-t", "gru", "--clip_len", "4", "--crop_dim", "64", "--batch_size", "2",
            "--warm_up_epochs", "1", "-lr", "0.002", "--predict_location", "-s", str(tmp_path), "--num_classes", "6",
            "--criterion", "loss"]
    model, losses, best = TR.main(TR.get_args(base + ["--num_epochs", "2"]))
    names = set(os.listdir(tmp_path))
    assert {"checkpoint_000.pt", "checkpoint_001.pt", "optim_001.pt", "loss.json", "config.json"} <= names and "optim_000.pt" not in names
    assert [l["epoch"] for l in losses] == [0, 1] and all(np.isfinite(l["train"]) and np.isfinite(l["val"]) for l in losses)
    sd = torch.load(tmp_path / "checkpoint_001.pt", weights_only=True)
    assert set(sd) == set(model.state_dict())
    model2, losses2, _ = TR.main(TR.get_args(base + ["--num_epochs", "3", "--resume"]))
    assert [l["epoch"] for l in losses2] == [0, 1, 2] and np.isfinite(losses2[2]["train"])
    # resumed from checkpoint_001 (minus the `_fc_out` tensors, which the reference always drops): a backbone tensor moved
    # on from the saved value by one more epoch of small steps, not from a fresh initialisation
    k = "_features.s1.b1.conv2.conv.weight"
    moved = (model2.state_dict()[k].cpu() - sd[k].cpu()).abs().max()
    assert 0 < float(moved) < 0.05, moved
    assert json.load(open(tmp_path / "config.json"))["num_classes"] == 6
    # the same data from a shard directory
    shard_dir = tmp_path / "shards"
    os.makedirs(shard_dir)
    for split, seed in (("train", 0), ("val", 1)):
        b = list(TR.SyntheticClips(4, 4, 4, 64, 7, seed))[0]
        np.savez(shard_dir / f"{split}_000.npz", frame=b["frame"].numpy(), label=b["label"].numpy(), xy=b["xy"].numpy())
    args = TR.get_args(["ds", str(shard_dir)] + base[2:-6] + ["-s", str(tmp_path / "run2"), "--num_classes", "6", "--num_epochs", "2", "--criterion", "loss"])
    _, losses3, _ = TR.main(args)
    assert len(losses3) == 2 and np.isfinite(losses3[1]["val"])


def test_train_driver_selects_the_best_epoch_by_validation_map(tmp_path):
    """--criterion map: from start_val_epoch on, whole validation videos go through evaluate() (the reference's
    train_e2e_spatial.py:1158-1178 and :656-827); the metric lands in loss.json and the predictions beside it.  The
    video directory format (val.json + val_videos/*.npy) gives the same numbers as the same data held in memory."""
    import json
    import os
    from spot_amd import evaluate as EV, train as TR
    base = ["ds", "synthetic:8", "-m", "rny008_gsm", "-t", "gru", "--clip_len", "8", "--crop_dim", "64", "--batch_size", "2",
            "--warm_up_epochs", "1", "-lr", "0.002", "--predict_location", "-s", str(tmp_path / "run"), "--num_classes", "6",
            "--criterion", "map", "--start_val_epoch", "1", "--num_epochs", "2"]
    args = TR.get_args(base)
    model, losses, best = TR.main(args)
    assert losses[0]["val_mAP"] == 0 and 0 <= losses[1]["val_mAP"] <= 1 and best == 1
    pred = json.load(open(tmp_path / "run" / "pred-val.1.json"))
    assert [p["video"] for p in pred] == ["synthetic0", "synthetic1"] and os.path.exists(tmp_path / "run" / "pred-val.1.recall.json.gz")
    assert json.load(open(tmp_path / "run" / "loss.json"))[1]["val_mAP"] == losses[1]["val_mAP"]
    # the same videos from files
    classes = TR.load_classes(args)
    mem = TR.get_val_videos(args, classes)
    d = tmp_path / "data"
    os.makedirs(d / "val_videos")
    json.dump(mem.labels, open(d / "val.json", "w"))
    for v, f in mem._frames.items():
        np.save(d / "val_videos" / (v + ".npy"), f.numpy())
    args.frame_dir = str(d)
    disk = TR.get_val_videos(args, classes)
    assert len(disk) == len(mem) == 6 and torch.equal(disk[4]["frame"], mem[4]["frame"]) and disk[4]["start"] == mem[4]["start"]
    a = EV.accumulate(model, mem, 6, True)
    b = EV.accumulate(model, disk, 6, True, batch_size=3)            # batch composition does not change a clip's scores
    for v in a:
        assert np.array_equal(a[v][1], b[v][1]) and a[v][1].min() == 1
        assert np.allclose(a[v][0], b[v][0], atol=2e-3) and np.allclose(a[v][2], b[v][2], atol=2e-3)
    m1 = EV.evaluate(model, mem, "VAL", classes, predict_location=True, quiet=True)
    assert abs(m1 - losses[1]["val_mAP"]) < 1e-9                  # the best epoch's weights were restored


def test_train_driver_reads_a_jpeg_frame_directory(tmp_path, monkeypatch):
    """The reference's own data layout (frame_dir/<video>/%06d.jpg + data/<dataset>/{train,val}.json + class.txt,
    dataset/frame.py:20-65,156-333, train_e2e_spatial.py:880-932): frames decoded on the host, handed over as uint8, resized
    and augmented on the GPU (--augment), validation mAP over whole videos cut from the frame directory clip by clip."""
    import json
    from PIL import Image
    from spot_amd import train as TR
    rng = np.random.default_rng(0)
    labels = []
    for v, n in (("a", 40), ("b", 30)):
        d = tmp_path / "frames" / v
        d.mkdir(parents=True)
        for f in range(n):
            Image.fromarray(rng.integers(0, 256, (72, 128, 3), dtype=np.uint8)).save(str(d / ("%06d.jpg" % f)), quality=90)
        labels.append({"video": v, "num_frames": n, "fps": 25,
                       "events": [{"frame": f, "label": "xy"[f % 2], "xy": [0.3, 0.6]} for f in range(3, n, 9)]})
    (tmp_path / "data" / "vnl_tiny").mkdir(parents=True)
    for split in ("train", "val"):
        (tmp_path / "data" / "vnl_tiny" / (split + ".json")).write_text(json.dumps(labels))
    (tmp_path / "data" / "vnl_tiny" / "class.txt").write_text("x\ny\n")
    monkeypatch.chdir(tmp_path)
    args = TR.get_args(["vnl_tiny", str(tmp_path / "frames"), "-m", "rny008_gsm", "-t", "gru", "--clip_len", "8", "--crop_dim", "64",
                        "--batch_size", "2", "--warm_up_epochs", "1", "-lr", "0.002", "--predict_location", "-s", str(tmp_path / "run"),
                        "--criterion", "map", "--start_val_epoch", "1", "--num_epochs", "2", "--debug_only", "-j", "0", "--augment",
                        "--dilate_len", "1", "--fg_upsample", "0.5"])
    model, losses, best = TR.main(args)
    assert len(losses) == 2 and all(np.isfinite(l["train"]) and np.isfinite(l["val"]) for l in losses) and 0 <= losses[1]["val_mAP"] <= 1
    pred = json.load(open(tmp_path / "run" / "pred-val.1.json"))
    assert [p["video"] for p in pred] == ["a", "b"]


def test_train_driver_fused_mode_and_cross_mode_resume(tmp_path):
    """--fused: epochs through the HIP loss/backward/AdamW step with the learning rate of the reference's schedule; the
    optimizer file keeps torch.optim.AdamW's format (moments and step count per parameter), so the run resumes through the
    reference-style loop, and a reference-style run resumes fused.  (With the minGRU head: a checkpoint that holds `_fc_out`
    tensors -- any GRU head -- makes the reference, and this driver, skip the optimizer state on resume.)"""
    import os
    from spot_amd import train as TR
    base = ["ds", "synthetic:8", "-m", "rny008_gsm", "-t", "mingru", "--clip_len", "4", "--crop_dim", "64", "--batch_size", "2",
            "--warm_up_epochs", "1", "-lr", "0.002", "--predict_location", "-s", str(tmp_path), "--num_classes", "6",
            "--criterion", "loss"]
    model, losses, _ = TR.main(TR.get_args(base + ["--num_epochs", "2", "--fused"]))
    assert [l["epoch"] for l in losses] == [0, 1] and all(np.isfinite(l["train"]) and np.isfinite(l["val"]) for l in losses)
    opt = torch.load(tmp_path / "optim_001.pt", weights_only=True)
    st = opt["optimizer_state_dict"]["state"]
    n_trainable = sum(p.requires_grad for p in model._model.parameters())
    assert len(st) == n_trainable and all(int(s["step"]) == 8 for s in st.values())          # 2 epochs x 4 steps
    eng = model._model.engine
    names = [n for n, _ in model._model.named_parameters()]
    i = names.index("_features.s2.b1.conv1.net.conv.weight")
    p = dict(model._model.named_parameters())[names[i]]
    o, n, shape = eng._goff[id(p)]
    assert torch.equal(st[i]["exp_avg"].cuda(), eng.adam_m[o:o + n].view(shape)) and float(st[i]["exp_avg_sq"].abs().sum()) > 0
    assert opt["lr_state_dict"]["_schedulers"][0]["last_epoch"] == 8
    sd1 = torch.load(tmp_path / "checkpoint_001.pt", weights_only=True)
    # resume through the reference-style loop (torch AdamW picks the moments up) ...
    model2, losses2, _ = TR.main(TR.get_args(base + ["--num_epochs", "3", "--resume"]))
    assert [l["epoch"] for l in losses2] == [0, 1, 2] and np.isfinite(losses2[2]["train"])
    st2 = torch.load(tmp_path / "optim_002.pt", weights_only=True)["optimizer_state_dict"]["state"]
    assert int(st2[i]["step"]) == 12
    # ... and again fused
    model3, losses3, _ = TR.main(TR.get_args(base + ["--num_epochs", "4", "--resume", "--fused"]))
    assert [l["epoch"] for l in losses3] == [0, 1, 2, 3] and model3._model.engine.step_count == 16
    k = "_features.s1.b1.conv2.conv.weight"
    moved = (model3.state_dict()[k].cpu() - sd1[k].cpu()).abs().max()
    assert 0 < float(moved) < 0.05, moved
    with pytest.raises(NotImplementedError):
        TR.main(TR.get_args(base + ["--num_epochs", "1", "--fused", "-ag", "2"]))
    # no location head, plain backbone, frame-wise classifier: the fused step takes no xy
    nl = [a for a in base if a != "--predict_location"]
    nl[nl.index("rny008_gsm")] = "rny008"
    nl[nl.index("mingru")] = ""
    nl[nl.index(str(tmp_path))] = str(tmp_path / "noloc")
    _, losses4, _ = TR.main(TR.get_args(nl + ["--num_epochs", "2", "--fused"]))
    assert len(losses4) == 2 and np.isfinite(losses4[1]["train"]) and losses4[1]["train"] > 0


def test_train_driver_under_a_launcher_env_uses_rccl_paths(tmp_path, monkeypatch):
    """The torchrun code path on one rank (this box has one GPU): process group over RCCL, weight broadcast, global loss
    denominators, bucketed gradient all-reduce, loss averaging, distributed whole-video evaluation, barriers."""
    import json
    import socket
    from spot_amd import train as TR
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    for k, v in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(port))):
        monkeypatch.setenv(k, v)
    args = TR.get_args(["ds", "synthetic:8", "-m", "rny008_gsm", "-t", "gru", "--clip_len", "8", "--crop_dim", "64", "--batch_size", "2",
                        "--warm_up_epochs", "1", "-lr", "0.002", "--predict_location", "-s", str(tmp_path), "--num_classes", "6",
                        "--criterion", "map", "--start_val_epoch", "0", "--num_epochs", "2"])
    model, losses, best = TR.main(args)
    assert not torch.distributed.is_initialized()
    assert len(losses) == 2 and all(np.isfinite(l["train"]) and 0 <= l["val_mAP"] <= 1 for l in losses) and best in (0, 1)
    assert json.load(open(tmp_path / "loss.json"))[1]["val_mAP"] == losses[1]["val_mAP"]
    assert (tmp_path / "pred-val.1.json").exists() and (tmp_path / "optim_001.pt").exists()


def test_inference_cli_on_predecoded_frames_reads_the_training_drivers_files(tmp_path):
    """python -m spot_amd.infer (the reference's inference_on_mp4.py entry point on decoded frames): architecture from the
    config.json the training driver wrote, weights from its checkpoint, 21 uint8 frames = 3 windows of 8 with a padded
    last one; the events equal run_inference on the same model."""
    import json
    from spot_amd import infer, train as TR
    model, _, _ = TR.main(TR.get_args(["ds", "synthetic:4", "-m", "rny008_gsm", "-t", "deeper_gru", "--clip_len", "8", "--crop_dim", "64",
                                       "--batch_size", "2", "--warm_up_epochs", "1", "-lr", "0.002", "--predict_location", "-s",
                                       str(tmp_path / "run"), "--num_classes", "9", "--num_epochs", "2", "--time_backward", "--criterion", "loss"]))
    frames = torch.randint(0, 256, (21, 3, 64, 64), generator=torch.Generator().manual_seed(3), dtype=torch.uint8)
    np.save(tmp_path / "v.npy", frames.numpy())
    ev = infer.main(["--frames", str(tmp_path / "v.npy"), "--checkpoint_path", str(tmp_path / "run" / "checkpoint_001.pt"),
                     "--save_dir", str(tmp_path / "out"), "--batch_size", "2", "--clip_len", "8", "--no_postprocess"])
    assert json.load(open(tmp_path / "out" / "predictions.json")) == ev
    want, scores = infer.run_inference(model, frames, "block dig net pass receive score serve set spike".split(), 8, batch_size=4,
                                       postprocess=False, video="v.npy")
    assert scores.shape == (21, 10) and [(e["frame"], e["label"]) for e in ev[0]["events"]] == [(e["frame"], e["label"]) for e in want[0]["events"]]
    assert all(abs(a["score"] - b["score"]) < 2e-3 for a, b in zip(ev[0]["events"], want[0]["events"]))


@pytest.mark.parametrize("loc", [False, True])
def test_plain_backbone_with_framewise_fc_head_vs_oracle(loc):
    """-m rny008 -t "" (no temporal shift, per-frame linear classifier: reference train_e2e_spatial.py:394-395), with and
    without the location head: clips of any length up to... any length (nothing in the model depends on clip_len), eval
    predictions and one training step's loss and head gradients against the fp32 oracle."""
    from oracle import stes_oracle as O
    from spot_amd.e2e import E2EModel
    om = O.Impl(7, "rny008", "", clip_len=4, predict_location=loc)
    O.randomize_(om, seed=13)
    m = E2EModel(7, "rny008", "", clip_len=4, modality="rgb", device="cuda", predict_location=loc, verbose=False)
    assert set(m._model.state_dict()) == set(om.state_dict())
    m._model.load_state_dict(om.state_dict(), strict=True)
    g = torch.Generator().manual_seed(14)
    for Tv in (4, 6):
        x = torch.randn(2, Tv, 3, 64, 64, generator=g)
        out, oout = m.predict(x), O.predict(om, x)
        assert len(out) == (3 if loc else 2)
        np.testing.assert_allclose(out[1], oout[1], atol=2e-2)
        if loc:
            np.testing.assert_allclose(out[2], oout[2], atol=2e-2)
    om.train(); m._model.train()
    x = torch.randn(2, 4, 3, 64, 64, generator=g)
    label = torch.randint(0, 7, (2, 4), generator=g); label[0, 0] = 0; label[1, 1] = 3
    od = om(x)
    xy = None
    if loc:
        with torch.no_grad():
            xy = (od["loc_feat"].sigmoid() + torch.where(torch.rand(2, 4, 2, generator=g) < 0.5, -0.2, 0.2)).clamp(0.0, 1.0)
        ol, _, _ = O.stes_loss(od["im_feat"], od["loc_feat"], label.flatten(), xy.reshape(-1, 2), 7)
    else:
        import torch.nn.functional as F
        ol = F.cross_entropy(od["im_feat"].reshape(-1, 7), label.flatten(), weight=torch.tensor([1.0] + [5.0] * 6))
    ol.backward()
    out = m.train_step(x.cuda(), label.cuda(), xy.cuda() if loc else None, lr=0.0, optimizer_step=False).cpu()
    assert abs(float(out[0]) - float(ol.detach())) < 3e-2 * abs(float(ol.detach())), (out, ol)
    for n, p in m._model.named_parameters():
        if n.startswith("_pred_fine"):
            r = rel(p.grad.cpu(), dict(om.named_parameters())[n].grad)
            assert r < 0.05, (n, r)
