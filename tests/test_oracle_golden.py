"""The CPU oracle against vectors produced by the reference's own modules
(tools/gen_golden.py).  fp32: GSM/TSM shift paths bit-exact; heads <= 1e-5 rel."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import stes_oracle as O

T = torch.from_numpy


@pytest.mark.parametrize("fold", [8, 16, 32, 80, 192])
@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("gate", ["rand", "zero"])
def test_gsm_matches_reference(golden, fold, mode, gate):
    z = golden(f"gsm_f{fold}_{mode}_{gate}")
    m = O.GSM(fold, int(z["T"]))
    with torch.no_grad():
        m.conv3D.weight.copy_(T(z["conv_w"])); m.conv3D.bias.copy_(T(z["conv_b"]))
        m.bn.weight.copy_(T(z["bn_w"])); m.bn.bias.copy_(T(z["bn_b"]))
        m.bn.running_mean.copy_(T(z["rm0"])); m.bn.running_var.copy_(T(z["rv0"]))
    m.train(mode == "train")
    x = T(z["x"]).clone().requires_grad_(True)
    y = m(x)
    assert np.array_equal(y.detach().numpy(), z["y"])          # bit-exact forward
    y.backward(T(z["gy"]))
    np.testing.assert_allclose(x.grad.numpy(), z["gx"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m.conv3D.weight.grad.numpy(), z["g_conv_w"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.conv3D.bias.grad.numpy(), z["g_conv_b"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.bn.weight.grad.numpy(), z["g_bn_w"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.bn.running_mean.numpy(), z["rm1"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(m.bn.running_var.numpy(), z["rv1"], rtol=1e-6, atol=1e-7)


def test_gsm_zero_gate_is_channel_permutation(golden):
    # reference init (conv3D = 0): identity + within-group interleave, e.g. fold 8 -> [0,2,1,3,4,6,5,7]
    assert [O.gsm_dest_channel(c, 8) for c in range(8)] == [0, 2, 1, 3, 4, 6, 5, 7]
    z = golden("gsm_f16_eval_zero")
    dest = [O.gsm_dest_channel(c, 16) for c in range(16)]
    assert np.array_equal(z["y"][:, dest], z["x"])


def test_gsm_fold_dims():
    # block-input channels of regnety_008 -> fold (reference model/shift.py:150)
    assert [O.gsm_fold_dim(c) for c in (32, 64, 128, 320, 768)] == [8, 16, 32, 80, 192]


def test_gated_shift(golden):
    z = golden("gated_shift")
    conv = nn.Conv2d(32, 16, 1, bias=False)
    gs = O.GatedShift(conv, 4, 4)
    assert gs.fold_dim == int(z["fold"])
    with torch.no_grad():
        conv.weight.copy_(T(z["conv_w"])); gs.gsm.conv3D.weight.copy_(T(z["gsm_w"])); gs.gsm.conv3D.bias.copy_(T(z["gsm_b"]))
    gs.eval()
    np.testing.assert_allclose(gs(T(z["x"])).detach().numpy(), z["y"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("c", [32, 64, 320])
def test_tsm(golden, c):
    z = golden(f"tsm_c{c}")
    x = T(z["x"]).clone().requires_grad_(True)
    y = O.tsm_shift(x, int(z["T"]), 8)
    assert np.array_equal(y.detach().numpy(), z["y"])
    assert np.array_equal(z["y"], z["y_inplace"]) and bool(z["aliased"])
    y.backward(T(z["gy"]))
    assert np.array_equal(x.grad.numpy(), z["gx"])


def _load_params(mod, z, prefix="p."):
    sd = {k[len(prefix):]: T(z[k]) for k in z.files if k.startswith(prefix)}
    mod.load_state_dict(sd, strict=True)


@pytest.mark.parametrize("L", [1, 3])
def test_gru_head(golden, L):
    z = golden(f"gru_L{L}")
    m = O.GRUPrediction(48, 7, 48, num_layers=L)
    _load_params(m, z)
    m.eval()
    x = T(z["x"]).clone().requires_grad_(True)
    y = m(x)
    np.testing.assert_allclose(y.detach().numpy(), z["y"], rtol=1e-5, atol=1e-6)
    y.backward(T(z["gy"]))
    np.testing.assert_allclose(x.grad.numpy(), z["gx"], rtol=1e-4, atol=1e-6)
    for n, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), z["g." + n], rtol=1e-4, atol=1e-5)


def test_loc_mlp(golden):
    z = golden("loc_mlp")
    m = nn.Sequential(O.MLP(32, 96, 32, 2), nn.Linear(32, 2))
    _load_params(m, z)
    x = T(z["x"]).clone().requires_grad_(True)
    y = m(x)
    np.testing.assert_allclose(y.detach().numpy(), z["y"], rtol=1e-5, atol=1e-6)
    y.backward(T(z["gy"]))
    np.testing.assert_allclose(x.grad.numpy(), z["gx"], rtol=1e-4, atol=1e-6)


def test_mingru(golden):
    z = golden("mingru")
    m = O.MinRNNPredictor(24, 40, 7, n_layers=3, rnn_type="mingru")
    _load_params(m, z)
    x = T(z["x"]).clone().requires_grad_(True)
    y = m(x)
    np.testing.assert_allclose(y.detach().numpy(), z["y"], rtol=1e-5, atol=1e-6)
    y.backward(T(z["gy"]))
    np.testing.assert_allclose(x.grad.numpy(), z["gx"], rtol=1e-4, atol=1e-6)
    for n, p in m.named_parameters():
        if "g." + n in z.files:
            np.testing.assert_allclose(p.grad.numpy(), z["g." + n], rtol=1e-4, atol=1e-5)


def test_channel_attention(golden):
    """The clip-level channel gate (reference model/modules.py:86-105): output and all three gradients."""
    z = golden("channel_attention")
    m = O.ChannelAttention(64, reduction=8)
    with torch.no_grad():
        m.fc[0].weight.copy_(T(z["w1"])); m.fc[2].weight.copy_(T(z["w2"]))
    assert set(m.state_dict()) == {"fc.0.weight", "fc.2.weight"}
    x = T(z["x"]).clone().requires_grad_(True)
    y = m(x)
    np.testing.assert_allclose(y.detach().numpy(), z["y"], rtol=1e-5, atol=1e-6)
    y.backward(T(z["dy"]))
    np.testing.assert_allclose(x.grad.numpy(), z["dx"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(m.fc[0].weight.grad.numpy(), z["dw1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(m.fc[2].weight.grad.numpy(), z["dw2"], rtol=1e-4, atol=1e-6)


def test_step_loss_adamw(golden):
    z = golden("step_heads")

    class Heads(nn.Module):
        def __init__(self):
            super().__init__()
            self._pred_fine = O.GRUPrediction(32, 7, 32, 1)
            self._pred_loc = nn.Sequential(O.MLP(32, 96, 32, 2), nn.Linear(32, 2))

    m = Heads()
    _load_params(m, z, "p0.")
    m.eval()
    opt = O.make_adamw(m, float(z["lr"]))
    x, label, xy = T(z["x"]), T(z["label"]), T(z["xy"])
    for it in range(2):
        loss, lc, ll = O.stes_loss(m._pred_fine(x), m._pred_loc(x), label.flatten(), xy.reshape(-1, 2), 7)
        loss.backward(); opt.step(); opt.zero_grad()
        np.testing.assert_allclose([loss.item(), lc.item(), ll.item()], z["losses"][it], rtol=1e-5)
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(v.numpy(), z["p2." + k], rtol=1e-4, atol=1e-6)


def test_postproc_events_bit_exact(golden):
    z = golden("postproc")
    classes = ["serve", "receive", "set", "attack", "block", "score"]
    ev = O.events_from_scores(z["scores"].copy(), z["locs"], classes)
    f = O.filter_events_by_score(ev, 0.15)
    s = O.nms_events(f, 3)
    lab = {c: i for i, c in enumerate(classes)}
    enc = lambda e: np.array([[x["frame"], lab[x["label"]]] for x in e], np.int64).reshape(-1, 2)
    assert np.array_equal(enc(f), z["filtered"])
    assert np.array_equal(enc(s), z["nms"])
    assert np.array_equal(np.array([e["score"] for e in s]), z["nms_scores"])


@pytest.mark.parametrize("arch,tarch,tag", [("rny008_gsm", "gru", "full_rny008_gsm_gru"),
                                            ("rny008_tsm", "mingru", "full_rny008_tsm_mingru")])
def test_full_model(golden, arch, tarch, tag):
    z = golden(tag)
    m = O.Impl(7, arch, tarch, clip_len=4, predict_location=True)
    O.randomize_(m, seed=3)
    assert abs(sum(p.abs().sum().item() for p in m.parameters()) - float(z["param_abs_sum"])) < 1e-3 * float(z["param_abs_sum"])
    x = T(z["x"])
    m.eval()
    with torch.no_grad():
        d = m(x)
    for k in ("im_feat", "loc_feat", "cnn_feat"):
        np.testing.assert_allclose(d[k].numpy(), z["eval." + k], rtol=1e-4, atol=1e-5)
    m.train()
    if tarch == "gru":
        m._pred_fine._dropout.p = 0.0
    d = m(x)
    for k in ("im_feat", "loc_feat", "cnn_feat"):
        np.testing.assert_allclose(d[k].detach().numpy(), z["train." + k], rtol=1e-4, atol=1e-5)
    loss, lc, ll = O.stes_loss(d["im_feat"], d["loc_feat"], T(z["label"]).flatten(), T(z["xy"]).reshape(-1, 2), 7)
    np.testing.assert_allclose([lc.item(), ll.item()], z["loss"], rtol=1e-5)
    loss.backward()
    norms = dict(zip(z["grad_names"].tolist(), z["grad_norms"].tolist()))
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        assert abs(p.grad.double().norm().item() - norms[n]) <= 2e-3 * norms[n] + 1e-7, n
        if "g." + n in z.files:
            if tarch == "gru":
                np.testing.assert_allclose(p.grad.numpy(), z["g." + n], rtol=2e-3, atol=1e-5 * (1 + norms[n]))
            else:
                # the in-place shift variant was generated through strided views: its fp32 pre-activations differ from this
                # run's in the last bits, which flips the odd ReLU at zero (one element changes a channel's gradient by
                # percents); the tensors agree to a few 1e-3 in relative L2, the reference's autograd quirks would show as 1e-1
                g = z["g." + n]
                assert np.linalg.norm(p.grad.numpy() - g) <= 5e-3 * np.linalg.norm(g), n
    for n, b in m.named_buffers():
        if "buf." + n in z.files:
            np.testing.assert_allclose(b.numpy(), z["buf." + n], rtol=1e-5, atol=1e-6)
