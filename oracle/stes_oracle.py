"""CPU oracle for the STES hot path (TEST INFRASTRUCTURE, not product code).

A plain PyTorch fp32 restatement, on the host CPU, of the per-frame feature +
temporal path of the reference (hoangqnguyen/spot):

    RegNet-Y backbone  ->  Gated / plain temporal shift in every bottleneck
    ->  [B, T, F] features  ->  BiGRU (or minGRU) head  +  (x, y) location MLP
    ->  weighted CE + masked L1 loss  ->  AdamW step.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product package (spot_amd/) never does: it runs hand-written
HIP kernels and fails loudly when they are missing.

How the oracle is pinned: tools/gen_golden.py imports the reference's own
model/shift.py, model/modules.py, model/min_gru.py, model/common.py and
util/score.py in the build container, runs them on seeded inputs and commits
the inputs/outputs under tests/golden/.  tests/test_oracle_golden.py checks
every function below against those vectors.  The RegNet-Y arithmetic itself
lives in the third-party package `timm` (requirements.txt:3, `timm>=0.5.4`,
no lock file) which is not installed anywhere here, so the backbone follows
timm's published regnet.py structure and is cross-checked against the
RegNet implementation that ships with `transformers` (weights mapped name to
name); at the timm boundary parity is otherwise UNPINNED.

Reference citations are relative to the reference repository root.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------
# optional bf16 storage emulation.  The HIP path keeps activations in bf16 and
# accumulates in fp32; `set_storage_rounding(True)` makes the oracle round the
# same tensors so that the two can be compared tightly.  Default: pure fp32.
# --------------------------------------------------------------------------
_ROUND = False


def set_storage_rounding(flag: bool) -> None:
    global _ROUND
    _ROUND = bool(flag)


class _RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def q(x: torch.Tensor) -> torch.Tensor:
    """Round to bf16 storage precision when emulation is on (identity otherwise)."""
    return _RoundBf16.apply(x) if _ROUND else x


# --------------------------------------------------------------------------
# Temporal shift: GSM  (reference model/shift.py:83-161)  and TSM (:21-78)
# --------------------------------------------------------------------------
def gsm_fold_dim(channels: int, n_div: int = 4) -> int:
    """reference model/shift.py:150"""
    return math.ceil(channels // n_div / 4) * 4


def gsm_dest_channel(c: int, fold: int) -> int:
    """Where source channel c of the fold slice lands after the within-group
    interleave of reference model/shift.py:124-128 (view(2, fold/4) -> permute)."""
    half, quarter = fold // 2, fold // 4
    g, rem = divmod(c, half)
    i, j = divmod(rem, quarter)
    return g * half + 2 * j + i


def gsm_forward(x, n_segment, conv_w, conv_b, bn_w, bn_b, bn_mean, bn_var,
                training, momentum=0.1, eps=1e-5):
    """Gated shift of an [N, fold, H, W] slice (N = clips * n_segment).

    reference model/shift.py:103-130.  Float op order kept: y = g*x; r = x - y;
    out = shift(y) + r.  Group 0 takes y from frame t+1, group 1 from t-1, zero
    at the clip edge.  Returns [N, fold, H, W].
    """
    n, fold, h, w = x.shape
    b = n // n_segment
    half = fold // 2
    x5 = x.reshape(b, n_segment, fold, h, w).permute(0, 2, 1, 3, 4).contiguous()  # [B,fold,T,H,W]
    xb = F.batch_norm(x5, bn_mean, bn_var, bn_w, bn_b, training, momentum, eps)
    gate = torch.tanh(F.conv3d(F.relu(xb), conv_w, conv_b, stride=1, padding=1, groups=2))  # [B,2,T,H,W]
    out = torch.empty_like(x5)
    dest = [gsm_dest_channel(c, fold) for c in range(fold)]
    for g in range(2):
        xs = x5[:, g * half:(g + 1) * half]
        yg = gate[:, g:g + 1] * xs
        rg = xs - yg
        sh = torch.zeros_like(yg)
        if g == 0:
            sh[:, :, :-1] = yg[:, :, 1:]
        else:
            sh[:, :, 1:] = yg[:, :, :-1]
        og = sh + rg
        out[:, dest[g * half:(g + 1) * half]] = og
    return out.permute(0, 2, 1, 3, 4).reshape(n, fold, h, w)


class GSM(nn.Module):
    """Parameter container with the reference's names (conv3D, bn); model/shift.py:84-95."""

    def __init__(self, f_plane, num_segments):
        super().__init__()
        self.conv3D = nn.Conv3d(f_plane, 2, (3, 3, 3), stride=1, padding=(1, 1, 1), groups=2)
        nn.init.constant_(self.conv3D.weight, 0)
        nn.init.constant_(self.conv3D.bias, 0)
        self.bn = nn.BatchNorm3d(f_plane)
        self.fPlane = f_plane
        self.num_segments = num_segments

    def forward(self, x):
        bn = self.bn
        if self.training and bn.track_running_stats:
            bn.num_batches_tracked += 1
        return gsm_forward(x, self.num_segments, self.conv3D.weight, self.conv3D.bias,
                           bn.weight, bn.bias, bn.running_mean, bn.running_var,
                           self.training, bn.momentum, bn.eps)


class GatedShift(nn.Module):
    """reference model/shift.py:135-161: fold slice through GSM, rest copied, then the wrapped conv."""

    def __init__(self, net, n_segment, n_div):
        super().__init__()
        channels = net.conv.in_channels if hasattr(net, "conv") else net.in_channels
        self.fold_dim = gsm_fold_dim(channels, n_div)
        self.gsm = GSM(self.fold_dim, n_segment)
        self.net = net
        self.n_segment = n_segment

    def forward(self, x):
        f = self.fold_dim
        y = torch.cat([q(self.gsm(x[:, :f])), x[:, f:]], dim=1)
        return self.net(y)


def tsm_shift(x, n_segment, fold_div=8):
    """Plain temporal shift, out-of-place; reference model/shift.py:37-52.
    First c//fold_div channels take frame t+1, the next c//fold_div take t-1."""
    nt, c, h, w = x.shape
    b = nt // n_segment
    x5 = x.reshape(b, n_segment, c, h, w)
    fold = c // fold_div
    out = torch.zeros_like(x5)
    out[:, :-1, :fold] = x5[:, 1:, :fold]
    out[:, 1:, fold:2 * fold] = x5[:, :-1, fold:2 * fold]
    out[:, :, 2 * fold:] = x5[:, :, 2 * fold:]
    return out.reshape(nt, c, h, w)


class _ReluLateMask(torch.autograd.Function):
    """ReLU whose backward mask can be replaced after the forward.  The reference's in-place shift (InplaceShift,
    model/shift.py:55-66) overwrites the storage of the previous layer's ReLU output through `.data`, and ReLU's backward
    multiplies by (saved output > 0): the mask therefore comes from the SHIFTED values."""

    @staticmethod
    def forward(ctx, y, holder):
        out = y.clamp_min(0)
        holder["mask"] = out > 0
        ctx.holder = holder
        return out

    @staticmethod
    def backward(ctx, g):
        return g * ctx.holder["mask"], None


def relu_late(y):
    """relu(y) carrying the handle a following TemporalShift uses to swap the backward mask."""
    holder = {}
    out = q(_ReluLateMask.apply(y, holder))
    out._late_mask = holder
    return out


class TemporalShift(nn.Module):
    """reference model/shift.py:21-35 with inplace=True semantics.  The reference shifts the caller's storage
    (InplaceShift, :55-66) and returns it; autograd sees three consequences, all reproduced here:
      * the wrapped conv reads the shifted tensor and its gradient passes through the adjoint shift (:68-78);
      * the block's residual shortcut still holds the ORIGINAL tensor object: it reads the shifted values, but its
        gradient reaches the previous layer without the adjoint shift (`last_shortcut`);
      * the previous layer's ReLU backward masks with the shifted values (`_ReluLateMask`)."""

    def __init__(self, net, n_segment, n_div=8):
        super().__init__()
        self.net = net
        self.n_segment = n_segment
        self.fold_div = n_div
        self.last_shortcut = None

    def forward(self, x):
        s = tsm_shift(x, self.n_segment, self.fold_div)
        holder = getattr(x, "_late_mask", None)
        if holder is not None and torch.is_grad_enabled():
            holder["mask"] = s.detach() > 0
        self.last_shortcut = x + (s - x).detach()          # value of s, gradient of x
        return self.net(s)


# --------------------------------------------------------------------------
# RegNet-Y (structure of timm models/regnet.py; see module docstring)
# --------------------------------------------------------------------------
class ConvBnAct(nn.Module):
    def __init__(self, cin, cout, k, stride=1, groups=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.apply_act = act

    def forward(self, x):
        x = q(self.conv(x))
        x = self.bn(x)
        return relu_late(x) if self.apply_act else x


class SEModule(nn.Module):
    def __init__(self, channels, rd_channels):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, rd_channels, 1, bias=True)
        self.fc2 = nn.Conv2d(rd_channels, channels, 1, bias=True)

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        s = F.relu(self.fc1(s))
        return x * torch.sigmoid(self.fc2(s))


class Bottleneck(nn.Module):
    def __init__(self, cin, cout, stride, group_size=16, se_ratio=0.25):
        super().__init__()
        groups = cout // group_size
        self.conv1 = ConvBnAct(cin, cout, 1)
        self.conv2 = ConvBnAct(cout, cout, 3, stride=stride, groups=groups)
        self.se = SEModule(cout, int(round(cin * se_ratio)))
        self.conv3 = ConvBnAct(cout, cout, 1, act=False)
        if cin != cout or stride != 1:
            self.downsample = ConvBnAct(cin, cout, 1, stride=stride, act=False)
        else:
            self.downsample = nn.Identity()

    def forward(self, x):
        shortcut = x
        y = self.conv1(x)
        if isinstance(self.conv1, TemporalShift):
            shortcut = self.conv1.last_shortcut         # in-place shift aliasing
        y = self.conv2(y)
        y = self.se(y)
        y = self.conv3(y)
        y = y + self.downsample(shortcut)
        return relu_late(y)


class _Head(nn.Module):
    def __init__(self, feat, num_classes=1000):
        super().__init__()
        self.fc = nn.Linear(feat, num_classes)

    def forward(self, x):
        return self.fc(x.mean((2, 3)))


REGNET_CFGS = {
    # name: (stem width, stage widths, stage depths, group size)
    "regnety_008": (32, (64, 128, 320, 768), (1, 3, 8, 2), 16),
    "regnety_002": (32, (24, 56, 152, 368), (1, 1, 4, 7), 8),
}


class RegNetY(nn.Module):
    def __init__(self, name="regnety_008", in_chans=3):
        super().__init__()
        stem_w, widths, depths, gs = REGNET_CFGS[name]
        self.stem = ConvBnAct(in_chans, stem_w, 3, stride=2)
        cin = stem_w
        for si, (w, d) in enumerate(zip(widths, depths)):
            blocks = OrderedDict()
            for bi in range(d):
                blocks[f"b{bi + 1}"] = Bottleneck(cin, w, 2 if bi == 0 else 1, gs)
                cin = w
            setattr(self, f"s{si + 1}", nn.Sequential(blocks))
        self.final_conv = nn.Identity()
        self.head = _Head(cin)
        self.num_features = cin

    def forward(self, x):
        x = q(x)
        x = q(self.stem(x))
        for s in (self.s1, self.s2, self.s3, self.s4):
            x = s(x)
        return self.head(x)


def make_temporal_shift(net: RegNetY, clip_len: int, is_gsm: bool) -> None:
    """RegNet branch of reference model/shift.py:164-170,191-205."""
    for stage in (net.s1, net.s2, net.s3, net.s4):
        for blk in stage.children():
            if is_gsm:
                blk.conv1 = GatedShift(blk.conv1, n_segment=clip_len, n_div=4)
            else:
                blk.conv1 = TemporalShift(blk.conv1, n_segment=clip_len, n_div=8)


# --------------------------------------------------------------------------
# Heads
# --------------------------------------------------------------------------
def gru_direction(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """One direction of one GRU layer, h0 = 0; x [B,T,I] -> [B,T,H].
    Gate order r, z, n (torch.nn.GRU, used at reference model/modules.py:24-30)."""
    bsz, t_len, _ = x.shape
    hid = w_hh.shape[1]
    gi_all = q(x @ w_ih.t() + b_ih)
    h = x.new_zeros(bsz, hid)
    outs = [None] * t_len
    order = range(t_len - 1, -1, -1) if reverse else range(t_len)
    for t in order:
        gi = gi_all[:, t]
        gh = h @ w_hh.t() + b_hh
        i_r, i_z, i_n = gi.chunk(3, 1)
        h_r, h_z, h_n = gh.chunk(3, 1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (1 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, 1)


class FCPrediction(nn.Module):
    """reference model/modules.py:7-17"""

    def __init__(self, feat_dim, num_classes):
        super().__init__()
        self._fc_out = nn.Linear(feat_dim, num_classes)

    def forward(self, x):
        b, t, _ = x.shape
        return self._fc_out(x.reshape(b * t, -1)).view(b, t, -1)


class ChannelAttention(nn.Module):
    """reference model/modules.py:86-105: per clip, mean over the T frames -> Linear (no bias) -> ReLU -> Linear (no bias)
    -> sigmoid -> every frame's features scaled by that per-channel gate."""

    def __init__(self, feat_dim, reduction=8):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(feat_dim, feat_dim // reduction, bias=False), nn.ReLU(),
                                nn.Linear(feat_dim // reduction, feat_dim, bias=False), nn.Sigmoid())

    def forward(self, x):
        y = self.fc[1](q(self.fc[0](q(x.mean(1)))))
        return q(x * torch.sigmoid(self.fc[2](q(y))).unsqueeze(1))


class GRUPrediction(nn.Module):
    """reference model/modules.py:20-36.  Parameters live in a torch.nn.GRU so the
    state-dict keys match; the arithmetic is the explicit recurrence above."""

    def __init__(self, feat_dim, num_classes, hidden_dim, num_layers=1):
        super().__init__()
        self._gru = nn.GRU(feat_dim, hidden_dim, num_layers=num_layers, batch_first=True, bidirectional=True)
        self._fc_out = FCPrediction(2 * hidden_dim, num_classes)
        self._dropout = nn.Dropout()
        self.num_layers = num_layers
        self.dropout_mask = None      # optional injected keep-mask (already scaled), for parity tests

    def forward(self, x):
        y = x
        for l in range(self.num_layers):
            outs = []
            for sfx, rev in (("", False), ("_reverse", True)):
                p = lambda n: getattr(self._gru, f"{n}_l{l}{sfx}")
                outs.append(gru_direction(y, p("weight_ih"), p("weight_hh"), p("bias_ih"), p("bias_hh"), rev))
            y = q(torch.cat(outs, 2))
        if self.dropout_mask is not None and self.training:
            y = y * self.dropout_mask
        else:
            y = self._dropout(y)
        return self._fc_out(y)


class MLP(nn.Module):
    """reference model/common.py:78-92"""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = q(F.relu(layer(x))) if i < self.num_layers - 1 else q(layer(x))
        return x


class MinGRU(nn.Module):
    """reference model/min_gru.py:69-92"""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.hidden_size = hidden_size
        self.linear_z = nn.Linear(input_size, hidden_size)
        self.linear_h_tilde = nn.Linear(input_size, hidden_size)

    def forward(self, x, h0):
        z = torch.sigmoid(q(self.linear_z(x)))
        ht = q(self.linear_h_tilde(x))
        h = h0
        outs = []
        for t in range(x.shape[1]):
            h = (1 - z[:, t]) * h + z[:, t] * ht[:, t]
            outs.append(h)
        return q(torch.stack(outs, 1))


class MinRNNPredictor(nn.Module):
    """reference model/min_gru.py:4-66 (mingru branch; MinLSTM is never constructed,
    train_e2e_spatial.py:389)"""

    def __init__(self, input_size, hidden_size, output_size, n_layers=1, rnn_type="minGRU", batch_first=True):
        super().__init__()
        if rnn_type.lower() != "mingru":
            raise ValueError("rnn_type must be 'minGRU'")
        self.rnn_layers = nn.ModuleList()
        d = input_size
        for _ in range(n_layers):
            self.rnn_layers.append(MinGRU(d, hidden_size))
            d = hidden_size
        self.fc = nn.Linear(hidden_size, output_size)
        self.h0 = nn.ParameterList(
            [nn.Parameter(torch.zeros(1, hidden_size), requires_grad=False) for _ in range(n_layers)])

    def forward(self, x):
        out = x
        for i, layer in enumerate(self.rnn_layers):
            out = layer(out, self.h0[i].expand(x.shape[0], -1))
        return self.fc(out)


# --------------------------------------------------------------------------
# Model graph (reference train_e2e_spatial.py:232-455), rny00x branch only
# --------------------------------------------------------------------------
MAX_GRU_HIDDEN_DIM = 768


class Impl(nn.Module):
    def __init__(self, num_classes, feature_arch, temporal_arch, clip_len, modality="rgb",
                 predict_location=False, pred_loc_arch="mlp", time_backward=False, use_channel_attention=False):
        super().__init__()
        self.time_backward = time_backward
        self.feature_arch = feature_arch
        self._use_channel_attention = use_channel_attention
        base = feature_arch.rsplit("_", 1)[0] if "_" in feature_arch else feature_arch
        if base not in ("rny002", "rny008"):
            raise NotImplementedError(feature_arch)
        in_ch = {"flow": 2, "bw": 1, "rgb": 3}[modality]
        features = RegNetY({"rny002": "regnety_002", "rny008": "regnety_008"}[base], in_chans=in_ch)
        feat_dim = features.head.fc.in_features
        features.head.fc = nn.Identity()
        self._require_clip_len = -1
        if feature_arch.endswith("_tsm"):
            make_temporal_shift(features, clip_len, is_gsm=False)
            self._require_clip_len = clip_len
        elif feature_arch.endswith("_gsm"):
            make_temporal_shift(features, clip_len, is_gsm=True)
            self._require_clip_len = clip_len
        self._features = features
        self._feat_dim = feat_dim
        # reference train_e2e_spatial.py:348-354
        self._channel_attention = ChannelAttention(feat_dim, reduction=8) if use_channel_attention else nn.Identity()
        hidden = min(feat_dim, MAX_GRU_HIDDEN_DIM)
        if "gru" in temporal_arch:
            if temporal_arch in ("gru", "deeper_gru"):
                self._pred_fine = GRUPrediction(feat_dim, num_classes, hidden,
                                                num_layers=3 if temporal_arch[0] == "d" else 1)
            elif temporal_arch == "mingru":
                self._pred_fine = MinRNNPredictor(feat_dim, hidden, num_classes, n_layers=3, rnn_type="mingru")
            else:
                raise NotImplementedError(temporal_arch)
        elif temporal_arch == "":
            self._pred_fine = FCPrediction(feat_dim, num_classes)
        else:
            raise NotImplementedError(temporal_arch)
        self._predict_location = predict_location
        if predict_location:
            if pred_loc_arch != "mlp":
                raise NotImplementedError(pred_loc_arch)
            self._pred_loc = nn.Sequential(MLP(hidden, hidden * 3, hidden, 2), nn.Linear(hidden, 2))

    def forward(self, x):
        b, true_t, c, h, w = x.shape
        if self.time_backward:
            x = x.flip(1)
        t = true_t
        if self._require_clip_len > 0:
            assert true_t <= self._require_clip_len, "Expected {}, got {}".format(self._require_clip_len, true_t)
            if true_t < self._require_clip_len:
                pad = x.new_zeros(b, self._require_clip_len - true_t, c, h, w)
                x = torch.cat([x, pad], 1)
                t = self._require_clip_len
        im = q(self._features(x.reshape(-1, c, h, w))).reshape(b, t, self._feat_dim)
        if self._use_channel_attention:            # over the padded clip, before the padding is removed (:436-442)
            im = self._channel_attention(im)
        if t != true_t:
            im = im[:, :true_t]
        loc = None
        if self._predict_location:
            loc = self._pred_loc(im)
            if self.time_backward:
                loc = loc.flip(1)
        ev = self._pred_fine(im)
        return {"im_feat": ev.flip(1) if self.time_backward else ev, "loc_feat": loc, "cnn_feat": im}


def randomize_(model: nn.Module, seed: int = 0, gsm_gain: float = 1.0) -> None:
    """Seeded synthetic weights (no pretrained weights exist offline).  GSM conv3D
    is drawn N(0, gsm_gain / sqrt(fan_in)) because its zero init (model/shift.py:89-90)
    makes the gate trivially zero; the fan-in scaling keeps the gate pre-activation O(1)
    at every fold width (a fixed std saturates tanh in the wide blocks and makes the
    network chaotic: fp32 and bf16 runs of the SAME code then differ by > 5 %).  BN affine
    and running stats are perturbed so that every term of the normalisation is exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "conv3D.weight" in name:
                p.copy_(torch.randn(p.shape, generator=g) * gsm_gain / math.sqrt(p[0].numel()))
            elif "conv3D.bias" in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            elif name.endswith("conv3.bn.weight"):
                # timm zero-initialises the last BatchNorm of every residual branch; a small non-zero gain keeps the
                # branch live while the random network stays well conditioned (gain 1 makes every block amplify a
                # perturbation ~1.3x in training mode: fp32 and bf16 runs of the same code then differ by > 20 %)
                p.copy_(0.25 * (1.0 + 0.2 * torch.randn(p.shape, generator=g)))
            elif name.endswith("bn.weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith("bn.bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * math.sqrt(2.0 / fan_in) * (0.5 if "_gru" in name else 1.0))
            elif "h0" in name:
                continue
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
        for name, b in model.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))


# --------------------------------------------------------------------------
# Loss + step  (reference train_e2e_spatial.py:522-593, model/common.py:61-75)
# --------------------------------------------------------------------------
def stes_loss(pred, loc, label, target_xy, num_classes, fg_weight=5.0):
    """Weighted-mean CE over frames + masked L1 on sigmoid(loc).
    pred [B,T,C], loc [B,T,2] or None, label [B*T] int64, target_xy [B*T,2]."""
    weight = torch.tensor([1.0] + [float(fg_weight)] * (num_classes - 1)) if fg_weight != 1 else None
    loss_cls = F.cross_entropy(pred.reshape(-1, num_classes), label, weight=weight)
    loss_loc = pred.new_zeros(())
    if loc is not None:
        mask = (label != 0).float().reshape(-1)
        l1 = (loc.reshape(-1, 2).sigmoid() - target_xy).abs().sum(-1)
        loss_loc = (l1 * mask).sum() / (mask.sum() + 1e-6)
    return loss_cls + loss_loc, loss_cls, loss_loc


def make_adamw(model: nn.Module, lr: float):
    """reference model/common.py:32-42: one group per tensor, `_pred_loc` at lr/5."""
    groups = [{"params": p, "lr": lr / 5 if "_pred_loc" in n else lr} for n, p in model.named_parameters()]
    return torch.optim.AdamW(groups)


def train_step(model: Impl, opt, frame, label, xy, num_classes, fg_weight=5.0):
    model.train()
    out = model(frame)
    loss, lc, ll = stes_loss(out["im_feat"], out["loc_feat"], label.flatten(), xy.reshape(-1, 2), num_classes, fg_weight)
    loss.backward()
    opt.step()
    opt.zero_grad()
    return float(loss.detach()), float(lc.detach()), float(ll.detach())


def predict(model: Impl, seq):
    """reference train_e2e_spatial.py:627-653 (fp32, eval)."""
    if seq.dim() == 4:
        seq = seq.unsqueeze(0)
    model.eval()
    with torch.no_grad():
        d = model(seq)
        scores = torch.softmax(d["im_feat"], 2).numpy()
        cls = torch.argmax(d["im_feat"], 2).numpy()
        if model._predict_location:
            return cls, scores, d["loc_feat"].reshape(seq.shape[0], -1, 2).sigmoid().numpy()
        return cls, scores


# --------------------------------------------------------------------------
# Sliding-window post-processing (reference inference_on_mp4.py:204-270,
# util/score.py:246-300)
# --------------------------------------------------------------------------
def accumulate_windows(n_frames, num_classes, windows):
    """windows: iterable of (start, scores [T,C], loc [T,2])."""
    scores = np.zeros((n_frames, num_classes), np.float32)
    support = np.zeros(n_frames, np.int32)
    locs = np.zeros((n_frames, 2), np.float32)
    for start, s, l in windows:
        end = min(start + s.shape[0], n_frames)
        scores[start:end] += s[:end - start]
        support[start:end] += 1
        locs[start:end] = l[:end - start]
    assert support.min() > 0
    scores /= support[:, None]
    return scores, locs


def events_from_scores(scores, locs, classes):
    pred = np.argmax(scores, axis=1)
    ev = []
    for i in range(pred.shape[0]):
        if pred[i] != 0:
            ev.append({"label": classes[pred[i] - 1], "frame": i,
                       "score": scores[i, pred[i]].item(), "xy": locs[i].tolist()})
    return ev


def filter_events_by_score(events, fg_threshold):
    return [e for e in events if e["score"] >= fg_threshold]


def nms_events(events, tol_t):
    """Per-label greedy grouping of reference util/score.py:268-300."""
    by_label = OrderedDict()
    for e in events:
        by_label.setdefault(e["label"], []).append(e)
    kept = []
    for evs in by_label.values():
        evs = sorted(evs, key=lambda e: e["frame"])
        i = 0
        while i < len(evs):
            cur = evs[i]
            j = i + 1
            while j < len(evs) and evs[j]["frame"] - cur["frame"] <= tol_t:
                if evs[j]["score"] > cur["score"]:
                    cur = evs[j]
                j += 1
            kept.append(cur)
            i = j
    return kept
