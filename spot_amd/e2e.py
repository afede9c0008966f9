"""E2EModel: the reference's model class (train_e2e_spatial.py:230-653) on the HIP engine.

Same constructor arguments, attributes (`device`, `_model`, `_num_classes`), methods (`epoch`, `predict`,
`get_optimizer`, `state_dict`, `load`) and state-dict keys.  `Impl.forward` returns the reference's dict
(`im_feat` logits, `loc_feat` pre-sigmoid, `cnn_feat`).  Under autograd the whole network is ONE
autograd node whose backward runs the engine's hand-written backward pass and leaves the parameter
gradients in `.grad` (views of the engine's flat gradient buffer), so the reference's training loop
(`loss.backward(); optimizer.step()` in model/common.py:61-75) works unchanged on top.
`E2EModel.train_step` is the fused path: loss, backward and AdamW all in HIP, no host synchronisation.
"""
from __future__ import annotations

from contextlib import nullcontext

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib
from .engine import BF, Engine
from .model.common import MLP, BaseRGBModel, step
from .model.min_gru import MinRNNPredictor
from .model.modules import ChannelAttention, FCPrediction, GRUPrediction
from .model.shift import make_temporal_shift
from .regnet import create_model

MAX_GRU_HIDDEN_DIM = 768          # reference train_e2e_spatial.py:42


class _EngineFn(torch.autograd.Function):
    """The network as one autograd node.  `hook` is a dummy leaf that makes the outputs require grad; parameter
    gradients are written straight into the flat buffer behind every `param.grad`."""

    @staticmethod
    def forward(ctx, x, hook, impl):
        eng = impl.engine
        logits, loc, feat = eng.forward(x, train=impl.training)   # x: fp32 (normalised) or uint8 (decoded) frames
        ctx.impl = impl
        ctx.has_loc = loc is not None
        outs = (logits.clone(), loc.clone() if loc is not None else logits.new_zeros(()), feat.float())
        return outs

    @staticmethod
    def backward(ctx, dlogits, dloc, dfeat):
        impl = ctx.impl
        eng = impl.engine
        N = eng.N
        fresh = all(p.grad is None for _, p in eng._params)
        eng.attach_grads(zero=fresh)
        nc = dlogits.shape[-1]
        ncp = (nc + 7) // 8 * 8
        dl = torch.zeros(N, ncp, dtype=BF, device=dlogits.device)
        dl[:, :nc] = dlogits.reshape(N, nc)
        dlc = None
        if ctx.has_loc and dloc is not None:
            dlc = torch.zeros(N, 8, dtype=BF, device=dlogits.device)
            dlc[:, :2] = dloc.reshape(N, 2)
        extra = dfeat            # the gradient w.r.t. the returned features (zeros when they are unused) is added on the device
        eng.backward(dl, dlc, extra)
        return None, None, None


class E2EModel(BaseRGBModel):

    class Impl(nn.Module):
        """reference train_e2e_spatial.py:232-468, rny008 branch (the only backbone of the hot path)."""

        def __init__(self, num_classes, feature_arch, temporal_arch, clip_len, modality, predict_location=False,
                     pred_loc_arch="mlp", time_backward=False, use_channel_attention=False):
            super().__init__()
            self._pred_loc_arch = pred_loc_arch
            self.time_backward = time_backward
            is_rgb = modality == "rgb"
            in_channels = {"flow": 2, "bw": 1, "rgb": 3}[modality]
            self.feature_arch = feature_arch
            self._use_channel_attention = use_channel_attention
            if feature_arch.startswith("rny008"):
                features = create_model("regnety_008", pretrained=is_rgb, in_chans=in_channels)
                feat_dim = features.head.fc.in_features
                features.head.fc = nn.Identity()
            else:
                raise NotImplementedError(feature_arch)
            self._channel_attention = ChannelAttention(feat_dim, reduction=8) if use_channel_attention else nn.Identity()
            self._require_clip_len = -1
            if feature_arch.endswith("_tsm"):
                make_temporal_shift(features, clip_len, is_gsm=False)
                self._require_clip_len = clip_len
            elif feature_arch.endswith("_gsm"):
                make_temporal_shift(features, clip_len, is_gsm=True)
                self._require_clip_len = clip_len
            self._features = features
            self._feat_dim = feat_dim
            self._clip_len = clip_len
            if "gru" in temporal_arch:
                hidden_dim = feat_dim
                if hidden_dim > MAX_GRU_HIDDEN_DIM:
                    hidden_dim = MAX_GRU_HIDDEN_DIM
                    print("Clamped GRU hidden dim: {} -> {}".format(feat_dim, hidden_dim))
                if temporal_arch in ("gru", "deeper_gru"):
                    self._pred_fine = GRUPrediction(feat_dim, num_classes, hidden_dim,
                                                    num_layers=3 if temporal_arch[0] == "d" else 1)
                elif temporal_arch == "mingru":
                    self._pred_fine = MinRNNPredictor(input_size=feat_dim, hidden_size=hidden_dim, output_size=num_classes,
                                                      n_layers=3, rnn_type="mingru", batch_first=True)
                else:
                    raise NotImplementedError(temporal_arch)
            elif temporal_arch == "":
                hidden_dim = feat_dim
                self._pred_fine = FCPrediction(feat_dim, num_classes)
            else:
                raise NotImplementedError(temporal_arch)
            self._predict_location = predict_location
            if self._predict_location:
                if pred_loc_arch == "mlp":
                    self._pred_loc = nn.Sequential(MLP(hidden_dim, hidden_dim * 3, output_dim=hidden_dim, num_layers=2),
                                                   nn.Linear(hidden_dim, 2))
                else:
                    raise NotImplementedError(f"Unimplemented location predictor: {pred_loc_arch}")
            self._engine = None
            self._hook = None

        @property
        def engine(self) -> Engine:
            if self._engine is None:
                self._engine = Engine(self, self._clip_len)
                self._hook = torch.zeros(1, device=self._engine.dev, requires_grad=True)
            return self._engine

        def load_state_dict(self, state_dict, strict=True, **kw):
            r = super().load_state_dict(state_dict, strict=strict, **kw)
            if self._engine is not None:
                self._engine.weights_dirty = True
            return r

        def forward(self, x):
            batch_size, true_clip_len, channels, height, width = x.shape
            if self.time_backward:
                x = x.flip(1)
            clip_len = true_clip_len
            pad_T = None
            if self._require_clip_len > 0:
                assert true_clip_len <= self._require_clip_len, "Expected {}, got {}".format(self._require_clip_len, true_clip_len)
                if true_clip_len < self._require_clip_len:
                    if x.dtype == torch.uint8:
                        pad_T = self._require_clip_len       # decoded frames: the stem kernel writes the zero frames
                    else:
                        x = F.pad(x, (0,) * 7 + (self._require_clip_len - true_clip_len,))
                    clip_len = self._require_clip_len
            eng = self.engine
            if self.training and torch.is_grad_enabled():
                if true_clip_len != clip_len:
                    raise NotImplementedError("training needs clips of exactly clip_len frames")
                logits, loc, feat = _EngineFn.apply(x, self._hook, self)
                if not self._predict_location:
                    loc = None
            else:
                # a padded clip is un-padded before the heads, as the reference does (train_e2e_spatial.py:440-442)
                logits, loc, feat = eng.forward(x, train=self.training, head_T=true_clip_len, pad_T=pad_T)
                logits, feat = logits.clone(), feat.float()
                loc = loc.clone() if loc is not None else None
            if self.time_backward:
                logits = logits.flip(1)
                loc = loc.flip(1) if loc is not None else None
            return {"im_feat": logits, "loc_feat": loc, "cnn_feat": feat}

        def print_stats(self):
            print(f"Model params:{sum(p.numel() for p in self.parameters()):,}")
            print(f"CNN features:{sum(p.numel() for p in self._features.parameters()):,}")
            print(f"Temporal Head:{sum(p.numel() for p in self._pred_fine.parameters()):,}")
            if hasattr(self, "_pred_loc"):
                print(f"Spatial Head:{sum(p.numel() for p in self._pred_loc.parameters()):,}")

    def __init__(self, num_classes, feature_arch, temporal_arch, clip_len, modality, device="cuda", predict_location=False,
                 multi_gpu=False, pred_loc_arch="mlp", time_backward=False, use_channel_attention=False, verbose=True):
        self.device = device
        self._multi_gpu = multi_gpu
        self._model = E2EModel.Impl(num_classes, feature_arch, temporal_arch, clip_len, modality, predict_location,
                                    pred_loc_arch=pred_loc_arch, time_backward=time_backward,
                                    use_channel_attention=use_channel_attention)
        if verbose:
            self._model.print_stats()
        # multi_gpu: the reference wraps nn.DataParallel (train_e2e_spatial.py:500-501); here data parallelism is one
        # process per GPU (spot_amd.dp.DataParallelTrainer over RCCL), so the flag does not change the module.
        self._model.to(device)
        self._num_classes = num_classes
        self._temporal_arch = temporal_arch

    # ------------------------------------------------------------------ reference API
    def epoch(self, loader, optimizer=None, scaler=None, lr_scheduler=None, acc_grad_iter=1, fg_weight=5):
        """reference train_e2e_spatial.py:507-625 (tqdm bar dropped; losses identical)."""
        if optimizer is None:
            self._model.eval()
        else:
            optimizer.zero_grad()
            self._model.train()
        ce_kwargs = {}
        if fg_weight != 1:
            ce_kwargs["weight"] = torch.FloatTensor([1] + [fg_weight] * (self._num_classes - 1)).to(self.device)
        sums = {"sum": 0.0, "cls": 0.0, "loc": 0.0, "contrast": 0.0}
        with torch.no_grad() if optimizer is None else nullcontext():
            for batch_idx, batch in enumerate(loader):
                frame = batch["frame"].to(self.device)
                label = batch["label"].to(self.device)
                if self._model._predict_location:
                    target_xy = batch["xy"].to(self.device).reshape(-1, 2)
                label = label.flatten() if len(label.shape) == 2 else label.view(-1, label.shape[-1])
                preds = self._model(frame)
                pred, loc = preds["im_feat"], preds["loc_feat"]
                loss_cls, loss_loc = 0.0, torch.zeros((), device=pred.device)
                if len(pred.shape) == 3:
                    pred = pred.unsqueeze(0)
                for i in range(pred.shape[0]):
                    loss_cls = loss_cls + F.cross_entropy(pred[i].reshape(-1, self._num_classes), label, **ce_kwargs)
                if self._model._predict_location:
                    event_mask = (label != 0).float().reshape(-1)
                    xy_loss = F.l1_loss(loc.reshape(-1, 2).sigmoid(), target_xy, reduction="none").sum(dim=-1)
                    loss_loc = (xy_loss * event_mask).sum() / (event_mask.sum() + 1e-6)
                loss = loss_cls + loss_loc
                if optimizer is not None:
                    step(optimizer, scaler, loss / acc_grad_iter, lr_scheduler=lr_scheduler,
                         backward_only=(batch_idx + 1) % acc_grad_iter != 0)
                    if (batch_idx + 1) % acc_grad_iter == 0:
                        self._model.engine.weights_dirty = True
                sums["sum"] += loss.detach().item()
                sums["cls"] += loss_cls.detach().item()
                sums["loc"] += loss_loc.detach().item()
        n = max(1, len(loader))
        return {k: v / n for k, v in sums.items()}

    def predict(self, seq, use_amp=False, presence_threshold=0.0):
        """reference train_e2e_spatial.py:627-653.  `use_amp` is accepted and ignored: the path always runs bf16
        storage with fp32 accumulation."""
        if not isinstance(seq, torch.Tensor):
            seq = torch.FloatTensor(np.asarray(seq))
        if len(seq.shape) == 4:
            seq = seq.unsqueeze(0)
        if seq.device != torch.device(self.device) and str(seq.device) != str(self.device):
            seq = seq.to(self.device)
        self._model.eval()
        with torch.no_grad():
            pred_dict = self._model(seq)
            pred_cls_score = torch.softmax(pred_dict["im_feat"], axis=2).cpu().numpy()
            pred_cls = torch.argmax(pred_dict["im_feat"], axis=2).cpu().numpy()
            if self._model._predict_location:
                pred_loc = pred_dict["loc_feat"].reshape(seq.shape[0], -1, 2).sigmoid().cpu().numpy()
                return pred_cls, pred_cls_score, pred_loc
            return pred_cls, pred_cls_score

    def predict_async(self, seq):
        """predict() without the host round trip: (class scores [B,T,classes], locations [B,T,2] | None) as device tensors of
        their own (the next forward does not overwrite them), nothing waited for.  A caller that walks a long video enqueues
        the next batch before it reads the previous one back, so the GPU is never idle while the host issues launches
        (reference inference_on_mp4.py:204-243 reads every batch back before it starts the next)."""
        if seq.device != torch.device(self.device) and str(seq.device) != str(self.device):
            seq = seq.to(self.device, non_blocking=True)
        if self._model.training:             # (eval() walks every module: 0.5 ms a call)
            self._model.eval()
        with torch.no_grad():
            pred_dict = self._model(seq)
            scores = torch.softmax(pred_dict["im_feat"], axis=2)
            loc = pred_dict["loc_feat"].reshape(seq.shape[0], -1, 2).sigmoid() if self._model._predict_location else None
        return scores, loc

    # ------------------------------------------------------------------ fused training step (HIP loss + backward + AdamW)
    def train_step(self, frame, label, xy, lr, fg_weight=5.0, grad_scale=1.0, ext_denoms=None, optimizer_step=True,
                   after_backward=None, grad_mul=1.0, on_ready=None):
        """One optimisation step entirely on the device.  frame [B,T,C,H,W] f32 (normalised) or uint8 (decoded), label [B,T] int64, xy [B,T,2] f32.
        Returns the device tensor {sum, cls, loc, sum of class weights, number of event frames} (no host sync)."""
        m = self._model
        m.train()
        eng = m.engine
        if frame.shape[1] != m._clip_len and m._require_clip_len > 0:
            raise lib.StesError("train_step expects clips of exactly clip_len frames")
        eng.forward(frame, train=True)
        eng.attach_grads(zero=True)
        out = eng.loss_backward(label.reshape(-1), xy.reshape(-1, 2) if xy is not None else None, self._num_classes,
                                fg_weight=fg_weight, grad_scale=grad_scale, ext_denoms=ext_denoms, on_ready=on_ready)
        if after_backward is not None:
            after_backward(eng)
        if optimizer_step:
            eng.adamw_step(lr, grad_mul=grad_mul)
        return out
