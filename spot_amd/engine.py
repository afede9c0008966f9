"""Host-side executor of the STES hot path on one MI355X.

Walks the parameter tree of `E2EModel.Impl` (RegNet-Y + gated/plain temporal shift + GRU/minGRU
head + location MLP; reference train_e2e_spatial.py:232-455) once, lays every learnable tensor
out in ONE flat fp32 buffer (with a twin buffer for gradients, so the data-parallel all-reduce and
the fused AdamW see contiguous memory), keeps bf16 MFMA-ready copies of the weights, and runs the
forward / backward pass as a fixed sequence of libstes_hip kernel launches on the current HIP stream.

Activations are NHWC bf16 ([frames*H*W][C]); accumulation, BatchNorm statistics, the GRU state,
losses and optimizer state are fp32.  Nothing here computes on the host or through torch operators:
torch supplies device memory, streams and (in spot_amd.dp) the RCCL process group.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os

import torch
import torch.nn as nn

from . import lib
from .model.modules import GRUPrediction
from .model.min_gru import MinRNNPredictor
from .model.shift import GatedShift, TemporalShift
from .regnet import ConvBnAct

BF = torch.bfloat16
F32 = torch.float32
# Weight-gradient kernels run on a side stream beside the input-gradient chain.  Off (STES_SERIAL=1, bench.py --serial, and
# during the per-kernel roofline pass) they run in program order on the main stream, which gives the same results.
OVERLAP = os.environ.get("STES_SERIAL", "0") != "1"


def overlap():
    """Side-stream execution is on (never in the library's deterministic mode, which fixes the order of every float
    accumulation and therefore wants all launches on one stream, in program order)."""
    return OVERLAP and lib.deterministic_ticket() is None


def _rup(x, m):
    return (x + m - 1) // m * m


class BN:
    """One BatchNorm site: folded scale/shift for the consumer's prologue, saved batch statistics, the
    backward coefficients, and four accumulators in the per-step zeroed statistics arena."""

    def __init__(self, eng, mod):
        self.mod, self.C = mod, mod.num_features
        Cc = self.C
        self.scale, self.shift, self.mean, self.inv, self.alpha, self.beta, self.delta = eng.f32_block(7, Cc)
        self.sum, self.sumsq, self.s_dy, self.s_dyr = eng.stats_block(4, Cc)
        self.eng = eng
        eng.bns.append(self)

    def finalize(self, count, training):
        if not training and self.eng._eval_bn_live:
            return              # evaluation: scale/shift folded from the running statistics by an earlier forward are current
        m = self.mod
        lib.call("stes_bn_finalize", lib.ptr(self.sum), lib.ptr(self.sumsq), float(count), lib.ptr(m.weight), lib.ptr(m.bias),
                 lib.ptr(m.running_mean), lib.ptr(m.running_var), float(m.momentum), float(m.eps), int(training),
                 lib.ptr(self.scale), lib.ptr(self.shift), lib.ptr(self.mean), lib.ptr(self.inv), self.C)

    def bwd_finalize(self, count, s_dy=None):
        g = self.eng.grad
        lib.call("stes_bn_bwd_finalize", lib.ptr(s_dy if s_dy is not None else self.s_dy), lib.ptr(self.s_dyr), lib.ptr(self.mean),
                 lib.ptr(self.inv), lib.ptr(self.mod.weight), float(count), lib.ptr(self.alpha), lib.ptr(self.beta),
                 lib.ptr(self.delta), lib.ptr(g(self.mod.weight)), lib.ptr(g(self.mod.bias)), self.C)

    @property
    def coef(self):
        return (self.alpha, self.beta, self.delta)


class Linear:
    """nn.Linear / 1x1 conv weight with its bf16 copy [N][K] and transposed copy [K][N8]."""

    def __init__(self, eng, weight, bias=None, n=None, k=None, kp=None, perm_fold=0):
        self.w, self.b = weight, bias
        self.N = n if n is not None else weight.shape[0]
        self.K = k if k is not None else weight[0].numel()
        self.Kp = kp if kp is not None else _rup(self.K, 8)           # padded K (stem: 9*Cin -> 32)
        self.Np = _rup(self.N, 8)
        self.w_bf = eng.bf16(self.N, self.Kp)
        self.wT_bf = eng.bf16(self.Kp, self.Np)
        eng.casts.append((weight, self.w_bf, self.N, self.K, self.Kp, self.N, 0, perm_fold))
        eng.casts.append((weight, self.wT_bf, self.N, self.K, self.Np, self.Kp, 1, perm_fold))


class Block:
    def __init__(self, eng, blk, name):
        self.eng, self.name = eng, name
        c1 = blk.conv1
        self.kind = "gsm" if isinstance(c1, GatedShift) else ("tsm" if isinstance(c1, TemporalShift) else "none")
        self.pre_shifted = False
        conv1 = c1.net if self.kind != "none" else c1
        self.cin, self.cmid, self.stride = conv1.conv.in_channels, conv1.conv.out_channels, blk.conv2.conv.stride[0]
        self.cout = blk.conv3.conv.out_channels
        if blk.conv2.conv.in_channels // blk.conv2.conv.groups != 16 or self.cmid % 64:
            raise NotImplementedError("grouped 3x3 conv kernel is built for group width 16 and channels % 64 == 0 (regnety_008)")
        self.fold = 0
        if self.kind == "gsm":
            self.fold = c1.fold_dim
            if not (self.fold == 8 or self.fold % 16 == 0) or self.fold > 192:
                raise NotImplementedError("gated shift kernels are built for fold 8 or a multiple of 16, up to 192")
        # the gated-shift output keeps the source channel order: conv1's input columns are permuted instead
        self.l1 = Linear(eng, conv1.conv.weight, perm_fold=self.fold)
        self.l3 = Linear(eng, blk.conv3.conv.weight)
        self.bn1, self.bn2, self.bn3 = BN(eng, conv1.bn), BN(eng, blk.conv2.bn), BN(eng, blk.conv3.bn)
        self.pool_fused = True
        self.w2 = blk.conv2.conv.weight
        G = self.cmid // 16
        self.wp2_f, self.wp2_d = eng.bf16(G * 5 * 512), eng.bf16(G * 5 * 512)
        eng.packs.append((self.w2, self.wp2_f, self.wp2_d, G, self.stride))
        self.se = blk.se
        self.rd = blk.se.fc1.out_channels
        if self.rd % 8:
            raise NotImplementedError("squeeze-excite reduction width must be a multiple of 8")
        self.se1, self.se2 = Linear(eng, blk.se.fc1.weight), Linear(eng, blk.se.fc2.weight)
        self.down = isinstance(blk.downsample, ConvBnAct)
        if self.down:
            self.ld = Linear(eng, blk.downsample.conv.weight)
            self.bnd = BN(eng, blk.downsample.bn)
        if self.kind == "gsm":
            self.gsm = c1.gsm
            self.bng = BN(eng, c1.gsm.bn)
            self.wt = torch.zeros(27 * self.fold, dtype=torch.bfloat16, device=eng.dev)
            eng.gsm_preps.append((c1.gsm.conv3D.weight, self.wt, self.fold))
        elif self.kind == "tsm":
            self.tsm_fold = self.cin // c1.fold_div

    # ------------------------------------------------------------------ shapes / buffers
    def plan(self, N, Hi, Wi, T, train):
        e = self.eng
        self.N, self.Hi, self.Wi, self.T = N, Hi, Wi, T
        s = self.stride
        self.Ho, self.Wo = (Hi - 1) // s + 1, (Wi - 1) // s + 1
        self.Mi, self.Mo = N * Hi * Wi, N * self.Ho * self.Wo
        a = e.act
        nm = self.name
        if self.kind == "gsm":
            self.Xc = a(nm + ".Xc", self.Mi * self.fold, BF)
            self.G = a(nm + ".G", self.Mi * self.fold, BF)
            self.gate = a(nm + ".gate", self.Mi * 2, F32)
        elif self.kind == "tsm":
            # the reference shifts the block input in place (model/shift.py:55-66): nobody reads the unshifted tensor afterwards,
            # so only the moving quarter of the channels is touched; other fold widths go through a shifted copy
            self.tsm_inplace = self.tsm_fold % 4 == 0 and self.cin % 4 == 0
            self.Xs = None if self.tsm_inplace else a(nm + ".Xs", self.Mi * self.cin, BF)
        # (X > 0) as one bit per element, written by the kernel that produces X and read by this block's input gradient in
        # place of X; the plain temporal shift masks with the shifted tensor instead (model/shift.py:55-66)
        self.xbits = a(nm + ".xbits", self.Mi * self.cin // 8, torch.uint8) if train and self.kind != "tsm" else None
        self.c1 = a(nm + ".c1", self.Mi * self.cmid, BF)
        self.c2 = a(nm + ".c2", self.Mo * self.cmid, BF)
        self.c3 = a(nm + ".c3", self.Mo * self.cout, BF)
        self.pooled = a(nm + ".pooled", N * self.cmid, BF)
        self.hidden = a(nm + ".hidden", N * self.rd, BF)
        self.segate = a(nm + ".segate", N * self.cmid, F32)
        # Where the last 1x1 conv runs on the tiled GEMM (768 channels), its operand relu(bn2(c2)) * gate is written out once:
        # the product with the operand prologue takes 217 us at 50,176 rows, the pass (40) plus the plain product (112) less,
        # and the weight gradient reads the same tensor instead of transforming its operand again.
        self.a2 = a(nm + ".a2", self.Mo * self.cmid, BF) if not lib.gemm_streams(self.Mo, self.cout, self.cmid) and self.Mo >= 8192 else None
        if self.down:
            self.sc = a(nm + ".sc", self.Mo * self.cout, BF)
        self.out = a(nm + ".out", self.Mo * self.cout, BF)
        return self.Ho, self.Wo

    # ------------------------------------------------------------------ forward
    def forward(self, X, train, nxt):
        """nxt: the next block when it has a gated shift (its fold slice copy and BatchNorm3d sums are produced here)."""
        e = self.eng
        p = lib.ptr
        Mi, Mo, cin, cmid, cout = self.Mi, self.Mo, self.cin, self.cmid, self.cout
        st = (lambda bn: (bn.sum, bn.sumsq)) if train else (lambda bn: None)
        self.X = X
        a_src, a0, k0 = X, None, 0
        if self.kind == "gsm":
            geom = lib.gsm_geom(self.N // self.T, self.T, self.Hi, self.Wi, self.fold)
            self.geom = geom
            self.bng.finalize(Mi, train)
            # gate conv (MFMA) and shift in one kernel: the narrow slices of the large maps (230 us instead of 360 at fold 8, 129
            # instead of 141 at fold 16) and the wide ones of the small maps (measured: faster from fold 80 up, slower at 32)
            if self.fold >= 80 or self.fold <= 16:
                lib.call("stes_gsm_forward", p(self.Xc), p(self.bng.scale), p(self.bng.shift), p(self.wt), p(self.gsm.conv3D.bias),
                         p(self.gate), p(self.G), C.byref(geom))
            else:
                lib.call("stes_gsm_gate", p(self.Xc), p(self.bng.scale), p(self.bng.shift), p(self.wt), p(self.gsm.conv3D.bias), p(self.gate),
                         C.byref(geom))
                lib.call("stes_gsm_shift", p(self.Xc), p(self.gate), p(self.G), C.byref(geom))
            a0, k0 = self.G, self.fold
        elif self.kind == "tsm":
            if self.tsm_inplace:
                if not self.pre_shifted:           # (in training the kernel that wrote X has shifted it on its way out)
                    lib.call("stes_tsm_shift_inplace", p(X), self.tsm_fold, cin, self.N // self.T, self.T, self.Hi, self.Wi, 0)
                self.pre_shifted = False
            else:
                lib.call("stes_tsm_shift", p(X), p(self.Xs), self.tsm_fold, cin, self.N // self.T, self.T, self.Hi, self.Wi, 0)
                a_src = self.Xs
        self.a_src = a_src
        if (not train and e.block_fused and cin == cmid == cout and self.l1.Kp == cin
                and lib.block_eval_takes(cmid, self.Hi, self.Wi, self.rd, k0, self.stride, self.down)):
            # evaluation, stage-3 identity blocks: every BatchNorm is a fixed affine map and one 14x14 image's activations fit
            # a compute unit's LDS, so the three convolutions and the squeeze-excite run per image in one kernel and only the
            # block input and output touch memory (c1, c2 and the pooled/hidden/gate vectors never exist)
            for bn in (self.bn1, self.bn2, self.bn3):
                bn.finalize(Mo, train)
            slc = (nxt.Xc, nxt.fold) if nxt is not None and nxt.kind == "gsm" else (None, 0)
            se = self.se
            lib.block_eval(a_src, a0, self.l1.w_bf, (self.bn1.scale, self.bn1.shift), self.wp2_f, (self.bn2.scale, self.bn2.shift),
                           self.se1.w_bf, se.fc1.bias, self.se2.w_bf, se.fc2.bias, self.l3.w_bf, (self.bn3.scale, self.bn3.shift),
                           self.out, slc[0], slc[1], self.N)
            return self.out
        HWo = self.Ho * self.Wo
        geo = (self.N, self.Hi, self.Wi, self.Ho, self.Wo, self.stride, cmid)
        pooled = False
        if (not train and e.block_fused and self.l1.Kp == cin
                and lib.block_front_takes(cin, cmid, self.Hi, self.Wi, self.stride, k0)):
            # evaluation, the blocks of stages 1 and 2: conv1's output (in the strided blocks, at the input resolution, the largest
            # tensor of the network) stays on chip between conv1 and conv2; c2 and the squeeze come out of one kernel
            self.bn1.finalize(Mi, train)
            self.bn2.finalize(Mo, train)
            lib.block_front(a_src, a0, k0, self.l1.w_bf, (self.bn1.scale, self.bn1.shift), self.wp2_f, (self.bn2.scale, self.bn2.shift),
                            self.c2, self.pooled, self.N, self.Hi, self.Wi, cin, cmid, self.stride)
            pooled = True
        else:
            lib.gemm(a_src, self.l1.w_bf, self.c1, Mi, cmid, cin, A0=a0, lda0=k0, k0=k0, stats=st(self.bn1))
            self.bn1.finalize(Mi, train)
        s2 = st(self.bn2)
        if not pooled and not train and self.pool_fused:
            # evaluation: conv2.bn is a fixed affine map, so the squeeze-excite squeeze is taken by the convolution on its way out
            self.bn2.finalize(Mo, train)
            try:
                lib.call("stes_conv3x3g_fwd_pool", p(self.c1), p(self.wp2_f), p(self.c2), p(self.bn1.scale), p(self.bn1.shift),
                         p(self.bn2.scale), p(self.bn2.shift), p(self.pooled), *geo)
                pooled = True
            except lib.StesError as err:
                if getattr(err, "rc", 0) != lib.ERR_UNSUPPORTED:
                    raise
                self.pool_fused = False             # an image too large for the window kernel: the two passes from now on
        if not pooled:
            lib.call("stes_conv3x3g_fwd", p(self.c1), p(self.wp2_f), p(self.c2), p(self.bn1.scale), p(self.bn1.shift),
                     p(s2[0]) if s2 else None, p(s2[1]) if s2 else None, *geo)
            self.bn2.finalize(Mo, train)
            # squeeze-excite: mean over H,W -> fc1 + ReLU -> fc2 -> sigmoid, the two FCs as MFMA GEMMs over the frames
            lib.call("stes_chan_pool", p(self.c2), p(self.bn2.scale), p(self.bn2.shift), None, None, p(self.pooled), self.N, HWo, cmid, 1.0 / HWo)
        se = self.se
        # (fc1 + ReLU and fc2 + sigmoid in one launch: two launches of a few microseconds sat in the chain between two large kernels)
        lib.gemm2(self.pooled, self.se1.w_bf, self.hidden, self.se2.w_bf, self.segate, self.N, cmid, self.rd, cmid,
                  bias1=se.fc1.bias, act1=1, bias2=se.fc2.bias, act2=2, ldw1=self.se1.Kp, ldw2=self.se2.Kp)
        def shortcut():
            if self.down:
                rowmap = (self.stride, self.Ho, self.Wo, self.Hi, self.Wi) if self.stride > 1 else None
                lib.gemm(a_src, self.ld.w_bf, self.sc, Mo, cout, cin, rowmap=rowmap, stats=st(self.bnd))
                self.bnd.finalize(Mo, train)
                return self.sc, self.bnd.scale, self.bnd.shift
            return a_src, None, None
        kw3 = dict(scale=self.bn2.scale, shift=self.bn2.shift, relu=True, gate=self.segate, gate_group=HWo)
        src3 = self.c2
        if self.a2 is not None:
            lib.call("stes_bn_act_gate", p(self.c2), p(self.bn2.scale), p(self.bn2.shift), p(self.segate), p(self.a2), self.N, HWo, cmid)
            src3, kw3 = self.a2, {}
        if not train and self.a2 is None and lib.gemm_has_tail(Mo, cout, cmid, gate_group=HWo):
            # evaluation: conv3.bn is a fixed affine map, so relu(bn3(conv3) + shortcut) leaves the conv3 GEMM's store phase
            # and the raw conv3 output never exists (one pass over the block output less)
            res, rs, rb = shortcut()
            self.bn3.finalize(Mo, train)
            slc = (nxt.Xc, nxt.fold) if nxt is not None and nxt.kind == "gsm" else (None, 0)
            lib.gemm(self.c2, self.l3.w_bf, self.out, Mo, cout, cmid, tail=(self.bn3.scale, self.bn3.shift, res, rs, rb) + slc, **kw3)
            return self.out
        lib.gemm(src3, self.l3.w_bf, self.c3, Mo, cout, cmid, stats=st(self.bn3), **kw3)
        self.bn3.finalize(Mo, train)
        res, rs, rb = shortcut()
        e.bn_act(self.c3, self.bn3, res, rs, rb, self.out, nxt, train, Mo, cout)
        return self.out

    # ------------------------------------------------------------------ backward
    def backward(self, dZ, sums_done=False):
        """dZ: gradient w.r.t. this block's output, already masked by (out > 0).  Returns the masked gradient
        w.r.t. the block input; the BatchNorm-backward sums of the layer in front of this block (the previous block's
        conv3/shortcut BatchNorms, or the stem's) are accumulated on the way."""
        e = self.eng
        p, g = lib.ptr, e.grad
        # Weight gradients are off the critical path (nothing in backward reads them): they run on a side stream beside the
        # input-gradient chain.  Scratch tensors alternate between two sets by block parity, and this block may reuse its set
        # only once the side stream has finished the block that used it last.
        par = self.index & 1
        sc = lambda name, numel, dtype=BF: e.scratch(f"{name}.p{par}", numel, dtype)
        main = torch.cuda.current_stream()
        if e.side_done[par] is not None:
            main.wait_event(e.side_done[par])

        def side(fn):
            if not overlap():
                return fn()
            e.side.wait_stream(main)
            with torch.cuda.stream(e.side):
                fn()
        Mi, Mo, cin, cmid, cout = self.Mi, self.Mo, self.cin, self.cmid, self.cout
        HWo = self.Ho * self.Wo
        X, a_src = self.X, self.a_src
        bn1, bn2, bn3 = self.bn1, self.bn2, self.bn3
        # --- conv3.bn (+ shortcut bn) backward
        # (the sums of dZ against conv3's -- and the shortcut conv's -- raw output: taken by the kernel that produced dZ, the
        # block behind this one, on its way out; only the last block's dZ comes from the pooling backward)
        if not sums_done:
            sums = [bn3.s_dy, bn3.s_dyr] + ([self.bnd.s_dyr] if self.down else [])
            lib.bn_bwd_reduce(dZ, self.c3, Mo, cout, sums, raw2=self.sc if self.down else None)
        bn3.bwd_finalize(Mo)
        # Where the input-gradient GEMM runs on the streaming kernel (stages 1-3) it applies the BatchNorm backward itself, in
        # its operand prologue, and writes the result (dc3) on the way for the weight gradient; elsewhere a separate pass does.
        dc3, dA3 = sc("dc3", Mo * cout), sc("dA3", Mo * cmid)
        if lib.gemm_streams(Mo, cmid, cout):
            lib.gemm(dZ, self.l3.wT_bf, dA3, Mo, cmid, cout, ldw=self.l3.Np, bn_bwd=(self.c3, bn3.coef, dc3))
        else:
            lib.bn_bwd_apply(dZ, self.c3, Mo, cout, bn3.coef, dc3)
            lib.gemm(dc3, self.l3.wT_bf, dA3, Mo, cmid, cout, ldw=self.l3.Np)
        # --- conv3 (1x1) weight gradient
        if self.a2 is not None:
            side(lambda: lib.wgrad(dc3, self.a2, g(self.l3.w), Mo, cout, cmid, ws=e.wgrad_ws(cout, cmid)))
        else:
            side(lambda: lib.wgrad(dc3, self.c2, g(self.l3.w), Mo, cout, cmid, scale=bn2.scale, shift=bn2.shift, relu=True,
                                   gate=self.segate, gate_group=HWo, ws=e.wgrad_ws(cout, cmid)))
        # --- squeeze-excite backward
        dgate = sc("se.dgate", self.N * cmid, F32)
        S4 = sc("se.S4", 4 * self.N * cmid, F32)
        dpre2, dpre1, dpool = sc("se.dpre2", self.N * cmid), sc("se.dpre1", self.N * self.rd), sc("se.dpool", self.N * cmid, F32)
        se, rd = self.se, self.rd
        lib.call("stes_se_bwd_pool", p(self.c2), p(bn2.scale), p(bn2.shift), p(dA3), p(S4), p(dgate), p(self.segate), p(dpre2),
                 self.N, HWo, cmid)
        lib.gemm2(dpre2, self.se2.wT_bf, dpre1, self.se1.wT_bf, dpool, self.N, cmid, rd, cmid, relu_mask1=self.hidden, ldmask1=rd,
                  out_scale=1.0 / HWo, ldw1=self.se2.Np, ldw2=self.se1.Np)
        def se_wgrads():
            lib.wgrad(dpre2, self.hidden, g(se.fc2.weight), self.N, cmid, rd, dbias=g(se.fc2.bias))
            lib.wgrad(dpre1, self.pooled, g(se.fc1.weight), self.N, rd, cmid, dbias=g(se.fc1.bias))
        side(se_wgrads)
        # --- conv2.bn backward: dY2 = (dA3*gate + dpool) * [bn2(c2) > 0]
        kw2 = dict(mul=self.segate, add=dpool, group=HWo, act=(self.c2, bn2.scale, bn2.shift))
        lib.call("stes_se_bwd_bnsums", p(S4), p(self.segate), p(dpool), p(bn2.s_dy), p(bn2.s_dyr), self.N, cmid)
        bn2.bwd_finalize(Mo)
        dc2 = sc("dc2", Mo * cmid)
        lib.bn_bwd_apply(dA3, self.c2, Mo, cmid, bn2.coef, dc2, **kw2)
        # --- grouped 3x3 conv
        geo = (self.N, self.Hi, self.Wi, self.Ho, self.Wo, self.stride, cmid)
        side(lambda: lib.call("stes_conv3x3g_wgrad", p(dc2), p(self.c1), p(bn1.scale), p(bn1.shift), p(g(self.w2)), *geo))
        dA2 = sc("dA2", Mi * cmid)
        # conv1.bn backward, first half: the input-gradient kernel masks with [bn1(c1) > 0] and accumulates the sums
        lib.call("stes_conv3x3g_dgrad", p(dc2), p(self.wp2_d), p(dA2), p(self.c1), p(bn1.scale), p(bn1.shift), p(bn1.s_dy), p(bn1.s_dyr), *geo)
        bn1.bwd_finalize(Mi)
        dc1 = sc("dc1", Mi * cmid)
        # --- conv1 (1x1): input gradient (with the BatchNorm backward in its prologue where it streams), weight gradient
        q0, k0 = (self.G, self.fold) if self.kind == "gsm" else (None, 0)
        dY1 = sc("dY1", Mi * cin)
        dG = sc("gsm.dG", Mi * self.fold) if self.kind == "gsm" else None
        kw1 = dict(ldw=self.l1.Np, C0=dG, ldc0=self.fold, n_split=self.fold)
        if lib.gemm_streams(Mi, cin, cmid) and lib.gemm_fuses_wgrad(Mi, cin, cmid):
            # input gradient and weight gradient in one launch: the BatchNorm-backward result (1.6 GB in stage 1) stays on chip
            lib.gemm(dA2, self.l1.wT_bf, dY1, Mi, cin, cmid, bn_bwd=(self.c1, bn1.coef, None),
                     wgrad_fused=(a_src, g(self.l1.w), q0, k0, k0, self.fold), **kw1)
        else:
            if lib.gemm_streams(Mi, cin, cmid):
                lib.gemm(dA2, self.l1.wT_bf, dY1, Mi, cin, cmid, bn_bwd=(self.c1, bn1.coef, dc1), **kw1)
            else:
                lib.bn_bwd_apply(dA2, self.c1, Mi, cmid, bn1.coef, dc1)
                lib.gemm(dc1, self.l1.wT_bf, dY1, Mi, cin, cmid, **kw1)
            side(lambda: lib.wgrad(dc1, a_src, g(self.l1.w), Mi, cmid, cin, Q0=q0, ldq0=k0, k0=k0, perm_fold=self.fold, ws=e.wgrad_ws(cmid, cin)))
        dXf = None
        if self.kind == "gsm":
            gs, bng, geom, Xc, fold = self.gsm, self.bng, self.geom, self.Xc, self.fold
            dpre = sc("gsm.dpre", Mi * 2, F32)
            lib.call("stes_gsm_bwd_gate", p(dG), p(Xc), p(self.gate), p(dpre), C.byref(geom))
            dxbn = sc("gsm.dxbn", Mi * fold)
            if fold >= 32:               # gate-conv backward in one kernel: dxbn, its BatchNorm3d sums, conv3D weight/bias gradient
                lib.call("stes_gsm_bwd_conv", p(dpre), p(Xc), p(bng.scale), p(bng.shift), p(gs.conv3D.weight), p(dxbn), p(bng.s_dy),
                         p(bng.s_dyr), p(g(gs.conv3D.weight)), p(g(gs.conv3D.bias)), p(e.gsm_ws(fold)), C.byref(geom))
            else:                        # narrow slices: dxbn and its sums in one kernel, the conv3D gradients beside it
                side(lambda: lib.call("stes_gsm_bwd_wgrad", p(dpre), p(Xc), p(bng.scale), p(bng.shift), p(g(gs.conv3D.weight)),
                                      p(g(gs.conv3D.bias)), C.byref(geom)))
                lib.call("stes_gsm_bwd_conv", p(dpre), p(Xc), p(bng.scale), p(bng.shift), p(gs.conv3D.weight), p(dxbn), p(bng.s_dy),
                         p(bng.s_dyr), None, None, None, C.byref(geom))
            bng.bwd_finalize(Mi)
            dXf = sc("gsm.dXf", Mi * fold)
            lib.call("stes_gsm_bwd_dx", p(dG), p(Xc), p(self.gate), p(dxbn), p(bng.alpha), p(bng.beta), p(bng.delta), p(dXf), C.byref(geom))
        elif self.kind == "tsm":
            if self.tsm_inplace:
                pass            # the adjoint shift is taken by stes_block_in_grad_tsm below while it reads dY1
            else:
                dY1s = sc("tsm.dY1s", Mi * cin)
                lib.call("stes_tsm_shift", p(dY1), p(dY1s), self.tsm_fold, cin, self.N // self.T, self.T, self.Hi, self.Wi, 1)
                dY1 = dY1s
        # --- shortcut
        if self.down:
            bnd = self.bnd
            bnd.bwd_finalize(Mo, s_dy=bn3.s_dy)
            dsc, Dsc = sc("dsc", Mo * cout), sc("Dsc", Mo * cin)
            rowmap = (self.stride, self.Ho, self.Wo, self.Hi, self.Wi) if self.stride > 1 else None
            if lib.gemm_streams(Mo, cin, cout) and lib.gemm_fuses_wgrad(Mo, cin, cout):
                lib.gemm(dZ, self.ld.wT_bf, Dsc, Mo, cin, cout, ldw=self.ld.Np, bn_bwd=(self.sc, bnd.coef, None),
                         wgrad_fused=(a_src, g(self.ld.w), None, 0, 0, 0, rowmap))
            else:
                if lib.gemm_streams(Mo, cin, cout):
                    lib.gemm(dZ, self.ld.wT_bf, Dsc, Mo, cin, cout, ldw=self.ld.Np, bn_bwd=(self.sc, bnd.coef, dsc))
                else:
                    lib.bn_bwd_apply(dZ, self.sc, Mo, cout, bnd.coef, dsc)
                    lib.gemm(dsc, self.ld.wT_bf, Dsc, Mo, cin, cout, ldw=self.ld.Np)
                side(lambda: lib.wgrad(dsc, a_src, g(self.ld.w), Mo, cout, cin, rowmap=rowmap))
            resid, ms = Dsc, self.stride
        else:
            resid, ms = dZ, 1
        dZp = e.scratch_pp("dZ", Mi * cin)
        # the ReLU that produced the block input is masked with the tensor the block actually consumed
        # (for the in-place plain shift of the reference that is the shifted tensor, model/shift.py:55-66)
        if self.index > 0:
            prev = e.blocks[self.index - 1]
            front = (prev.c3, prev.sc if prev.down else None, prev.bn3.s_dy, prev.bn3.s_dyr, prev.bnd.s_dyr if prev.down else None)
        else:
            front = (e.c0, None, e.bn0.s_dy, e.bn0.s_dyr, None)
        if self.kind == "tsm" and self.tsm_inplace:
            lib.call("stes_block_in_grad_tsm", p(dY1), cin, None, 0, p(resid), ms, self.Ho, self.Wo, self.Hi, self.Wi, p(a_src),
                     p(dZp), Mi, cin, *[p(t) for t in front], p(self.xbits), self.tsm_fold, self.T)
        else:
            lib.call("stes_block_in_grad", p(dY1), cin, p(dXf), self.fold, p(resid), ms, self.Ho, self.Wo, self.Hi, self.Wi, p(a_src),
                     p(dZp), Mi, cin, *[p(t) for t in front], p(self.xbits))
        e.side_done[par] = e.side.record_event()
        return dZp


class GRULayer:
    def __init__(self, eng, gru, l, in_dim, H):
        sfx = ("", "_reverse")
        P = lambda n: [getattr(gru, f"{n}_l{l}{s}") for s in sfx]
        self.wih, self.whh, self.bih, self.bhh = P("weight_ih"), P("weight_hh"), P("bias_ih"), P("bias_hh")
        self.I, self.H = in_dim, H
        if H % 32 or in_dim % 8:
            raise NotImplementedError("GRU kernels need hidden % 32 == 0 and input % 8 == 0")
        # the two directions are adjacent in the flat parameter buffer (Engine._flatten order): view them as one tensor
        self.wih_all = eng.pair_view(self.wih)          # [6H, I]
        self.bih_all = eng.pair_view(self.bih)          # [6H]
        self.whh_all = eng.pair_view(self.whh)          # [2*3H, H]
        self.bhh_all = eng.pair_view(self.bhh)          # [2*3H]
        self.lin = Linear(eng, self.wih_all, n=6 * H, k=in_dim)
        self.whh_bf, self.whhT_bf = eng.bf16(2, 3 * H, H), eng.bf16(2, H, 3 * H)
        for d in range(2):
            eng.casts.append((self.whh[d], self.whh_bf[d], 3 * H, H, H, 3 * H, 0, 0))
            eng.casts.append((self.whh[d], self.whhT_bf[d], 3 * H, H, 3 * H, H, 1, 0))


class Engine:
    STATS_FLOATS = 1 << 20

    def __init__(self, impl: nn.Module, clip_len: int):
        dev = next(impl.parameters()).device
        if dev.type != "cuda":
            raise lib.StesError("spot_amd runs on an MI355X through libstes_hip; the model must be on a cuda (ROCm) device")
        lib.lib()
        self.impl, self.dev, self.clip_len = impl, dev, clip_len
        self.casts, self.packs, self.gsm_preps = [], [], []
        self.input_mean, self.input_std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)   # uint8 input only
        self.side = torch.cuda.Stream(device=dev)      # weight-gradient kernels (see Block.backward)
        self.comm = torch.cuda.Stream(device=dev)      # hands finished gradient buckets to the all-reduce (see backward)
        self.side_done = [None, None]
        self._stats = torch.zeros(self.STATS_FLOATS, device=dev)
        self._stats_used = 0
        self._scratch, self._acts, self._pp = {}, {}, {}
        self._cast_table = None
        self.bns = []                                  # every BatchNorm site (BN.__init__ registers itself)
        self._eval_bn_key, self._eval_bn_live = None, False
        self.block_fused = os.environ.get("STES_BLOCK_FUSED", "1") != "0"    # evaluation: stage-3 identity blocks as one kernel each
        self._grad_views = None
        self._grad_of = {}
        self._flatten()
        feats = impl._features
        self.in_chans = feats.stem.conv.in_channels
        if feats.stem.conv.out_channels != 32:
            raise NotImplementedError("stem kernels are built for 32 output channels")
        self.stem = Linear(self, feats.stem.conv.weight, kp=32)    # [32][9*Cin] -> K padded to 32
        self.bn0 = BN(self, feats.stem.bn)
        self.blocks = []
        for sname in ("s1", "s2", "s3", "s4"):
            for bname, blk in getattr(feats, sname).named_children():
                self.blocks.append(Block(self, blk, f"{sname}.{bname}"))
                self.blocks[-1].index = len(self.blocks) - 1
        self.feat_dim = self.blocks[-1].cout
        self._build_heads()
        self.weights_dirty = True
        self.shape_key = None
        self.step_count = 0
        self.dropout_seed = 0x5EED
        self.dropout_mask_override = None
        self.loss_buf = torch.zeros(8, device=dev)

    # ------------------------------------------------------------------ memory helpers
    def f32_block(self, n, Cc):
        t = torch.zeros(n, Cc, device=self.dev)
        return [t[i] for i in range(n)]

    def stats_block(self, n, Cc):
        cp = _rup(Cc, 4)
        if self._stats_used + n * cp > self.STATS_FLOATS:
            raise lib.StesError("statistics arena exhausted")
        out = []
        for _ in range(n):
            out.append(self._stats[self._stats_used:self._stats_used + Cc])
            self._stats_used += cp
        return out

    def bf16(self, *shape):
        return torch.zeros(*shape, dtype=BF, device=self.dev)

    def scratch(self, name, numel, dtype=BF):
        t = self._scratch.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            if t is not None:
                t.record_stream(self.side)       # the side stream may still be reading the tensor being replaced
            t = torch.empty(numel, dtype=dtype, device=self.dev)
            self._scratch[name] = t
        return t

    def wgrad_ws(self, n1, n2):
        """Slab workspace of the 320 x 320 weight-gradient kernel (None for other shapes).  One buffer: the weight gradients
        that use it run one after the other on one stream."""
        if n1 != 320 or n2 != 320:
            return None
        t = self._scratch.get("wgrad.ws")
        if t is None:
            t = self._scratch["wgrad.ws"] = torch.empty(lib.lib().stes_wgrad_ws_floats(), device=self.dev)
        return t

    def gsm_ws(self, fold):
        """Workspace of stes_gsm_bwd_conv: zero when first handed over, left zero by every call."""
        n = lib.lib().stes_gsm_bwd_conv_ws_floats(fold)
        t = self._scratch.get("gsm.ws")
        if t is None or t.numel() < n:
            t = self._scratch["gsm.ws"] = torch.zeros(n, device=self.dev)
        return t

    def scratch_pp(self, name, numel):
        """ping-pong scratch: the result of one block's backward is the input of the next"""
        i = self._pp.get(name, 0) ^ 1
        self._pp[name] = i
        return self.scratch(f"{name}.{i}", numel)

    def act(self, name, numel, dtype):
        t = self._acts.get(name)
        if t is None or t.numel() != numel or t.dtype != dtype:
            t = torch.empty(numel, dtype=dtype, device=self.dev)
            self._acts[name] = t
        return t

    def bn_act(self, c, bn, res, rs, rb, out, nxt, train, M, Cc):
        """out = relu(bn(c) + residual); for a following gated-shift block also its compact fold slice and, in training,
        the BatchNorm3d sums of that slice."""
        p = lib.ptr
        fold, ssum, ssq, slc = 0, None, None, None
        if nxt is not None and nxt.kind == "gsm":
            fold, slc = nxt.fold, nxt.Xc
            if train:
                ssum, ssq = nxt.bng.sum, nxt.bng.sumsq
        if nxt is not None and nxt.kind == "tsm" and nxt.tsm_inplace and train:
            # the next block's temporal shift is taken in this kernel's store: `out` is the shifted tensor
            lib.call("stes_bn_act_residual_tsm", p(c), p(bn.scale), p(bn.shift), p(res), p(rs), p(rb), p(out), M, Cc, nxt.tsm_fold,
                     self.T, M // self.N)
            nxt.pre_shifted = True
            return
        bits = nxt.xbits if nxt is not None and train else None
        lib.call("stes_bn_act_residual", p(c), p(bn.scale), p(bn.shift), p(res), p(rs), p(rb), p(out), p(ssum), p(ssq), fold, p(slc), M, Cc,
                 p(bits))

    # ------------------------------------------------------------------ flat parameters
    def _flatten(self):
        impl = self.impl
        named = [(n, p) for n, p in impl.named_parameters() if p.requires_grad]
        # GRU: put the two directions of each tensor kind next to each other; minGRU: the gate and the candidate
        # projection of a layer (so that each pair is one [2H, I] matrix in the flat buffer)
        def key(item):
            n = item[0]
            if "._gru." in n or n.startswith("_pred_fine._gru."):
                base = n.replace("_reverse", "")
                return (base, 1 if n.endswith("_reverse") else 0)
            if ".rnn_layers." in n and (".linear_z." in n or ".linear_h_tilde." in n):
                base = n.replace(".linear_h_tilde.", ".linear_z.")
                return (base, 1 if ".linear_h_tilde." in n else 0)
            return None
        gru = sorted([it for it in named if key(it) is not None], key=key)
        rest = [it for it in named if key(it) is None]
        rest_main = [it for it in rest if "_pred_loc" not in it[0]]
        # backbone first, whatever the registration order (the gradient buckets below are contiguous ranges)
        main = ([it for it in rest_main if it[0].startswith("_features.")]
                + [it for it in rest_main if not it[0].startswith("_features.")] + gru)
        loc = [it for it in rest if "_pred_loc" in it[0]]
        order = main + loc
        offs, total = {}, 0
        for n, p_ in order:
            offs[n] = total
            total += _rup(p_.numel(), 16)
        self.n_main = sum(_rup(p_.numel(), 16) for _, p_ in main)
        self.n_total = total
        # gradient buckets for the data-parallel all-reduce, in the order the backward pass completes them: heads
        # (end of the buffer) first, then s4, s3, and finally s2 + s1 + stem
        def first(prefix):
            return min(offs[n] for n, _ in order if n.startswith(prefix))
        heads_lo = min(offs[n] for n, _ in order if not n.startswith("_features."))
        self.buckets = {"heads": (heads_lo, total), "s4": (first("_features.s4."), heads_lo),
                        "s3": (first("_features.s3."), first("_features.s4.")), "rest": (0, first("_features.s3."))}
        self.flat = torch.zeros(total, device=self.dev)
        self.flat_grad = torch.zeros(total, device=self.dev)
        self.adam_m = self.adam_v = None
        self._goff = {}
        self._params = []
        for n, p_ in order:
            o = offs[n]
            view = self.flat[o:o + p_.numel()].view(p_.shape)
            view.copy_(p_.data)
            p_.data = view
            self._goff[id(p_)] = (o, p_.numel(), p_.shape)
            self._params.append((n, p_))
        self._views_ptr = {id(p_): p_.data_ptr() for _, p_ in order}

    def check_views(self):
        for _, p_ in self._params:
            if p_.data_ptr() != self._views_ptr[id(p_)]:
                raise lib.StesError("model parameters were re-allocated after the engine was built (call .to(device) before the first forward)")

    def grad(self, param):
        v = self._grad_of.get(id(param))
        if v is None:
            o, n, shape = self._goff[id(param)]
            v = self._grad_of[id(param)] = self.flat_grad[o:o + n].view(shape)
        return v

    def pair_view(self, params):
        o0, n0, s0 = self._goff[id(params[0])]
        o1, n1, s1 = self._goff[id(params[1])]
        assert o1 == o0 + n0 and n0 == n1 and n0 % 16 == 0, "GRU direction tensors must be adjacent in the flat buffer"
        t = self.flat[o0:o0 + 2 * n0].view(2 * s0[0], *s0[1:])
        self._goff[id(t)] = (o0, 2 * n0, t.shape)
        self._keep = getattr(self, "_keep", []) + [t]
        return t

    def attach_grads(self, zero):
        """Expose the flat gradient buffer as .grad of every parameter (torch optimizers then work unchanged)."""
        if zero:
            self.flat_grad.zero_()
        if self._grad_views is None:          # the views are made once: ~300 tensor constructions per step would stall the launch queue
            self._grad_views = []
            for _, p_ in self._params:
                o, n, shape = self._goff[id(p_)]
                self._grad_views.append(self.flat_grad[o:o + n].view(shape))
        for (_, p_), v in zip(self._params, self._grad_views):
            if p_.grad is not v:
                p_.grad = v

    # ------------------------------------------------------------------ heads
    def _build_heads(self):
        impl = self.impl
        F = self.feat_dim
        pf = impl._pred_fine
        self.head_kind = "gru" if isinstance(pf, GRUPrediction) else ("mingru" if isinstance(pf, MinRNNPredictor) else "fc")
        if self.head_kind == "gru":
            H = pf._gru.hidden_size
            self.gru_layers = [GRULayer(self, pf._gru, l, F if l == 0 else 2 * H, H) for l in range(pf._gru.num_layers)]
            self.fc = Linear(self, pf._fc_out._fc_out.weight, pf._fc_out._fc_out.bias)
            self.drop_p = float(pf._dropout.p)
        elif self.head_kind == "mingru":
            self.fc = Linear(self, pf.fc.weight, pf.fc.bias)
            # gate and candidate projections of a layer as one [2H, I] matrix (adjacent in the flat buffer, see _flatten)
            self.min_lin = []
            for layer in pf.rnn_layers:
                w = self.pair_view([layer.linear_z.weight, layer.linear_h_tilde.weight])
                b = self.pair_view([layer.linear_z.bias, layer.linear_h_tilde.bias])
                self.min_lin.append(Linear(self, w, b))
        else:
            self.fc = Linear(self, pf._fc_out.weight, pf._fc_out.bias)
        # channel attention over the clip (reference train_e2e_spatial.py:348-354,437-438): two bias-free linears
        self.ca = None
        if getattr(impl, "_use_channel_attention", False):
            fc = impl._channel_attention.fc
            self.ca = (Linear(self, fc[0].weight), Linear(self, fc[2].weight))
            if self.ca[0].N % 8:
                raise NotImplementedError("channel attention: reduced width must be a multiple of 8")
        self.loc = None
        if getattr(impl, "_predict_location", False):
            mlp, last = impl._pred_loc[0], impl._pred_loc[1]
            self.loc = [Linear(self, l.weight, l.bias) for l in mlp.layers] + [Linear(self, last.weight, last.bias)]

    # ------------------------------------------------------------------ weights
    def _bn_key(self):
        return sum(b.mod.running_mean._version + b.mod.running_var._version for b in self.bns)

    def refresh_weights(self):
        self.check_views()
        self._eval_bn_key = None
        p = lib.ptr
        if self._cast_table is None:
            # one launch for every bf16 weight copy: the descriptor table lives on the device (sources are views into the
            # flat parameter buffer and destinations are fixed, so it is built once)
            arr = (lib.CastDesc * len(self.casts))()
            blk = 0
            for d, (src, dst, rows, cols, ld, dst_rows, tr, pf) in zip(arr, self.casts):
                d.src, d.dst, d.rows, d.cols, d.ld, d.dst_rows, d.transpose, d.perm_fold, d.block0 = (
                    src.data_ptr(), dst.data_ptr(), rows, cols, ld, dst_rows, tr, pf, blk)
                blk += (dst_rows * ld + lib.CAST_CHUNK - 1) // lib.CAST_CHUNK
            self._cast_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev)
            self._cast_blocks = blk
        lib.call("stes_cast_bf16_multi", p(self._cast_table), len(self.casts), self._cast_blocks)
        for w3d, wt, fold in self.gsm_preps:
            lib.call("stes_gsm_prep", p(w3d), p(wt), fold)
        for w, wf, wd, G, stride in self.packs:
            lib.call("stes_pack_conv3x3g", p(w), p(wf), p(wd), G, stride)
        self.weights_dirty = False

    # ------------------------------------------------------------------ planning
    def plan(self, B, T, H, W, train, head_T=None):
        Th = head_T if head_T is not None else T
        key = (B, T, H, W, train, Th)
        if key == self.shape_key:
            return
        self._acts.clear()
        N = B * T
        self.B, self.T, self.N, self.H, self.W = B, T, N, H, W
        self.H0, self.W0 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        self.M0 = N * self.H0 * self.W0
        a = self.act
        self.patches = a("stem.patches", self.M0 * 32, BF)
        self.c0 = a("stem.c0", self.M0 * 32, BF)
        self.X0 = a("stem.X0", self.M0 * 32, BF)
        h, w = self.H0, self.W0
        for blk in self.blocks:
            h, w = blk.plan(N, h, w, T, train)
        self.Hl, self.Wl = h, w
        Fd = self.feat_dim
        self.feat = a("feat", N * Fd, BF)
        # feat_in: what the heads see (the attended features when channel attention is on)
        self.feat_in = self.feat
        if self.ca is not None:
            self.feat_in = a("ca.out", N * Fd, BF)
            self.ca_pool, self.ca_hidden = a("ca.pool", B * Fd, BF), a("ca.hidden", B * self.ca[0].N, BF)
            self.ca_gate = a("ca.gate", B * Fd, F32)
        # heads may run on the first Th <= T frames of every clip (a zero-padded last window is un-padded before the
        # heads, reference train_e2e_spatial.py:440-442)
        self.Th, self.Nh = Th, B * Th
        self.feat_h = self.feat_in if Th == T else a("feat_h", B * Th * Fd, BF)
        N_bb, N, T = N, B * Th, Th
        if self.head_kind == "gru":
            for i, L in enumerate(self.gru_layers):
                Hh = L.H
                L.gi = a(f"gru{i}.gi", N * 6 * Hh, BF)
                L.y = a(f"gru{i}.y", N * 2 * Hh, BF)
                L.hstate = a(f"gru{i}.hs", lib.lib().stes_gru_ws_floats(B, Hh, 0), F32)     # large enough for the one-launch recurrence
                L.saved = a(f"gru{i}.saved", 2 * T * B * 5 * Hh, F32) if train else None
                L.hprev = a(f"gru{i}.hprev", 2 * T * B * Hh, BF) if train else None
            Hh = self.gru_layers[-1].H
            self.ydrop = a("gru.ydrop", N * 2 * Hh, BF)
            self.dmask = a("gru.mask", N * 2 * Hh, BF)
        elif self.head_kind == "mingru":
            Hh = self.impl._pred_fine.hidden_size
            self.min_pre = [a(f"min{i}.pre", N * 2 * Hh, BF) for i in range(len(self.min_lin))]
            self.min_h = [a(f"min{i}.h", N * Hh, BF) for i in range(len(self.min_lin))]
        self.logits = a("logits", N * self.fc.N, F32)
        if self.loc is not None:
            self.loc_h1 = a("loc.h1", N * self.loc[0].N, BF)
            self.loc_h2 = a("loc.h2", N * self.loc[1].N, BF)
            self.loc_out = a("loc.out", N * 2, F32)
        self.shape_key = key
        N, T = N_bb, self.T

    # ------------------------------------------------------------------ forward
    def forward(self, x, train, head_T=None, pad_T=None):
        """x: [B, T, C, H, W] on the device (T == clip_len when a temporal shift is present): fp32 as the reference loader
        hands it over (already normalised), or uint8 decoded frames, which the stem normalises itself with `input_mean` /
        `input_std` (reference dataset/frame.py:16-17,29,124).  A uint8 batch of shorter clips is padded to pad_T frames
        inside the stem kernel (zero frames after normalisation, reference train_e2e_spatial.py:427-430).
        Returns (logits [B,Th,classes] f32, loc [B,Th,2] f32 | None, feat [B,Th,F] bf16), Th = head_T or T."""
        B, Tx, Cin, H, W = x.shape
        if Cin != self.in_chans:
            raise lib.StesError(f"expected {self.in_chans} input channels, got {Cin}")
        u8 = x.dtype == torch.uint8
        T = pad_T if (u8 and pad_T) else Tx
        if T < Tx:
            raise lib.StesError(f"clips of {Tx} frames cannot be padded to {T}")
        if not u8 and (x.dtype != F32 or not x.is_contiguous()):
            x = x.contiguous().float()
        if u8 and not x.is_contiguous():
            x = x.contiguous()
        self.plan(B, T, H, W, train, head_T)
        if self.weights_dirty or train:
            self.refresh_weights()
        # Evaluation folds the running statistics into scale/shift (61 launches of ~5 us, 2.5 % of a forward pass): once, until
        # the weights or the statistics change.  Our own kernels change them only in a training forward (which resets the key
        # through refresh_weights); changes from outside (load_state_dict, in-place tensor ops) show in the version counters.
        bn_key = None if train else self._bn_key()
        self._eval_bn_live = bn_key is not None and bn_key == self._eval_bn_key
        self._eval_bn_key = None
        p = lib.ptr
        N = self.N
        if train:
            self._stats[:self._stats_used].zero_()
        self._x = x
        mean = std = None
        if u8:
            mean = (C.c_float * 3)(*self.input_mean[:Cin], *([0.0] * (3 - Cin)))
            std = (C.c_float * 3)(*self.input_std[:Cin], *([1.0] * (3 - Cin)))
        blocks = self.blocks
        if train:
            # one pass: patches (the weight gradient reads them), the raw output and its column statistics
            lib.call("stes_stem_conv", p(x), int(u8), mean, std, p(self.stem.w_bf), p(self.patches), p(self.c0), p(self.bn0.sum),
                     p(self.bn0.sumsq), None, None, None, 0, B, T, Tx, Cin, H, W)
            self.bn0.finalize(self.M0, train)
            self.bn_act(self.c0, self.bn0, None, None, None, self.X0, blocks[0], train, self.M0, 32)
        else:
            # evaluation: BatchNorm + ReLU (and the first block's gated-shift slice) in the convolution's store phase
            self.bn0.finalize(self.M0, train)
            slc, fold = (blocks[0].Xc, blocks[0].fold) if blocks[0].kind == "gsm" else (None, 0)
            lib.call("stes_stem_conv", p(x), int(u8), mean, std, p(self.stem.w_bf), None, p(self.X0), None, None, p(self.bn0.scale),
                     p(self.bn0.shift), p(slc), fold, B, T, Tx, Cin, H, W)
        X = self.X0
        for i, blk in enumerate(blocks):
            X = blk.forward(X, train, blocks[i + 1] if i + 1 < len(blocks) else None)
        HWl = self.Hl * self.Wl
        lib.call("stes_chan_pool", p(X), None, None, None, None, p(self.feat), N, HWl, self.feat_dim, 1.0 / HWl)
        self._last_out = X
        self._eval_bn_key = bn_key                     # every site is folded now
        if self.ca is not None:
            # gate[b] = sigmoid(W2 relu(W1 mean_t feat[b,t])) over all T frames of the (padded) clip; feat_in = feat * gate
            c1, c2 = self.ca
            Fd = self.feat_dim
            lib.call("stes_chan_pool", p(self.feat), None, None, None, None, p(self.ca_pool), B, T, Fd, 1.0 / T)
            lib.gemm(self.ca_pool, c1.w_bf, self.ca_hidden, B, c1.N, Fd, act=1)
            lib.gemm(self.ca_hidden, c2.w_bf, self.ca_gate, B, Fd, c1.N, act=2)
            lib.call("stes_clip_gate_fwd", p(self.feat), p(self.ca_gate), p(self.feat_in), B, T, Fd)
        if self.Th != T:
            self.feat_h.view(B, self.Th, self.feat_dim).copy_(self.feat_in.view(B, T, self.feat_dim)[:, :self.Th])
        return self._heads_forward(train)

    def _heads_forward(self, train):
        N, B, T, Fd = self.Nh, self.B, self.Th, self.feat_dim
        feat = self.feat_h
        p = lib.ptr
        loc = None
        main = torch.cuda.current_stream()
        loc_on_side = False
        if self.loc is not None:
            # the location MLP does not depend on the temporal head, and the recurrence (a hundred small dependent launches)
            # leaves most of the GPU idle: the MLP runs beside it on the side stream
            loc_on_side = overlap() and self.head_kind == "gru"
            if loc_on_side:
                self.side.wait_stream(main)
            with torch.cuda.stream(self.side) if loc_on_side else contextlib.nullcontext():
                l1, l2, l3 = self.loc
                lib.gemm(feat, l1.w_bf, self.loc_h1, N, l1.N, l1.K, bias=l1.b, act=1)
                lib.gemm(self.loc_h1, l2.w_bf, self.loc_h2, N, l2.N, l2.K, bias=l2.b)
                lib.gemm(self.loc_h2, l3.w_bf, self.loc_out, N, l3.N, l3.K, bias=l3.b)
            loc = self.loc_out.view(B, T, 2)
        if self.head_kind == "gru":
            xin, in_dim = feat, Fd
            for L in self.gru_layers:
                lib.gemm(xin, L.lin.w_bf, L.gi, N, 6 * L.H, in_dim, bias=L.bih_all)
                a = lib.gru_args(gi=L.gi, whh=L.whh_bf, whhT=L.whhT_bf, bhh=L.bhh_all, y=L.y, hstate=L.hstate, saved=L.saved,
                                 hprev_bf=L.hprev, B=B, T=T, H=L.H, ws_floats=L.hstate.numel())
                lib.check(lib.lib().stes_gru_forward(C.byref(a), lib.stream()), "stes_gru_forward")
                xin, in_dim = L.y, 2 * L.H
            self.fc_in = xin
            self.mask_used = None
            if train and (self.drop_p > 0 or self.dropout_mask_override is not None):
                if self.dropout_mask_override is not None:
                    self.dmask.copy_(self.dropout_mask_override.reshape(-1).to(BF))
                    self._mul(xin, self.dmask, self.ydrop)
                else:
                    self.dropout_seed = (self.dropout_seed * 1664525 + 1013904223) & 0xFFFFFFFF
                    lib.call("stes_dropout", p(xin), p(self.ydrop), p(self.dmask), xin.numel(), self.drop_p, self.dropout_seed)
                self.fc_in, self.mask_used = self.ydrop, self.dmask
            lib.gemm(self.fc_in, self.fc.w_bf, self.logits, N, self.fc.N, in_dim, bias=self.fc.b)
        elif self.head_kind == "mingru":
            xin, in_dim = feat, Fd
            for i, lin in enumerate(self.min_lin):
                Hh = lin.N // 2
                lib.gemm(xin, lin.w_bf, self.min_pre[i], N, 2 * Hh, in_dim, bias=lin.b)
                lib.call("stes_mingru_scan_fwd", p(self.min_pre[i]), p(self.impl._pred_fine.h0[i]), p(self.min_h[i]), B, T, Hh)
                xin, in_dim = self.min_h[i], Hh
            self.fc_in = xin
            lib.gemm(xin, self.fc.w_bf, self.logits, N, self.fc.N, in_dim, bias=self.fc.b)
        else:
            self.fc_in = feat
            lib.gemm(feat, self.fc.w_bf, self.logits, N, self.fc.N, Fd, bias=self.fc.b)
        if loc_on_side:
            main.wait_stream(self.side)
        return self.logits.view(B, T, self.fc.N), loc, feat.view(B, T, Fd)

    def _mul(self, a, m, out):
        # injected dropout mask (parity tests): out = a * m through the BatchNorm-apply kernel's arithmetic is overkill;
        # a [N*2H] x mask elementwise is expressed with the dropout kernel's output format by the caller.
        torch.mul(a, m, out=out)

    # ------------------------------------------------------------------ backward
    def backward(self, dlogits_bf, dloc_bf, dfeat_extra=None, on_ready=None):
        """dlogits_bf [N, 8k] bf16 (padded rows), dloc_bf [N, 8] bf16 or None.  Accumulates into the flat gradient buffer.
        on_ready(lo, hi) is called as soon as flat_grad[lo:hi] is final (gradient all-reduce overlapped with backward)."""
        N, B, T, Fd = self.N, self.B, self.T, self.feat_dim
        p, g, sc = lib.ptr, self.grad, self.scratch
        if self.Th != T:
            raise NotImplementedError("backward needs clips of exactly clip_len frames")
        dfa = sc("dfeat.a", N * Fd, F32)
        dfb = None
        fc = self.fc
        ncp = dlogits_bf.shape[1]
        has_loc = self.loc is not None and dloc_bf is not None

        def loc_backward():
            l1, l2, l3 = self.loc
            lib.wgrad(dloc_bf, self.loc_h2, g(l3.w), N, l3.N, l3.K, ldp=8, dbias=g(l3.b))
            dh2 = sc("loc.dh2", N * l2.N)
            lib.gemm(dloc_bf, l3.wT_bf, dh2, N, l2.N, 8, ldw=l3.Np)
            lib.wgrad(dh2, self.loc_h1, g(l2.w), N, l2.N, l2.K, dbias=g(l2.b))
            dh1 = sc("loc.dh1", N * l1.N)
            lib.gemm(dh2, l2.wT_bf, dh1, N, l1.N, l2.N, ldw=l2.Np, relu_mask=self.loc_h1, ldmask=l1.N)
            lib.wgrad(dh1, self.feat_in, g(l1.w), N, l1.N, Fd, dbias=g(l1.b))
            d = sc("dfeat.b", N * Fd, F32)
            lib.gemm(dh1, l1.wT_bf, d, N, Fd, l1.N, ldw=l1.Np)
            return d
        # the location MLP's backward beside the recurrence's (see _heads_forward)
        main0 = torch.cuda.current_stream()
        loc_on_side = overlap() and has_loc and self.head_kind == "gru"
        if loc_on_side:
            self.side.wait_stream(main0)
            with torch.cuda.stream(self.side):
                dfb = loc_backward()
        if self.head_kind == "gru":
            Hh = self.gru_layers[-1].H
            in_dim = 2 * Hh
            lib.wgrad(dlogits_bf, self.fc_in, g(fc.w), N, fc.N, in_dim, ldp=ncp, dbias=g(fc.b))
            dy = sc("gru.dy", N * in_dim, F32)
            lib.gemm(dlogits_bf, fc.wT_bf, dy, N, in_dim, ncp, ldw=fc.Np)
            mask = self.mask_used
            for li in range(len(self.gru_layers) - 1, -1, -1):
                L = self.gru_layers[li]
                Hh = L.H
                dgi, dgh = sc("gru.dgi", N * 6 * Hh), sc("gru.dgh", 2 * T * B * 3 * Hh)
                dhs = sc("gru.dhs", lib.lib().stes_gru_ws_floats(B, Hh, 1), F32)
                a = lib.gru_args(gi=L.gi, whh=L.whh_bf, whhT=L.whhT_bf, bhh=L.bhh_all, y=L.y, hstate=L.hstate, saved=L.saved,
                                 hprev_bf=L.hprev, dy=dy, ld_dy=2 * Hh, mask=mask, dgi=dgi, dgh=dgh, dhstate=dhs, B=B, T=T, H=Hh,
                                 ws_floats=dhs.numel())
                lib.check(lib.lib().stes_gru_backward(C.byref(a), lib.stream()), "stes_gru_backward")
                xin = self.feat_in if li == 0 else self.gru_layers[li - 1].y
                lib.wgrad(dgi, xin, g(L.wih_all), N, 6 * Hh, L.I, dbias=g(L.bih_all))
                gw, gb = g(L.whh_all), g(L.bhh_all)
                dgh_v, hp_v = dgh.view(2, T * B, 3 * Hh), L.hprev.view(2, T * B, Hh)
                for d in range(2):
                    lib.wgrad(dgh_v[d], hp_v[d], gw[d * 3 * Hh:(d + 1) * 3 * Hh], T * B, 3 * Hh, Hh, dbias=gb[d * 3 * Hh:(d + 1) * 3 * Hh])
                if li == 0:
                    lib.gemm(dgi, L.lin.wT_bf, dfa, N, Fd, 6 * Hh, ldw=L.lin.Np)
                else:
                    dy = sc(f"gru.dy{li}", N * L.I, F32)
                    lib.gemm(dgi, L.lin.wT_bf, dy, N, L.I, 6 * Hh, ldw=L.lin.Np)
                mask = None
        elif self.head_kind == "mingru":
            Hh = self.min_lin[-1].N // 2
            lib.wgrad(dlogits_bf, self.fc_in, g(fc.w), N, fc.N, Hh, ldp=ncp, dbias=g(fc.b))
            dh = sc("min.dh", N * Hh, F32)
            lib.gemm(dlogits_bf, fc.wT_bf, dh, N, Hh, ncp, ldw=fc.Np)
            for i in range(len(self.min_lin) - 1, -1, -1):
                lin = self.min_lin[i]
                Hh = lin.N // 2
                dpre = sc("min.dpre", N * 2 * Hh)
                lib.call("stes_mingru_scan_bwd", p(self.min_pre[i]), p(self.impl._pred_fine.h0[i]), p(self.min_h[i]), p(dh), Hh, p(dpre), B, T, Hh)
                xin = self.feat_in if i == 0 else self.min_h[i - 1]
                lib.wgrad(dpre, xin, g(lin.w), N, 2 * Hh, lin.K, dbias=g(lin.b))
                if i == 0:
                    lib.gemm(dpre, lin.wT_bf, dfa, N, Fd, 2 * Hh, ldw=lin.Np)
                else:
                    dh = sc(f"min.dh{i}", N * lin.K, F32)
                    lib.gemm(dpre, lin.wT_bf, dh, N, lin.K, 2 * Hh, ldw=lin.Np)
        else:
            lib.wgrad(dlogits_bf, self.feat_in, g(fc.w), N, fc.N, Fd, ldp=ncp, dbias=g(fc.b))
            lib.gemm(dlogits_bf, fc.wT_bf, dfa, N, Fd, ncp, ldw=fc.Np)
        if has_loc and not loc_on_side:
            dfb = loc_backward()
        if loc_on_side:
            main0.wait_stream(self.side)
        if dfeat_extra is not None:
            dfa.add_(dfeat_extra.reshape(-1).float())
        if self.ca is not None:
            # y = x * gate(mean_t x): d gate = sum_t dy * x, through the two linears back to the mean, then
            # dx = dy * gate + d mean / T  (written over dfa; dfb is consumed here)
            c1, c2 = self.ca
            dgate, dpool = sc("ca.dgate", B * Fd, F32), sc("ca.dpool", B * Fd, F32)
            dpre2, dpre1 = sc("ca.dpre2", B * Fd), sc("ca.dpre1", B * c1.N)
            lib.call("stes_clip_gate_bwd", p(dfa), p(dfb), p(self.feat), None, None, p(dgate), None, B, T, Fd)
            lib.call("stes_sigmoid_bwd", p(dgate), p(self.ca_gate), p(dpre2), B * Fd)
            lib.gemm(dpre2, c2.wT_bf, dpre1, B, c1.N, Fd, ldw=c2.Np, relu_mask=self.ca_hidden, ldmask=c1.N)
            lib.gemm(dpre1, c1.wT_bf, dpool, B, Fd, c1.N, ldw=c1.Np, out_scale=1.0 / T)
            lib.wgrad(dpre2, self.ca_hidden, g(c2.w), B, Fd, c1.N)
            lib.wgrad(dpre1, self.ca_pool, g(c1.w), B, c1.N, Fd)
            lib.call("stes_clip_gate_bwd", p(dfa), p(dfb), None, p(self.ca_gate), p(dpool), None, p(dfa), B, T, Fd)
            dfb = None
        # --- backbone
        HWl = self.Hl * self.Wl
        dZ = self.scratch_pp("dZ", self.N * HWl * Fd)
        lib.call("stes_pool_bwd", p(dfa), p(dfb), p(self._last_out), p(dZ), N, HWl, Fd)
        if on_ready:
            on_ready(*self.buckets["heads"])
        main = torch.cuda.current_stream()
        for blk in reversed(self.blocks):
            dZ = blk.backward(dZ, sums_done=blk is not self.blocks[-1])
            if on_ready and blk.name in ("s4.b1", "s3.b1"):
                # The bucket is final once BOTH streams have passed this point.  The reduction is handed over from a third
                # stream that waits for the two, so the input-gradient chain does not stall behind the queued weight gradients.
                self.comm.wait_stream(main)
                self.comm.wait_stream(self.side)
                with torch.cuda.stream(self.comm):
                    on_ready(*self.buckets[blk.name[:2]])
        # stem: dZ is masked by (X0 > 0) already and its sums against c0 were taken by s1.b1's backward
        bn0 = self.bn0
        bn0.bwd_finalize(self.M0)
        K = self.stem.K
        # the stem has no input gradient: its BatchNorm backward is applied in the weight-gradient kernel's operand prologue
        lib.wgrad(dZ, self.patches, g(self.stem.w), self.M0, 32, K, ldq=32, lddw=K, bn_bwd=(self.c0, bn0.coef))
        main.wait_stream(self.side)                  # every gradient is final from here on (optimizer, next forward)
        if on_ready:
            on_ready(*self.buckets["rest"])

    # ------------------------------------------------------------------ fused training step
    def loss_backward(self, label, xy, num_classes, fg_weight=5.0, grad_scale=1.0, ext_denoms=None, on_ready=None):
        N = self.N
        p = lib.ptr
        ncp = _rup(num_classes, 8)
        dl = self.scratch("dlogits", N * ncp).view(N, ncp)
        dloc = self.scratch("dloc", N * 8).view(N, 8) if self.loc is not None else None
        lib.call("stes_loss", p(self.logits), num_classes, p(self.loc_out) if self.loc is not None else None, 2, p(label), p(xy), N,
                 num_classes, float(fg_weight), float(grad_scale), p(ext_denoms), p(self.loss_buf), p(dl), p(dloc))
        self.backward(dl, dloc, on_ready=on_ready)
        return self.loss_buf

    def adamw_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, loc_lr_div=5.0, grad_mul=1.0):
        """torch.optim.AdamW semantics with the reference's two learning rates (model/common.py:32-42) over the flat buffer."""
        if self.adam_m is None:
            self.adam_m, self.adam_v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.step_count += 1
        p = lib.ptr
        n0, n1 = self.n_main, self.n_total - self.n_main
        lib.call("stes_adamw", p(self.flat), p(self.flat_grad), p(self.adam_m), p(self.adam_v), n0, float(lr), betas[0], betas[1], eps,
                 weight_decay, self.step_count, float(grad_mul))
        if n1 > 0:
            o = n0
            lib.call("stes_adamw", p(self.flat[o:]), p(self.flat_grad[o:]), p(self.adam_m[o:]), p(self.adam_v[o:]), n1,
                     float(lr) / loc_lr_div, betas[0], betas[1], eps, weight_decay, self.step_count, float(grad_mul))
        self.weights_dirty = True
