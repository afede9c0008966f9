"""Each HIP kernel of libstes_hip against fp32 torch (CPU) on bf16-exact inputs, through the C-ABI.
Tolerances: outputs stored as bf16 -> 1 bf16 ulp relative (8e-3) plus accumulation noise."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def bfr(*shape, g, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def nhwc(x):           # [N,C,H,W] -> [N*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def nchw(x, n, h, w):  # [N*H*W, C] -> [N,C,H,W]
    return x.reshape(n, h, w, -1).permute(0, 3, 1, 2).contiguous()


_KEEP = []


def dev(t):
    """Host tensor -> device tensor that stays alive until the test module is torn down (raw pointers are
    handed to the C-ABI, so a temporary must not be returned to the caching allocator before the kernel ran)."""
    t = t.detach().contiguous().cuda()
    _KEEP.append(t)
    if len(_KEEP) > 512:
        torch.cuda.synchronize()
        del _KEEP[:256]
    return t


def close(a, b, rtol=1e-2, atol=1e-2):
    torch.testing.assert_close(a.float().cpu(), b.float().cpu(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("stride,H,W,C,N", [(1, 14, 14, 64, 3), (2, 16, 12, 64, 3), (1, 7, 7, 128, 3), (2, 28, 28, 128, 3), (2, 9, 7, 64, 3),
                                               (1, 28, 28, 128, 3), (2, 56, 56, 64, 3), (1, 30, 50, 64, 3), (2, 25, 13, 320, 3), (1, 13, 25, 192, 3),
                                               (2, 40, 199, 64, 3),
                                               # many images: a workgroup of the column-strip kernel owns a range of rows that starts
                                               # and ends inside images (7 rows x 1,500 images over about a thousand workgroups)
                                               (1, 7, 7, 64, 1500), (1, 4, 4, 320, 900), (1, 14, 9, 64, 333), (1, 2, 2, 128, 2000),
                                               # several images per workgroup of the LDS-window kernel on maps of fewer than 8 output pixels
                                               # (the pooled sums of the four waves need more staging space than such a map's pixels)
                                               (2, 4, 4, 128, 600), (2, 3, 3, 64, 1100), (2, 4, 6, 64, 1100),
                                               # the stride-2 row-ring kernels (maps 112, 56, 28 and 14 wide, even height): every
                                               # width, heights of one and a few output rows, more images than resident workgroups
                                               # (a workgroup's images follow one another without a pause in its row requests)
                                               (2, 112, 112, 64, 5), (2, 20, 28, 64, 3), (2, 6, 14, 64, 9), (2, 2, 56, 64, 2), (2, 2, 14, 128, 5),
                                               (2, 28, 28, 64, 1300), (2, 14, 14, 192, 2100),
                                               # the stride-1 row-ring kernel (7 wide, eight rows a step): fewer rows than a step, two
                                               # steps, a last step of one row
                                               (1, 5, 7, 64, 40), (1, 16, 7, 128, 300), (1, 9, 7, 64, 3),
                                               # the row-ring weight gradient (stride 1; 28, 14, 7 wide): heights that end inside a step
                                               # and inside a reduction block of two or four rows
                                               (1, 9, 14, 64, 5), (1, 3, 28, 64, 4), (1, 14, 14, 320, 700)])
def test_conv3x3g(stride, H, W, C, N):
    from spot_amd import lib
    g = torch.Generator().manual_seed(3)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    x = bfr(N, C, H, W, g=g)
    w = torch.randn(C, 16, 3, 3, generator=g) * 0.1
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    wb = w.to(BF).float()
    xin = torch.relu(x.float() * scale[None, :, None, None] + shift[None, :, None, None]).to(BF).float().requires_grad_(True)
    wref = wb.clone().requires_grad_(True)
    ref = F.conv2d(xin, wref, stride=stride, padding=1, groups=C // 16)
    gy = bfr(*ref.shape, g=g)
    ref.backward(gy.float())

    wp_f = torch.empty(C // 16 * 5 * 512, dtype=BF, device="cuda"); wp_d = torch.empty_like(wp_f)
    lib.call("stes_pack_conv3x3g", lib.ptr(dev(w)), lib.ptr(wp_f), lib.ptr(wp_d), C // 16, stride)
    out = torch.zeros(N * Ho * Wo, C, dtype=BF, device="cuda")
    s1 = torch.zeros(C, device="cuda"); s2 = torch.zeros(C, device="cuda")
    xd = dev(nhwc(x))
    lib.call("stes_conv3x3g_fwd", lib.ptr(xd), lib.ptr(wp_f), lib.ptr(out), lib.ptr(dev(scale)), lib.ptr(dev(shift)),
             lib.ptr(s1), lib.ptr(s2), N, H, W, Ho, Wo, stride, C)
    torch.cuda.synchronize()
    close(nchw(out.cpu(), N, Ho, Wo), ref.detach(), 1e-2, 2e-2)
    close(s1, out.float().sum(0), 1e-3, 1e-2)
    close(s2, (out.float() ** 2).sum(0), 1e-3, 1e-2)
    # forward with the squeeze-excite squeeze taken on the way out (evaluation): the same output bits, the pooled values of
    # stes_chan_pool up to the order of the fp32 sum, the same bits again on a second run and for a sub-batch
    ps, pb = dev(torch.rand(C, generator=g) + 0.5), dev(torch.randn(C, generator=g) * 0.3)
    out2 = torch.zeros_like(out)
    pooled = torch.full((N, C), 9.0, dtype=BF, device="cuda")
    args = (lib.ptr(dev(scale)), lib.ptr(dev(shift)), lib.ptr(ps), lib.ptr(pb))
    try:
        lib.call("stes_conv3x3g_fwd_pool", lib.ptr(xd), lib.ptr(wp_f), lib.ptr(out2), *args, lib.ptr(pooled), N, H, W, Ho, Wo, stride, C)
        fused_pool = True
    except lib.StesError:
        fused_pool = False                # refused for images too wide for the LDS window kernel, never silently wrong
        assert W > 150
    want = torch.empty(N, C, dtype=BF, device="cuda")
    lib.call("stes_chan_pool", lib.ptr(out), lib.ptr(ps), lib.ptr(pb), None, None, lib.ptr(want), N, Ho * Wo, C, 1.0 / (Ho * Wo))
    if fused_pool:
        pooled_b = torch.empty(1, C, dtype=BF, device="cuda")
        lib.call("stes_conv3x3g_fwd_pool", lib.ptr(xd), lib.ptr(wp_f), lib.ptr(torch.empty_like(out)), *args, lib.ptr(pooled_b), 1, H, W, Ho, Wo, stride, C)
        torch.cuda.synchronize()
        if (stride == 1 and W in (7, 14, 28)) or (stride == 2 and H % 2 == 0 and W in (112, 56, 28, 14)):
            # these shapes' plain forward is the row-ring kernel (K = 16 MFMAs, one tap each), the pooled forward the window
            # kernel (K = 32, two taps each): the fp32 sums are taken in another order
            close(out2.float(), out.float(), 1e-2, 2e-2)
        else:
            assert torch.equal(out2.view(torch.int16), out.view(torch.int16))
        close(pooled.float(), want.float(), 1e-2, 1e-3)
        assert torch.equal(pooled_b[0].view(torch.int16), pooled[0].view(torch.int16))
    # dgrad
    din = torch.zeros(N * H * W, C, dtype=BF, device="cuda")
    gyd = dev(nhwc(gy))
    lib.call("stes_conv3x3g_dgrad", lib.ptr(gyd), lib.ptr(wp_d), lib.ptr(din), None, None, None, None, None, N, H, W, Ho, Wo, stride, C)
    torch.cuda.synchronize()
    close(nchw(din.cpu(), N, H, W), xin.grad, 1e-2, 2e-2)
    # dgrad fused with the ReLU mask of the producing BatchNorm and its backward sums
    dinm = torch.zeros(N * H * W, C, dtype=BF, device="cuda")
    sdy = torch.zeros(C, device="cuda"); sdyr = torch.zeros(C, device="cuda")
    lib.call("stes_conv3x3g_dgrad", lib.ptr(gyd), lib.ptr(wp_d), lib.ptr(dinm), lib.ptr(xd), lib.ptr(dev(scale)), lib.ptr(dev(shift)),
             lib.ptr(sdy), lib.ptr(sdyr), N, H, W, Ho, Wo, stride, C)
    torch.cuda.synchronize()
    xr = nhwc(x).float()
    on = (xr * scale + shift) > 0
    assert torch.equal(dinm.cpu(), torch.where(on, din.cpu(), torch.zeros((), dtype=BF)))
    close(sdy, dinm.float().cpu().sum(0), 1e-3, 1e-2)
    close(sdyr, (dinm.float().cpu() * xr).sum(0), 1e-3, 2e-2)
    # wgrad
    dw = torch.zeros(C, 16, 3, 3, device="cuda")
    lib.call("stes_conv3x3g_wgrad", lib.ptr(gyd), lib.ptr(xd), lib.ptr(dev(scale)), lib.ptr(dev(shift)), lib.ptr(dw),
             N, H, W, Ho, Wo, stride, C)
    torch.cuda.synchronize()
    close(dw, wref.grad, 1e-2, 2e-2 * math.sqrt(N * Ho * Wo))


@pytest.mark.parametrize("stride,H,W,C,N", [(2, 28, 28, 64, 5), (2, 14, 14, 128, 7), (1, 14, 14, 64, 9), (1, 7, 7, 128, 11), (1, 28, 28, 64, 3)])
def test_conv3x3g_row_ring_without_input_transform_and_without_sums(stride, H, W, C, N):
    """The row-ring kernels with their optional parts off: no BatchNorm + ReLU on the way in, no sums on the way out (forward
    and weight gradient on the raw input), and the same launch twice gives the same bits (a workgroup's images follow one
    another through one ring: nothing of an earlier image may be left in it)."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(11)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    x = bfr(N, C, H, W, g=g)
    w = torch.randn(C, 16, 3, 3, generator=g) * 0.1
    xin = x.float().requires_grad_(True)
    wref = w.to(BF).float().requires_grad_(True)
    ref = F.conv2d(xin, wref, stride=stride, padding=1, groups=C // 16)
    gy = bfr(*ref.shape, g=g)
    ref.backward(gy.float())
    wp_f = torch.empty(C // 16 * 5 * 512, dtype=BF, device="cuda"); wp_d = torch.empty_like(wp_f)
    lib.call("stes_pack_conv3x3g", lib.ptr(dev(w)), lib.ptr(wp_f), lib.ptr(wp_d), C // 16, stride)
    xd, gyd = dev(nhwc(x)), dev(nhwc(gy))
    outs = []
    for _ in range(2):
        out = torch.full((N * Ho * Wo, C), 3.0, dtype=BF, device="cuda")
        lib.call("stes_conv3x3g_fwd", lib.ptr(xd), lib.ptr(wp_f), lib.ptr(out), None, None, None, None, N, H, W, Ho, Wo, stride, C)
        outs.append(out)
    dw = torch.zeros(C, 16, 3, 3, device="cuda")
    lib.call("stes_conv3x3g_wgrad", lib.ptr(gyd), lib.ptr(xd), None, None, lib.ptr(dw), N, H, W, Ho, Wo, stride, C)
    torch.cuda.synchronize()
    close(nchw(outs[0].cpu(), N, Ho, Wo), ref.detach(), 1e-2, 2e-2)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    close(dw, wref.grad, 1e-2, 2e-2 * math.sqrt(N * Ho * Wo))


def test_gru_one_launch_timeout_is_loud():
    """A wait of the one-launch GRU recurrence that runs out must not pass unnoticed: with the spin bound at zero (every wait
    runs out) the outputs carry NaN and the next stes_gru_* call reports the failure; with
    the bound restored the same call works again."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(5)
    B, T, H = 4, 12, 768
    gi = (torch.randn(B, T, 2, 3 * H, generator=g) * 0.7).to(BF).cuda()
    whh = (torch.randn(2, 3 * H, H, generator=g) / H ** 0.5).to(BF).cuda()
    bhh = (torch.randn(2, 3 * H, generator=g) * 0.1).cuda()

    def run():
        y = torch.zeros(B, T, 2 * H, dtype=BF, device="cuda")
        hs = torch.zeros(lib.lib().stes_gru_ws_floats(B, H, 0), device="cuda")
        a = lib.gru_args(gi=gi, whh=whh, whhT=None, bhh=bhh, y=y, hstate=hs, saved=None, hprev_bf=None, B=B, T=T, H=H, ws_floats=hs.numel())
        rc = lib.lib().stes_gru_forward(C.byref(a), lib.stream())
        torch.cuda.synchronize()
        return rc, y

    rc, good = run()
    assert rc == 0 and bool(torch.isfinite(good.float()).all())
    try:
        assert lib.lib().stes_gru_set_spin_limit(0) == 0
        rc, bad = run()
        assert rc == 0                                         # the failure happens on the device, after the call returned
        assert bool(torch.isnan(bad.float()).any())
        rc, _ = run()                                          # the next call refuses, once, and names the reason
        assert rc == -2 and b"timed out" in lib.lib().stes_last_error()
    finally:
        lib.lib().stes_gru_set_spin_limit(1 << 20)
    rc, again = run()
    assert rc == 0 and torch.equal(again.view(torch.int16), good.view(torch.int16))


def test_im2col_stem():
    from spot_amd import lib
    g = torch.Generator().manual_seed(4)
    N, H, W = 2, 12, 10
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(32, 3, 3, 3, generator=g)
    Ho, Wo = H // 2, W // 2
    p = torch.empty(N * Ho * Wo, 32, dtype=BF, device="cuda")
    lib.call("stes_im2col_stem", lib.ptr(dev(x)), lib.ptr(p), N, 3, H, W)
    torch.cuda.synchronize()
    ref = F.conv2d(x.to(BF).float(), w, stride=2, padding=1)
    got = p.float().cpu()[:, :27] @ w.reshape(32, 27).t()
    close(nchw(got, N, Ho, Wo), ref, 1e-4, 1e-4)
    assert float(p.float().cpu()[:, 27:].abs().max()) == 0.0


@pytest.mark.parametrize("B,T,Tv,H,W", [(2, 4, 4, 12, 10), (3, 4, 2, 9, 15), (1, 5, 1, 7, 7)])
def test_im2col_stem_u8_equals_fp32_path_on_the_loaders_tensor(B, T, Tv, H, W):
    """Decoded uint8 frames through the stem = `.float()/255` + Normalize (reference dataset/frame.py:29,124) through the
    fp32 stem, bit for bit, including the clips padded at the end with zero frames (train_e2e_spatial.py:427-430)."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(40)
    u = torch.randint(0, 256, (B, Tv, 3, H, W), generator=g, dtype=torch.uint8)
    u[0, 0, :, 0, :3] = torch.tensor([0, 255, 128], dtype=torch.uint8)
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    xf = (u.float() / 255 - mean[:, None, None]) / std[:, None, None]
    xf = F.pad(xf, (0,) * 7 + (T - Tv,))
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    a = torch.full((B * T * Ho * Wo, 32), 7.0, dtype=BF, device="cuda"); b = torch.full_like(a, 9.0)
    lib.call("stes_im2col_stem", lib.ptr(dev(xf.reshape(B * T, 3, H, W))), lib.ptr(a), B * T, 3, H, W)
    m3, s3 = (C.c_float * 3)(*mean.tolist()), (C.c_float * 3)(*std.tolist())
    lib.call("stes_im2col_stem_u8", lib.ptr(dev(u)), lib.ptr(b), m3, s3, B, T, Tv, 3, H, W)
    torch.cuda.synchronize()
    assert torch.equal(a.cpu().view(torch.int16), b.cpu().view(torch.int16))
    if Tv < T:
        assert float(b.view(B, T, -1)[:, Tv:].float().abs().max()) == 0.0
    with pytest.raises(lib.StesError):
        lib.call("stes_im2col_stem_u8", lib.ptr(dev(u)), lib.ptr(b), m3, s3, B, Tv - 1 if Tv > 1 else 0, Tv, 3, H, W)


@pytest.mark.parametrize("B,T,Tv,H,W,u8", [(2, 3, 3, 12, 10, False), (3, 4, 2, 9, 15, True), (1, 5, 1, 7, 7, True), (2, 2, 2, 64, 64, False),
                                             (1, 2, 2, 224, 224, True)])
def test_stem_conv_one_pass_equals_patches_then_gemm(B, T, Tv, H, W, u8):
    """The one-pass stem convolution gives, bit for bit, the patches of stes_im2col_stem(_u8), and the output / column
    statistics / evaluation tail of the product of those patches with the weight matrix."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(41)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    M = B * T * Ho * Wo
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    m3, s3 = (C.c_float * 3)(*mean.tolist()), (C.c_float * 3)(*std.tolist())
    w = torch.zeros(32, 32); w[:, :27] = torch.randn(32, 27, generator=g) * 0.3
    wb = dev(w.to(BF))
    pat = torch.full((M, 32), 7.0, dtype=BF, device="cuda")
    if u8:
        x = dev(torch.randint(0, 256, (B, Tv, 3, H, W), generator=g, dtype=torch.uint8))
        lib.call("stes_im2col_stem_u8", lib.ptr(x), lib.ptr(pat), m3, s3, B, T, Tv, 3, H, W)
    else:
        x = dev(torch.randn(B * T, 3, H, W, generator=g))
        lib.call("stes_im2col_stem", lib.ptr(x), lib.ptr(pat), B * T, 3, H, W)
    torch.cuda.synchronize()
    ref = (pat.float() @ wb.float().t()).to(BF)                       # fp32 products of bf16 operands, rounded once
    got_p = torch.full_like(pat, 9.0); got = torch.full_like(pat, 9.0)
    ssum = torch.zeros(32, device="cuda"); ssq = torch.zeros(32, device="cuda")
    args = (lib.ptr(x), int(u8), m3 if u8 else None, s3 if u8 else None, lib.ptr(wb))
    lib.call("stes_stem_conv", *args, lib.ptr(got_p), lib.ptr(got), lib.ptr(ssum), lib.ptr(ssq), None, None, None, 0, B, T, Tv, 3, H, W)
    torch.cuda.synchronize()
    assert torch.equal(got_p.view(torch.int16), pat.view(torch.int16))
    # (the accumulation order over k differs between MFMA shapes: one bf16 ulp)
    close(got.float().cpu(), ref.float().cpu(), 2e-2, 1e-2)
    close(ssum.cpu(), got.float().sum(0).cpu(), 1e-3 * M ** 0.5, 1e-4)
    close(ssq.cpu(), (got.float() ** 2).sum(0).cpu(), 1e-3 * M ** 0.5, 1e-4)
    # evaluation form: relu(scale * c + shift) and the first 8 channels as a compact slice; no patches
    sc = dev(torch.rand(32, generator=g) + 0.5); sh = dev(torch.randn(32, generator=g) * 0.2)
    ev = torch.full_like(pat, 9.0); sl = torch.full((M, 8), 9.0, dtype=BF, device="cuda")
    lib.call("stes_stem_conv", *args, None, lib.ptr(ev), None, None, lib.ptr(sc), lib.ptr(sh), lib.ptr(sl), 8, B, T, Tv, 3, H, W)
    torch.cuda.synchronize()
    want = torch.relu((pat.float() @ wb.float().t()) * sc + sh)
    close(ev.float().cpu(), want.cpu(), 2e-2, 1e-2)
    assert torch.equal(sl.view(torch.int16), ev[:, :8].contiguous().view(torch.int16))
    with pytest.raises(lib.StesError):
        lib.call("stes_stem_conv", *args, None, lib.ptr(got), None, None, None, None, None, 0, B, T, Tv, 3, H, W)


@pytest.mark.parametrize("B,T,H,W,Cc,fold", [(2, 5, 4, 6, 32, 4), (1, 8, 2, 2, 64, 8), (2, 7, 3, 3, 320, 40), (3, 1, 2, 2, 32, 4)])
def test_block_tail_takes_the_next_blocks_temporal_shift_in_its_store(B, T, H, W, Cc, fold):
    """stes_bn_act_residual_tsm against stes_bn_act_residual followed by the temporal shift, bit for bit."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(B + T + Cc)
    M = B * T * H * W
    c, res = dev(bfr(M, Cc, g=g)), dev(bfr(M, Cc, g=g))
    sc, sh = dev(torch.rand(Cc, generator=g) + 0.5), dev(torch.randn(Cc, generator=g) * 0.3)
    plain = torch.empty(M, Cc, dtype=BF, device="cuda"); ref = torch.empty_like(plain)
    lib.call("stes_bn_act_residual", lib.ptr(c), lib.ptr(sc), lib.ptr(sh), lib.ptr(res), None, None, lib.ptr(plain), None, None, 0, None, M, Cc, None)
    lib.call("stes_tsm_shift", lib.ptr(plain), lib.ptr(ref), fold, Cc, B, T, H, W, 0)
    out = torch.full((M, Cc), 7.0, dtype=BF, device="cuda")
    lib.call("stes_bn_act_residual_tsm", lib.ptr(c), lib.ptr(sc), lib.ptr(sh), lib.ptr(res), None, None, lib.ptr(out), M, Cc, fold, T, H * W)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("B,T,H,W,Cc,fold", [(2, 5, 4, 6, 32, 4), (1, 8, 2, 2, 64, 8), (2, 7, 3, 3, 320, 40)])
def test_block_in_grad_takes_the_adjoint_temporal_shift_on_its_way(B, T, H, W, Cc, fold):
    """stes_block_in_grad_tsm against stes_tsm_shift(backward) followed by stes_block_in_grad, bit for bit (output and the
    BatchNorm-backward sums' inputs are the same values in the same order)."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(B + T + Cc)
    M = B * T * H * W
    dY, X, raw = dev(bfr(M, Cc, g=g)), dev(bfr(M, Cc, g=g)), dev(bfr(M, Cc, g=g))
    outs = []
    for fused in (False, True):
        out = torch.empty(M, Cc, dtype=BF, device="cuda")
        sums = [torch.zeros(Cc, device="cuda") for _ in range(2)]
        if fused:
            lib.call("stes_block_in_grad_tsm", lib.ptr(dY), Cc, None, 0, lib.ptr(dY), 1, H, W, H, W, lib.ptr(X), lib.ptr(out), M, Cc,
                     lib.ptr(raw), None, lib.ptr(sums[0]), lib.ptr(sums[1]), None, None, fold, T)
        else:
            sh = torch.empty_like(dY)
            lib.call("stes_tsm_shift", lib.ptr(dY), lib.ptr(sh), fold, Cc, B, T, H, W, 1)
            # (the residual is the unshifted tensor in both forms)
            lib.call("stes_block_in_grad", lib.ptr(sh), Cc, None, 0, lib.ptr(dY), 1, H, W, H, W, lib.ptr(X), lib.ptr(out), M, Cc,
                     lib.ptr(raw), None, lib.ptr(sums[0]), lib.ptr(sums[1]), None, None)
        torch.cuda.synchronize()
        outs.append((out, sums))
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
    for a, b in zip(outs[0][1], outs[1][1]):
        close(a.cpu(), b.cpu(), 1e-5, 1e-4)


@pytest.mark.parametrize("B,T,H,W,Cc,fold", [(2, 5, 3, 4, 32, 4), (1, 8, 2, 2, 64, 8), (3, 19, 1, 5, 320, 40), (1, 1, 2, 3, 32, 4), (2, 64, 7, 7, 768, 96), (2, 2, 3, 3, 64, 8), (1, 9, 5, 5, 128, 16)])
def test_tsm_shift_in_place_equals_the_copying_shift(B, T, H, W, Cc, fold):
    """The temporal shift (reference model/shift.py:37-52) applied in place -- only the moving channels are touched, frames walked
    in the direction the values come from -- against the out-of-place kernel, forward and adjoint, bit for bit."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(B * 100 + T)
    X = dev(bfr(B * T * H * W, Cc, g=g))
    for backward in (0, 1):
        ref = torch.empty_like(X)
        lib.call("stes_tsm_shift", lib.ptr(X), lib.ptr(ref), fold, Cc, B, T, H, W, backward)
        Y = X.clone()
        lib.call("stes_tsm_shift_inplace", lib.ptr(Y), fold, Cc, B, T, H, W, backward)
        torch.cuda.synchronize()
        assert torch.equal(Y.view(torch.int16), ref.view(torch.int16)), backward


@pytest.mark.parametrize("N,cin,cmid,Wd,fold,stride", [(5, 32, 64, 112, 8, 2), (9, 64, 128, 56, 16, 2), (600, 64, 128, 56, 16, 2), (3, 32, 64, 112, 0, 2),
                                                        (11, 128, 128, 28, 32, 1), (900, 128, 128, 28, 32, 1), (2, 128, 128, 28, 0, 1),
                                                        (7, 128, 320, 28, 32, 2), (700, 128, 320, 28, 32, 2), (3, 128, 320, 28, 0, 2),
                                                        (1, 32, 64, 112, 8, 2), (1, 128, 128, 28, 32, 1), (1, 128, 320, 28, 32, 2)])
def test_block_front_eval_equals_the_kernel_sequence(N, cin, cmid, Wd, fold, stride):
    """conv1 -> bn1/ReLU -> grouped conv2 (+ squeeze) of a bottleneck in one kernel against stes_gemm followed by
    stes_conv3x3g_fwd_pool: same roundings, so c2 agrees to fp32 summation order (an odd bf16 ulp) and the squeeze to its bf16
    rounding."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(7 * N + cin)
    M = N * Wd * Wd
    Mo = M // (stride * stride)
    X = dev(bfr(M, cin, g=g).abs())
    G = dev(bfr(M, fold, g=g).abs()) if fold else None
    W1 = dev((torch.randn(cmid, cin, generator=g) / cin ** 0.5).to(BF))
    w2 = dev(torch.randn(cmid, 16, 3, 3, generator=g) / 12.0)
    wpf = torch.empty(cmid // 16 * 5 * 512, dtype=BF, device="cuda"); wpd = torch.empty_like(wpf)
    lib.call("stes_pack_conv3x3g", lib.ptr(w2), lib.ptr(wpf), lib.ptr(wpd), cmid // 16, stride)
    bns = [(dev(torch.rand(cmid, generator=g) + 0.5), dev(torch.randn(cmid, generator=g) * 0.2)) for _ in range(2)]
    c1 = torch.empty(M, cmid, dtype=BF, device="cuda")
    ref = torch.empty(Mo, cmid, dtype=BF, device="cuda"); ref_pool = torch.empty(N, cmid, dtype=BF, device="cuda")
    if fold:
        lib.gemm(X, W1, c1, M, cmid, cin, A0=G, lda0=fold, k0=fold)
    else:
        lib.gemm(X, W1, c1, M, cmid, cin)
    lib.call("stes_conv3x3g_fwd_pool", lib.ptr(c1), lib.ptr(wpf), lib.ptr(ref), lib.ptr(bns[0][0]), lib.ptr(bns[0][1]), lib.ptr(bns[1][0]),
             lib.ptr(bns[1][1]), lib.ptr(ref_pool), N, Wd, Wd, Wd // stride, Wd // stride, stride, cmid)
    out = torch.full((Mo, cmid), 9.0, dtype=BF, device="cuda"); pool = torch.full((N, cmid), 9.0, dtype=BF, device="cuda")
    lib.block_front(X, G, fold, W1, bns[0], wpf, bns[1], out, pool, N, Wd, Wd, cin, cmid, stride)
    torch.cuda.synchronize()
    a, b = out.float().cpu(), ref.float().cpu()
    assert float((a - b).norm() / b.norm()) < 2e-3, float((a - b).norm() / b.norm())
    close(a, b, 2e-2, 2e-2)
    close(pool.float().cpu(), ref_pool.float().cpu(), 1e-2, 1e-2)


@pytest.mark.parametrize("N,with_slice,fold", [(43, True, 80), (300, False, 80), (45, False, 0), (257, True, 80), (513, False, 0)])
def test_block_eval_s3_equals_the_kernel_sequence(N, with_slice, fold):
    """A stage-3 bottleneck's evaluation forward in one kernel against the sequence it replaces (stes_gemm with the gated-shift
    split, stes_conv3x3g_fwd_pool, stes_gemm_small2, stes_gemm with the block-tail epilogue): same roundings, so the results
    agree to one bf16 ulp here and there (fp32 summation order inside the products)."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(31 + N)
    Cc, HW, rd = 320, 196, 80
    M = N * HW
    X = dev(bfr(M, Cc, g=g).abs()); G = dev(bfr(M, fold, g=g).abs()) if fold else None
    W1 = dev((torch.randn(Cc, Cc, generator=g) / Cc ** 0.5).to(BF)); W3 = dev((torch.randn(Cc, Cc, generator=g) / Cc ** 0.5).to(BF))
    w2 = dev(torch.randn(Cc, 16, 3, 3, generator=g) / 12.0)
    wpf = torch.empty(Cc // 16 * 5 * 512, dtype=BF, device="cuda"); wpd = torch.empty_like(wpf)
    lib.call("stes_pack_conv3x3g", lib.ptr(w2), lib.ptr(wpf), lib.ptr(wpd), Cc // 16, 1)
    Wse1 = dev((torch.randn(rd, Cc, generator=g) / Cc ** 0.5).to(BF)); Wse2 = dev((torch.randn(Cc, rd, generator=g) / rd ** 0.5).to(BF))
    bse1 = dev(torch.randn(rd, generator=g) * 0.1); bse2 = dev(torch.randn(Cc, generator=g) * 0.1)
    bns = [(dev(torch.rand(Cc, generator=g) + 0.5), dev(torch.randn(Cc, generator=g) * 0.2)) for _ in range(3)]
    # ---- the kernel sequence
    c1 = torch.empty(M, Cc, dtype=BF, device="cuda"); c2 = torch.empty_like(c1); ref = torch.empty_like(c1)
    pooled = torch.empty(N, Cc, dtype=BF, device="cuda"); hidden = torch.empty(N, rd, dtype=BF, device="cuda"); gate = torch.empty(N, Cc, device="cuda")
    ref_slice = torch.full((M, 80), 7.0, dtype=BF, device="cuda")
    if fold:
        lib.gemm(X, W1, c1, M, Cc, Cc, A0=G, lda0=fold, k0=fold)
    else:
        lib.gemm(X, W1, c1, M, Cc, Cc)
    lib.call("stes_conv3x3g_fwd_pool", lib.ptr(c1), lib.ptr(wpf), lib.ptr(c2), lib.ptr(bns[0][0]), lib.ptr(bns[0][1]), lib.ptr(bns[1][0]),
             lib.ptr(bns[1][1]), lib.ptr(pooled), N, 14, 14, 14, 14, 1, Cc)
    lib.gemm2(pooled, Wse1, hidden, Wse2, gate, N, Cc, rd, Cc, bias1=bse1, act1=1, bias2=bse2, act2=2)
    lib.gemm(c2, W3, ref, M, Cc, Cc, scale=bns[1][0], shift=bns[1][1], relu=True, gate=gate, gate_group=HW,
             tail=(bns[2][0], bns[2][1], X, None, None, ref_slice if with_slice else None, 80 if with_slice else 0))
    # ---- one kernel
    out = torch.full((M, Cc), 9.0, dtype=BF, device="cuda"); slc = torch.full((M, 80), 9.0, dtype=BF, device="cuda")
    lib.block_eval(X, G, W1, bns[0], wpf, bns[1], Wse1, bse1, Wse2, bse2, W3, bns[2], out, slc if with_slice else None, 80 if with_slice else 0, N)
    torch.cuda.synchronize()
    a, b = out.float().cpu(), ref.float().cpu()
    assert float((a - b).norm() / b.norm()) < 2e-3, float((a - b).norm() / b.norm())
    close(a, b, 2e-2, 2e-2)
    if with_slice:
        assert torch.equal(slc.view(torch.int16), out[:, :80].contiguous().view(torch.int16))
        assert float((slc.float() - ref_slice.float()).norm() / ref_slice.float().norm()) < 2e-3


def test_bn_act_gate_is_the_gemm_prologue_written_out():
    """stes_bn_act_gate = relu(scale * c + shift) * gate per image, rounded once; the plain product of it equals the product with
    the operand prologue bit for bit (same arithmetic in both places)."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(9)
    N, HW, Cc, No = 64, 49, 768, 128            # (large enough for both products to run on the tiled kernel)
    c = bfr(N * HW, Cc, g=g); gate = dev(torch.rand(N, Cc, generator=g))
    sc, sh = dev(torch.rand(Cc, generator=g) + 0.5), dev(torch.randn(Cc, generator=g) * 0.3)
    out = torch.full((N * HW, Cc), 7.0, dtype=BF, device="cuda")
    lib.call("stes_bn_act_gate", lib.ptr(dev(c)), lib.ptr(sc), lib.ptr(sh), lib.ptr(gate), lib.ptr(out), N, HW, Cc)
    torch.cuda.synchronize()
    want = (torch.relu(c.float().cuda() * sc + sh) * gate.repeat_interleave(HW, 0)).to(BF)
    close(out.float().cpu(), want.float().cpu(), 1e-2, 1e-3)          # (the kernel contracts scale * c + shift into one fma: one bf16 ulp)
    W = dev((torch.randn(No, Cc, generator=g) / Cc ** 0.5).to(BF))
    a = torch.empty(N * HW, No, dtype=BF, device="cuda"); b = torch.empty_like(a)
    lib.gemm(out, W, a, N * HW, No, Cc)
    lib.gemm(dev(c), W, b, N * HW, No, Cc, scale=sc, shift=sh, relu=True, gate=gate, gate_group=HW)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))


def test_bn_finalize_and_act_residual():
    from spot_amd import lib
    g = torch.Generator().manual_seed(5)
    M, Cc = 500, 64
    c = bfr(M, Cc, g=g); res = bfr(M, Cc, g=g)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    rm, rv = torch.randn(Cc, generator=g), torch.rand(Cc, generator=g) + 0.5
    bn = torch.nn.BatchNorm1d(Cc)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    bn.train()
    ref = torch.relu(bn(c.float()) + res.float())
    cf = c.float()
    s1, s2 = dev(cf.sum(0)), dev((cf * cf).sum(0))
    scale, shift, mean, inv = (torch.empty(Cc, device="cuda") for _ in range(4))
    rmd, rvd = dev(rm.clone()), dev(rv.clone())
    lib.call("stes_bn_finalize", lib.ptr(s1), lib.ptr(s2), float(M), lib.ptr(dev(gamma)), lib.ptr(dev(beta)), lib.ptr(rmd), lib.ptr(rvd),
             0.1, 1e-5, 1, lib.ptr(scale), lib.ptr(shift), lib.ptr(mean), lib.ptr(inv), Cc)
    out = torch.empty(M, Cc, dtype=BF, device="cuda")
    st1 = torch.zeros(16, device="cuda"); st2 = torch.zeros(16, device="cuda")
    slc = torch.zeros(M, 16, dtype=BF, device="cuda")
    rbits = torch.zeros(M, Cc // 8, dtype=torch.uint8, device="cuda")
    lib.call("stes_bn_act_residual", lib.ptr(dev(c)), lib.ptr(scale), lib.ptr(shift), lib.ptr(dev(res)), None, None, lib.ptr(out),
             lib.ptr(st1), lib.ptr(st2), 16, lib.ptr(slc), M, Cc, lib.ptr(rbits))
    torch.cuda.synchronize()
    close(out, ref.detach(), 1e-2, 1e-2)
    want = ((out.view(M, Cc // 8, 8) > 0).to(torch.int32) << torch.arange(8, dtype=torch.int32, device="cuda")).sum(-1).to(torch.uint8)
    assert torch.equal(rbits, want)
    close(rmd, bn.running_mean, 1e-5, 1e-5); close(rvd, bn.running_var, 1e-4, 1e-5)
    close(st1, out.float()[:, :16].sum(0), 1e-4, 1e-3); close(st2, (out.float()[:, :16] ** 2).sum(0), 1e-4, 1e-3)
    assert torch.equal(slc, out[:, :16])
    # eval mode
    lib.call("stes_bn_finalize", None, None, 0.0, lib.ptr(dev(gamma)), lib.ptr(dev(beta)), lib.ptr(dev(rm)), lib.ptr(dev(rv)),
             0.1, 1e-5, 0, lib.ptr(scale), lib.ptr(shift), None, None, Cc)
    bn.eval()
    with torch.no_grad():
        bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
        ref = torch.relu(bn(c.float()))
    lib.call("stes_bn_act_residual", lib.ptr(dev(c)), lib.ptr(scale), lib.ptr(shift), None, None, None, lib.ptr(out), None, None, 0, None, M, Cc, None)
    torch.cuda.synchronize()
    close(out, ref, 1e-2, 1e-2)


@pytest.mark.parametrize("Cc,R,HW", [(64, 8, 49), (320, 80, 30), (768, 192, 12)])
def test_se(Cc, R, HW):
    from spot_amd import lib
    g = torch.Generator().manual_seed(6)
    N = 5
    c2 = bfr(N * HW, Cc, g=g)
    scale, shift = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    w1 = (torch.randn(R, Cc, generator=g) * Cc ** -0.5).requires_grad_(True); b1 = torch.randn(R, generator=g).requires_grad_(True)
    w2 = (torch.randn(Cc, R, generator=g) * R ** -0.5).requires_grad_(True); b2 = torch.randn(Cc, generator=g).requires_grad_(True)
    a = torch.relu(c2.float() * scale + shift).view(N, HW, Cc)
    pooled = a.mean(1).detach().requires_grad_(True)
    hidden = torch.relu(pooled @ w1.t() + b1)
    gate = torch.sigmoid(hidden @ w2.t() + b2)
    dgate = torch.randn(N, Cc, generator=g)
    gate.backward(dgate)
    # the product's sequence (engine.Block): pooled/hidden in bf16, both FCs and their gradients on the MFMA GEMM kernels
    w1b, w2b = dev(w1.detach().to(BF)), dev(w2.detach().to(BF))
    w1T, w2T = dev(w1.detach().t().contiguous().to(BF)), dev(w2.detach().t().contiguous().to(BF))
    pd = torch.empty(N, Cc, dtype=BF, device="cuda")
    lib.call("stes_chan_pool", lib.ptr(dev(c2)), lib.ptr(dev(scale)), lib.ptr(dev(shift)), None, None, lib.ptr(pd), N, HW, Cc, 1.0 / HW)
    hd = torch.empty(N, R, dtype=BF, device="cuda"); gd = torch.empty(N, Cc, device="cuda")
    lib.gemm(pd, w1b, hd, N, R, Cc, bias=dev(b1.detach()), act=1)
    lib.gemm(hd, w2b, gd, N, Cc, R, bias=dev(b2.detach()), act=2)
    torch.cuda.synchronize()
    close(pd, pooled, 1e-2, 1e-2); close(hd, hidden, 2e-2, 2e-2); close(gd, gate, 2e-2, 1e-2)
    d2 = torch.empty(N, Cc, dtype=BF, device="cuda"); d1 = torch.empty(N, R, dtype=BF, device="cuda"); dp = torch.empty(N, Cc, device="cuda")
    lib.call("stes_sigmoid_bwd", lib.ptr(dev(dgate)), lib.ptr(gd), lib.ptr(d2), N * Cc)
    lib.gemm(d2, w2T, d1, N, R, Cc, relu_mask=hd, ldmask=R)
    lib.gemm(d1, w1T, dp, N, Cc, R, out_scale=0.5)
    dw2 = torch.zeros(Cc, R, device="cuda"); db2 = torch.zeros(Cc, device="cuda")
    dw1 = torch.zeros(R, Cc, device="cuda"); db1 = torch.zeros(R, device="cuda")
    lib.wgrad(d2, hd, dw2, N, Cc, R, dbias=db2)
    lib.wgrad(d1, pd, dw1, N, R, Cc, dbias=db1)
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.float().cpu() - b).norm() / (b.norm() + 1e-9))
    assert rel(dp, 0.5 * pooled.grad) < 2e-2
    for got, ref in ((dw2, w2.grad), (db2, b2.grad), (dw1, w1.grad), (db1, b1.grad)):
        assert rel(got, ref) < 2e-2
    # pooling with a multiplicand (SE backward reduction), fp32 output
    mul = bfr(N * HW, Cc, g=g); pf = torch.empty(N, Cc, device="cuda")
    lib.call("stes_chan_pool", lib.ptr(dev(c2)), lib.ptr(dev(scale)), lib.ptr(dev(shift)), lib.ptr(dev(mul)), lib.ptr(pf), None, N, HW, Cc, 1.0)
    torch.cuda.synchronize()
    close(pf, (a * mul.float().view(N, HW, Cc)).sum(1), 1e-3, 1e-3)


@pytest.mark.parametrize("Cc,HW", [(64, 49), (320, 30), (768, 12)])
def test_se_backward_pool_gives_gate_gradient_and_bn_sums(Cc, HW):
    """stes_se_bwd_pool + stes_se_bwd_bnsums against the direct formulas: dgate = sum_hw relu(bn(c2)) * dA3 and the two
    BatchNorm-backward sums of dy = (dA3*gate + dpool) * [bn(c2) > 0]."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(Cc + HW)
    N = 6
    c2, dA = bfr(N * HW, Cc, g=g), bfr(N * HW, Cc, g=g)
    scale, shift = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    gate, dpool = torch.rand(N, Cc, generator=g), torch.randn(N, Cc, generator=g) * 0.1
    x = c2.float().view(N, HW, Cc); d = dA.float().view(N, HW, Cc)
    pre = x * scale + shift
    dgate_ref = (torch.relu(pre) * d).sum(1)
    dy = (d * gate[:, None] + dpool[:, None]) * (pre > 0)
    S = torch.empty(4, N, Cc, device="cuda"); dg = torch.empty(N, Cc, device="cuda")
    dpre = torch.empty(N, Cc, dtype=BF, device="cuda")
    lib.call("stes_se_bwd_pool", lib.ptr(dev(c2)), lib.ptr(dev(scale)), lib.ptr(dev(shift)), lib.ptr(dev(dA)), lib.ptr(S), lib.ptr(dg),
             lib.ptr(dev(gate)), lib.ptr(dpre), N, HW, Cc)
    s_dy = torch.zeros(Cc, device="cuda"); s_dyr = torch.zeros(Cc, device="cuda")
    lib.call("stes_se_bwd_bnsums", lib.ptr(S), lib.ptr(dev(gate)), lib.ptr(dev(dpool)), lib.ptr(s_dy), lib.ptr(s_dyr), N, Cc)
    torch.cuda.synchronize()
    close(dg, dgate_ref, 1e-3, 1e-3)
    close(dpre.float(), dgate_ref * gate * (1 - gate), 1e-2, 1e-2)      # the sigmoid backward on the way out (bf16)
    close(s_dy, dy.sum((0, 1)), 1e-3, 1e-3)
    close(s_dyr, (dy * x).sum((0, 1)), 1e-3, 2e-3)


def test_bn_backward_chain():
    """dY = (D*mul + add) * relu-mask, then BatchNorm (training) backward: reduce -> finalize -> apply."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(7)
    N, HW, Cc = 4, 25, 64
    M = N * HW
    raw = bfr(M, Cc, g=g).float().requires_grad_(True)
    gamma = (torch.rand(Cc, generator=g) + 0.5).requires_grad_(True); beta = torch.randn(Cc, generator=g).requires_grad_(True)
    D = bfr(M, Cc, g=g); mul = torch.rand(N, Cc, generator=g); add = torch.randn(N, Cc, generator=g) * 0.1
    mean = raw.mean(0); var = raw.var(0, unbiased=False); inv = (var + 1e-5).rsqrt()
    yb = (raw - mean) * inv * gamma + beta
    act = torch.relu(yb)
    dA = (D.float().view(N, HW, Cc) * mul[:, None] + add[:, None]).reshape(M, Cc)
    act.backward(dA)
    scale = (gamma * inv).detach(); shift = (beta - mean * gamma * inv).detach()
    sums = [torch.zeros(Cc, device="cuda") for _ in range(2)]
    rawd = dev(raw.to(BF))
    kw = dict(mul=dev(mul), add=dev(add), group=HW, act=(rawd, dev(scale), dev(shift)))
    lib.bn_bwd_reduce(dev(D), rawd, M, Cc, sums, **kw)
    coef = [torch.empty(Cc, device="cuda") for _ in range(3)]
    dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
    lib.call("stes_bn_bwd_finalize", lib.ptr(sums[0]), lib.ptr(sums[1]), lib.ptr(dev(mean)), lib.ptr(dev(inv)), lib.ptr(dev(gamma)),
             float(M), lib.ptr(coef[0]), lib.ptr(coef[1]), lib.ptr(coef[2]), lib.ptr(dg), lib.ptr(db), Cc)
    out = torch.empty(M, Cc, dtype=BF, device="cuda")
    lib.bn_bwd_apply(dev(D), rawd, M, Cc, coef, out, **kw)
    torch.cuda.synchronize()
    close(dg, gamma.grad, 1e-3, 1e-3); close(db, beta.grad, 1e-3, 1e-3)
    close(out, raw.grad, 1e-2, 1e-2)


def test_block_in_grad_and_pool_bwd():
    from spot_amd import lib
    g = torch.Generator().manual_seed(8)
    N, Hi, Wi, Cc, fold = 2, 6, 8, 32, 8
    Ho, Wo = 3, 4
    M = N * Hi * Wi
    dY = bfr(M, Cc, g=g); dXf = bfr(M, fold, g=g); X = bfr(M, Cc, g=g); Dsc = bfr(N * Ho * Wo, Cc, g=g)
    up = torch.zeros(N, Hi, Wi, Cc)
    up[:, ::2, ::2] = Dsc.float().view(N, Ho, Wo, Cc)
    ref = (torch.cat([dXf.float(), dY.float()[:, fold:]], 1) + up.view(M, Cc)) * (X.float() > 0)
    out = torch.empty(M, Cc, dtype=BF, device="cuda")
    d = dev
    lib.call("stes_block_in_grad", lib.ptr(d(dY)), Cc, lib.ptr(d(dXf)), fold, lib.ptr(d(Dsc)), 2, Ho, Wo, Hi, Wi, lib.ptr(d(X)), lib.ptr(out), M, Cc,
             None, None, None, None, None, None)
    torch.cuda.synchronize()
    close(out, ref, 1e-2, 1e-2)
    ref = (torch.cat([dXf.float(), dY.float()[:, fold:]], 1) + dY.float()) * (X.float() > 0)
    lib.call("stes_block_in_grad", lib.ptr(d(dY)), Cc, lib.ptr(d(dXf)), fold, lib.ptr(d(dY)), 1, Hi, Wi, Hi, Wi, lib.ptr(d(X)), lib.ptr(out), M, Cc,
             None, None, None, None, None, None)
    torch.cuda.synchronize()
    close(out, ref, 1e-2, 1e-2)
    # the same with the BatchNorm-backward sums of the layer in front taken on the way out (of the rounded values)
    plain = out.clone()
    r1, r2 = bfr(M, Cc, g=g), bfr(M, Cc, g=g)
    sums = [torch.zeros(Cc, device="cuda") for _ in range(3)]
    lib.call("stes_block_in_grad", lib.ptr(d(dY)), Cc, lib.ptr(d(dXf)), fold, lib.ptr(d(dY)), 1, Hi, Wi, Hi, Wi, lib.ptr(d(X)), lib.ptr(out), M, Cc,
             lib.ptr(d(r1)), lib.ptr(d(r2)), *[lib.ptr(t) for t in sums], None)
    torch.cuda.synchronize()
    assert torch.equal(out, plain)
    o = out.float().cpu()
    close(sums[0], o.sum(0), 1e-4, 1e-3)
    close(sums[1], (o * r1.float()).sum(0), 1e-4, 1e-3)
    close(sums[2], (o * r2.float()).sum(0), 1e-4, 1e-3)
    sums = [torch.zeros(Cc, device="cuda") for _ in range(2)]
    lib.call("stes_block_in_grad", lib.ptr(d(dY)), Cc, lib.ptr(d(dXf)), fold, lib.ptr(d(Dsc)), 2, Ho, Wo, Hi, Wi, lib.ptr(d(X)), lib.ptr(out), M, Cc,
             lib.ptr(d(r1)), None, lib.ptr(sums[0]), lib.ptr(sums[1]), None, None)
    torch.cuda.synchronize()
    o = out.float().cpu()
    close(sums[0], o.sum(0), 1e-4, 1e-3)
    close(sums[1], (o * r1.float()).sum(0), 1e-4, 1e-3)
    # the mask as one bit per element (what stes_bn_act_residual writes beside X) instead of X itself
    bits = (X.float() > 0).view(M, Cc // 8, 8).to(torch.int32)
    bits = (bits << torch.arange(8, dtype=torch.int32)).sum(-1).to(torch.uint8)
    keep, sums2 = out.clone(), [torch.zeros(Cc, device="cuda") for _ in range(2)]
    out.zero_()
    lib.call("stes_block_in_grad", lib.ptr(d(dY)), Cc, lib.ptr(d(dXf)), fold, lib.ptr(d(Dsc)), 2, Ho, Wo, Hi, Wi, None, lib.ptr(out), M, Cc,
             lib.ptr(d(r1)), None, lib.ptr(sums2[0]), lib.ptr(sums2[1]), None, lib.ptr(d(bits)))
    torch.cuda.synchronize()
    assert torch.equal(out, keep)
    close(sums2[0], sums[0], 1e-5, 1e-4); close(sums2[1], sums[1], 1e-5, 1e-4)        # atomics: order varies
    with pytest.raises(lib.StesError):
        lib.call("stes_block_in_grad", lib.ptr(d(dY)), Cc, lib.ptr(d(dXf)), fold, lib.ptr(d(Dsc)), 2, Ho, Wo, Hi, Wi, lib.ptr(d(X)), lib.ptr(out),
                 M, Cc, lib.ptr(d(r1)), None, None, None, None, None)
    da, db = torch.randn(N, Cc, generator=g), torch.randn(N, Cc, generator=g)
    ref = ((da + db)[:, None] / (Hi * Wi) * (X.float().view(N, Hi * Wi, Cc) > 0)).reshape(M, Cc)
    lib.call("stes_pool_bwd", lib.ptr(d(da)), lib.ptr(d(db)), lib.ptr(d(X)), lib.ptr(out), N, Hi * Wi, Cc)
    torch.cuda.synchronize()
    close(out, ref, 1e-2, 1e-3)


def _gsm_run(z, mode, fused=False):
    """Run the GSM kernels on a golden case; returns y and the gradients in the reference's NCHW layout and channel order
    (the kernels keep source channel order; dest(c) is the reference's within-group interleave).  fused: the one-kernel
    forward (stes_gsm_forward) and the one-kernel gate-conv backward (stes_gsm_bwd_conv) in place of the kernel sequences."""
    from spot_amd import lib
    x = torch.from_numpy(z["x"]); gy = torch.from_numpy(z["gy"])
    NT, fold, H, W = x.shape
    T = int(z["T"]); B = NT // T
    M = NT * H * W
    half, quarter = fold // 2, fold // 4
    dest = [(c // half) * half + 2 * ((c % half) % quarter) + (c % half) // quarter for c in range(fold)]
    d = lambda a: dev(torch.from_numpy(np.asarray(a)))
    # a strided block input [M, fold + 24]: the compact slice copy and its statistics come from stes_col_stats
    Xfull = torch.zeros(M, fold + 24, dtype=BF); Xfull[:, :fold] = nhwc(x).to(BF)
    Xfull = dev(Xfull)
    Xc = torch.empty(M, fold, dtype=BF, device="cuda")
    geom = lib.gsm_geom(B, T, H, W, fold)
    s1 = torch.zeros(fold, device="cuda"); s2 = torch.zeros(fold, device="cuda")
    lib.call("stes_col_stats", lib.ptr(Xfull), fold + 24, fold, M, lib.ptr(s1), lib.ptr(s2), lib.ptr(Xc))
    scale, shift, mean, inv = (torch.empty(fold, device="cuda") for _ in range(4))
    rm, rv = d(z["rm0"]).clone(), d(z["rv0"]).clone()
    gam = d(z["bn_w"])
    lib.call("stes_bn_finalize", lib.ptr(s1), lib.ptr(s2), float(M), lib.ptr(gam), lib.ptr(d(z["bn_b"])), lib.ptr(rm), lib.ptr(rv),
             0.1, 1e-5, int(mode == "train"), lib.ptr(scale), lib.ptr(shift), lib.ptr(mean), lib.ptr(inv), fold)
    w3d, b3d = d(z["conv_w"]).contiguous(), d(z["conv_b"]).contiguous()
    wt = torch.empty(27 * fold, dtype=torch.bfloat16, device="cuda")
    lib.call("stes_gsm_prep", lib.ptr(w3d), lib.ptr(wt), fold)
    gate = torch.empty(M, 2, device="cuda")
    G = torch.empty(M, fold, dtype=BF, device="cuda")
    if fused:
        lib.call("stes_gsm_forward", lib.ptr(Xc), lib.ptr(scale), lib.ptr(shift), lib.ptr(wt), lib.ptr(b3d), lib.ptr(gate), lib.ptr(G),
                 C.byref(geom))
    else:
        lib.call("stes_gsm_gate", lib.ptr(Xc), lib.ptr(scale), lib.ptr(shift), lib.ptr(wt), lib.ptr(b3d), lib.ptr(gate), C.byref(geom))
        lib.call("stes_gsm_shift", lib.ptr(Xc), lib.ptr(gate), lib.ptr(G), C.byref(geom))
    y = torch.empty(NT, fold, H, W)
    y[:, dest] = nchw(G.float().cpu(), NT, H, W)
    res = {"y": y, "rm": rm.cpu(), "rv": rv.cpu()}
    if mode == "train":
        dG = dev(nhwc(gy[:, dest]).to(BF))                       # gradient w.r.t. G in source order
        dpre = torch.empty(M, 2, device="cuda")
        lib.call("stes_gsm_bwd_gate", lib.ptr(dG), lib.ptr(Xc), lib.ptr(gate), lib.ptr(dpre), C.byref(geom))
        dw = torch.zeros_like(w3d); db = torch.zeros(2, device="cuda")
        dxbn = torch.empty(M, fold, dtype=BF, device="cuda")
        sd = torch.zeros(fold, device="cuda"); sdx = torch.zeros(fold, device="cuda")
        if fused:                                                # (narrow slices take no workspace)
            ws = torch.zeros(lib.lib().stes_gsm_bwd_conv_ws_floats(fold), device="cuda") if fold >= 32 else None
            lib.call("stes_gsm_bwd_conv", lib.ptr(dpre), lib.ptr(Xc), lib.ptr(scale), lib.ptr(shift), lib.ptr(w3d), lib.ptr(dxbn),
                     lib.ptr(sd), lib.ptr(sdx), lib.ptr(dw), lib.ptr(db), lib.ptr(ws), C.byref(geom))
            assert ws is None or not bool(ws.any())
        else:
            lib.call("stes_gsm_bwd_wgrad", lib.ptr(dpre), lib.ptr(Xc), lib.ptr(scale), lib.ptr(shift), lib.ptr(dw), lib.ptr(db), C.byref(geom))
            lib.call("stes_gsm_dxbn", lib.ptr(dpre), lib.ptr(Xc), lib.ptr(scale), lib.ptr(shift), lib.ptr(w3d), lib.ptr(dxbn), C.byref(geom))
            lib.bn_bwd_reduce(dxbn, Xc, M, fold, [sd, sdx])
        coef = [torch.empty(fold, device="cuda") for _ in range(3)]
        dgam = torch.zeros(fold, device="cuda"); dbet = torch.zeros(fold, device="cuda")
        lib.call("stes_bn_bwd_finalize", lib.ptr(sd), lib.ptr(sdx), lib.ptr(mean), lib.ptr(inv), lib.ptr(gam), float(M),
                 lib.ptr(coef[0]), lib.ptr(coef[1]), lib.ptr(coef[2]), lib.ptr(dgam), lib.ptr(dbet), fold)
        dXf = torch.empty(M, fold, dtype=BF, device="cuda")
        lib.call("stes_gsm_bwd_dx", lib.ptr(dG), lib.ptr(Xc), lib.ptr(gate), lib.ptr(dxbn), lib.ptr(coef[0]), lib.ptr(coef[1]),
                 lib.ptr(coef[2]), lib.ptr(dXf), C.byref(geom))
        res.update(gx=nchw(dXf.float().cpu(), NT, H, W), dw=dw.cpu(), db=db.cpu(), dgam=dgam.cpu(), dbet=dbet.cpu())
    torch.cuda.synchronize()
    return res


@pytest.mark.parametrize("B,T,H,W,fold", [(2, 3, 9, 20, 8), (2, 3, 9, 30, 8), (1, 5, 13, 50, 16), (1, 3, 30, 30, 32), (1, 9, 14, 14, 80),
                                          (1, 5, 7, 7, 192), (1, 11, 40, 36, 16), (1, 4, 25, 13, 32),
                                          # full-length clips on the small maps of a 64 x 64 crop
                                          (1, 64, 16, 16, 16), (1, 64, 4, 4, 80), (2, 64, 2, 2, 192), (1, 64, 1, 1, 192)])
def test_gsm_gate_tiling(B, T, H, W, fold):
    """The gate kernel's row tiles, frame segments, ragged strips and both thread mappings against conv3d
    (reference model/shift.py:108-113) on the same bf16-rounded operands."""
    from spot_amd import lib
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B + T + H + W + fold)
    half, M = fold // 2, B * T * H * W
    Xc = torch.randn(M, fold, generator=g).to(BF)
    sc, sh = torch.rand(fold, generator=g) + 0.5, torch.randn(fold, generator=g) * 0.3
    w3d = torch.randn(2, half, 3, 3, 3, generator=g) / (27 * half) ** 0.5
    b3d = torch.randn(2, generator=g) * 0.1
    wt = torch.empty(27 * fold, dtype=BF, device="cuda")
    lib.call("stes_gsm_prep", lib.ptr(dev(w3d)), lib.ptr(wt), fold)
    gate = torch.full((M, 2), 9.0, device="cuda")
    geom = lib.gsm_geom(B, T, H, W, fold)
    lib.call("stes_gsm_gate", lib.ptr(dev(Xc)), lib.ptr(dev(sc)), lib.ptr(dev(sh)), lib.ptr(wt), lib.ptr(dev(b3d)), lib.ptr(gate), C.byref(geom))
    torch.cuda.synchronize()
    xb = torch.relu(Xc.float() * sc + sh).to(BF).float().view(B, T, H, W, fold).permute(0, 4, 1, 2, 3)
    ref = torch.tanh(F.conv3d(xb, w3d.to(BF).float(), b3d, padding=1, groups=2)).permute(0, 2, 3, 4, 1).reshape(M, 2)
    close(gate, ref, 1e-4, 1e-4)


@pytest.mark.parametrize("B,T,H,W,fold", [(1, 3, 30, 30, 32), (2, 9, 14, 14, 80), (1, 5, 7, 7, 192), (1, 4, 25, 13, 32), (1, 1, 14, 14, 80),
                                          (1, 2, 5, 40, 48), (1, 3, 3, 100, 32)])
def test_gsm_fused_forward_and_backward_conv(B, T, H, W, fold):
    """stes_gsm_forward against conv3d + the shift formula (reference model/shift.py:103-130) on the same bf16-rounded
    operands, and stes_gsm_bwd_conv against the kernel sequence it replaces, over ragged row tiles, single-frame clips, tile
    rows that are not a multiple of 16 positions and every channel-tile count."""
    from spot_amd import lib
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B + T + H + W + fold)
    half, M = fold // 2, B * T * H * W
    Xc = torch.randn(M, fold, generator=g).to(BF)
    sc, sh = torch.rand(fold, generator=g) + 0.5, torch.randn(fold, generator=g) * 0.3
    w3d = torch.randn(2, half, 3, 3, 3, generator=g) / (27 * half) ** 0.5
    b3d = torch.randn(2, generator=g) * 0.1
    wt = torch.empty(27 * fold, dtype=BF, device="cuda")
    lib.call("stes_gsm_prep", lib.ptr(dev(w3d)), lib.ptr(wt), fold)
    gate = torch.full((M, 2), 9.0, device="cuda")
    G = torch.full((M, fold), 9.0, dtype=BF, device="cuda")
    geom = lib.gsm_geom(B, T, H, W, fold)
    dXc, dsc, dsh, dw3d, db3d = dev(Xc), dev(sc), dev(sh), dev(w3d), dev(b3d)
    lib.call("stes_gsm_forward", lib.ptr(dXc), lib.ptr(dsc), lib.ptr(dsh), lib.ptr(wt), lib.ptr(db3d), lib.ptr(gate), lib.ptr(G), C.byref(geom))
    torch.cuda.synchronize()
    xb = torch.relu(Xc.float() * sc + sh).to(BF).float().view(B, T, H, W, fold).permute(0, 4, 1, 2, 3)
    ref = torch.tanh(F.conv3d(xb, w3d.to(BF).float(), b3d, padding=1, groups=2)).permute(0, 2, 3, 4, 1).reshape(M, 2)
    close(gate, ref, 1e-4, 1e-4)
    # the shift on the kernel's own gate values: bit-exact against the two-kernel path's shift
    G2 = torch.empty_like(G)
    lib.call("stes_gsm_shift", lib.ptr(dXc), lib.ptr(gate), lib.ptr(G2), C.byref(geom))
    torch.cuda.synchronize()
    assert torch.equal(G, G2)
    # backward of the gate conv
    dpre = dev(torch.randn(M, 2, generator=g) * 0.1)
    outs = []
    ws = torch.zeros(lib.lib().stes_gsm_bwd_conv_ws_floats(fold), device="cuda")
    for fused in (0, 1, 2):                                   # the kernel sequence; fused with and without the workspace
        dw = torch.zeros(2, half, 27, device="cuda"); db = torch.zeros(2, device="cuda")
        dxbn = torch.full((M, fold), 9.0, dtype=BF, device="cuda")
        sd = torch.zeros(fold, device="cuda"); sdx = torch.zeros(fold, device="cuda")
        if fused:
            lib.call("stes_gsm_bwd_conv", lib.ptr(dpre), lib.ptr(dXc), lib.ptr(dsc), lib.ptr(dsh), lib.ptr(dw3d), lib.ptr(dxbn),
                     lib.ptr(sd), lib.ptr(sdx), lib.ptr(dw), lib.ptr(db), lib.ptr(ws) if fused == 1 else None, C.byref(geom))
        else:
            lib.call("stes_gsm_bwd_wgrad", lib.ptr(dpre), lib.ptr(dXc), lib.ptr(dsc), lib.ptr(dsh), lib.ptr(dw), lib.ptr(db), C.byref(geom))
            lib.call("stes_gsm_dxbn", lib.ptr(dpre), lib.ptr(dXc), lib.ptr(dsc), lib.ptr(dsh), lib.ptr(dw3d), lib.ptr(dxbn), C.byref(geom))
            lib.bn_bwd_reduce(dxbn, dXc, M, fold, [sd, sdx])
        torch.cuda.synchronize()
        outs.append((dxbn.float().cpu(), dw.cpu(), db.cpu(), sd.cpu(), sdx.cpu()))
    # the fused kernel rounds the gate gradient and the conv weights to bf16 for MFMA: 2^-9 relative per product
    assert not bool(ws.any())                                 # the workspace is handed back zeroed
    for k in (1, 2):
        for name, a, b_ in zip(("dxbn", "dw", "db", "s_dy", "s_dyr"), outs[k], outs[0]):
            assert float((a - b_).norm() / (b_.norm() + 1e-12)) < 8e-3, (name, k)


@pytest.mark.parametrize("B,T,H,W,fold", [(2, 3, 9, 20, 8), (1, 1, 5, 7, 8), (2, 9, 9, 30, 8), (1, 5, 13, 50, 16), (1, 11, 40, 36, 16), (1, 64, 16, 16, 16),
                                          (1, 70, 3, 112, 8), (1, 2, 24, 112, 8), (2, 7, 20, 56, 16), (1, 4, 1, 1, 8), (1, 6, 2, 100, 16)])
def test_gsm_narrow_slice_forward_in_one_kernel(B, T, H, W, fold):
    """stes_gsm_forward on slices of 8 and 16 channels (frame segments with their four extra frames, ragged row tiles, rows
    that are not a multiple of 16 positions, single frames and single pixels): the gate against conv3d (reference
    model/shift.py:108-110) on the same bf16-rounded operands, the output bit for bit against the shift kernel run on the
    kernel's own gate values."""
    from spot_amd import lib
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B + T + H + W + fold)
    half, M = fold // 2, B * T * H * W
    Xc = torch.randn(M, fold, generator=g).to(BF)
    sc, sh = torch.rand(fold, generator=g) + 0.5, torch.randn(fold, generator=g) * 0.3
    w3d = torch.randn(2, half, 3, 3, 3, generator=g) / (27 * half) ** 0.5
    b3d = torch.randn(2, generator=g) * 0.1
    wt = torch.empty(27 * fold, dtype=BF, device="cuda")
    lib.call("stes_gsm_prep", lib.ptr(dev(w3d)), lib.ptr(wt), fold)
    gate = torch.full((M, 2), 9.0, device="cuda")
    G = torch.full((M, fold), 9.0, dtype=BF, device="cuda")
    geom = lib.gsm_geom(B, T, H, W, fold)
    dXc = dev(Xc)
    lib.call("stes_gsm_forward", lib.ptr(dXc), lib.ptr(dev(sc)), lib.ptr(dev(sh)), lib.ptr(wt), lib.ptr(dev(b3d)), lib.ptr(gate), lib.ptr(G), C.byref(geom))
    torch.cuda.synchronize()
    xb = torch.relu(Xc.float() * sc + sh).to(BF).float().view(B, T, H, W, fold).permute(0, 4, 1, 2, 3)
    ref = torch.tanh(F.conv3d(xb, w3d.to(BF).float(), b3d, padding=1, groups=2)).permute(0, 2, 3, 4, 1).reshape(M, 2)
    close(gate, ref, 1e-4, 1e-4)
    G2 = torch.empty_like(G)
    lib.call("stes_gsm_shift", lib.ptr(dXc), lib.ptr(gate), lib.ptr(G2), C.byref(geom))
    torch.cuda.synchronize()
    assert torch.equal(G.view(torch.int16), G2.view(torch.int16))
    # backward of the gate conv: dxbn bit for bit (same products, the tap order of the sequence's kernel), the sums within fp32
    # summation order, the conv3D gradients only when asked for
    dpre = dev(torch.randn(M, 2, generator=g) * 0.1)
    dsc, dsh, dw3d = dev(sc), dev(sh), dev(w3d)
    dxa = torch.full((M, fold), 9.0, dtype=BF, device="cuda"); dxb = torch.full_like(dxa, 7.0)
    sa = torch.zeros(2, fold, device="cuda"); sb = torch.zeros(2, fold, device="cuda")
    dwa = torch.zeros(2, half, 27, device="cuda"); dba = torch.zeros(2, device="cuda"); dwb = torch.zeros_like(dwa); dbb = torch.zeros_like(dba)
    lib.call("stes_gsm_dxbn", lib.ptr(dpre), lib.ptr(dXc), lib.ptr(dsc), lib.ptr(dsh), lib.ptr(dw3d), lib.ptr(dxa), C.byref(geom))
    lib.bn_bwd_reduce(dxa, dXc, M, fold, [sa[0], sa[1]])
    lib.call("stes_gsm_bwd_wgrad", lib.ptr(dpre), lib.ptr(dXc), lib.ptr(dsc), lib.ptr(dsh), lib.ptr(dwa), lib.ptr(dba), C.byref(geom))
    lib.call("stes_gsm_bwd_conv", lib.ptr(dpre), lib.ptr(dXc), lib.ptr(dsc), lib.ptr(dsh), lib.ptr(dw3d), lib.ptr(dxb), lib.ptr(sb[0]), lib.ptr(sb[1]),
             lib.ptr(dwb), lib.ptr(dbb), None, C.byref(geom))
    torch.cuda.synchronize()
    close(dxb.float().cpu(), dxa.float().cpu(), 1e-2, 1e-6)
    assert float((sb - sa).norm() / (sa.norm() + 1e-12)) < 1e-4
    assert torch.equal(dwa, dwb) or float((dwa - dwb).norm() / dwa.norm()) < 1e-5
    sb.zero_(); dwb.zero_()
    lib.call("stes_gsm_bwd_conv", lib.ptr(dpre), lib.ptr(dXc), lib.ptr(dsc), lib.ptr(dsh), lib.ptr(dw3d), lib.ptr(dxb), lib.ptr(sb[0]), lib.ptr(sb[1]),
             None, None, None, C.byref(geom))
    torch.cuda.synchronize()
    assert float((sb - sa).norm() / (sa.norm() + 1e-12)) < 1e-4 and not bool(dwb.any())


@pytest.mark.parametrize("fold", [8, 16, 32, 80, 192])
@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("fused", [False, True])
def test_gsm_vs_reference_golden(golden, fold, mode, fused):
    """Inputs of the golden vectors rounded to bf16 (the kernels' storage type); expected values are the
    reference's fp32 outputs -> tolerance 2 bf16 ulp + the input rounding + the gate conv weights rounded to bf16 (like
    every other conv weight on the path; up to 2592 products per gate value at fold 192).  fused: the one-kernel forward and
    the one-kernel gate-conv backward."""
    z = golden(f"gsm_f{fold}_{mode}_rand")
    r = _gsm_run(z, mode, fused)
    close(r["y"], torch.from_numpy(z["y"]), 2e-2, 5e-2)
    assert float((r["y"] - torch.from_numpy(z["y"])).norm() / torch.from_numpy(z["y"]).norm()) < 6e-3
    if mode == "train":
        close(r["rm"], torch.from_numpy(z["rm1"]), 1e-2, 2e-3); close(r["rv"], torch.from_numpy(z["rv1"]), 1e-2, 2e-3)
        gx, gxr = r["gx"], torch.from_numpy(z["gx"])
        assert float((gx - gxr).norm() / gxr.norm()) < 2e-2
        for k, kr in (("dw", "g_conv_w"), ("db", "g_conv_b"), ("dgam", "g_bn_w"), ("dbet", "g_bn_b")):
            a, b = r[k].flatten(), torch.from_numpy(z[kr]).flatten()
            assert float((a - b).norm() / (b.norm() + 1e-6)) < 3e-2, k


@pytest.mark.parametrize("fold,fused", [(8, False), (16, False), (32, False), (80, False), (192, False), (8, True), (16, True), (32, True), (80, True),
                                        (192, True)])
def test_gsm_zero_gate_shift_indexing_bit_exact(golden, fold, fused):
    """conv3D = 0 (the reference's init): output is exactly the within-group channel interleave of the bf16 input."""
    z = golden(f"gsm_f{fold}_eval_zero")
    r = _gsm_run(z, "eval", fused)
    xb = torch.from_numpy(z["x"]).to(BF).float()
    yref = torch.from_numpy(z["y"])
    # reference output is a pure permutation of its input, so the bf16 version must be the same permutation of the bf16 input
    half, quarter = fold // 2, fold // 4
    dest = [(c // half) * half + 2 * ((c % half) % quarter) + (c % half) // quarter for c in range(fold)]
    assert torch.equal(yref[:, dest], torch.from_numpy(z["x"]))
    assert torch.equal(r["y"][:, dest], xb)


@pytest.mark.parametrize("c", [32, 64, 320])
def test_tsm_bit_exact(golden, c):
    from spot_amd import lib
    z = golden(f"tsm_c{c}")
    x = torch.from_numpy(z["x"]).to(BF); T = int(z["T"])
    NT, Cc, H, W = x.shape
    X = dev(nhwc(x.float()).to(BF)); out = torch.empty_like(X)
    lib.call("stes_tsm_shift", lib.ptr(X), lib.ptr(out), Cc // 8, Cc, NT // T, T, H, W, 0)
    gy = torch.from_numpy(z["gy"]).to(BF)
    GY = dev(nhwc(gy.float()).to(BF)); gx = torch.empty_like(GY)
    lib.call("stes_tsm_shift", lib.ptr(GY), lib.ptr(gx), Cc // 8, Cc, NT // T, T, H, W, 1)
    torch.cuda.synchronize()
    from oracle import stes_oracle as O
    assert torch.equal(nchw(out.float().cpu(), NT, H, W), O.tsm_shift(x.float(), T, 8))
    xr = x.float().requires_grad_(True)
    O.tsm_shift(xr, T, 8).backward(gy.float())
    assert torch.equal(nchw(gx.float().cpu(), NT, H, W), xr.grad)


def _pack_gru(sd, l, H):
    sfx = ("", "_reverse")
    wih = torch.cat([sd[f"_gru.weight_ih_l{l}{s}"] for s in sfx], 0)
    bih = torch.cat([sd[f"_gru.bias_ih_l{l}{s}"] for s in sfx], 0)
    whh = torch.stack([sd[f"_gru.weight_hh_l{l}{s}"] for s in sfx], 0)
    bhh = torch.stack([sd[f"_gru.bias_hh_l{l}{s}"] for s in sfx], 0)
    return wih, bih, whh, bhh


def test_loc_mlp_vs_reference_golden(golden):
    """The location head (reference train_e2e_spatial.py:400-410, model/common.py:78-92: Linear-ReLU-Linear, then Linear) as
    the engine runs it -- three stes_gemm forward, stes_gemm / stes_wgrad backward -- against the reference module's own
    output, input gradient and parameter gradients."""
    from spot_amd import lib
    z = golden("loc_mlp")
    x = torch.from_numpy(z["x"]); gy = torch.from_numpy(z["gy"])
    M, I = x.shape[0] * x.shape[1], x.shape[2]
    W = [torch.from_numpy(z["p." + k + ".weight"]) for k in ("0.layers.0", "0.layers.1", "1")]
    Bs = [dev(torch.from_numpy(z["p." + k + ".bias"])) for k in ("0.layers.0", "0.layers.1", "1")]
    Wb = [dev(w.to(BF)) for w in W]
    WT = []
    for w in W:                                            # [K][N8]: transposed copies with the output width padded to 8
        t = torch.zeros(w.shape[1], (w.shape[0] + 7) // 8 * 8)
        t[:, :w.shape[0]] = w.t()
        WT.append(dev(t.to(BF)))
    xb = dev(x.reshape(M, I).to(BF))
    h1 = torch.empty(M, 96, dtype=BF, device="cuda"); h2 = torch.empty(M, 32, dtype=BF, device="cuda"); y = torch.empty(M, 2, device="cuda")
    lib.gemm(xb, Wb[0], h1, M, 96, I, bias=Bs[0], act=1)
    lib.gemm(h1, Wb[1], h2, M, 32, 96, bias=Bs[1])
    lib.gemm(h2, Wb[2], y, M, 2, 32, bias=Bs[2])
    torch.cuda.synchronize()
    close(y.cpu().reshape(z["y"].shape), torch.from_numpy(z["y"]), 2e-2, 2e-2)
    # backward, as Engine.backward runs it
    dy8 = torch.zeros(M, 8); dy8[:, :2] = gy.reshape(M, 2)
    dy8 = dev(dy8.to(BF))
    g = [torch.zeros_like(w).cuda() for w in W]; gb = [torch.zeros_like(b) for b in Bs]
    lib.wgrad(dy8, h2, g[2], M, 2, 32, ldp=8, dbias=gb[2])
    dh2 = torch.empty(M, 32, dtype=BF, device="cuda")
    lib.gemm(dy8, WT[2], dh2, M, 32, 8, ldw=WT[2].shape[1])
    lib.wgrad(dh2, h1, g[1], M, 32, 96, dbias=gb[1])
    dh1 = torch.empty(M, 96, dtype=BF, device="cuda")
    lib.gemm(dh2, WT[1], dh1, M, 96, 32, ldw=WT[1].shape[1], relu_mask=h1, ldmask=96)
    lib.wgrad(dh1, xb, g[0], M, 96, I, dbias=gb[0])
    dx = torch.empty(M, I, device="cuda")
    lib.gemm(dh1, WT[0], dx, M, I, 96, ldw=WT[0].shape[1])
    torch.cuda.synchronize()
    ref = lambda k: torch.from_numpy(z[k])
    assert float((dx.cpu().reshape(z["gx"].shape) - ref("gx")).norm() / ref("gx").norm()) < 2e-2
    for i, k in enumerate(("0.layers.0", "0.layers.1", "1")):
        assert float((g[i].cpu() - ref("g." + k + ".weight")).norm() / ref("g." + k + ".weight").norm()) < 2e-2, k
        assert float((gb[i].cpu() - ref("g." + k + ".bias")).norm() / ref("g." + k + ".bias").norm()) < 2e-2, k


@pytest.mark.parametrize("B,T,H", [(16, 64, 768), (9, 5, 256), (16, 3, 96), (3, 7, 768)])
def test_gru_one_launch_recurrence_bit_identical(B, T, H):
    """The whole recurrence in one launch (hidden state exchanged between workgroups through the workspace) against the
    launch-per-step form: forward outputs, everything saved for the backward, and the backward's gate gradients, bit for bit."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(B * 1000 + T * 10 + H)
    M = B * T
    gi = dev((torch.randn(M, 6 * H, generator=g) * 0.5).to(BF))
    whh = torch.randn(2, 3 * H, H, generator=g) / H ** 0.5
    whh_d, whhT_d = dev(whh.to(BF)), dev(whh.transpose(1, 2).contiguous().to(BF))
    bhh = dev(torch.randn(2, 3 * H, generator=g) * 0.1)
    dy = dev(torch.randn(B, T, 2 * H, generator=g))
    L = lib.lib()
    outs = []
    for one_launch in (False, True):
        nf = L.stes_gru_ws_floats(B, H, 0) if one_launch else 4 * B * H
        nb = L.stes_gru_ws_floats(B, H, 1) if one_launch else 4 * B * H
        y = torch.zeros(B, T, 2 * H, dtype=BF, device="cuda")
        hstate = torch.full((nf,), 7.0, device="cuda")        # (the workspace need not be cleared by the caller)
        saved = torch.zeros(2, T, B, 5, H, device="cuda"); hprev = torch.zeros(2, T, B, H, dtype=BF, device="cuda")
        a = lib.gru_args(gi=gi, whh=whh_d, whhT=whhT_d, bhh=bhh, y=y, hstate=hstate, saved=saved, hprev_bf=hprev, B=B, T=T, H=H,
                         ws_floats=nf if one_launch else 0)
        lib.check(L.stes_gru_forward(C.byref(a), lib.stream()), "gru fwd")
        dgi = torch.zeros(M, 6 * H, dtype=BF, device="cuda"); dgh = torch.zeros(2, T, B, 3 * H, dtype=BF, device="cuda")
        dhs = torch.full((nb,), 7.0, device="cuda")
        a = lib.gru_args(gi=gi, whh=whh_d, whhT=whhT_d, bhh=bhh, y=y, hstate=hstate, saved=saved, hprev_bf=hprev, dy=dy, ld_dy=2 * H,
                         dgi=dgi, dgh=dgh, dhstate=dhs, B=B, T=T, H=H, ws_floats=nb if one_launch else 0)
        lib.check(L.stes_gru_backward(C.byref(a), lib.stream()), "gru bwd")
        torch.cuda.synchronize()
        outs.append((y, saved, hprev, dgi, dgh))
    assert float(outs[0][0].float().abs().max()) > 0.05        # a live recurrence
    for name, a_, b_ in zip(("y", "saved", "hprev", "dgi", "dgh"), outs[0], outs[1]):
        assert torch.equal(a_, b_), name


def test_gru_layer_vs_reference_golden(golden):
    """One bidirectional GRU layer forward + BPTT against torch.nn.GRU as run by the reference's GRUPrediction."""
    from spot_amd import lib
    z = golden("gru_L1")
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}
    x = torch.from_numpy(z["x"])
    B, T, I = x.shape
    H = 48
    # hidden size 48 is not a multiple of 32: embed into H' = 64 with zero padding (zero weights keep extra units at 0)
    Hp = 64
    wih, bih, whh, bhh = _pack_gru(sd, 0, H)

    def pad_gates(w, cols=None):   # [.., 3H, X] -> [.., 3H', X']
        shp = list(w.shape)
        lead = shp[:-2] if w.dim() > 2 else []
        w3 = w.reshape(*lead, 3, H, shp[-1])
        out = torch.zeros(*lead, 3, Hp, cols or shp[-1])
        out[..., :H, :shp[-1]] = w3
        return out.reshape(*lead, 3 * Hp, cols or shp[-1])

    Ip = 64
    xp = torch.zeros(B, T, Ip); xp[..., :I] = x
    wih_p = torch.cat([pad_gates(wih[:3 * H], Ip), pad_gates(wih[3 * H:], Ip)], 0)            # [6H', I']
    bih_p = torch.cat([pad_gates(bih[:3 * H, None]).flatten(), pad_gates(bih[3 * H:, None]).flatten()])
    whh_p = torch.stack([pad_gates(whh[d], Hp) for d in range(2)])                               # [2, 3H', H']
    bhh_p = torch.stack([pad_gates(bhh[d][:, None]).flatten() for d in range(2)])
    M = B * T
    gi = torch.empty(M, 6 * Hp, dtype=BF, device="cuda")
    lib.gemm(dev(xp.reshape(M, Ip).to(BF)), dev(wih_p.to(BF)), gi, M, 6 * Hp, Ip, bias=dev(bih_p))
    y = torch.zeros(B, T, 2 * Hp, dtype=BF, device="cuda")
    hstate = torch.zeros(2, 2, B, Hp, device="cuda"); saved = torch.zeros(2, T, B, 5, Hp, device="cuda")
    hprev = torch.zeros(2, T, B, Hp, dtype=BF, device="cuda")
    whh_d = dev(whh_p.to(BF)); whhT_d = dev(whh_p.transpose(1, 2).to(BF)); bhh_d = dev(bhh_p)
    a = lib.gru_args(gi=gi, whh=whh_d, whhT=whhT_d, bhh=bhh_d, y=y, hstate=hstate, saved=saved, hprev_bf=hprev, B=B, T=T, H=Hp)
    lib.check(lib.lib().stes_gru_forward(C.byref(a), lib.stream()), "gru fwd")
    torch.cuda.synchronize()
    ref_gru = torch.nn.GRU(I, H, batch_first=True, bidirectional=True)
    ref_gru.load_state_dict({k[5:]: v for k, v in sd.items() if k.startswith("_gru.")})
    xr = x.clone().requires_grad_(True)
    yr, _ = ref_gru(xr)
    yk = y.float().cpu()
    got = torch.cat([yk[..., :H], yk[..., Hp:Hp + H]], -1)
    close(got, yr.detach(), 2e-2, 2e-2)
    assert float(yk[..., H:Hp].abs().max()) == 0.0
    # backward
    g = torch.Generator().manual_seed(1)
    dy = torch.randn(B, T, 2 * H, generator=g)
    yr.backward(dy)
    dyp = torch.zeros(B, T, 2 * Hp); dyp[..., :H] = dy[..., :H]; dyp[..., Hp:Hp + H] = dy[..., H:]
    dgi = torch.zeros(M, 6 * Hp, dtype=BF, device="cuda"); dgh = torch.zeros(2, T, B, 3 * Hp, dtype=BF, device="cuda")
    dhs = torch.zeros(2, 2, B, Hp, device="cuda")
    a = lib.gru_args(gi=gi, whh=whh_d, whhT=whhT_d, bhh=bhh_d, y=y, hstate=hstate, saved=saved, hprev_bf=hprev,
                     dy=dev(dyp), ld_dy=2 * Hp, dgi=dgi, dgh=dgh, dhstate=dhs, B=B, T=T, H=Hp)
    lib.check(lib.lib().stes_gru_backward(C.byref(a), lib.stream()), "gru bwd")
    dx = torch.empty(M, Ip, device="cuda")
    lib.gemm(dgi, dev(wih_p.t().to(BF)), dx, M, Ip, 6 * Hp)
    dwhh = torch.zeros(2, 3 * Hp, Hp, device="cuda"); dbhh = torch.zeros(2, 3 * Hp, device="cuda")
    for d in range(2):
        lib.wgrad(dgh[d], hprev[d], dwhh[d], T * B, 3 * Hp, Hp, dbias=dbhh[d])
    torch.cuda.synchronize()
    close(dx.cpu()[:, :I].reshape(B, T, I), xr.grad, 3e-2, 3e-2)
    for d, s in enumerate(("", "_reverse")):
        ref_w = getattr(ref_gru, f"weight_hh_l0{s}").grad
        got_w = dwhh[d].cpu().reshape(3, Hp, Hp)[:, :H, :H].reshape(3 * H, H)
        assert float((got_w - ref_w).norm() / ref_w.norm()) < 3e-2
        ref_b = getattr(ref_gru, f"bias_hh_l0{s}").grad
        got_b = dbhh[d].cpu().reshape(3, Hp)[:, :H].reshape(-1)
        assert float((got_b - ref_b).norm() / ref_b.norm()) < 3e-2


def test_mingru_scan():
    from spot_amd import lib
    g = torch.Generator().manual_seed(2)
    B, T, H = 3, 9, 40
    pre = bfr(B, T, 2 * H, g=g)
    pr = pre.float().requires_grad_(True)
    z = torch.sigmoid(pr[..., :H]); cand = pr[..., H:]
    h0 = torch.randn(H, generator=g) * 0.3
    h = h0.expand(B, H); hs = []
    for t in range(T):
        h = (1 - z[:, t]) * h + z[:, t] * cand[:, t]
        hs.append(h)
    hs = torch.stack(hs, 1)
    dh = torch.randn(B, T, H, generator=g)
    hs.backward(dh)
    hd = torch.empty(B, T, H, dtype=BF, device="cuda")
    h0d = dev(h0)
    lib.call("stes_mingru_scan_fwd", lib.ptr(dev(pre)), lib.ptr(h0d), lib.ptr(hd), B, T, H)
    dpre = torch.empty(B, T, 2 * H, dtype=BF, device="cuda")
    lib.call("stes_mingru_scan_bwd", lib.ptr(dev(pre)), lib.ptr(h0d), lib.ptr(hd), lib.ptr(dev(dh)), H, lib.ptr(dpre), B, T, H)
    torch.cuda.synchronize()
    close(hd, hs.detach(), 8e-3, 2e-3)
    close(dpre, pr.grad, 2e-2, 5e-3)


def test_loss_and_adamw_vs_reference_golden(golden):
    from spot_amd import lib
    from oracle import stes_oracle as O
    g = torch.Generator().manual_seed(3)
    M, Cc = 200, 7
    logits = torch.randn(M, Cc, generator=g).requires_grad_(True); loc = torch.randn(M, 2, generator=g).requires_grad_(True)
    label = torch.randint(0, Cc, (M,), generator=g); label[:50] = 0
    xy = torch.rand(M, 2, generator=g)
    loss, lc, ll = O.stes_loss(logits.view(1, M, Cc), loc.view(1, M, 2), label, xy, Cc)
    loss.backward()
    out = torch.zeros(5, device="cuda")
    dl = torch.empty(M, 8, dtype=BF, device="cuda"); dloc = torch.empty(M, 8, dtype=BF, device="cuda")
    d = dev
    lib.call("stes_loss", lib.ptr(d(logits)), Cc, lib.ptr(d(loc)), 2, lib.ptr(d(label)), lib.ptr(d(xy)), M, Cc, 5.0, 1.0, None,
             lib.ptr(out), lib.ptr(dl), lib.ptr(dloc))
    torch.cuda.synchronize()
    close(out[:3], torch.stack([loss, lc, ll]).detach(), 1e-5, 1e-5)
    close(dl[:, :Cc], logits.grad, 1e-2, 1e-5); close(dloc[:, :2], loc.grad, 1e-2, 1e-5)
    assert float(dl[:, Cc:].float().abs().max()) == 0 and float(dloc[:, 2:].float().abs().max()) == 0
    # AdamW against torch.optim.AdamW (the reference's optimiser) over 3 steps
    n = 1000
    p = torch.randn(n, generator=g); p0 = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([p0], lr=1e-2)
    pd = dev(p); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        p0.grad = gr.clone(); opt.step()
        lib.call("stes_adamw", lib.ptr(pd), lib.ptr(dev(gr)), lib.ptr(m), lib.ptr(v), n, 1e-2, 0.9, 0.999, 1e-8, 1e-2, step, 1.0)
    torch.cuda.synchronize()
    close(pd, p0.detach(), 1e-5, 1e-6)


def test_cast_and_dropout():
    from spot_amd import lib
    g = torch.Generator().manual_seed(4)
    w = torch.randn(7, 20, generator=g)
    a = torch.empty(8, 24, dtype=BF, device="cuda"); b = torch.empty(24, 8, dtype=BF, device="cuda")
    lib.call("stes_cast_bf16", lib.ptr(dev(w)), lib.ptr(a), 7, 20, 24, 8, 0, 0)
    lib.call("stes_cast_bf16", lib.ptr(dev(w)), lib.ptr(b), 7, 20, 8, 24, 1, 0)
    # gated-shift column interleave folded into the cast: destination column c reads source column dest(c)
    pa = torch.empty(8, 24, dtype=BF, device="cuda"); pb = torch.empty(24, 8, dtype=BF, device="cuda")
    lib.call("stes_cast_bf16", lib.ptr(dev(w)), lib.ptr(pa), 7, 20, 24, 8, 0, 8)
    lib.call("stes_cast_bf16", lib.ptr(dev(w)), lib.ptr(pb), 7, 20, 8, 24, 1, 8)
    x = dev(bfr(4096, g=g)); out = torch.empty_like(x); mask = torch.empty_like(x)
    lib.call("stes_dropout", lib.ptr(x), lib.ptr(out), lib.ptr(mask), 4096, 0.5, 1234)
    torch.cuda.synchronize()
    ref = torch.zeros(8, 24); ref[:7, :20] = w.to(BF).float()
    assert torch.equal(a.float().cpu(), ref)
    assert torch.equal(b.float().cpu()[:20, :7], w.to(BF).float().t()) and float(b.float().cpu()[20:].abs().max()) == 0
    dest8 = [0, 2, 1, 3, 4, 6, 5, 7]
    wperm = w.clone(); wperm[:, :8] = w[:, dest8]
    assert torch.equal(pa.float().cpu()[:7, :20], wperm.to(BF).float())
    assert torch.equal(pb.float().cpu()[:20, :7], wperm.to(BF).float().t())
    mk = mask.float().cpu()
    assert set(mk.unique().tolist()) <= {0.0, 2.0} and 0.4 < float((mk > 0).float().mean()) < 0.6
    assert torch.equal(out.float().cpu(), (x.float().cpu() * mk).to(BF).float())


def test_cast_bf16_multi_equals_single_launches():
    """One launch over a device table of weight copies = the per-tensor launches (plain, transposed, padded, fold-permuted)."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(77)
    cases = [(40, 24, 24, 40, 0, 0), (40, 24, 40, 24, 1, 0), (33, 27, 32, 40, 0, 0), (64, 96, 96, 64, 0, 16), (64, 96, 64, 96, 1, 16),
             (7, 768, 768, 8, 0, 0), (7, 768, 8, 768, 1, 0), (300, 24, 24, 300, 0, 0), (300, 24, 304, 24, 1, 0),     # (the last two: several chunks)
             (320, 768, 320, 768, 1, 80), (2304, 770, 2304, 776, 1, 0), (100, 90, 104, 90, 1, 0)]                    # (many tiles, ragged on both sides)
    srcs = [dev(torch.randn(r, c, generator=g)) for r, c, *_ in cases]
    single = [torch.full((dr * ld,), 3.0, dtype=BF, device="cuda") for _, _, ld, dr, _, _ in cases]
    multi = [torch.full((dr * ld,), 4.0, dtype=BF, device="cuda") for _, _, ld, dr, _, _ in cases]
    arr = (lib.CastDesc * len(cases))()
    blk = 0
    for d, s, a, b, (r, c, ld, dr, tr, pf) in zip(arr, srcs, single, multi, cases):
        lib.call("stes_cast_bf16", lib.ptr(s), lib.ptr(a), r, c, ld, dr, tr, pf)
        d.src, d.dst, d.rows, d.cols, d.ld, d.dst_rows, d.transpose, d.perm_fold, d.block0 = s.data_ptr(), b.data_ptr(), r, c, ld, dr, tr, pf, blk
        blk += (dr * ld + lib.CAST_CHUNK - 1) // lib.CAST_CHUNK
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    lib.call("stes_cast_bf16_multi", lib.ptr(table), len(cases), blk)
    torch.cuda.synchronize()
    for a, b in zip(single, multi):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
