"""stes_gemm / stes_wgrad against fp32 torch on bf16-exact inputs (GPU)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


_KEEP = []


def dev(t):
    # raw pointers go to the C-ABI: keep every device tensor alive until the module ends
    t = t.detach().contiguous().cuda()
    _KEEP.append(t)
    return t


def _bf(*shape, g, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K,kernel", [
    (128, 64, 32, "gemm_small_kernel"), (300, 128, 64, "gemm_small_kernel"), (1000, 320, 128, "gemm_small_kernel"),
    (77, 7, 1536, "gemm_small_kernel"), (513, 768, 320, "gemm_small_kernel"), (1024, 80, 8, "gemm_small_kernel"),
    (256, 2304, 768, "gemm_kernel<64>"),          # few 128-wide tiles -> 64-wide ones
    (4100, 320, 328, "gemm_kernel<64>"),          # a width that is a multiple of 64 but not of 128
    (16384, 384, 328, "gemm_kernel<128>"), (9000, 40, 1000, "gemm_kernel<64>")])
def test_gemm_plain(M, N, K, kernel):
    """Every dispatch target of the no-prologue product (the latency-bound small kernel up to 2^28 multiply-adds, the
    tiled kernel with 64- or 128-wide tiles) against fp32 torch, bf16 and fp32 outputs, with bias."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(M + N + K)
    A, W = _bf(M, K, g=g), _bf(N, K, g=g, scale=K ** -0.5)
    bias = torch.randn(N, generator=g)
    ref = A.float() @ W.float().t() + bias
    for out_dtype in (torch.bfloat16, torch.float32):
        C = torch.full((M, N), 7.0, dtype=out_dtype, device="cuda")
        lib.TIMING = []
        try:
            lib.gemm(dev(A), dev(W), C, M, N, K, bias=dev(bias))
            assert [t[0] for t in lib.TIMING] == [kernel]
        finally:
            lib.TIMING = None
        torch.cuda.synchronize()
        tol = 2e-2 if out_dtype == torch.bfloat16 else 2e-3
        torch.testing.assert_close(C.float().cpu(), ref, rtol=tol, atol=tol)


def test_gemm_small_kernel_epilogues():
    """The small kernel's epilogue options as the squeeze-excite backward uses them: transposed weights with a padded
    leading dimension, ReLU mask, output scale, fp32 output; sigmoid / ReLU activations; odd sizes."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(77)
    M, N, K, ldw = 1000, 72, 200, 208
    A = _bf(M, K, g=g)
    Wp = torch.zeros(N, ldw, dtype=torch.bfloat16)
    Wp[:, :K] = _bf(N, K, g=g, scale=K ** -0.5)
    mask = _bf(M, N, g=g)
    ref = (A.float() @ Wp[:, :K].float().t()) * 0.25 * (mask.float() > 0)
    C = torch.full((M, N), 7.0, device="cuda")
    lib.gemm(dev(A), dev(Wp), C, M, N, K, ldw=ldw, relu_mask=dev(mask), ldmask=N, out_scale=0.25)
    torch.testing.assert_close(C.cpu(), ref, rtol=2e-3, atol=2e-3)
    for act, fn in ((1, torch.relu), (2, torch.sigmoid)):
        Cb = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        lib.gemm(dev(A), dev(Wp), Cb, M, N, K, ldw=ldw, act=act)
        torch.testing.assert_close(Cb.float().cpu(), fn(A.float() @ Wp[:, :K].float().t()), rtol=2e-2, atol=2e-2)


def test_gemm_prologue_stats_compact_and_rowmap():
    from spot_amd import lib
    g = torch.Generator().manual_seed(5)
    n_img, Hi, Wi, K, N, fold = 3, 6, 8, 64, 128, 16
    Ho, Wo = Hi // 2, Wi // 2
    X = _bf(n_img * Hi * Wi, K, g=g)
    G = _bf(n_img * Hi * Wi, fold, g=g)
    W = _bf(N, K, g=g, scale=K ** -0.5)
    scale, shift = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    gate = torch.rand(n_img, K, generator=g)
    # reference: rows of the stride-2 sampled image, first `fold` columns from G
    full = torch.cat([G.float(), X.float()[:, fold:]], 1).view(n_img, Hi, Wi, K)[:, ::2, ::2].reshape(-1, K)
    a = torch.relu(full * scale + shift)
    a = (a.view(n_img, Ho * Wo, K) * gate[:, None]).reshape(-1, K).to(torch.bfloat16).float()
    ref = a @ W.float().t()
    M = n_img * Ho * Wo
    C = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    s1 = torch.zeros(N, device="cuda"); s2 = torch.zeros(N, device="cuda")
    lib.gemm(dev(X), dev(W), C, M, N, K, A0=dev(G), lda0=fold, k0=fold, scale=dev(scale), shift=dev(shift),
             relu=True, gate=dev(gate), gate_group=Ho * Wo, rowmap=(2, Ho, Wo, Hi, Wi), stats=(s1, s2))
    torch.cuda.synchronize()
    torch.testing.assert_close(C.float().cpu(), ref, rtol=2e-2, atol=2e-2)
    cb = C.float().cpu()
    torch.testing.assert_close(s1.cpu(), cb.sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(s2.cpu(), (cb * cb).sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("N,K,mode", [(32, 32, "plain"), (64, 32, "rowmap"), (32, 64, "mask"), (64, 64, "gate"), (128, 64, "compact"),
                                      (128, 128, "gate"), (64, 128, "mask"), (320, 128, "plain"), (320, 320, "gate"), (128, 320, "mask"),
                                      (160, 96, "compact"), (768, 320, "gate"), (768, 320, "mask"), (256, 64, "plain")])
def test_gemm_streaming_path(N, K, mode):
    """Shapes that take the weights-resident streaming kernel (K <= 320, N in {32,64,128,160,320}, M >= 8192), with each
    fused option it serves: BN+ReLU(+SE gate) prologue and output statistics, compact first columns, stride-2 row map,
    ReLU mask + split output + bias."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(N * 7 + K)
    n_img, Ho, Wo = 3, 53, 52
    M = n_img * Ho * Wo                                   # 8268 rows, ragged last tile
    W = _bf(N, K, g=g, scale=K ** -0.5)
    Cout = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    kw, ref_a = {}, None
    if mode == "rowmap":
        Hi, Wi = 2 * Ho, 2 * Wo
        X = _bf(n_img * Hi * Wi, K, g=g)
        ref_a = X.float().view(n_img, Hi, Wi, K)[:, ::2, ::2].reshape(M, K)
        kw = dict(rowmap=(2, Ho, Wo, Hi, Wi))
    else:
        X = _bf(M, K, g=g)
        ref_a = X.float()
    if mode == "compact":
        fold = 16
        G = _bf(M, fold, g=g)
        ref_a = torch.cat([G.float(), ref_a[:, fold:]], 1)
        kw.update(A0=dev(G), lda0=fold, k0=fold)
    if mode in ("gate", "compact", "rowmap"):
        scale, shift = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
        ref_a = torch.relu(ref_a * scale + shift)
        kw.update(scale=dev(scale), shift=dev(shift), relu=True)
    if mode == "gate":
        gate = torch.rand(n_img, K, generator=g)
        ref_a = (ref_a.view(n_img, Ho * Wo, K) * gate[:, None]).reshape(M, K)
        kw.update(gate=dev(gate), gate_group=Ho * Wo)
    ref = ref_a.to(torch.bfloat16).float() @ W.float().t()
    s1 = torch.zeros(N, device="cuda"); s2 = torch.zeros(N, device="cuda")
    C0 = None
    if mode == "mask":
        bias = torch.randn(N, generator=g)
        mask = _bf(M, N, g=g)
        ref = (0.5 * ref + bias) * (mask.float() > 0)
        C0 = torch.full((M, 16), 5.0, dtype=torch.bfloat16, device="cuda")
        kw.update(bias=dev(bias), relu_mask=dev(mask), ldmask=N, C0=C0, ldc0=16, n_split=16, out_scale=0.5)
    else:
        kw.update(stats=(s1, s2))
    lib.gemm(dev(X), dev(W), Cout, M, N, K, **kw)
    torch.cuda.synchronize()
    got = Cout.float().cpu()
    if mode == "mask":
        assert bool((got[:, :16] == 7.0).all())            # the first columns went to the compact tensor
        got[:, :16] = C0.float().cpu()
    torch.testing.assert_close(got, ref, rtol=2e-2, atol=2e-2)
    if mode != "mask":
        torch.testing.assert_close(s1.cpu(), got.sum(0), rtol=1e-3, atol=5e-2)
        torch.testing.assert_close(s2.cpu(), (got * got).sum(0), rtol=1e-3, atol=5e-2)


@pytest.mark.parametrize("M,N1,N2", [(256, 64, 32), (1000, 128, 64), (4099, 320, 320), (2048, 768, 768), (500, 7, 1536), (3000, 64, 64)])
def test_wgrad(M, N1, N2):
    from spot_amd import lib
    g = torch.Generator().manual_seed(M + N1)
    ldp = (N1 + 7) // 8 * 8
    P = torch.zeros(M, ldp, dtype=torch.bfloat16)
    P[:, :N1] = _bf(M, N1, g=g)
    Q = _bf(M, N2, g=g)
    scale, shift = torch.rand(N2, generator=g) + 0.5, torch.randn(N2, generator=g) * 0.3
    q = torch.relu(Q.float() * scale + shift).to(torch.bfloat16).float()
    ref = P[:, :N1].float().t() @ q
    dW = torch.zeros(N1, N2, device="cuda"); db = torch.zeros(N1, device="cuda")
    lib.wgrad(dev(P), dev(Q), dW, M, N1, N2, ldp=ldp, scale=dev(scale), shift=dev(shift), relu=True, dbias=db)
    torch.cuda.synchronize()
    torch.testing.assert_close(dW.cpu(), ref, rtol=2e-3, atol=2e-3 * M ** 0.5)
    torch.testing.assert_close(db.cpu(), P[:, :N1].float().sum(0), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("slab", [False, True])
@pytest.mark.parametrize("mode", ["plain", "prologue", "gated", "compact", "rowmap"])
def test_wgrad_strip_320(mode, slab):
    """The 320 x 320 weight gradient of stage 3 (strip kernel: LDS-DMA ring, the Q prologue in place in LDS) with every operand
    option its callers use: BatchNorm + ReLU (+ squeeze-excite gate) on Q, the gated shift's compact source for the first
    columns with the channel interleave on the output, the strided row map of the shortcut convolution."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(len(mode))
    HW = 14 * 14
    N = 48
    M = N * HW                                                    # rows of P: 9,408 = 294 steps of 32
    P = _bf(M, 320, g=g)
    kw = {}
    if mode == "rowmap":                                          # Q rows live on a 28x28 map, P rows on its stride-2 subsample
        Hi = Wi = 28; Ho = Wo = 14
        Qfull = _bf(N * Hi * Wi, 320, g=g)
        q = Qfull.float().view(N, Hi, Wi, 320)[:, ::2, ::2].reshape(M, 320)
        Qd = dev(Qfull)
        kw["rowmap"] = (2, Ho, Wo, Hi, Wi)
    else:
        Q = _bf(M, 320, g=g)
        q = Q.float()
        Qd = dev(Q)
    if mode in ("prologue", "gated"):
        scale, shift = torch.rand(320, generator=g) + 0.5, torch.randn(320, generator=g) * 0.3
        q = torch.relu(q * scale + shift)
        kw.update(scale=dev(scale), shift=dev(shift), relu=True)
        if mode == "gated":
            gate = torch.rand(N, 320, generator=g)
            q = q * gate.repeat_interleave(HW, 0)
            kw.update(gate=dev(gate), gate_group=HW)
        q = q.to(torch.bfloat16).float()
    ref = P.float().t() @ q
    if mode == "compact":                                         # first 80 columns from a compact tensor, written through dest(c)
        fold = 80
        Q0 = _bf(M, fold, g=g)
        q2 = q.clone(); q2[:, :fold] = Q0.float()
        half, quarter = fold // 2, fold // 4
        dest = [(c // half) * half + 2 * ((c % half) % quarter) + (c % half) // quarter for c in range(fold)]
        full = P.float().t() @ q2
        ref = full.clone()
        ref[:, dest] = full[:, :fold]
        kw.update(Q0=dev(Q0), ldq0=fold, k0=fold, perm_fold=fold)
    if slab:                                                     # per-chunk slabs + a second launch instead of fp32 atomics
        kw["ws"] = torch.full((lib.lib().stes_wgrad_ws_floats(),), float("nan"), device="cuda")
    dW = torch.ones(320, 320, device="cuda")                      # the result is ADDED to dW
    lib.wgrad(dev(P), Qd, dW, M, 320, 320, **kw)
    torch.cuda.synchronize()
    torch.testing.assert_close(dW.cpu() - 1.0, ref, rtol=2e-3, atol=2e-3 * M ** 0.5)


@pytest.mark.parametrize("N,K", [(32, 64), (64, 128), (128, 128), (128, 320), (320, 320)])
def test_gemm_bn_backward_prologue_equals_apply_then_gemm(N, K):
    """The streaming GEMM's BatchNorm-backward prologue (operand = alpha*dY + beta*raw + delta, rounded to bf16, also stored
    for the weight gradient) against stes_bn_bwd_apply followed by the plain GEMM: same bits in both outputs, with the split
    output the gated-shift blocks use; shapes outside the streaming kernel are refused."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(N + 3 * K)
    M = 3 * 53 * 52
    D, raw, W = _bf(M, K, g=g), _bf(M, K, g=g), _bf(N, K, g=g, scale=K ** -0.5)
    coef = [dev(torch.randn(K, generator=g) * s) for s in (1.0, 0.3, 0.1)]
    dD, dR, dW_ = dev(D), dev(raw), dev(W)
    dc_ref = torch.full((M, K), 3.0, dtype=torch.bfloat16, device="cuda")
    lib.bn_bwd_apply(dD, dR, M, K, coef, dc_ref)
    split = 16 if N >= 64 else 0
    mk = lambda: (torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda"),
                  torch.full((M, 16), 5.0, dtype=torch.bfloat16, device="cuda") if split else None)
    C_ref, C0_ref = mk()
    lib.gemm(dc_ref, dW_, C_ref, M, N, K, C0=C0_ref, ldc0=16, n_split=split)
    C_got, C0_got = mk()
    dc_got = torch.full((M, K), 4.0, dtype=torch.bfloat16, device="cuda")
    lib.gemm(dD, dW_, C_got, M, N, K, C0=C0_got, ldc0=16, n_split=split, bn_bwd=(dR, coef, dc_got))
    torch.cuda.synchronize()
    assert torch.equal(dc_got.view(torch.int16), dc_ref.view(torch.int16))
    assert torch.equal(C_got[:, split:].view(torch.int16), C_ref[:, split:].view(torch.int16))
    if split:
        assert torch.equal(C0_got.view(torch.int16), C0_ref.view(torch.int16))
    # without the side output
    C2, _ = mk()
    lib.gemm(dD, dW_, C2, M, N, K, bn_bwd=(dR, coef, None))
    torch.cuda.synchronize()
    assert torch.equal(C2[:, split:].view(torch.int16), C_ref[:, split:].view(torch.int16))
    assert lib.gemm_streams(M, N, K)
    with pytest.raises(lib.StesError):
        lib.gemm(dD, dW_, C2, 4096, N, K, bn_bwd=(dR, coef, None))          # too few rows for the streaming kernel


@pytest.mark.parametrize("M,k0", [(3 * 53 * 52, 8), (8192 + 19, 0), (16 * 1024, 16)])
def test_gemm_with_fused_weight_gradient_equals_gemm_then_wgrad(M, k0):
    """K = 64, N = 32 (stage 1's first 1x1 conv): the input-gradient GEMM with the BatchNorm-backward prologue also produces the
    layer's weight gradient.  Against the two-launch form (prologue GEMM writing the operand, stes_wgrad reading it): the GEMM
    output bit for bit, the weight gradient to fp32 summation order -- with the gated-shift operand split and column
    permutation, and a row count that is not a multiple of the 32-row tile."""
    from spot_amd import lib
    N, K = 32, 64
    g = torch.Generator().manual_seed(M + k0)
    D, raw, W, Q = _bf(M, K, g=g), _bf(M, K, g=g), _bf(N, K, g=g, scale=K ** -0.5), _bf(M, N, g=g)
    Q0 = _bf(M, max(k0, 8), g=g)
    coef = [dev(torch.randn(K, generator=g) * s) for s in (1.0, 0.3, 0.1)]
    dD, dR, dW_, dQ, dQ0 = dev(D), dev(raw), dev(W), dev(Q), dev(Q0)
    assert lib.gemm_fuses_wgrad(M, N, K)
    C_ref = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
    dc = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
    lib.gemm(dD, dW_, C_ref, M, N, K, bn_bwd=(dR, coef, dc))
    gw_ref = torch.ones(K, N, device="cuda")
    lib.wgrad(dc, dQ, gw_ref, M, K, N, Q0=dQ0 if k0 else None, ldq0=Q0.shape[1], k0=k0, perm_fold=k0)
    C_got = torch.full((M, N), 9.0, dtype=torch.bfloat16, device="cuda")
    gw_got = torch.ones(K, N, device="cuda")
    lib.gemm(dD, dW_, C_got, M, N, K, bn_bwd=(dR, coef, None), wgrad_fused=(dQ, gw_got, dQ0 if k0 else None, Q0.shape[1], k0, k0))
    torch.cuda.synchronize()
    assert torch.equal(C_got.view(torch.int16), C_ref.view(torch.int16))
    torch.testing.assert_close(gw_got, gw_ref, rtol=1e-4, atol=1e-3 * M ** 0.5 * 1e-1)
    # and against the definition
    qeff = Q.float().clone()
    if k0:
        qeff[:, :k0] = Q0.float()[:, :k0]
    want = dc.float().cpu().t() @ qeff
    if k0:
        half, quarter = k0 // 2, k0 // 4
        dest = [(c // half) * half + 2 * ((c % half) % quarter) + (c % half) // quarter for c in range(k0)]
        perm = want.clone()
        perm[:, dest] = want[:, :k0]
        want = perm
    torch.testing.assert_close(gw_got.cpu() - 1.0, want, rtol=2e-3, atol=2e-3 * M ** 0.5)
    with pytest.raises(lib.StesError):            # another shape: refused, not silently dropped
        lib.gemm(dev(_bf(M, 128, g=g)), dev(_bf(64, 128, g=g)), torch.empty(M, 64, dtype=torch.bfloat16, device="cuda"), M, 64, 128,
                 bn_bwd=(dev(_bf(M, 128, g=g)), [dev(torch.randn(128)) for _ in range(3)], None),
                 wgrad_fused=(dev(_bf(M, 64, g=g)), torch.zeros(128, 64, device="cuda"), None, 0, 0, 0))


@pytest.mark.parametrize("n,Ho,Wo", [(5, 41, 43), (3, 56, 56)])
def test_gemm_with_fused_weight_gradient_of_a_strided_shortcut(n, Ho, Wo):
    """The same for a stride-2 1x1 shortcut conv: the input gradient lives at the output resolution (M = n*Ho*Wo rows) and the
    weight gradient's Q rows are the input pixels (n, 2ho, 2wo) of the [n, Hi, Wi] block input."""
    from spot_amd import lib
    N, K, Hi, Wi = 32, 64, 2 * Ho + 1, 2 * Wo
    M = n * Ho * Wo
    g = torch.Generator().manual_seed(M)
    D, raw, W, Q = _bf(M, K, g=g), _bf(M, K, g=g), _bf(N, K, g=g, scale=K ** -0.5), _bf(n * Hi * Wi, N, g=g)
    coef = [dev(torch.randn(K, generator=g) * s) for s in (1.0, 0.3, 0.1)]
    dD, dR, dW_, dQ = dev(D), dev(raw), dev(W), dev(Q)
    rowmap = (2, Ho, Wo, Hi, Wi)
    C_ref = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    dc = torch.empty(M, K, dtype=torch.bfloat16, device="cuda")
    lib.gemm(dD, dW_, C_ref, M, N, K, bn_bwd=(dR, coef, dc))
    gw_ref = torch.zeros(K, N, device="cuda")
    lib.wgrad(dc, dQ, gw_ref, M, K, N, rowmap=rowmap)
    C_got = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    gw_got = torch.zeros(K, N, device="cuda")
    lib.gemm(dD, dW_, C_got, M, N, K, bn_bwd=(dR, coef, None), wgrad_fused=(dQ, gw_got, None, 0, 0, 0, rowmap))
    torch.cuda.synchronize()
    assert torch.equal(C_got.view(torch.int16), C_ref.view(torch.int16))
    torch.testing.assert_close(gw_got, gw_ref, rtol=1e-4, atol=1e-4 * M ** 0.5)
    qs = Q.float().view(n, Hi, Wi, N)[:, 0:2 * Ho:2, 0:2 * Wo:2].reshape(M, N)
    torch.testing.assert_close(gw_got.cpu(), dc.float().cpu().t() @ qs, rtol=2e-3, atol=2e-3 * M ** 0.5)


@pytest.mark.parametrize("N1,N2", [(32, 32), (64, 24), (32, 128)])
def test_wgrad_bn_backward_prologue_equals_apply_then_wgrad(N1, N2):
    """stes_wgrad with the BatchNorm-backward prologue on P (the stem's path: no input gradient, so no GEMM to do it)
    against stes_bn_bwd_apply followed by the plain weight gradient, including the bias gradient."""
    from spot_amd import lib
    g = torch.Generator().manual_seed(N1 * 5 + N2)
    M = 9001
    D, raw, Q = _bf(M, N1, g=g), _bf(M, N1, g=g), _bf(M, N2 + (-N2) % 8, g=g)
    coef = [dev(torch.randn(N1, generator=g) * s) for s in (1.0, 0.3, 0.1)]
    dD, dR, dQ = dev(D), dev(raw), dev(Q)
    dc = torch.empty(M, N1, dtype=torch.bfloat16, device="cuda")
    lib.bn_bwd_apply(dD, dR, M, N1, coef, dc)
    ref, got = torch.zeros(N1, N2, device="cuda"), torch.zeros(N1, N2, device="cuda")
    bref, bgot = torch.zeros(N1, device="cuda"), torch.zeros(N1, device="cuda")
    ldq = Q.shape[1]
    lib.wgrad(dc, dQ, ref, M, N1, N2, ldq=ldq, dbias=bref)
    lib.wgrad(dD, dQ, got, M, N1, N2, ldq=ldq, dbias=bgot, bn_bwd=(dR, coef))
    torch.cuda.synchronize()
    scale = float(ref.abs().max()) + 1e-6
    assert float((got - ref).abs().max()) <= 2e-3 * scale          # same operands; only the order of the fp32 atomics differs
    assert float((bgot - bref).abs().max()) <= 2e-3 * (float(bref.abs().max()) + 1e-6)
    if N2 > 64:
        with pytest.raises(lib.StesError):
            lib.wgrad(dD, dQ, got, M, 128, N2, ldq=ldq, bn_bwd=(dR, coef))


@pytest.mark.parametrize("N,K,down,fold", [(64, 64, True, 16), (128, 128, False, 32), (320, 320, False, 80), (320, 320, True, 0)])
def test_gemm_block_tail_epilogue_equals_gemm_then_bn_act_residual(N, K, down, fold):
    """Evaluation forward of a bottleneck's last 1x1 conv: relu(bn3(conv3) + shortcut) from the GEMM's store phase, with the
    next block's compact gated-shift slice, against the two-launch form (GEMM writing the raw product, stes_bn_act_residual
    reading it) bit for bit -- identity and BatchNorm'ed shortcut, ragged row count."""
    from spot_amd import lib
    n_img, HW = 3, 53 * 52
    M = n_img * HW
    g = torch.Generator().manual_seed(N + K + fold)
    X, W, R = _bf(M, K, g=g), _bf(N, K, g=g, scale=K ** -0.5), _bf(M, N, g=g)
    scale, shift = dev(torch.rand(K, generator=g) + 0.5), dev(torch.randn(K, generator=g) * 0.3)
    gate = dev(torch.rand(n_img, K, generator=g))
    es, eb = dev(torch.rand(N, generator=g) + 0.5), dev(torch.randn(N, generator=g) * 0.3)
    rs, rb = (dev(torch.rand(N, generator=g) + 0.5), dev(torch.randn(N, generator=g) * 0.3)) if down else (None, None)
    dX, dW_, dR = dev(X), dev(W), dev(R)
    kw = dict(scale=scale, shift=shift, relu=True, gate=gate, gate_group=HW)
    assert lib.gemm_has_tail(M, N, K)
    c3 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    lib.gemm(dX, dW_, c3, M, N, K, **kw)
    out_ref = torch.full((M, N), 3.0, dtype=torch.bfloat16, device="cuda")
    slc_ref = torch.full((M, max(fold, 8)), 4.0, dtype=torch.bfloat16, device="cuda")
    lib.call("stes_bn_act_residual", lib.ptr(c3), lib.ptr(es), lib.ptr(eb), lib.ptr(dR), lib.ptr(rs), lib.ptr(rb), lib.ptr(out_ref), None, None,
             fold, lib.ptr(slc_ref) if fold else None, M, N, None)
    out = torch.full((M, N), 5.0, dtype=torch.bfloat16, device="cuda")
    slc = torch.full((M, max(fold, 8)), 6.0, dtype=torch.bfloat16, device="cuda")
    lib.TIMING = []
    try:
        lib.gemm(dX, dW_, out, M, N, K, tail=(es, eb, dR, rs, rb, slc if fold else None, fold), **kw)
        assert lib.TIMING[0][0].startswith("gemm_stream_tail_kernel<")
    finally:
        lib.TIMING = None
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), out_ref.view(torch.int16))
    if fold:
        assert torch.equal(slc.view(torch.int16), slc_ref.view(torch.int16))
    with pytest.raises(lib.StesError):            # a shape outside the three: refused
        lib.gemm(dX, dW_, out, M, N, K, tail=(es, eb, dR, rs, rb, None, 0), stats=(torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")), **kw)


def test_gemm_stem_tail_epilogue_equals_gemm_then_bn_act():
    """The stem in evaluation: relu(bn(stem conv)) and the first block's 8-channel gated-shift slice from the stem GEMM's store
    phase (no prologue, no residual), bit for bit against GEMM + stes_bn_act_residual."""
    from spot_amd import lib
    M, N, K, fold = 3 * 57 * 55, 32, 32, 8
    g = torch.Generator().manual_seed(11)
    X, W = dev(_bf(M, K, g=g)), dev(_bf(N, K, g=g, scale=K ** -0.5))
    es, eb = dev(torch.rand(N, generator=g) + 0.5), dev(torch.randn(N, generator=g) * 0.3)
    c0 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    lib.gemm(X, W, c0, M, N, K)
    out_ref = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    slc_ref = torch.empty(M, fold, dtype=torch.bfloat16, device="cuda")
    lib.call("stes_bn_act_residual", lib.ptr(c0), lib.ptr(es), lib.ptr(eb), None, None, None, lib.ptr(out_ref), None, None, fold, lib.ptr(slc_ref),
             M, N, None)
    out = torch.full((M, N), 5.0, dtype=torch.bfloat16, device="cuda")
    slc = torch.full((M, fold), 6.0, dtype=torch.bfloat16, device="cuda")
    lib.gemm(X, W, out, M, N, K, tail=(es, eb, None, None, None, slc, fold))
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), out_ref.view(torch.int16)) and torch.equal(slc.view(torch.int16), slc_ref.view(torch.int16))


@pytest.mark.parametrize("M,K1,N1,N2,f32", [(1024, 320, 80, 320, True), (1024, 64, 8, 64, True), (50, 768, 192, 768, False), (17, 128, 32, 128, True),
                                           (1024, 768, 192, 768, True), (3, 8, 8, 8, False), (100, 1024, 256, 1024, True)])
def test_gemm_small2_equals_two_small_products(M, K1, N1, N2, f32):
    """The squeeze-excite FC pair in one launch (H = relu(A W1^T + b1), C = sigmoid(H W2^T + b2)) and its backward pair (masked
    H, scaled fp32 C) against two stes_gemm calls: same roundings (H bf16 in between), fp32 summation order aside."""
    from spot_amd import lib
    BF = torch.bfloat16

    def close(a, b, rtol, atol):
        assert torch.allclose(a, b, rtol=rtol, atol=atol), float((a - b).abs().max())
    g = torch.Generator().manual_seed(M + K1 + N1 + N2)
    A = (torch.randn(M, K1, generator=g)).to(BF).cuda()
    W1 = (torch.randn(N1, K1, generator=g) / K1 ** 0.5).to(BF).cuda(); W2 = (torch.randn(N2, N1, generator=g) / N1 ** 0.5).to(BF).cuda()
    b1 = torch.randn(N1, generator=g).cuda() * 0.1; b2 = torch.randn(N2, generator=g).cuda() * 0.1
    mask = torch.randn(M, N1, generator=g).to(BF).cuda()
    for fwd in (True, False):
        H1 = torch.full((M, N1), 7.0, dtype=BF, device="cuda"); H2 = torch.full_like(H1, 9.0)
        dt = torch.float32 if f32 else BF
        C1 = torch.full((M, N2), 7.0, dtype=dt, device="cuda"); C2 = torch.full_like(C1, 9.0)
        if fwd:
            lib.gemm(A, W1, H1, M, N1, K1, bias=b1, act=1)
            lib.gemm(H1, W2, C1, M, N2, N1, bias=b2, act=2)
            lib.gemm2(A, W1, H2, W2, C2, M, K1, N1, N2, bias1=b1, act1=1, bias2=b2, act2=2)
        else:
            lib.gemm(A, W1, H1, M, N1, K1, relu_mask=mask, ldmask=N1)
            lib.gemm(H1, W2, C1, M, N2, N1, out_scale=0.25)
            lib.gemm2(A, W1, H2, W2, C2, M, K1, N1, N2, relu_mask1=mask, ldmask1=N1, out_scale=0.25)
        torch.cuda.synchronize()
        close(H2.float(), H1.float(), 1e-2, 1e-2)
        close(C2.float(), C1.float(), 2e-2, 1e-2)
        if not fwd:
            assert bool(((H2.float() == 0) | (mask.float() > 0)).all())
