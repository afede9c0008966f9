// Grouped 3x3 convolution (group width 16) of the RegNet-Y bottleneck as implicit GEMM on
// v_mfma_f32_16x16x32_bf16, plus the stem's im2col.
//
// Per group the GEMM is [16 out ch] x [K = 9 taps * 16 in ch] x [pixels].  One MFMA K-step (32)
// covers two taps; lane (p = lane&15, q = lane>>4) supplies, for pixel p, the 8 channels
// (q&1)*8.. of tap 2*step + (q>>1): one 16-byte NHWC load.  Weights sit pre-packed in fragment
// order in registers for the wave's lifetime; the output leaves as 4 consecutive channels per lane.
//   forward : BatchNorm-apply + ReLU of the previous layer fused into the operand load, batch
//             statistics of the output reduced in registers (one atomic per channel per wave);
//   dgrad   : same kernel, transposed/stride-aware tap geometry;
//   wgrad   : reduction over pixels -> both operands transposed through LDS with ds_read_b64_tr_b16.
#include "stes_common.h"

namespace {

// ------------------------------------------------------------------ weight packing
// w [G*16][16][3][3] fp32 ->  fwd [G][5][64][8]: lane(i=co,q): tap=2s+(q>>1), ci=(q&1)*8+j
//                             dgr [G][5][64][8]: lane(i=ci,q): tap=2s+(q>>1), co=(q&1)*8+j
__global__ void pack_conv3x3g_kernel(const float* __restrict__ w, bf16* __restrict__ fwd, bf16* __restrict__ dgr, int G) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = G * 5 * 64 * 8;
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, s = (idx >> 9) % 5, g = idx / (5 * 512);
    const int i = lane & 15, q = lane >> 4;
    const int tap = 2 * s + (q >> 1), c = (q & 1) * 8 + j;
    float vf = 0.f, vd = 0.f;
    if (tap < 9) {
        vf = w[((long)(g * 16 + i) * 16 + c) * 9 + tap];
        vd = w[((long)(g * 16 + c) * 16 + i) * 9 + tap];
    }
    fwd[idx] = (bf16)vf;
    dgr[idx] = (bf16)vd;
}

struct ConvGeom {
    int N, Hs, Ws, Hd, Wd, stride, C, G;
    uint32_t Md;
    FastDiv fd_wd, fd_hd;
};

template <int MODE, int STRIDE>   // MODE 0 forward (dst = conv output), 1 dgrad (dst = conv input)
__global__ __launch_bounds__(256) void conv3x3g_kernel(const bf16* __restrict__ src, const bf16* __restrict__ wp,
                                                       bf16* __restrict__ dst, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* col_sum, float* col_sumsq,
                                                       ConvGeom gm, int n_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float red[2][64];
    const bool active = blockIdx.y * 4 + wave < gm.G;
    const int g = active ? blockIdx.y * 4 + wave : gm.G - 1;      // idle waves shadow the last group and store nothing
    const int p = lane & 15, q = lane >> 4;
    const int C = gm.C;
    bf16x8 wf[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) wf[s] = *reinterpret_cast<const bf16x8*>(wp + (((long)g * 5 + s) * 64 + lane) * 8);
    const int cin0 = g * 16 + (q & 1) * 8;
    float sc[8], sh[8];
    const bool affine = (MODE == 0) && scale != nullptr;
    if (affine) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc[i] = scale[cin0 + i]; sh[i] = shift[cin0 + i]; }
    }
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    const bool stats = (MODE == 0) && col_sum != nullptr;
    const int Hs = gm.Hs, Ws = gm.Ws;

    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint32_t m = (uint32_t)t * 16 + p;
        const bool valid = active && m < gm.Md;
        uint32_t row, wd_, n_, hd_;
        fdivmod(valid ? m : 0u, gm.fd_wd, row, wd_);
        fdivmod(row, gm.fd_hd, n_, hd_);
        const int hd = (int)hd_, wd = (int)wd_;
        const long nbase = (long)n_ * Hs * Ws;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int tap = 2 * s + (q >> 1);
            bf16x8 frag = zero8();
            if (valid && tap < 9) {
                const int kh = tap / 3, kw = tap - kh * 3;
                int hs, ws; bool ok;
                if (MODE == 0) {
                    hs = hd * STRIDE + kh - 1; ws = wd * STRIDE + kw - 1;
                    ok = hs >= 0 && hs < Hs && ws >= 0 && ws < Ws;
                } else {
                    const int th = hd + 1 - kh, tw = wd + 1 - kw;
                    ok = th >= 0 && tw >= 0 && (STRIDE == 1 || (((th | tw) & 1) == 0));
                    hs = STRIDE == 1 ? th : th >> 1; ws = STRIDE == 1 ? tw : tw >> 1;
                    ok = ok && hs < Hs && ws < Ws;
                }
                if (ok) {
                    frag = *reinterpret_cast<const bf16x8*>(src + (nbase + (long)hs * Ws + ws) * C + cin0);
                    if (affine) {
                        float f[8];
                        unpack8(frag, f);
#pragma unroll
                        for (int i = 0; i < 8; ++i) f[i] = fmaxf(f[i] * sc[i] + sh[i], 0.f);
                        frag = pack8(f);
                    }
                }
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s], frag, acc, 0, 0, 0);
        }
        if (valid) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[r];
            *reinterpret_cast<bf16x4*>(dst + (long)m * C + g * 16 + q * 4) = o;
            if (stats) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float v = (float)o[r]; ssum[r] += v; ssq[r] += v * v; }
            }
        }
    }
    if (stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = ssum[r], b = ssq[r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
            if (p == 0) { red[0][wave * 16 + q * 4 + r] = a; red[1][wave * 16 + q * 4 + r] = b; }
        }
        __syncthreads();
        // one contiguous 64-channel atomic per workgroup and statistic instead of 32 four-lane ones
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6, c = blockIdx.y * 64 + (threadIdx.x & 63);
            if (c < gm.C) atomicAdd((which ? col_sumsq : col_sum) + c, red[which][threadIdx.x & 63]);
        }
    }
}

// ------------------------------------------------------------------ wgrad
constexpr int TH = 4, TW = 8;                 // 32 output pixels per reduction step
constexpr int MAX_HALO = 9 * 17;              // stride 2: (TH-1)*2+3 by (TW-1)*2+3

__device__ __forceinline__ int rho32(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

__global__ __launch_bounds__(256) void conv3x3g_wgrad_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ in,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             float* __restrict__ dw, int N, int Hi, int Wi, int Ho, int Wo,
                                                             int stride, int C, int n_items, FastDiv fd_tiles, FastDiv fd_tw,
                                                             FastDiv fd_iw) {
    // one LDS block: staging tiles while walking items, then re-used to lay the result out for coalesced atomics
    __shared__ __attribute__((aligned(16))) unsigned char smem[9216 * 4];
    bf16* Pt = reinterpret_cast<bf16*>(smem);                     // [4 groups][32 px (rho)][16]
    bf16* It = reinterpret_cast<bf16*>(smem + 4 * 32 * 16 * 2);    // [4 groups][MAX_HALO][16]
    float* red = reinterpret_cast<float*>(smem);                  // [4 groups][16 co][16 ci][9]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int gq = blockIdx.y;
    const int IH = (TH - 1) * stride + 3, IW = (TW - 1) * stride + 3;
    const int halo = IH * IW;
    const int chunk = tid & 7, cgrp = chunk >> 1, chalf = chunk & 1;
    const int cbase = gq * 64 + chunk * 8;
    float sc[8], sh[8];
    const bool affine = scale != nullptr;
    if (affine) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc[i] = scale[cbase + i]; sh[i] = shift[cbase + i]; }
    }
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int px_lo = 8 * fq + (fr >> 2), px_hi = px_lo + 4;
    const int cc = 4 * (fr & 3);
    const int a_lo = (wave * 32 + rho32(px_lo)) * 16 + cc, a_hi = (wave * 32 + rho32(px_hi)) * 16 + cc;
    const int hb_lo = ((px_lo >> 3) * stride) * IW + (px_lo & 7) * stride;
    const int hb_hi = ((px_hi >> 3) * stride) * IW + (px_hi & 7) * stride;

    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        uint32_t n, tr, trow, tcol;
        fdivmod((uint32_t)item, fd_tiles, n, tr);
        fdivmod(tr, fd_tw, trow, tcol);
        const int ho0 = (int)trow * TH, wo0 = (int)tcol * TW;
        {   // dOut tile: thread -> pixel tid/8, chunk tid%8
            const int px = tid >> 3;
            const int ho = ho0 + (px >> 3), wo = wo0 + (px & 7);
            bf16x8 v = zero8();
            if (ho < Ho && wo < Wo) v = *reinterpret_cast<const bf16x8*>(dout + (((long)n * Ho + ho) * Wo + wo) * C + cbase);
            *reinterpret_cast<bf16x8*>(&Pt[(cgrp * 32 + rho32(px)) * 16 + chalf * 8]) = v;
        }
        for (int i = tid; i < halo * 8; i += 256) {
            const uint32_t hp = (uint32_t)i >> 3;
            uint32_t hr, hc;
            fdivmod(hp, fd_iw, hr, hc);
            const int hi = ho0 * stride - 1 + (int)hr, wi = wo0 * stride - 1 + (int)hc;
            bf16x8 v = zero8();
            if (hi >= 0 && hi < Hi && wi >= 0 && wi < Wi) {
                v = *reinterpret_cast<const bf16x8*>(in + (((long)n * Hi + hi) * Wi + wi) * C + cbase);
                if (affine) {
                    float f[8];
                    unpack8(v, f);
#pragma unroll
                    for (int k = 0; k < 8; ++k) f[k] = fmaxf(f[k] * sc[k] + sh[k], 0.f);
                    v = pack8(f);
                }
            }
            *reinterpret_cast<bf16x8*>(&It[(cgrp * MAX_HALO + hp) * 16 + chalf * 8]) = v;
        }
        __syncthreads();
        const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(&Pt[a_lo]));
        const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(&Pt[a_hi]));
        const bf16x8 fa = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int off = (t / 3) * IW + (t % 3);
            const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (bf16x4 __attribute__((address_space(3)))*)(&It[(wave * MAX_HALO + hb_lo + off) * 16 + cc]));
            const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (bf16x4 __attribute__((address_space(3)))*)(&It[(wave * MAX_HALO + hb_hi + off) * 16 + cc]));
            const bf16x8 fb = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
    // result -> LDS in the checkpoint layout [co][ci][tap] of the 64 channels of this workgroup, then contiguous atomics
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * 16 + fq * 4 + r) * 16 + fr) * 9 + t] = acc[t][r];
    __syncthreads();
    float* dst = dw + (long)gq * 9216;
    for (int i = tid; i < 9216; i += 256) atomicAdd(dst + i, red[i]);
}

// ------------------------------------------------------------------ stem im2col
// x [N,3,H,W] fp32 (NCHW) -> patches [N*Ho*Wo][32] bf16, k = ci*9 + kh*3 + kw (27 used, rest zero); 3x3 stride 2 pad 1
__global__ __launch_bounds__(256) void im2col_stem_kernel(const float* __restrict__ x, bf16* __restrict__ out, int N, int Cin,
                                                          int H, int W, int Ho, int Wo) {
    const long total = (long)N * Ho * Wo;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (long)gridDim.x * blockDim.x) {
        const int wo = (int)(m % Wo); const long t = m / Wo; const int ho = (int)(t % Ho); const long n = t / Ho;
        float f[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) f[k] = 0.f;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            if (ci >= Cin) break;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int hi = ho * 2 + kh - 1;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int wi = wo * 2 + kw - 1;
                    if (hi >= 0 && hi < H && wi >= 0 && wi < W) f[ci * 9 + kh * 3 + kw] = x[((n * Cin + ci) * H + hi) * (long)W + wi];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (bf16)f[c * 8 + i];
            *reinterpret_cast<bf16x8*>(out + m * 32 + c * 8) = v;
        }
    }
}

}  // namespace

#define S(s) ((hipStream_t)(s))

extern "C" int stes_pack_conv3x3g(const float* w, uint16_t* wp_fwd, uint16_t* wp_dgrad, int G, void* stream) {
    STES_CHECK(w && wp_fwd && wp_dgrad && G > 0, STES_ERR_ARG, "stes_pack_conv3x3g: null pointer");
    const int total = G * 5 * 64 * 8;
    hipLaunchKernelGGL(pack_conv3x3g_kernel, dim3(stes_div_up(total, 256)), dim3(256), 0, S(stream), w, (bf16*)wp_fwd, (bf16*)wp_dgrad, G);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

static int conv_launch(int mode, const uint16_t* src, const uint16_t* wp, uint16_t* dst, const float* scale, const float* shift,
                       float* col_sum, float* col_sumsq, int N, int Hi, int Wi, int Ho, int Wo, int stride, int C, void* stream) {
    STES_CHECK(src && wp && dst, STES_ERR_ARG, "stes_conv3x3g: null pointer");
    STES_CHECK(C % 64 == 0, STES_ERR_ARG, "stes_conv3x3g: channels must be a multiple of 64 (groups of 16, 4 per workgroup)");
    STES_CHECK(stride == 1 || stride == 2, STES_ERR_ARG, "stes_conv3x3g: stride must be 1 or 2");
    STES_CHECK(Ho == (Hi + 2 - 3) / stride + 1 && Wo == (Wi + 2 - 3) / stride + 1, STES_ERR_ARG, "stes_conv3x3g: output size does not match");
    STES_CHECK((scale == nullptr) == (shift == nullptr) && (col_sum == nullptr) == (col_sumsq == nullptr), STES_ERR_ARG, "stes_conv3x3g: paired pointers");
    ConvGeom gm;
    gm.N = N; gm.stride = stride; gm.C = C; gm.G = C / 16;
    if (mode == 0) { gm.Hs = Hi; gm.Ws = Wi; gm.Hd = Ho; gm.Wd = Wo; }
    else { gm.Hs = Ho; gm.Ws = Wo; gm.Hd = Hi; gm.Wd = Wi; }
    const long md = (long)N * gm.Hd * gm.Wd;
    STES_CHECK(md <= STES_MAX_ROWS && (long)N * gm.Hs * gm.Ws <= STES_MAX_ROWS, STES_ERR_UNSUPPORTED, "stes_conv3x3g: more than 2^31 pixels");
    gm.Md = (uint32_t)md;
    gm.fd_wd = make_fastdiv(gm.Wd); gm.fd_hd = make_fastdiv(gm.Hd);
    const int n_tiles = stes_div_up(md, 16);
    const int gy = gm.G / 4;
    int gx = n_tiles;
    const int cap = 4096 / gy > 1 ? 4096 / gy : 1;
    if (gx > cap) gx = cap;
#define CONV_LAUNCH(MODE, ST) hipLaunchKernelGGL((conv3x3g_kernel<MODE, ST>), dim3(gx, gy), dim3(256), 0, S(stream), (const bf16*)src, \
        (const bf16*)wp, (bf16*)dst, scale, shift, col_sum, col_sumsq, gm, n_tiles)
    if (mode == 0) { if (stride == 1) CONV_LAUNCH(0, 1); else CONV_LAUNCH(0, 2); }
    else { if (stride == 1) CONV_LAUNCH(1, 1); else CONV_LAUNCH(1, 2); }
#undef CONV_LAUNCH
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_conv3x3g_fwd(const uint16_t* in, const uint16_t* wp_fwd, uint16_t* out, const float* in_scale,
                                 const float* in_shift, float* col_sum, float* col_sumsq, int N, int Hi, int Wi, int Ho, int Wo,
                                 int stride, int C, void* stream) {
    return conv_launch(0, in, wp_fwd, out, in_scale, in_shift, col_sum, col_sumsq, N, Hi, Wi, Ho, Wo, stride, C, stream);
}

extern "C" int stes_conv3x3g_dgrad(const uint16_t* dout, const uint16_t* wp_dgrad, uint16_t* din, int N, int Hi, int Wi, int Ho,
                                   int Wo, int stride, int C, void* stream) {
    return conv_launch(1, dout, wp_dgrad, din, nullptr, nullptr, nullptr, nullptr, N, Hi, Wi, Ho, Wo, stride, C, stream);
}

extern "C" int stes_conv3x3g_wgrad(const uint16_t* dout, const uint16_t* in, const float* in_scale, const float* in_shift, float* dw,
                                   int N, int Hi, int Wi, int Ho, int Wo, int stride, int C, void* stream) {
    STES_CHECK(dout && in && dw, STES_ERR_ARG, "stes_conv3x3g_wgrad: null pointer");
    STES_CHECK(C % 64 == 0 && (stride == 1 || stride == 2), STES_ERR_ARG, "stes_conv3x3g_wgrad: C % 64, stride 1|2");
    STES_CHECK(Ho == (Hi + 2 - 3) / stride + 1 && Wo == (Wi + 2 - 3) / stride + 1, STES_ERR_ARG, "stes_conv3x3g_wgrad: output size does not match");
    const int th = stes_div_up(Ho, TH), tw = stes_div_up(Wo, TW);
    const int n_items = N * th * tw;
    const int gy = C / 64;
    int gx = n_items;
    const int cap = 2048 / gy > 1 ? 2048 / gy : 1;
    if (gx > cap) gx = cap;
    const int IW = (TW - 1) * stride + 3;
    hipLaunchKernelGGL(conv3x3g_wgrad_kernel, dim3(gx, gy), dim3(256), 0, S(stream), (const bf16*)dout, (const bf16*)in, in_scale, in_shift,
                       dw, N, Hi, Wi, Ho, Wo, stride, C, n_items, make_fastdiv(th * tw), make_fastdiv(tw), make_fastdiv(IW));
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_im2col_stem(const float* x, uint16_t* patches, int N, int Cin, int H, int W, void* stream) {
    STES_CHECK(x && patches && N > 0 && Cin >= 1 && Cin <= 3, STES_ERR_ARG, "stes_im2col_stem: bad argument");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    long blocks = ((long)N * Ho * Wo + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(im2col_stem_kernel, dim3((int)blocks), dim3(256), 0, S(stream), x, (bf16*)patches, N, Cin, H, W, Ho, Wo);
    STES_LAUNCH_CHECK();
    return STES_OK;
}
