// Clip augmentation on the GPU: one geometric resampling through a per-clip homography plus the colour chain of the training
// loader (reference dataset/frame.py:78-140: RandomChoice of crops, RandomChoice of flip / zoom-out / perspective / rotation /
// affine, Resize, then hue, saturation, brightness, contrast, Gaussian blur, Normalize; same parameters for every frame of a
// clip, :169-170).  Decoded uint8 frames in, the loader's normalised fp32 tensor (or uint8) out.
//
// The work is byte traffic: every output pixel gathers at most four source pixels per channel.  Two launches, because contrast
// needs the mean grey value of the whole frame after the brightness step:
//   1. augment_warp_kernel   gather + hue + saturation + brightness -> fp32 planes, per-workgroup partial sums of the grey value
//   2. augment_finish_kernel frame mean (partials added in a fixed order: results do not depend on scheduling), contrast,
//                            5x5 Gaussian blur with reflected borders, Normalize -> output
// Clips with neither contrast nor blur (9 in 16 at the loader's p = 0.25) are finished by the first launch.
// Threads run along x so that stores are whole rows; the gathers of neighbouring threads fall into neighbouring source bytes
// for every map that is close to a similarity, and the source frame (267 KB at 398x224) stays in L2 while it is being read.
#include "stes_common.h"

// every product and sum below is rounded on its own, so that a plain fp32 restatement gives the same bits
#pragma clang fp contract(off)

namespace {

constexpr int AUG_THREADS = 256;
constexpr int AUG_VB = 8;          // 256-pixel pieces of a frame per workgroup (a workgroup per piece is bound by dispatch)
constexpr int AUG_PARAMS = STES_AUGMENT_PARAMS;

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float grey_of(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }
// ratio*a + (1-ratio)*b clamped to [0,1]: the blend every ColorJitter step is made of
__device__ __forceinline__ float blend(float a, float b, float ratio) { return clamp01(ratio * a + (1.f - ratio) * b); }

// the byte as a float (0 outside the frame); the division by 255 follows the interpolation: one per channel instead of four
__device__ __forceinline__ float fetch(const uint8_t* __restrict__ plane, int x, int y, int W, int H) {
    return (x >= 0 && x < W && y >= 0 && y < H) ? (float)plane[(long)y * W + x] : 0.f;
}

// The horizontal neighbours x0, x0 + 1 of row y in one two-byte load (any byte address): the gathers are bound by load
// instructions per lane, not by bytes, and this halves them.  Pairs that straddle the frame's edge take single bytes.
typedef uint16_t u16_any __attribute__((aligned(1)));
__device__ __forceinline__ void fetch_pair(const uint8_t* __restrict__ plane, int x0, int y, int W, int H, float& a, float& b) {
    a = 0.f; b = 0.f;
    if (y < 0 || y >= H) return;
    const uint8_t* row = plane + (long)y * W;
    if (x0 >= 0 && x0 + 1 < W) {
        const unsigned v = *reinterpret_cast<const u16_any*>(row + x0);
        a = (float)(v & 0xffu); b = (float)(v >> 8);
    } else {
        if (x0 >= 0 && x0 < W) a = (float)row[x0];
        if (x0 + 1 >= 0 && x0 + 1 < W) b = (float)row[x0 + 1];
    }
}

// hue shift through HSV (the formulas of torchvision's _rgb_to_hsv / _hsv_to_rgb)
__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float shift) {
    const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
    const bool eq = maxc == minc;
    const float cr = maxc - minc;
    const float s = cr / (eq ? 1.f : maxc);
    const float div = eq ? 1.f : cr;
    const float rc = (maxc - r) / div, gc = (maxc - g) / div, bc = (maxc - b) / div;
    const float hr = (maxc == r) ? bc - gc : 0.f;
    const float hg = (maxc == g && maxc != r) ? 2.f + rc - bc : 0.f;
    const float hb = (maxc != g && maxc != r) ? 4.f + gc - rc : 0.f;
    float h = fmodf((hr + hg + hb) / 6.f + 1.f, 1.f);
    h = h + shift;
    h = h - floorf(h);                       // python's % 1.0
    const float h6 = h * 6.f;
    const float fl = floorf(h6);
    const float f = h6 - fl;
    const int i = ((int)fl) % 6;
    const float v = maxc;
    const float p = clamp01(v * (1.f - s));
    const float q = clamp01(v * (1.f - s * f));
    const float t = clamp01(v * (1.f - s * (1.f - f)));
    switch (i) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

// A clip with neither contrast nor blur needs no second pass: its frames leave this kernel finished (out_f / out_u), and the
// finish kernel skips them.
__global__ __launch_bounds__(AUG_THREADS) void augment_warp_kernel(const uint8_t* __restrict__ x, float* __restrict__ tmp,
                                                                    float* __restrict__ partial, const float* __restrict__ params,
                                                                    float* __restrict__ out_f, uint8_t* __restrict__ out_u,
                                                                    int T, int Hs, int Ws, int Ho, int Wo, int nblk) {
    __shared__ float red[AUG_THREADS / 64];
    const int frame = blockIdx.y;
    const float* __restrict__ P = params + (long)(frame / T) * AUG_PARAMS;
    for (int vb = blockIdx.x * AUG_VB; vb < (int)(blockIdx.x + 1) * AUG_VB && vb < nblk; ++vb) {
    const int pix = vb * AUG_THREADS + threadIdx.x;
    const bool live = pix < Ho * Wo;
    float grey = 0.f;
    if (live) {
        const int oy = pix / Wo, ox = pix - oy * Wo;
        const float px = (float)ox + 0.5f, py = (float)oy + 0.5f;
        const float w = P[6] * px + P[7] * py + P[8];
        const float sx = (P[0] * px + P[1] * py + P[2]) / w - 0.5f;
        const float sy = (P[3] * px + P[4] * py + P[5]) / w - 0.5f;
        const uint8_t* __restrict__ src = x + (long)frame * 3 * Hs * Ws;
        float c[3];
        // (a map that leaves the frame, or a vanishing denominator, reads the fill value 0 like torchvision's fill=0)
        const bool finite = sx > -2.f && sx < (float)Ws + 1.f && sy > -2.f && sy < (float)Hs + 1.f;
        if (!finite) {
            c[0] = c[1] = c[2] = 0.f;
        } else if (P[9] != 0.f) {               // nearest (ties to even, like grid_sample)
            const int xi = (int)rintf(sx), yi = (int)rintf(sy);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) c[ch] = fetch(src + (long)ch * Hs * Ws, xi, yi, Ws, Hs) / 255.f;
        } else {
            const float x0f = floorf(sx), y0f = floorf(sy);
            const int x0 = (int)x0f, y0 = (int)y0f;
            const float wx = sx - x0f, wy = sy - y0f;
            const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy), w10 = (1.f - wx) * wy, w11 = wx * wy;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const uint8_t* pl = src + (long)ch * Hs * Ws;
                float v00, v01, v10, v11;
                fetch_pair(pl, x0, y0, Ws, Hs, v00, v01);
                fetch_pair(pl, x0, y0 + 1, Ws, Hs, v10, v11);
                c[ch] = (((v00 * w00 + v01 * w01) + v10 * w10) + v11 * w11) / 255.f;
            }
        }
        if (P[10] != 0.f) hue_shift(c[0], c[1], c[2], P[11]);
        if (P[12] != 0.f) {
            const float gsat = grey_of(c[0], c[1], c[2]);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) c[ch] = blend(c[ch], gsat, P[13]);
        }
        if (P[14] != 0.f) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) c[ch] = blend(c[ch], 0.f, P[15]);
        }
        const long o = (long)frame * 3 * Ho * Wo + pix;
        if (P[16] == 0.f && P[18] == 0.f) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                if (out_u) out_u[o + (long)ch * Ho * Wo] = (uint8_t)rintf(clamp01(c[ch]) * 255.f);
                else out_f[o + (long)ch * Ho * Wo] = (c[ch] - P[24 + ch]) / P[27 + ch];
            }
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) tmp[o + (long)ch * Ho * Wo] = c[ch];
            grey = grey_of(c[0], c[1], c[2]);
        }
    }
    if (P[16] == 0.f) continue;                // (the whole workgroup: P belongs to the frame) no contrast, no frame mean
    // fixed-order sum of the piece's grey values: lanes by xor butterfly, then the four waves in order
    grey = wave_sum(grey);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = grey;
    __syncthreads();
    if (threadIdx.x == 0) partial[(long)frame * nblk + vb] = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    }
}

template <bool U8>
__global__ __launch_bounds__(AUG_THREADS) void augment_finish_kernel(const float* __restrict__ tmp, const float* __restrict__ partial,
                                                                      const float* __restrict__ params, float* __restrict__ out_f,
                                                                      uint8_t* __restrict__ out_u, int T, int Ho, int Wo, int nblk) {
    __shared__ float mean_s;
    const int frame = blockIdx.y;
    const float* __restrict__ P = params + (long)(frame / T) * AUG_PARAMS;
    const bool contrast = P[16] != 0.f, blur = P[18] != 0.f;
    if (!contrast && !blur) return;            // finished by the warp kernel
    if (contrast && threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < nblk; ++i) s += partial[(long)frame * nblk + i];
        mean_s = s / (float)(Ho * Wo);
    }
    __syncthreads();
    for (int vb = blockIdx.x * AUG_VB; vb < (int)(blockIdx.x + 1) * AUG_VB && vb < nblk; ++vb) {
    const int pix = vb * AUG_THREADS + threadIdx.x;
    if (pix >= Ho * Wo) return;
    const int oy = pix / Wo, ox = pix - oy * Wo;
    const float cf = P[17];
    const float mean = contrast ? mean_s : 0.f;
    const float* __restrict__ src = tmp + (long)frame * 3 * Ho * Wo;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float* __restrict__ pl = src + (long)ch * Ho * Wo;
        float v;
        if (!blur) {
            v = pl[pix];
            if (contrast) v = blend(v, mean, cf);
        } else {
            v = 0.f;
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) {
                int yy = oy + dy - 2;
                yy = yy < 0 ? -yy : (yy >= Ho ? 2 * Ho - 2 - yy : yy);
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) {
                    int xx = ox + dx - 2;
                    xx = xx < 0 ? -xx : (xx >= Wo ? 2 * Wo - 2 - xx : xx);
                    float t = pl[(long)yy * Wo + xx];
                    if (contrast) t = blend(t, mean, cf);
                    v = v + (P[19 + dy] * P[19 + dx]) * t;
                }
            }
        }
        if constexpr (U8) {
            out_u[(long)frame * 3 * Ho * Wo + (long)ch * Ho * Wo + pix] = (uint8_t)rintf(clamp01(v) * 255.f);
        } else {
            out_f[(long)frame * 3 * Ho * Wo + (long)ch * Ho * Wo + pix] = (v - P[24 + ch]) / P[27 + ch];
        }
    }
    }
}

}  // namespace

extern "C" int64_t stes_augment_ws_floats(int n_frames, int Ho, int Wo) {
    const long nblk = stes_div_up((long)Ho * Wo, AUG_THREADS);
    return (long)n_frames * 3 * Ho * Wo + (long)n_frames * nblk;
}

extern "C" int stes_augment_u8(const uint8_t* x, float* out_f32, uint8_t* out_u8, const float* params, float* ws, int64_t ws_floats,
                               int n_clips, int T, int Hs, int Ws, int Ho, int Wo, void* stream) {
    STES_CHECK(x && params && ws && (out_f32 != nullptr) != (out_u8 != nullptr), STES_ERR_ARG,
               "stes_augment_u8: x, params, ws and exactly one of out_f32 / out_u8 are required");
    STES_CHECK(n_clips > 0 && T > 0 && Hs > 0 && Ws > 0 && Ho >= 3 && Wo >= 3, STES_ERR_ARG, "stes_augment_u8: bad sizes");
    const long n_frames = (long)n_clips * T;
    STES_CHECK(n_frames <= 65535, STES_ERR_ARG, "stes_augment_u8: at most 65535 frames per call");
    STES_CHECK((long)Hs * Ws < (1L << 30) && (long)Ho * Wo < (1L << 30), STES_ERR_ARG, "stes_augment_u8: frame too large");
    STES_CHECK(ws_floats >= stes_augment_ws_floats((int)n_frames, Ho, Wo), STES_ERR_ARG, "stes_augment_u8: workspace too small");
    const int nblk = stes_div_up((long)Ho * Wo, AUG_THREADS);
    float* tmp = ws;
    float* partial = ws + n_frames * 3 * Ho * Wo;
    const dim3 grid(stes_div_up(nblk, AUG_VB), (unsigned)n_frames);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(augment_warp_kernel, grid, dim3(AUG_THREADS), 0, s, x, tmp, partial, params, out_f32, out_u8, T, Hs, Ws, Ho, Wo, nblk);
    if (out_u8)
        hipLaunchKernelGGL(augment_finish_kernel<true>, grid, dim3(AUG_THREADS), 0, s, tmp, partial, params, out_f32, out_u8, T, Ho, Wo, nblk);
    else
        hipLaunchKernelGGL(augment_finish_kernel<false>, grid, dim3(AUG_THREADS), 0, s, tmp, partial, params, out_f32, out_u8, T, Ho, Wo, nblk);
    STES_LAUNCH_CHECK();
    return STES_OK;
}
