// Grouped 3x3 convolution (groups of 16 channels) with stride 2, forward, for the first block of a stage (timm Bottleneck conv2
// as the reference builds it, train_e2e_spatial.py:305-314): input = raw output of the convolution in front, BatchNorm + ReLU
// applied on the way in; output raw, with its per-channel sum and sum of squares for the BatchNorm behind it.
// A workgroup (4 waves) owns 64 channels (grouped convolution never mixes them, so a pixel's 128 bytes are whole cache lines)
// and walks an image's output rows.  Per output row ro:
//   (a) input rows 2ro and 2ro+1 come from memory as 16-byte chunks (requested two steps ahead), are transformed and go to a
//       ring of four row slots in LDS (row 2ro-1 is there from the step before, row -1 is zeros; a zero pixel either side);
//   (b) the output row is convolved from three slots: wave = group, lane = output column, nine K = 16 MFMAs per 16-column
//       fragment on 8-byte LDS reads -- with stride 2 an input pixel is read 2.25 times, against ten times for an im2col
//       operand -- (pixel stride padded by 8 bytes: two pixels apart is four banks apart); the result leaves through a
//       staging row as 16-byte chunks; the statistics are summed in registers over all the workgroup's images and committed
//       once (one atomic per channel and workgroup).
#include "stes_common.h"
#include "stes_conv_ring.h"
#include <stdlib.h>

namespace {

typedef short r_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned r_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned r_u32x4 __attribute__((ext_vector_type(4)));
#define CR_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float cr_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}
// A operand of one tap from the k-step-pair pack of stes_pack_conv3x3g ([G][5][64][8])
__device__ __forceinline__ r_s16x4 cr_tap_frag(const bf16* __restrict__ wp, int g, int slot, int lane) {
    const int m = lane & 15, q = lane >> 4;
    const int packed_lane = ((slot & 1) * 2 + (q >> 1)) * 16 + m;
    return *reinterpret_cast<const r_s16x4*>(wp + (((long)g * 5 + (slot >> 1)) * 64 + packed_lane) * 8 + 4 * (q & 1));
}

template <int WI>
struct RingShape {
    static constexpr int WO = WI / 2, CF = (WO + 15) / 16;
    static constexpr int PXS = 64 + 4;                      // ring: elements per pixel (8 bytes of padding)
    static constexpr int SLOT = (WI + 2) * PXS;             // a row slot with a zero pixel either side
    static constexpr int SPX = 64 + 8;                      // staging row: elements per pixel
    static constexpr int CHUNKS = 2 * WI * 8, CPT = (CHUNKS + 255) / 256;      // 16-byte chunks of a step's two input rows
    static constexpr size_t LDS = (size_t)(4 * SLOT + WO * SPX) * 2;
};

// POOL (evaluation): the squeeze-excite squeeze of every image on the way out -- mean over its pixels of relu(pool_scale * out +
// pool_shift), bf16 [N][C].  A wave owns its 16 channels for the whole image, so the sum needs no other wave.
template <int WI, bool POOL>
__global__ __launch_bounds__(256) void conv_ring2_kernel(const bf16* __restrict__ in, const bf16* __restrict__ wp, bf16* __restrict__ out,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         float* col_sum, float* col_sumsq, const float* __restrict__ pool_scale,
                                                         const float* __restrict__ pool_shift, bf16* __restrict__ pool_out, float pool_norm,
                                                         int N, int HI, int C, StesDet det) {
    typedef RingShape<WI> S;
    extern __shared__ __attribute__((aligned(16))) unsigned char cr_smem[];
    bf16* ring = reinterpret_cast<bf16*>(cr_smem);                          // [4][SLOT]
    bf16* stage = ring + 4 * S::SLOT;                                       // [WO][SPX]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int HO = HI / 2, slice = blockIdx.y, cb = slice * 64;
    for (int i = tid; i < 4 * S::SLOT / 2; i += 256) reinterpret_cast<unsigned*>(ring)[i] = 0u;     // borders stay zero for good
    r_s16x4 wA[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) wA[a][b] = cr_tap_frag(wp, slice * 4 + wave, a * 3 + b, lane);
    // the input transform of this thread's eight channels (a thread's chunks all have channel piece tid & 7)
    const int chunk = tid & 7;
    const bool affine = scale != nullptr;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = affine ? scale[cb + chunk * 8 + e] : 1.f; sh[e] = affine ? shift[cb + chunk * 8 + e] : 0.f; }
    const int ch0 = wave * 16 + 4 * fq;                                     // this lane's output channels within the slice
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
    float ps[4], pb[4], psum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) { ps[e] = POOL ? pool_scale[cb + ch0 + e] : 0.f; pb[e] = POOL ? pool_shift[cb + ch0 + e] : 0.f; }
    __syncthreads();

    // the workgroup's steps (image, output row) in one sequence, input rows requested two steps ahead -- across images too
    const int nimg = blockIdx.x < N ? (N - 1 - blockIdx.x) / gridDim.x + 1 : 0;
    const int total = nimg * HO;
    int fn = blockIdx.x, fro = 0;                                           // the step the next request is for
    // Every request is made by every thread on every step (threads without a chunk and steps past the workgroup's last
    // repeat a valid address and drop the result): the wait in front of a buffer's use is then counted in instructions --
    // "all but the other buffer's requests" -- where a request under a condition would make it "everything outstanding",
    // the rows asked for a moment ago included.
    auto fetch = [&](bf16x8 (&dst)[S::CPT]) {
        const bf16* src = in + ((long)min(fn, N - 1) * HI + 2 * fro) * WI * C + cb + chunk * 8;
#pragma unroll
        for (int k = 0; k < S::CPT; ++k) {
            const int i = k < S::CHUNKS / 256 ? tid + 256 * k : min(tid + 256 * k, S::CHUNKS - 1);        // chunk i: pixel i >> 3 of the two-row strip
            dst[k] = *reinterpret_cast<const bf16x8*>(src + (long)(i >> 3) * C);
        }
        if (++fro == HO) { fro = 0; fn += gridDim.x; }
    };
    int n = blockIdx.x, ro = 0, pn = 0, pro = -1;                           // this step, the step before (-1: nothing staged)
    auto flush = [&]() {                                                    // the staged output row of the step before leaves as whole chunks
        if (pro >= 0) {
            bf16* dst = out + (((long)pn * HO + pro) * S::WO) * C + cb;
            for (int idx = tid; idx < S::WO * 8; idx += 256) {
                const int co = idx >> 3, c0 = (idx & 7) * 8;
                *reinterpret_cast<bf16x8*>(dst + (long)co * C + c0) = *reinterpret_cast<const bf16x8*>(stage + co * S::SPX + c0);
            }
        }
    };
    auto step = [&](bf16x8 (&raw)[S::CPT], bool live) {
        if (ro == 0) {                                                      // row -1 of an image: zeros (slot 0 holds rows 3, 7, .. later)
            for (int i = tid; i < WI * 32; i += 256) reinterpret_cast<unsigned*>(ring + ((i >> 5) + 1) * S::PXS)[i & 31] = 0u;
        }
        // (a) rows 2ro and 2ro+1, transformed, into their slots (row r lives in slot (r + 1) & 3)
#pragma unroll
        for (int k = 0; k < S::CPT; ++k) {
            const int i = tid + 256 * k;
            if (i < S::CHUNKS) {
                const int p = i >> 3, row = p >= WI ? 1 : 0, px = p - row * WI;
                bf16x8 v = raw[k];
                if (affine) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (bf16)fmaxf((float)v[e] * sc[e] + sh[e], 0.f);
                }
                const r_u32x4 w = __builtin_bit_cast(r_u32x4, v);
                bf16* dstp = ring + ((2 * ro + row + 1) & 3) * S::SLOT + (px + 1) * S::PXS + chunk * 8;
                *reinterpret_cast<r_u32x2*>(dstp) = (r_u32x2){w[0], w[1]};
                *reinterpret_cast<r_u32x2*>(dstp + 4) = (r_u32x2){w[2], w[3]};
            }
        }
        fetch(raw);                                                         // the rows of two steps on
        flush();
        __syncthreads();
        // (b) output row ro: source rows 2ro-1, 2ro, 2ro+1 = slots (2ro + a) & 3
        f32x4 acc[S::CF];
#pragma unroll
        for (int f = 0; f < S::CF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const bf16* row = ring + ((2 * ro + a) & 3) * S::SLOT + ch0;
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int f = 0; f < S::CF; ++f) {
                    const int pxi = min(2 * (16 * f + fr) + b, WI + 1);      // ring pixel index of column 2co + b - 1
                    const r_u32x2 rawb = *reinterpret_cast<const r_u32x2*>(row + pxi * S::PXS);
                    acc[f] = CR_MFMA16(wA[a][b], __builtin_bit_cast(r_s16x4, rawb), acc[f]);
                }
        }
#pragma unroll
        for (int f = 0; f < S::CF; ++f) {
            const int co = 16 * f + fr;
            if (co < S::WO) {
                bf16x4 ob;
#pragma unroll
                for (int e = 0; e < 4; ++e) ob[e] = (bf16)acc[f][e];
                *reinterpret_cast<bf16x4*>(stage + co * S::SPX + ch0) = ob;
                if (live) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float fv = (float)ob[e];
                        if constexpr (POOL) psum[e] += fmaxf(fv * ps[e] + pb[e], 0.f);
                        else { ssum[e] += fv; ssq[e] += fv * fv; }
                    }
                }
            }
        }
        if constexpr (POOL) {
            if (live && ro == HO - 1) {                                    // the image's squeeze
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = cr_row16_sum(psum[e]) * pool_norm;
                    if (fr == 0) pool_out[(long)n * C + cb + ch0 + e] = (bf16)t;
                    psum[e] = 0.f;
                }
            }
        }
        __syncthreads();
        pn = n; pro = live ? ro : -1;
        if (++ro == HO) { ro = 0; n += gridDim.x; }
    };
    bf16x8 rawA[S::CPT], rawB[S::CPT];
    fetch(rawA);
    fetch(rawB);
#pragma unroll 1
    for (int s = 0; s < total; s += 2) {                                    // (an odd count ends with a step whose results are dropped)
        step(rawA, true);
        step(rawB, s + 1 < total);
    }
    flush();
    if (col_sum) {
        // one 64-lane atomic per workgroup and array (a lane per channel): the workgroups end together and their additions to
        // the same 64 addresses queue up at the L2, so every instruction less is time off the kernel's tail
        float* red = reinterpret_cast<float*>(stage);                       // [2][64]
        __syncthreads();
        float a[4], b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = cr_row16_sum(ssum[e]); b[e] = cr_row16_sum(ssq[e]); }
        if (fr == 0) {
            *reinterpret_cast<f32x4*>(red + ch0) = (f32x4){a[0], a[1], a[2], a[3]};
            *reinterpret_cast<f32x4*>(red + 64 + ch0) = (f32x4){b[0], b[1], b[2], b[3]};
        }
        __syncthreads();
        det_enter(det);
        if (wave == 0) {
            atomicAdd(col_sum + cb + lane, red[lane]);
            atomicAdd(col_sumsq + cb + lane, red[64 + lane]);
        }
        det_leave(det);
    }
}

// ---------------------------------------------------------------------------------------------- input gradient
// dX of the same convolution: dX[h][w] gathers dY[(h + 1 - a) / 2][(w + 1 - b) / 2] over the taps (a, b) whose parities
// match, so the 2 x 2 output pixels under dY pixel (y, j) take 1, 2, 2 and 4 taps -- nine K = 16 MFMAs on four 8-byte reads
// (dY pixels j and j + 1 of rows y and y + 1), nothing multiplied by zero.  The walk is the forward kernel's: per step one new
// dY row enters a ring of three slots (a zero pixel at the right, a zero row after the last), two output rows are staged
// (even and odd columns in planes of their own: lane = j writes without bank conflicts) and leave as 16-byte chunks, on the
// way masked by the ReLU in front (raw input of that ReLU and its BatchNorm affine given) and summed for the BatchNorm
// backward: s_dy += dX, s_dyr += dX * raw.  dY rows and the raw rows under the mask are requested two steps ahead.
template <int WI>
struct DgradShape {
    static constexpr int WO = WI / 2, CF = (WO + 15) / 16;
    static constexpr int PXS = 64 + 8;                      // elements per pixel, ring and staging: 36 banks, so the 32 lanes
                                                            // of an 8-byte access pass fall on 32 different bank pairs
    static constexpr int SLOT = (WO + 1) * PXS;
    static constexpr int YCH = 2 * WO * 8, YPT = (YCH + 255) / 256;     // chunks of dY rows y (an image's first step only) and y + 1
    static constexpr int MCH = 2 * WI * 8, MPT = (MCH + 255) / 256;     // chunks of a step's two output rows
    static constexpr size_t LDS = (size_t)(3 * SLOT + 4 * WO * PXS) * 2;
};

template <int WI>
__global__ __launch_bounds__(256) void dgrad_ring2_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ wp, bf16* __restrict__ dx,
                                                          const bf16* __restrict__ mask_raw, const float* __restrict__ mscale,
                                                          const float* __restrict__ mshift, float* s_dy, float* s_dyr, int N, int HI, int C,
                                                          StesDet det) {
    typedef DgradShape<WI> S;
    extern __shared__ __attribute__((aligned(16))) unsigned char cr_smem[];
    bf16* ring = reinterpret_cast<bf16*>(cr_smem);                          // [3][SLOT]
    bf16* stage = ring + 3 * S::SLOT;                                       // [row parity][column parity][WO][PXS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int HO = HI / 2, slice = blockIdx.y, cb = slice * 64;
    for (int i = tid; i < 3 * S::SLOT / 2; i += 256) reinterpret_cast<unsigned*>(ring)[i] = 0u;     // the right-hand pixels stay zero
    // taps by (k-step, half) of the parity-class pack of stes_pack_conv3x3g: (1,1) | (1,0),(1,2) | (0,1),(2,1) | (0,0),(0,2) | (2,0),(2,2)
    const int g = slice * 4 + wave;
    const r_s16x4 w11 = cr_tap_frag(wp, g, 0, lane), w10 = cr_tap_frag(wp, g, 2, lane), w12 = cr_tap_frag(wp, g, 3, lane);
    const r_s16x4 w01 = cr_tap_frag(wp, g, 4, lane), w21 = cr_tap_frag(wp, g, 5, lane), w00 = cr_tap_frag(wp, g, 6, lane);
    const r_s16x4 w02 = cr_tap_frag(wp, g, 7, lane), w20 = cr_tap_frag(wp, g, 8, lane), w22 = cr_tap_frag(wp, g, 9, lane);
    const int chunk = tid & 7;
    const bool masked = mask_raw != nullptr;
    float sc[8], sh[8], ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = masked ? mscale[cb + chunk * 8 + e] : 0.f; sh[e] = masked ? mshift[cb + chunk * 8 + e] : 0.f;
        ssum[e] = 0.f; ssq[e] = 0.f;
    }
    const int ch0 = wave * 16 + 4 * fq;
    __syncthreads();

    const int nimg = blockIdx.x < N ? (N - 1 - blockIdx.x) / gridDim.x + 1 : 0;
    const int total = nimg * HO;
    int yn = blockIdx.x, yy = 0;                                            // the step the next dY request is for
    auto fetch_y = [&](bf16x8 (&dst)[S::YPT]) {
        if (yn < N) {
            const bf16* src = dy + ((long)yn * HO + yy) * S::WO * C + cb + chunk * 8;
#pragma unroll
            for (int k = 0; k < S::YPT; ++k) {
                const int i = tid + 256 * k;
                if (i < S::YCH) {
                    const int p = i >> 3;                                   // pixel of the strip of rows yy and yy + 1
                    dst[k] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                    if (p < S::WO ? yy == 0 : yy + 1 < HO) dst[k] = *reinterpret_cast<const bf16x8*>(src + (long)p * C);
                }
            }
        }
        if (++yy == HO) { yy = 0; yn += gridDim.x; }
    };
    int mn = blockIdx.x, my = 0;                                            // the step the next request of raw rows is for
    auto fetch_m = [&](bf16x8 (&dst)[S::MPT]) {
        if (masked && mn < N) {
            const bf16* src = mask_raw + ((long)mn * HI + 2 * my) * WI * C + cb + chunk * 8;
#pragma unroll
            for (int k = 0; k < S::MPT; ++k) {
                const int i = tid + 256 * k;
                if (i < S::MCH) dst[k] = *reinterpret_cast<const bf16x8*>(src + (long)(i >> 3) * C);
            }
        }
        if (++my == HO) { my = 0; mn += gridDim.x; }
    };
    int n = blockIdx.x, y = 0, s0 = 0;                                      // this step; ring slot of dY row y
    auto step = [&](bf16x8 (&raw)[S::YPT], bf16x8 (&m)[S::MPT]) {
        if (y == 0) s0 = 0;
        const int s1 = s0 == 2 ? 0 : s0 + 1;
        // (a) dY row y + 1 (zeros after the last row) into the slot behind row y's; row y itself on an image's first step
#pragma unroll
        for (int k = 0; k < S::YPT; ++k) {
            const int i = tid + 256 * k;
            if (i < S::YCH) {
                const int p = i >> 3, row = p >= S::WO ? 1 : 0, px = p - row * S::WO;
                if (row == 1 || y == 0) *reinterpret_cast<bf16x8*>(ring + (row ? s1 : s0) * S::SLOT + px * S::PXS + chunk * 8) = raw[k];
            }
        }
        fetch_y(raw);
        __syncthreads();
        // (b) output rows 2y and 2y + 1
        const bf16* rowy = ring + s0 * S::SLOT + ch0;
        const bf16* rowy1 = ring + s1 * S::SLOT + ch0;
#pragma unroll
        for (int f = 0; f < S::CF; ++f) {
            const int j = 16 * f + fr, jc = min(j, S::WO - 1);
            const r_s16x4 r0 = *reinterpret_cast<const r_s16x4*>(rowy + jc * S::PXS);
            const r_s16x4 r1 = *reinterpret_cast<const r_s16x4*>(rowy + (jc + 1) * S::PXS);
            const r_s16x4 t0 = *reinterpret_cast<const r_s16x4*>(rowy1 + jc * S::PXS);
            const r_s16x4 t1 = *reinterpret_cast<const r_s16x4*>(rowy1 + (jc + 1) * S::PXS);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            f32x4 ee = CR_MFMA16(w11, r0, z);
            f32x4 eo = CR_MFMA16(w10, r1, z);
            f32x4 oe = CR_MFMA16(w01, t0, z);
            f32x4 oo = CR_MFMA16(w00, t1, z);
            eo = CR_MFMA16(w12, r0, eo);
            oe = CR_MFMA16(w21, r0, oe);
            oo = CR_MFMA16(w02, t0, oo);
            oo = CR_MFMA16(w20, r1, oo);
            oo = CR_MFMA16(w22, r0, oo);
            if (j < S::WO) {
                bf16x4 o0, o1, o2, o3;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o0[e] = (bf16)ee[e]; o1[e] = (bf16)eo[e]; o2[e] = (bf16)oe[e]; o3[e] = (bf16)oo[e]; }
                bf16* at = stage + j * S::PXS + ch0;
                *reinterpret_cast<bf16x4*>(at) = o0;
                *reinterpret_cast<bf16x4*>(at + S::WO * S::PXS) = o1;
                *reinterpret_cast<bf16x4*>(at + 2 * S::WO * S::PXS) = o2;
                *reinterpret_cast<bf16x4*>(at + 3 * S::WO * S::PXS) = o3;
            }
        }
        __syncthreads();
        // the two rows leave: 2 * WI pixels, contiguous in memory
        bf16* dst = dx + ((long)n * HI + 2 * y) * WI * C + cb + chunk * 8;
#pragma unroll
        for (int k = 0; k < S::MPT; ++k) {
            const int i = tid + 256 * k;
            if (i < S::MCH) {
                const int p = i >> 3, row = p >= WI ? 1 : 0, w = p - row * WI;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + ((row * 2 + (w & 1)) * S::WO + (w >> 1)) * S::PXS + chunk * 8);
                if (masked) {
                    const bf16x8 r = m[k];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float rf = (float)r[e];
                        if (!(rf * sc[e] + sh[e] > 0.f)) v[e] = (bf16)0.f;
                        const float fv = (float)v[e];
                        ssum[e] += fv; ssq[e] += fv * rf;
                    }
                }
                *reinterpret_cast<bf16x8*>(dst + (long)p * C) = v;
            }
        }
        fetch_m(m);
        s0 = s1;
        if (++y == HO) { y = 0; n += gridDim.x; }
    };
    bf16x8 rawA[S::YPT], rawB[S::YPT], mA[S::MPT], mB[S::MPT];
    fetch_y(rawA);
    fetch_m(mA);
    fetch_y(rawB);
    fetch_m(mB);
#pragma unroll 1
    for (int s = 0; s < total; s += 2) {
        step(rawA, mA);
        if (s + 1 < total) step(rawB, mB);
    }
    if (masked) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {                                      // lanes with the same eight channels
            ssum[e] += __shfl_xor(ssum[e], 8, 64); ssum[e] += __shfl_xor(ssum[e], 16, 64); ssum[e] += __shfl_xor(ssum[e], 32, 64);
            ssq[e] += __shfl_xor(ssq[e], 8, 64); ssq[e] += __shfl_xor(ssq[e], 16, 64); ssq[e] += __shfl_xor(ssq[e], 32, 64);
        }
        float* red = reinterpret_cast<float*>(ring);                        // [4 waves][2][64]
        __syncthreads();
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[wave * 128 + lane * 8 + e] = ssum[e]; red[wave * 128 + 64 + lane * 8 + e] = ssq[e]; }
        }
        __syncthreads();
        det_enter(det);
        if (wave == 0) {                                                    // one 64-lane atomic per workgroup and array, as in the forward
            atomicAdd(s_dy + cb + lane, (red[lane] + red[128 + lane]) + (red[256 + lane] + red[384 + lane]));
            atomicAdd(s_dyr + cb + lane, (red[64 + lane] + red[192 + lane]) + (red[320 + lane] + red[448 + lane]));
        }
        det_leave(det);
    }
}

template <int WI>
int dgrad_launch(const uint16_t* dy, const uint16_t* wp, uint16_t* dx, const uint16_t* mask_raw, const float* mscale, const float* mshift,
                 float* s_dy, float* s_dyr, int N, int Hi, int C, void* stream) {
    typedef DgradShape<WI> S;
    static_assert(S::LDS >= 4 * 128 * sizeof(float) && 3 * S::SLOT * 2 >= 4 * 128 * sizeof(float), "the statistics reduction reuses the ring");
    static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad_ring2_kernel<WI>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    STES_CHECK(big_lds, STES_ERR_UNSUPPORTED, "stes_conv3x3g_dgrad: cannot raise the dynamic LDS limit");
    static int resident = 0;
    if (!resident) {
        int per_cu = 0, dev = 0, cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dgrad_ring2_kernel<WI>, 256, S::LDS) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        resident = per_cu * (cus < 1 ? 256 : cus);
    }
    const int slices = C / 64;
    int gx = resident / slices;
    if (gx < 1) gx = 1;
    if (gx > N) gx = N;
    hipLaunchKernelGGL((dgrad_ring2_kernel<WI>), dim3(gx, slices), dim3(256), S::LDS, (hipStream_t)stream, (const bf16*)dy, (const bf16*)wp, (bf16*)dx,
                       (const bf16*)mask_raw, mscale, mshift, s_dy, s_dyr, N, Hi, C, mask_raw ? stes_det_take(gx * slices) : StesDet{nullptr, 0u, 0u});
    STES_LAUNCH_CHECK();
    return STES_OK;
}

// ---------------------------------------------------------------------------------------------- stride 1
// The identity blocks' convolution (maps 28, 14 and 7 wide), forward and input gradient in one kernel: the input gradient of a
// stride-1 convolution is the same convolution with the taps mirrored (and the channel roles exchanged, which the weight pack
// for the gradient already holds).  R output rows per step from a ring of R + 2 input rows (slots: a power of two); a 7-wide
// map puts two output rows into one 16-lane fragment.  Forward: BatchNorm + ReLU on the way in, sum and sum of squares of the
// output on the way out.  Gradient: nothing on the way in; the ReLU mask and s_dy += dX, s_dyr += dX * raw on the way out.
template <int W, int R>
struct Ring1Shape {
    static constexpr bool PACK = W <= 8;                    // two output rows per fragment
    static constexpr int CF = (W + 15) / 16;
    static constexpr int NS = R + 2 <= 4 ? 4 : R + 2 <= 8 ? 8 : 16;          // ring slots; input row r lives in slot (r + 1) & (NS - 1)
    static_assert(R % 2 == 0 && R + 2 <= NS, "a step's R + 2 input rows are in the ring together");
    static constexpr int PXS = 64 + 8;
    static constexpr int SLOT = (W + 2) * PXS;
    static constexpr int ICH = (R + 1) * W * 8, IPT = (ICH + 255) / 256;      // chunks of input rows R t (an image's first step only) .. R t + R
    static constexpr int OCH = R * W * 8, OPT = (OCH + 255) / 256;            // chunks of a step's output rows
    static constexpr size_t LDS = (size_t)(NS * SLOT + R * W * PXS) * 2;
};

template <int W, int R, bool DGRAD, bool POOL>
__global__ __launch_bounds__(256) void conv_ring1_kernel(const bf16* __restrict__ in, const bf16* __restrict__ wp, bf16* __restrict__ out,
                                                         const bf16* __restrict__ mask_raw, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* sum_a, float* sum_b,
                                                         const float* __restrict__ pool_scale, const float* __restrict__ pool_shift,
                                                         bf16* __restrict__ pool_out, float pool_norm, int N, int H, int C, StesDet det) {
    static_assert(!(DGRAD && POOL), "the squeeze belongs to the forward");
    typedef Ring1Shape<W, R> S;
    extern __shared__ __attribute__((aligned(16))) unsigned char cr_smem[];
    bf16* ring = reinterpret_cast<bf16*>(cr_smem);                          // [NS][SLOT]
    bf16* stage = ring + S::NS * S::SLOT;                                   // [R][W][PXS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int slice = blockIdx.y, cb = slice * 64, HT = (H + R - 1) / R;    // steps per image
    for (int i = tid; i < S::NS * S::SLOT / 2; i += 256) reinterpret_cast<unsigned*>(ring)[i] = 0u;     // the border pixels stay zero
    r_s16x4 wA[3][3];                                                       // wA[a][b] multiplies input row h + a - 1, column w + b - 1
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) wA[a][b] = cr_tap_frag(wp, slice * 4 + wave, DGRAD ? 8 - (a * 3 + b) : a * 3 + b, lane);
    const int chunk = tid & 7;
    const bool affine = scale != nullptr;                                   // forward: the input transform; gradient: the mask's
    const bool masked = DGRAD && mask_raw != nullptr;
    const bool stats = sum_a != nullptr;
    float sc[8], sh[8], ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = affine ? scale[cb + chunk * 8 + e] : 1.f; sh[e] = affine ? shift[cb + chunk * 8 + e] : 0.f;
        ssum[e] = 0.f; ssq[e] = 0.f;
    }
    const int ch0 = wave * 16 + 4 * fq;
    float ps[4], pb[4], psum[4] = {0.f, 0.f, 0.f, 0.f};                     // POOL: see conv_ring2_kernel
#pragma unroll
    for (int e = 0; e < 4; ++e) { ps[e] = POOL ? pool_scale[cb + ch0 + e] : 0.f; pb[e] = POOL ? pool_shift[cb + ch0 + e] : 0.f; }
    __syncthreads();

    const int nimg = blockIdx.x < N ? (N - 1 - blockIdx.x) / gridDim.x + 1 : 0;
    const int total = nimg * HT;
    int in_n = blockIdx.x, in_t = 0;                                        // the step the next request of input rows is for
    // (every request is made by every thread on every step and unwanted results are dropped, so that the waits are counted
    // in instructions: see the stride-2 forward kernel)
    auto fetch_in = [&](bf16x8 (&dst)[S::IPT]) {
        const bf16* src = in + ((long)min(in_n, N - 1) * H + R * in_t) * W * C + cb + chunk * 8;
        const int last = (H - R * in_t) * W - 1;                            // last pixel from row R t on inside the image
#pragma unroll
        for (int k = 0; k < S::IPT; ++k) {
            const int i = k < S::ICH / 256 ? tid + 256 * k : min(tid + 256 * k, S::ICH - 1);
            const int p = i >> 3, c = p / W;                                // chunk row c: input row R t + c
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (long)min(p, last) * C);
            dst[k] = (c == 0 ? in_t == 0 : R * in_t + c < H) ? v : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
        if (++in_t == HT) { in_t = 0; in_n += gridDim.x; }
    };
    int m_n = blockIdx.x, m_t = 0;                                          // the step the next request of raw rows under the mask is for
    const bf16* mbase = masked ? mask_raw : in;                             // (without a mask: any readable map of the size; dropped)
    auto fetch_m = [&](bf16x8 (&dst)[S::OPT]) {
        const bf16* src = mbase + ((long)min(m_n, N - 1) * H + R * m_t) * W * C + cb + chunk * 8;
        const int last = min(R, H - R * m_t) * W - 1;
#pragma unroll
        for (int k = 0; k < S::OPT; ++k) {
            const int i = k < S::OCH / 256 ? tid + 256 * k : min(tid + 256 * k, S::OCH - 1);
            dst[k] = *reinterpret_cast<const bf16x8*>(src + (long)min(i >> 3, last) * C);
        }
        if (++m_t == HT) { m_t = 0; m_n += gridDim.x; }
    };
    int n = blockIdx.x, t = 0;
    auto step = [&](bf16x8 (&raw)[S::IPT], bf16x8 (&m)[S::OPT], bool live) {
        const int h0 = R * t;                                               // first output row of the step
        if (t == 0) {                                                       // row -1 of an image: zeros
            for (int i = tid; i < W * 32; i += 256) reinterpret_cast<unsigned*>(ring + ((i >> 5) + 1) * S::PXS)[i & 31] = 0u;
        }
        // (a) input rows h0 + 1 .. h0 + R (zeros below the image) into the slots of rows no longer needed; row h0 on a first step
#pragma unroll
        for (int k = 0; k < S::IPT; ++k) {
            const int i = tid + 256 * k;
            if (i < S::ICH) {
                const int p = i >> 3, c = p / W, px = p - c * W;
                if (c > 0 || t == 0) {
                    bf16x8 v = raw[k];
                    if (!DGRAD && affine && h0 + c < H) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (bf16)fmaxf((float)v[e] * sc[e] + sh[e], 0.f);
                    }
                    *reinterpret_cast<bf16x8*>(ring + ((h0 + c + 1) & (S::NS - 1)) * S::SLOT + (px + 1) * S::PXS + chunk * 8) = v;
                }
            }
        }
        fetch_in(raw);
        __syncthreads();
        // (b) output rows h0 .. h0 + R - 1: row h of the output reads input rows h - 1, h, h + 1 = slots (h + a) & (NS - 1)
        if constexpr (S::PACK) {
#pragma unroll
            for (int u = 0; u < R / 2; ++u) {
                const int rr = 2 * u + (fr >> 3), col = min(fr & 7, W - 1);
                if (h0 + 2 * u < H) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const bf16* row = ring + ((h0 + rr + a) & (S::NS - 1)) * S::SLOT + col * S::PXS + ch0;
#pragma unroll
                        for (int b = 0; b < 3; ++b) acc = CR_MFMA16(wA[a][b], *reinterpret_cast<const r_s16x4*>(row + b * S::PXS), acc);
                    }
                    if ((fr & 7) < W) {
                        bf16x4 ob;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ob[e] = (bf16)acc[e];
                        *reinterpret_cast<bf16x4*>(stage + (rr * W + col) * S::PXS + ch0) = ob;
                        if constexpr (POOL) {
                            if (live && h0 + rr < H) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) psum[e] += fmaxf((float)ob[e] * ps[e] + pb[e], 0.f);
                            }
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                if (h0 + rr < H) {
                    f32x4 acc[S::CF];
#pragma unroll
                    for (int f = 0; f < S::CF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const bf16* row = ring + ((h0 + rr + a) & (S::NS - 1)) * S::SLOT + ch0;
#pragma unroll
                        for (int b = 0; b < 3; ++b)
#pragma unroll
                            for (int f = 0; f < S::CF; ++f) {
                                const int col = min(16 * f + fr, W - 1);
                                acc[f] = CR_MFMA16(wA[a][b], *reinterpret_cast<const r_s16x4*>(row + (col + b) * S::PXS), acc[f]);
                            }
                    }
#pragma unroll
                    for (int f = 0; f < S::CF; ++f) {
                        const int col = 16 * f + fr;
                        if (col < W) {
                            bf16x4 ob;
#pragma unroll
                            for (int e = 0; e < 4; ++e) ob[e] = (bf16)acc[f][e];
                            *reinterpret_cast<bf16x4*>(stage + (rr * W + col) * S::PXS + ch0) = ob;
                            if constexpr (POOL) {
                                if (live) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) psum[e] += fmaxf((float)ob[e] * ps[e] + pb[e], 0.f);
                                }
                            }
                        }
                    }
                }
            }
        }
        if constexpr (POOL) {
            if (live && t == HT - 1) {                                     // the image's squeeze
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s = cr_row16_sum(psum[e]) * pool_norm;
                    if (fr == 0) pool_out[(long)n * C + cb + ch0 + e] = (bf16)s;
                    psum[e] = 0.f;
                }
            }
        }
        __syncthreads();
        // the rows leave as 16-byte chunks, contiguous in memory
        bf16* dst = out + ((long)n * H + h0) * W * C + cb + chunk * 8;
        const int pix = live ? min(R, H - h0) * W : 0;
#pragma unroll
        for (int k = 0; k < S::OPT; ++k) {
            const int i = tid + 256 * k;
            if (i < S::OCH && (i >> 3) < pix) {
                const int p = i >> 3;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + p * S::PXS + chunk * 8);
                if (masked) {
                    const bf16x8 r = m[k];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float rf = (float)r[e];
                        if (!(rf * sc[e] + sh[e] > 0.f)) v[e] = (bf16)0.f;
                        const float fv = (float)v[e];
                        ssum[e] += fv; ssq[e] += fv * rf;
                    }
                } else if (stats) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float fv = (float)v[e]; ssum[e] += fv; ssq[e] += fv * fv; }
                }
                *reinterpret_cast<bf16x8*>(dst + (long)p * C) = v;
            }
        }
        fetch_m(m);
        if (++t == HT) { t = 0; n += gridDim.x; }
    };
    bf16x8 rawA[S::IPT], rawB[S::IPT], mA[S::OPT], mB[S::OPT];
    fetch_in(rawA);
    fetch_m(mA);
    fetch_in(rawB);
    fetch_m(mB);
#pragma unroll 1
    for (int s = 0; s < total; s += 2) {                                    // (an odd count ends with a step whose results are dropped)
        step(rawA, mA, true);
        step(rawB, mB, s + 1 < total);
    }
    if (stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {                                      // lanes with the same eight channels
            ssum[e] += __shfl_xor(ssum[e], 8, 64); ssum[e] += __shfl_xor(ssum[e], 16, 64); ssum[e] += __shfl_xor(ssum[e], 32, 64);
            ssq[e] += __shfl_xor(ssq[e], 8, 64); ssq[e] += __shfl_xor(ssq[e], 16, 64); ssq[e] += __shfl_xor(ssq[e], 32, 64);
        }
        float* red = reinterpret_cast<float*>(ring);                        // [4 waves][2][64]
        __syncthreads();
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[wave * 128 + lane * 8 + e] = ssum[e]; red[wave * 128 + 64 + lane * 8 + e] = ssq[e]; }
        }
        __syncthreads();
        det_enter(det);
        if (wave == 0) {
            atomicAdd(sum_a + cb + lane, (red[lane] + red[128 + lane]) + (red[256 + lane] + red[384 + lane]));
            atomicAdd(sum_b + cb + lane, (red[64 + lane] + red[192 + lane]) + (red[320 + lane] + red[448 + lane]));
        }
        det_leave(det);
    }
}

template <int W, int R, bool DGRAD, bool POOL = false>
int ring1_launch(const uint16_t* in, const uint16_t* wp, uint16_t* out, const uint16_t* mask_raw, const float* scale, const float* shift,
                 float* sum_a, float* sum_b, int N, int H, int C, void* stream, const float* pool_scale = nullptr,
                 const float* pool_shift = nullptr, uint16_t* pool_out = nullptr, float pool_norm = 0.f) {
    typedef Ring1Shape<W, R> S;
    static_assert(S::NS * S::SLOT * 2 >= 4 * 128 * sizeof(float), "the statistics reduction reuses the ring");
    static_assert(S::LDS <= 64 * 1024, "within the default dynamic LDS limit");
    static int resident = 0;
    if (!resident) {
        int per_cu = 0, dev = 0, cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_ring1_kernel<W, R, DGRAD, POOL>, 256, S::LDS) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        resident = per_cu * (cus < 1 ? 256 : cus);
    }
    const int slices = C / 64;
    int gx = resident / slices;
    if (gx < 1) gx = 1;
    if (gx > N) gx = N;
    hipLaunchKernelGGL((conv_ring1_kernel<W, R, DGRAD, POOL>), dim3(gx, slices), dim3(256), S::LDS, (hipStream_t)stream, (const bf16*)in, (const bf16*)wp,
                       (bf16*)out, (const bf16*)mask_raw, scale, shift, sum_a, sum_b, pool_scale, pool_shift, (bf16*)pool_out, pool_norm, N, H, C, sum_a ? stes_det_take(gx * slices) : StesDet{nullptr, 0u, 0u});
    STES_LAUNCH_CHECK();
    return STES_OK;
}

// ---------------------------------------------------------------------------------------------- weight gradient, stride 1
// dW[co][ci][a][b] += sum over pixels of dY[h][w][co] * x[h + a - 1][w + b - 1][ci] with x = ReLU(BatchNorm(raw input)): per
// group a 16 x 16 product per tap whose reduction runs over pixels, so both MFMA operands are wanted pixel-major while memory
// is channel-major -- the transposing LDS read (ds_read_tr16_b64: 16 lanes read 4 pixels x 16 channels, each lane leaves with
// 4 pixels of one channel) does that turn.  The tile kernel of stes_conv.hip stages a 4 x 8 pixel tile and its 6 x 10 halo per
// reduction step, so at stride 1 an input pixel is fetched 1.9 times (532 MB counted against 257 MB on the 14 x 14 maps).  Here
// the workgroup walks the image rows as the forward kernel does: input rows in the ring (each fetched once), the step's dY
// rows beside them, and a reduction block of 32 pixel slots (one row of a 28-wide map, two rows of 14, four rows of 7; unused
// slots point the dY operand at a zero pixel) takes 2 + 18 transposing reads and 9 MFMAs.  The 9 accumulators stay in registers
// over all the workgroup's images and leave through the atomics of the tile kernel.
template <int W, int R>
struct WgradRingShape {
    typedef Ring1Shape<W, R> In;                            // the input ring is the forward kernel's
    static constexpr int RPB = W > 16 ? 1 : W > 8 ? 2 : 4;  // rows per reduction block of 32 slots
    static constexpr int SPR = 32 / RPB;                    // slots per row
    static constexpr int NB = R / RPB;                      // reduction blocks per step
    static constexpr int YCH = R * W * 8, YPT = (YCH + 255) / 256;
    static constexpr int RING = In::NS * In::SLOT, STG = R * W * In::PXS;     // elements
    static constexpr size_t USED = (size_t)(RING + STG + In::PXS) * 2;       // + the zero pixel
    static constexpr size_t LDS = USED > 4608 * sizeof(float) ? USED : 4608 * sizeof(float);      // the result layout reuses it all
    static_assert(R % RPB == 0 && W <= SPR, "whole reduction blocks per step");
};

template <int W, int R>
__global__ __launch_bounds__(256) void wgrad_ring1_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ in,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ dw, int N, int H, int C, StesDet det) {
    typedef WgradRingShape<W, R> S;
    typedef Ring1Shape<W, R> I;
    extern __shared__ __attribute__((aligned(16))) unsigned char cr_smem[];
    bf16* ring = reinterpret_cast<bf16*>(cr_smem);                          // [NS][SLOT]
    bf16* stage = ring + S::RING;                                           // dY rows of the step: [R][W][PXS]
    bf16* zpix = stage + S::STG;                                            // a pixel of zeros
    float* red = reinterpret_cast<float*>(cr_smem);                         // [2 groups][16 co][16 ci][9] at the end
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int slice = blockIdx.y, cb = slice * 64, HT = (H + R - 1) / R;
    for (int i = tid; i < (S::RING + S::STG + I::PXS) / 2; i += 256) reinterpret_cast<unsigned*>(ring)[i] = 0u;
    const int chunk = tid & 7;
    const bool affine = scale != nullptr;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = affine ? scale[cb + chunk * 8 + e] : 1.f; sh[e] = affine ? shift[cb + chunk * 8 + e] : 0.f; }
    // this lane's two pixel slots of a reduction block (a transposing read takes slots 8 fq + fr / 4 and that + 4) and its four channels
    const int cc = wave * 16 + 4 * (fr & 3);
    int s_row[2], s_px[2];
    bool s_ok[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int slot = 8 * fq + (fr >> 2) + 4 * hh;
        s_row[hh] = slot / S::SPR;
        const int px = slot - s_row[hh] * S::SPR;
        s_ok[hh] = px < W;
        s_px[hh] = min(px, W - 1);
    }
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int nimg = blockIdx.x < N ? (N - 1 - blockIdx.x) / gridDim.x + 1 : 0;
    const int total = nimg * HT;
    // (every request is made by every thread on every step and unwanted results are dropped, so that the waits are counted
    // in instructions: see the stride-2 forward kernel)
    int in_n = blockIdx.x, in_t = 0;
    auto fetch_in = [&](bf16x8 (&dst)[I::IPT]) {
        const bf16* src = in + ((long)min(in_n, N - 1) * H + R * in_t) * W * C + cb + chunk * 8;
        const int last = (H - R * in_t) * W - 1;
#pragma unroll
        for (int k = 0; k < I::IPT; ++k) {
            const int i = k < I::ICH / 256 ? tid + 256 * k : min(tid + 256 * k, I::ICH - 1);
            const int p = i >> 3, c = p / W;                                // chunk row c: input row R t + c
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (long)min(p, last) * C);
            dst[k] = (c == 0 ? in_t == 0 : R * in_t + c < H) ? v : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
        if (++in_t == HT) { in_t = 0; in_n += gridDim.x; }
    };
    int y_n = blockIdx.x, y_t = 0;
    auto fetch_y = [&](bf16x8 (&dst)[S::YPT]) {
        const bf16* src = dout + ((long)min(y_n, N - 1) * H + R * y_t) * W * C + cb + chunk * 8;
        const int pix = min(R, H - R * y_t) * W;
#pragma unroll
        for (int k = 0; k < S::YPT; ++k) {
            const int i = k < S::YCH / 256 ? tid + 256 * k : min(tid + 256 * k, S::YCH - 1);
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (long)min(i >> 3, pix - 1) * C);
            dst[k] = (i >> 3) < pix ? v : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
        if (++y_t == HT) { y_t = 0; y_n += gridDim.x; }
    };
    int t = 0;
    auto step = [&](bf16x8 (&raw)[I::IPT], bf16x8 (&gy)[S::YPT], bool live) {
        const int h0 = R * t;
        if (t == 0) {                                                       // row -1 of an image: zeros
            for (int i = tid; i < W * 32; i += 256) reinterpret_cast<unsigned*>(ring + ((i >> 5) + 1) * I::PXS)[i & 31] = 0u;
        }
#pragma unroll
        for (int k = 0; k < I::IPT; ++k) {
            const int i = tid + 256 * k;
            if (i < I::ICH) {
                const int p = i >> 3, c = p / W, px = p - c * W;
                if (c > 0 || t == 0) {
                    bf16x8 v = raw[k];
                    if (affine && h0 + c < H) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (bf16)fmaxf((float)v[e] * sc[e] + sh[e], 0.f);
                    }
                    *reinterpret_cast<bf16x8*>(ring + ((h0 + c + 1) & (I::NS - 1)) * I::SLOT + (px + 1) * I::PXS + chunk * 8) = v;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < S::YPT; ++k) {
            const int i = tid + 256 * k;
            if (i < S::YCH) *reinterpret_cast<bf16x8*>(stage + (i >> 3) * I::PXS + chunk * 8) = live ? gy[k] : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
        fetch_in(raw);
        fetch_y(gy);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < S::NB; ++u) {
            if (h0 + u * S::RPB < H) {                                      // (rows below the image hold zeros in both operands anyway)
                const int r0 = u * S::RPB + s_row[0], r1 = u * S::RPB + s_row[1];
                const bf16* ya = s_ok[0] ? stage + (r0 * W + s_px[0]) * I::PXS + cc : zpix + (cc & 15);
                const bf16* yb = s_ok[1] ? stage + (r1 * W + s_px[1]) * I::PXS + cc : zpix + (cc & 15);
                const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)ya);
                const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)yb);
                const bf16x8 fa = __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const bf16* xa = ring + ((h0 + r0 + a) & (I::NS - 1)) * I::SLOT + s_px[0] * I::PXS + cc;
                    const bf16* xb = ring + ((h0 + r1 + a) & (I::NS - 1)) * I::SLOT + s_px[1] * I::PXS + cc;
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(xa + b * I::PXS));
                        const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(xb + b * I::PXS));
                        const bf16x8 fb = __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7);
                        acc[a * 3 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[a * 3 + b], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
        if (++t == HT) t = 0;
    };
    bf16x8 rawA[I::IPT], rawB[I::IPT], gyA[S::YPT], gyB[S::YPT];
    fetch_in(rawA);
    fetch_y(gyA);
    fetch_in(rawB);
    fetch_y(gyB);
#pragma unroll 1
    for (int s = 0; s < total; s += 2) {                                    // (an odd count ends with a step on zero dY rows)
        step(rawA, gyA, true);
        step(rawB, gyB, s + 1 < total);
    }
    // result -> LDS in the checkpoint layout [co][ci][tap] of this workgroup's 64 channels, two groups at a time, then
    // contiguous atomics (the tile kernel's way out)
    float* dst = dw + (long)slice * 9216;
    det_enter(det);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((wave >> 1) == h) {
#pragma unroll
            for (int tt = 0; tt < 9; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(((wave & 1) * 16 + fq * 4 + r) * 16 + fr) * 9 + tt] = acc[tt][r];
        }
        __syncthreads();
        for (int i = tid; i < 4608; i += 256) atomicAdd(dst + h * 4608 + i, red[i]);
        __syncthreads();
    }
    det_leave(det);
}

template <int W, int R>
int wgrad_ring1_launch(const uint16_t* dout, const uint16_t* in, const float* scale, const float* shift, float* dw, int N, int H, int C,
                       void* stream) {
    typedef WgradRingShape<W, R> S;
    static_assert(S::LDS <= 64 * 1024, "within the default dynamic LDS limit");
    static int resident = 0;
    if (!resident) {
        int per_cu = 0, dev = 0, cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgrad_ring1_kernel<W, R>, 256, S::LDS) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        resident = per_cu * (cus < 1 ? 256 : cus);
    }
    const int slices = C / 64;
    int gx = resident / slices;
    if (gx < 1) gx = 1;
    if (gx > N) gx = N;
    hipLaunchKernelGGL((wgrad_ring1_kernel<W, R>), dim3(gx, slices), dim3(256), S::LDS, (hipStream_t)stream, (const bf16*)dout, (const bf16*)in, scale,
                       shift, dw, N, H, C, stes_det_take(gx * slices));
    STES_LAUNCH_CHECK();
    return STES_OK;
}

template <int WI, bool POOL>
int ring_launch(const uint16_t* in, const uint16_t* wp, uint16_t* out, const float* scale, const float* shift, float* col_sum,
                float* col_sumsq, const float* pool_scale, const float* pool_shift, uint16_t* pool_out, float pool_norm, int N, int Hi, int C,
                void* stream) {
    typedef RingShape<WI> S;
    static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ring2_kernel<WI, POOL>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    STES_CHECK(big_lds, STES_ERR_UNSUPPORTED, "stes_conv3x3g: cannot raise the dynamic LDS limit");
    static int resident = 0;
    if (!resident) {
        int per_cu = 0, dev = 0, cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_ring2_kernel<WI, POOL>, 256, S::LDS) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        resident = per_cu * (cus < 1 ? 256 : cus);
    }
    const int slices = C / 64;
    int gx = resident / slices;
    if (gx < 1) gx = 1;
    if (gx > N) gx = N;
    const dim3 grid(gx, slices);
    hipLaunchKernelGGL((conv_ring2_kernel<WI, POOL>), grid, dim3(256), S::LDS, (hipStream_t)stream, (const bf16*)in, (const bf16*)wp, (bf16*)out, scale,
                       shift, col_sum, col_sumsq, pool_scale, pool_shift, (bf16*)pool_out, pool_norm, N, Hi, C, col_sum ? stes_det_take(gx * slices) : StesDet{nullptr, 0u, 0u});
    STES_LAUNCH_CHECK();
    return STES_OK;
}

}  // namespace

bool stes_conv_ring_takes(int mode, int stride, int Hi, int Wi, bool pooled) {
    static const int on = getenv("STES_CONV_RING") ? atoi(getenv("STES_CONV_RING")) : 1;
    (void)pooled;            // (with the squeeze-excite squeeze on the way out as well)
    return on && mode == 0 && stride == 2 && Hi % 2 == 0 && (Wi == 112 || Wi == 56 || Wi == 28 || Wi == 14);
}

bool stes_wgrad_ring_takes(int stride, int Wi) {
    // (alone on 1,024 images: 28 x 28 x 128 101 vs 126 us, 14 x 14 x 320 64 vs 85, 7 x 7 x 768 52 vs 61; the step with the weight
    // gradients in program order 35.05 vs 35.19 ms, with them on the side stream unchanged -- what it saves there is traffic)
    static const int on = getenv("STES_CONV_RING") ? atoi(getenv("STES_CONV_RING")) : 1;
    return on && stride == 1 && (Wi == 14 || Wi == 28 || Wi == 7);
}

int stes_wgrad_ring_launch(const uint16_t* dout, const uint16_t* in, const float* scale, const float* shift, float* dw, int N, int H, int W,
                           int C, void* stream) {
    if (W == 28) return wgrad_ring1_launch<28, 2>(dout, in, scale, shift, dw, N, H, C, stream);
    if (W == 14) return wgrad_ring1_launch<14, 8>(dout, in, scale, shift, dw, N, H, C, stream);       // (64 us; 70 at four rows a step, 74 at two)
    if (W == 7) return wgrad_ring1_launch<7, 8>(dout, in, scale, shift, dw, N, H, C, stream);
    return stes_set_error(STES_ERR_UNSUPPORTED, "stes_wgrad_ring: map width");
}

bool stes_conv_ring1_takes(int mode, int stride, int Wi, bool pooled) {
    static const int on = getenv("STES_CONV_RING") ? atoi(getenv("STES_CONV_RING")) : 1;
    // (measured on 1,024 images against the window and column-strip kernels, forward / masked gradient, the kernel alone with its
    // operands warm in the Infinity Cache: 7 x 7 x 768 48 vs 67 us and 51 vs 96; 28 x 28 x 128 107 vs 144 and 139 vs 175;
    // 14 x 14 x 320 74 vs 73 and 93 vs 117 -- inside the training step, operands from memory, the 14-wide forward is 100 us with the
    // column-strip kernel and the step 0.13 ms shorter with this one.  With a request under a condition anywhere in the step all
    // of these were slower than the kernels they replace: the waits then cover the rows asked for a moment ago.)
    return on && stride == 1 && (!pooled || mode == 0) && (Wi == 7 || Wi == 14 || Wi == 28) && (mode == 0 || mode == 1);
}

int stes_conv_ring1_launch(int mode, const uint16_t* in, const uint16_t* wp, uint16_t* out, const uint16_t* mask_raw, const float* scale,
                           const float* shift, float* sum_a, float* sum_b, const float* pool_scale, const float* pool_shift,
                           uint16_t* pool_out, float pool_norm, int N, int H, int W, int C, void* stream) {
#define RING1(WW, RR)                                                                                                                  \
    if (W == WW && mode == 0 && pool_out)                                                                                              \
        return ring1_launch<WW, RR, false, true>(in, wp, out, nullptr, scale, shift, nullptr, nullptr, N, H, C, stream, pool_scale, pool_shift, \
                                                 pool_out, pool_norm);                                                                 \
    if (W == WW) return mode == 0 ? ring1_launch<WW, RR, false>(in, wp, out, nullptr, scale, shift, sum_a, sum_b, N, H, C, stream)     \
                                  : ring1_launch<WW, RR, true>(in, wp, out, mask_raw, scale, shift, sum_a, sum_b, N, H, C, stream)
    // (14 wide, rows per step: forward 68 us at eight, 72 at four, 85 at two; masked gradient 100 / 93 / 110)
    RING1(7, 8);
    if (mode == 0) { RING1(14, 8); }
    RING1(14, 4);
    RING1(28, 2);
#undef RING1
    return stes_set_error(STES_ERR_UNSUPPORTED, "stes_conv_ring1: map width");
}

bool stes_dgrad_ring_takes(int mode, int stride, int Hi, int Wi) {
    static const int on = getenv("STES_CONV_RING") ? atoi(getenv("STES_CONV_RING")) : 1;
    return on && mode == 1 && stride == 2 && Hi % 2 == 0 && (Wi == 112 || Wi == 56 || Wi == 28 || Wi == 14);
}

int stes_dgrad_ring_launch(const uint16_t* dy, const uint16_t* wp, uint16_t* dx, const uint16_t* mask_raw, const float* mscale,
                           const float* mshift, float* s_dy, float* s_dyr, int N, int Hi, int Wi, int C, void* stream) {
    if (Wi == 112) return dgrad_launch<112>(dy, wp, dx, mask_raw, mscale, mshift, s_dy, s_dyr, N, Hi, C, stream);
    if (Wi == 56) return dgrad_launch<56>(dy, wp, dx, mask_raw, mscale, mshift, s_dy, s_dyr, N, Hi, C, stream);
    if (Wi == 28) return dgrad_launch<28>(dy, wp, dx, mask_raw, mscale, mshift, s_dy, s_dyr, N, Hi, C, stream);
    if (Wi == 14) return dgrad_launch<14>(dy, wp, dx, mask_raw, mscale, mshift, s_dy, s_dyr, N, Hi, C, stream);
    return stes_set_error(STES_ERR_UNSUPPORTED, "stes_dgrad_ring: map width");
}

int stes_conv_ring_launch(const uint16_t* in, const uint16_t* wp, uint16_t* out, const float* scale, const float* shift, float* col_sum,
                          float* col_sumsq, const float* pool_scale, const float* pool_shift, uint16_t* pool_out, float pool_norm, int N,
                          int Hi, int Wi, int C, void* stream) {
#define RING2(WW)                                                                                                                           \
    if (Wi == WW)                                                                                                                           \
        return pool_out ? ring_launch<WW, true>(in, wp, out, scale, shift, nullptr, nullptr, pool_scale, pool_shift, pool_out, pool_norm, N, Hi, C, stream) \
                        : ring_launch<WW, false>(in, wp, out, scale, shift, col_sum, col_sumsq, nullptr, nullptr, nullptr, 0.f, N, Hi, C, stream)
    RING2(112);
    RING2(56);
    RING2(28);
    RING2(14);
#undef RING2
    return stes_set_error(STES_ERR_UNSUPPORTED, "stes_conv_ring: map width");
}
