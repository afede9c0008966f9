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
#include <mutex>
#include "stes_common.h"
#include "stes_conv_strip.h"
#include "stes_conv_ring.h"

namespace {

// ------------------------------------------------------------------ weight packing
// w [G*16][16][3][3] fp32 ->  fwd [G][5][64][8]: lane(i=co,q): tap=2s+(q>>1), ci=(q&1)*8+j
//                             dgr [G][5][64][8]: lane(i=ci,q): tap=2s+(q>>1), co=(q&1)*8+j
// stride-2 input gradient: taps grouped by output-pixel parity class (row parity, column parity); k-step s pairs the taps
//   s=0 (even,even): (1,1), -      s=1 (even,odd): (1,0),(1,2)      s=2 (odd,even): (0,1),(2,1)
//   s=3,4 (odd,odd): (0,0),(0,2) and (2,0),(2,2)           (tap index kh*3+kw; the first of a pair reads source index + 1)
__constant__ int kParityTaps[5][2] = {{4, -1}, {3, 5}, {1, 7}, {0, 2}, {6, 8}};

__global__ void pack_conv3x3g_kernel(const float* __restrict__ w, bf16* __restrict__ fwd, bf16* __restrict__ dgr, int G, int stride) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = G * 5 * 64 * 8;
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, s = (idx >> 9) % 5, g = idx / (5 * 512);
    const int i = lane & 15, q = lane >> 4;
    const int tap = 2 * s + (q >> 1), c = (q & 1) * 8 + j;
    const int tap_d = stride == 2 ? kParityTaps[s][q >> 1] : (tap < 9 ? tap : -1);
    float vf = 0.f, vd = 0.f;
    if (tap < 9) vf = w[((long)(g * 16 + i) * 16 + c) * 9 + tap];
    if (tap_d >= 0) vd = w[((long)(g * 16 + c) * 16 + i) * 9 + tap_d];
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
                                                       ConvGeom gm, int n_tiles, StesDet det) {
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
        det_enter(det);
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6, c = blockIdx.y * 64 + (threadIdx.x & 63);
            if (c < gm.C) atomicAdd((which ? col_sumsq : col_sum) + c, red[which][threadIdx.x & 63]);
        }
        det_leave(det);
    }
}

// ------------------------------------------------------------------ forward / stride-1 input gradient, LDS row window
// Workgroup = (image, 64-channel block = 4 groups, one per wave).  It walks down the image keeping the input rows of the
// current step in an LDS window (pixel stride 160 bytes: 128 of data + 32 of padding, which spreads the fragment reads
// over the banks), so every input element is fetched from memory once and BatchNorm + ReLU is applied once, at the store
// into LDS.  The window always starts at slot 0: the one or two rows a step shares with the next one are carried over
// through registers, which makes every operand address "per-lane constant of the fragment + per-lane constant of the
// k-step" (one add per read; with a ring the row wrap cost more vector instructions than the convolution itself).
// The next step's rows travel global -> registers while the current rows are convolved.  Results are staged in LDS and
// leave as whole 128-byte pixel rows; a thread always stores the same 8 channels, so the output statistics take 16
// registers.
struct RowsGeom {
    int N, Hs, Ws, Hd, Wd, C;
    int RS, NR;                        // output rows per step, window rows (RS * stride + overlap)
    FastDiv fd_wd, fd_px8;             // Wd; 8 * Ws (16-byte chunks per input row of the channel block)
};
constexpr int PSB = 160;               // LDS bytes per pixel
constexpr int CONV_MAXI = 7;           // staged 16-byte chunks per thread per step
constexpr int CONV_OVI = 4;            // carried-over 16-byte chunks per thread
constexpr int CONV_NF = 3;             // 16-pixel fragments in flight per wave
constexpr int CONV_NG = 5;             // fragment groups per step: at most 15 fragments = 240 pixels
constexpr int CONV_MAXP = 196;         // output pixels per step (host side keeps RS * Wd below this)
static_assert(CONV_NF * CONV_NG * 16 >= CONV_MAXP, "fragment groups must cover a step");

// POOL (forward, evaluation): the squeeze of the squeeze-excite block that follows, pool_out[n][c] = pool_norm * sum over the
// image of relu(pool_scale[c]*y + pool_shift[c]) on the rounded outputs y (the BatchNorm over y is a fixed affine map there),
// so the separate pooling pass does not read y again.  A workgroup owns whole images of its 64 channels: no atomics, and the
// sum is taken in a fixed order (per thread over its pixels, lanes by xor butterfly, the four waves in order).
template <int STRIDE, bool FLIP, bool POOL = false>
__global__ __launch_bounds__(256, 2) void conv3x3g_rows_kernel(const bf16* __restrict__ src, const bf16* __restrict__ wp,
                                                               bf16* __restrict__ dst, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, float* col_sum, float* col_sumsq,
                                                               const bf16* __restrict__ mask_raw, RowsGeom gm,
                                                               const float* __restrict__ pool_scale, const float* __restrict__ pool_shift,
                                                               bf16* __restrict__ pool_out, float pool_norm, StesDet det) {
    // FLIP with mask_raw: the result is the gradient at a ReLU(BatchNorm(raw)) output; it is masked with
    // [scale*raw + shift > 0] on its way out and the BatchNorm-backward sums (sum dy -> col_sum, sum dy*raw -> col_sumsq)
    // are accumulated here, which saves the separate reduction pass over both tensors.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, q = lane >> 4;
    const int ncb = gm.C >> 6;
    const int cb = blockIdx.x % ncb, n_first = blockIdx.x / ncb, n_step = gridDim.x / ncb;   // gridDim.x is a multiple of ncb
    const int g = cb * 4 + wave;
    const int Hs = gm.Hs, Ws = gm.Ws, Hd = gm.Hd, Wd = gm.Wd, C = gm.C, RS = gm.RS, NR = gm.NR;
    const int ROWB = (Ws + 2) * PSB;
    constexpr int OV = STRIDE == 1 ? 2 : 1;           // window rows shared with the next step
    constexpr int NEW = STRIDE;                       // new input rows per output row
    unsigned char* ring = smem;
    const unsigned ring_lds = (unsigned)(size_t)smem; // LDS byte address of the window
    unsigned char* stage = smem + NR * ROWB;          // [RS*Wd][128 B]
    // (at least 1 KiB of staging: the per-image pooled sums of the four waves pass through it, and a 2 x 2 map stages only 512 bytes)
    float* red = reinterpret_cast<float*>(stage + (RS * Wd * 128 < 1024 ? 1024 : RS * Wd * 128));   // [2][64]

    bf16x8 wf[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) wf[s] = *reinterpret_cast<const bf16x8*>(wp + (((long)g * 5 + s) * 64 + lane) * 8);
    const bool masked = FLIP && mask_raw != nullptr;
    const bool affine = !FLIP && scale != nullptr;
    const bool stats = col_sum != nullptr;
    const int chunk = tid & 7;                        // this thread's 8 channels in the staging / store phases
    float sc[8], sh[8];
    if (affine || masked) { load8f(scale + cb * 64 + chunk * 8, sc); load8f(shift + cb * 64 + chunk * 8, sh); }
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }

    // Operand addresses.  Output pixel (orow, ocol) of the step under tap (kh, kw) reads window row orow*STRIDE + kh and
    // pixel column ocol*STRIDE + kw (column 0 is the left halo); the flipped (input gradient) form reads row orow + 2 - kh
    // and column ocol + 2 - kw.  fbase: fragment f, this lane's pixel; tapc: k-step s, this lane's tap 2s + (q >> 1).  The
    // tenth tap does not exist: its weights are zero and it re-reads the ninth tap's operand.
    const int P_full = RS * Wd;
    int* fbase = reinterpret_cast<int*>(red + 128);   // [fragments][16 pixels]: a table, not registers (the kernel is short of them)
    for (int i = tid; i < CONV_NF * CONV_NG * 16; i += 256) {
        int pi = i;
        if (pi >= P_full) pi = P_full - 1;
        uint32_t orow, ocol;
        fdivmod((uint32_t)pi, gm.fd_wd, orow, ocol);
        fbase[i] = (int)orow * STRIDE * ROWB + (int)ocol * STRIDE * PSB;
    }
    int tapc[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int t0 = 2 * s, t1 = 2 * s + 1 < 9 ? 2 * s + 1 : 8;
        const int r0 = FLIP ? 2 - t0 / 3 : t0 / 3, c0 = FLIP ? 2 - t0 % 3 : t0 % 3;
        const int r1 = FLIP ? 2 - t1 / 3 : t1 / 3, c1 = FLIP ? 2 - t1 % 3 : t1 % 3;
        tapc[s] = (q < 2 ? r0 * ROWB + c0 * PSB : r1 * ROWB + c1 * PSB) + wave * 32 + (q & 1) * 16;
    }

    // zero the left / right halo columns of every window row once; they are never written again
    for (int i = tid; i < NR * 2 * 8; i += 256) {
        const int r = i >> 4, side = (i >> 3) & 1, ch = i & 7;
        *reinterpret_cast<bf16x8*>(ring + r * ROWB + (side ? (Ws + 1) * PSB : 0) + ch * 16) = zero8();
    }
    if constexpr (POOL) {                                // red holds the pooling affine instead of statistics
        // (each word written by one thread only: zeroing all of red first would race with these stores across waves)
        if (tid < 128) red[tid] = tid < 64 ? pool_scale[cb * 64 + tid] : pool_shift[cb * 64 + tid - 64];
    } else if (tid < 128) red[tid] = 0.f;

    const bf16* img = nullptr;                           // this image, this channel block (uniform)
    bf16x8 regs[CONV_MAXI];
    // Item k of this thread in the staging phases: 16-byte chunk `chunk` of pixel (tid >> 3) + 32k of the rows being moved,
    // kept as (row << 16 | column); the division happens once here.  The packed value passes through an empty asm at
    // every use: the arithmetic on it is then redone there (a few full-rate instructions) instead of its results being
    // kept in registers across the convolution, which is short of them.
    int item[CONV_MAXI];
#pragma unroll
    for (int k = 0; k < CONV_MAXI; ++k) {
        const uint32_t pxi = (uint32_t)(tid >> 3) + 32u * k;
        const uint32_t r = fdiv(pxi * 8u, gm.fd_px8);
        item[k] = (int)((r << 16) | (pxi - r * (uint32_t)Ws));
    }
    // rows [first, first + count) of the input -> registers; rows outside the image are zero rows
    auto fetch = [&](int first, int count) {
#pragma unroll
        for (int k = 0; k < CONV_MAXI; ++k) {
            int e = item[k];
            asm volatile("" : "+v"(e));
            const int r = e >> 16, px = e & 0xffff, ir = first + r;
            regs[k] = zero8();
            if (r < count && ir >= 0 && ir < Hs)
                regs[k] = *reinterpret_cast<const bf16x8*>(img + (__mul24(__mul24(ir, Ws) + px, C) + chunk * 8));
        }
    };
    // registers -> window rows [slot, slot + count); `first` is the input row of the first one
    auto commit = [&](int first, int count, int slot) {
#pragma unroll
        for (int k = 0; k < CONV_MAXI; ++k) {
            int e = item[k];
            asm volatile("" : "+v"(e));
            const int r = e >> 16, px = e & 0xffff, ir = first + r;
            if (r < count) {
                bf16x8 v = regs[k];
                if (affine && ir >= 0 && ir < Hs) {
                    float f[8];
                    unpack8(v, f);
#pragma unroll
                    for (int e2 = 0; e2 < 8; ++e2) f[e2] = fmaxf(f[e2] * sc[e2] + sh[e2], 0.f);
                    v = pack8(f);
                }
                *reinterpret_cast<bf16x8*>(ring + __mul24(slot + r, ROWB) + __mul24(px + 1, PSB) + chunk * 16) = v;
            }
        }
    };
    // the last OV window rows -> registers -> the first OV rows (the rows the next step shares with this one)
    bf16x8 carry[CONV_OVI];
    auto carry_move = [&](bool load) {
#pragma unroll
        for (int k = 0; k < CONV_OVI; ++k) {
            int e = item[k];
            asm volatile("" : "+v"(e));
            const int r = e >> 16, px = e & 0xffff;
            if (r < OV) {
                unsigned char* at = ring + __mul24(px + 1, PSB) + chunk * 16;
                if (load) carry[k] = *reinterpret_cast<const bf16x8*>(at + __mul24(NR - OV + r, ROWB));
                else *reinterpret_cast<bf16x8*>(at + __mul24(r, ROWB)) = carry[k];
            }
        }
    };

    // Small images: the whole first window fits the staging registers, and then the next image's first window is requested
    // while this image's last step is convolved (otherwise every image of a one-step geometry would wait out a memory
    // round trip with nothing to do).
    const bool merged = NR * Ws * 8 <= CONV_MAXI * 256;
    bool have = false;                                   // the registers already hold this image's first window
    const long img_off = (long)cb * 64;
    const int nsteps = (Hd + RS - 1) / RS;
    for (int n = n_first; n < gm.N; n += n_step) {       // images of this workgroup; statistics stay in registers
    __syncthreads();                                     // the previous image's window and staging are no longer read
    if (merged) {
        if (!have) { img = src + (long)n * Hs * Ws * C + img_off; fetch(-1, NR); }
        commit(-1, NR, 0);
        have = false;
    } else {
        img = src + (long)n * Hs * Ws * C + img_off;
        fetch(-1, OV);
        commit(-1, OV, 0);
        fetch(OV - 1, RS * NEW);
    }
    for (int st = 0; st < nsteps; ++st) {
        const int first_new = OV - 1 + st * RS * NEW;
        if (st > 0) carry_move(false);
        if (!(merged && st == 0)) commit(first_new, RS * NEW, OV);
        __syncthreads();
        if (st + 1 < nsteps) fetch(first_new + RS * NEW, RS * NEW);
        else if (merged && n + n_step < gm.N) { img = src + (long)(n + n_step) * Hs * Ws * C + img_off; fetch(-1, NR); have = true; }
        const int row0 = st * RS;
        const int rows_here = (row0 + RS <= Hd) ? RS : Hd - row0;
        const int P = rows_here * Wd;
        const int nfrag = (P + 15) >> 4;
        const long obase = ((long)(n * Hd + row0) * Wd) * C + cb * 64 + chunk * 8;
        // the raw values under the mask, asked for before the convolution that hides their latency
        bf16x8 mraw[CONV_MAXI];
        if (masked) {
            int t = tid;
            asm volatile("" : "+v"(t));
#pragma unroll
            for (int k = 0; k < CONV_MAXI; ++k) {
                const int px = (t >> 3) + k * 32;
                mraw[k] = zero8();
                if (px < P) mraw[k] = *reinterpret_cast<const bf16x8*>(mask_raw + obase + (long)px * C);
            }
        }
        // Fragments of 16 pixels, CONV_NF at a time: all their operand reads are issued before the first MFMA and the
        // accumulation chains are independent.  The reads go through asm with counted waits: left to itself the compiler
        // interleaves read, wait and MFMA, and the loop runs at one LDS round trip per k-step.  (A second register set that
        // keeps the next group's reads in flight during the MFMAs measured no faster and costs 60 registers.)
#pragma unroll 1
        for (int gi = 0; gi < CONV_NG; ++gi) {
            if (gi * CONV_NF < nfrag) {
                bf16x8 frag[5][CONV_NF];
                unsigned fb[CONV_NF];
#pragma unroll
                for (int u = 0; u < CONV_NF; ++u) fb[u] = ring_lds + (unsigned)fbase[(gi * CONV_NF + u) * 16 + p];
#pragma unroll
                for (int s = 0; s < 5; ++s)
#pragma unroll
                    for (int u = 0; u < CONV_NF; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(frag[s][u]) : "v"(fb[u] + (unsigned)tapc[s]));
                f32x4 acc[CONV_NF];
#pragma unroll
                for (int u = 0; u < CONV_NF; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                static_assert(CONV_NF == 3, "the counted waits below release three reads each");
#define CONV_KSTEP(S, CNT)                                                                                                    \
                asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(frag[S][0]), "+v"(frag[S][1]), "+v"(frag[S][2]));            \
                _Pragma("unroll") for (int u = 0; u < CONV_NF; ++u)                                                           \
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[S], frag[S][u], acc[u], 0, 0, 0);                     \
                __builtin_amdgcn_sched_barrier(0);
                CONV_KSTEP(0, 12) CONV_KSTEP(1, 9) CONV_KSTEP(2, 6) CONV_KSTEP(3, 3) CONV_KSTEP(4, 0)
#undef CONV_KSTEP
#pragma unroll
                for (int u = 0; u < CONV_NF; ++u) {
                    const int pi = (gi * CONV_NF + u) * 16 + p;
                    if (pi < P) {
                        bf16x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[u][r];
                        *reinterpret_cast<bf16x4*>(stage + pi * 128 + wave * 32 + q * 8) = o;
                    }
                }
            }
        }
        if (st + 1 < nsteps) carry_move(true);
        __syncthreads();
        // staged pixels -> global, 128 contiguous bytes per pixel (fully contiguous when C == 64)
        bf16* out = dst + obase;
        int ts = tid;
        asm volatile("" : "+v"(ts));
#pragma unroll
        for (int k = 0; k < CONV_MAXI; ++k) {
            const int px = (ts >> 3) + k * 32;
            if (px >= P) break;
            bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + px * 128 + chunk * 16);
            if (masked) {
                const bf16x8 raw = mraw[k];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float r = (float)raw[e];
                    if (!(r * sc[e] + sh[e] > 0.f)) v[e] = (bf16)0.f;
                    const float fv = (float)v[e];
                    ssum[e] += fv; ssq[e] += fv * r;
                }
                *reinterpret_cast<bf16x8*>(out + (long)px * C) = v;
                continue;
            }
            *reinterpret_cast<bf16x8*>(out + (long)px * C) = v;
            if constexpr (POOL) {
                float a[8], b[8];
                load8f(red + chunk * 8, a); load8f(red + 64 + chunk * 8, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) ssum[e] += fmaxf((float)v[e] * a[e] + b[e], 0.f);
            } else if (stats) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float fv = (float)v[e]; ssum[e] += fv; ssq[e] += fv * fv; }
            }
        }
    }
    if constexpr (POOL) {                                // this image's pooled values
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = ssum[e];
            t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
            ssum[e] = t;                                 // lanes with the same 8 channels hold the wave's sum
        }
        __syncthreads();                                 // the store phase no longer reads the staging buffer
        float* ps = reinterpret_cast<float*>(stage);     // [4 waves][64 channels]
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ps[wave * 64 + lane * 8 + e] = ssum[e];
        }
        __syncthreads();
        if (tid < 64) pool_out[(long)n * C + cb * 64 + tid] = (bf16)((((ps[tid] + ps[64 + tid]) + ps[128 + tid]) + ps[192 + tid]) * pool_norm);
#pragma unroll
        for (int e = 0; e < 8; ++e) ssum[e] = 0.f;
    }
    }
    if (stats) {
        STES_DET_LDS(det, _Pragma("unroll") for (int e = 0; e < 8; ++e) { atomicAdd(&red[chunk * 8 + e], ssum[e]); atomicAdd(&red[64 + chunk * 8 + e], ssq[e]); });
        __syncthreads();
        det_enter(det);
        if (tid < 128) atomicAdd((tid < 64 ? col_sum : col_sumsq) + cb * 64 + (tid & 63), red[tid]);
        det_leave(det);
    }
}

// ------------------------------------------------------------------ stride-2 input gradient, LDS row ring
// dX[h,w] gathers dY[(h+1-kh)/2, (w+1-kw)/2] over the taps whose parities match, so output pixels are processed in
// fragments of one parity class: 1, 1, 1 and 2 MFMA k-steps per 16 pixels instead of 5 with three quarters of the operand
// zero.  dY rows y0..y0+R (one row of lookahead) sit in the LDS ring and produce output rows 2*y0 .. 2*(y0+R)-1.
struct Dgrad2Geom {
    int N, Ho, Wo, Hi, Wi, C;          // dY is [Ho][Wo], dX is [Hi][Wi]
    int R;                             // dY rows per step
    FastDiv fd_px8;                    // 8 * Wo
};

template <bool BATCH>
__global__ __launch_bounds__(256, 2) void conv3x3g_dgrad2_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ wp,
                                                                 bf16* __restrict__ dx, const bf16* __restrict__ mask_raw,
                                                                 const float* __restrict__ mscale, const float* __restrict__ mshift,
                                                                 float* s_dy, float* s_dyr, Dgrad2Geom gm, StesDet det) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, q = lane >> 4;
    const int ncb = gm.C >> 6;
    const int cb = blockIdx.x % ncb, n_first = blockIdx.x / ncb, n_step = gridDim.x / ncb;
    const int g = cb * 4 + wave;
    const int Ho = gm.Ho, Wo = gm.Wo, Hi = gm.Hi, Wi = gm.Wi, C = gm.C, R = gm.R, NR = R + 1;
    const int ROWB = (Wo + 2) * PSB;
    unsigned char* ring = smem;
    unsigned char* stage = smem + NR * ROWB;          // [2R*Wi][128 B]
    float* red = reinterpret_cast<float*>(stage + 2 * R * Wi * 128);   // [2][64]
    const bool masked = mask_raw != nullptr;          // see conv3x3g_rows_kernel
    float sc[8], sh[8], ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; sc[e] = 0.f; sh[e] = 0.f; }
    if (masked) { load8f(mscale + cb * 64 + (tid & 7) * 8, sc); load8f(mshift + cb * 64 + (tid & 7) * 8, sh); }
    if (tid < 128) red[tid] = 0.f;
    bf16x8 wf[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) wf[s] = *reinterpret_cast<const bf16x8*>(wp + (((long)g * 5 + s) * 64 + lane) * 8);
    const int chunk = tid & 7;
    for (int i = tid; i < NR * 2 * 8; i += 256) {     // halo columns stay zero
        const int r = i >> 4, side = (i >> 3) & 1, ch = i & 7;
        *reinterpret_cast<bf16x8*>(ring + r * ROWB + (side ? (Wo + 1) * PSB : 0) + ch * 16) = zero8();
    }
    const bf16* img = nullptr;                           // this image, this channel block (uniform)
    bf16x8 regs[CONV_MAXI];
    // staging items as in conv3x3g_rows_kernel: (row << 16 | column) of pixel (tid >> 3) + 32k, divided once
    int item[CONV_MAXI];
#pragma unroll
    for (int k = 0; k < CONV_MAXI; ++k) {
        const uint32_t pxi = (uint32_t)(tid >> 3) + 32u * k;
        const uint32_t r = fdiv(pxi * 8u, gm.fd_px8);
        item[k] = (int)((r << 16) | (pxi - r * (uint32_t)Wo));
    }
    auto fetch = [&](int first, int count) {
#pragma unroll
        for (int k = 0; k < CONV_MAXI; ++k) {
            int e = item[k];
            asm volatile("" : "+v"(e));
            const int r = e >> 16, px = e & 0xffff, ir = first + r;
            regs[k] = zero8();
            if (r < count && ir < Ho) regs[k] = *reinterpret_cast<const bf16x8*>(img + (__mul24(__mul24(ir, Wo) + px, C) + chunk * 8));
        }
    };
    auto commit = [&](int first, int count) {
        const int slot0 = first % NR;                    // uniform; count <= NR, so one conditional subtraction wraps
#pragma unroll
        for (int k = 0; k < CONV_MAXI; ++k) {
            int e = item[k];
            asm volatile("" : "+v"(e));
            const int r = e >> 16, px = e & 0xffff;
            if (r < count) {
                int sl = slot0 + r;
                sl -= sl >= NR ? NR : 0;
                *reinterpret_cast<bf16x8*>(ring + __mul24(sl, ROWB) + __mul24(px + 1, PSB) + chunk * 16) = regs[k];
            }
        }
    };
    const int ne = (Wi + 1) >> 1, no = Wi >> 1;       // even / odd output columns per row
    const int fe = (ne + 15) >> 4, fo = (no + 15) >> 4;
    const int nsteps = (Ho + R - 1) / R;
    // (as in conv3x3g_rows_kernel: when row 0 and the first step's rows fit the staging registers together, the next image's
    // first rows are requested while this image's last step is computed)
    const bool merged = (1 + R) * Wo * 8 <= CONV_MAXI * 256;
    bool have = false;
    const long img_off = (long)cb * 64;
    const int hi_half = q >> 1;                           // second tap of the pair
    const int lanec = wave * 32 + (q & 1) * 16;
    const bool pre = 2 * R * Wi <= CONV_MAXI * 32;        // a step's output pixels fit the mask prefetch registers
    for (int n = n_first; n < gm.N; n += n_step) {
        __syncthreads();
        if (merged) {
            if (!have) { img = dy + (long)n * Ho * Wo * C + img_off; fetch(0, 1 + R); }
            commit(0, 1 + R);
            have = false;
        } else {
            img = dy + (long)n * Ho * Wo * C + img_off;
            fetch(0, 1);
            commit(0, 1);
            fetch(1, R);
        }
        for (int st = 0; st < nsteps; ++st) {
            const int y0 = st * R;
            if (!(merged && st == 0)) commit(y0 + 1, R);
            __syncthreads();
            if (st + 1 < nsteps) fetch(y0 + R + 1, R);
            else if (merged && n + n_step < gm.N) { img = dy + (long)(n + n_step) * Ho * Wo * C + img_off; fetch(0, 1 + R); have = true; }
            int out_rows = 2 * R;
            if (2 * y0 + out_rows > Hi) out_rows = Hi - 2 * y0;
            const int P = out_rows * Wi;
            const long obase = ((long)(n * Hi + 2 * y0) * Wi) * C + cb * 64 + chunk * 8;
            // the raw values under the mask, asked for before the convolution that hides their latency
            bf16x8 mraw[CONV_MAXI];
            if (masked && pre) {
                int t = tid;
                asm volatile("" : "+v"(t));
#pragma unroll
                for (int k = 0; k < CONV_MAXI; ++k) {
                    const int px = (t >> 3) + k * 32;
                    mraw[k] = zero8();
                    if (px < P) mraw[k] = *reinterpret_cast<const bf16x8*>(mask_raw + obase + (long)px * C);
                }
            }
            const int ybase = y0 % NR;
            if constexpr (BATCH) {
            // Work items (output row rr, column parity b, fragment f) in that order, four at a time: the operand reads of a
            // group are all issued (through asm, see conv3x3g_rows_kernel) before its first MFMA; every item issues two reads
            // (the second repeats the first where only one is used).  The tap geometry is that of the plain loop below.
            const unsigned ring_lds = (unsigned)(size_t)smem;
            int rr = 0, bb = 0, ff = 0;
            while (rr < out_rows) {
                bf16x8 fA[4], fB[4];
                int it_w[4], it_st[4], it_j[4];    // weight set a*2+b (-1: no item), staging pixel, column index
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    it_w[u] = -1;
                    if (rr < out_rows) {
                        const int a = rr & 1;
                        int sy = ybase + (rr >> 1), sy1 = sy + 1;            // ring slots of dY rows y and y + 1
                        sy -= sy >= NR ? NR : 0;
                        sy1 -= sy1 >= NR ? NR : 0;
                        const int npx = bb ? no : ne;
                        int j = ff * 16 + p;
                        it_j[u] = j;
                        if (j >= npx) j = npx - 1;
                        const unsigned colb = ring_lds + (unsigned)(__mul24(j + 1, PSB) + lanec + (bb && !hi_half ? PSB : 0));
                        const unsigned rowA = (unsigned)__mul24(a ? ((bb == 0 && hi_half) ? sy : sy1) : sy, ROWB);
                        const unsigned rowB = (unsigned)__mul24(sy, ROWB);
                        asm volatile("ds_read_b128 %0, %1" : "=v"(fA[u]) : "v"(colb + rowA));
                        asm volatile("ds_read_b128 %0, %1" : "=v"(fB[u]) : "v"(colb + rowB));
                        it_w[u] = a * 2 + bb;
                        it_st[u] = rr * Wi + bb;
                        if (++ff >= (bb ? fo : fe)) { ff = 0; if (++bb == 2) { bb = 0; ++rr; } }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fA[0]), "+v"(fA[1]), "+v"(fA[2]), "+v"(fA[3]), "+v"(fB[0]), "+v"(fB[1]), "+v"(fB[2]), "+v"(fB[3]));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (it_w[u] >= 0) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        if (it_w[u] == 0) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], fA[u], acc, 0, 0, 0);
                        else if (it_w[u] == 1) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], fA[u], acc, 0, 0, 0);
                        else if (it_w[u] == 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2], fA[u], acc, 0, 0, 0);
                        else {
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[3], fA[u], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[4], fB[u], acc, 0, 0, 0);
                        }
                        const int npx = (it_w[u] & 1) ? no : ne;
                        if (it_j[u] < npx) {
                            bf16x4 o;
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[r];
                            *reinterpret_cast<bf16x4*>(stage + (it_st[u] + 2 * it_j[u]) * 128 + wave * 32 + q * 8) = o;
                        }
                    }
                }
            }
            } else
            for (int rr = 0; rr < out_rows; ++rr) {
                const int a = rr & 1;
                int sy = ybase + (rr >> 1), sy1 = sy + 1;                 // ring slots of dY rows y and y + 1
                sy -= sy >= NR ? NR : 0;
                sy1 -= sy1 >= NR ? NR : 0;
                const int slot_y = sy * ROWB, slot_y1 = sy1 * ROWB;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int npx = b ? no : ne, nf = b ? fo : fe;
                    for (int f = 0; f < nf; ++f) {
                        int j = f * 16 + p;
                        const bool valid = j < npx;
                        if (!valid) j = npx - 1;
                        const int colb = (j + 1) * PSB + lanec;                       // source column j; + PSB for j + 1
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        if (a == 0 && b == 0) {
                            bf16x8 frag = zero8();
                            if (!hi_half) frag = *reinterpret_cast<const bf16x8*>(ring + slot_y + colb);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], frag, acc, 0, 0, 0);
                        } else if (a == 0) {          // (1,0): column j + 1, (1,2): column j
                            const bf16x8 frag = *reinterpret_cast<const bf16x8*>(ring + slot_y + colb + (hi_half ? 0 : PSB));
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], frag, acc, 0, 0, 0);
                        } else if (b == 0) {          // (0,1): row y + 1, (2,1): row y
                            const bf16x8 frag = *reinterpret_cast<const bf16x8*>(ring + (hi_half ? slot_y : slot_y1) + colb);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2], frag, acc, 0, 0, 0);
                        } else {                      // (0,0),(0,2): row y + 1; (2,0),(2,2): row y; first of a pair column j + 1
                            const int cc = colb + (hi_half ? 0 : PSB);
                            const bf16x8 f0 = *reinterpret_cast<const bf16x8*>(ring + slot_y1 + cc);
                            const bf16x8 f1 = *reinterpret_cast<const bf16x8*>(ring + slot_y + cc);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[3], f0, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[4], f1, acc, 0, 0, 0);
                        }
                        if (valid) {
                            bf16x4 o;
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[r];
                            *reinterpret_cast<bf16x4*>(stage + (rr * Wi + 2 * j + b) * 128 + wave * 32 + q * 8) = o;
                        }
                    }
                }
            }
            __syncthreads();
            bf16* out = dx + obase;
            auto emit = [&](int px, const bf16x8& raw) {
                bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + px * 128 + chunk * 16);
                if (masked) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float r = (float)raw[e];
                        if (!(r * sc[e] + sh[e] > 0.f)) v[e] = (bf16)0.f;
                        const float fv = (float)v[e];
                        ssum[e] += fv; ssq[e] += fv * r;
                    }
                }
                *reinterpret_cast<bf16x8*>(out + (long)px * C) = v;
            };
            if (pre) {
                int ts = tid;
                asm volatile("" : "+v"(ts));
#pragma unroll
                for (int k = 0; k < CONV_MAXI; ++k) {
                    const int px = (ts >> 3) + k * 32;
                    if (px < P) emit(px, mraw[k]);
                }
            } else {
                for (int px = tid >> 3; px < P; px += 32)
                    emit(px, masked ? *reinterpret_cast<const bf16x8*>(mask_raw + obase + (long)px * C) : zero8());
            }
        }
    }
    if (masked) {
        STES_DET_LDS(det, _Pragma("unroll") for (int e = 0; e < 8; ++e) { atomicAdd(&red[chunk * 8 + e], ssum[e]); atomicAdd(&red[64 + chunk * 8 + e], ssq[e]); });
        __syncthreads();
        det_enter(det);
        if (tid < 128) atomicAdd((tid < 64 ? s_dy : s_dyr) + cb * 64 + (tid & 63), red[tid]);
        det_leave(det);
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
                                                             FastDiv fd_iw, StesDet det) {
    // one LDS block: staging tiles while walking items, then re-used to lay the result out for coalesced atomics
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 32 * 16 * 2 + 4 * MAX_HALO * 16 * 2];
    static_assert(sizeof(smem) >= 4608 * 4, "the result layout of two groups must fit the staging tiles");
    bf16* Pt = reinterpret_cast<bf16*>(smem);                     // [4 groups][32 px (rho)][16]
    bf16* It = reinterpret_cast<bf16*>(smem + 4 * 32 * 16 * 2);    // [4 groups][MAX_HALO][16]
    float* red = reinterpret_cast<float*>(smem);                  // [2 groups][16 co][16 ci][9]
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

    // The tiles of the next item travel global -> registers while the current one is multiplied (a tile is only 9 MFMAs of
    // work per wave, far less than a memory round trip); BatchNorm + ReLU is applied when the registers go to LDS, so the
    // loads are not waited for where they are issued.
    constexpr int MAXH = (MAX_HALO * 8 + 255) / 256;               // halo chunks per thread (5 at stride 2, 2 at stride 1)
    bf16x8 r_out, r_in[MAXH];
    unsigned in_ok = 0;                                            // bit k: r_in[k] came from inside the image
    auto fetch = [&](int item) {
        uint32_t n, tr, trow, tcol;
        fdivmod((uint32_t)item, fd_tiles, n, tr);
        fdivmod(tr, fd_tw, trow, tcol);
        const int ho0 = (int)trow * TH, wo0 = (int)tcol * TW;
        {   // dOut tile: thread -> pixel tid/8, chunk tid%8
            const int px = tid >> 3;
            const int ho = ho0 + (px >> 3), wo = wo0 + (px & 7);
            r_out = zero8();
            if (ho < Ho && wo < Wo) r_out = *reinterpret_cast<const bf16x8*>(dout + (((long)n * Ho + ho) * Wo + wo) * C + cbase);
        }
        in_ok = 0;
#pragma unroll
        for (int k = 0; k < MAXH; ++k) {
            const int i = tid + k * 256;
            r_in[k] = zero8();
            if (i < halo * 8) {
                const uint32_t hp = (uint32_t)i >> 3;
                uint32_t hr, hc;
                fdivmod(hp, fd_iw, hr, hc);
                const int hi = ho0 * stride - 1 + (int)hr, wi = wo0 * stride - 1 + (int)hc;
                if (hi >= 0 && hi < Hi && wi >= 0 && wi < Wi) {
                    r_in[k] = *reinterpret_cast<const bf16x8*>(in + (((long)n * Hi + hi) * Wi + wi) * C + cbase);
                    in_ok |= 1u << k;
                }
            }
        }
    };
    // Neighbouring items are neighbouring tiles of one image and share their halo pixels.  Workgroups go to the 8 XCDs (each
    // with its own L2) round-robin by index, so the index is permuted: XCD j walks a contiguous eighth of every round of items
    // and the shared pixels come out of its L2 instead of being fetched from memory once per tile.
    int first = blockIdx.x;
    if ((gridDim.x & 7) == 0) first = (blockIdx.x >> 3) + (blockIdx.x & 7) * (gridDim.x >> 3);
    if (first < n_items) fetch(first);
    for (int item = first; item < n_items; item += gridDim.x) {
        *reinterpret_cast<bf16x8*>(&Pt[(cgrp * 32 + rho32(tid >> 3)) * 16 + chalf * 8]) = r_out;
#pragma unroll
        for (int k = 0; k < MAXH; ++k) {
            const int i = tid + k * 256;
            if (i < halo * 8) {
                bf16x8 v = r_in[k];
                if (affine && (in_ok >> k & 1)) {
                    float f[8];
                    unpack8(v, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e] * sc[e] + sh[e], 0.f);
                    v = pack8(f);
                }
                *reinterpret_cast<bf16x8*>(&It[(cgrp * MAX_HALO + (i >> 3)) * 16 + chalf * 8]) = v;
            }
        }
        __syncthreads();
        if (item + (int)gridDim.x < n_items) fetch(item + gridDim.x);
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
    // result -> LDS in the checkpoint layout [co][ci][tap] of the 64 channels of this workgroup, then contiguous atomics; two
    // groups at a time, so that this layout needs no more LDS than the staging tiles (24 KiB: six workgroups per CU)
    float* dst = dw + (long)gq * 9216;
    det_enter(det);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((wave >> 1) == h) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(((wave & 1) * 16 + fq * 4 + r) * 16 + fr) * 9 + t] = acc[t][r];
        }
        __syncthreads();
        for (int i = tid; i < 4608; i += 256) atomicAdd(dst + h * 4608 + i, red[i]);
        __syncthreads();
    }
    det_leave(det);
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

// The same patches straight from decoded uint8 frames [B*Tv,Cin,H,W]: value = (u/255 - mean[ci]) / std[ci] in fp32, the
// arithmetic of reference dataset/frame.py:29 followed by :124 (and inference_on_mp4.py:523-524), so the bf16 patches are
// the ones the fp32 kernel gives on the loader's tensor.  Clips of Tv frames are laid out as clips of T frames: frames
// t >= Tv are the zero padding of reference train_e2e_spatial.py:427-430, which is zero AFTER normalisation.
struct StemNorm { float mean[3], stdv[3]; };

__global__ __launch_bounds__(256) void im2col_stem_u8_kernel(const uint8_t* __restrict__ x, bf16* __restrict__ out, StemNorm nm,
                                                             int B, int T, int Tv, int Cin, int H, int W, int Ho, int Wo) {
    const long total = (long)B * T * Ho * Wo;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (long)gridDim.x * blockDim.x) {
        const int wo = (int)(m % Wo); const long t0 = m / Wo; const int ho = (int)(t0 % Ho); const long n = t0 / Ho;
        const int tt = (int)(n % T); const long b = n / T;
        float f[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) f[k] = 0.f;
        if (tt < Tv) {
            const long ns = b * Tv + tt;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                if (ci >= Cin) break;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int hi = ho * 2 + kh - 1;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int wi = wo * 2 + kw - 1;
                        if (hi >= 0 && hi < H && wi >= 0 && wi < W) {
                            const float v = (float)x[((ns * Cin + ci) * H + hi) * (long)W + wi] / 255.0f;
                            f[ci * 9 + kh * 3 + kw] = (v - nm.mean[ci]) / nm.stdv[ci];
                        }
                    }
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

// ------------------------------------------------------------------ stem convolution in one pass
// The stem's 3x3 stride-2 convolution (3 -> 32 channels) without the round trip of its patches through memory.  A unit of
// work is STEM_R output rows of one image: the workgroup stages the 2*STEM_R + 1 input rows they need (three channel planes,
// bf16, data column 0 at element 4 of a row so that four columns are one aligned 8-byte LDS write; elements 3 and 4 + W are
// the zero halo) with every load of the unit issued before the first is waited for, and each wave then builds, for 16 output
// pixels at a time, the B operand of a 16x16x32 MFMA straight from there (lane (pixel, q) gathers taps 8q..8q+7 of its pixel:
// k = ci*9 + kh*3 + kw, zero from 27 up) against the two 16-channel halves of the weight matrix as A operands.  The result
// D[channel 4q + i][pixel] gives every lane four consecutive channels of its pixel twice: two 8-byte stores, no transposition.
// Workgroups walk over units with a grid stride and commit their column statistics once at the end (one atomic per channel and
// workgroup: the 64 sums live in two cache lines, so their number decides the cost).
// Training: the raw output, its column statistics and -- because the weight gradient reads them -- the patches, written from
// the operand registers.  Evaluation: relu(scale * c + shift) instead of the raw output and, on request, the first channels
// of every pixel as the next layer's compact gated-shift slice; nothing else is written.
// U8: x holds decoded uint8 frames and the normalisation (u / 255 - mean) / std of im2col_stem_u8_kernel is applied when the
// rows are staged; clips of Tv frames are laid out as clips of T frames, the missing ones zero.
// VEC: W is a multiple of 4 and x is aligned for four-column loads.
constexpr int STEM_R = 4;
constexpr int STEM_ROWS = 2 * STEM_R + 1;
constexpr int STEM_LD = 6;                                                   // four-column loads in flight per thread

struct StemGeom { int T, Tv, Cin, H, W, Ho, Wo, groups, LW; long units; };

template <bool U8, bool EVAL, bool VEC>
__global__ __launch_bounds__(256) void stem_conv_kernel(const void* __restrict__ xin, StemNorm nm, const bf16* __restrict__ wbf,
                                                        bf16* __restrict__ patches, bf16* __restrict__ out, float* col_sum,
                                                        float* col_sumsq, const float* __restrict__ e_scale,
                                                        const float* __restrict__ e_shift, bf16* __restrict__ e_slice, int e_slice_n,
                                                        StemGeom gm, StesDet det) {
    extern __shared__ __attribute__((aligned(16))) unsigned char stem_smem[];
    bf16* rows = reinterpret_cast<bf16*>(stem_smem);                        // [3 channels][STEM_ROWS][LW]
    const int LW = gm.LW, W = gm.W, H = gm.H, Wo = gm.Wo, Ho = gm.Ho, Cin = gm.Cin;
    float* red = reinterpret_cast<float*>(rows + 3 * STEM_ROWS * LW);        // [2][4 waves][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 15, q = lane >> 4;
    const int nq = (W + 3) >> 2, quads = 3 * STEM_ROWS * nq;                 // four-column pieces of a row, of a unit
    // this lane's eight taps, as offsets from the staged element of (input row 2*orow - 1 of the unit's first, column 2*wo - 1)
    int toff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j;
        const int ci = k / 9, kh = (k - ci * 9) / 3, kw = k - ci * 9 - kh * 3;
        toff[j] = k < 27 ? (ci * STEM_ROWS + kh) * LW + kw + 3 : -1;
    }
    const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wbf + (long)p * 32 + q * 8);            // A operand: row = output channel
    const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wbf + (long)(16 + p) * 32 + q * 8);
    float es[8], eb[8];
    if (EVAL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            es[i] = e_scale[4 * q + i]; es[4 + i] = e_scale[16 + 4 * q + i];
            eb[i] = e_shift[4 * q + i]; eb[4 + i] = e_shift[16 + 4 * q + i];
        }
    }
    float ssum[8], ssq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ssum[i] = ssq[i] = 0.f;
    // zero halo columns (never written by the staging below)
    for (int i = tid; i < 3 * STEM_ROWS * 2; i += 256) rows[(i >> 1) * LW + ((i & 1) ? 4 + W : 3)] = (bf16)0.f;

    for (long unit = blockIdx.x; unit < gm.units; unit += gridDim.x) {
        const int n = (int)(unit / gm.groups), grp = (int)(unit - (long)n * gm.groups);   // image (of B * T), group of output rows
        const int ho0 = STEM_R * grp;
        long src_img = n;
        bool live = true;
        if (U8) { const int b = n / gm.T, t = n - b * gm.T; live = t < gm.Tv; src_img = (long)b * gm.Tv + t; }
        // ---- stage input rows 2*ho0 - 1 .. 2*ho0 + 2*STEM_R - 1 of the three planes
        for (int e0 = 0; e0 < quads; e0 += 256 * STEM_LD) {
            float4 v[STEM_LD];
            int at[STEM_LD];
#pragma unroll
            for (int u = 0; u < STEM_LD; ++u) {
                const int e = e0 + u * 256 + tid;
                const int rp = e / nq, c4 = e - rp * nq;                     // row-plane (ci * STEM_ROWS + row), piece
                const int ci = rp / STEM_ROWS, r = 2 * ho0 - 1 + (rp - ci * STEM_ROWS);
                at[u] = e < quads ? rp * LW + 4 + 4 * c4 : -1;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < quads && live && ci < Cin && r >= 0 && r < H) {
                    const long o = ((src_img * Cin + ci) * H + r) * (long)W + 4 * c4;
                    if (U8) {
                        const float mu = ci == 0 ? nm.mean[0] : ci == 1 ? nm.mean[1] : nm.mean[2];
                        const float sd = ci == 0 ? nm.stdv[0] : ci == 1 ? nm.stdv[1] : nm.stdv[2];
                        const uint8_t* xs = reinterpret_cast<const uint8_t*>(xin) + o;
                        unsigned b4 = 0;
                        if (VEC) b4 = *reinterpret_cast<const unsigned*>(xs);
                        else {
#pragma unroll
                            for (int z = 0; z < 4; ++z) if (4 * c4 + z < W) b4 |= (unsigned)xs[z] << (8 * z);
                        }
                        v[u].x = ((float)(b4 & 255u) / 255.0f - mu) / sd;         v[u].y = ((float)((b4 >> 8) & 255u) / 255.0f - mu) / sd;
                        v[u].z = ((float)((b4 >> 16) & 255u) / 255.0f - mu) / sd; v[u].w = ((float)(b4 >> 24) / 255.0f - mu) / sd;
                    } else {
                        const float* xs = reinterpret_cast<const float*>(xin) + o;
                        if (VEC) v[u] = *reinterpret_cast<const float4*>(xs);
                        else {
                            v[u].x = xs[0];
                            if (4 * c4 + 1 < W) v[u].y = xs[1];
                            if (4 * c4 + 2 < W) v[u].z = xs[2];
                            if (4 * c4 + 3 < W) v[u].w = xs[3];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < STEM_LD; ++u) {
                if (at[u] < 0) continue;
                bf16x4 h;
                h[0] = (bf16)v[u].x; h[1] = (bf16)v[u].y; h[2] = (bf16)v[u].z; h[3] = (bf16)v[u].w;
                if (VEC) *reinterpret_cast<bf16x4*>(rows + at[u]) = h;
                else {
                    const int c0 = at[u] - (at[u] / LW) * LW - 4;                  // first column of the piece
#pragma unroll
                    for (int z = 0; z < 4; ++z) if (c0 + z < W) rows[at[u] + z] = h[z];   // (column W is the zero halo)
                }
            }
        }
        __syncthreads();
        const int nrow = min(STEM_R, Ho - ho0);
        const int npx = nrow * Wo;                                           // output pixels of the unit, row-major over its rows
        const long pix0 = ((long)n * Ho + ho0) * Wo;
        for (int t0 = wave * 16; t0 < npx; t0 += 64) {
            const bool ok = t0 + p < npx;
            const int px = ok ? t0 + p : npx - 1;
            const int orow = (px >= Wo) + (px >= 2 * Wo) + (px >= 3 * Wo);
            static_assert(STEM_R == 4, "the row of a pixel is found by three comparisons");
            const int wo = px - orow * Wo;
            const bf16* base = rows + 2 * orow * LW + 2 * wo;
            bf16x8 a;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16 t = base[toff[j] >= 0 ? toff[j] : 3];
                a[j] = toff[j] >= 0 ? t : (bf16)0.f;
            }
            if (!EVAL && ok) *reinterpret_cast<bf16x8*>(patches + (pix0 + px) * 32 + q * 8) = a;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, a, z, 0, 0, 0);       // D[channel 4q + i][pixel p]
            const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, a, z, 0, 0, 0);
            bf16x4 o0, o1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v0 = d0[i], v1 = d1[i];
                if (EVAL) { v0 = fmaxf(v0 * es[i] + eb[i], 0.f); v1 = fmaxf(v1 * es[4 + i] + eb[4 + i], 0.f); }
                o0[i] = (bf16)v0; o1[i] = (bf16)v1;
                if (!EVAL) {
                    const float f0 = ok ? (float)o0[i] : 0.f, f1 = ok ? (float)o1[i] : 0.f;
                    ssum[i] += f0; ssq[i] += f0 * f0; ssum[4 + i] += f1; ssq[4 + i] += f1 * f1;
                }
            }
            if (ok) {
                bf16* orow_p = out + (pix0 + px) * 32;
                *reinterpret_cast<bf16x4*>(orow_p + 4 * q) = o0;
                *reinterpret_cast<bf16x4*>(orow_p + 16 + 4 * q) = o1;
                if (EVAL && e_slice) {
                    if (4 * q < e_slice_n) *reinterpret_cast<bf16x4*>(e_slice + (pix0 + px) * e_slice_n + 4 * q) = o0;
                    if (16 + 4 * q < e_slice_n) *reinterpret_cast<bf16x4*>(e_slice + (pix0 + px) * e_slice_n + 16 + 4 * q) = o1;
                }
            }
        }
        __syncthreads();                                                     // the rows are overwritten by the next unit
    }
    if (!EVAL && col_sum) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float a = ssum[i], b = ssq[i];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { a += __shfl_xor(a, m, 64); b += __shfl_xor(b, m, 64); }
            if (p == 0) {
                const int c = (i < 4 ? 0 : 16) + 4 * q + (i & 3);
                red[wave * 32 + c] = a; red[128 + wave * 32 + c] = b;
            }
        }
        __syncthreads();
        det_enter(det);
        if (tid < 64) {
            const int which = tid >> 5, c = tid & 31;
            const float* r = red + which * 128 + c;
            atomicAdd((which ? col_sumsq : col_sum) + c, (r[0] + r[32]) + (r[64] + r[96]));
        }
        det_leave(det);
    }
}

}  // namespace

#define S(s) ((hipStream_t)(s))

extern "C" int stes_pack_conv3x3g(const float* w, uint16_t* wp_fwd, uint16_t* wp_dgrad, int G, int stride, void* stream) {
    STES_CHECK(w && wp_fwd && wp_dgrad && G > 0 && (stride == 1 || stride == 2), STES_ERR_ARG, "stes_pack_conv3x3g: bad argument");
    const int total = G * 5 * 64 * 8;
    hipLaunchKernelGGL(pack_conv3x3g_kernel, dim3(stes_div_up(total, 256)), dim3(256), 0, S(stream), w, (bf16*)wp_fwd, (bf16*)wp_dgrad, G, stride);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

static int conv_launch(int mode, const uint16_t* src, const uint16_t* wp, uint16_t* dst, const float* scale, const float* shift,
                       float* col_sum, float* col_sumsq, const uint16_t* mask_raw, int N, int Hi, int Wi, int Ho, int Wo, int stride,
                       int C, void* stream, const float* pool_scale = nullptr, const float* pool_shift = nullptr,
                       uint16_t* pool_out = nullptr, float pool_norm = 0.f) {
    STES_CHECK(src && wp && dst, STES_ERR_ARG, "stes_conv3x3g: null pointer");
    STES_CHECK(C % 64 == 0, STES_ERR_ARG, "stes_conv3x3g: channels must be a multiple of 64 (groups of 16, 4 per workgroup)");
    STES_CHECK(stride == 1 || stride == 2, STES_ERR_ARG, "stes_conv3x3g: stride must be 1 or 2");
    STES_CHECK(Ho == (Hi + 2 - 3) / stride + 1 && Wo == (Wi + 2 - 3) / stride + 1, STES_ERR_ARG, "stes_conv3x3g: output size does not match");
    STES_CHECK((scale == nullptr) == (shift == nullptr) && (col_sum == nullptr) == (col_sumsq == nullptr), STES_ERR_ARG, "stes_conv3x3g: paired pointers");
    if (stes_conv_ring1_takes(mode, stride, Wi, pool_out != nullptr))
        return stes_conv_ring1_launch(mode, src, wp, dst, mask_raw, scale, shift, col_sum, col_sumsq, pool_scale, pool_shift, pool_out, pool_norm, N,
                                      Hi, Wi, C, stream);
    if (stes_conv_strip_takes(mode, stride, Wi))         // 14 x 14 and 7 x 7 maps, forward: operands straight from memory, no LDS window
        return stes_conv_strip_launch(src, wp, dst, scale, shift, col_sum, col_sumsq, N, Hi, Wi, C, stream, pool_scale, pool_shift, pool_out,
                                      pool_norm);
    if (stes_conv_ring_takes(mode, stride, Hi, Wi, pool_out != nullptr))     // the strided forward of the large maps: read-each-pixel-2.25-times ring
        return stes_conv_ring_launch(src, wp, dst, scale, shift, col_sum, col_sumsq, pool_scale, pool_shift, pool_out, pool_norm, N, Hi, Wi, C, stream);
    if (stes_dgrad_ring_takes(mode, stride, Hi, Wi))
        return stes_dgrad_ring_launch(src, wp, dst, mask_raw, scale, shift, col_sum, col_sumsq, N, Hi, Wi, C, stream);
    if (mode == 0 || stride == 1) {
        // LDS row-ring kernel when its ring + staging fit
        RowsGeom rg;
        rg.N = N; rg.C = C;
        if (mode == 0) { rg.Hs = Hi; rg.Ws = Wi; rg.Hd = Ho; rg.Wd = Wo; }
        else { rg.Hs = Ho; rg.Ws = Wo; rg.Hd = Hi; rg.Wd = Wi; }
        const int st = mode == 0 ? stride : 1;
        auto lds_of = [&](int rs) {
            const int nr = st == 1 ? rs + 2 : 2 * rs + 1;
            const size_t staging = (size_t)rs * rg.Wd * 128 < 1024 ? 1024 : (size_t)rs * rg.Wd * 128;    // see the kernel
            return (size_t)nr * (rg.Ws + 2) * PSB + staging + 128 * sizeof(float) + CONV_NF * CONV_NG * 16 * sizeof(int);
        };
        int cap = CONV_MAXP / rg.Wd; if (cap < 1) cap = 1; if (cap > rg.Hd) cap = rg.Hd;
        // more than one step per image: the rows carried from step to step must fit their registers
        const bool carry_ok = (st == 1 ? 2 : 1) * rg.Ws * 8 <= CONV_OVI * 256;

        int best = 0, best_steps = 1 << 30;
        for (int rs = cap; rs >= 1; --rs) {
            if ((long)rs * st * rg.Ws * 8 > CONV_MAXI * 256 || lds_of(rs) > 72 * 1024 || (rs < rg.Hd && !carry_ok) || rs * rg.Wd > CONV_MAXP) continue;
            const int steps = stes_div_up(rg.Hd, rs);
            if (steps < best_steps) { best_steps = steps; best = rs; }
        }
        if (best == 0 && (long)st * rg.Ws * 8 <= CONV_MAXI * 256 && lds_of(1) <= 150 * 1024 && (rg.Hd == 1 || carry_ok) && rg.Wd <= CONV_MAXP) best = 1;
        if (best > 0 && (long)N * (C / 64) < (1L << 31)) {
            rg.RS = best; rg.NR = st == 1 ? best + 2 : 2 * best + 1;
            rg.fd_wd = make_fastdiv(rg.Wd); rg.fd_px8 = make_fastdiv(rg.Ws * 8);
            const size_t lds = lds_of(best);
            const int ncb = C / 64;
            static const bool big_lds = [] {
                bool ok = true;
                ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3g_rows_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
                ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3g_rows_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
                ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3g_rows_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
                ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3g_rows_kernel<1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
                ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3g_rows_kernel<2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
                return ok;
            }();
            // as many workgroups as are resident at once: every further one would only add device-scope statistics
            // atomics, which serialise at the end of the launch (residency cached per kernel form and LDS size)
            const int form = mode == 1 ? 2 : (stride == 1 ? 0 : 1);
            static size_t seen_lds[3][8];
            static int seen_val[3][8], n_seen[3] = {0, 0, 0};
            static std::mutex mu;                  // entry points may be called from any host thread
            std::lock_guard<std::mutex> lock(mu);
            int resident = 0;
            for (int i = 0; i < n_seen[form]; ++i) if (seen_lds[form][i] == lds) resident = seen_val[form][i];
            if (resident == 0) {
                hipError_t e = hipSuccess;
                if (form == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, conv3x3g_rows_kernel<1, true>, 256, lds);
                else if (form == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, conv3x3g_rows_kernel<1, false>, 256, lds);
                else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, conv3x3g_rows_kernel<2, false>, 256, lds);
                if (e != hipSuccess || resident < 1) resident = 1;
                if (n_seen[form] < 8) { seen_lds[form][n_seen[form]] = lds; seen_val[form][n_seen[form]] = resident; ++n_seen[form]; }
            }
            int per = 256 * resident / ncb; if (per < 1) per = 1; if (per > N) per = N;
            const dim3 grid(per * ncb);
            if (big_lds || lds <= 64 * 1024) {
#define ROWS_LAUNCH(ST, FL, PL) hipLaunchKernelGGL((conv3x3g_rows_kernel<ST, FL, PL>), grid, dim3(256), lds, S(stream), (const bf16*)src, (const bf16*)wp, \
        (bf16*)dst, scale, shift, col_sum, col_sumsq, (const bf16*)mask_raw, rg, pool_scale, pool_shift, (bf16*)pool_out, pool_norm, \
        col_sum ? stes_det_take(grid.x) : StesDet{nullptr, 0u, 0u})
                if (mode == 1) ROWS_LAUNCH(1, true, false);
                else if (pool_out && stride == 1) ROWS_LAUNCH(1, false, true);
                else if (pool_out) ROWS_LAUNCH(2, false, true);
                else if (stride == 1) ROWS_LAUNCH(1, false, false);
                else ROWS_LAUNCH(2, false, false);
#undef ROWS_LAUNCH
                STES_LAUNCH_CHECK();
                return STES_OK;
            }
        }
    }
    STES_CHECK(!pool_out, STES_ERR_UNSUPPORTED, "stes_conv3x3g_fwd_pool: image too large for the LDS window kernel; run stes_chan_pool");
    if (mode == 1 && stride == 2) {
        Dgrad2Geom dg;
        dg.N = N; dg.Ho = Ho; dg.Wo = Wo; dg.Hi = Hi; dg.Wi = Wi; dg.C = C;
        auto lds_of = [&](int r) { return (size_t)(r + 1) * (Wo + 2) * PSB + (size_t)2 * r * Wi * 128 + 128 * sizeof(float); };
        int r = 112 / Wi; if (r < 1) r = 1; if (r > Ho) r = Ho;
        while (r > 1 && ((long)r * Wo * 8 > CONV_MAXI * 256 || lds_of(r) > 72 * 1024)) --r;
        STES_CHECK((long)r * Wo * 8 <= CONV_MAXI * 256 && lds_of(r) <= 150 * 1024, STES_ERR_UNSUPPORTED,
                   "stes_conv3x3g_dgrad: image too wide for the stride-2 LDS tile");
        dg.R = r; dg.fd_px8 = make_fastdiv(Wo * 8);
        const size_t lds = lds_of(r);
        static const bool big = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3g_dgrad2_kernel<true>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                                hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3g_dgrad2_kernel<false>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
        STES_CHECK(big || lds <= 64 * 1024, STES_ERR_UNSUPPORTED, "stes_conv3x3g_dgrad: cannot raise the dynamic LDS limit");
        static size_t seen_lds[8]; static int seen_val[8], n_seen = 0;
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        int resident = 0;
        for (int i = 0; i < n_seen; ++i) if (seen_lds[i] == lds) resident = seen_val[i];
        if (resident == 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, conv3x3g_dgrad2_kernel<true>, 256, lds) != hipSuccess || resident < 1) resident = 1;
            if (n_seen < 8) { seen_lds[n_seen] = lds; seen_val[n_seen] = resident; ++n_seen; }
        }
        const int ncb = C / 64;
        int per = 256 * resident / ncb; if (per < 1) per = 1; if (per > N) per = N;
        // with the mask the batched fragment loop is the faster one (943 vs 1077 us at 112 x 112 x 64, 490 vs 555 at 56 x 56 x 128),
        // without it the plain loop (667 vs 761 us): measured, both forms kept
        if (mask_raw)
            hipLaunchKernelGGL(conv3x3g_dgrad2_kernel<true>, dim3(per * ncb), dim3(256), lds, S(stream), (const bf16*)src, (const bf16*)wp, (bf16*)dst,
                               (const bf16*)mask_raw, scale, shift, col_sum, col_sumsq, dg, stes_det_take(per * ncb));
        else
            hipLaunchKernelGGL(conv3x3g_dgrad2_kernel<false>, dim3(per * ncb), dim3(256), lds, S(stream), (const bf16*)src, (const bf16*)wp, (bf16*)dst,
                               (const bf16*)mask_raw, scale, shift, col_sum, col_sumsq, dg, StesDet{nullptr, 0u, 0u});
        STES_LAUNCH_CHECK();
        return STES_OK;
    }
    STES_CHECK(!mask_raw, STES_ERR_UNSUPPORTED, "stes_conv3x3g_dgrad: image too wide for the fused mask + BatchNorm sums");
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
        (const bf16*)wp, (bf16*)dst, scale, shift, col_sum, col_sumsq, gm, n_tiles, col_sum ? stes_det_take(gx * gy) : StesDet{nullptr, 0u, 0u})
    if (mode == 0) { if (stride == 1) CONV_LAUNCH(0, 1); else CONV_LAUNCH(0, 2); }
    else CONV_LAUNCH(1, 1);
#undef CONV_LAUNCH
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_conv3x3g_fwd(const uint16_t* in, const uint16_t* wp_fwd, uint16_t* out, const float* in_scale,
                                 const float* in_shift, float* col_sum, float* col_sumsq, int N, int Hi, int Wi, int Ho, int Wo,
                                 int stride, int C, void* stream) {
    return conv_launch(0, in, wp_fwd, out, in_scale, in_shift, col_sum, col_sumsq, nullptr, N, Hi, Wi, Ho, Wo, stride, C, stream);
}

extern "C" int stes_conv3x3g_fwd_pool(const uint16_t* in, const uint16_t* wp_fwd, uint16_t* out, const float* in_scale,
                                      const float* in_shift, const float* pool_scale, const float* pool_shift, uint16_t* pool_out,
                                      int N, int Hi, int Wi, int Ho, int Wo, int stride, int C, void* stream) {
    STES_CHECK(pool_scale && pool_shift && pool_out, STES_ERR_ARG, "stes_conv3x3g_fwd_pool: the pooling affine and its output are required");
    return conv_launch(0, in, wp_fwd, out, in_scale, in_shift, nullptr, nullptr, nullptr, N, Hi, Wi, Ho, Wo, stride, C, stream, pool_scale,
                       pool_shift, pool_out, 1.0f / (float)(Ho * Wo));
}

extern "C" int stes_conv3x3g_dgrad(const uint16_t* dout, const uint16_t* wp_dgrad, uint16_t* din, const uint16_t* mask_raw,
                                   const float* mask_scale, const float* mask_shift, float* s_dy, float* s_dyr, int N, int Hi,
                                   int Wi, int Ho, int Wo, int stride, int C, void* stream) {
    STES_CHECK(!mask_raw || (mask_scale && mask_shift && s_dy && s_dyr), STES_ERR_ARG, "stes_conv3x3g_dgrad: the mask needs its affine and both sums");
    return conv_launch(1, dout, wp_dgrad, din, mask_raw ? mask_scale : nullptr, mask_raw ? mask_shift : nullptr, mask_raw ? s_dy : nullptr,
                       mask_raw ? s_dyr : nullptr, mask_raw, N, Hi, Wi, Ho, Wo, stride, C, stream);
}

extern "C" int stes_conv3x3g_wgrad(const uint16_t* dout, const uint16_t* in, const float* in_scale, const float* in_shift, float* dw,
                                   int N, int Hi, int Wi, int Ho, int Wo, int stride, int C, void* stream) {
    STES_CHECK(dout && in && dw, STES_ERR_ARG, "stes_conv3x3g_wgrad: null pointer");
    STES_CHECK(C % 64 == 0 && (stride == 1 || stride == 2), STES_ERR_ARG, "stes_conv3x3g_wgrad: C % 64, stride 1|2");
    STES_CHECK(Ho == (Hi + 2 - 3) / stride + 1 && Wo == (Wi + 2 - 3) / stride + 1, STES_ERR_ARG, "stes_conv3x3g_wgrad: output size does not match");
    if (stes_wgrad_ring_takes(stride, Wi)) return stes_wgrad_ring_launch(dout, in, in_scale, in_shift, dw, N, Hi, Wi, C, stream);
    const int th = stes_div_up(Ho, TH), tw = stes_div_up(Wo, TW);
    const int n_items = N * th * tw;
    const int gy = C / 64;
    int gx = n_items;
    // exactly one resident set of workgroups (three per CU, by registers): measured over grid sizes, the time has a sharp
    // minimum there (364 us at 768 workgroups, 494 at 832, 396 at 640 for 112 x 112 x 64) -- a second, partial round runs on a
    // part-empty GPU, and every workgroup ends with 36 KiB of atomics into the same weight gradient
    static const int resident = [] {
        int occ = 0;
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv3x3g_wgrad_kernel, 256, 0) == hipSuccess && occ > 0 ? occ : 1;
    }();
    const int cap = 256 * resident / gy > 1 ? 256 * resident / gy : 1;
    if (gx > cap) gx = cap;
    if (gx > 8) gx &= ~7;                      // a multiple of 8: see the kernel's item permutation
    const int IW = (TW - 1) * stride + 3;
    hipLaunchKernelGGL(conv3x3g_wgrad_kernel, dim3(gx, gy), dim3(256), 0, S(stream), (const bf16*)dout, (const bf16*)in, in_scale, in_shift,
                       dw, N, Hi, Wi, Ho, Wo, stride, C, n_items, make_fastdiv(th * tw), make_fastdiv(tw), make_fastdiv(IW),
                       stes_det_take(gx * gy));
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_im2col_stem_u8(const uint8_t* x, uint16_t* patches, const float* mean, const float* stdv, int B, int T,
                                   int T_valid, int Cin, int H, int W, void* stream) {
    STES_CHECK(x && patches && mean && stdv && B > 0 && T > 0 && T_valid > 0 && T_valid <= T && Cin >= 1 && Cin <= 3 && H > 0 && W > 0,
               STES_ERR_ARG, "stes_im2col_stem_u8: bad argument");
    StemNorm nm;
    for (int c = 0; c < 3; ++c) {
        nm.mean[c] = c < Cin ? mean[c] : 0.f;
        nm.stdv[c] = c < Cin ? stdv[c] : 1.f;
        STES_CHECK(nm.stdv[c] > 0.f, STES_ERR_ARG, "stes_im2col_stem_u8: std must be positive");
    }
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long total = (long)B * T * Ho * Wo;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(im2col_stem_u8_kernel, dim3((int)blocks), dim3(256), 0, S(stream), x, (bf16*)patches, nm, B, T, T_valid, Cin, H, W,
                       Ho, Wo);
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

// Workgroups of a stem kernel that are resident at once on the whole GPU (its grid: the units are walked with a grid stride).
template <class K>
static long stem_resident(K kernel, size_t lds) {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return (long)per_cu * (cus > 0 ? cus : 256);
}

// x: [N][Cin][H][W] fp32 (x_is_u8 == 0), or [B*T_valid][Cin][H][W] uint8 laid out as B clips of T frames (the missing frames
// zero after normalisation) with HOST mean/std of Cin floats.  w_bf: the stem weight [32][32] bf16, row = output channel,
// column k = ci*9 + kh*3 + kw.  Training form (e_scale == NULL): out = raw conv output [N*Ho*Wo][32], patches likewise
// (required), column statistics added to col_sum / col_sumsq when given.  Evaluation form: out = relu(e_scale*c + e_shift),
// e_slice (optional) = its first e_slice_n channels as [N*Ho*Wo][e_slice_n].
extern "C" int stes_stem_conv(const void* x, int x_is_u8, const float* mean, const float* stdv, const uint16_t* w_bf, uint16_t* patches,
                              uint16_t* out, float* col_sum, float* col_sumsq, const float* e_scale, const float* e_shift,
                              uint16_t* e_slice, int e_slice_n, int B, int T, int T_valid, int Cin, int H, int W, void* stream) {
    STES_CHECK(x && w_bf && out && B > 0 && T > 0 && T_valid > 0 && T_valid <= T && Cin >= 1 && Cin <= 3 && H > 0 && W > 0, STES_ERR_ARG,
               "stes_stem_conv: bad argument");
    STES_CHECK(x_is_u8 || T_valid == T, STES_ERR_ARG, "stes_stem_conv: only uint8 clips are padded in time");
    STES_CHECK((e_scale == nullptr) == (e_shift == nullptr) && (col_sum == nullptr) == (col_sumsq == nullptr), STES_ERR_ARG, "stes_stem_conv: paired pointers");
    const bool eval = e_scale != nullptr;
    STES_CHECK(eval || patches, STES_ERR_ARG, "stes_stem_conv: the training form writes the patches");
    STES_CHECK(!eval || (!col_sum && (!e_slice || (e_slice_n > 0 && e_slice_n <= 32 && e_slice_n % 4 == 0))), STES_ERR_ARG, "stes_stem_conv: evaluation form arguments");
    StemNorm nm;
    for (int c = 0; c < 3; ++c) {
        nm.mean[c] = (x_is_u8 && c < Cin) ? mean[c] : 0.f;
        nm.stdv[c] = (x_is_u8 && c < Cin) ? stdv[c] : 1.f;
        STES_CHECK(nm.stdv[c] > 0.f, STES_ERR_ARG, "stes_stem_conv: std must be positive");
    }
    STES_CHECK(!x_is_u8 || (mean && stdv), STES_ERR_ARG, "stes_stem_conv: uint8 input needs mean and std");
    StemGeom gm;
    gm.T = T; gm.Tv = T_valid; gm.Cin = Cin; gm.H = H; gm.W = W;
    gm.Ho = (H - 1) / 2 + 1; gm.Wo = (W - 1) / 2 + 1; gm.groups = (gm.Ho + STEM_R - 1) / STEM_R;
    gm.LW = (W + 8 + 3) & ~3;
    gm.units = (long)B * T * gm.groups;
    STES_CHECK((long)B * T * gm.Ho * gm.Wo < (1L << 31) / 32, STES_ERR_UNSUPPORTED, "stes_stem_conv: too many pixels");
    const size_t lds = (size_t)3 * STEM_ROWS * gm.LW * 2 + 2 * 4 * 32 * sizeof(float);
    STES_CHECK(lds <= 64 * 1024, STES_ERR_UNSUPPORTED, "stes_stem_conv: image too wide");
    const bool vec = W % 4 == 0 && (uintptr_t)x % (x_is_u8 ? 4 : 16) == 0;
    const bool det_on = col_sum != nullptr;
#define STEM_LAUNCH(U, E, V) do { \
        const long cap = stem_resident(stem_conv_kernel<U, E, V>, lds); \
        const unsigned blocks = (unsigned)(gm.units < cap ? gm.units : cap); \
        const StesDet det = det_on ? stes_det_take(blocks) : StesDet{nullptr, 0u, 0u}; \
        hipLaunchKernelGGL((stem_conv_kernel<U, E, V>), dim3(blocks), dim3(256), lds, S(stream), x, nm, (const bf16*)w_bf, \
            (bf16*)patches, (bf16*)out, col_sum, col_sumsq, e_scale, e_shift, (bf16*)e_slice, e_slice_n, gm, det); } while (0)
#define STEM_LAUNCH_V(U, E) do { if (vec) STEM_LAUNCH(U, E, true); else STEM_LAUNCH(U, E, false); } while (0)
    if (x_is_u8) { if (eval) STEM_LAUNCH_V(true, true); else STEM_LAUNCH_V(true, false); }
    else { if (eval) STEM_LAUNCH_V(false, true); else STEM_LAUNCH_V(false, false); }
#undef STEM_LAUNCH_V
#undef STEM_LAUNCH
    STES_LAUNCH_CHECK();
    return STES_OK;
}
