// Evaluation forward of a whole stage-3 bottleneck (RegNet-Y 800MF: 14 x 14 x 320, stride 1, identity shortcut) in one kernel,
// one image per workgroup pass.  timm Bottleneck as the reference builds it (train_e2e_spatial.py:305-314) with the gated
// shift's result G substituted for the first 80 input channels of conv1 (model/shift.py:157-161), BatchNorms in evaluation
// mode (fixed affine maps):
//   c1 = [G | X] W1^T;  c2 = conv3x3_{groups of 16}(relu(bn1(c1)));  s = mean_hw relu(bn2(c2));
//   gate = sigmoid(W_se2 relu(W_se1 s + b1) + b2);  c3 = (relu(bn2(c2)) * gate) W3^T;  out = relu(bn3(c3) + X).
// As separate kernels these are seven passes over a 128 MB tensor per block; here an image's activations (196 x 320 bf16 =
// 125 KB) stay in LDS between the three convolutions and the block costs one read of X and G, one more read of X for the
// shortcut (nowhere to keep it: LDS is full) and one write of the result.  512 threads, one workgroup per compute unit:
//   phase 1  conv1 from the input rows in LDS: the 20 column fragments of the 320 x 320 product go to the 8 waves as 3 or 2
//            each (a SIMD runs one wave of each kind), every wave takes all 13 row tiles, so each weight fragment is fetched
//            by exactly one wave (the 205 KB matrix is L2 traffic for every image); result -> relu(bn1(.)) -> LDS in place
//            of the input once every wave has finished reading;
//   phase 2  conv2 in place: a wave owns a group of 16 channels at a time; lane = image column (a 16-column fragment holds a
//            14-pixel row plus zero borders), every source row is read once and feeds nine K = 16 MFMAs, the taps' column
//            shifts are DPP row shifts of the accumulators; raw c2 (bf16) replaces the group's channels row by row, a row
//            behind the reads; the squeeze is summed on the way;
//   phase 3  the two squeeze-excite products on the 320-vector, then relu(bn2(c2)) * gate in place;
//   phase 4  conv3 like phase 1; the rounded product goes back to LDS and leaves in a chunk-wise pass that adds bn3 and the
//            shortcut, writes the next block's gated-shift slice, and puts the workgroup's next image's input chunk where it
//            has just taken the product's out (the next input's reads fly together with the shortcut's reads and the writes).
// Measured per image on a compute unit: products 7.5 us each (MFMA time alone: 4.3), conv2 14 (VALU issue), squeeze-excite 8,
// epilogues 5, the pass over memory 23 (all compute units are in it at the same time and together they move 5 TB/s).
// Every rounding is the one the kernel sequence makes (raw conv outputs rounded to bf16 before their BatchNorm, the squeeze
// and the FC vector in bf16, the block-tail arithmetic of bn_act_residual), so results agree to fp32 summation order.
#include "stes_common.h"

namespace {

constexpr int BE_C = 320, BE_W = 14, BE_HW = 196, BE_RD = 80, BE_FOLD = 80;
constexpr int BE_LD = BE_C + 8;                 // LDS row stride (elements): 656 bytes (672, conflict-free for the b128 fragment reads, measured no faster)
constexpr int BE_THREADS = 512;
constexpr int BE_CHUNKS = BE_HW * (BE_C / 8);    // 16-byte chunks of an image
constexpr int BE_BATCH = 16;                    // chunks a thread has in flight in the passes over memory
constexpr int BE_ROWS = 208;                    // 13 tiles of 16 rows
constexpr int BE_LDS = BE_ROWS * BE_LD * 2 + (7 * BE_C + 128) * 4;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2b __attribute__((ext_vector_type(2)));
#define BE_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float be_from_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}
__device__ __forceinline__ float be_from_prev(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
// sum over the 16 lanes of a row, in every lane (row rotations: no LDS round trips)
__device__ __forceinline__ float be_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}
// A operand of one tap from the k-step-pair pack of stes_pack_conv3x3g ([G][5][64][8], see stes_conv_strip.hip)
__device__ __forceinline__ s16x4 be_tap_frag(const bf16* __restrict__ wp, int g, int slot, int lane) {
    const int m = lane & 15, q = lane >> 4;
    const int packed_lane = ((slot & 1) * 2 + (q >> 1)) * 16 + m;
    return *reinterpret_cast<const s16x4*>(wp + (((long)g * 5 + (slot >> 1)) * 64 + packed_lane) * 8 + 4 * (q & 1));
}

// one of the two 320 x 320 products: D[n][m] += W[n][k] A[m][k], NC column fragments (16 output channels each, from fragment
// col0) for all 13 row tiles.  Every weight fragment is fetched by exactly one wave of the workgroup (the 205 KB matrix comes
// from L2 for every image, and L2 gives a compute unit about 70 GB/s), BE_AHEAD k-steps ahead; the activations come from LDS,
// a row tile's next fragment as soon as its products are issued.
constexpr int BE_AHEAD = 2, BE_RING = 6;      // weight fragments: k-steps ahead; activation fragments: row tiles ahead
template <int NC>
__device__ __forceinline__ void be_gemm(const bf16* __restrict__ Wm, const bf16* act, int col0, int fr, int fq, f32x4 (&acc)[13][3]) {
#pragma unroll
    for (int i = 0; i < 13; ++i)
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16* __restrict__ wrow = Wm + (long)(col0 * 16 + fr) * BE_C + fq * 8;          // + j * 16 * BE_C + ks * 32
    const bf16* arow = act + fr * BE_LD + fq * 8;                                        // + i * 16 * BE_LD + ks * 32
    bf16x8 b[BE_AHEAD + 1][NC], a[BE_RING];
#pragma unroll
    for (int s = 0; s < BE_AHEAD; ++s)
#pragma unroll
        for (int j = 0; j < NC; ++j) b[s][j] = *reinterpret_cast<const bf16x8*>(wrow + j * 16 * BE_C + s * 32);
#pragma unroll
    for (int q = 0; q < BE_RING; ++q) a[q] = *reinterpret_cast<const bf16x8*>(arow + q * 16 * BE_LD);
    // (fully unrolled: the ring slots are compile-time, so no fragment is copied between registers, which would wait for its
    // load; the compiler-only fence after each step keeps the loads where they are written: without it the fragments carried
    // across steps are turned into carried pointers and every load sinks to the step that uses it; the scheduling barrier
    // after each row tile keeps the LDS read issued there behind that tile's products instead of bunched at the end of the step)
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
        if (ks + BE_AHEAD < 10) {
#pragma unroll
            for (int j = 0; j < NC; ++j)
                b[(ks + BE_AHEAD) % (BE_AHEAD + 1)][j] = *reinterpret_cast<const bf16x8*>(wrow + j * 16 * BE_C + (ks + BE_AHEAD) * 32);
        }
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const int q = ks * 13 + i, slot = q % BE_RING;                  // the activation fragments in the order they are used
#pragma unroll
            for (int j = 0; j < NC; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ks % (BE_AHEAD + 1)][j], a[slot], acc[i][j], 0, 0, 0);   // D[n 4fq + r][m fr]
            if (q + BE_RING < 130)
                a[slot] = *reinterpret_cast<const bf16x8*>(arow + ((q + BE_RING) % 13) * 16 * BE_LD + ((q + BE_RING) / 13) * 32);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" ::: "memory");
    }
}

// the epilogue of a product: f(value, channel) of every element goes to LDS as bf16 in place of the activations
template <int NC, class F>
__device__ __forceinline__ void be_store(bf16* act, int col0, int fr, int fq, const f32x4 (&acc)[13][3], F f) {
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int m = i * 16 + fr;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int nb = (col0 + j) * 16 + fq * 4;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f(acc[i][j][r], nb + r);
            *reinterpret_cast<bf16x4*>(act + m * BE_LD + nb) = o;
        }
    }
}

// The grouped 3x3 convolution of an image in place, NG groups of 16 channels (g0, g0 + 8, ..) side by side in one wave: their
// chains (LDS read -> nine K = 16 MFMAs -> column shifts -> rounding -> LDS write) are independent, and one chain alone leaves
// the wave waiting for most of a row's time.  lane = image column fr - 1 (a 16-column fragment holds a 14-pixel row plus zero
// borders); every source row is read once, a row ahead of its use, and feeds the accumulators of the three output rows it
// belongs to; raw c2 (bf16) replaces the group's channels a row behind the reads; the squeeze is summed on the way.
template <int NG>
__device__ __forceinline__ void be_conv2(bf16* act, const bf16* __restrict__ wp2, const float* s2, const float* b2, float* gate, int g0,
                                         int fr, int fq, int lane) {
    s16x4 wA[NG][3][3];
    float sc[NG][4], sh[NG][4], ssum[NG][4];
    f32x4 a0[NG][3], a1[NG][3];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int g = g0 + 8 * k;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) wA[k][a][b] = be_tap_frag(wp2, g, a * 3 + b, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) { sc[k][i] = s2[g * 16 + 4 * fq + i]; sh[k][i] = b2[g * 16 + 4 * fq + i]; ssum[k][i] = 0.f; }
#pragma unroll
        for (int x = 0; x < 3; ++x) { a0[k][x] = zero; a1[k][x] = zero; }
    }
    const bool colin = fr >= 1 && fr <= BE_W;
    bf16* lane_base = act + (fr - 1) * BE_LD + g0 * 16 + 4 * fq;            // + k * 128 + row * 14 * BE_LD
    u32x2b nxt[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        nxt[k] = (u32x2b){0u, 0u};
        if (colin) nxt[k] = *reinterpret_cast<const u32x2b*>(lane_base + k * 128);
    }
#pragma unroll 1
    for (int r = 0; r <= BE_W; ++r) {                                       // source row r (row 14: the zero row below the image)
        s16x4 b[NG];
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            b[k] = __builtin_bit_cast(s16x4, nxt[k]);
            nxt[k] = (u32x2b){0u, 0u};
            if (colin && r + 1 < BE_W) nxt[k] = *reinterpret_cast<const u32x2b*>(lane_base + k * 128 + (r + 1) * BE_W * BE_LD);
        }
        f32x4 done[NG][3];
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                done[k][x] = BE_MFMA16(wA[k][2][x], b[k], a0[k][x]);
                a0[k][x] = BE_MFMA16(wA[k][1][x], b[k], a1[k][x]);
                a1[k][x] = BE_MFMA16(wA[k][0][x], b[k], zero);
            }
        if (r >= 1) {                                                       // output row r - 1 is complete; its source row was read two steps ago
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                bf16x4 ob;
#pragma unroll
                for (int i = 0; i < 4; ++i) ob[i] = (bf16)(done[k][1][i] + be_from_next(done[k][2][i]) + be_from_prev(done[k][0][i]));
                if (colin) {
                    *reinterpret_cast<bf16x4*>(lane_base + k * 128 + (r - 1) * BE_W * BE_LD) = ob;
#pragma unroll
                    for (int i = 0; i < 4; ++i) ssum[k][i] += fmaxf((float)ob[i] * sc[k][i] + sh[k][i], 0.f);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NG; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = be_row16_sum(ssum[k][i]) * (1.0f / BE_HW);
            if (fr == 0) gate[(g0 + 8 * k) * 16 + 4 * fq + i] = (float)(bf16)s;   // the squeeze, rounded like the kernel sequence's bf16 vector
        }
}

__global__ __launch_bounds__(BE_THREADS) void block_eval_s3_kernel(const stes_block_eval_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char be_smem[];
    bf16* act = reinterpret_cast<bf16*>(be_smem);                            // [208][BE_LD]
    float* cst = reinterpret_cast<float*>(be_smem + BE_ROWS * BE_LD * 2);    // s1 b1 s2 b2 s3 b3 [320] each
    float* s1 = cst, *b1 = cst + BE_C, *s2 = cst + 2 * BE_C, *b2 = cst + 3 * BE_C, *s3 = cst + 4 * BE_C, *b3 = cst + 5 * BE_C;
    float* gate = cst + 6 * BE_C;                                            // [320]: the squeeze (as bf16 values), then the gate
    float* hid = gate + BE_C;                                                // [80] (+ pad)
    const int tid = threadIdx.x, wave = tid >> 6;
    // the products' 20 column fragments: waves 0-3 take three, waves 4-7 two (a SIMD runs one wave of each kind)
    const bool wide = wave < 4;
    const int col0 = wide ? 5 * wave : 5 * (wave - 4) + 3;
    for (int i = tid; i < BE_C; i += BE_THREADS) {
        s1[i] = p.s1[i]; b1[i] = p.b1[i]; s2[i] = p.s2[i]; b2[i] = p.b2[i]; s3[i] = p.s3[i]; b3[i] = p.b3[i];
    }
    const bf16* __restrict__ X = (const bf16*)p.X;
    const bf16* __restrict__ G = (const bf16*)p.G;
    const int fold = p.fold;                          // 80 with a gated shift in front of conv1, 0 without
    static_assert(BE_THREADS * BE_BATCH >= BE_CHUNKS, "a thread's chunks of an image are one batch");
    // the first image's input rows (the first 80 channels from G, the rest from X) go to LDS in whole 16-byte chunks; the later
    // images' arrive in the pass that writes the image before them out
    if (blockIdx.x < p.N) {
        const long row0 = (long)blockIdx.x * BE_HW;
        const int tv = tid;
        bf16x8 v[BE_BATCH];
#pragma unroll
        for (int u = 0; u < BE_BATCH; ++u) {
            const int idx = min(u * BE_THREADS + tv, BE_CHUNKS - 1);         // (the clamped tail moves the last chunk again)
            const int m = idx / (BE_C / 8), c0 = (idx - m * (BE_C / 8)) * 8;
            const bf16* src = c0 < fold ? G + (row0 + m) * fold + c0 : X + (row0 + m) * BE_C + c0;
            v[u] = *reinterpret_cast<const bf16x8*>(src);
        }
#pragma unroll
        for (int u = 0; u < BE_BATCH; ++u) {
            const int idx = min(u * BE_THREADS + tv, BE_CHUNKS - 1);
            const int m = idx / (BE_C / 8), c0 = (idx - m * (BE_C / 8)) * 8;
            *reinterpret_cast<bf16x8*>(act + m * BE_LD + c0) = v[u];
        }
    }
    __syncthreads();
    for (int n = blockIdx.x; n < p.N; n += gridDim.x) {
        const long row0 = (long)n * BE_HW;
        // (tv: the thread index behind an opaque copy, so that the passes' per-chunk addresses are computed where they are used
        // instead of being hoisted out of the image loop into hundreds of registers)
        int tv = tid;
        asm volatile("" : "+v"(tv));
        const int fr = tv & 15, fq = (tv >> 4) & 3, lane = tv & 63;
        f32x4 acc[13][3];
        // ---------------- phase 1: conv1 from the input rows in LDS; its result replaces them once every wave has finished reading
        {
            const auto bn_relu = [&](float v, int ch) { return (bf16)fmaxf((float)(bf16)v * s1[ch] + b1[ch], 0.f); };
            // (the raw product rounded as the kernel sequence stores it, then its BatchNorm and ReLU)
            if (wide) {
                be_gemm<3>((const bf16*)p.W1, act, col0, fr, fq, acc);
                __syncthreads();
                be_store<3>(act, col0, fr, fq, acc, bn_relu);
            } else {
                be_gemm<2>((const bf16*)p.W1, act, col0, fr, fq, acc);
                __syncthreads();
                be_store<2>(act, col0, fr, fq, acc, bn_relu);
            }
        }
        __syncthreads();
        // ---------------- phase 2: grouped 3x3 convolution in place, squeeze on the way
#pragma unroll 1
        for (int g = wave; g < BE_C / 16; g += BE_THREADS / 64) be_conv2<1>(act, (const bf16*)p.wp2, s2, b2, gate, g, fr, fq, lane);
        __syncthreads();
        // ---------------- phase 3: squeeze-excite FCs, then relu(bn2(c2)) * gate in place
        // hidden[o] = relu(W1[o] . s + b): four threads per output (waves 0-4)
        const bool se_thread = tid < BE_C;                                   // (wave-uniform: 320 = five waves)
        if (se_thread) {
            const int o = tv >> 2, part = tv & 3;
            const bf16* w = (const bf16*)p.Wse1 + (long)o * BE_C + part * 80;
            float h = 0.f;
#pragma unroll
            for (int k = 0; k < 80; k += 8) {
                float f[8];
                load8b(w + k, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) h += f[e] * gate[part * 80 + k + e];
            }
            h += __shfl_xor(h, 1, 64); h += __shfl_xor(h, 2, 64);
            if (part == 0) hid[o] = (float)(bf16)fmaxf(h + p.bse1[o], 0.f);
        }
        __syncthreads();
        if (se_thread) {
            const bf16* w = (const bf16*)p.Wse2 + (long)tv * BE_RD;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < BE_RD; k += 8) {
                float f[8];
                load8b(w + k, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) v += f[e] * hid[k + e];
            }
            v += p.bse2[tv];
            gate[tv] = 1.f / (1.f + __expf(-v));       // (every read of the squeeze in gate[] was before the barrier above)
        }
        __syncthreads();
#pragma unroll 2
        for (int idx = tv; idx < BE_CHUNKS; idx += BE_THREADS) {
            const int m = idx / (BE_C / 8), c0 = (idx - m * (BE_C / 8)) * 8;
            float f[8];
            load8b(act + m * BE_LD + c0, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e] * s2[c0 + e] + b2[c0 + e], 0.f) * gate[c0 + e];
            store8b(act + m * BE_LD + c0, f);
        }
        __syncthreads();
        // ---------------- phase 4: conv3 from LDS, tail in a row-wise pass
        {
            const auto round_only = [](float v, int) { return (bf16)v; };
            if (wide) {
                be_gemm<3>((const bf16*)p.W3, act, col0, fr, fq, acc);
                __syncthreads();                                                 // every wave is done reading the activations
                be_store<3>(act, col0, fr, fq, acc, round_only);
            } else {
                be_gemm<2>((const bf16*)p.W3, act, col0, fr, fq, acc);
                __syncthreads();
                be_store<2>(act, col0, fr, fq, acc, round_only);
            }
        }
        __syncthreads();
        // the pass that writes the image out: bn3 and the shortcut in whole chunks (the arithmetic of bn_act_residual_kernel, in
        // its order); the thread that has taken a chunk of the product out of LDS puts the next image's input chunk in its place,
        // so the reads of the next input are in flight together with the shortcut's reads and the result's writes
        {
            const int nn = n + gridDim.x;
            const bool more = nn < p.N;
            const long row0n = (long)(more ? nn : n) * BE_HW;
            bf16x8 xr[BE_BATCH], nx[BE_BATCH];
#pragma unroll
            for (int u = 0; u < BE_BATCH; ++u) {
                const int idx = min(u * BE_THREADS + tv, BE_CHUNKS - 1);
                xr[u] = *reinterpret_cast<const bf16x8*>(X + row0 * BE_C + (long)idx * 8);
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < BE_BATCH; ++u) {
                    const int idx = min(u * BE_THREADS + tv, BE_CHUNKS - 1);
                    const int m = idx / (BE_C / 8), c0 = (idx - m * (BE_C / 8)) * 8;
                    const bf16* src = c0 < fold ? G + (row0n + m) * fold + c0 : X + (row0n + m) * BE_C + c0;
                    nx[u] = *reinterpret_cast<const bf16x8*>(src);
                }
            }
#pragma unroll
            for (int u = 0; u < BE_BATCH; ++u) {
                const int idx = u * BE_THREADS + tv;
                if (idx >= BE_CHUNKS) continue;
                const int m = idx / (BE_C / 8), c0 = (idx - m * (BE_C / 8)) * 8;
                float f[8];
                load8b(act + m * BE_LD + c0, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = f[e] * s3[c0 + e] + b3[c0 + e];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += (float)xr[u][e];
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)fmaxf(f[e], 0.f);
                *reinterpret_cast<bf16x8*>((bf16*)p.out + row0 * BE_C + (long)idx * 8) = v;
                if (p.slice && c0 < p.slice_n) *reinterpret_cast<bf16x8*>((bf16*)p.slice + (row0 + m) * p.slice_n + c0) = v;
                if (more) *reinterpret_cast<bf16x8*>(act + m * BE_LD + c0) = nx[u];
            }
        }
        __syncthreads();                                                     // the next image's input is in place
    }
}

}  // namespace

extern "C" int stes_block_eval_s3(const stes_block_eval_args* a, void* stream) {
    STES_CHECK(a && a->X && (a->G || a->fold == 0) && a->W1 && a->wp2 && a->Wse1 && a->Wse2 && a->W3 && a->out, STES_ERR_ARG, "stes_block_eval_s3: null pointer");
    STES_CHECK(a->s1 && a->b1 && a->s2 && a->b2 && a->s3 && a->b3 && a->bse1 && a->bse2, STES_ERR_ARG, "stes_block_eval_s3: null pointer");
    STES_CHECK(a->N > 0, STES_ERR_ARG, "stes_block_eval_s3: no images");
    STES_CHECK(a->C == BE_C && a->H == BE_W && a->W == BE_W && a->rd == BE_RD && (a->fold == BE_FOLD || a->fold == 0), STES_ERR_UNSUPPORTED,
               "stes_block_eval_s3: built for 14 x 14 x 320 with an 80-channel squeeze and an 80-channel gated-shift slice (or none)");
    STES_CHECK(!a->slice || (a->slice_n > 0 && a->slice_n % 8 == 0 && a->slice_n <= BE_C), STES_ERR_ARG, "stes_block_eval_s3: slice width");
    static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(&block_eval_s3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    160 * 1024) == hipSuccess;
    STES_CHECK(big_lds, STES_ERR_UNSUPPORTED, "stes_block_eval_s3: cannot raise the dynamic LDS limit");
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) cus = 256;
    const int grid = a->N < cus ? a->N : cus;
    hipLaunchKernelGGL(block_eval_s3_kernel, dim3(grid), dim3(BE_THREADS), BE_LDS, (hipStream_t)stream, *a);
    STES_LAUNCH_CHECK();
    return STES_OK;
}
