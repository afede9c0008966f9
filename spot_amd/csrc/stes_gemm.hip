// bf16 MFMA GEMMs for the 1x1 convolutions and Linear layers of the STES path.
//
// stes_gemm : C[M,N] = act(pro(A) W^T + bias), tall-and-skinny (M = frames*H*W is huge, K and N
//             are channel counts).  128 x BN output tile per workgroup pass, 4 waves (2x2), BK = 32,
//             register-staged double-buffered LDS.  The BatchNorm-apply/ReLU/SE-gate "prologue" runs
//             on A between the global load and the LDS store, so the normalised activations of the
//             previous layer are never written to HBM; BatchNorm batch statistics of the output are
//             reduced in registers over every M tile the workgroup walks and leave as one atomic per
//             column per workgroup.
// stes_wgrad: dW[N1,N2] += P^T pro(Q); the reduction runs over the rows, so both operands are
//             consumed transposed: tiles land in LDS row-major and are fetched with
//             ds_read_b64_tr_b16 straight into MFMA fragment order.
#include <cstdlib>
#include <mutex>
#include "stes_common.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;

// 16-byte chunk swizzle for [rows][32] bf16 tiles (64-byte rows): chunk' = chunk ^ SWZ[(row>>2)&3]
// makes every ds_read_b128 lane group of the MFMA fragment read hit 16 distinct 16-byte slots.
__device__ __forceinline__ int swz(int row, int chunk) {
    const int g = (row >> 2) & 3;
    return chunk ^ ((4 - g) & 3);      // g = 0,1,2,3 -> 0,3,2,1
}

struct RowInfo {
    const bf16* a;       // main source row pointer (nullptr: row out of range -> zeros)
    const bf16* a0;      // compact source row pointer
    const float* gate;   // gate row pointer
};

struct RowDivs { FastDiv hw, wo, gate; };

template <int BN>
__global__ __launch_bounds__(256) void gemm_kernel(const stes_gemm_args p, int m_tiles, int n_tiles, int m_slots, RowDivs dv) {
    constexpr int NI = BN / 32;          // 16-wide n fragments per wave (wave covers BN/2 columns)
    constexpr int BLOADS = BN / 64;      // W-tile chunk loads per thread
    // two k-steps of 32 per barrier (KB = 2 tiles per buffer): the barrier and the LDS round trip are paid once per 32 MFMAs
    constexpr int KB = 2;
    __shared__ __attribute__((aligned(16))) bf16 As[2][KB * BM * BK];
    __shared__ __attribute__((aligned(16))) bf16 Bs[2][KB * BN * BK];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    // block -> (n tile, m slot); the n tiles of one m slot share blockIdx % 8 (one XCD's L2).
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int nt = idx % n_tiles;
    const int slot = (idx / n_tiles) * 8 + xcd;
    if (slot >= m_slots) return;
    const int n0 = nt * BN;

    const bf16* __restrict__ A = (const bf16*)p.A;
    const bf16* __restrict__ A0 = (const bf16*)p.A0;
    const bf16* __restrict__ W = (const bf16*)p.W;
    const int K = p.K, N = p.N;
    const long M = p.M;
    const int nk = (K + KB * BK - 1) / (KB * BK);
    const bool has_affine = p.a_scale != nullptr;
    const bool has_gate = p.a_gate != nullptr;
    const bool pro = has_affine || has_gate;

    const int ld_chunk = tid & 3, ld_row = tid >> 2;   // A/W tile loader coordinates

    float csum[NI][4], csq[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { csum[i][r] = 0.f; csq[i][r] = 0.f; }
    const bool stats = p.col_sum != nullptr;
    const float out_scale = p.out_scale != 0.f ? p.out_scale : 1.f;

    for (int mt = slot; mt < m_tiles; mt += m_slots) {
        const long m0 = (long)mt * BM;
        RowInfo ri[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long m = m0 + ld_row + h * 64;
            ri[h].a = nullptr; ri[h].a0 = nullptr; ri[h].gate = nullptr;
            if (m < M) {
                long src = m;
                if (p.map_stride > 1) {
                    uint32_t n, rem, ho, wo;
                    fdivmod((uint32_t)m, dv.hw, n, rem);
                    fdivmod(rem, dv.wo, ho, wo);
                    src = ((long)n * p.Hi + (long)ho * p.map_stride) * p.Wi + (long)wo * p.map_stride;
                }
                ri[h].a = A + src * p.lda;
                if (A0) ri[h].a0 = A0 + src * p.lda0;
                if (has_gate) ri[h].gate = p.a_gate + (long)fdiv((uint32_t)m, dv.gate) * K;
            }
        }
        const bf16* wrow[BLOADS];
#pragma unroll
        for (int h = 0; h < BLOADS; ++h) {
            const int n = n0 + ld_row + h * 64;
            wrow[h] = (n < N) ? W + (long)n * p.ldw : nullptr;
        }

        f32x4 acc[NI][4];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        bf16x8 ra[KB][2], rb[KB][BLOADS];
        auto gload = [&](int ks) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int k = (ks * KB + kb) * BK + ld_chunk * 8;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (ri[h].a != nullptr && k < K) {
                        const bf16* src = (k < p.k0) ? ri[h].a0 + k : ri[h].a + k;
                        ra[kb][h] = *reinterpret_cast<const bf16x8*>(src);
                    } else {
                        ra[kb][h] = zero8();
                    }
                }
#pragma unroll
                for (int h = 0; h < BLOADS; ++h)
                    rb[kb][h] = (wrow[h] != nullptr && k < K) ? *reinterpret_cast<const bf16x8*>(wrow[h] + k) : zero8();
            }
        };
        auto lstore = [&](int ks, int buf) {
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            const int k = (ks * KB + kb) * BK + ld_chunk * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x8 v = ra[kb][h];
                if (pro && ri[h].a != nullptr && k < K) {
                    float f[8];
                    unpack8(v, f);
                    if (has_affine) {
                        const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.a_scale + k);
                        const f32x4 s1 = *reinterpret_cast<const f32x4*>(p.a_scale + k + 4);
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.a_shift + k);
                        const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.a_shift + k + 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) { f[i] = f[i] * s0[i] + b0[i]; f[i + 4] = f[i + 4] * s1[i] + b1[i]; }
                        if (p.a_relu) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) f[i] = fmaxf(f[i], 0.f);
                        }
                    }
                    if (has_gate) {
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(ri[h].gate + k);
                        const f32x4 g1 = *reinterpret_cast<const f32x4*>(ri[h].gate + k + 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) { f[i] *= g0[i]; f[i + 4] *= g1[i]; }
                    }
                    v = pack8(f);
                }
                const int row = ld_row + h * 64;
                *reinterpret_cast<bf16x8*>(&As[buf][kb * BM * BK + row * BK + swz(row, ld_chunk) * 8]) = v;
            }
#pragma unroll
            for (int h = 0; h < BLOADS; ++h) {
                const int row = ld_row + h * 64;
                *reinterpret_cast<bf16x8*>(&Bs[buf][kb * BN * BK + row * BK + swz(row, ld_chunk) * 8]) = rb[kb][h];
            }
          }
        };

        gload(0);
        lstore(0, 0);
        __syncthreads();
        for (int ks = 0; ks < nk; ++ks) {
            const int buf = ks & 1;
            if (ks + 1 < nk) gload(ks + 1);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                bf16x8 fa[NI], fb[4];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int row = wn * (BN / 2) + i * 16 + fr;
                    fa[i] = *reinterpret_cast<const bf16x8*>(&Bs[buf][kb * BN * BK + row * BK + swz(row, fq) * 8]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = wm * 64 + j * 16 + fr;
                    fb[j] = *reinterpret_cast<const bf16x8*>(&As[buf][kb * BM * BK + row * BK + swz(row, fq) * 8]);
                }
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
            if (ks + 1 < nk) lstore(ks + 1, buf ^ 1);
            __syncthreads();
        }

        // epilogue: lane holds, per (i, j), columns n = n0 + wn*BN/2 + i*16 + fq*4 + r of row m = m0 + wm*64 + j*16 + fr
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int nb = n0 + wn * (BN / 2) + i * 16 + fq * 4;
            float bias[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (nb + r < N) bias[r] = p.bias[nb + r];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long m = m0 + wm * 64 + j * 16 + fr;
                if (m >= M) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[i][j][r] * out_scale + bias[r];
                    if (p.act == 1) v[r] = fmaxf(v[r], 0.f);
                    else if (p.act == 2) v[r] = 1.f / (1.f + __expf(-v[r]));
                }
                if (p.relu_mask) {
                    const bf16* mk = (const bf16*)p.relu_mask + m * p.ldmask + nb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (nb + r < N && !((float)mk[r] > 0.f)) v[r] = 0.f;
                }
                if (p.c_f32) {
                    float* c = (float*)p.C + m * p.ldc + nb;
                    if (nb + 3 < N && (p.ldc & 3) == 0) {
                        *reinterpret_cast<f32x4*>(c) = (f32x4){v[0], v[1], v[2], v[3]};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (nb + r < N) c[r] = v[r];
                    }
                } else {
                    bf16* c = (nb < p.n_split) ? (bf16*)p.C0 + m * p.ldc0 + nb : (bf16*)p.C + m * p.ldc + nb;
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { o[r] = (bf16)v[r]; v[r] = (float)o[r]; }
                    if (nb + 3 < N && (p.ldc & 3) == 0) {
                        *reinterpret_cast<bf16x4*>(c) = o;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (nb + r < N) c[r] = o[r];
                    }
                }
                if (stats) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { csum[i][r] += v[r]; csq[i][r] += v[r] * v[r]; }
                }
            }
        }
    }

    if (stats) {
        // per-wave column sums -> LDS (the A tile is free: the k loop ended on a barrier) -> one contiguous atomic per column
        // per workgroup; scattered few-lane atomics on a handful of cache lines serialise in L2
        float* red = reinterpret_cast<float*>(&As[0][0]);      // [2 (wm)][2 (sum, sumsq)][BN]
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = csum[i][r], q = csq[i][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
                const int nl = wn * (BN / 2) + i * 16 + fq * 4 + r;
                if (fr == 0) { red[(wm * 2 + 0) * BN + nl] = s; red[(wm * 2 + 1) * BN + nl] = q; }
            }
        __syncthreads();
        for (int t = tid; t < 2 * BN; t += 256) {
            const int which = t / BN, nl = t - which * BN;
            if (n0 + nl < N) atomicAdd((which ? p.col_sumsq : p.col_sum) + n0 + nl, red[which * BN + nl] + red[(2 + which) * BN + nl]);
        }
    }
}

// ------------------------------------------------------------------ streaming GEMM for small K, N (weights resident in LDS)
// The tall-and-skinny layers (K <= 320, column slice NB <= 160) are bound by HBM, not by MFMA.  Here the weight slice
// [NB][K] sits in LDS for the workgroup's lifetime and every wave streams its own 32-row tiles with no workgroup barrier:
// A fragments go global -> registers in MFMA operand layout (lane (fr,fq): row fr, k = 8fq..8fq+7: one 16-byte load), four
// k-steps ahead of their use; the BatchNorm/ReLU/gate prologue runs in registers.  The output tile is staged through a
// wave-private LDS buffer so that it leaves as whole rows (1 KiB per store instruction) and so that the BatchNorm
// statistics of the rounded output cost 16 registers per lane (a lane always reads back the same 8 columns).
#ifndef STES_STREAM_DEPTH
#define STES_STREAM_DEPTH 4
#endif
// PRO selects the operand prologue: 0 none or BatchNorm+ReLU (a_scale/a_shift), 1 the same plus the squeeze-excite gate,
// 2 BatchNorm backward: A is the gradient at a BatchNorm's output, a_raw the raw tensor under that BatchNorm, and the operand
// is bf16(alpha*A + beta*a_raw + delta) -- what stes_bn_bwd_apply would have written; it is written here too (a_out, by the
// workgroups of column slice 0) for the weight-gradient kernel, which saves that kernel's pass and one read of its result.
template <int NI, int WAVES, int PRO>   // NB = 16*NI columns per workgroup
__global__ __launch_bounds__(WAVES * 64) void gemm_stream_kernel(const stes_gemm_args p, int row_tiles, int n_slices, int m_slots,
                                                                 RowDivs dv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool GATE = PRO == 1, BNB = PRO == 2;
    constexpr int NB = NI * 16, D = STES_STREAM_DEPTH;
    constexpr int CPR = NB / 8;                 // 16-byte chunks per output row
    constexpr int RPP = 64 / CPR;               // output rows per store pass of a wave
    constexpr int SRS = NB * 2 + 16;            // staging row stride (bytes)
    const int K = p.K, N = p.N, nk = K >> 5;
    const long M = p.M;
    const int RS = 2 * K + 32;                  // weight row stride (bytes): (RS/16) % 4 == 2 keeps the b128 fragment reads conflict-free
    unsigned char* wl = smem;
    unsigned char* stage = smem + NB * RS + (threadIdx.x >> 6) * 16 * SRS;
    float* sc_s = reinterpret_cast<float*>(smem + NB * RS + WAVES * 16 * SRS);
    float* sh_s = sc_s + K;
    float* red = sh_s + K;                      // [2][NB]
    float* de_s = red + 2 * NB;                 // [K], BNB only (sc_s = alpha, sh_s = beta, de_s = delta)
    // GATE only: this wave's copy of the gate rows of the (at most two) images its current tile touches, refreshed once per
    // tile -- read at the point of use from global memory the gate cost a cache round trip in every k-step
    // (it lives in the wave's output staging buffer: the gate is read in the k loop, the staging buffer in the tile epilogue)
    float* gl = reinterpret_cast<float*>(stage);               // [2][K]: 8*K bytes <= 16*SRS since K <= 2*NB + 16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int slice = idx % n_slices;
    const int slot = (idx / n_slices) * 8 + xcd;
    const int n0 = slice * NB;
    const bf16* __restrict__ A = (const bf16*)p.A;
    const bf16* __restrict__ A0 = (const bf16*)p.A0;
    const bf16* __restrict__ R = (const bf16*)p.a_raw;
    const bool has_affine = !BNB && p.a_scale != nullptr;
    const bool stats = p.col_sum != nullptr;
    const float out_scale = p.out_scale != 0.f ? p.out_scale : 1.f;

    {   // weight slice, prologue constants, statistics scratch
        const bf16* __restrict__ W = (const bf16*)p.W;
        const int kch = K >> 3;
        for (int c = tid; c < NB * kch; c += WAVES * 64) {
            const int r = c / kch, kc = c - r * kch;
            const bf16x8 v = (n0 + r < N) ? *reinterpret_cast<const bf16x8*>(W + (long)(n0 + r) * p.ldw + kc * 8) : zero8();
            *reinterpret_cast<bf16x8*>(wl + r * RS + kc * 16) = v;
        }
        if (has_affine) for (int k = tid; k < K; k += WAVES * 64) { sc_s[k] = p.a_scale[k]; sh_s[k] = p.a_shift[k]; }
        if (BNB) for (int k = tid; k < K; k += WAVES * 64) { sc_s[k] = p.bn_alpha[k]; sh_s[k] = p.bn_beta[k]; de_s[k] = p.bn_delta[k]; }
        for (int k = tid; k < 2 * NB; k += WAVES * 64) red[k] = 0.f;
    }
    __syncthreads();
    if (slot >= m_slots) return;

    // ---- cursors: the fetch cursor runs D k-steps ahead of the compute cursor
    const long tstride = (long)m_slots * WAVES;
    long f_tile = (long)slot * WAVES + wave, c_tile = f_tile;
    int f_ks = 0, c_ks = 0;
    const bf16* f_a[2]; const bf16* f_a0[2]; const bf16* f_r[2];
    auto f_rows = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long m = f_tile * 32 + j * 16 + fr;
            f_a[j] = nullptr; f_a0[j] = nullptr; f_r[j] = nullptr;
            if (f_tile < row_tiles && m < M) {
                long src = m;
                if (p.map_stride > 1) {
                    uint32_t n, rem, ho, wo;
                    fdivmod((uint32_t)m, dv.hw, n, rem);
                    fdivmod(rem, dv.wo, ho, wo);
                    src = ((long)n * p.Hi + (long)ho * p.map_stride) * p.Wi + (long)wo * p.map_stride;
                }
                f_a[j] = A + src * p.lda;
                if (A0) f_a0[j] = A0 + src * p.lda0;
                if (BNB) f_r[j] = R + src * p.lda;
            }
        }
    };
    const float* c_gate[2] = {nullptr, nullptr};
    auto c_rows = [&]() {
        if (GATE) {
            if (c_tile >= row_tiles) return;
            long m0 = c_tile * 32, m1 = m0 + 31;
            if (m1 >= M) m1 = M - 1;
            const uint32_t i0 = fdiv((uint32_t)m0, dv.gate), i1 = fdiv((uint32_t)m1, dv.gate);      // i1 <= i0 + 1 (gate_group >= 32)
            __builtin_amdgcn_wave_barrier();                    // the previous tile's reads of gl are done
            for (int c = lane; c < K / 2; c += 64) {            // 2 rows x K floats as float4 chunks
                const int r = c >= K / 4, cc = c - r * (K / 4);
                const f32x4 v = *reinterpret_cast<const f32x4*>(p.a_gate + (long)(r ? i1 : i0) * K + cc * 4);
                *reinterpret_cast<f32x4*>(gl + r * K + cc * 4) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                long m = m0 + j * 16 + fr;
                if (m >= M) m = M - 1;
                c_gate[j] = gl + (fdiv((uint32_t)m, dv.gate) - i0) * K;
            }
        }
    };
    bf16x8 ring[D][2];
    bf16x8 ring_r[BNB ? D : 1][2];
    auto fetch = [&](int d) {
        const int k = f_ks * 32 + fq * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (f_a[j] != nullptr) ring[d][j] = *reinterpret_cast<const bf16x8*>((k < p.k0 ? f_a0[j] : f_a[j]) + k);
            else ring[d][j] = zero8();
            if (BNB) ring_r[d][j] = f_r[j] != nullptr ? *reinterpret_cast<const bf16x8*>(f_r[j] + k) : zero8();
        }
        if (++f_ks == nk) { f_ks = 0; f_tile += tstride; f_rows(); }
    };

    f32x4 acc[NI][2];
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    const int o_chunk = lane % CPR, o_rsub = lane / CPR;        // store phase: this lane's 8 columns and row phase
    const bool o_active = o_rsub < RPP;

    f_rows(); c_rows();
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d);

    while (c_tile < row_tiles) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (c_tile >= row_tiles) break;
            if (c_ks == 0) {
#pragma unroll
                for (int i = 0; i < NI; ++i) { acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            }
            // ---- prologue on the two A fragments of this k-step
            bf16x8 fb[2] = {ring[d][0], ring[d][1]};
            const int k = c_ks * 32 + fq * 8;
            if (BNB) {
                float al[8], be[8], de[8];
                load8f(sc_s + k, al); load8f(sh_s + k, be); load8f(de_s + k, de);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float f[8], r[8];
                    unpack8(fb[j], f);
                    unpack8(ring_r[d][j], r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = al[e] * f[e] + be[e] * r[e] + de[e];
                    fb[j] = pack8(f);
                    const long m = c_tile * 32 + j * 16 + fr;
                    if (p.a_out && slice == 0 && m < M) *reinterpret_cast<bf16x8*>(p.a_out + m * p.lda_out + k) = fb[j];
                }
            }
            if (has_affine || GATE) {
                float s[8], b[8];
                if (has_affine) { load8f(sc_s + k, s); load8f(sh_s + k, b); }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float f[8];
                    unpack8(fb[j], f);
                    if (has_affine) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = f[e] * s[e] + b[e];
                        if (p.a_relu) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
                        }
                    }
                    if (GATE) {
                        float g[8];
                        load8f(c_gate[j] + k, g);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] *= g[e];
                    }
                    fb[j] = pack8(f);
                }
            }
            // refill two slots at a time: consecutive k-steps are the two halves of the same 128-byte lines of A, and issued
            // back to back the second load finds the line in (or on its way to) the vector L1 instead of fetching it again
            if (d & 1) { fetch(d - 1); fetch(d); }
            // ---- MFMA: D[n][m] += W[n][k] A[m][k]
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wl + (i * 16 + fr) * RS + k * 2);
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, fb[0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, fb[1], acc[i][1], 0, 0, 0);
            }
            if (++c_ks < nk) continue;
            // ---- tile done: bias / activation in registers, then through the staging buffer, 16 rows at a time
            c_ks = 0;
            const long m_base = c_tile * 32;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int nb = n0 + i * 16 + fq * 4;
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[i][j][r] * out_scale + (p.bias ? p.bias[nb + r] : 0.f);
                        if (p.act == 1) v = fmaxf(v, 0.f);
                        else if (p.act == 2) v = 1.f / (1.f + __expf(-v));
                        o[r] = (bf16)v;
                    }
                    *reinterpret_cast<bf16x4*>(stage + fr * SRS + (i * 16 + fq * 4) * 2) = o;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (o_active) {
                    const int col = n0 + o_chunk * 8;
#pragma unroll
                    for (int rr = 0; rr < (16 + RPP - 1) / RPP; ++rr) {
                        const int row = rr * RPP + o_rsub;
                        const long m = m_base + j * 16 + row;
                        if (row < 16 && m < M) {
                            bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + row * SRS + o_chunk * 16);
                            if (p.relu_mask) {
                                const bf16x8 mk = *reinterpret_cast<const bf16x8*>((const bf16*)p.relu_mask + m * p.ldmask + col);
#pragma unroll
                                for (int e = 0; e < 8; ++e) if (!((float)mk[e] > 0.f)) v[e] = (bf16)0.f;
                            }
                            bf16* dst = (col < p.n_split) ? (bf16*)p.C0 + m * p.ldc0 + col : (bf16*)p.C + m * p.ldc + col;
                            *reinterpret_cast<bf16x8*>(dst) = v;
                            if (stats) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; ssum[e] += f; ssq[e] += f * f; }
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            c_tile += tstride;
            c_rows();
        }
    }
    if (stats) {
        if (o_active) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { atomicAdd(&red[o_chunk * 8 + e], ssum[e]); atomicAdd(&red[NB + o_chunk * 8 + e], ssq[e]); }
        }
        __syncthreads();
        for (int t = tid; t < 2 * NB; t += WAVES * 64) {
            const int which = t / NB, nl = t - which * NB;
            atomicAdd((which ? p.col_sumsq : p.col_sum) + n0 + nl, red[t]);
        }
    }
}

// ------------------------------------------------------------------ small GEMM (squeeze-excite and classifier matrices)
// A few thousand rows at most and no operand prologue: the product is a few microseconds of latency, not bandwidth.  A
// workgroup owns 16 rows x 64 columns; its four waves split K and read both operands straight from global memory in MFMA
// operand layout (all loads of a pass in flight together), partial sums meet in LDS, wave 0 runs the epilogue.
__global__ __launch_bounds__(256) void gemm_small_kernel(const stes_gemm_args p) {
    constexpr int NI = 4, U = 4;
    __shared__ float red[3][64][NI * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const long M = p.M;
    const int N = p.N, K = p.K;
    const long m0 = (long)blockIdx.x * 16;
    const int n0 = blockIdx.y * 16 * NI;
    const bf16* __restrict__ arow = (m0 + fr < M) ? (const bf16*)p.A + (m0 + fr) * p.lda : nullptr;
    const bf16* __restrict__ wrow[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) wrow[i] = (n0 + i * 16 + fr < N) ? (const bf16*)p.W + (long)(n0 + i * 16 + fr) * p.ldw : nullptr;
    f32x4 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = (K + 31) / 32;
    for (int ks0 = wave; ks0 < nk; ks0 += 4 * U) {
        bf16x8 fa[U], fb[U][NI];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = (ks0 + 4 * u) * 32 + fq * 8;
            const bool on = k < K;                       // K is a multiple of 8: a chunk is inside or outside as a whole
            fa[u] = (on && arow) ? *reinterpret_cast<const bf16x8*>(arow + k) : zero8();
#pragma unroll
            for (int i = 0; i < NI; ++i) fb[u][i] = (on && wrow[i]) ? *reinterpret_cast<const bf16x8*>(wrow[i] + k) : zero8();
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u][i], fa[u], acc[i], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave - 1][lane][i * 4 + r] = acc[i][r];
    }
    __syncthreads();
    if (wave != 0) return;
    // lane holds columns n = n0 + i*16 + fq*4 + r of row m = m0 + fr
    const long m = m0 + fr;
    if (m >= M) return;
    const float out_scale = p.out_scale != 0.f ? p.out_scale : 1.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int nb = n0 + i * 16 + fq * 4;
        if (nb >= N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = (acc[i][r] + red[0][lane][i * 4 + r] + red[1][lane][i * 4 + r] + red[2][lane][i * 4 + r]) * out_scale;
            if (p.bias && nb + r < N) v[r] += p.bias[nb + r];
            if (p.act == 1) v[r] = fmaxf(v[r], 0.f);
            else if (p.act == 2) v[r] = 1.f / (1.f + __expf(-v[r]));
        }
        if (p.relu_mask) {
            const bf16* mk = (const bf16*)p.relu_mask + m * p.ldmask + nb;
#pragma unroll
            for (int r = 0; r < 4; ++r) if (nb + r < N && !((float)mk[r] > 0.f)) v[r] = 0.f;
        }
        if (p.c_f32) {
            float* c = (float*)p.C + m * p.ldc + nb;
            if (nb + 3 < N && (p.ldc & 3) == 0) *reinterpret_cast<f32x4*>(c) = (f32x4){v[0], v[1], v[2], v[3]};
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (nb + r < N) c[r] = v[r];
            }
        } else {
            bf16* c = (bf16*)p.C + m * p.ldc + nb;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
            if (nb + 3 < N && (p.ldc & 3) == 0) *reinterpret_cast<bf16x4*>(c) = o;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (nb + r < N) c[r] = o[r];
            }
        }
    }
}

// ------------------------------------------------------------------ wgrad
__device__ __forceinline__ int rho(int m) {  // LDS row of tile row m: swap bits 2 and 3
    return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1);
}
// within-group channel interleave of the gated shift (reference model/shift.py:124-128)
__device__ __forceinline__ int gsm_dest(int c, int fold) {
    const int half = fold >> 1, quarter = fold >> 2;
    const int g = c / half, rem = c - g * half;
    const int i = rem / quarter, j = rem - i * quarter;
    return g * half + 2 * j + i;
}

// T x T output tile (T = 128 or 64) by WAVES waves arranged 2 x WAVES/2.  Eight waves on the 128 tile halve every wave's
// accumulators and staging registers, so four waves fit a SIMD instead of two: the stage-3/4 shapes wait on memory latency.
template <int T, int WAVES>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(WAVES == 8 ? 4 : 1))) void wgrad_kernel(const stes_wgrad_args p, int m_chunk, int t1, int t2, int m_blocks, RowDivs dv) {
    constexpr int WK = T == 128 ? 64 : 32;   // rows (reduction) per step and barrier: the large tile amortises the barrier
                                             // over two MFMA k-blocks; for the HBM-bound small tile that only costs residency
    constexpr int LDT = T + 16;              // padded row (elements): 32 extra bytes
    constexpr int NT = WAVES * 64;
    constexpr int FI = T / 32;               // 16-wide fragments per wave of P (N1 side)
    constexpr int FJ = T / (8 * WAVES);      // ... of Q (N2 side)
    constexpr int CH = T / 8;                // 16-byte chunks per tile row
    constexpr int LOADS = (WK * CH) / NT;    // chunk loads per thread per operand
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];      // wgrad_lds_bytes<T>()
    typedef bf16 TileBuf[WK * LDT];
    TileBuf* Ps = reinterpret_cast<TileBuf*>(wg_smem);                            // [2]
    TileBuf* Qs = Ps + 2;                                                        // [2]
    float* bias_acc = reinterpret_cast<float*>(Qs + 2);                          // [T]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int W2N = WAVES / 2;
    const int w1 = wave / W2N, w2 = wave % W2N;
    const int fr = lane & 15, fq = lane >> 4;
    // workgroup -> (tile, row chunk): the t1*t2 tiles of a row chunk are consecutive workgroups of ONE XCD (workgroups go
    // round-robin over the 8 XCDs), so a chunk of P and Q comes from HBM once and its other readers hit that XCD's L2
    // (with few row chunks that would leave XCDs idle: then m_blocks < 0 selects chunk-major order over all XCDs)
    int tile, mblk;
    if (m_blocks > 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = idx % (t1 * t2);
        mblk = (idx / (t1 * t2)) * 8 + xcd;
    } else {
        m_blocks = -m_blocks;
        mblk = blockIdx.x % m_blocks;
        tile = blockIdx.x / m_blocks;
    }
    const int by = tile % t1, bz = tile / t1;
    const int n1_0 = by * T, n2_0 = bz * T;
    const long m_begin = (long)mblk * m_chunk;
    const long m_end = (m_begin + m_chunk < p.M) ? m_begin + m_chunk : p.M;
    if (mblk >= m_blocks || m_begin >= m_end) return;

    const bf16* __restrict__ P = (const bf16*)p.P;
    const bf16* __restrict__ Q = (const bf16*)p.Q;
    const bf16* __restrict__ Q0 = (const bf16*)p.Q0;
    const bool has_affine = p.q_scale != nullptr, has_gate = p.q_gate != nullptr;
    const bool do_bias = p.dbias != nullptr && bz == 0;
    if (do_bias && tid < T) bias_acc[tid] = 0.f;

    const int lc = tid % CH, lr = tid / CH;          // loader: chunk, row (row step NT/CH)
    constexpr int RSTEP = NT / CH;
    const int c1 = n1_0 + lc * 8, c2 = n2_0 + lc * 8;
    const bool c1_ok = c1 < p.N1, c2_ok = c2 < p.N2;

    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bsum[i] = 0.f;

    // BatchNorm-backward prologue on P (small tile only: its 24 coefficient registers do not fit the capped 128-tile)
    const bf16* __restrict__ PR = T == 64 ? (const bf16*)p.p_raw : nullptr;
    float pal[8], pbe[8], pde[8];
    if (T == 64) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool on = PR != nullptr && c1_ok;
            pal[i] = on ? p.bn_alpha[c1 + i] : 1.f; pbe[i] = on ? p.bn_beta[c1 + i] : 0.f; pde[i] = on ? p.bn_delta[c1 + i] : 0.f;
        }
    }
    float qsc[8], qsh[8];      // this thread's 8 columns of the prologue's scale / shift
#pragma unroll
    for (int i = 0; i < 8; ++i) { qsc[i] = (has_affine && c2_ok) ? p.q_scale[c2 + i] : 1.f; qsh[i] = (has_affine && c2_ok) ? p.q_shift[c2 + i] : 0.f; }
    bf16x8 rp[LOADS], rq[LOADS], rr[T == 64 ? LOADS : 1];
    long qm[LOADS];   // logical row of each staged q chunk (for the gate lookup)
    auto gload = [&](long mb) {
#pragma unroll
        for (int h = 0; h < LOADS; ++h) {
            const long m = mb + lr + h * RSTEP;
            qm[h] = m;
            if (m < m_end) {
                rp[h] = c1_ok ? *reinterpret_cast<const bf16x8*>(P + m * p.ldp + c1) : zero8();
                if (T == 64 && PR) rr[h] = c1_ok ? *reinterpret_cast<const bf16x8*>(PR + m * p.ldp + c1) : zero8();
                long src = m;
                if (p.map_stride > 1) {
                    uint32_t n, rem, ho, wo;
                    fdivmod((uint32_t)m, dv.hw, n, rem);
                    fdivmod(rem, dv.wo, ho, wo);
                    src = ((long)n * p.Hi + (long)ho * p.map_stride) * p.Wi + (long)wo * p.map_stride;
                }
                if (c2_ok) rq[h] = (c2 < p.k0) ? *reinterpret_cast<const bf16x8*>(Q0 + src * p.ldq0 + c2)
                                               : *reinterpret_cast<const bf16x8*>(Q + src * p.ldq + c2);
                else rq[h] = zero8();
            } else {
                rp[h] = zero8(); rq[h] = zero8();
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < LOADS; ++h) {
            const int row = lr + h * RSTEP;
            bf16x8 v = rq[h];
            if ((has_affine || has_gate) && qm[h] < m_end && c2_ok) {
                float f[8];
                unpack8(v, f);
                if (has_affine) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        f[i] = f[i] * qsc[i] + qsh[i];
                        if (p.q_relu) f[i] = fmaxf(f[i], 0.f);
                    }
                }
                if (has_gate) {
                    const float* g = p.q_gate + (long)fdiv((uint32_t)qm[h], dv.gate) * p.N2 + c2;
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] *= g[i];
                }
                v = pack8(f);
            }
            *reinterpret_cast<bf16x8*>(&Qs[buf][rho(row) * LDT + lc * 8]) = v;
            if (T == 64 && PR && qm[h] < m_end && c1_ok) {
                float f[8], r[8];
                unpack8(rp[h], f);
                unpack8(rr[h], r);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] = pal[i] * f[i] + pbe[i] * r[i] + pde[i];
                rp[h] = pack8(f);
            }
            *reinterpret_cast<bf16x8*>(&Ps[buf][rho(row) * LDT + lc * 8]) = rp[h];
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 8; ++i) bsum[i] += (float)rp[h][i];
            }
        }
    };

    gload(m_begin);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (long mb = m_begin; mb < m_end; mb += WK) {
        const bool more = mb + WK < m_end;
        if (more) gload(mb + WK);
        // transposed fragment reads: lane supplies row 8*fq + (fr>>2) (+4) of a 32-row k-block, columns 4*(fr&3)..+3
        const int r_lo = rho(8 * fq + (fr >> 2)), r_hi = rho(8 * fq + (fr >> 2) + 4);
        const int cc = 4 * (fr & 3);
#pragma unroll
        for (int kk = 0; kk < WK / 32; ++kk) {
            bf16x8 fa[FI], fb[FJ];
            const bf16* ps = &Ps[buf][kk * 32 * LDT];
            const bf16* qs = &Qs[buf][kk * 32 * LDT];
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int col = w1 * (T / 2) + i * 16 + cc;
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(ps + r_lo * LDT + col));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(ps + r_hi * LDT + col));
                fa[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < FJ; ++j) {
                const int col = w2 * (T / W2N) + j * 16 + cc;
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(qs + r_lo * LDT + col));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(qs + r_hi * LDT + col));
                fb[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // D[row = n1 (fq*4 + r)][col = n2 (fr)]
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
            const int n2 = n2_0 + w2 * (T / W2N) + j * 16 + fr;
            const int n2d = n2 < p.perm_fold ? gsm_dest(n2, p.perm_fold) : n2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n1 = n1_0 + w1 * (T / 2) + i * 16 + fq * 4 + r;
                if (n1 < p.N1 && n2 < p.N2) atomicAdd(p.dW + (long)n1 * p.lddw + n2d, acc[i][j][r]);
            }
        }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&bias_acc[lc * 8 + i], bsum[i]);
        __syncthreads();
        if (tid < T && n1_0 + tid < p.N1) atomicAdd(p.dbias + n1_0 + tid, bias_acc[tid]);
    }
}

template <int T, int WAVES>
static size_t wgrad_lds_bytes() {
    static const bool raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   160 * 1024) == hipSuccess;
    (void)raised;
    return (size_t)4 * (T == 128 ? 64 : 32) * (T + 16) * sizeof(bf16) + T * sizeof(float);
}

// Workgroups of one streaming-GEMM instantiation resident per CU at a given dynamic LDS size (cached: queried once per size).
template <int NI, int WAVES, int PRO>
static int stream_residency(size_t lds) {
    static std::mutex mu;                          // entry points may be called from any host thread (autograd runs backward on its own)
    std::lock_guard<std::mutex> lock(mu);
    static bool attr_done = false;
    static size_t seen[8];
    static int val[8];
    static int n_seen = 0;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_kernel<NI, WAVES, PRO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    for (int i = 0; i < n_seen; ++i) if (seen[i] == lds) return val[i];
    int per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_stream_kernel<NI, WAVES, PRO>, WAVES * 64, lds) != hipSuccess || per_cu < 1)
        per_cu = 1;
    if (n_seen < 8) { seen[n_seen] = lds; val[n_seen] = per_cu; ++n_seen; }
    return per_cu;
}

}  // namespace

extern "C" int stes_gemm(const stes_gemm_args* a, void* stream) {
    STES_CHECK(a && a->A && a->W && a->C, STES_ERR_ARG, "stes_gemm: null pointer");
    STES_CHECK(a->M > 0 && a->N > 0 && a->K > 0, STES_ERR_ARG, "stes_gemm: empty shape");
    STES_CHECK(a->K % 8 == 0 && a->lda % 8 == 0 && a->ldw % 8 == 0, STES_ERR_ARG, "stes_gemm: K, lda, ldw must be multiples of 8");
    STES_CHECK(a->k0 == 0 || (a->A0 && a->k0 % 8 == 0 && a->lda0 % 8 == 0 && a->k0 <= a->K), STES_ERR_ARG, "stes_gemm: bad compact source");
    STES_CHECK((a->a_scale == nullptr) == (a->a_shift == nullptr), STES_ERR_ARG, "stes_gemm: scale/shift must come together");
    STES_CHECK(!a->a_gate || a->gate_group > 0, STES_ERR_ARG, "stes_gemm: gate_group");
    STES_CHECK(a->map_stride <= 1 || (a->Ho > 0 && a->Wo > 0 && a->Hi > 0 && a->Wi > 0), STES_ERR_ARG, "stes_gemm: row map dims");
    STES_CHECK((a->col_sum == nullptr) == (a->col_sumsq == nullptr), STES_ERR_ARG, "stes_gemm: stats pointers");
    STES_CHECK(!(a->col_sum && a->c_f32), STES_ERR_ARG, "stes_gemm: stats need bf16 output");
    STES_CHECK(a->n_split == 0 || (a->C0 && !a->c_f32 && a->n_split % 4 == 0 && a->ldc0 % 4 == 0 && a->ldc0 >= a->n_split), STES_ERR_ARG, "stes_gemm: split output");
    STES_CHECK(a->M <= STES_MAX_ROWS, STES_ERR_UNSUPPORTED, "stes_gemm: more than 2^31 rows");
    const bool bnb = a->a_raw != nullptr;
    STES_CHECK(!bnb || (a->bn_alpha && a->bn_beta && a->bn_delta && !a->a_scale && !a->a_gate && a->k0 == 0 && a->map_stride <= 1 &&
                        (!a->a_out || (a->lda_out % 8 == 0 && a->lda_out >= a->K))),
               STES_ERR_ARG, "stes_gemm: the BatchNorm-backward prologue needs alpha/beta/delta and excludes the other prologues");
    STES_CHECK(bnb || !a->a_out, STES_ERR_ARG, "stes_gemm: a_out without a_raw");
    RowDivs dv;
    dv.hw = make_fastdiv(a->map_stride > 1 ? a->Ho * a->Wo : 1);
    dv.wo = make_fastdiv(a->map_stride > 1 ? a->Wo : 1);
    dv.gate = make_fastdiv(a->a_gate ? a->gate_group : 1);
    // tall-and-skinny bf16 layers with small K and N: weights resident in LDS, rows streamed (gemm_stream_kernel)
    {
        int nb = 0;
        if (a->N == 32 || a->N == 64 || a->N == 128 || a->N == 160) nb = a->N;
        else if (a->N == 320) nb = 160;
        else if (a->N % 128 == 0 && a->N <= 1024) nb = 128;     // wide layers with a short K (stage 4's first 1x1 conv): 128-column slices
        const bool fits = nb > 0 && a->K % 32 == 0 && a->K <= 320 && !a->c_f32 && a->M >= 8192 && a->ldc % 8 == 0 &&
                          (a->k0 == 0 || a->k0 % 8 == 0) && (a->n_split == 0 || (a->n_split % 8 == 0 && a->ldc0 % 8 == 0)) &&
                          (!a->relu_mask || a->ldmask % 8 == 0) && (!a->a_gate || (nb >= 64 && a->K % 8 == 0 && a->gate_group >= 32 && a->K <= 4 * nb + 32));
        if (fits) {
            const int ni = nb / 16, waves = nb >= 128 ? 8 : 4;
            const size_t lds = (size_t)nb * (2 * a->K + 32) + (size_t)waves * 16 * (nb * 2 + 16) +
                               (size_t)((bnb ? 3 : 2) * a->K + 2 * nb) * sizeof(float);
            const int n_slices = a->N / nb;
            const int row_tiles = stes_div_up(a->M, 32);
            int per_cu = 1;
#define STREAM_CASE(NI_, W_, G_)                                                                                                   \
    do {                                                                                                                           \
        per_cu = stream_residency<NI_, W_, G_>(lds);                                                                               \
        int m_slots = stes_div_up(row_tiles, W_);                                                                                  \
        const int cap = 256 * per_cu / n_slices > 8 ? 256 * per_cu / n_slices : 8;                                                 \
        if (m_slots > cap) m_slots = cap;                                                                                          \
        const int grid = stes_div_up(m_slots, 8) * n_slices * 8;                                                                   \
        hipLaunchKernelGGL((gemm_stream_kernel<NI_, W_, G_>), dim3(grid), dim3(W_ * 64), lds, (hipStream_t)stream, *a, row_tiles,   \
                           n_slices, m_slots, dv);                                                                                 \
    } while (0)
            const int pro = bnb ? 2 : (a->a_gate != nullptr ? 1 : 0);
            if (ni == 2) { if (pro == 2) STREAM_CASE(2, 4, 2); else STREAM_CASE(2, 4, 0); }
            else if (ni == 4) { if (pro == 2) STREAM_CASE(4, 4, 2); else if (pro == 1) STREAM_CASE(4, 4, 1); else STREAM_CASE(4, 4, 0); }
            else if (ni == 8) { if (pro == 2) STREAM_CASE(8, 8, 2); else if (pro == 1) STREAM_CASE(8, 8, 1); else STREAM_CASE(8, 8, 0); }
            else { if (pro == 2) STREAM_CASE(10, 8, 2); else if (pro == 1) STREAM_CASE(10, 8, 1); else STREAM_CASE(10, 8, 0); }
#undef STREAM_CASE
            STES_LAUNCH_CHECK();
            return STES_OK;
        }
    }
    STES_CHECK(!bnb, STES_ERR_UNSUPPORTED, "stes_gemm: the BatchNorm-backward prologue exists for the streaming shapes only "
                                           "(N in {32,64,128,160,320} or a multiple of 128 up to 1024, K % 32 == 0, K <= 320, M >= 8192, bf16 output); run stes_bn_bwd_apply first");
    // no prologue and at most 2^28 multiply-adds: the latency-bound small kernel
    if (!a->a_scale && !a->a_gate && a->k0 == 0 && a->map_stride <= 1 && !a->col_sum && a->n_split == 0 &&
        (double)a->M * a->N * a->K <= (double)(1L << 28) && a->M <= 16L * 65535) {
        hipLaunchKernelGGL(gemm_small_kernel, dim3((unsigned)stes_div_up(a->M, 16), (unsigned)stes_div_up(a->N, 64)), dim3(256), 0,
                           (hipStream_t)stream, *a);
        STES_LAUNCH_CHECK();
        return STES_OK;
    }
    const int m_tiles = stes_div_up(a->M, BM);
    // fewer than 300 128-wide tiles (the head and squeeze-excite matrices: 1,024 rows) leave CUs idle: 64-wide tiles double
    // the workgroups (measured 20-30 % faster on those shapes)
    // (and a width like 320 = 5 x 64 wastes a sixth of 3 x 128: 5-20 % faster on 64-wide tiles)
    const bool wide = a->N > 64 && (long)m_tiles * stes_div_up(a->N, 128) >= 300 && !(a->N % 128 != 0 && a->N % 64 == 0);
    const int bn = wide ? 128 : 64;
    const int n_tiles = stes_div_up(a->N, bn);
    int m_slots = m_tiles;
    const int cap = 2048 / n_tiles > 8 ? 2048 / n_tiles : 8;
    if (m_slots > cap) m_slots = cap;
    const int slot_groups = stes_div_up(m_slots, 8);
    const int grid = slot_groups * n_tiles * 8;
    if (wide) hipLaunchKernelGGL(gemm_kernel<128>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a, m_tiles, n_tiles, m_slots, dv);
    else hipLaunchKernelGGL(gemm_kernel<64>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a, m_tiles, n_tiles, m_slots, dv);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_wgrad(const stes_wgrad_args* a, void* stream) {
    STES_CHECK(a && a->P && a->Q && a->dW, STES_ERR_ARG, "stes_wgrad: null pointer");
    STES_CHECK(a->M > 0 && a->N1 > 0 && a->N2 > 0, STES_ERR_ARG, "stes_wgrad: empty shape");
    STES_CHECK(a->ldp % 8 == 0 && a->ldq % 8 == 0 && a->ldq >= (a->N2 + 7) / 8 * 8, STES_ERR_ARG, "stes_wgrad: ldp, ldq must be multiples of 8 and cover N2 rounded up to 8");
    STES_CHECK(a->N2 % 8 == 0 || (!a->q_scale && !a->q_gate), STES_ERR_ARG, "stes_wgrad: a prologue needs N2 % 8 == 0");
    STES_CHECK(a->k0 == 0 || (a->Q0 && a->k0 % 8 == 0 && a->ldq0 % 8 == 0), STES_ERR_ARG, "stes_wgrad: bad compact source");
    STES_CHECK(a->ldp >= (a->N1 + 7) / 8 * 8, STES_ERR_ARG, "stes_wgrad: ldp must cover N1 rounded up to 8 (pad the operand)");
    const bool big = a->N1 > 64 && a->N2 > 64;
    STES_CHECK(!a->p_raw || (a->bn_alpha && a->bn_beta && a->bn_delta), STES_ERR_ARG, "stes_wgrad: the P prologue needs alpha/beta/delta");
    STES_CHECK(!a->p_raw || !big, STES_ERR_UNSUPPORTED, "stes_wgrad: the P prologue exists for the 64-wide tile only (N1 <= 64 or N2 <= 64)");
    const int t = big ? 128 : 64;
    const int t1 = stes_div_up(a->N1, t), t2 = stes_div_up(a->N2, t);
    // Split the rows so that the grid has about `total` workgroups, at least 256 rows each.  The 128-tile holds two
    // workgroups per CU (LDS), and 512 of them -- one resident set -- beat finer splits (every extra chunk adds a tile of
    // atomics) and coarser ones (one workgroup per CU hides no latency); with one to three tiles and short chunks, or the
    // many tiles of the head matrices, one workgroup per CU is enough.  The 64-tile is HBM-bound and likes more, smaller ones.
    const int nt = t1 * t2;
    long total = 1024;
    if (big) total = (nt >= 64 || (nt <= 3 && a->M * nt <= (1L << 20))) ? 256 : 512;
    long want = total / nt;
    if (want < 1) want = 1;
    long chunk = (a->M + want - 1) / want;
    if (chunk < 256) chunk = 256;
    chunk = (chunk + 63) / 64 * 64;
    const int mb = stes_div_up(a->M, chunk);
    STES_CHECK(a->M <= STES_MAX_ROWS, STES_ERR_UNSUPPORTED, "stes_wgrad: more than 2^31 rows");
    RowDivs dv;
    dv.hw = make_fastdiv(a->map_stride > 1 ? a->Ho * a->Wo : 1);
    dv.wo = make_fastdiv(a->map_stride > 1 ? a->Wo : 1);
    dv.gate = make_fastdiv(a->q_gate ? a->gate_group : 1);
    // several tiles re-reading each row chunk and enough chunks to balance 8 XCDs: tiles of a chunk share an XCD
    const bool xcd_map = t1 * t2 >= 4 && mb >= 48;   // (measured: with 3 tiles or fewer the chunk-major order is faster)
    const int grid = xcd_map ? (mb + 7) / 8 * 8 * t1 * t2 : mb * t1 * t2;
    const int mb_arg = xcd_map ? mb : -mb;
    if (big) {
        const size_t lds = wgrad_lds_bytes<128, 8>();
        hipLaunchKernelGGL((wgrad_kernel<128, 8>), dim3(grid), dim3(512), lds, (hipStream_t)stream, *a, (int)chunk, t1, t2, mb_arg, dv);
    } else {
        const size_t lds = wgrad_lds_bytes<64, 4>();
        hipLaunchKernelGGL((wgrad_kernel<64, 4>), dim3(grid), dim3(256), lds, (hipStream_t)stream, *a, (int)chunk, t1, t2, mb_arg, dv);
    }
    STES_LAUNCH_CHECK();
    return STES_OK;
}
