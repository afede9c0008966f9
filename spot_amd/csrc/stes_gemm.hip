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
#include "stes_gemm_stream.h"

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
__global__ __launch_bounds__(256, BN == 128 ? 2 : 3) void gemm_kernel(const stes_gemm_args p, int m_tiles, int n_tiles, int m_slots, RowDivs dv, StesDet det) {
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
    if (slot >= m_slots) { det_enter(det); det_leave(det); return; }     // (deterministic mode: the turn still has to pass)
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
    // Row-store epilogue (full bf16 column tiles): a wave's 64 x BN/2 result goes through its own slice of the A buffer and
    // leaves as 16-byte pieces that make up whole row segments.  Stored straight from the accumulator layout a store
    // instruction covers 16 rows x 32 bytes, and those partial lines cost as much time as the k loop at K = 768.
    constexpr int WN = BN / 2, CPR = WN / 8;            // columns per wave, 16-byte pieces per row
    const bool row_store = !p.c_f32 && !p.relu_mask && (p.ldc & 7) == 0 && n0 + BN <= N &&
                           (p.n_split == 0 || ((p.n_split & 7) == 0 && (p.ldc0 & 7) == 0));
    // (its column sums live in csum[0..1][] / csq[0..1][]: a lane always reads back the same 8 columns)

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
        if (row_store) {
            // (the k loop ended on a barrier: nobody reads the operand buffers any more)
            unsigned char* stg = reinterpret_cast<unsigned char*>(&As[0][0]) + wave * (64 * WN * 2);
            auto piece = [](int row, int c) { return c ^ (CPR == 8 ? (row & 7) : ((row >> 1) & 3)); };   // bank spread
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int nb = n0 + wn * WN + i * 16 + fq * 4;
                float bias[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bias[r] = p.bias[nb + r];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[i][j][r] * out_scale + bias[r];
                        if (p.act == 1) v = fmaxf(v, 0.f);
                        else if (p.act == 2) v = 1.f / (1.f + __expf(-v));
                        o[r] = (bf16)v;
                    }
                    const int row = j * 16 + fr;
                    *reinterpret_cast<bf16x4*>(stg + row * (WN * 2) + piece(row, i * 2 + (fq >> 1)) * 16 + (fq & 1) * 8) = o;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int c = lane % CPR, rsub = lane / CPR;
            const int col = n0 + wn * WN + c * 8;
            const bool to_c0 = col < p.n_split;
            bf16* const o_base = to_c0 ? (bf16*)p.C0 + col : (bf16*)p.C + col;
            const long o_ld = to_c0 ? p.ldc0 : p.ldc;
#pragma unroll
            for (int ps = 0; ps < 64 / (64 / CPR); ++ps) {
                const int row = ps * (64 / CPR) + rsub;
                const long m = m0 + wm * 64 + row;
                if (m < M) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * (WN * 2) + piece(row, c) * 16);
                    *reinterpret_cast<bf16x8*>(o_base + m * o_ld) = v;
                    if (stats) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; csum[e >> 2][e & 3] += f; csq[e >> 2][e & 3] += f * f; }
                    }
                }
            }
            __syncthreads();                  // the next tile's operands go into the same buffer
            continue;
        }
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
        if (row_store) {
            // lanes with the same lane % CPR hold the same 8 columns
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float sv = csum[e >> 2][e & 3], q = csq[e >> 2][e & 3];
#pragma unroll
                for (int o = CPR; o < 64; o <<= 1) { sv += __shfl_xor(sv, o, 64); q += __shfl_xor(q, o, 64); }
                const int nl = wn * WN + (lane % CPR) * 8 + e;
                if (lane < CPR) { red[(wm * 2 + 0) * BN + nl] = sv; red[(wm * 2 + 1) * BN + nl] = q; }
            }
        } else
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
        det_enter(det);
        for (int t = tid; t < 2 * BN; t += 256) {
            const int which = t / BN, nl = t - which * BN;
            if (n0 + nl < N) atomicAdd((which ? p.col_sumsq : p.col_sum) + n0 + nl, red[which * BN + nl] + red[(2 + which) * BN + nl]);
        }
        det_leave(det);
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

// ------------------------------------------------------------------ two chained small GEMMs (squeeze-excite FC pair)
// Workgroup = 16 rows of both products.  Product 1: the four waves take the column fragments of H round-robin, each over the
// whole K1, operands straight from global memory in MFMA layout (the weight rows as the A operand, so a lane ends up with four
// consecutive columns of one row); H goes to memory and, as bf16, to LDS.  Product 2 reads H from LDS and W2 from memory the
// same way.  Every wave's loads of a pass are in flight together; no cross-wave reduction.
__device__ __forceinline__ float small_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.f / (1.f + __expf(-v));
    return v;
}

__global__ __launch_bounds__(256) void gemm_small2_kernel(const stes_gemm2_args p) {
    constexpr int HS = 256 + 8;                                   // LDS row stride of H (elements): 16-byte rows, bank-shifted
    __shared__ __attribute__((aligned(16))) bf16 Hs[16 * HS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const long m0 = (long)blockIdx.x * 16, m = m0 + fr;
    const bool row_ok = m < p.M;
    const bf16* __restrict__ arow = (const bf16*)p.A + (row_ok ? m : 0) * p.lda;
    // zero padding of H up to a multiple of 32 columns (the second product's k-steps)
    const int n1p = (p.N1 + 31) & ~31;
    for (int i = threadIdx.x; i < 16 * (n1p - p.N1); i += 256) Hs[(i / (n1p - p.N1)) * HS + p.N1 + i % (n1p - p.N1)] = (bf16)0.f;
    // ---- product 1
    const int nf1 = (p.N1 + 15) >> 4, nk1 = (p.K1 + 31) >> 5;
    for (int f = wave; f < nf1; f += 4) {
        const int n = f * 16 + fr;
        const bf16* __restrict__ wrow = (const bf16*)p.W1 + (long)(n < p.N1 ? n : 0) * p.ldw1;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int ks0 = 0; ks0 < nk1; ks0 += 8) {
            bf16x8 fa[8], fb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = (ks0 + u) * 32 + fq * 8;
                const bool on = k < p.K1;                           // K1 is a multiple of 8: a chunk is inside or outside as a whole
                fa[u] = (on && row_ok) ? *reinterpret_cast<const bf16x8*>(arow + k) : zero8();
                fb[u] = (on && n < p.N1) ? *reinterpret_cast<const bf16x8*>(wrow + k) : zero8();
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u], fa[u], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u], fa[u], acc0, 0, 0, 0);
            }
        }
        // lane holds columns nb..nb+3 of row m
        const int nb = f * 16 + fq * 4;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc0[r] + acc1[r];
            if (p.bias1 && nb + r < p.N1) v += p.bias1[nb + r];
            v = small_act(v, p.act1);
            if (p.relu_mask1 && row_ok && nb + r < p.N1 && !((float)((const bf16*)p.relu_mask1)[m * p.ldmask1 + nb + r] > 0.f)) v = 0.f;
            o[r] = (bf16)((nb + r < p.N1) ? v : 0.f);
        }
        *reinterpret_cast<bf16x4*>(&Hs[fr * HS + nb]) = o;
        if (row_ok && p.H) {
            bf16* h = (bf16*)p.H + m * p.ldh + nb;
            if (nb + 3 < p.N1 && (p.ldh & 3) == 0) *reinterpret_cast<bf16x4*>(h) = o;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (nb + r < p.N1) h[r] = o[r];
            }
        }
    }
    __syncthreads();
    // ---- product 2
    const int nf2 = (p.N2 + 15) >> 4, nk2 = n1p >> 5;                // at most 8 k-steps
    bf16x8 ha[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ha[u] = u < nk2 ? *reinterpret_cast<const bf16x8*>(&Hs[fr * HS + u * 32 + fq * 8]) : zero8();
    const float out_scale = p.out_scale != 0.f ? p.out_scale : 1.f;
    for (int f0 = wave; f0 < nf2; f0 += 8) {
        // two column fragments at a time: their loads are in flight together
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        bf16x8 fb[2][8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = (f0 + 4 * q) * 16 + fr;
            const bf16* __restrict__ wrow = (const bf16*)p.W2 + (long)(n < p.N2 ? n : 0) * p.ldw2;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = u * 32 + fq * 8;
                fb[q][u] = (u < nk2 && k < p.N1 && n < p.N2 && f0 + 4 * q < nf2) ? *reinterpret_cast<const bf16x8*>(wrow + k) : zero8();
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[q][u], ha[u], acc[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = f0 + 4 * q;
            const int nb = f * 16 + fq * 4;
            if (f >= nf2 || !row_ok || nb >= p.N2) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[q][r];
                if (p.bias2 && nb + r < p.N2) v[r] += p.bias2[nb + r];
                v[r] = small_act(v[r], p.act2) * out_scale;
            }
            if (p.c_f32) {
                float* c = (float*)p.C + m * p.ldc + nb;
                if (nb + 3 < p.N2 && (p.ldc & 3) == 0) *reinterpret_cast<f32x4*>(c) = (f32x4){v[0], v[1], v[2], v[3]};
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (nb + r < p.N2) c[r] = v[r];
                }
            } else {
                bf16* c = (bf16*)p.C + m * p.ldc + nb;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
                if (nb + 3 < p.N2 && (p.ldc & 3) == 0) *reinterpret_cast<bf16x4*>(c) = o;
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (nb + r < p.N2) c[r] = o[r];
                }
            }
        }
    }
}

// The same pair of products with every operand brought into LDS first.  The kernel above walks its operands in dependent
// rounds of loads (the k-batches of a column fragment, then the fragments of the second product: seven memory round trips
// for the stage-3 shapes), and a round trip is what the pair costs: 15 us alone, 40 us beside the weight-gradient stream.
// Here the 16 rows of A, W1 and W2 (up to 128 KiB together) are requested at once -- 512 threads, at most 16 chunks of 16
// bytes each, all in flight together, with the bias and mask values of the epilogues behind them -- and land in LDS rows
// padded by 16 bytes; both products then read MFMA operands from LDS.  One round trip.  Sums are taken in the order of the
// kernel above (even and odd k-steps of the first product in two accumulators), so both forms give the same bits.
constexpr int S2_THREADS = 512, S2_MAXC = 16;
struct Small2Lds {
    int sa, sw2;                    // LDS row strides (elements) of A / W1 rows and of W2 / H rows
    int n1r, n2r;                   // W1 and W2 rows rounded up to whole fragments
    int c_a, c_w1, c_all;           // chunk counts: A, A + W1, all
    FastDiv fd_kc1, fd_kc2;         // chunks per row of A / W1 and of W2
};

__global__ __launch_bounds__(S2_THREADS) void gemm_small2_lds_kernel(const stes_gemm2_args p, const Small2Lds g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s2_smem[];
    bf16* As = reinterpret_cast<bf16*>(s2_smem);                   // [16][sa]
    bf16* W1s = As + 16 * g.sa;                                    // [n1r][sa]
    bf16* W2s = W1s + g.n1r * g.sa;                                // [n2r][sw2]
    bf16* Hs = W2s + g.n2r * g.sw2;                                // [16][sw2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const long m0 = (long)blockIdx.x * 16, m = m0 + fr;
    const bool row_ok = m < p.M;
    // ---- every operand chunk of the workgroup, requested together
    bf16x8 regs[S2_MAXC];
    int at[S2_MAXC];                                               // LDS element offset, -1: no chunk
#pragma unroll
    for (int k = 0; k < S2_MAXC; ++k) {
        const int i = tid + k * S2_THREADS;
        at[k] = -1;
        if (i < g.c_all) {
            uint32_t r, c;
            if (i < g.c_a) {
                fdivmod((uint32_t)i, g.fd_kc1, r, c);
                if (m0 + r < p.M) { regs[k] = *reinterpret_cast<const bf16x8*>((const bf16*)p.A + (m0 + r) * p.lda + c * 8); at[k] = (int)(r * g.sa + c * 8); }
            } else if (i < g.c_w1) {
                fdivmod((uint32_t)(i - g.c_a), g.fd_kc1, r, c);
                regs[k] = *reinterpret_cast<const bf16x8*>((const bf16*)p.W1 + (long)r * p.ldw1 + c * 8);
                at[k] = (int)((16 + r) * g.sa + c * 8);
            } else {
                fdivmod((uint32_t)(i - g.c_w1), g.fd_kc2, r, c);
                regs[k] = *reinterpret_cast<const bf16x8*>((const bf16*)p.W2 + (long)r * p.ldw2 + c * 8);
                at[k] = (int)((16 + g.n1r) * g.sa + r * g.sw2 + c * 8);
            }
        }
    }
    // the epilogues' bias and mask values (first round of each product; later rounds read them where they are used)
    const int nf1 = (p.N1 + 15) >> 4, nf2 = (p.N2 + 15) >> 4;
    float b1v[4] = {0.f, 0.f, 0.f, 0.f}, b2v[3][4];
    bool keep1[4] = {true, true, true, true};
    {
        const int nb = wave * 16 + fq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (p.bias1 && nb + r < p.N1) b1v[r] = p.bias1[nb + r];
            if (p.relu_mask1 && row_ok && nb + r < p.N1) keep1[r] = (float)((const bf16*)p.relu_mask1)[m * p.ldmask1 + nb + r] > 0.f;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = (wave + 8 * q) * 16 + fq * 4 + r;
                b2v[q][r] = (p.bias2 && n < p.N2) ? p.bias2[n] : 0.f;
            }
    }
    // padding (k beyond K1 and N1, rows beyond M, N1 and N2) is zero: everything is cleared while the loads are on their way
    {
        const int words = ((16 + g.n1r) * g.sa + (g.n2r + 16) * g.sw2) / 8;
        for (int i = tid; i < words; i += S2_THREADS) reinterpret_cast<bf16x8*>(s2_smem)[i] = zero8();
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < S2_MAXC; ++k)
        if (at[k] >= 0) *reinterpret_cast<bf16x8*>(As + at[k]) = regs[k];
    __syncthreads();
    // ---- product 1: a column fragment of H per wave
    const int nk1 = (p.K1 + 31) >> 5;
    for (int f = wave; f < nf1; f += 8) {
        const bf16* arow = As + fr * g.sa + fq * 8;
        const bf16* wrow = W1s + (f * 16 + fr) * g.sa + fq * 8;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int ks = 0;
        for (; ks + 1 < nk1; ks += 2) {
            const bf16x8 fa0 = *reinterpret_cast<const bf16x8*>(arow + ks * 32), fb0 = *reinterpret_cast<const bf16x8*>(wrow + ks * 32);
            const bf16x8 fa1 = *reinterpret_cast<const bf16x8*>(arow + ks * 32 + 32), fb1 = *reinterpret_cast<const bf16x8*>(wrow + ks * 32 + 32);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0, fa0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1, fa1, acc1, 0, 0, 0);
        }
        if (ks < nk1)
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(wrow + ks * 32), *reinterpret_cast<const bf16x8*>(arow + ks * 32), acc0, 0, 0, 0);
        const int nb = f * 16 + fq * 4;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc0[r] + acc1[r];
            if (f < 8) v += b1v[r];
            else if (p.bias1 && nb + r < p.N1) v += p.bias1[nb + r];
            v = small_act(v, p.act1);
            if (f < 8) { if (!keep1[r]) v = 0.f; }
            else if (p.relu_mask1 && row_ok && nb + r < p.N1 && !((float)((const bf16*)p.relu_mask1)[m * p.ldmask1 + nb + r] > 0.f)) v = 0.f;
            o[r] = (bf16)((nb + r < p.N1) ? v : 0.f);
        }
        *reinterpret_cast<bf16x4*>(&Hs[fr * g.sw2 + nb]) = o;
        if (row_ok && p.H) {
            bf16* h = (bf16*)p.H + m * p.ldh + nb;
            if (nb + 3 < p.N1 && (p.ldh & 3) == 0) *reinterpret_cast<bf16x4*>(h) = o;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (nb + r < p.N1) h[r] = o[r];
            }
        }
    }
    __syncthreads();
    // ---- product 2
    const int nk2 = ((p.N1 + 31) & ~31) >> 5;                          // at most 8 k-steps
    bf16x8 ha[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ha[u] = u < nk2 ? *reinterpret_cast<const bf16x8*>(&Hs[fr * g.sw2 + u * 32 + fq * 8]) : zero8();
    const float out_scale = p.out_scale != 0.f ? p.out_scale : 1.f;
    int round = 0;
    for (int f = wave; f < nf2; f += 8, ++round) {
        const bf16* wrow = W2s + (f * 16 + fr) * g.sw2 + fq * 8;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < nk2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(wrow + u * 32), ha[u], acc, 0, 0, 0);
        const int nb = f * 16 + fq * 4;
        if (!row_ok || nb >= p.N2) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = acc[r];
            if (round == 0) v[r] += b2v[0][r];
            else if (round == 1) v[r] += b2v[1][r];
            else if (round == 2) v[r] += b2v[2][r];
            else if (p.bias2 && nb + r < p.N2) v[r] += p.bias2[nb + r];
            v[r] = small_act(v[r], p.act2) * out_scale;
        }
        if (p.c_f32) {
            float* c = (float*)p.C + m * p.ldc + nb;
            if (nb + 3 < p.N2 && (p.ldc & 3) == 0) *reinterpret_cast<f32x4*>(c) = (f32x4){v[0], v[1], v[2], v[3]};
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (nb + r < p.N2) c[r] = v[r];
            }
        } else {
            bf16* c = (bf16*)p.C + m * p.ldc + nb;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
            if (nb + 3 < p.N2 && (p.ldc & 3) == 0) *reinterpret_cast<bf16x4*>(c) = o;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (nb + r < p.N2) c[r] = o[r];
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
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(WAVES == 8 ? 4 : 1))) void wgrad_kernel(const stes_wgrad_args p, int m_chunk, int t1, int t2, int m_blocks, RowDivs dv, StesDet det) {
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
    float* qaff = bias_acc + T;                                                  // [2][T]: the Q prologue's scale, shift for this tile's columns

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
    if (mblk >= m_blocks || m_begin >= m_end) { det_enter(det); det_leave(det); return; }

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
    // Small tile: the prologue's scale / shift live in LDS (read when a staged chunk is transformed), so that the registers hold
    // the squeeze-excite gate values of the chunks in flight instead -- those come from memory and are fetched with the operands
    // (64 x 64 at 3.2 M rows: 155 us instead of 175).  The capped large tile has no registers for them (the same change spills
    // and measures 4 % slower at 768 x 768): it keeps scale / shift in registers and reads the gate when it transforms.
    constexpr bool GATE_AHEAD = T == 64;
    float qsc[GATE_AHEAD ? 1 : 8], qsh[GATE_AHEAD ? 1 : 8];
    if constexpr (GATE_AHEAD) {
        if (has_affine && tid < T) {
            const bool on = n2_0 + tid < p.N2;
            qaff[tid] = on ? p.q_scale[n2_0 + tid] : 1.f; qaff[T + tid] = on ? p.q_shift[n2_0 + tid] : 0.f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { qsc[i] = (has_affine && c2_ok) ? p.q_scale[c2 + i] : 1.f; qsh[i] = (has_affine && c2_ok) ? p.q_shift[c2 + i] : 0.f; }
    }
    bf16x8 rp[LOADS], rq[LOADS], rr[T == 64 ? LOADS : 1];
    f32x4 rg[GATE_AHEAD ? LOADS : 1][2];   // gate values of each staged q chunk's 8 columns
    long qm[LOADS];                        // logical row of each staged q chunk
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
                if (GATE_AHEAD && has_gate && c2_ok) {
                    const float* g = p.q_gate + (long)fdiv((uint32_t)m, dv.gate) * p.N2 + c2;
                    rg[h][0] = *reinterpret_cast<const f32x4*>(g); rg[h][1] = *reinterpret_cast<const f32x4*>(g + 4);
                }
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
                if constexpr (GATE_AHEAD) {
                    if (has_affine) {
                        const f32x4 s0 = *reinterpret_cast<const f32x4*>(qaff + lc * 8), s1 = *reinterpret_cast<const f32x4*>(qaff + lc * 8 + 4);
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(qaff + T + lc * 8), b1 = *reinterpret_cast<const f32x4*>(qaff + T + lc * 8 + 4);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            f[i] = f[i] * (i < 4 ? s0[i & 3] : s1[i & 3]) + (i < 4 ? b0[i & 3] : b1[i & 3]);
                            if (p.q_relu) f[i] = fmaxf(f[i], 0.f);
                        }
                    }
                    if (has_gate) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) f[i] *= rg[h][i >> 2][i & 3];
                    }
                } else {
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
    if (GATE_AHEAD && has_affine) __syncthreads();       // the scale / shift columns are in LDS
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
    if (do_bias) {
        STES_DET_LDS(det, _Pragma("unroll") for (int i = 0; i < 8; ++i) atomicAdd(&bias_acc[lc * 8 + i], bsum[i]));
        __syncthreads();
    }
    det_enter(det);
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
    if (do_bias && tid < T && n1_0 + tid < p.N1) atomicAdd(p.dbias + n1_0 + tid, bias_acc[tid]);
    det_leave(det);
}

// ------------------------------------------------------------------ weight gradient, 320 x 320 (stage 3): strip kernel
// dW[320][320] = P^T pro(Q) is bound by reading P and Q once; the 128 x 128 tiles above read each operand three times (from
// L2, when the tiles of a row chunk stay in step) behind a single 64-row prefetch.  Here a workgroup of eight waves owns
// ALL 320 columns of P against a 160-column half of Q for its row chunk (accumulators: 5 x 5 fragments per wave), so P
// is read twice and Q once, and the operands travel global -> LDS directly (no staging registers) through a ring of four
// 32-row steps: three steps' loads (90 KiB per CU) are always in flight, one workgroup barrier per step.
// LDS image: 1 KiB pieces of 16 rows x 32 channels, one per load instruction (lane = row, 16-byte chunk; rows 8-15 hold
// their two 32-byte halves swapped, a swizzle applied to the per-lane SOURCE address, which makes the transposed
// fragment reads conflict-free).  The Q prologue (BatchNorm + ReLU + squeeze-excite gate) runs in place in LDS, each
// wave on the pieces it loaded itself, before the step's barrier.  M must be a multiple of 32.
constexpr int WS_N1 = 320, WS_NQ = 160, WS_STAGES = 4, WS_PIECES = 30, WS_GMAX = 12;
constexpr int WS_STAGE_BYTES = WS_PIECES * 1024;

template <bool PRO, bool SLAB>
__global__ __launch_bounds__(512) void wgrad_strip_kernel(const stes_wgrad_args p, int m_chunk, RowDivs dv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_smem[];
    float* sc_s = reinterpret_cast<float*>(ws_smem + WS_STAGES * WS_STAGE_BYTES);      // [160]
    float* sh_s = sc_s + WS_NQ;                                                         // [160]
    float* gate_s = sh_s + WS_NQ;                                                       // [WS_GMAX][160]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    // the two halves of a row chunk share blockIdx % 8, one XCD's L2: dY is then fetched from memory once, not once per half
    const int qhalf = (blockIdx.x >> 3) & 1, mblk = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7);
    const long m_begin = (long)mblk * m_chunk;
    const long m_end = (m_begin + m_chunk < p.M) ? m_begin + m_chunk : p.M;
    if (m_begin >= m_end) return;
    const int nsteps = (int)((m_end - m_begin) >> 5);
    const bf16* __restrict__ P = (const bf16*)p.P;
    const bf16* __restrict__ Q = (const bf16*)p.Q;
    const bf16* __restrict__ Q0 = (const bf16*)p.Q0;
    const int img0 = PRO && p.q_gate ? (int)fdiv((uint32_t)m_begin, dv.gate) : 0;
    if (PRO) {
        for (int i = tid; i < WS_NQ; i += 512) { sc_s[i] = p.q_scale ? p.q_scale[qhalf * WS_NQ + i] : 1.f; sh_s[i] = p.q_scale ? p.q_shift[qhalf * WS_NQ + i] : 0.f; }
        const int nimg = p.q_gate ? (int)fdiv((uint32_t)(m_end - 1), dv.gate) - img0 + 1 : 0;
        for (int i = tid; i < nimg * WS_NQ; i += 512) {
            const int g = i / WS_NQ, c = i - g * WS_NQ;
            gate_s[i] = p.q_gate[(long)(img0 + g) * p.N2 + qhalf * WS_NQ + c];
        }
    }
    // ---- this wave's pieces of a step: q = wave, wave + 8, ... < 30; q < 20: P piece (row half q / 10, 32-column piece q % 10)
    const int n_mine = wave < 6 ? 4 : 3;
    const int lrow = lane >> 2, lchunk = (lane & 3) ^ (2 * ((lrow >> 3) & 1));            // row in the piece, SOURCE chunk of this lane
    auto issue = [&](int step) {
        const long mrow = m_begin + (long)step * 32;
        unsigned char* stage = ws_smem + (step % WS_STAGES) * WS_STAGE_BYTES;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = wave + 8 * u;
            if (q >= WS_PIECES) break;
            const bf16* src;
            if (q < 20) {
                const int rp = q / 10, cp = q - rp * 10;
                src = P + (mrow + rp * 16 + lrow) * p.ldp + cp * 32 + lchunk * 8;
            } else {
                const int q2 = q - 20, rp = q2 / 5, cp = q2 - rp * 5;
                long m = mrow + rp * 16 + lrow;
                if (p.map_stride > 1) {
                    uint32_t n, rem, ho, wo;
                    fdivmod((uint32_t)m, dv.hw, n, rem);
                    fdivmod(rem, dv.wo, ho, wo);
                    m = ((long)n * p.Hi + (long)ho * p.map_stride) * p.Wi + (long)wo * p.map_stride;
                }
                const int c = qhalf * WS_NQ + cp * 32 + lchunk * 8;
                src = c < p.k0 ? Q0 + m * p.ldq0 + c : Q + m * p.ldq + c;
            }
            // (as inline asm: the compiler drains every outstanding load before the first LDS read that follows an LDS-DMA it
            // knows about, which would serialise the ring; the waits below are counted by hand)
            const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(stage + q * 1024));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
        }
    };
    // ---- Q prologue in place, on the pieces this wave loaded (their bytes have landed: the wave waited for its own loads)
    auto transform = [&](int step) {
        unsigned char* stage = ws_smem + (step % WS_STAGES) * WS_STAGE_BYTES;
        const long mrow = m_begin + (long)step * 32;
#pragma unroll
        for (int u = 2; u < 4; ++u) {
            const int q = wave + 8 * u;
            if (q < 20 || q >= WS_PIECES) continue;
            const int q2 = q - 20, rp = q2 / 5, cp = q2 - rp * 5;
            const int c = cp * 32 + lchunk * 8;                                           // channel within the 160-column half
            bf16x8* cell = reinterpret_cast<bf16x8*>(stage + q * 1024 + lane * 16);
            float f[8], sc[8], sh[8];
            unpack8(*cell, f);
            load8f(sc_s + c, sc); load8f(sh_s + c, sh);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f[e] = f[e] * sc[e] + sh[e];
                if (p.q_relu) f[e] = fmaxf(f[e], 0.f);
            }
            if (p.q_gate) {
                float g[8];
                load8f(gate_s + ((int)fdiv((uint32_t)(mrow + rp * 16 + lrow), dv.gate) - img0) * WS_NQ + c, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] *= g[e];
            }
            *cell = pack8(f);
        }
    };

    f32x4 acc[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int w1 = wave & 3, w2 = wave >> 2;               // 80 columns of P x 80 columns of Q per wave
    // transposed fragment reads: lane group fq supplies rows 8 (fq & 1) + (fr >> 2) (+ 4) of row half fq >> 1
    const int roff = ((fq >> 1) * 10) * 1024 + (8 * (fq & 1) + (fr >> 2)) * 64 + (fr & 3) * 8;   // P pieces; Q: row half stride 5 pieces
    const int qoff = 20 * 1024 + ((fq >> 1) * 5) * 1024 + (8 * (fq & 1) + (fr >> 2)) * 64 + (fr & 3) * 8;
    const int sw = 2 * (fq & 1);                           // rows 8-15 of a piece hold their 32-byte halves swapped

    __syncthreads();                                       // prologue tables
#pragma unroll 1
    for (int s0 = 0; s0 < WS_STAGES - 1 && s0 < nsteps; ++s0) issue(s0);
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
        // this wave's loads of `step` have landed when at most the later steps' loads are outstanding
        const int later = nsteps - 1 - step < WS_STAGES - 2 ? nsteps - 1 - step : WS_STAGES - 2;
        if (n_mine == 4) {
            if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (later >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (PRO) { transform(step); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();                      // every wave's pieces of `step` are in LDS; everyone is done with step - 1
        if (step + WS_STAGES - 1 < nsteps) issue(step + WS_STAGES - 1);
        const unsigned char* stage = ws_smem + (step % WS_STAGES) * WS_STAGE_BYTES;
        bf16x8 fa[5], fb[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int col = w1 * 80 + i * 16;
            const unsigned char* a = stage + roff + (col >> 5) * 1024 + ((2 * ((col >> 4) & 1)) ^ sw) * 16;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(a + 256));
            fa[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int col = w2 * 80 + j * 16;
            const unsigned char* b = stage + qoff + (col >> 5) * 1024 + ((2 * ((col >> 4) & 1)) ^ sw) * 16;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(b));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(b + 256));
            fb[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    // D[row = n1 (4 fq + r)][col = n2 (fr)]
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int n2 = qhalf * WS_NQ + w2 * 80 + j * 16 + fr;
            const int n2d = n2 < p.perm_fold ? gsm_dest(n2, p.perm_fold) : n2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n1 = w1 * 80 + i * 16 + fq * 4 + r;
                // (with a workspace: this chunk's own slab, plain stores, summed by wgrad_strip_fold_kernel -- 128 chunks of fp32
                // atomics into the same 400 KiB cost 42 us of the kernel's 102)
                if (SLAB) p.ws[((long)mblk * WS_N1 + n1) * (2 * WS_NQ) + n2d] = acc[i][j][r];
                else atomicAdd(p.dW + (long)n1 * p.lddw + n2d, acc[i][j][r]);
            }
        }
}

// dW += sum of the row chunks' slabs [mb][320][320]; grid (320*320/256, slab groups): a thread adds up its group's slabs for one
// element (coalesced across the workgroup) and then adds once to dW
__global__ __launch_bounds__(256) void wgrad_strip_fold_kernel(const float* __restrict__ ws, int mb, float* __restrict__ dW, long lddw,
                                                               StesDet det) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int per = (mb + gridDim.y - 1) / gridDim.y, s0 = blockIdx.y * per, s1 = s0 + per < mb ? s0 + per : mb;
    constexpr int NEL = WS_N1 * 2 * WS_NQ;
    float a = 0.f;
    int sl = s0;
    for (; sl + 4 <= s1; sl += 4) {
        const float v0 = ws[(long)sl * NEL + i], v1 = ws[(long)(sl + 1) * NEL + i], v2 = ws[(long)(sl + 2) * NEL + i], v3 = ws[(long)(sl + 3) * NEL + i];
        a += (v0 + v1) + (v2 + v3);
    }
    for (; sl < s1; ++sl) a += ws[(long)sl * NEL + i];
    det_enter(det);
    if (s1 > s0) atomicAdd(dW + (long)(i / (2 * WS_NQ)) * lddw + i % (2 * WS_NQ), a);
    det_leave(det);
}

template <int T, int WAVES>
static size_t wgrad_lds_bytes() {
    static const bool raised = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   160 * 1024) == hipSuccess;
    (void)raised;
    return (size_t)4 * (T == 128 ? 64 : 32) * (T + 16) * sizeof(bf16) + 3 * T * sizeof(float);
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
    // tall-and-skinny bf16 layers with small K and N: weights resident in LDS, rows streamed (stes_gemm_stream.hip)
    STES_CHECK(!a->wg_dw || (bnb && a->wg_q && !a->a_out && a->wg_ldq >= a->N && a->wg_lddw >= a->N && a->wg_k0 % 8 == 0 && (a->wg_k0 == 0 || a->wg_q0)),
               STES_ERR_ARG, "stes_gemm: the fused weight gradient needs the BatchNorm-backward prologue, Q and no a_out");
    STES_CHECK(!a->wg_dw || a->wg_map_stride <= 1 || (a->Ho > 0 && a->Wo > 0 && a->Hi > 0 && a->Wi > 0 && a->M % ((long)a->Ho * a->Wo) == 0),
               STES_ERR_ARG, "stes_gemm: row map dims of the fused weight gradient's Q");
    STES_CHECK(!a->e_scale || (a->e_shift && (!a->e_res || (a->e_ldres % 8 == 0 && a->e_ldres >= a->N)) && (a->e_res || !a->e_rscale) && !a->col_sum && !a->c_f32 && a->n_split == 0 &&
                               (a->e_rscale == nullptr) == (a->e_rshift == nullptr) && a->e_slice_n % 8 == 0 && a->e_slice_n <= a->N &&
                               (a->e_slice_n == 0 || a->e_slice)),
               STES_ERR_ARG, "stes_gemm: the block-tail epilogue needs scale and shift, bf16 output, no statistics and no split output");
    if (stes_gemm_stream(a, stream)) {
        STES_LAUNCH_CHECK();
        return STES_OK;
    }
    STES_CHECK(!a->e_scale, STES_ERR_UNSUPPORTED, "stes_gemm: the block-tail epilogue exists on the streaming kernel only; run stes_bn_act_residual");
    STES_CHECK(!a->wg_dw, STES_ERR_UNSUPPORTED, "stes_gemm: the fused weight gradient is built for K = 64, N = 32 on the streaming kernel");
    RowDivs dv;
    dv.hw = make_fastdiv(a->map_stride > 1 ? a->Ho * a->Wo : 1);
    dv.wo = make_fastdiv(a->map_stride > 1 ? a->Wo : 1);
    dv.gate = make_fastdiv(a->a_gate ? a->gate_group : 1);
    STES_CHECK(!bnb, STES_ERR_UNSUPPORTED, "stes_gemm: the BatchNorm-backward prologue exists for the streaming shapes only "
                                           "(N in {32,64,128,160,320} or a multiple of 128 up to 1024, K in {64,128,320}, M >= 8192, bf16 output, no epilogue option); run stes_bn_bwd_apply first");
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
    // Workgroups walk row tiles `m_slots` apart.  With more tiles than fit the GPU at once the grid is one resident set, not
    // more: further workgroups would run as a partial round beside an emptying GPU.
    static const int res128 = [] { int n = 0; return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_kernel<128>, 256, 0) == hipSuccess && n > 0 ? n : 1; }();
    static const int res64 = [] { int n = 0; return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_kernel<64>, 256, 0) == hipSuccess && n > 0 ? n : 1; }();
    const int resident = 256 * (wide ? res128 : res64);
    int m_slots = m_tiles;
    int slot_groups = stes_div_up(m_slots, 8);
    if ((long)slot_groups * 8 * n_tiles > resident) {
        slot_groups = resident / (8 * n_tiles) > 1 ? resident / (8 * n_tiles) : 1;
        m_slots = slot_groups * 8;
    }
    const int grid = slot_groups * n_tiles * 8;
    const StesDet det = a->col_sum ? stes_det_take(grid) : StesDet{nullptr, 0u, 0u};
    if (wide) hipLaunchKernelGGL(gemm_kernel<128>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a, m_tiles, n_tiles, m_slots, dv, det);
    else hipLaunchKernelGGL(gemm_kernel<64>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a, m_tiles, n_tiles, m_slots, dv, det);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gemm_small2(const stes_gemm2_args* a, void* stream) {
    STES_CHECK(a && a->A && a->W1 && a->W2 && a->C, STES_ERR_ARG, "stes_gemm_small2: null pointer");
    STES_CHECK(a->M > 0 && a->K1 > 0 && a->N1 > 0 && a->N2 > 0, STES_ERR_ARG, "stes_gemm_small2: empty shape");
    STES_CHECK(a->K1 % 8 == 0 && a->N1 % 8 == 0 && a->lda % 8 == 0 && a->ldw1 % 8 == 0 && a->ldw2 % 8 == 0, STES_ERR_ARG,
               "stes_gemm_small2: K1, N1 and the leading dimensions must be multiples of 8");
    STES_CHECK(a->N1 <= 256 && a->K1 <= 1024 && a->N2 <= 1024, STES_ERR_UNSUPPORTED, "stes_gemm_small2: N1 <= 256, K1 <= 1024, N2 <= 1024");
    STES_CHECK(a->act1 >= 0 && a->act1 <= 2 && a->act2 >= 0 && a->act2 <= 2, STES_ERR_ARG, "stes_gemm_small2: act is 0, 1 or 2");
    STES_CHECK(a->M <= (1L << 31), STES_ERR_UNSUPPORTED, "stes_gemm_small2: too many rows");
    // everything in LDS when it fits (the squeeze-excite pairs of stages 1 to 3 do, stage 4's 768-wide ones do not)
    Small2Lds g;
    const int k1p = (a->K1 + 31) & ~31, n1p = (a->N1 + 31) & ~31;
    g.sa = k1p + 8; g.sw2 = n1p + 8;
    g.n1r = (a->N1 + 15) & ~15; g.n2r = (a->N2 + 15) & ~15;
    const int kc1 = a->K1 / 8, kc2 = a->N1 / 8;
    g.c_a = 16 * kc1; g.c_w1 = g.c_a + a->N1 * kc1; g.c_all = g.c_w1 + a->N2 * kc2;
    g.fd_kc1 = make_fastdiv(kc1); g.fd_kc2 = make_fastdiv(kc2);
    const size_t lds = ((size_t)(16 + g.n1r) * g.sa + (size_t)(g.n2r + 16) * g.sw2) * 2;
    static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_small2_lds_kernel),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    static const bool lds_form = !(getenv("STES_SMALL2_LDS") && atoi(getenv("STES_SMALL2_LDS")) == 0);
    if (lds_form && big_lds && lds <= 150 * 1024 && g.c_all <= S2_MAXC * S2_THREADS)
        hipLaunchKernelGGL(gemm_small2_lds_kernel, dim3((unsigned)((a->M + 15) / 16)), dim3(S2_THREADS), lds, (hipStream_t)stream, *a, g);
    else
        hipLaunchKernelGGL(gemm_small2_kernel, dim3((unsigned)((a->M + 15) / 16)), dim3(256), 0, (hipStream_t)stream, *a);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int64_t stes_wgrad_ws_floats(void) { return 128L * WS_N1 * 2 * WS_NQ; }

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
    STES_CHECK(a->M <= STES_MAX_ROWS, STES_ERR_UNSUPPORTED, "stes_wgrad: more than 2^31 rows");
    // stage 3's 320 x 320 products: the strip kernel (P read twice, Q once, LDS-DMA ring)
    if (a->N1 == WS_N1 && a->N2 == 2 * WS_NQ && a->M % 32 == 0 && a->M >= 32 * 256 && !a->dbias && !a->p_raw && a->ldp % 8 == 0 &&
        (a->k0 == 0 || a->k0 % 8 == 0)) {
        const long chunk = (a->M / 32 + 127) / 128 * 32;                                // 128 row chunks x 2 halves = 256 workgroups
        const int mb = stes_div_up(a->M, chunk);
        const bool pro = a->q_scale || a->q_gate;
        const long imgs = a->q_gate ? chunk / a->gate_group + 2 : 0;
        if (imgs <= WS_GMAX) {
            RowDivs dv;
            dv.hw = make_fastdiv(a->map_stride > 1 ? a->Ho * a->Wo : 1);
            dv.wo = make_fastdiv(a->map_stride > 1 ? a->Wo : 1);
            dv.gate = make_fastdiv(a->q_gate ? a->gate_group : 1);
            const size_t lds = (size_t)WS_STAGES * WS_STAGE_BYTES + (size_t)(2 + WS_GMAX) * WS_NQ * sizeof(float);
            const bool slab = a->ws && a->ws_floats >= (int64_t)mb * WS_N1 * 2 * WS_NQ;
            // (deterministic mode: without the slab workspace the strip kernel's results meet in atomics whose order nothing
            // fixes; the tiled kernel below commits in turn)
            const bool det_on = stes_det_take(0).ticket != nullptr;
            static const bool raised = [] {
                bool ok = true;
                const void* fns[4] = {reinterpret_cast<const void*>(&wgrad_strip_kernel<true, true>), reinterpret_cast<const void*>(&wgrad_strip_kernel<true, false>),
                                      reinterpret_cast<const void*>(&wgrad_strip_kernel<false, true>), reinterpret_cast<const void*>(&wgrad_strip_kernel<false, false>)};
                for (const void* f : fns) ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
                return ok;
            }();
            if (raised && (slab || !det_on)) {
                const dim3 grid(stes_div_up(mb, 8) * 16), block(512);       // (chunk, half) -> block: see the kernel
                hipStream_t st = (hipStream_t)stream;
                if (pro && slab) hipLaunchKernelGGL((wgrad_strip_kernel<true, true>), grid, block, lds, st, *a, (int)chunk, dv);
                else if (pro) hipLaunchKernelGGL((wgrad_strip_kernel<true, false>), grid, block, lds, st, *a, (int)chunk, dv);
                else if (slab) hipLaunchKernelGGL((wgrad_strip_kernel<false, true>), grid, block, lds, st, *a, (int)chunk, dv);
                else hipLaunchKernelGGL((wgrad_strip_kernel<false, false>), grid, block, lds, st, *a, (int)chunk, dv);
                STES_LAUNCH_CHECK();
                if (slab) {
                    hipLaunchKernelGGL(wgrad_strip_fold_kernel, dim3(WS_N1 * 2 * WS_NQ / 256, 8), dim3(256), 0, st, a->ws, mb, a->dW, (long)a->lddw,
                                       stes_det_take(WS_N1 * 2 * WS_NQ / 256 * 8));
                    STES_LAUNCH_CHECK();
                }
                return STES_OK;
            }
        }
    }
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
        hipLaunchKernelGGL((wgrad_kernel<128, 8>), dim3(grid), dim3(512), lds, (hipStream_t)stream, *a, (int)chunk, t1, t2, mb_arg, dv,
                           stes_det_take(grid));
    } else {
        const size_t lds = wgrad_lds_bytes<64, 4>();
        hipLaunchKernelGGL((wgrad_kernel<64, 4>), dim3(grid), dim3(256), lds, (hipStream_t)stream, *a, (int)chunk, t1, t2, mb_arg, dv,
                           stes_det_take(grid));
    }
    STES_LAUNCH_CHECK();
    return STES_OK;
}
