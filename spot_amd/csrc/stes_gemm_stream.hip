// Streaming bf16 MFMA GEMM for the tall-and-skinny 1x1 convolutions (K in {32, 64, 128, 320}, column slice <= 160).
//
// These layers are bound by HBM, not by MFMA.  The weight slice [NB][K] sits in LDS for the workgroup's lifetime and every
// wave streams its own row tiles with no workgroup barrier.  A fragments go global -> registers in MFMA operand layout
// (lane (fr,fq): row fr, k = 8fq..8fq+7: one 16-byte load) through a ring that holds TA whole tiles: slot (t, ks) is
// refilled for the tile TA tiles further on as soon as k-step ks of tile t has consumed it, so a wave always has TA tiles
// of operand bytes in flight.  Everything about a tile that is the same in all lanes (tile index, row base, byte offsets)
// is kept in scalar registers, the k-step is a compile-time constant, and no load sits under a branch: then the compiler
// counts its outstanding loads (s_waitcnt vmcnt(N)) instead of draining the queue (vmcnt(0)) at every use.
// The operand prologue of k-step ks+1 (BatchNorm + ReLU + squeeze-excite gate, or BatchNorm backward) is written in front
// of the MFMAs of k-step ks, so its VALU work issues in the shadow of the matrix pipe.
// The output tile is staged through a wave-private LDS buffer so that it leaves as whole rows (up to 1 KiB per store
// instruction) and so that the BatchNorm statistics of the rounded output cost 16 registers per lane (a lane always reads
// back the same 8 columns).
#include <mutex>
#include <type_traits>
#include "stes_common.h"
#include "stes_gemm_stream.h"

namespace {

// f(integral_constant<int, I>) for I = B .. E-1, in order
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// PRO selects the operand prologue: 0 none; 1 BatchNorm (+ ReLU) (+ squeeze-excite gate): relu?(A*scale + shift) * gate;
// 2 BatchNorm backward: A is the gradient at a BatchNorm's output, a_raw the raw tensor under that BatchNorm, and the operand
// is bf16(alpha*A + beta*a_raw + delta) -- what stes_bn_bwd_apply would have written; it is written here too (a_out, by the
// workgroups of column slice 0) for the weight-gradient kernel, which saves that kernel's pass and one read of its result.
// NK = K/32 k-steps per tile, TA tiles in the ring, J 16-row fragments per tile.
// WG: the layer's weight gradient dW[K][N] += pro(A)^T Q in the same pass (PRO 2, J = 2, one column slice).  A tile's 32 rows are
// one MFMA k-step of the transposed product: the transformed operand and the Q tile go to a wave-private LDS area as
// [32 rows][16 columns] sub-tiles (rows permuted by rho32 like the weight-gradient kernels' tiles) and come back through
// ds_read_b64_tr_b16 as fragments with the rows in the contraction dimension; K/16 x N/16 accumulator tiles per wave, added to
// wg_dw with atomics when the wave has walked its tiles.  The transformed operand then never travels through memory.
__device__ __forceinline__ int rho32(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }
__device__ __forceinline__ int wg_dest(int c, int fold) {    // within-group channel interleave of the gated shift (model/shift.py:124-128)
    const int half = fold >> 1, quarter = fold >> 2;
    const int g = c / half, rem = c - g * half;
    const int i = rem / quarter, j = rem - i * quarter;
    return g * half + 2 * j + i;
}

// TAIL: the block-tail epilogue relu(e_scale*c + e_shift + residual) on the staged rows (evaluation forward of a bottleneck's
// last 1x1 conv); its own instantiations, because the residual rows it keeps in flight cost registers every other form needs.
template <int NI, int WAVES, int PRO, int NK, int TA, int J, bool WG, bool TAIL = false>   // NB = 16*NI columns per workgroup
__device__ __forceinline__ void gemm_stream_body(const stes_gemm_args& p, int row_tiles, int n_slices, int m_slots, const StreamDivs& dv,
                                                 const StesDet& det) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(!WG || (PRO == 2 && J == 2), "the fused weight gradient rides on the BatchNorm-backward form with 32-row tiles");
    constexpr bool AFF = PRO == 1, BNB = PRO == 2;
    constexpr int NB = NI * 16, TR = 16 * J, K = NK * 32;
    constexpr int CPR = NB / 8;                 // 16-byte chunks per output row
    constexpr int RPP = 64 / CPR;               // output rows per store pass of a wave
    constexpr int RIT = (16 + RPP - 1) / RPP;   // store passes per 16 rows
    constexpr int SRS = NB * 2 + 16;            // staging row stride (bytes)
    constexpr int RS = 2 * K + 32;              // weight row stride (bytes): (RS/16) % 4 == 2 keeps the b128 fragment reads conflict-free
    constexpr int GCH = (K / 2 + 63) / 64;      // AFF: 16-byte chunks per lane of the two gate rows [2][K] f32
    static_assert(!AFF || 8 * K <= 16 * SRS, "the gate rows live in the wave's staging buffer");
    const int N = p.N;
    const long M = p.M;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform -> scalar registers
    const int fr = lane & 15, fq = lane >> 4;
    unsigned char* wl = smem;
    unsigned char* stage = smem + NB * RS + wave * 16 * SRS;
    float* sc_s = reinterpret_cast<float*>(smem + NB * RS + WAVES * 16 * SRS);
    float* sh_s = sc_s + K;
    float* de_s = sh_s + K;                     // BNB: sc_s = alpha, sh_s = beta, de_s = delta
    float* red = de_s + K;                      // [2][NB]
    float* epi_l = red + 2 * NB;                // [4][NB]: block-tail epilogue scale, shift, residual scale, residual shift
    constexpr int WG_PSUB = K / 16, WG_QSUB = NB / 16;                         // [32][16] bf16 sub-tiles (1 KiB each) of pro(A) and of Q
    unsigned char* wgl = reinterpret_cast<unsigned char*>(epi_l + (TAIL ? 4 * NB : 0)) + wave * (WG_PSUB + WG_QSUB) * 1024;
    // AFF: this wave's copy of the gate rows of the (at most two) images its current tile touches, or ones without a gate.
    // It lives in the wave's output staging buffer: the gate is read in the k loop, the staging buffer in the tile epilogue.
    float* gl = reinterpret_cast<float*>(stage);

    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int slice = idx % n_slices;
    const int slot = (idx / n_slices) * 8 + xcd;
    const int n0 = slice * NB;
    const bf16* __restrict__ A = (const bf16*)p.A;
    const bf16* __restrict__ A0 = (const bf16*)p.A0;
    const bf16* __restrict__ R = (const bf16*)p.a_raw;
    const bool stats = p.col_sum != nullptr;
    const float relu_floor = p.a_relu ? 0.f : -__builtin_inff();

    {   // weight slice, prologue constants, statistics scratch
        const bf16* __restrict__ W = (const bf16*)p.W;
        constexpr int kch = K >> 3;
        for (int c = tid; c < NB * kch; c += WAVES * 64) {
            const int r = c / kch, kc = c - r * kch;
            const bf16x8 v = (n0 + r < N) ? *reinterpret_cast<const bf16x8*>(W + (long)(n0 + r) * p.ldw + kc * 8) : zero8();
            *reinterpret_cast<bf16x8*>(wl + r * RS + kc * 16) = v;
        }
        if (AFF) for (int k = tid; k < K; k += WAVES * 64) { sc_s[k] = p.a_scale[k]; sh_s[k] = p.a_shift[k]; }
        if (BNB) for (int k = tid; k < K; k += WAVES * 64) { sc_s[k] = p.bn_alpha[k]; sh_s[k] = p.bn_beta[k]; de_s[k] = p.bn_delta[k]; }
        for (int k = tid; k < 2 * NB; k += WAVES * 64) red[k] = 0.f;
        if constexpr (TAIL) for (int k = tid; k < NB; k += WAVES * 64) {
            epi_l[k] = p.e_scale[n0 + k]; epi_l[NB + k] = p.e_shift[n0 + k];
            epi_l[2 * NB + k] = p.e_rscale ? p.e_rscale[n0 + k] : 1.f; epi_l[3 * NB + k] = p.e_rscale ? p.e_rshift[n0 + k] : 0.f;
        }
    }
    __syncthreads();
    if (slot >= m_slots) { det_enter(det); det_leave(det); return; }     // (deterministic mode: the turn still has to pass)

    const long tstride = (long)m_slots * WAVES;
    long c_tile = (long)slot * WAVES + wave;                       // scalar

    // ---- operand row pointers of a tile.  Rows past the end (the last tile's tail, the ring running ahead of the last
    // tile) read row M-1 instead: their results are never stored, and a load that is always issued needs no branch.
    const bf16* fa[J]; const bf16* fa0[J]; const bf16* fraw[J];
    auto rows_of = [&](long tile) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            long m = tile * TR + j * 16 + fr;
            if (m >= M) m = M - 1;
            long src = m;
            if (p.map_stride > 1) {
                uint32_t n, rem, ho, wo;
                fdivmod((uint32_t)m, dv.hw, n, rem);
                fdivmod(rem, dv.wo, ho, wo);
                src = ((long)n * p.Hi + (long)ho * p.map_stride) * p.Wi + (long)wo * p.map_stride;
            }
            fa[j] = A + src * p.lda + fq * 8;
            fa0[j] = A0 ? A0 + src * p.lda0 + fq * 8 : fa[j];
            fraw[j] = BNB ? R + src * p.lda + fq * 8 : fa[j];
        }
    };
    bf16x8 ring[TA][NK][J];
    bf16x8 ring_r[BNB ? TA : 1][BNB ? NK : 1][J];
    // WG: Q tile of 32 rows x NB columns = 32 * NB/8 16-byte chunks, NB/16 per lane; chunk c of a lane is (row, 8 columns)
    constexpr int WG_QC = WG ? NB / 16 : 1;
    bf16x8 ring_q[WG ? TA : 1][WG_QC];
    const bf16* fq_src[WG_QC];
    f32x4 accw[WG ? WG_PSUB : 1][WG ? WG_QSUB : 1];
    if constexpr (WG) {
#pragma unroll
        for (int a = 0; a < WG_PSUB; ++a)
#pragma unroll
            for (int b = 0; b < WG_QSUB; ++b) accw[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto q_rows_of = [&](long tile) {
        if constexpr (WG) {
#pragma unroll
            for (int c = 0; c < WG_QC; ++c) {
                const int id = lane + 64 * c, row = id / (NB / 8), cc = id % (NB / 8);
                long m = tile * TR + row;
                if (m >= M) m = M - 1;
                if (p.wg_map_stride > 1) {
                    uint32_t n, rem, ho, wo;
                    fdivmod((uint32_t)m, dv.hw, n, rem);
                    fdivmod(rem, dv.wo, ho, wo);
                    m = ((long)n * p.Hi + (long)ho * p.wg_map_stride) * p.Wi + (long)wo * p.wg_map_stride;
                }
                fq_src[c] = (cc * 8 < p.wg_k0) ? (const bf16*)p.wg_q0 + m * p.wg_ldq0 + cc * 8 : (const bf16*)p.wg_q + m * p.wg_ldq + cc * 8;
            }
        }
    };
    auto q_refill = [&](auto tc) {
        if constexpr (WG) {
            constexpr int t = decltype(tc)::value;
#pragma unroll
            for (int c = 0; c < WG_QC; ++c) ring_q[t][c] = *reinterpret_cast<const bf16x8*>(fq_src[c]);
        }
    };
    // slot (t, ks) <- k-step ks of the tile whose pointers are current
    auto refill = [&](auto tc, auto kc) {
        constexpr int t = decltype(tc)::value, ks = decltype(kc)::value;
        const bool compact = ks * 32 + fq * 8 < p.k0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const bf16* src = compact ? fa0[j] : fa[j];
            ring[t][ks][j] = *reinterpret_cast<const bf16x8*>(src + ks * 32);
            if constexpr (BNB) ring_r[t][ks][j] = *reinterpret_cast<const bf16x8*>(fraw[j] + ks * 32);
        }
    };

    // ---- AFF: gate rows of a tile, global -> registers (issued a tile ahead), registers -> LDS after the epilogue
    f32x4 greg[AFF ? GCH : 1];
    const float* c_gate[J];
#pragma unroll
    for (int j = 0; j < J; ++j) c_gate[j] = gl;
    auto gate_fetch = [&](long tile) {
        if (!AFF) return;
        if (p.a_gate == nullptr) {
#pragma unroll
            for (int q = 0; q < GCH; ++q) greg[q] = (f32x4){1.f, 1.f, 1.f, 1.f};
            return;
        }
        long m0 = tile * TR, m1 = m0 + TR - 1;
        if (m0 >= M) m0 = M - 1;
        if (m1 >= M) m1 = M - 1;
        const uint32_t i0 = fdiv((uint32_t)m0, dv.gate), i1 = fdiv((uint32_t)m1, dv.gate);      // i1 <= i0 + 1 (gate_group >= 32)
#pragma unroll
        for (int q = 0; q < GCH; ++q) {
            int c = lane + 64 * q;
            if (c >= K / 2) c = K / 2 - 1;                          // (a duplicate of the last chunk: no branch around the load)
            const int r = c >= K / 4, cc = c - r * (K / 4);
            greg[q] = *reinterpret_cast<const f32x4*>(p.a_gate + (long)(r ? i1 : i0) * K + cc * 4);
        }
    };
    auto gate_commit = [&](long tile) {          // the registers hold the rows of `tile`; call once the staging buffer is free
        if (!AFF) return;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < GCH; ++q) {
            int c = lane + 64 * q;
            if (c >= K / 2) c = K / 2 - 1;
            *reinterpret_cast<f32x4*>(gl + c * 4) = greg[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (p.a_gate != nullptr) {
            long m0 = tile * TR;
            if (m0 >= M) m0 = M - 1;
            const uint32_t i0 = fdiv((uint32_t)m0, dv.gate);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                long m = tile * TR + j * 16 + fr;
                if (m >= M) m = M - 1;
                c_gate[j] = gl + (fdiv((uint32_t)m, dv.gate) - i0) * K;
            }
        }
    };

    // ---- operand prologue of slot (t, ks): ring registers -> MFMA fragments
    auto prologue = [&](auto tc, auto kc, bf16x8 (&fb)[J]) {
        constexpr int t = decltype(tc)::value, ks = decltype(kc)::value;
        const int k = ks * 32 + fq * 8;
#pragma unroll
        for (int j = 0; j < J; ++j) fb[j] = ring[t][ks][j];
        if constexpr (BNB) {
            float al[8], be[8], de[8];
            load8f(sc_s + k, al); load8f(sh_s + k, be); load8f(de_s + k, de);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                float f[8], r[8];
                unpack8(fb[j], f);
                unpack8(ring_r[t][ks][j], r);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = al[e] * f[e] + be[e] * r[e] + de[e];
                fb[j] = pack8(f);
                const long m = c_tile * TR + j * 16 + fr;
                if (p.a_out && slice == 0 && m < M) *reinterpret_cast<bf16x8*>(p.a_out + m * p.lda_out + k) = fb[j];
                if constexpr (WG) {               // sub-tile k/16, row rho32(j*16+fr), column half (k/8)&1; rows past the end add nothing
                    *reinterpret_cast<bf16x8*>(wgl + (k >> 4) * 1024 + rho32(j * 16 + fr) * 32 + ((k >> 3) & 1) * 16) = m < M ? fb[j] : zero8();
                }
            }
        }
        if constexpr (AFF) {
            float s[8], b[8];
            load8f(sc_s + k, s); load8f(sh_s + k, b);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                float f[8], g[8];
                unpack8(fb[j], f);
                load8f(c_gate[j] + k, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e] * s[e] + b[e], relu_floor) * g[e];
                fb[j] = pack8(f);
            }
        }
    };

    f32x4 acc[NI][J];
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    // store phase: this lane's 8 columns, its row phase, and the tensor its columns go to (the compact one below n_split)
    const int o_chunk = lane % CPR, o_rsub = lane / CPR;
    const bool o_active = o_rsub < RPP;
    const int col = n0 + o_chunk * 8;
    const bool to_c0 = col < p.n_split;
    bf16* const o_base = to_c0 ? (bf16*)p.C0 + col : (bf16*)p.C + col;
    const uint32_t o_ld = (uint32_t)(to_c0 ? p.ldc0 : p.ldc);
    // block-tail epilogue: relu(scale*c + shift + residual) on the staged rows; the residual rows of a 16-row group are asked
    // for before the group is staged
    constexpr bool tail = TAIL;
    const bf16* __restrict__ e_res = (const bf16*)p.e_res;

    // ---- fill the ring: TA tiles
    static_for<0, TA>([&](auto tc) {
        rows_of(c_tile + decltype(tc)::value * tstride);
        static_for<0, NK>([&](auto kc) { refill(tc, kc); });
        q_rows_of(c_tile + decltype(tc)::value * tstride);
        q_refill(tc);
    });
    gate_fetch(c_tile);
    gate_commit(c_tile);

    while (c_tile < row_tiles) {
        static_for<0, TA>([&](auto tc) {
            // tiles past the end (the last round of a ring of several tiles) run with every row masked
            rows_of(c_tile + TA * tstride);                         // what this tile's slots are refilled with
            if constexpr (WG) {                                     // this tile's Q rows -> LDS sub-tiles, the slot refilled
#pragma unroll
                for (int c = 0; c < WG_QC; ++c) {
                    const int id = lane + 64 * c, row = id / (NB / 8), cc = id % (NB / 8);
                    *reinterpret_cast<bf16x8*>(wgl + (WG_PSUB + (cc >> 1)) * 1024 + rho32(row) * 32 + (cc & 1) * 16) = ring_q[decltype(tc)::value][c];
                }
                q_rows_of(c_tile + TA * tstride);
                q_refill(tc);
            }
            gate_fetch(c_tile + tstride);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < J; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bf16x8 fbc[J], fbn[J];
            prologue(tc, std::integral_constant<int, 0>{}, fbc);
            refill(tc, std::integral_constant<int, 0>{});
            static_for<0, NK>([&](auto kc) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (ks + 1 < NK) {
                    prologue(tc, std::integral_constant<int, ks + 1>{}, fbn);
                    refill(tc, std::integral_constant<int, ks + 1>{});
                }
                // D[n][m] += W[n][k] A[m][k]
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wl + (i * 16 + fr) * RS + (ks * 32 + fq * 8) * 2);
#pragma unroll
                    for (int j = 0; j < J; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, fbc[j], acc[i][j], 0, 0, 0);
                }
                if constexpr (ks + 1 < NK) {
#pragma unroll
                    for (int j = 0; j < J; ++j) fbc[j] = fbn[j];
                }
            });
            if constexpr (WG) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // fragment of a [32][16] sub-tile with the 32 rows as the contraction dimension (as conv3x3g_wgrad_kernel)
                const int px_lo = 8 * fq + (fr >> 2), cc4 = 4 * (fr & 3);
                const int o_lo = rho32(px_lo) * 32 + cc4 * 2, o_hi = rho32(px_lo + 4) * 32 + cc4 * 2;
                auto frag_of = [&](int sub) {
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(wgl + sub * 1024 + o_lo));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(wgl + sub * 1024 + o_hi));
                    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                };
                bf16x8 qf[WG_QSUB];
#pragma unroll
                for (int b = 0; b < WG_QSUB; ++b) qf[b] = frag_of(WG_PSUB + b);
#pragma unroll
                for (int a = 0; a < WG_PSUB; ++a) {
                    const bf16x8 pf = frag_of(a);
#pragma unroll
                    for (int b = 0; b < WG_QSUB; ++b) accw[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, qf[b], accw[a][b], 0, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
            }
            // ---- tile done: through the staging buffer, 16 rows at a time
            const long m_base = c_tile * TR;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const long mrow0 = m_base + j * 16;                  // scalar
                long left = M - mrow0;
                const int rows_here = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
                const long toff = to_c0 ? mrow0 * p.ldc0 : mrow0 * p.ldc;
                bf16x8 rres[TAIL ? RIT : 1];
                if constexpr (TAIL) {
                    if (tail) {
#pragma unroll
                        for (int rr = 0; rr < RIT; ++rr) {
                            const int row = rr * RPP + o_rsub;
                            rres[rr] = (e_res && o_active && row < rows_here) ? *reinterpret_cast<const bf16x8*>(e_res + (mrow0 + row) * p.e_ldres + col) : zero8();
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[i][j][r];
                    *reinterpret_cast<bf16x4*>(stage + fr * SRS + (i * 16 + fq * 4) * 2) = o;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int rr = 0; rr < RIT; ++rr) {
                    const int row = rr * RPP + o_rsub;
                    if (o_active && row < rows_here) {
                        bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + row * SRS + o_chunk * 16);
                        if constexpr (TAIL) {
                            if (tail) {                 // the arithmetic of bn_act_residual_kernel, in its order
                                float f[8], r[8], a[8], b[8];
                                unpack8(v, f);
                                unpack8(rres[rr], r);
                                load8f(epi_l + o_chunk * 8, a); load8f(epi_l + NB + o_chunk * 8, b);
#pragma unroll
                                for (int e = 0; e < 8; ++e) f[e] = f[e] * a[e] + b[e];
                                if (p.e_rscale) {
                                    load8f(epi_l + 2 * NB + o_chunk * 8, a); load8f(epi_l + 3 * NB + o_chunk * 8, b);
#pragma unroll
                                    for (int e = 0; e < 8; ++e) f[e] += r[e] * a[e] + b[e];
                                } else if (e_res) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) f[e] += r[e];
                                }
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = (bf16)fmaxf(f[e], 0.f);
                                if (col < p.e_slice_n) *reinterpret_cast<bf16x8*>((bf16*)p.e_slice + (mrow0 + row) * p.e_slice_n + col) = v;
                            }
                        }
                        *reinterpret_cast<bf16x8*>(o_base + toff + __umul24((uint32_t)row, o_ld)) = v;
                        if (stats) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; ssum[e] += f; ssq[e] += f * f; }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            c_tile += tstride;
            gate_commit(c_tile);
        });
    }
    if (stats) {
        STES_DET_LDS(det, if (o_active) {
            _Pragma("unroll") for (int e = 0; e < 8; ++e) { atomicAdd(&red[o_chunk * 8 + e], ssum[e]); atomicAdd(&red[NB + o_chunk * 8 + e], ssq[e]); }
        });
        __syncthreads();
    }
    det_enter(det);
    if constexpr (WG) {                  // D[k][n]: lane column n = fr, rows k = 4 fq + r
        // (every wave holds a partial of the whole weight gradient)
        STES_DET_WAVES(det, _Pragma("unroll") for (int a = 0; a < WG_PSUB; ++a) {
            _Pragma("unroll") for (int b = 0; b < WG_QSUB; ++b) {
                const int n = b * 16 + fr;
                const int nd = n < p.wg_perm_fold ? wg_dest(n, p.wg_perm_fold) : n;
                _Pragma("unroll") for (int r = 0; r < 4; ++r) atomicAdd(p.wg_dw + (long)(a * 16 + fq * 4 + r) * p.wg_lddw + nd, accw[a][b][r]);
            }
        });
    }
    if (stats) {
        for (int t = tid; t < 2 * NB; t += WAVES * 64) {
            const int which = t / NB, nl = t - which * NB;
            atomicAdd((which ? p.col_sumsq : p.col_sum) + n0 + nl, red[t]);
        }
    }
    det_leave(det);
}

template <int NI, int WAVES, int PRO, int NK, int TA, int J>
__global__ __launch_bounds__(WAVES * 64) void gemm_stream_kernel(const stes_gemm_args p, int row_tiles, int n_slices, int m_slots,
                                                                 StreamDivs dv, StesDet det) {
    gemm_stream_body<NI, WAVES, PRO, NK, TA, J, false>(p, row_tiles, n_slices, m_slots, dv, det);
}

// the same with the layer's weight gradient in the same pass (see gemm_stream_body)
template <int NI, int WAVES, int PRO, int NK, int TA, int J>
__global__ __launch_bounds__(WAVES * 64, 2) void gemm_stream_wg_kernel(const stes_gemm_args p, int row_tiles, int n_slices, int m_slots,
                                                                    StreamDivs dv, StesDet det) {
    gemm_stream_body<NI, WAVES, PRO, NK, TA, J, true>(p, row_tiles, n_slices, m_slots, dv, det);
}

// the same with the block-tail epilogue (see gemm_stream_body)
template <int NI, int WAVES, int PRO, int NK, int TA, int J>
__global__ __launch_bounds__(WAVES * 64) void gemm_stream_tail_kernel(const stes_gemm_args p, int row_tiles, int n_slices, int m_slots,
                                                                      StreamDivs dv, StesDet det) {
    gemm_stream_body<NI, WAVES, PRO, NK, TA, J, false, true>(p, row_tiles, n_slices, m_slots, dv, det);
}

// Workgroups of one instantiation resident per CU at a given dynamic LDS size (queried once; the LDS size is fixed by
// the template arguments).  Entry points may be called from any host thread: the cache sits behind a mutex.
template <int NI, int WAVES, int PRO, int NK, int TA, int J, bool WG, bool TAIL>
static int stream_residency(size_t lds) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 1;
        hipError_t e;
        if constexpr (WG) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_wg_kernel<NI, WAVES, PRO, NK, TA, J>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_stream_wg_kernel<NI, WAVES, PRO, NK, TA, J>, WAVES * 64, lds);
        } else if constexpr (TAIL) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_tail_kernel<NI, WAVES, PRO, NK, TA, J>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_stream_tail_kernel<NI, WAVES, PRO, NK, TA, J>, WAVES * 64, lds);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_kernel<NI, WAVES, PRO, NK, TA, J>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_stream_kernel<NI, WAVES, PRO, NK, TA, J>, WAVES * 64, lds);
        }
        if (e != hipSuccess || n < 1) n = 1;
        per_cu = n;
    }
    return per_cu;
}

template <int NI, int WAVES, int PRO, int NK, int TA, int J, bool WG = false, bool TAIL = false>
static void launch(const stes_gemm_args* a, hipStream_t stream, const StreamDivs& dv) {
    constexpr int NB = NI * 16, K = NK * 32;
    const size_t lds = (size_t)NB * (2 * K + 32) + (size_t)WAVES * 16 * (NB * 2 + 16) + (size_t)(3 * K + (TAIL ? 6 : 2) * NB) * sizeof(float) +
                       (WG ? (size_t)WAVES * (K / 16 + NB / 16) * 1024 : 0);
    const int n_slices = a->N / NB;
    const int row_tiles = stes_div_up(a->M, 16 * J);
    const int per_cu = stream_residency<NI, WAVES, PRO, NK, TA, J, WG, TAIL>(lds);
    // one resident set of workgroups: further ones would only add statistics atomics and weight-slice loads
    int m_slots = stes_div_up(row_tiles, WAVES);
    const int cap = 256 * per_cu / n_slices > 8 ? 256 * per_cu / n_slices : 8;
    if (m_slots > cap) m_slots = cap;
    const int grid = stes_div_up(m_slots, 8) * n_slices * 8;
    const StesDet det = (WG || a->col_sum) ? stes_det_take(grid) : StesDet{nullptr, 0u, 0u};
    if constexpr (WG)
        hipLaunchKernelGGL((gemm_stream_wg_kernel<NI, WAVES, PRO, NK, TA, J>), dim3(grid), dim3(WAVES * 64), lds, stream, *a, row_tiles, n_slices,
                           m_slots, dv, det);
    else if constexpr (TAIL)
        hipLaunchKernelGGL((gemm_stream_tail_kernel<NI, WAVES, PRO, NK, TA, J>), dim3(grid), dim3(WAVES * 64), lds, stream, *a, row_tiles, n_slices,
                           m_slots, dv, det);
    else
        hipLaunchKernelGGL((gemm_stream_kernel<NI, WAVES, PRO, NK, TA, J>), dim3(grid), dim3(WAVES * 64), lds, stream, *a, row_tiles, n_slices,
                           m_slots, dv, det);
}

template <int NI, int WAVES, int PRO>
static bool launch_k(const stes_gemm_args* a, hipStream_t stream, const StreamDivs& dv) {
    // ring: K = 320 one 16-row tile (10 k-steps) ahead; K = 128 one 32-row tile; K = 64 two; K = 32 four
    if (a->e_scale) {           // block-tail epilogue: the last 1x1 conv of a stage-1, -2 or -3 bottleneck (BatchNorm + ReLU + gate prologue)
        if constexpr (PRO == 1 && NI == 10) { if (a->K == 320) { launch<NI, WAVES, PRO, 10, 1, 1, false, true>(a, stream, dv); return true; } }
        if constexpr (PRO == 1 && NI == 8) { if (a->K == 128) { launch<NI, WAVES, PRO, 4, 1, 2, false, true>(a, stream, dv); return true; } }
        if constexpr (PRO == 1 && NI == 4) { if (a->K == 64) { launch<NI, WAVES, PRO, 2, 2, 2, false, true>(a, stream, dv); return true; } }
        // the stem: BatchNorm + ReLU of the 3x3 stem conv (K = 32 patch columns), no residual
        if constexpr (PRO == 0 && NI == 2) { if (a->K == 32) { launch<NI, WAVES, PRO, 1, 4, 2, false, true>(a, stream, dv); return true; } }
        return false;
    }
    switch (a->K) {
        case 320:
            if (a->wg_dw) return false;
            // (with the gate prologue the two gate rows must fit the wave's staging buffer: slices of 128 columns and more)
            if constexpr (PRO == 0) { launch<NI, WAVES, PRO, 10, 2, 1>(a, stream, dv); return true; } else if constexpr (PRO != 1 || NI >= 8) { launch<NI, WAVES, PRO, 10, 1, 1>(a, stream, dv); return true; } else return false;
        // (the fused weight gradient at K = 128, N = 64 needs 128 more accumulator registers than two waves per SIMD have: 319 spills)
        case 128: if (a->wg_dw) return false; launch<NI, WAVES, PRO, 4, 1, 2>(a, stream, dv); return true;
        case 64:
            if constexpr (PRO == 2 && NI == 2) { if (a->wg_dw) { launch<NI, WAVES, PRO, 2, 2, 2, true>(a, stream, dv); return true; } }
            if (a->wg_dw) return false;
            launch<NI, WAVES, PRO, 2, 2, 2>(a, stream, dv); return true;
        case 32: if (a->wg_dw) return false; if constexpr (PRO != 2) { launch<NI, WAVES, PRO, 1, 4, 2>(a, stream, dv); return true; } else return false;
        default: return false;
    }
}

}  // namespace

int stes_stream_slice(const stes_gemm_args* a) {
    int nb = 0;
    if (a->N == 32 || a->N == 64 || a->N == 128 || a->N == 160) nb = a->N;
    else if (a->N == 320) nb = 160;    // (80-column slices, two workgroups per CU: 99 us instead of 81 -- the operand loads double)
    else if (a->N % 128 == 0 && a->N <= 1024) nb = 128;     // wide layers with a short K (stage 4's first 1x1 conv): 128-column slices
    const bool bnb = a->a_raw != nullptr;
    const bool k_ok = a->K == 320 || a->K == 128 || a->K == 64 || (a->K == 32 && !bnb);
    const bool fits = nb > 0 && k_ok && !a->c_f32 && a->M >= 8192 && a->ldc % 8 == 0 && (a->k0 == 0 || a->k0 % 8 == 0) &&
                      (a->n_split == 0 || (a->n_split % 8 == 0 && a->ldc0 % 8 == 0)) &&
                      // epilogue options live in the tiled kernels only
                      !a->relu_mask && !a->bias && a->act == 0 && (a->out_scale == 0.f || a->out_scale == 1.f) &&
                      (!a->e_scale || (!a->wg_dw && !a->col_sum && a->n_split == 0)) &&
                      // the gate rows [2][K] f32 share the wave's staging buffer; a tile spans at most two images
                      (!a->a_scale || 8 * a->K <= 16 * (nb * 2 + 16)) && (!a->a_gate || (a->a_scale && a->gate_group >= 32)) &&
                      (long)a->M * (a->ldc > a->ldc0 ? a->ldc : a->ldc0) < (1L << 40);
    return fits ? nb : 0;
}

int stes_gemm_stream(const stes_gemm_args* a, void* stream) {
    const int nb = stes_stream_slice(a);
    if (nb == 0) return 0;
    StreamDivs dv;
    const bool mapped = a->map_stride > 1 || (a->wg_dw && a->wg_map_stride > 1);
    dv.hw = make_fastdiv(mapped ? a->Ho * a->Wo : 1);
    dv.wo = make_fastdiv(mapped ? a->Wo : 1);
    dv.gate = make_fastdiv(a->a_gate ? a->gate_group : 1);
    const int pro = a->a_raw ? 2 : (a->a_scale ? 1 : 0);
    hipStream_t s = (hipStream_t)stream;
    bool ok = false;
#define BY_PRO(NI_, W_) (pro == 2 ? launch_k<NI_, W_, 2>(a, s, dv) : (pro == 1 ? launch_k<NI_, W_, 1>(a, s, dv) : launch_k<NI_, W_, 0>(a, s, dv)))
    switch (nb) {
        case 32: ok = BY_PRO(2, 4); break;
        case 64: ok = BY_PRO(4, 4); break;
        case 128: ok = BY_PRO(8, 8); break;
        default: ok = BY_PRO(10, 8); break;
    }
#undef BY_PRO
    return ok ? 1 : 0;
}
