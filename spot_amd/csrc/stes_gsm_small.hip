// Gated Shift Module (reference model/shift.py:103-130) on slices of 8 or 16 channels -- the blocks at 112 x 112 and 56 x 56,
// where a position's slice is one or two 16-byte chunks and the fp32 gate [M][2] is half as large as the slice itself.
//
// Forward in one kernel: gate = tanh(conv3d_{3x3x3, groups=2}(relu(bn3d(x)))) and G = shift(gate * x) + (x - gate * x).
// A workgroup owns a tile of TH full rows of one clip and walks a segment of frames.  Per frame step f it
//   - takes the raw slice of frame f from memory exactly once: every thread owns up to four 16-byte chunks of the tile (the
//     same ones in every frame) plus a share of the two halo rows, keeps the raw values of its chunks for the last four
//     frames in registers, and writes relu(bn(x)) into a three-frame ring of halo tiles in LDS;
//   - convolves frame f - 1 on the matrix pipe: the 27 taps x `fold` channels are the contraction dimension (a tap's
//     channels of one position are ONE ds_read_b128 of the ring, which is exactly an MFMA B fragment with the position as
//     the column), the two groups are rows 0 and 1 of the A operand (weights, zero outside a row's group; rows 2..15 are
//     unused -- the matrix pipe has the time), so a 16-position tile costs ceil(27 * fold / 32) reads and MFMAs and no
//     other arithmetic;
//   - applies tanh to the frame's pre-activations spread over all threads (they travel through a three-frame ring of gates in
//     LDS, from where the frame's gate rows also leave for memory as one contiguous run);
//   - emits G of frame f - 2 from the registers and the gate ring: whole 16-byte chunks, each written once.
// Nothing is read twice from memory except the halo rows (2 of TH + 2) and the four extra frames at a segment's ends.
#include "stes_common.h"
#include "stes_gsm_small.h"

namespace {

constexpr int SG_OWN = 4;        // 16-byte chunks of the tile per thread
constexpr int SG_HALO = 2;       // ... of the two halo rows
constexpr int SG_UNIT = 7;       // MFMA k-steps whose operand reads are in flight together

struct SmallGeom {
    int B, T, H, W, TH, TSEG;
    FastDiv fd_W, fd_WN;         // W, W * chunks per position
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// tanh(x) = 1 - 2 / (1 + e^(2x)) on the hardware exp2 and reciprocal (1 ulp each): absolute error below 3e-7, saturating
// correctly at both ends (e^(2x) = inf -> 1, 0 -> -1); a sixth of the instructions of the library routine
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

template <int FOLD>
__global__ __launch_bounds__(256) void gsm_fwd_small_kernel(const bf16* __restrict__ Xc, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const bf16* __restrict__ wt,
                                                            const float* __restrict__ b3d, float* __restrict__ gate, bf16* __restrict__ G,
                                                            SmallGeom gm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
    constexpr int NCH = FOLD / 8, HALF = FOLD / 2;
    constexpr int NQ = 27 * NCH;                 // 8-channel pieces of the contraction dimension
    constexpr int KS = (NQ + 3) / 4;             // MFMA k-steps of 32
    const int T = gm.T, H = gm.H, W = gm.W, TH = gm.TH, W2 = W + 2;
    const int h0 = blockIdx.x * TH, t0 = blockIdx.y * gm.TSEG, b = blockIdx.z;
    const int t1 = t0 + gm.TSEG < T ? t0 + gm.TSEG : T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int slot_bytes = (TH + 2) * W2 * FOLD * 2;
    const int npos = TH * W;                     // positions of the tile (rows past the image included)
    const int rows_in = H - h0 < TH ? H - h0 : TH;
    unsigned char* ring = sg_smem;                                                   // [3][TH + 2][W + 2][FOLD] bf16
    float* sc_s = reinterpret_cast<float*>(sg_smem + 3 * slot_bytes);                // [FOLD] scale, [FOLD] shift
    float* sh_s = sc_s + FOLD;
    f32x2* gring = reinterpret_cast<f32x2*>(sh_s + FOLD);                            // [3][npos] (group 0, group 1) + one spare entry
    const long plane = (long)H * W;

    if (tid < FOLD) { sc_s[tid] = scale[tid]; sh_s[tid] = shift[tid]; }
    // the halo columns stay zero in every slot
    for (int i = tid; i < 3 * (TH + 2) * 2 * NCH; i += 256) {
        const int ch = i % NCH, side = (i / NCH) & 1, r = i / (2 * NCH);             // r: row over the three slots
        *reinterpret_cast<bf16x8*>(ring + ((long)(r * W2 + (side ? W + 1 : 0)) * NCH + ch) * 16) = zero8();
    }

    // ---- this thread's chunks: offset inside a frame (-1: outside the image), offset inside a ring slot (-1: none)
    int goff[SG_OWN], loff[SG_OWN], opos[SG_OWN], och[SG_OWN];
#pragma unroll
    for (int j = 0; j < SG_OWN; ++j) {
        const int u = tid + 256 * j;
        goff[j] = -1; loff[j] = -1; opos[j] = 0; och[j] = 0;
        if (u < npos * NCH) {
            uint32_t o, ch, hr, ww;
            if (NCH == 1) { o = u; ch = 0; } else { o = u / NCH; ch = u % NCH; }
            fdivmod(o, gm.fd_W, hr, ww);
            opos[j] = (int)o; och[j] = (int)ch;
            loff[j] = ((((int)hr + 1) * W2 + (int)ww + 1) * NCH + (int)ch) * 16;
            if (h0 + (int)hr < H) goff[j] = ((h0 + (int)hr) * W + (int)ww) * FOLD + (int)ch * 8;
        }
    }
    int hgoff[SG_HALO], hloff[SG_HALO];
#pragma unroll
    for (int j = 0; j < SG_HALO; ++j) {
        const int v = tid + 256 * j;
        hgoff[j] = -1; hloff[j] = -1;
        if (v < 2 * W * NCH) {
            uint32_t side, rem;
            fdivmod((uint32_t)v, gm.fd_WN, side, rem);
            const int ww = (int)rem / NCH, ch = (int)rem % NCH;
            const int hh = side ? h0 + TH : h0 - 1;
            hloff[j] = (((side ? TH + 1 : 0) * W2 + ww + 1) * NCH + ch) * 16;
            if (hh >= 0 && hh < H) hgoff[j] = (hh * W + ww) * FOLD + ch * 8;
        }
    }

    // ---- weights as MFMA A fragments: row 0 = group 0, row 1 = group 1, piece q = 4s + fq of k-step s is (tap, 8 channels)
    bf16x8 afrag[KS];
    int rel_s[KS], kt_s[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int q = 4 * s + fq;
        const bool on = q < NQ;
        const int tap = on ? q / NCH : 0, chunk = on ? q % NCH : 0;
        const int kt = tap / 9, kh = (tap - kt * 9) / 3, kw = tap - kt * 9 - kh * 3;
        kt_s[s] = kt;
        rel_s[s] = ((kh * W2 + kw) * NCH + chunk) * 16;
        bf16x8 a = zero8();
        if (on && fr < 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = chunk * 8 + e;
                if (c / HALF == fr) a[e] = wt[((long)fr * 27 + tap) * HALF + (c - fr * HALF)];
            }
        }
        afrag[s] = a;
    }
    const float bias0 = b3d[0], bias1 = b3d[1];

    // Branch-free staging: a thread without a chunk (or whose chunk lies outside the image) loads a valid address anyway,
    // writes to a spare 16 bytes of LDS, and selects zeros -- per-chunk branches cost more than the work they skip.
    const int dump = (3 * slot_bytes + 2 * FOLD * 4 + (3 * npos + 1) * 8 + 15) & ~15;    // spare 16 bytes behind everything else
    int goffc[SG_OWN], loffc[SG_OWN], hgoffc[SG_HALO], hloffc[SG_HALO], hch[SG_HALO];
#pragma unroll
    for (int j = 0; j < SG_OWN; ++j) { goffc[j] = goff[j] >= 0 ? goff[j] : 0; loffc[j] = loff[j] >= 0 ? loff[j] : dump; }
#pragma unroll
    for (int j = 0; j < SG_HALO; ++j) {
        hgoffc[j] = hgoff[j] >= 0 ? hgoff[j] : 0; hloffc[j] = hloff[j] >= 0 ? hloff[j] : dump;
        hch[j] = NCH > 1 && hloff[j] >= 0 ? (hloff[j] >> 4) % NCH : 0;
    }
    // (fold 8: every chunk has the same eight channels, so their scale / shift stay in registers)
    float sc8[FOLD == 8 ? 8 : 1], sh8[FOLD == 8 ? 8 : 1];
    if constexpr (FOLD == 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc8[e] = scale[e]; sh8[e] = shift[e]; }
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto keep_if = [](const bf16x8& v, bool on) {                                         // v or zeros, without a branch
        u32x4 r = __builtin_bit_cast(u32x4, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = on ? r[k] : 0u;
        return __builtin_bit_cast(bf16x8, r);
    };
    bf16x8 xin[SG_OWN], xh[SG_HALO];
    bf16x8 x1[SG_OWN], x2[SG_OWN], x3[SG_OWN];       // raw chunks of frames f - 1, f - 2, f - 3 once frame f is committed
#pragma unroll
    for (int j = 0; j < SG_OWN; ++j) { x1[j] = zero8(); x2[j] = zero8(); x3[j] = zero8(); xin[j] = zero8(); }
#pragma unroll
    for (int j = 0; j < SG_HALO; ++j) xh[j] = zero8();
    auto issue = [&](int f) {                        // frame f -> registers (a frame outside the clip: nothing, commit writes zeros)
        if (f < 0 || f >= T) return;
        const bf16* base = Xc + ((long)b * T + f) * plane * FOLD;
#pragma unroll
        for (int j = 0; j < SG_OWN; ++j) xin[j] = *reinterpret_cast<const bf16x8*>(base + goffc[j]);
#pragma unroll
        for (int j = 0; j < SG_HALO; ++j) xh[j] = *reinterpret_cast<const bf16x8*>(base + hgoffc[j]);
    };
    auto bn_relu = [&](const bf16x8& v, int ch) {
        bf16x8 o;
        if constexpr (FOLD == 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaxf((float)v[e] * sc8[e] + sh8[e], 0.f);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaxf((float)v[e] * sc_s[ch * 8 + e] + sh_s[ch * 8 + e], 0.f);
        }
        return o;
    };
    auto commit = [&](int f) {                       // registers -> BatchNorm + ReLU -> ring slot f mod 3; padding stays exactly zero
        const bool live = f >= 0 && f < T;
        unsigned char* slot = ring + ((f + 3) % 3) * slot_bytes;
#pragma unroll
        for (int j = 0; j < SG_OWN; ++j) {
            x3[j] = x2[j]; x2[j] = x1[j]; x1[j] = xin[j];
            *reinterpret_cast<bf16x8*>((loff[j] >= 0 ? slot : ring) + loffc[j]) = keep_if(bn_relu(xin[j], och[j]), live && goff[j] >= 0);
        }
#pragma unroll
        for (int j = 0; j < SG_HALO; ++j)
            *reinterpret_cast<bf16x8*>((hloff[j] >= 0 ? slot : ring) + hloffc[j]) = keep_if(bn_relu(xh[j], hch[j]), live && hgoff[j] >= 0);
    };

    __syncthreads();                                 // scale / shift visible
    issue(t0 - 2);
    for (int f = t0 - 2; f <= t1 + 1; ++f) {
        // after commit(f) the chunk registers hold x1 = frame f, x2 = f - 1, x3 = f - 2; the emission below needs the frame
        // before those too, which is what x3 held before this commit
        bf16x8 x4[SG_OWN];
#pragma unroll
        for (int j = 0; j < SG_OWN; ++j) x4[j] = x3[j];
        commit(f);
        if (f + 1 <= t1 + 1) issue(f + 1);
        __syncthreads();
        const int t = f - 1;                         // frame whose gate is computed now
        const bool do_gate = t >= 0 && t < T && t >= t0 - 1 && t <= t1;
        f32x2* gt = gring + ((t + 3) % 3) * npos;
        if (do_gate) {
            int base_s[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) base_s[s] = rel_s[s] + ((t + 2 + kt_s[s]) % 3) * slot_bytes;
            // Software pipeline over units of SG_UNIT k-steps: the fragment reads of the next unit are issued (all of them, ahead
            // of a scheduling barrier -- left alone, the compiler keeps one read in flight and every MFMA waits out an LDS round
            // trip) before the MFMAs of the current one, which run as two independent accumulation chains.  Buffer A holds
            // the first unit of a pair, buffer B the second: the two halves of a tile's k-steps (fold 16) or two tiles (fold 8).
            constexpr int UPT = KS / SG_UNIT;                                   // units per tile
            static_assert(KS % SG_UNIT == 0 && (UPT == 1 || UPT == 2), "7 or 14 k-steps");
            const int ntile = (npos + 15) >> 4, tiles_w = (ntile - wave + 3) >> 2;     // tiles of this wave
            const int npair = UPT == 2 ? tiles_w : (tiles_w + 1) >> 1;
            bf16x8 bva[SG_UNIT], bvb[UPT == 2 ? 1 : SG_UNIT];    // (fold 16 has no registers for the second buffer: its two units share one)
            auto tile_of = [&](int pair, int second) { return wave + 4 * (UPT == 2 ? pair : 2 * pair + second); };
            auto pos_of = [&](int tile) {
                const int o = tile * 16 + fr;
                const uint32_t oc = (uint32_t)(o < npos ? o : npos - 1);
                return (int)(oc + 2 * fdiv(oc, gm.fd_W)) * NCH * 16;          // row * (W + 2) + column = o + 2 * row
            };
            auto load_a = [&](int pair) {
                const int posoff = pos_of(tile_of(pair, 0));
#pragma unroll
                for (int i = 0; i < SG_UNIT; ++i) bva[i] = *reinterpret_cast<const bf16x8*>(ring + base_s[i] + posoff);
            };
            auto load_b = [&](int pair) {
                const int posoff = pos_of(tile_of(pair, 1));
#pragma unroll
                for (int i = 0; i < SG_UNIT; ++i) {
                    if constexpr (UPT == 2) bva[i] = *reinterpret_cast<const bf16x8*>(ring + base_s[SG_UNIT + i] + posoff);
                    else bvb[i] = *reinterpret_cast<const bf16x8*>(ring + base_s[i] + posoff);
                }
            };
            auto put = [&](int tile, const f32x4& c0, const f32x4& c1) {
                const int o = tile * 16 + fr;                                   // (lanes that hold no gate write to the spare entry)
                gt[(fq == 0 && o < npos) ? o : 3 * npos - (t + 3) % 3 * npos] = (f32x2){c0[0] + c1[0], c0[1] + c1[1]};
            };
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (UPT == 2) {
                for (int pair = 0; pair < npair; ++pair) {
                    load_a(pair);
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 acc0 = zero4, acc1 = zero4;                           // D[row 4fq + i][position fr]
#pragma unroll
                    for (int i = 0; i < SG_UNIT; ++i) {
                        if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[i], bva[i], acc1, 0, 0, 0);
                        else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[i], bva[i], acc0, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    load_b(pair);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < SG_UNIT; ++i) {
                        if (i & 1) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[SG_UNIT + i], bva[i], acc0, 0, 0, 0);
                        else acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[SG_UNIT + i], bva[i], acc1, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    put(tile_of(pair, 1), acc0, acc1);
                }
            } else {
                load_a(0);
                for (int pair = 0; pair < npair; ++pair) {
                    load_b(pair);
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 acc0 = zero4, acc1 = zero4;
#pragma unroll
                    for (int i = 0; i < SG_UNIT; ++i) {
                        if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[i], bva[i], acc1, 0, 0, 0);
                        else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[i], bva[i], acc0, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    put(tile_of(pair, 0), acc0, acc1);
                    acc0 = zero4; acc1 = zero4;
                    if (pair + 1 < npair) load_a(pair + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < SG_UNIT; ++i) {
                        if (i & 1) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[i], bvb[i], acc0, 0, 0, 0);
                        else acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[i], bvb[i], acc1, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    put(tile_of(pair, 1), acc0, acc1);
                }
            }
        }
        __syncthreads();
        if (do_gate) {
            const bool store = t >= t0 && t < t1;
            float* gout = gate + (((long)b * T + t) * plane + (long)h0 * W) * 2;
            float* gl = reinterpret_cast<float*>(gt);
            for (int i = tid; i < rows_in * W * 2; i += 256) {
                const float v = fast_tanh(gl[i] + ((i & 1) ? bias1 : bias0));
                gl[i] = v;
                if (store) gout[i] = v;
            }
        }
        __syncthreads();
        const int tau = f - 2;                       // frame whose output is complete now: x[tau - 1] = x4, x[tau] = x3, x[tau + 1] = x2
        if (tau >= t0 && tau < t1) {
            const f32x2* g_cur = gring + ((tau + 3) % 3) * npos;
            const f32x2* g_nxt = gring + ((tau + 4) % 3) * npos;
            const f32x2* g_prv = gring + ((tau + 2) % 3) * npos;
            const bool has_n = tau + 1 < T, has_p = tau - 1 >= 0;
            bf16* obase = G + ((long)b * T + tau) * plane * FOLD;
#pragma unroll
            for (int j = 0; j < SG_OWN; ++j) {
                const f32x2 gc = g_cur[opos[j]];
                const float gn = has_n ? g_nxt[opos[j]][0] : 0.f, gp = has_p ? g_prv[opos[j]][1] : 0.f;
                bf16x8 out;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool grp1 = NCH == 1 ? e >= 4 : och[j] >= NCH / 2;
                    const float xv = (float)x3[j][e];
                    const float y = (grp1 ? gc[1] : gc[0]) * xv;
                    const float r = xv - y;
                    const float ys = grp1 ? (has_p ? gp * (float)x4[j][e] : 0.f) : (has_n ? gn * (float)x2[j][e] : 0.f);
                    out[e] = (bf16)(ys + r);
                }
                if (goff[j] >= 0) *reinterpret_cast<bf16x8*>(obase + goff[j]) = out;
            }
        }
    }
}

// ------------------------------------------------------------------ backward of the gate conv on the narrow slices
// dxbn[m,c] = [bn(x) > 0] * sum_tap w[c][tap] * dpre[m - off(tap), g(c)] (bf16 [M][FOLD]) and the BatchNorm3d-backward sums
// s_dy += sum_m dxbn, s_dyr += sum_m dxbn * x of the rounded result, in one pass.  A workgroup owns TH full rows of one clip
// and walks a segment of frames with a three-frame ring of halo tiles of the gate gradient (both groups of a position as one
// 8-byte entry) in LDS, so the gate gradient is read once.  A thread owns up to four positions with all FOLD channels; the
// nine (frame, row) tap rows are the outer loop, so that a tap row's 3 x FOLD weights (wave-uniform: scalar registers) are
// fetched once per frame and workgroup instead of once per position.  The sums stay in registers over the whole segment and
// leave through one atomic per channel and workgroup.
template <int FOLD, int SB_PPT>   // SB_PPT: positions per thread (its accumulators are SB_PPT x FOLD registers)
__global__ __launch_bounds__(256) void gsm_dxbn_small_kernel(const float* __restrict__ dpre, const bf16* __restrict__ Xc,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ w3d, bf16* __restrict__ dxbn, float* s_dy,
                                                             float* s_dyr, SmallGeom gm, StesDet det) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
    constexpr int HALF = FOLD / 2, NCH = FOLD / 8;
    const int T = gm.T, H = gm.H, W = gm.W, TH = gm.TH, W2 = W + 2, rows = TH + 2;
    const int h0 = blockIdx.x * TH, t0 = blockIdx.y * gm.TSEG, b = blockIdx.z;
    const int t1 = t0 + gm.TSEG < T ? t0 + gm.TSEG : T;
    const int tid = threadIdx.x;
    const int slot = rows * W2;                                  // entries of a ring slot
    f32x2* ring = reinterpret_cast<f32x2*>(sg_smem);             // [3][TH + 2][W + 2] (group 0, group 1)
    float* red = reinterpret_cast<float*>(ring + 3 * slot);      // [2][FOLD]
    const long plane = (long)H * W;
    const int npos = TH * W;
    for (int i = tid; i < 2 * FOLD; i += 256) red[i] = 0.f;

    // this thread's positions: entry of the tile's interior in a slot, offset inside a frame (-1: none / outside the image)
    int pent[SB_PPT], poff[SB_PPT];
#pragma unroll
    for (int j = 0; j < SB_PPT; ++j) {
        const int o = tid + 256 * j;
        uint32_t hr, ww;
        fdivmod((uint32_t)(o < npos ? o : 0), gm.fd_W, hr, ww);
        pent[j] = ((int)hr + 1) * W2 + (int)ww + 1;
        poff[j] = (o < npos && h0 + (int)hr < H) ? (h0 + (int)hr) * W + (int)ww : -1;
    }
    auto stage = [&](int f) {                                    // gate gradient of frame f -> ring slot f mod 3 (zeros outside)
        f32x2* dst = ring + ((f + 3) % 3) * slot;
        const bool live = f >= 0 && f < T;
        const float* src = dpre + ((long)b * T + (live ? f : 0)) * plane * 2;
        for (int i = tid; i < slot; i += 256) {
            uint32_t hr, wp;
            fdivmod((uint32_t)i, gm.fd_WN, hr, wp);               // (fd_WN: W + 2 here)
            const int hh = h0 + (int)hr - 1, ww = (int)wp - 1;
            f32x2 v = {0.f, 0.f};
            if (live && hh >= 0 && hh < H && ww >= 0 && ww < W) v = *reinterpret_cast<const f32x2*>(src + ((long)hh * W + ww) * 2);
            dst[i] = v;
        }
    };
    float sdy[FOLD], sdx[FOLD];
#pragma unroll
    for (int c = 0; c < FOLD; ++c) { sdy[c] = 0.f; sdx[c] = 0.f; }

    stage(t0 - 1);
    stage(t0);
    for (int t = t0; t < t1; ++t) {
        stage(t + 1);                                            // (slot of frame t - 2: its readers passed the barrier below)
        __syncthreads();
        float acc[SB_PPT][FOLD];
#pragma unroll
        for (int j = 0; j < SB_PPT; ++j)
#pragma unroll
            for (int c = 0; c < FOLD; ++c) acc[j][c] = 0.f;
        // output position q = p - (k - 1): tap (kt, kh, kw) reads the entry at frame t + 1 - kt, row + 1 - kh, column + 1 - kw
#pragma unroll 1
        for (int r = 0; r < 9; ++r) {                            // (a real loop: unrolled, all 27 x FOLD weights are fetched at once and spill)
            const int kt = r / 3, kh = r - kt * 3;
            const f32x2* sl = ring + ((t + 4 - kt) % 3) * slot + (1 - kh) * W2 + 1;
            const float* wr = w3d + r * 3;
            f32x2 dp[SB_PPT][3];
#pragma unroll
            for (int j = 0; j < SB_PPT; ++j)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) dp[j][kw] = sl[pent[j] - kw];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
                for (int c = 0; c < FOLD; ++c) {
                    const float wv = wr[c * 27 + kw];            // [2][HALF][27] flat = [FOLD][27]: wave-uniform
#pragma unroll
                    for (int j = 0; j < SB_PPT; ++j) acc[j][c] += wv * (c < HALF ? dp[j][kw][0] : dp[j][kw][1]);
                }
            }
        }
        const long fbase = ((long)b * T + t) * plane;
#pragma unroll
        for (int j = 0; j < SB_PPT; ++j) {
            const long m = fbase + (poff[j] >= 0 ? poff[j] : 0);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                float x[8], o[8];
                load8b(Xc + m * FOLD + ch * 8, x);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = ch * 8 + e;
                    o[e] = (x[e] * scale[c] + shift[c] > 0.f) ? acc[j][c] : 0.f;
                }
                const bf16x8 ob = pack8(o);
                if (poff[j] >= 0) {
                    *reinterpret_cast<bf16x8*>(dxbn + m * FOLD + ch * 8) = ob;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float r = (float)ob[e]; sdy[ch * 8 + e] += r; sdx[ch * 8 + e] += r * x[e]; }
                }
            }
        }
        __syncthreads();                                         // the slot of frame t - 1 is overwritten next
    }
    // sums: lanes -> wave (shuffles) -> workgroup (LDS) -> memory
#pragma unroll
    for (int c = 0; c < FOLD; ++c) {
        float a = sdy[c], d = sdx[c];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { a += __shfl_xor(a, m, 64); d += __shfl_xor(d, m, 64); }
        sdy[c] = a; sdx[c] = d;
    }
    STES_DET_WAVES(det, if ((tid & 63) == 0) {
        _Pragma("unroll") for (int c = 0; c < FOLD; ++c) { atomicAdd(&red[c], sdy[c]); atomicAdd(&red[FOLD + c], sdx[c]); }
    });
    __syncthreads();
    det_enter(det);
    if (tid < 2 * FOLD) atomicAdd((tid < FOLD ? s_dy : s_dyr) + (tid < FOLD ? tid : tid - FOLD), red[tid]);
    det_leave(det);
}

}  // namespace

int stes_gsm_small_forward(const uint16_t* Xc, const float* scale, const float* shift, const uint16_t* wt, const float* b3d, float* gate,
                           uint16_t* G, int B, int T, int H, int W, int fold, void* stream) {
    STES_CHECK(fold == 8 || fold == 16, STES_ERR_UNSUPPORTED, "stes_gsm_forward: the narrow-slice kernel is built for fold 8 and 16");
    const int nch = fold / 8;
    STES_CHECK(2 * W * nch <= 256 * SG_HALO && W * nch <= 256 * SG_OWN, STES_ERR_UNSUPPORTED, "stes_gsm_forward: image row too wide for the tile");
    auto lds_of = [&](int th) { return (size_t)3 * (th + 2) * (W + 2) * fold * 2 + (size_t)3 * th * W * 8 + 8 + 2 * fold * sizeof(float) + 32; };
    int th = 256 * SG_OWN / (W * nch);
    if (th > H) th = H;
    while (th > 1 && lds_of(th) > 78 * 1024) --th;      // two workgroups per compute unit
    // even out the row tiles (a ragged last tile costs a whole workgroup)
    const int rt = stes_div_up(H, th);
    th = stes_div_up(H, rt);
    STES_CHECK(lds_of(th) <= 160 * 1024, STES_ERR_UNSUPPORTED, "stes_gsm_forward: tile does not fit LDS");
    // frames per segment: every segment stages four frames and convolves two more than it emits; fewest steps over all rounds
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long slots = 2L * (cus > 0 ? cus : 256);
    int tseg = T; long best = -1;
    for (int cand = T; cand >= 4 || cand == T; cand = (cand + 1) / 2) {
        const long wgs = (long)rt * B * stes_div_up(T, cand);
        const long cost = ((wgs + slots - 1) / slots) * (cand + 4);
        if (best < 0 || cost < best) { best = cost; tseg = cand; }
        if (cand <= 4) break;
    }
    SmallGeom gm;
    gm.B = B; gm.T = T; gm.H = H; gm.W = W; gm.TH = th; gm.TSEG = tseg;
    gm.fd_W = make_fastdiv(W); gm.fd_WN = make_fastdiv(W * nch);
    const size_t lds = lds_of(th);
    const dim3 grid(rt, stes_div_up(T, tseg), B);
#define SMALL_CASE(F)                                                                                                              \
    do {                                                                                                                           \
        static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(&gsm_fwd_small_kernel<F>),                   \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;     \
        STES_CHECK(big_lds || lds <= 64 * 1024, STES_ERR_UNSUPPORTED, "stes_gsm_forward: cannot raise the dynamic LDS limit");     \
        hipLaunchKernelGGL((gsm_fwd_small_kernel<F>), grid, dim3(256), lds, (hipStream_t)stream, (const bf16*)Xc, scale, shift,   \
                           (const bf16*)wt, b3d, gate, (bf16*)G, gm);                                                              \
    } while (0)
    if (fold == 8) SMALL_CASE(8); else SMALL_CASE(16);
#undef SMALL_CASE
    STES_LAUNCH_CHECK();
    return STES_OK;
}

int stes_gsm_small_dxbn(const float* dpre, const uint16_t* Xc, const float* scale, const float* shift, const float* w3d, uint16_t* dxbn,
                        float* s_dy, float* s_dyr, int B, int T, int H, int W, int fold, void* stream) {
    // (32 channels: accumulators and sums of two positions already fill the register file -- one wave per SIMD)
    STES_CHECK(fold == 8 || fold == 16, STES_ERR_UNSUPPORTED, "stes_gsm_bwd_conv: the narrow-slice kernel is built for fold 8 and 16");
    const int SB_PPT = 4;
    STES_CHECK(W <= 256 * SB_PPT, STES_ERR_UNSUPPORTED, "stes_gsm_bwd_conv: image row too wide for the tile");
    auto lds_of = [&](int th) { return (size_t)3 * (th + 2) * (W + 2) * 8 + 2 * fold * sizeof(float); };
    int th = 256 * SB_PPT / W;
    if (th > H) th = H;
    while (th > 1 && lds_of(th) > 40 * 1024) --th;
    const int rt = stes_div_up(H, th);
    th = stes_div_up(H, rt);
    // frames per segment: a segment stages two frames more than it produces; at least about four workgroups per compute unit
    int tseg = T;
    while (tseg > 8 && (long)rt * B * stes_div_up(T, tseg) < 1024) tseg = (tseg + 1) / 2;
    SmallGeom gm;
    gm.B = B; gm.T = T; gm.H = H; gm.W = W; gm.TH = th; gm.TSEG = tseg;
    gm.fd_W = make_fastdiv(W); gm.fd_WN = make_fastdiv(W + 2);
    const dim3 grid(rt, stes_div_up(T, tseg), B);
    const StesDet det = stes_det_take(grid.x * grid.y * grid.z);
#define DXBN_CASE(F, P) hipLaunchKernelGGL((gsm_dxbn_small_kernel<F, P>), grid, dim3(256), lds_of(th), (hipStream_t)stream, dpre, (const bf16*)Xc, \
                                          scale, shift, w3d, (bf16*)dxbn, s_dy, s_dyr, gm, det)
    if (fold == 8) DXBN_CASE(8, 4); else DXBN_CASE(16, 4);
#undef DXBN_CASE
    STES_LAUNCH_CHECK();
    return STES_OK;
}
