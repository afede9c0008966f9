// Gated Shift Module (reference model/shift.py:83-161) and plain temporal shift (:21-78) on bf16 activations.
//
// The module acts on the first `fold` channels of the block input.  Its producer writes that slice a second time as a
// COMPACT tensor Xc [M][fold] (rows m = ((clip*T + t)*H + h)*W + w), so every kernel here streams unit-stride rows.
// Channel order everywhere is the SOURCE order; the reference's within-group interleave (view/permute at
// model/shift.py:124-128) is folded into the column order of the 1x1 convolution that consumes the result.
//
//   gate   : g[m,grp] = tanh(conv3d_{3x3x3,groups=2}(relu(bn3d(x)))) -- one workgroup per (clip, frame, row tile, group):
//            BN+ReLU applied once while the 3-frame halo tile is staged into LDS, taps unrolled from LDS with the conv
//            weights coming through scalar loads (they are wave-uniform);
//   shift  : G[m,c] = y[t+1 | t-1] + (x - y), y = g*x  (group 0 pulls t+1, group 1 pulls t-1, zero at the clip edge);
//   backward: gate-gradient (elementwise), conv3d weight gradient (reduction over positions on MFMA, operands
//            transposed through LDS with ds_read_b64_tr_b16), gradient at the BatchNorm3d output (27-tap gather of the
//            gate gradient staged in LDS, weights scalar), final input gradient (elementwise).
#include "stes_common.h"
#include "stes_gsm_small.h"

namespace {

struct Geom {
    int B, T, H, W, fold;
    long M;
    FastDiv fd_plane, fd_T, fd_W, fd_H, fd_nch;     // H*W, T, W, H, vector chunks per row
};
// frame index t of row m
__device__ __forceinline__ int frame_of(uint32_t m, const Geom& gm) {
    const uint32_t f = fdiv(m, gm.fd_plane);
    return (int)(f - fdiv(f, gm.fd_T) * gm.T);
}

// Workgroup -> (row tile, frame, clip) for the kernels that read a frame's two neighbours: consecutive workgroups go round-robin
// over the 8 XCDs, so with frames as the fast index the three readers of a frame sat on three XCDs and every L2 fetched it
// again.  Here a clip belongs to XCD (clip mod 8) and the frames of a clip follow each other on that XCD.  Launch
// 8 * ceil(B / 8) * T * row_tiles workgroups; false: no work for this one.
__device__ __forceinline__ bool xcd_clip_map(int row_tiles, int T, int B, int& rt, int& t, int& b) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int per_clip = T * row_tiles;
    const int cl = j / per_clip, rem = j - cl * per_clip;
    b = xcd + 8 * cl;
    t = rem / row_tiles;
    rt = rem - t * row_tiles;
    return b < B;
}

// The same idea for the elementwise kernels that read a row's neighbour one frame away: with clips a multiple of 8 (and a
// grid that is one), XCD x walks the clips x, x + 8, ... whole, so a frame and its neighbours pass through one L2.  `body(i)`
// gets the flat element index either way.
template <class F>
__device__ __forceinline__ void gsm_flat_walk(uint32_t total, uint32_t per_clip, int B, F&& body) {
    if ((B & 7) == 0 && (gridDim.x & 7) == 0) {
        const uint32_t xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, nbx = gridDim.x >> 3;
        for (uint32_t b = xcd; b < (uint32_t)B; b += 8)
            for (uint32_t i = jb * 256 + threadIdx.x; i < per_clip; i += nbx * 256) body(b * per_clip + i);
    } else {
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) body(i);
    }
}

__device__ __forceinline__ int dest_of(int c, int fold) {
    const int half = fold >> 1, quarter = fold >> 2;
    const int g = c / half, rem = c - g * half;
    const int i = rem / quarter, j = rem - i * quarter;
    return g * half + 2 * j + i;
}

// ------------------------------------------------------------------ weight prep: fp32 [2][half][27] -> bf16 [2][27][half]
__global__ void gsm_prep_kernel(const float* __restrict__ w, bf16* __restrict__ wt, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * half * 27) return;
    const int tap = i % 27, c = (i / 27) % half, g = i / (27 * half);
    wt[((long)g * 27 + tap) * half + c] = (bf16)w[i];
}

// ------------------------------------------------------------------ gate
// Workgroup = (row tile, frame segment, clip).  It walks its frames with a three-slot ring of BN+ReLU'd halo tiles in LDS
// (all `fold` channels of a position side by side), so every input frame is fetched once per workgroup; the next frame's
// tile travels global -> registers while the current frame is convolved.  Waves 0,1 evaluate group 0, waves 2,3 group 1:
// the conv weights (bf16 pairs) are wave-uniform scalar loads feeding v_dot2c_f32_bf16.
struct GateTile {
    int TH, TSEG, PS;                  // rows per tile, frames per segment, LDS position stride in 16-byte units
    FastDiv fd_nch, fd_W2, fd_spr;     // 16-byte chunks per position, W + 2, output strips per row
};
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// acc + a.lo*b.lo + a.hi*b.hi on packed bf16 pairs
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "s"(b), "v"(a));
    return acc;
}

// HALF: channels per group as a compile-time constant (0: read from the geometry).  With it the channel loop unrolls, the
// LDS reads and the scalar weight loads of a whole (kt, kh) tap row are issued together and waited for once -- at one or
// two waves per SIMD the kernel is bound by exactly those latencies.
template <int V, int WB, int HALF>   // V: channels per LDS read (4 when half == 4, else 8); WB: outputs per thread along w (4 or 1)
__global__ __launch_bounds__(256) void gsm_gate_kernel(const bf16* __restrict__ Xc, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const uint32_t* __restrict__ wt,
                                                       const float* __restrict__ b3d, float* __restrict__ gate, Geom gm, GateTile gt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int MAXI = 8;
    const int fold = gm.fold, half = HALF ? HALF : (fold >> 1), nch = fold >> 3;
    const int W = gm.W, H = gm.H, T = gm.T, W2 = W + 2, rows = gt.TH + 2;
    const int h0 = blockIdx.x * gt.TH, t0 = blockIdx.y * gt.TSEG, b = blockIdx.z;
    const int t1 = (t0 + gt.TSEG < T) ? t0 + gt.TSEG : T;
    const int slot_bytes = (rows * W2 + 4) * gt.PS * 16;      // + 4 positions of slack for ragged strips
    float* sc_s = reinterpret_cast<float*>(smem + 3 * slot_bytes);
    float* sh_s = sc_s + fold;
    for (int i = threadIdx.x; i < fold; i += 256) { sc_s[i] = scale[i]; sh_s[i] = shift[i]; }
    const long plane = (long)H * W;
    const int items = rows * W2 * nch;

    // per-thread staging slots: global element offset inside a frame (-1: zero padding), LDS byte offset, first channel
    int goff[MAXI], loff[MAXI], ch0[MAXI];
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
        const int i = threadIdx.x + k * 256;
        goff[k] = -1; loff[k] = -1; ch0[k] = 0;
        if (i < items) {
            uint32_t pos, ch, hr, wp;
            fdivmod((uint32_t)i, gt.fd_nch, pos, ch);
            fdivmod(pos, gt.fd_W2, hr, wp);
            const int hh = h0 + (int)hr - 1, ww = (int)wp - 1;
            loff[k] = ((int)pos * gt.PS + (int)ch) * 16;
            ch0[k] = (int)ch * 8;
            if (hh >= 0 && hh < H && ww >= 0 && ww < W) goff[k] = (hh * W + ww) * fold + (int)ch * 8;
        }
    }
    bf16x8 stage[MAXI];
    auto fetch = [&](int f) {           // frame f of this clip -> registers (zeros outside the clip)
        const bool live = f >= 0 && f < T;
        const bf16* base = Xc + ((long)b * T + (live ? f : 0)) * plane * fold;
#pragma unroll
        for (int k = 0; k < MAXI; ++k) stage[k] = (live && goff[k] >= 0) ? *reinterpret_cast<const bf16x8*>(base + goff[k]) : zero8();
        return live;
    };
    auto commit = [&](int f, bool live) {   // registers -> BN + ReLU -> ring slot f mod 3; padding stays exactly zero
        unsigned char* slot = smem + ((f + 3) % 3) * slot_bytes;
#pragma unroll
        for (int k = 0; k < MAXI; ++k) {
            if (loff[k] < 0) continue;
            bf16x8 v = stage[k];
            if (live && goff[k] >= 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16)fmaxf((float)v[e] * sc_s[ch0[k] + e] + sh_s[ch0[k] + e], 0.f);
            }
            *reinterpret_cast<bf16x8*>(slot + loff[k]) = v;
        }
    };

    __syncthreads();                     // scale/shift visible
    bool live = fetch(t0 - 1); commit(t0 - 1, live);
    live = fetch(t0); commit(t0, live);
    live = fetch(t0 + 1);
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7), tg = threadIdx.x & 127;     // wave-uniform -> SGPR
    const float bias = b3d[grp];
    const uint32_t* wg = wt + (grp * 27 * half) / 2;      // [27][half/2] bf16 pairs, wave-uniform
    const int nstrips = gt.TH * (int)gt.fd_spr.d;
    const int pstride = gt.PS * 16;
    typedef uint32_t xvec __attribute__((ext_vector_type(V / 2)));
    for (int t = t0; t < t1; ++t) {
        __syncthreads();                 // frame t-2's slot is no longer read
        commit(t + 1, live);
        __syncthreads();
        if (t + 1 < t1) live = fetch(t + 2);
        const unsigned char* s0 = smem + ((t + 2) % 3) * slot_bytes;     // frame t-1
        const unsigned char* s1 = smem + (t % 3) * slot_bytes;
        const unsigned char* s2 = smem + ((t + 1) % 3) * slot_bytes;
        for (int s = tg; s < nstrips; s += 128) {
            uint32_t hr, sw;
            fdivmod((uint32_t)s, gt.fd_spr, hr, sw);
            const int hh = h0 + (int)hr, w0 = (int)sw * WB;
            if (hh >= H) break;
            // WB == 4: one accumulator per output of the strip; WB == 1: four partial sums of the one output --
            // either way four independent v_dot2c chains
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const unsigned char* sl = kt == 0 ? s0 : (kt == 1 ? s1 : s2);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const unsigned char* rowp = sl + (((int)hr + kh) * W2 + w0) * pstride + grp * half * 2;
                    const uint32_t* wrow = wg + (((kt * 3 + kh) * 3) * half) / 2;
                    auto chunk = [&](int c) {
                        if (WB == 4) {
                            xvec x[6];
#pragma unroll
                            for (int j = 0; j < 6; ++j) x[j] = *reinterpret_cast<const xvec*>(rowp + j * pstride + c * 2);
#pragma unroll
                            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                                for (int e = 0; e < V / 2; ++e) {
                                    const uint32_t wv = wrow[(kw * half + c) / 2 + e];
#pragma unroll
                                    for (int o = 0; o < 4; ++o) acc[o] = dot2_bf16(x[o + kw][e], wv, acc[o]);
                                }
                        } else {
#pragma unroll
                            for (int kw = 0; kw < 3; ++kw) {
                                const xvec x = *reinterpret_cast<const xvec*>(rowp + kw * pstride + c * 2);
#pragma unroll
                                for (int e = 0; e < V / 2; ++e) {
                                    const int a = V == 4 ? ((kw & 1) * 2 + e) : e;        // four independent chains either way
                                    acc[a] = dot2_bf16(x[e], wrow[(kw * half + c) / 2 + e], acc[a]);
                                }
                            }
                        }
                    };
                    if (HALF > 0) {
#pragma unroll
                        for (int c = 0; c < HALF; c += V) chunk(c);
                    } else {
                        for (int c = 0; c < half; c += V) chunk(c);
                    }
                }
            }
            // the dot instructions are opaque to the compiler's hazard recogniser: a different VALU opcode may not read or
            // overwrite a v_dot2c result for 4 wait states, so drain the pipeline before the accumulators are consumed
            asm volatile("s_nop 4" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            const long m = ((long)b * T + t) * plane + (long)hh * W + w0;
            if (WB == 4) {
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (w0 + o < W) gate[(m + o) * 2 + grp] = tanhf(acc[o] + bias);
            } else {
                gate[m * 2 + grp] = tanhf((acc[0] + acc[1]) + (acc[2] + acc[3]) + bias);
            }
        }
    }
}

// ------------------------------------------------------------------ shift (forward), thread per (row, V-channel chunk)
template <int V>
__global__ __launch_bounds__(256) void gsm_shift_kernel(const bf16* __restrict__ Xc, const float* __restrict__ gate,
                                                        bf16* __restrict__ G, Geom gm) {
    typedef __bf16 vecb __attribute__((ext_vector_type(V)));
    const int fold = gm.fold, half = fold >> 1, nch = fold / V;
    const long plane = (long)gm.H * gm.W;
    const uint32_t total = (uint32_t)(gm.M * nch);
    gsm_flat_walk(total, (uint32_t)(gm.T * plane * nch), gm.B, [&](uint32_t i) {
        uint32_t mu, chk;
        fdivmod(i, gm.fd_nch, mu, chk);
        const long m = mu; const int c0 = (int)chk * V;
        const int g = c0 >= half;
        const int t = frame_of(mu, gm);
        const vecb x = *reinterpret_cast<const vecb*>(Xc + m * fold + c0);
        const float gt = gate[m * 2 + g];
        const int tn = g == 0 ? t + 1 : t - 1;
        const bool has = tn >= 0 && tn < gm.T;
        const long mn = g == 0 ? m + plane : m - plane;
        vecb xn;
        float gn = 0.f;
        if (has) { xn = *reinterpret_cast<const vecb*>(Xc + mn * fold + c0); gn = gate[mn * 2 + g]; }
        vecb o;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xv = (float)x[k];
            const float y = gt * xv;
            const float r = xv - y;
            const float ys = has ? gn * (float)xn[k] : 0.f;
            o[k] = (bf16)(ys + r);
        }
        *reinterpret_cast<vecb*>(G + m * fold + c0) = o;
    });
}

// ------------------------------------------------------------------ backward: gate pre-activation gradient
// dpre[m,g] = (1 - gate^2) * sum_{c in g} x[m,c] * (dG[m -+ plane, c] - dG[m, c]);  thread per (m, g)
template <int V>
__global__ __launch_bounds__(256) void gsm_bwd_gate_kernel(const bf16* __restrict__ dG, const bf16* __restrict__ Xc,
                                                           const float* __restrict__ gate, float* __restrict__ dpre, Geom gm) {
    typedef __bf16 vecb __attribute__((ext_vector_type(V)));
    const int fold = gm.fold, half = fold >> 1;
    const long plane = (long)gm.H * gm.W;
    const uint32_t total = (uint32_t)(gm.M * 2);
    gsm_flat_walk(total, (uint32_t)(gm.T * plane * 2), gm.B, [&](uint32_t i) {
        const long m = i >> 1; const int g = (int)(i & 1);
        const int t = frame_of(i >> 1, gm);
        const int tp = g == 0 ? t - 1 : t + 1;                 // frame whose output used y[t]
        const bool has = tp >= 0 && tp < gm.T;
        const long mp = g == 0 ? m - plane : m + plane;
        const long base = m * fold + g * half, basep = mp * fold + g * half;
        float acc = 0.f;
        for (int c = 0; c < half; c += V) {
            const vecb x = *reinterpret_cast<const vecb*>(Xc + base + c);
            const vecb d0 = *reinterpret_cast<const vecb*>(dG + base + c);
            if (has) {
                const vecb d1 = *reinterpret_cast<const vecb*>(dG + basep + c);
#pragma unroll
                for (int k = 0; k < V; ++k) acc += ((float)d1[k] - (float)d0[k]) * (float)x[k];
            } else {
#pragma unroll
                for (int k = 0; k < V; ++k) acc -= (float)d0[k] * (float)x[k];
            }
        }
        const float gt = gate[i];
        dpre[i] = acc * (1.f - gt * gt);
    });
}

// The same for groups of 16 channels and more (half % 8 == 0, half >= 16): thread per 16-byte chunk, so that a wave's loads are
// one contiguous run of rows instead of 64 runs of `half` channels; the chunk sums of a (row, group) meet in LDS.
__global__ __launch_bounds__(256) void gsm_bwd_gate_rows_kernel(const bf16* __restrict__ dG, const bf16* __restrict__ Xc,
                                                                const float* __restrict__ gate, float* __restrict__ dpre, Geom gm,
                                                                int rows_per_block, FastDiv fd_nch) {
    __shared__ float part[256];
    const int fold = gm.fold, nch = fold >> 3, hch = nch >> 1;
    const long plane = (long)gm.H * gm.W;
    const long n_blocks = (gm.M + rows_per_block - 1) / rows_per_block;
    uint32_t r, ch;
    fdivmod((uint32_t)threadIdx.x, fd_nch, r, ch);
    const bool active = (int)r < rows_per_block;
    const int g = (int)ch >= hch;
    // (row blocks in XCD-local clip order where the shapes allow it: see gsm_flat_walk)
    const long bpc = (long)gm.T * plane / rows_per_block;                   // row blocks per clip
    const bool by_xcd = (gm.B & 7) == 0 && (gridDim.x & 7) == 0 && bpc * rows_per_block == (long)gm.T * plane;
    const long nbx = gridDim.x >> 3, per_xcd = by_xcd ? (gm.B >> 3) * bpc : 0;
    const long first = by_xcd ? (long)(blockIdx.x >> 3) : (long)blockIdx.x, last = by_xcd ? per_xcd : n_blocks, step = by_xcd ? nbx : (long)gridDim.x;
    for (long it = first; it < last; it += step) {
        const long blk = by_xcd ? ((blockIdx.x & 7) + 8 * (it / bpc)) * bpc + it % bpc : it;
        const long m = blk * rows_per_block + r;
        float acc = 0.f;
        if (active && m < gm.M) {
            const int t = frame_of((uint32_t)m, gm);
            const int tp = g == 0 ? t - 1 : t + 1;             // frame whose output used y[t]
            const long off = m * fold + (long)ch * 8;
            float x[8], d0[8];
            load8b(Xc + off, x); load8b(dG + off, d0);
            if (tp >= 0 && tp < gm.T) {
                float d1[8];
                load8b(dG + off + (g == 0 ? -plane : plane) * fold, d1);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += (d1[k] - d0[k]) * x[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc -= d0[k] * x[k];
            }
        }
        part[threadIdx.x] = acc;
        __syncthreads();
        if ((int)threadIdx.x < rows_per_block * 2) {
            const int rr = threadIdx.x >> 1, gg = threadIdx.x & 1;
            const long mm = blk * rows_per_block + rr;
            if (mm < gm.M) {
                float sum = 0.f;
                for (int c = 0; c < hch; ++c) sum += part[rr * nch + gg * hch + c];
                const float gt = gate[mm * 2 + gg];
                dpre[mm * 2 + gg] = sum * (1.f - gt * gt);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ backward: conv3d weight / bias gradient on MFMA
// dw[grp][c][tap] += sum_p xbr[p, c] * dpre[p - off(tap), grp]: reduction over positions p.
// Per 32 positions: A = Dm^T (rows = taps, 27 -> 32), B = Xb (cols = channels), both read transposed from LDS.
// grid (workgroups, 2 groups); each workgroup walks 32-position chunks and keeps its accumulators in registers.
__device__ __forceinline__ int rho32(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

template <int NCT, int POS>   // NCT: 16-channel tiles per group (half <= 16*NCT); POS: positions per wave pass
__global__ __launch_bounds__(256) void gsm_bwd_wgrad_kernel(const float* __restrict__ dpre, const bf16* __restrict__ Xc,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ dw, float* __restrict__ db, Geom gm, long n_chunks,
                                                            FastDiv fd_nch8, StesDet det) {
    // POS == 64: lane = position, one 8-byte load brings the gate gradient of both groups, the groups take turns in the
    //            Dm buffer;  POS == 32: lane = (position, group), both groups' Dm rows are live together.
    constexpr int LDD = 48;                                        // Dm row: 32 taps + 16 pad (elements)
    constexpr int LDX = 32 * NCT + 16;                             // Xb row: both groups' channels + tile overhang + pad
    __shared__ __attribute__((aligned(16))) bf16 Dm[4][64 * LDD];  // per wave: [64 rows (rho within 32)][taps]
    __shared__ __attribute__((aligned(16))) bf16 Xb[4][POS * LDX]; // per wave: [POS positions (rho within 32)][channels]
    static_assert(sizeof(bf16) * 4 * POS * LDX >= sizeof(float) * 32 * NCT * 27, "result staging must fit the operand buffer");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int fold = gm.fold, half = fold >> 1, nch = fold >> 3;
    const int H = gm.H, W = gm.W, T = gm.T;
    const long plane = (long)H * W;
    f32x4 acc[2][2][NCT];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NCT; ++j) acc[g][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum0 = 0.f, bsum1 = 0.f;
    bf16* dm = Dm[wave];
    bf16* xb = Xb[wave];
    const int r_lo = rho32(8 * fq + (fr >> 2)), r_hi = rho32(8 * fq + (fr >> 2) + 4);
    const int cc = 4 * (fr & 3);
    const int pl = POS == 64 ? lane : (lane & 31), lg = POS == 64 ? 0 : (lane >> 5);
    const int my_row = (pl & 32) + rho32(pl & 31);

    auto mma = [&](int g, int dm_row0, int x_row0) {
        bf16x8 fa[2], fb[NCT];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(dm + (dm_row0 + r_lo) * LDD + i * 16 + cc));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(dm + (dm_row0 + r_hi) * LDD + i * 16 + cc));
            fa[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(xb + (x_row0 + r_lo) * LDX + g * half + j * 16 + cc));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(xb + (x_row0 + r_hi) * LDX + g * half + j * 16 + cc));
            fb[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NCT; ++j) acc[g][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[g][i][j], 0, 0, 0);
    };
    auto put_row = [&](int row, const bf16 (&v)[32]) {
        bf16x8* dst = reinterpret_cast<bf16x8*>(dm + row * LDD);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = v[q * 8 + k];
            dst[q] = o;
        }
    };

    // every wave owns whole chunks: no workgroup barriers inside the loop, only wave-local LDS traffic
    const long wstride = (long)gridDim.x * 4;
    for (long ck = (long)blockIdx.x * 4 + wave; ck < n_chunks; ck += wstride) {
        const long p0 = ck * POS;
        const long m = p0 + pl;
        const bool in = m < gm.M;
        uint32_t r1, w_u, r2, h_u;
        fdivmod(in ? (uint32_t)m : 0u, gm.fd_W, r1, w_u);
        fdivmod(r1, gm.fd_H, r2, h_u);
        const int w_ = (int)w_u, h_ = (int)h_u, t_ = (int)(r2 - fdiv(r2, gm.fd_T) * T);
        // tap k reads the gate gradient at offset -(k - 1): k = 0 needs coordinate + 1 inside, k = 2 needs coordinate - 1 inside
        const bool vt[3] = {t_ + 1 < T, true, t_ >= 1}, vh[3] = {h_ + 1 < H, true, h_ >= 1}, vw[3] = {w_ + 1 < W, true, w_ >= 1};
        bf16 v0[32], v1[32];
#pragma unroll
        for (int tap = 0; tap < 32; ++tap) {
            float a = 0.f, b = 0.f;
            if (tap < 27) {
                const int kt = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                const bool ok = in && vt[kt] && vh[kh] && vw[kw];
                const long src = ok ? m - (long)(kt - 1) * plane - (long)(kh - 1) * W - (kw - 1) : (in ? m : 0);
                if (POS == 64) {
                    const float2 d = *reinterpret_cast<const float2*>(dpre + src * 2);
                    a = ok ? d.x : 0.f; b = ok ? d.y : 0.f;
                } else {
                    const float d = dpre[src * 2 + lg];
                    a = ok ? d : 0.f;
                }
                if (tap == 13) { bsum0 += a; bsum1 += b; }
            }
            v0[tap] = (bf16)a; v1[tap] = (bf16)b;
        }
        // Xb: relu(bn(x)) for POS positions x fold channels, 8 channels per item (coalesced over the compact rows)
        for (int i = lane; i < POS * nch; i += 64) {
            const int ps = (int)fdiv((uint32_t)i, fd_nch8), c0 = (i - ps * nch) * 8;
            const long mm = p0 + ps;
            bf16x8 v = zero8();
            if (mm < gm.M) {
                const bf16x8 x = *reinterpret_cast<const bf16x8*>(Xc + mm * fold + c0);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (bf16)fmaxf((float)x[k] * scale[c0 + k] + shift[c0 + k], 0.f);
            }
            *reinterpret_cast<bf16x8*>(xb + ((ps & 32) + rho32(ps & 31)) * LDX + c0) = v;
        }
        if (POS == 64) {
            put_row(my_row, v0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            mma(0, 0, 0); mma(0, 32, 32);
            __builtin_amdgcn_wave_barrier();
            put_row(my_row, v1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            mma(1, 0, 0); mma(1, 32, 32);
            __builtin_amdgcn_wave_barrier();
        } else {
            put_row(lg * 32 + my_row, v0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            mma(0, 0, 0); mma(1, 32, 0);
            __builtin_amdgcn_wave_barrier();
        }
    }
    // reduce the four waves in LDS, laid out as the parameter tensor [2][half][27], then contiguous atomics
    __syncthreads();
    float* red = reinterpret_cast<float*>(&Xb[0][0]);
    for (int i = threadIdx.x; i < fold * 27; i += 256) red[i] = 0.f;
    __syncthreads();
    STES_DET_LDS(det, _Pragma("unroll") for (int g = 0; g < 2; ++g) {
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {
            _Pragma("unroll") for (int j = 0; j < NCT; ++j) {
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {
                    const int tap = i * 16 + fq * 4 + r, c = j * 16 + fr;
                    if (tap < 27 && c < half) atomicAdd(&red[(g * half + c) * 27 + tap], acc[g][i][j][r]);
                }
            }
        }
    });
    __syncthreads();
    det_enter(det);
    for (int i = threadIdx.x; i < fold * 27; i += 256) atomicAdd(dw + i, red[i]);
    if (POS == 64) {
        bsum0 = wave_sum(bsum0); bsum1 = wave_sum(bsum1);
        STES_DET_WAVES(det, if (lane == 0) { atomicAdd(db, bsum0); atomicAdd(db + 1, bsum1); });
    } else {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) bsum0 += __shfl_xor(bsum0, o, 64);
        STES_DET_WAVES(det, if ((lane & 31) == 0) atomicAdd(db + lg, bsum0));
    }
    det_leave(det);
}

// ------------------------------------------------------------------ backward: gradient at the BatchNorm3d output
// dxbn[m,c] = [bn(x) > 0] * sum_tap w[c][tap] * dpre[m - off(tap), g]   -> bf16 [M][fold]
// workgroup = (row tile, frame, clip*2+group); dpre halo tile in LDS; thread = position, taps in registers,
// weights [c][27] wave-uniform (scalar loads)
template <int V>
__global__ __launch_bounds__(256) void gsm_dxbn_kernel(const float* __restrict__ dpre, const bf16* __restrict__ Xc,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ w3d, bf16* __restrict__ dxbn, Geom gm, int TH) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __bf16 vecb __attribute__((ext_vector_type(V)));
    float* tile = (float*)smem;                            // [3][TH+2][W+2]
    const int fold = gm.fold, half = fold >> 1;
    const int grp = blockIdx.z & 1, b = blockIdx.z >> 1, t = blockIdx.y, h0 = blockIdx.x * TH;
    const int W = gm.W, H = gm.H, W2 = W + 2, rows = TH + 2;
    const long plane = (long)H * W;
    const int items = 3 * rows * W2;
    for (int i = threadIdx.x; i < items; i += 256) {
        const int wp = i % W2; int r = i / W2;
        const int hr = r % rows, kt = r / rows;
        const int tt = t + kt - 1, hh = h0 + hr - 1, ww = wp - 1;
        float v = 0.f;
        if (tt >= 0 && tt < gm.T && hh >= 0 && hh < H && ww >= 0 && ww < W)
            v = dpre[(((long)(b * gm.T + tt)) * plane + (long)hh * W + ww) * 2 + grp];
        tile[i] = v;
    }
    __syncthreads();
    const float* wg = w3d + (long)grp * half * 27;         // [half][27]
    const float* sc = scale + grp * half;
    const float* sh = shift + grp * half;
    const int nout = TH * W;
    for (int o = threadIdx.x; o < nout; o += 256) {
        const int hr = o / W, ww = o - hr * W;
        const int hh = h0 + hr;
        if (hh >= H) continue;
        float dp[27];
        // output position q = p - (k-1) <=> tile index (1 - (kt-1), hr + 1 - (kh-1), ww + 1 - (kw-1))
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    dp[(kt * 3 + kh) * 3 + kw] = tile[((2 - kt) * rows + hr + 2 - kh) * W2 + ww + 2 - kw];
        const long m = ((long)(b * gm.T + t)) * plane + (long)hh * W + ww;
        const bf16* xrow = Xc + m * fold + grp * half;
        bf16* orow = dxbn + m * fold + grp * half;
        for (int c = 0; c < half; c += V) {
            const vecb x = *reinterpret_cast<const vecb*>(xrow + c);
            vecb o8;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float* wc = wg + (c + k) * 27;
                float a = 0.f;
#pragma unroll
                for (int tp = 0; tp < 27; ++tp) a += wc[tp] * dp[tp];
                const float xbn = (float)x[k] * sc[c + k] + sh[c + k];
                o8[k] = (bf16)(xbn > 0.f ? a : 0.f);
            }
            *reinterpret_cast<vecb*>(orow + c) = o8;
        }
    }
}

// The same for the narrow slices (fold 8 and 16, the high-resolution stages): thread = position doing BOTH groups, so that a
// position's row of `fold` channels is loaded and stored whole (16 or 32 bytes per lane, contiguous across the wave) and
// the gate gradient of both groups arrives as one 8-byte load.  workgroup = (row tile, frame, clip).
template <int FOLD>
__global__ __launch_bounds__(256) void gsm_dxbn_rows_kernel(const float* __restrict__ dpre, const bf16* __restrict__ Xc,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ w3d, bf16* __restrict__ dxbn, Geom gm, int TH) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int HALF = FOLD / 2;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2* tile = (f32x2*)smem;                            // [3][TH+2][W+2] (group 0, group 1)
    const int b = blockIdx.z, t = blockIdx.y, h0 = blockIdx.x * TH;
    const int W = gm.W, H = gm.H, W2 = W + 2, rows = TH + 2;
    const long plane = (long)H * W;
    const int items = 3 * rows * W2;
    for (int i = threadIdx.x; i < items; i += 256) {
        const int wp = i % W2; int r = i / W2;
        const int hr = r % rows, kt = r / rows;
        const int tt = t + kt - 1, hh = h0 + hr - 1, ww = wp - 1;
        f32x2 v = {0.f, 0.f};
        if (tt >= 0 && tt < gm.T && hh >= 0 && hh < H && ww >= 0 && ww < W)
            v = *reinterpret_cast<const f32x2*>(dpre + (((long)(b * gm.T + tt)) * plane + (long)hh * W + ww) * 2);
        tile[i] = v;
    }
    __syncthreads();
    const int nout = TH * W;
    for (int o = threadIdx.x; o < nout; o += 256) {
        const int hr = o / W, ww = o - hr * W;
        const int hh = h0 + hr;
        if (hh >= H) continue;
        f32x2 dp[27];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    dp[(kt * 3 + kh) * 3 + kw] = tile[((2 - kt) * rows + hr + 2 - kh) * W2 + ww + 2 - kw];
        const long m = ((long)(b * gm.T + t)) * plane + (long)hh * W + ww;
        float x[FOLD], out[FOLD];
#pragma unroll
        for (int c = 0; c < FOLD; c += 8) load8b(Xc + m * FOLD + c, *reinterpret_cast<float(*)[8]>(&x[c]));
#pragma unroll
        for (int c = 0; c < FOLD; ++c) {
            const float* wc = w3d + c * 27;                // [2][HALF][27] flat = [FOLD][27]: wave-uniform scalar loads
            float a = 0.f;
#pragma unroll
            for (int tp = 0; tp < 27; ++tp) a += wc[tp] * (c < HALF ? dp[tp][0] : dp[tp][1]);
            out[c] = (x[c] * scale[c] + shift[c] > 0.f) ? a : 0.f;
        }
#pragma unroll
        for (int c = 0; c < FOLD; c += 8) store8b(dxbn + m * FOLD + c, *reinterpret_cast<float(*)[8]>(&out[c]));
    }
}

// ------------------------------------------------------------------ backward: input gradient (elementwise)
// dXf[m,c] = dG[m,c] + gate[m,g]*(dG[m -+ plane, c] - dG[m,c]) + alpha[c]*dxbn[m,c] + beta[c]*x[m,c] + delta[c]
template <int V>
__global__ __launch_bounds__(256) void gsm_bwd_dx_kernel(const bf16* __restrict__ dG, const bf16* __restrict__ Xc,
                                                         const float* __restrict__ gate, const bf16* __restrict__ dxbn,
                                                         const float* __restrict__ alpha, const float* __restrict__ beta,
                                                         const float* __restrict__ delta, bf16* __restrict__ dXf, Geom gm) {
    typedef __bf16 vecb __attribute__((ext_vector_type(V)));
    const int fold = gm.fold, half = fold >> 1, nch = fold / V;
    const long plane = (long)gm.H * gm.W;
    const uint32_t total = (uint32_t)(gm.M * nch);
    gsm_flat_walk(total, (uint32_t)(gm.T * plane * nch), gm.B, [&](uint32_t i) {
        uint32_t mu, chk;
        fdivmod(i, gm.fd_nch, mu, chk);
        const long m = mu; const int c0 = (int)chk * V;
        const int g = c0 >= half;
        const int t = frame_of(mu, gm);
        const int tp = g == 0 ? t - 1 : t + 1;
        const bool has = tp >= 0 && tp < gm.T;
        const long mp = g == 0 ? m - plane : m + plane;
        const vecb d0 = *reinterpret_cast<const vecb*>(dG + m * fold + c0);
        const vecb x = *reinterpret_cast<const vecb*>(Xc + m * fold + c0);
        const vecb db = *reinterpret_cast<const vecb*>(dxbn + m * fold + c0);
        vecb d1;
        if (has) d1 = *reinterpret_cast<const vecb*>(dG + mp * fold + c0);
        const float gt = gate[m * 2 + g];
        vecb o;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float dr = (float)d0[k];
            const float dyt = (has ? (float)d1[k] : 0.f) - dr;
            o[k] = (bf16)(dr + dyt * gt + alpha[c0 + k] * (float)db[k] + beta[c0 + k] * (float)x[k] + delta[c0 + k]);
        }
        *reinterpret_cast<vecb*>(dXf + m * fold + c0) = o;
    });
}

// ------------------------------------------------------------------ forward in one kernel (fold >= 32): gate conv on MFMA + shift
// Workgroup = (row tile, frame t, clip).  The frame's output needs the gates of frames t (both groups), t+1 (group 0) and
// t-1 (group 1).  Per group the conv3d is a product over channels first: Z[tap(kh,kw)][pos] = sum_{kt,c} w[kt,kh,kw,c] *
// relu(bn(x[frame t'+kt-1][pos][c])), accumulated by MFMA over the three frames (A = positions x channels straight from
// global memory in operand layout, BatchNorm + ReLU in registers; B = channels x 9 taps), for the two target frames t' of
// the group at once (every fetched fragment feeds both).  Z goes to LDS tap-major with a one-pixel zero frame, and the gate
// is the 9-term spatial gather sum of it.  The shift then runs over the tile's rows in whole 16-byte chunks.
struct FwdTile {
    int TH, ZPp;                       // rows per tile; padded positions of one Z plane: (TH + 2) * (W + 2), made odd
    FastDiv fd_W, fd_nch;
};

template <int KS>   // 32-channel k-steps per group: ceil(half / 32)
__global__ __launch_bounds__(256) void gsm_fwd_fused_kernel(const bf16* __restrict__ Xc, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const bf16* __restrict__ wt,
                                                            const float* __restrict__ b3d, float* __restrict__ gate,
                                                            bf16* __restrict__ G, Geom gm, FwdTile ft) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int fold = gm.fold, half = fold >> 1, nch = fold >> 3;
    const int W = gm.W, H = gm.H, T = gm.T, ZW = W + 2, TH = ft.TH, ZPp = ft.ZPp;
    int rtile, t, b;
    if (!xcd_clip_map((H + TH - 1) / TH, T, gm.B, rtile, t, b)) return;
    const int h0 = rtile * TH;
    const int NP = TH * W, PH = (TH + 2) * W;
    float* zbuf = reinterpret_cast<float*>(smem);         // [2 targets][16 taps][ZPp]
    float* gl = zbuf + 32 * ZPp;                           // [group][target][NP]: gates of (t, g) and of the group's neighbour frame
    float* sc_s = gl + 4 * NP;
    float* sh_s = sc_s + fold;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    for (int i = tid; i < 32 * ZPp; i += 256) zbuf[i] = 0.f;
    for (int i = tid; i < fold; i += 256) { sc_s[i] = scale[i]; sh_s[i] = shift[i]; }
    __syncthreads();
    const long plane = (long)H * W;
    const bf16* clip = Xc + (long)b * T * plane * fold;
    const int ntiles = (PH + 15) >> 4;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int tb = g == 0 ? t + 1 : t - 1;             // the group's second target frame
        const bool has_b = tb >= 0 && tb < T;
        const int f0 = t - 1 - g;                          // input frames f0 .. f0 + 3
        bf16x8 wf[3][KS];                                  // B fragments: lane (fr = tap (kh,kw), fq): 8 channels of w[g][kt][tap]
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c0 = ks * 32 + 8 * fq;
                wf[kt][ks] = (fr < 9 && c0 + 8 <= half) ? *reinterpret_cast<const bf16x8*>(wt + (long)(g * 27 + kt * 9 + fr) * half + c0) : zero8();
            }
        const float bias = b3d[g];
        // (no prefetch across tiles: the registers it takes cost residency, and with few tiles per wave what hides the
        // loads' latency is the other workgroups of the CU)
        for (int i = wave; i < ntiles; i += 4) {
            const int q = i * 16 + fr;                     // position of the tile with its halo rows: row h0 - 1 + q / W
            uint32_t r, w;
            fdivmod((uint32_t)q, ft.fd_W, r, w);
            const int h = h0 - 1 + (int)r;
            const bool pv = q < PH && h >= 0 && h < H;
            const long off = pv ? ((long)h * W + (long)w) * fold + g * half : 0;
            bf16x8 raw[4][KS];
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                int f = f0 + fi;
                f = f < 0 ? 0 : (f >= T ? T - 1 : f);      // (frames outside the clip: a valid address, the value is not used)
                const bf16* base = clip + (long)f * plane * fold + off;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int c0 = ks * 32 + 8 * fq;
                    raw[fi][ks] = (pv && c0 + 8 <= half) ? *reinterpret_cast<const bf16x8*>(base + c0) : zero8();
                }
            }
            f32x4 acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                const int f = f0 + fi;
                const int kta = fi - g, ktb = fi - 1 + g;  // temporal tap of frame f for target t and for target tb
                if (f < 0 || f >= T) continue;             // zero padding in time (wave-uniform)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int c0 = ks * 32 + 8 * fq;
                    bf16x8 v = zero8();
                    if (pv && c0 + 8 <= half) {
                        float fv[8];
                        unpack8(raw[fi][ks], fv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) fv[e] = fmaxf(fv[e] * sc_s[g * half + c0 + e] + sh_s[g * half + c0 + e], 0.f);
                        v = pack8(fv);
                    }
                    if (kta >= 0 && kta < 3) acc_a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, wf[kta < 0 ? 0 : (kta > 2 ? 2 : kta)][ks], acc_a, 0, 0, 0);
                    if (ktb >= 0 && ktb < 3 && has_b) acc_b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, wf[ktb < 0 ? 0 : (ktb > 2 ? 2 : ktb)][ks], acc_b, 0, 0, 0);
                }
            }
            // D[pos = 4 fq + e][tap = fr]
            if (fr < 9) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qq = i * 16 + 4 * fq + e;
                    if (qq < PH) {
                        uint32_t rr, ww;
                        fdivmod((uint32_t)qq, ft.fd_W, rr, ww);
                        const int zp = (int)rr * ZW + (int)ww + 1;
                        zbuf[fr * ZPp + zp] = acc_a[e];
                        zbuf[(16 + fr) * ZPp + zp] = acc_b[e];
                    }
                }
            }
        }
        __syncthreads();
        for (int p = tid; p < NP; p += 256) {
            uint32_t r, w;
            fdivmod((uint32_t)p, ft.fd_W, r, w);
            const int zc = ((int)r + 1) * ZW + (int)w + 1;
            float pa = bias, pb = bias;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int o = zc + (kh - 1) * ZW + (kw - 1);
                    pa += zbuf[(kh * 3 + kw) * ZPp + o];
                    pb += zbuf[(16 + kh * 3 + kw) * ZPp + o];
                }
            gl[(g * 2) * NP + p] = tanhf(pa);
            gl[(g * 2 + 1) * NP + p] = tanhf(pb);
        }
        __syncthreads();
    }
    // shift: G[m,c] = y[t+1 | t-1] + (x - y), y = gate * x (same operation order as gsm_shift_kernel)
    const int items = NP * nch;
    for (int it0 = 0; it0 < items; it0 += 1024) {         // four items per thread, their loads issued together
        long mrow[4]; int pp[4], cc0[4]; bool ok[4];
        bf16x8 xv4[4], xn4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = it0 + u * 256 + tid;
            uint32_t p, ch, r, w;
            fdivmod((uint32_t)(it < items ? it : 0), ft.fd_nch, p, ch);
            fdivmod(p, ft.fd_W, r, w);
            const int h = h0 + (int)r;
            ok[u] = it < items && h < H;
            pp[u] = (int)p; cc0[u] = (int)ch * 8;
            const long m = ((long)b * T + t) * plane + (long)(ok[u] ? h : h0) * W + (long)w;
            mrow[u] = m;
            const int g = cc0[u] >= half;
            const int tn = g == 0 ? t + 1 : t - 1;
            const bool has = tn >= 0 && tn < T;
            xv4[u] = *reinterpret_cast<const bf16x8*>(Xc + m * fold + cc0[u]);
            xn4[u] = *reinterpret_cast<const bf16x8*>(Xc + (has ? (g == 0 ? m + plane : m - plane) : m) * fold + cc0[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;
            const int g = cc0[u] >= half, p = pp[u];
            const int tn = g == 0 ? t + 1 : t - 1;
            const bool has = tn >= 0 && tn < T;
            const float gt = gl[(g * 2) * NP + p];
            const float gn = has ? gl[(g * 2 + 1) * NP + p] : 0.f;
            bf16x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xv = (float)xv4[u][k];
                const float y = gt * xv;
                const float rr = xv - y;
                const float ys = has ? gn * (float)xn4[u][k] : 0.f;
                o[k] = (bf16)(ys + rr);
            }
            *reinterpret_cast<bf16x8*>(G + mrow[u] * fold + cc0[u]) = o;
            if (cc0[u] == 0) { gate[mrow[u] * 2] = gl[p]; gate[mrow[u] * 2 + 1] = gl[2 * NP + p]; }
        }
    }
}

// ------------------------------------------------------------------ backward of the gate conv in one kernel (fold >= 32)
// Workgroup = (row tile, frame, clip).  From the gate-gradient halo tile (3 frames, one-pixel frame, both groups) it builds
// the im2col matrix Ag[pos][tap] = dpre[pos - off(tap), g] once per group in LDS (bf16) and uses it twice on MFMA:
//   dw[g][c][tap] += sum_pos relu(bn(x[pos,c])) * Ag[pos,tap]   (reduction over positions: both operands read transposed),
//   dxb[c][pos]    = sum_tap w[g][c][tap] * Ag[pos,tap]          (masked with bn(x) > 0, rounded to bf16 = dxbn),
// and accumulates the BatchNorm3d-backward sums (sum dxbn, sum dxbn * x) of the values it writes.  The x tile sits in LDS
// with rows permuted for conflict-free transposed reads; dxbn overwrites it in place and leaves as whole rows.
// sum over the 16 lanes of a DPP row (lanes with the same lane >> 4); the total lands in the row's last lane.  Row shifts run
// in the VALU: the LDS-crossbar shuffles (ds_bpermute) cost a hundred cycles per dependent step.
__device__ __forceinline__ float row_sum_to_last(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));   // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));   // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));   // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));   // row_shr:8
    return v;
}
__device__ __forceinline__ int prow(int p) { return (p & ~31) | rho32(p & 31); }
struct BwdTile {
    int TH, NPp, SX;                   // rows per tile; positions rounded up to 32; x tile row stride (bytes)
    FastDiv fd_W, fd_nch;
};

template <int NCT>   // 16-channel tiles per group: ceil(half / 16)
__global__ __launch_bounds__(256) void gsm_bwd_conv_kernel(const float* __restrict__ dpre, const bf16* __restrict__ Xc,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ w3d, bf16* __restrict__ dxbn,
                                                           float* __restrict__ s_dy, float* __restrict__ s_dyr,
                                                           float* __restrict__ dw, float* __restrict__ db, float* __restrict__ ws,
                                                           int n_rep, Geom gm, BwdTile bt, StesDet det) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SA = 96;                                 // Ag row stride (bytes): 32 taps + 16 pad, an odd multiple of 32
    const int fold = gm.fold, half = fold >> 1, nch = fold >> 3;
    const int W = gm.W, H = gm.H, T = gm.T, ZW = W + 2, TH = bt.TH, rows = TH + 2, NPp = bt.NPp, SX = bt.SX;
    int rtile, t, b;
    if (!xcd_clip_map((H + TH - 1) / TH, T, gm.B, rtile, t, b)) { det_enter(det); det_leave(det); return; }   // (deterministic mode: the turn still passes)
    const int h0 = rtile * TH;
    const int NP = TH * W;
    unsigned char* xt = smem;                                                  // [NPp][SX] (+64 slack)
    unsigned char* ag = xt + (size_t)NPp * SX + 64;                            // [NPp][SA]
    unsigned char* wsm = ag + (size_t)NPp * SA;                                // [fold][32 taps] bf16: conv weights, zero-padded
    float* dt = reinterpret_cast<float*>(wsm + (size_t)fold * 64);             // [3][rows][ZW][2]
    float* red = dt + 3 * rows * ZW * 2;                                       // [fold][27]
    float* sums = red + fold * 27;                                             // [2][fold]
    float* dbs = sums + 2 * fold;                                              // [2]
    float* sc_s = dbs + 2;                                                     // [fold]
    float* sh_s = sc_s + fold;                                                 // [fold]
    float* slab = sh_s + fold;                                                 // [4 waves][2][fold]: BatchNorm-backward partial sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const long plane = (long)H * W;
    const long m0 = ((long)b * T + t) * plane;

    // ---- stage.  Under load a dependent global load costs microseconds, so everything the workgroup reads leaves in ONE
    // batch up front (x tile, gate-gradient halo, conv weights, BatchNorm constants); the phases below touch LDS only.
    constexpr int XB = 8, DB = 4, WB = 6;                  // loads per thread in the first batch of each kind
    const int n_x = NP * nch, n_d = 3 * rows * ZW, n_w = fold * 27 / 4;
    auto x_src = [&](int i, int& dst) -> const bf16* {    // chunk i of the tile's rows: source address, LDS offset (-1: none)
        uint32_t p, ch, r, w;
        fdivmod((uint32_t)(i < n_x ? i : 0), bt.fd_nch, p, ch);
        fdivmod(p, bt.fd_W, r, w);
        const bool ok = i < n_x && h0 + (int)r < H;
        dst = i < n_x ? prow((int)p) * SX + (int)ch * 16 : -1;
        return ok ? Xc + (m0 + (long)(h0 + (int)r) * W + (long)w) * fold + ch * 8 : nullptr;
    };
    auto d_src = [&](int i) -> const float* {             // cell i of the halo tile (nullptr: zero padding)
        const int wp = i % ZW; const int r2 = i / ZW;
        const int hr = r2 % rows, kt = r2 / rows;
        const int tt = t + kt - 1, hh = h0 + hr - 1, ww = wp - 1;
        return (i < n_d && tt >= 0 && tt < T && hh >= 0 && hh < H && ww >= 0 && ww < W)
                   ? dpre + (((long)b * T + tt) * plane + (long)hh * W + ww) * 2 : nullptr;
    };
    auto w_put = [&](int i, const f32x4& v) {             // four consecutive floats of w3d [fold][27] -> bf16 [fold][32]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = i * 4 + e, c = k / 27, tap = k - c * 27;
            reinterpret_cast<bf16*>(wsm)[c * 32 + tap] = (bf16)v[e];
        }
    };
    {
        bf16x8 xv[XB]; int xd[XB];
        float2 dv[DB];
        f32x4 wv[WB];
#pragma unroll
        for (int u = 0; u < XB; ++u) {
            const bf16* src = x_src(u * 256 + tid, xd[u]);
            xv[u] = *reinterpret_cast<const bf16x8*>(src ? src : Xc);
            if (!src) xv[u] = zero8();
        }
#pragma unroll
        for (int u = 0; u < DB; ++u) {
            const float* src = d_src(u * 256 + tid);
            dv[u] = *reinterpret_cast<const float2*>(src ? src : dpre);
            if (!src) dv[u] = make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < WB; ++u) {
            const int i = u * 256 + tid;
            wv[u] = *reinterpret_cast<const f32x4*>(w3d + 4 * (i < n_w ? i : 0));
        }
        const float scv = tid < fold ? scale[tid] : 0.f, shv = tid < fold ? shift[tid] : 0.f;
        // (LDS that the loads do not fill: zero rows past the tile, weight padding, accumulators)
        for (int i = tid; i < (NPp - NP) * (SX / 16); i += 256) {
            const int pr = i / (SX / 16), chunk = i - pr * (SX / 16);
            *reinterpret_cast<bf16x8*>(xt + (size_t)prow(NP + pr) * SX + (size_t)chunk * 16) = zero8();
        }
        for (int i = tid; i < NPp * (SA / 16); i += 256) *reinterpret_cast<bf16x8*>(ag + (size_t)i * 16) = zero8();
        for (int i = tid; i < fold * 4; i += 256) *reinterpret_cast<bf16x8*>(wsm + (size_t)i * 16) = zero8();
        for (int i = tid; i < fold * 27 + 2 * fold + 2; i += 256) red[i] = 0.f;  // red, sums and dbs are adjacent
        __syncthreads();                                   // (the weight rows are zeroed before their 27 taps arrive)
#pragma unroll
        for (int u = 0; u < XB; ++u) if (xd[u] >= 0) *reinterpret_cast<bf16x8*>(xt + xd[u]) = xv[u];
#pragma unroll
        for (int u = 0; u < DB; ++u) if (u * 256 + tid < n_d) *reinterpret_cast<float2*>(dt + 2 * (u * 256 + tid)) = dv[u];
#pragma unroll
        for (int u = 0; u < WB; ++u) if (u * 256 + tid < n_w) w_put(u * 256 + tid, wv[u]);
        if (tid < fold) { sc_s[tid] = scv; sh_s[tid] = shv; }
        // what the first batch did not cover (larger tiles or wider slices than the benchmarked ones)
        for (int i = XB * 256 + tid; i < n_x; i += 256) {
            int dst; const bf16* src = x_src(i, dst);
            *reinterpret_cast<bf16x8*>(xt + dst) = src ? *reinterpret_cast<const bf16x8*>(src) : zero8();
        }
        for (int i = DB * 256 + tid; i < n_d; i += 256) {
            const float* src = d_src(i);
            *reinterpret_cast<float2*>(dt + 2 * i) = src ? *reinterpret_cast<const float2*>(src) : make_float2(0.f, 0.f);
        }
        for (int i = WB * 256 + tid; i < n_w; i += 256) w_put(i, *reinterpret_cast<const f32x4*>(w3d + 4 * i));
    }
    __syncthreads();

    // this lane's eight taps 8 fq .. 8 fq + 7 as offsets into the halo tile relative to a position's own cell
    int toff[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int tap = 8 * fq + k;
        const int kt = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        toff[k] = tap < 27 ? ((2 - kt) * rows + (1 - kh)) * ZW + (1 - kw) : -1000000;
    }
    const int r_lo = rho32(8 * fq + (fr >> 2)), r_hi = rho32(8 * fq + (fr >> 2) + 4);
    const int cc = 4 * (fr & 3);
    const int ptiles = (NP + 15) >> 4;

#pragma unroll 1
    for (int g = 0; g < 2; ++g) {
        // ---- (a) Ag for this group: lane (fr = position, fq): taps 8 fq .. 8 fq + 7 -> one 16-byte chunk of the row
        float bsum = 0.f;
        for (int i = wave; i < ptiles; i += 4) {
            const int p = i * 16 + fr;
            uint32_t r, w;
            fdivmod((uint32_t)p, bt.fd_W, r, w);
            const bool pv = p < NP && h0 + (int)r < H;
            const int zb = ((int)r + 1) * ZW + (int)w + 1;
            float f[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = (pv && toff[k] > -1000000) ? dt[2 * (zb + toff[k]) + g] : 0.f;
            *reinterpret_cast<bf16x8*>(ag + (size_t)prow(p) * SA + fq * 16) = pack8(f);
            if (fq == 1) bsum += f[5];                      // tap 13, the centre: the position's own gate gradient (conv bias gradient)
        }
        bsum = wave_sum(bsum);
        STES_DET_LDS(det, if (lane == 0) atomicAdd(&dbs[g], bsum));
        __syncthreads();
        // ---- (b) conv weight gradient: a wave owns whole output tiles D[16 channels][16 taps] and walks every 32-position
        // step itself, so the result needs no reduction between waves (LDS float atomics cost ~2 cycles per lane)
        for (int ot = wave; ot < 2 * NCT; ot += 4) {
            const int j = ot >> 1, n2 = ot & 1;
            const int c = j * 16 + fr;
            const bool cv = c < half;                      // (columns past the group hold other data: select, do not multiply)
            const float sca = cv ? sc_s[g * half + c] : 0.f, sha = cv ? sh_s[g * half + c] : 0.f;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < (NPp >> 5); ++s) {
                const unsigned char* xs = xt + (size_t)(s * 32) * SX + (g * half + j * 16 + cc) * 2;
                const unsigned char* as = ag + (size_t)(s * 32) * SA + (n2 * 16 + cc) * 2;
                const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(as + r_lo * SA));
                const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(as + r_hi * SA));
                const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(xs + r_lo * SX));
                const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(xs + r_hi * SX));
                float f[8];
                unpack8(__builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7), f);
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = cv ? fmaxf(f[k] * sca + sha, 0.f) : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pack8(f), __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7), acc, 0, 0, 0);
            }
            // D[c = 4 fq + e][tap = fr]
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c4 = j * 16 + 4 * fq + e, tap = n2 * 16 + fr;
                if (c4 < half && tap < 27) red[(g * half + c4) * 27 + tap] = acc[e];
            }
        }
        __syncthreads();                                   // every wave is done reading this group's columns of the x tile
        // ---- (c) dxb[c][pos] = sum_tap w[c][tap] Ag[pos][tap], masked and rounded, in place of x; BatchNorm-backward sums
        {
            bf16x8 wfr[NCT];
            float scc[NCT][4], shc[NCT][4], s1[NCT][4], s2[NCT][4];
#pragma unroll
            for (int j = 0; j < NCT; ++j) {
                const int c = j * 16 + fr;
                wfr[j] = c < half ? *reinterpret_cast<const bf16x8*>(wsm + (size_t)(g * half + c) * 64 + fq * 16) : zero8();
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c4 = j * 16 + 4 * fq + e;
                    scc[j][e] = c4 < half ? sc_s[g * half + c4] : 0.f;
                    shc[j][e] = c4 < half ? sh_s[g * half + c4] : 0.f;
                    s1[j][e] = 0.f; s2[j][e] = 0.f;
                }
            }
            for (int i = wave; i < ptiles; i += 4) {
                const int p = i * 16 + fr;
                uint32_t r, w;
                fdivmod((uint32_t)p, bt.fd_W, r, w);
                const bool pv = p < NP && h0 + (int)r < H;
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(ag + (size_t)prow(p) * SA + fq * 16);
#pragma unroll
                for (int j = 0; j < NCT; ++j) {
                    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[j], fb, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    const int c4 = j * 16 + 4 * fq;
                    if (pv && c4 < half) {
                        bf16x4* cell = reinterpret_cast<bf16x4*>(xt + (size_t)prow(p) * SX + (g * half + c4) * 2);
                        const bf16x4 x4 = *cell;
                        bf16x4 o4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float xv = (float)x4[e];
                            const bf16 o = (bf16)((xv * scc[j][e] + shc[j][e] > 0.f) ? d[e] : 0.f);
                            o4[e] = o;
                            s1[j][e] += (float)o;
                            s2[j][e] += (float)o * xv;
                        }
                        *cell = o4;
                    }
                }
            }
            // the 16 lanes of a DPP row hold the same channels: sum over positions in the VALU, then this wave's partial sums
            // go to its own LDS slab (plain stores)
#pragma unroll
            for (int j = 0; j < NCT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a1 = row_sum_to_last(s1[j][e]), a2 = row_sum_to_last(s2[j][e]);
                    const int c4 = j * 16 + 4 * fq + e;
                    if (fr == 15 && c4 < half) { slab[(wave * 2) * fold + g * half + c4] = a1; slab[(wave * 2 + 1) * fold + g * half + c4] = a2; }
                }
        }
        __syncthreads();                                   // Ag is rebuilt for the next group
    }
    for (int i = tid; i < 2 * fold; i += 256) sums[i] = (slab[i] + slab[2 * fold + i]) + (slab[4 * fold + i] + slab[6 * fold + i]);
    __syncthreads();
    // ---- results: dxbn rows, statistics, weight gradient
    for (int i = tid; i < NP * nch; i += 256) {
        uint32_t p, ch, r, w;
        fdivmod((uint32_t)i, bt.fd_nch, p, ch);
        fdivmod(p, bt.fd_W, r, w);
        if (h0 + (int)r < H)
            *reinterpret_cast<bf16x8*>(dxbn + (m0 + (long)(h0 + (int)r) * W + (long)w) * fold + ch * 8) =
                *reinterpret_cast<const bf16x8*>(xt + (size_t)prow((int)p) * SX + ch * 16);
    }
    // weight gradient, statistics, bias gradient: [fold*27][fold][fold][2] as laid out in LDS.  With a workspace the adds go to
    // one of its n_rep replicas (every workgroup adding to the same few KiB serialises in the memory-side atomic units) and
    // gsm_bwd_fold_kernel adds the replicas up; without one they go to the results directly.
    const int nws = fold * 27 + 2 * fold + 2;
    det_enter(det);
    if (ws) {
        float* rep = ws + (size_t)(blockIdx.x % n_rep) * nws;
        for (int i = tid; i < nws; i += 256) atomicAdd(rep + i, red[i]);
    } else {
        for (int i = tid; i < fold * 27; i += 256) atomicAdd(dw + i, red[i]);
        for (int i = tid; i < fold; i += 256) { atomicAdd(s_dy + i, sums[i]); atomicAdd(s_dyr + i, sums[fold + i]); }
        if (tid < 2) atomicAdd(db + tid, dbs[tid]);
    }
    det_leave(det);
}

// adds the workspace replicas of gsm_bwd_conv_kernel to the results and leaves the workspace zero for the next call
__global__ __launch_bounds__(256) void gsm_bwd_fold_kernel(float* __restrict__ ws, int n_rep, int fold, float* __restrict__ s_dy,
                                                           float* __restrict__ s_dyr, float* __restrict__ dw, float* __restrict__ db) {
    const int nws = fold * 27 + 2 * fold + 2;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nws) return;
    float a = 0.f;
    for (int r0 = 0; r0 < n_rep; r0 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = r0 + k < n_rep ? ws[(size_t)(r0 + k) * nws + i] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a += v[k];
            if (r0 + k < n_rep) ws[(size_t)(r0 + k) * nws + i] = 0.f;
        }
    }
    if (i < fold * 27) dw[i] += a;
    else if (i < fold * 27 + fold) s_dy[i - fold * 27] += a;
    else if (i < fold * 27 + 2 * fold) s_dyr[i - fold * 27 - fold] += a;
    else db[i - fold * 27 - 2 * fold] += a;
}

// ------------------------------------------------------------------ column statistics / slice copy of a strided tensor
__global__ __launch_bounds__(256) void col_stats_kernel(const bf16* __restrict__ X, long ldx, int ncols, long M, float* sum, float* sumsq,
                                                        bf16* __restrict__ copy, StesDet det) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 2 * ncols; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const long gt = (long)blockIdx.x * blockDim.x + threadIdx.x, TT = (long)gridDim.x * blockDim.x;
    const int c = (int)(gt % ncols);
    float a = 0.f, b = 0.f;
    for (long m = gt / ncols; m < M; m += TT / ncols) {
        const bf16 xv = X[m * ldx + c];
        if (copy) copy[m * ncols + c] = xv;
        const float v = (float)xv; a += v; b += v * v;
    }
    if (sum) {
        STES_DET_LDS(det, atomicAdd(&lds[c], a); atomicAdd(&lds[ncols + c], b));
        __syncthreads();
        det_enter(det);
        for (int i = threadIdx.x; i < ncols; i += blockDim.x) { atomicAdd(sum + i, lds[i]); atomicAdd(sumsq + i, lds[ncols + i]); }
        det_leave(det);
    }
}

// ------------------------------------------------------------------ plain temporal shift (TSM)
// dir = +1 forward: out[t, c<f] = x[t+1], out[t, f<=c<2f] = x[t-1], rest copied; dir = -1: the adjoint
__global__ __launch_bounds__(256) void tsm_shift_kernel(const bf16* __restrict__ X, bf16* __restrict__ out, int fold, int C, int T,
                                                        long plane, long M, int dir) {
    const long total = M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C; const int c = (int)(i - m * C);
        const int t = (int)((m / plane) % T);
        bf16 v = X[i];
        if (c < 2 * fold) {
            const int step = ((c < fold) ? 1 : -1) * dir;
            const int ts = t + step;
            v = (ts >= 0 && ts < T) ? X[i + (long)step * plane * C] : (bf16)0.f;
        }
        out[i] = v;
    }
}

// The same shift in place, touching only the 2 * fold channels that move (a quarter of the tensor at fold = C / 8).  A thread
// owns four channels of one pixel of one clip and walks the clip's frames in the direction its values come from, eight frames
// per trip with all loads of a trip ahead of its stores: a value is read before the trip that overwrites it.
constexpr int TSM_TRIP = 8;
__global__ __launch_bounds__(256) void tsm_shift_inplace_kernel(bf16* X, int fold, int C, int T, long plane, long items, int chunks, int dir) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= items) return;
    const long bp = i / chunks;                       // clip * plane + pixel
    const int q = (int)(i - bp * chunks);
    const long b = bp / plane, px = bp - b * plane;
    const int step = (4 * q < fold ? 1 : -1) * dir;   // out[t] = in[t + step]
    const long fstride = plane * C;                   // elements between frames
    bf16* base = X + (b * T * plane + px) * C + 4 * q;
    // walk u = 0 .. T-1 over frames t(u) = step > 0 ? u : T-1-u, so that the source of t(u) is always t(u + 1)
    const long first = step > 0 ? 0 : (long)(T - 1) * fstride;
    const long adv = step > 0 ? fstride : -fstride;
    for (int u0 = 0; u0 < T; u0 += TSM_TRIP) {
        u32x2 v[TSM_TRIP];
#pragma unroll
        for (int k = 0; k < TSM_TRIP; ++k) {
            const int us = u0 + k + 1;                // the source position of u0 + k
            v[k] = (u32x2){0u, 0u};
            if (us < T) v[k] = *reinterpret_cast<const u32x2*>(base + first + us * adv);
        }
#pragma unroll
        for (int k = 0; k < TSM_TRIP; ++k)
            if (u0 + k < T) *reinterpret_cast<u32x2*>(base + first + (u0 + k) * adv) = v[k];
    }
}

int flat_blocks(long total, int tpb, int ncols) {
    int a = ncols, b = tpb;
    while (b) { int t = a % b; a = b; b = t; }
    const int unit = ncols / a;
    long want = (total + tpb - 1) / tpb;
    if (want > 2048) want = 2048;
    long g = (want + unit - 1) / unit * unit;
    return (int)(g < unit ? unit : g);
}

int check_geom(const stes_gsm_geom* g) {
    STES_CHECK(g && g->B > 0 && g->T > 0 && g->H > 0 && g->W > 0, STES_ERR_ARG, "gsm: bad geometry");
    STES_CHECK(g->fold == 8 || (g->fold >= 16 && g->fold % 16 == 0), STES_ERR_ARG, "gsm: fold must be 8 or a multiple of 16");
    STES_CHECK(g->fold <= 192, STES_ERR_ARG, "gsm: fold > 192 is not built");
    STES_CHECK((long)g->B * g->T * g->H * g->W * (g->fold / 4) <= 0xffffffffL, STES_ERR_UNSUPPORTED, "gsm: tensor too large");
    return STES_OK;
}

Geom to_geom(const stes_gsm_geom* g) {
    Geom gm;
    gm.B = g->B; gm.T = g->T; gm.H = g->H; gm.W = g->W; gm.fold = g->fold;
    gm.M = (long)g->B * g->T * g->H * g->W;
    gm.fd_plane = make_fastdiv(g->H * g->W); gm.fd_T = make_fastdiv(g->T); gm.fd_W = make_fastdiv(g->W); gm.fd_H = make_fastdiv(g->H);
    gm.fd_nch = make_fastdiv(g->fold == 8 ? 2 : g->fold / 8);
    return gm;
}

// rows per tile so that the halo tile stays under ~48 KiB
int tile_rows(const Geom& gm, int bytes_per_pos) {
    int th = gm.H;
    while (th > 1 && 3L * (th + 2) * (gm.W + 2) * bytes_per_pos > 48 * 1024) th = (th + 1) / 2;
    return th;
}

}  // namespace

#define S(s) ((hipStream_t)(s))

extern "C" int stes_gsm_prep(const float* w3d, uint16_t* wt, int fold, void* stream) {
    STES_CHECK(w3d && wt && fold >= 8, STES_ERR_ARG, "stes_gsm_prep: bad argument");
    const int n = fold * 27;
    hipLaunchKernelGGL(gsm_prep_kernel, dim3(stes_div_up(n, 256)), dim3(256), 0, S(stream), w3d, (bf16*)wt, fold / 2);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gsm_gate(const uint16_t* Xc, const float* scale, const float* shift, const uint16_t* wt, const float* b3d,
                             float* gate, const stes_gsm_geom* g, void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(Xc && scale && shift && wt && b3d && gate, STES_ERR_ARG, "stes_gsm_gate: null pointer");
    const Geom gm = to_geom(g);
    const int nch = gm.fold / 8;
    GateTile gt;
    gt.PS = (nch >= 4 && nch % 2 == 0) ? nch + 1 : nch;         // odd 16-byte stride: conflict-free ds_read_b128 across positions
    // one output per thread (WB = 1): measured at least as fast as four-wide strips at every stage once the channel loop
    // unrolls, and it keeps more threads busy on the small images; the WB = 4 form of the kernel is not instantiated
    const int spr = gm.W;
    // + 4 positions: the last strip of a ragged row may read (and discard) past the tile's end
    auto slot = [&](int t) { return (size_t)((t + 2) * (gm.W + 2) + 4) * gt.PS * 16; };
    auto lds_of = [&](int t) { return 3 * slot(t) + (size_t)gm.fold * 2 * sizeof(float); };
    auto fits = [&](int t) { return lds_of(t) <= 150 * 1024 && (long)(t + 2) * (gm.W + 2) * nch <= 8 * 256; };
    // rows per tile: best product of thread utilisation (128 threads per group walk TH*spr strips), row-tile quantisation,
    // halo overhead and residency -- with one or two waves per SIMD the kernel waits on LDS and scalar-load latency, so
    // the workgroups that fit a CU's 160 KiB of LDS count (up to four)
    int th = 1; double best = -1.0;
    for (int t = 1; t <= gm.H; ++t) {
        if (!fits(t)) break;
        const int strips = t * spr;
        const int resident = (int)std::min<size_t>(4, 160 * 1024 / lds_of(t));
        const double quant = gm.H / ((double)t * stes_div_up(gm.H, t));       // a ragged last tile costs a whole workgroup
        const double eff = (double)strips / (128.0 * stes_div_up(strips, 128)) * quant * quant * quant * quant * t / (t + 2.0) *
                           resident / 4.0;
        if (eff > best) { best = eff; th = t; }
    }
    STES_CHECK(fits(th), STES_ERR_UNSUPPORTED, "stes_gsm_gate: image row too wide for the LDS tile");
    const int rt = stes_div_up(gm.H, th);
    int tseg = gm.T;
    while (tseg > 4 && (long)rt * gm.B * stes_div_up(gm.T, tseg) < 2048) tseg = (tseg + 1) / 2;
    gt.TH = th; gt.TSEG = tseg;
    gt.fd_nch = make_fastdiv(nch); gt.fd_W2 = make_fastdiv(gm.W + 2); gt.fd_spr = make_fastdiv(spr);
    const size_t lds = lds_of(th);
    const dim3 grid(rt, stes_div_up(gm.T, tseg), gm.B);
    const int half = gm.fold / 2;
#define GATE_CASE(V, WB, HF)                                                                                                       \
    do {                                                                                                                           \
        static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(&gsm_gate_kernel<V, WB, HF>),                \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;     \
        STES_CHECK(big_lds || lds <= 64 * 1024, STES_ERR_UNSUPPORTED, "stes_gsm_gate: cannot raise the dynamic LDS limit");        \
        hipLaunchKernelGGL((gsm_gate_kernel<V, WB, HF>), grid, dim3(256), lds, S(stream), (const bf16*)Xc, scale, shift,           \
                           (const uint32_t*)wt, b3d, gate, gm, gt);                                                                \
    } while (0)
    // the channel counts of RegNet-Y 800MF's gated shifts get their own instantiation, anything else the generic loop
    if (gm.fold == 8) GATE_CASE(4, 1, 4);
    else if (half == 8) GATE_CASE(8, 1, 8);
    else if (half == 16) GATE_CASE(8, 1, 16);
    else if (half == 40) GATE_CASE(8, 1, 40);
    else if (half == 96) GATE_CASE(8, 1, 96);
    else GATE_CASE(8, 1, 0);
#undef GATE_CASE
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gsm_shift(const uint16_t* Xc, const float* gate, uint16_t* G, const stes_gsm_geom* g, void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(Xc && gate && G, STES_ERR_ARG, "stes_gsm_shift: null pointer");
    const Geom gm = to_geom(g);
    const int V = gm.fold == 8 ? 4 : 8;
    long blocks = (gm.M * (gm.fold / V) + 255) / 256; if (blocks > 8192) blocks = 8192;
    if (V == 4) hipLaunchKernelGGL(gsm_shift_kernel<4>, dim3((int)blocks), dim3(256), 0, S(stream), (const bf16*)Xc, gate, (bf16*)G, gm);
    else hipLaunchKernelGGL(gsm_shift_kernel<8>, dim3((int)blocks), dim3(256), 0, S(stream), (const bf16*)Xc, gate, (bf16*)G, gm);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gsm_bwd_gate(const uint16_t* dG, const uint16_t* Xc, const float* gate, float* dpre,
                                 const stes_gsm_geom* g, void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(dG && Xc && gate && dpre, STES_ERR_ARG, "stes_gsm_bwd_gate: bad argument");
    const Geom gm = to_geom(g);
    const int nch = gm.fold / 8;
    if (gm.fold >= 32 && gm.fold % 16 == 0 && nch <= 128) {
        const int rpb = 256 / nch;
        long nb = (gm.M + rpb - 1) / rpb; if (nb > 16384) nb = 16384;
        hipLaunchKernelGGL(gsm_bwd_gate_rows_kernel, dim3((int)nb), dim3(256), 0, S(stream), (const bf16*)dG, (const bf16*)Xc, gate, dpre,
                           gm, rpb, make_fastdiv(nch));
        STES_LAUNCH_CHECK();
        return STES_OK;
    }
    long blocks = (gm.M * 2 + 255) / 256; if (blocks > 8192) blocks = 8192;
    if (gm.fold == 8) hipLaunchKernelGGL(gsm_bwd_gate_kernel<4>, dim3((int)blocks), dim3(256), 0, S(stream), (const bf16*)dG, (const bf16*)Xc, gate, dpre, gm);
    else hipLaunchKernelGGL(gsm_bwd_gate_kernel<8>, dim3((int)blocks), dim3(256), 0, S(stream), (const bf16*)dG, (const bf16*)Xc, gate, dpre, gm);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gsm_bwd_wgrad(const float* dpre, const uint16_t* Xc, const float* scale, const float* shift, float* dw,
                                  float* db, const stes_gsm_geom* g, void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(dpre && Xc && scale && shift && dw && db, STES_ERR_ARG, "stes_gsm_bwd_wgrad: null pointer");
    const Geom gm = to_geom(g);
    const int half = gm.fold / 2;
    const int nct = (half + 15) / 16;
    const int pos = nct == 1 ? 64 : 32;
    const long n_chunks = (gm.M + pos - 1) / pos;
    const FastDiv fd_nch8 = make_fastdiv(gm.fold / 8);
    // >= 8 chunks per wave, and never more workgroups than are resident at once (a partial second round of a grid-stride
    // kernel runs on a part-empty GPU; the 49 KiB and 132 registers of the 64-position form allow three per CU)
#define LAUNCH_WG(N, P)                                                                                                            \
    do {                                                                                                                           \
        static const int resident = [] {                                                                                           \
            int n = 0;                                                                                                             \
            return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gsm_bwd_wgrad_kernel<N, P>, 256, 0) == hipSuccess && n > 0 ? 256 * n : 256; \
        }();                                                                                                                       \
        long gx = (n_chunks + 31) / 32;                                                                                            \
        if (gx > resident) gx = resident;                                                                                          \
        hipLaunchKernelGGL((gsm_bwd_wgrad_kernel<N, P>), dim3((int)gx), dim3(256), 0, S(stream), dpre, (const bf16*)Xc, scale, shift, dw, db, gm, n_chunks, fd_nch8, stes_det_take((unsigned)gx)); \
    } while (0)
    switch (nct) {
        case 1: LAUNCH_WG(1, 64); break;
        case 2: LAUNCH_WG(2, 32); break;
        case 3: LAUNCH_WG(3, 32); break;
        case 4: LAUNCH_WG(4, 32); break;
        case 5: LAUNCH_WG(5, 32); break;
        default: LAUNCH_WG(6, 32); break;
    }
#undef LAUNCH_WG
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gsm_dxbn(const float* dpre, const uint16_t* Xc, const float* scale, const float* shift, const float* w3d,
                             uint16_t* dxbn, const stes_gsm_geom* g, void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(dpre && Xc && scale && shift && w3d && dxbn, STES_ERR_ARG, "stes_gsm_dxbn: null pointer");
    const Geom gm = to_geom(g);
    if (gm.fold == 8 || gm.fold == 16) {
        const int th = tile_rows(gm, 8);
        const size_t lds2 = (size_t)3 * (th + 2) * (gm.W + 2) * 8;
        const dim3 grid2(stes_div_up(gm.H, th), gm.T, gm.B);
        if (gm.fold == 8)
            hipLaunchKernelGGL(gsm_dxbn_rows_kernel<8>, grid2, dim3(256), lds2, S(stream), dpre, (const bf16*)Xc, scale, shift, w3d, (bf16*)dxbn, gm, th);
        else
            hipLaunchKernelGGL(gsm_dxbn_rows_kernel<16>, grid2, dim3(256), lds2, S(stream), dpre, (const bf16*)Xc, scale, shift, w3d, (bf16*)dxbn, gm, th);
        STES_LAUNCH_CHECK();
        return STES_OK;
    }
    const int th = tile_rows(gm, 4);
    const size_t lds = (size_t)3 * (th + 2) * (gm.W + 2) * 4;
    const dim3 grid(stes_div_up(gm.H, th), gm.T, gm.B * 2);
    if (gm.fold == 8)
        hipLaunchKernelGGL(gsm_dxbn_kernel<4>, grid, dim3(256), lds, S(stream), dpre, (const bf16*)Xc, scale, shift, w3d, (bf16*)dxbn, gm, th);
    else
        hipLaunchKernelGGL(gsm_dxbn_kernel<8>, grid, dim3(256), lds, S(stream), dpre, (const bf16*)Xc, scale, shift, w3d, (bf16*)dxbn, gm, th);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gsm_bwd_dx(const uint16_t* dG, const uint16_t* Xc, const float* gate, const uint16_t* dxbn,
                               const float* alpha, const float* beta, const float* delta, uint16_t* dXf, const stes_gsm_geom* g,
                               void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(dG && Xc && gate && dxbn && alpha && beta && delta && dXf, STES_ERR_ARG, "stes_gsm_bwd_dx: null pointer");
    const Geom gm = to_geom(g);
    const int V = gm.fold == 8 ? 4 : 8;
    long blocks = (gm.M * (gm.fold / V) + 255) / 256; if (blocks > 8192) blocks = 8192;
    if (V == 4) hipLaunchKernelGGL(gsm_bwd_dx_kernel<4>, dim3((int)blocks), dim3(256), 0, S(stream), (const bf16*)dG, (const bf16*)Xc, gate,
                                   (const bf16*)dxbn, alpha, beta, delta, (bf16*)dXf, gm);
    else hipLaunchKernelGGL(gsm_bwd_dx_kernel<8>, dim3((int)blocks), dim3(256), 0, S(stream), (const bf16*)dG, (const bf16*)Xc, gate,
                            (const bf16*)dxbn, alpha, beta, delta, (bf16*)dXf, gm);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_col_stats(const uint16_t* X, int64_t ldx, int ncols, int64_t M, float* sum, float* sumsq, uint16_t* copy,
                              void* stream) {
    STES_CHECK(X && ncols > 0 && ncols <= ldx && M > 0 && (sum == nullptr) == (sumsq == nullptr) && (sum || copy), STES_ERR_ARG, "stes_col_stats: bad argument");
    const int blocks = flat_blocks(M * ncols, 256, ncols);
    hipLaunchKernelGGL(col_stats_kernel, dim3(blocks), dim3(256), (size_t)(2 * ncols) * sizeof(float), S(stream), (const bf16*)X, (long)ldx,
                       ncols, (long)M, sum, sumsq, (bf16*)copy, sum ? stes_det_take(blocks) : StesDet{nullptr, 0u, 0u});
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_tsm_shift(const uint16_t* X, uint16_t* out, int fold, int C, int B, int T, int H, int W, int backward,
                              void* stream) {
    STES_CHECK(X && out && X != out && fold >= 0 && 2 * fold <= C && B > 0 && T > 0, STES_ERR_ARG, "stes_tsm_shift: bad argument (out must not alias X)");
    const long plane = (long)H * W, M = (long)B * T * plane;
    long blocks = (M * C + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(tsm_shift_kernel, dim3((int)blocks), dim3(256), 0, S(stream), (const bf16*)X, (bf16*)out, fold, C, T, plane, M,
                       backward ? -1 : 1);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_tsm_shift_inplace(uint16_t* X, int fold, int C, int B, int T, int H, int W, int backward, void* stream) {
    STES_CHECK(X && fold >= 0 && 2 * fold <= C && B > 0 && T > 0 && H > 0 && W > 0, STES_ERR_ARG, "stes_tsm_shift_inplace: bad argument");
    STES_CHECK(fold % 4 == 0 && C % 4 == 0, STES_ERR_UNSUPPORTED, "stes_tsm_shift_inplace: fold and C must be multiples of 4 (8-byte pieces)");
    if (fold == 0) return STES_OK;
    const long plane = (long)H * W;
    const int chunks = 2 * fold / 4;
    const long items = (long)B * plane * chunks;
    STES_CHECK((items + 255) / 256 <= 0x7fffffffL, STES_ERR_UNSUPPORTED, "stes_tsm_shift_inplace: tensor too large");
    hipLaunchKernelGGL(tsm_shift_inplace_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, S(stream), (bf16*)X, fold, C, T, plane,
                       items, chunks, backward ? -1 : 1);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

namespace {
// rows per tile: about 256 positions, tiles of equal height
int fused_tile_rows(const Geom& gm) {
    const int thmax = gm.W >= 256 ? 1 : 256 / gm.W;
    const int nt = stes_div_up(gm.H, thmax);
    return stes_div_up(gm.H, nt);
}
}  // namespace

extern "C" int stes_gsm_forward(const uint16_t* Xc, const float* scale, const float* shift, const uint16_t* wt, const float* b3d,
                                float* gate, uint16_t* G, const stes_gsm_geom* g, void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(Xc && scale && shift && wt && b3d && gate && G, STES_ERR_ARG, "stes_gsm_forward: null pointer");
    // narrow slices (the high-resolution stages): their own kernel, taps x channels as the contraction dimension
    if (g->fold < 32) return stes_gsm_small_forward(Xc, scale, shift, wt, b3d, gate, G, g->B, g->T, g->H, g->W, g->fold, stream);
    const Geom gm = to_geom(g);
    FwdTile ft;
    ft.TH = fused_tile_rows(gm);
    ft.ZPp = ((ft.TH + 2) * (gm.W + 2)) | 1;
    ft.fd_W = make_fastdiv(gm.W); ft.fd_nch = make_fastdiv(gm.fold / 8);
    const size_t lds = ((size_t)32 * ft.ZPp + 4 * ft.TH * gm.W + 2 * gm.fold) * sizeof(float);
    STES_CHECK(lds <= 64 * 1024, STES_ERR_UNSUPPORTED, "stes_gsm_forward: image row too wide for the LDS tile");
    const dim3 grid(8 * stes_div_up(gm.B, 8) * gm.T * stes_div_up(gm.H, ft.TH));       // (see xcd_clip_map)
    const int ks = (gm.fold / 2 + 31) / 32;
#define FWD_CASE(K) hipLaunchKernelGGL((gsm_fwd_fused_kernel<K>), grid, dim3(256), lds, S(stream), (const bf16*)Xc, scale, shift, (const bf16*)wt, b3d, gate, (bf16*)G, gm, ft)
    if (ks == 1) FWD_CASE(1); else if (ks == 2) FWD_CASE(2); else FWD_CASE(3);
#undef FWD_CASE
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gsm_bwd_conv_ws_floats(int fold) { return STES_GSM_WS_REPLICAS * (fold * 27 + 2 * fold + 2); }

extern "C" int stes_gsm_bwd_conv(const float* dpre, const uint16_t* Xc, const float* scale, const float* shift, const float* w3d,
                                 uint16_t* dxbn, float* s_dy, float* s_dyr, float* dw, float* db, float* ws, const stes_gsm_geom* g,
                                 void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(dpre && Xc && scale && shift && w3d && dxbn && s_dy && s_dyr, STES_ERR_ARG, "stes_gsm_bwd_conv: null pointer");
    STES_CHECK((dw == nullptr) == (db == nullptr), STES_ERR_ARG, "stes_gsm_bwd_conv: dw and db come together");
    if (g->fold < 32) {
        // narrow slices: dxbn and its sums in one kernel; the conv3D weight gradient (when asked for) is its own kernel behind it
        rc = stes_gsm_small_dxbn(dpre, Xc, scale, shift, w3d, dxbn, s_dy, s_dyr, g->B, g->T, g->H, g->W, g->fold, stream);
        if (rc == STES_OK && dw) rc = stes_gsm_bwd_wgrad(dpre, Xc, scale, shift, dw, db, g, stream);
        return rc;
    }
    STES_CHECK(dw && db, STES_ERR_ARG, "stes_gsm_bwd_conv: the wide-slice kernel always computes the conv3D gradients");
    const Geom gm = to_geom(g);
    BwdTile bt;
    bt.TH = fused_tile_rows(gm);
    const int np = bt.TH * gm.W;
    bt.NPp = (np + 31) / 32 * 32;
    bt.SX = ((gm.fold * 2 + 31) / 32 | 1) * 32;
    bt.fd_W = make_fastdiv(gm.W); bt.fd_nch = make_fastdiv(gm.fold / 8);
    const size_t lds = (size_t)bt.NPp * bt.SX + 64 + (size_t)bt.NPp * 96 + (size_t)gm.fold * 64 +
                       ((size_t)3 * (bt.TH + 2) * (gm.W + 2) * 2 + gm.fold * 27 + 12 * gm.fold + 2) * sizeof(float);
    STES_CHECK(lds <= 160 * 1024, STES_ERR_UNSUPPORTED, "stes_gsm_bwd_conv: tile does not fit LDS");
    const dim3 grid(8 * stes_div_up(gm.B, 8) * gm.T * stes_div_up(gm.H, bt.TH));      // (see xcd_clip_map)
    const int nct = (gm.fold / 2 + 15) / 16;
#define BWD_CASE(N)                                                                                                                \
    do {                                                                                                                           \
        static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(&gsm_bwd_conv_kernel<N>),                    \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;     \
        STES_CHECK(big_lds || lds <= 64 * 1024, STES_ERR_UNSUPPORTED, "stes_gsm_bwd_conv: cannot raise the dynamic LDS limit");    \
        hipLaunchKernelGGL((gsm_bwd_conv_kernel<N>), grid, dim3(256), lds, S(stream), dpre, (const bf16*)Xc, scale, shift, w3d,   \
                           (bf16*)dxbn, s_dy, s_dyr, dw, db, ws, STES_GSM_WS_REPLICAS, gm, bt, stes_det_take(grid.x * grid.y * grid.z)); \
    } while (0)
    switch (nct) {
        case 1: BWD_CASE(1); break;
        case 2: BWD_CASE(2); break;
        case 3: BWD_CASE(3); break;
        case 4: BWD_CASE(4); break;
        case 5: BWD_CASE(5); break;
        default: BWD_CASE(6); break;
    }
#undef BWD_CASE
    STES_LAUNCH_CHECK();
    if (ws) {
        const int nws = gm.fold * 27 + 2 * gm.fold + 2;
        hipLaunchKernelGGL(gsm_bwd_fold_kernel, dim3(stes_div_up(nws, 256)), dim3(256), 0, S(stream), ws, STES_GSM_WS_REPLICAS, gm.fold,
                           s_dy, s_dyr, dw, db);
        STES_LAUNCH_CHECK();
    }
    return STES_OK;
}
