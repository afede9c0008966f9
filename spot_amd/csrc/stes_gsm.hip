// Gated Shift Module (reference model/shift.py:83-161) and plain temporal shift (:21-78) on
// NHWC bf16 activations.  Rows are m = ((clip*T + t)*H + h)*W + w; the module acts on the first
// `fold` channels of each row and never crosses a clip boundary.
//
//   gate  : g[m, grp] = tanh(conv3d_{3x3x3, groups=2}(relu(bn3d(x)))[m, grp])          (fp32)
//   shift : out[m, dest(c)] = y[t+1 | t-1] + (x - y),  y = g*x  (group 0 pulls t+1, group 1 t-1,
//           zero at the clip edge), dest = within-group interleave of the reference's view/permute
//   backward: gate-gradient, conv3d weight/bias gradient, BatchNorm3d reductions, input gradient.
// The fold slice is a small fraction of the tensor (<= 1/4 of the channels); all kernels are
// gather-style and lean on L2 for the 27-point neighbourhood.
#include "stes_common.h"

namespace {

struct Geom {
    int B, T, H, W, ldx, fold;
    long M;
};

__device__ __forceinline__ int dest_of(int c, int fold) {
    const int half = fold >> 1, quarter = fold >> 2;
    const int g = c / half, rem = c - g * half;
    const int i = rem / quarter, j = rem - i * quarter;
    return g * half + 2 * j + i;
}

struct Pos { int b, t, h, w; };
__device__ __forceinline__ Pos decode(long m, const Geom& gm) {
    Pos p;
    p.w = (int)(m % gm.W); long r = m / gm.W;
    p.h = (int)(r % gm.H); r /= gm.H;
    p.t = (int)(r % gm.T); p.b = (int)(r / gm.T);
    return p;
}

// ------------------------------------------------------------------ gate
// one thread per (row, group); weights [2][half][27] fp32 in LDS
__global__ __launch_bounds__(256) void gsm_gate_kernel(const bf16* __restrict__ X, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ w3d,
                                                       const float* __restrict__ b3d, float* __restrict__ gate, Geom gm) {
    extern __shared__ float lds[];     // w[half*27] + scale[half] + shift[half]
    const int half = gm.fold >> 1, grp = blockIdx.y;
    float* sw = lds; float* ss = lds + half * 27; float* sb = ss + half;
    for (int i = threadIdx.x; i < half * 27; i += blockDim.x) sw[i] = w3d[(long)grp * half * 27 + i];
    for (int i = threadIdx.x; i < half; i += blockDim.x) { ss[i] = scale[grp * half + i]; sb[i] = shift[grp * half + i]; }
    __syncthreads();
    const float bias = b3d[grp];
    const long plane = (long)gm.H * gm.W;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < gm.M; m += (long)gridDim.x * blockDim.x) {
        const Pos p = decode(m, gm);
        float acc = bias;
        for (int kt = 0; kt < 3; ++kt) {
            const int t = p.t + kt - 1;
            if (t < 0 || t >= gm.T) continue;
            for (int kh = 0; kh < 3; ++kh) {
                const int h = p.h + kh - 1;
                if (h < 0 || h >= gm.H) continue;
                for (int kw = 0; kw < 3; ++kw) {
                    const int w = p.w + kw - 1;
                    if (w < 0 || w >= gm.W) continue;
                    const bf16* xp = X + (m + (long)(kt - 1) * plane + (long)(kh - 1) * gm.W + (kw - 1)) * gm.ldx + grp * half;
                    const int tap = (kt * 3 + kh) * 3 + kw;
                    for (int c = 0; c < half; c += 4) {
                        const bf16x4 v = *reinterpret_cast<const bf16x4*>(xp + c);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc += sw[(c + i) * 27 + tap] * fmaxf((float)v[i] * ss[c + i] + sb[c + i], 0.f);
                    }
                }
            }
        }
        gate[m * 2 + grp] = tanhf(acc);
    }
}

// ------------------------------------------------------------------ shift (forward)
// thread per (row, destination channel)
__global__ __launch_bounds__(256) void gsm_shift_kernel(const bf16* __restrict__ X, const float* __restrict__ gate,
                                                        bf16* __restrict__ G, Geom gm) {
    const int fold = gm.fold, half = fold >> 1, quarter = fold >> 2;
    const long plane = (long)gm.H * gm.W;
    const long total = gm.M * fold;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / fold; const int d = (int)(i - m * fold);
        const int g = d / half, rem = d - g * half;
        const int c = g * half + (rem & 1) * quarter + (rem >> 1);      // source channel of destination d
        const int t = (int)((m / plane) % gm.T);
        const float x = (float)X[m * gm.ldx + c];
        const float y = gate[m * 2 + g] * x;
        const float r = x - y;
        float ys = 0.f;
        const int tn = g == 0 ? t + 1 : t - 1;
        if (tn >= 0 && tn < gm.T) {
            const long mn = g == 0 ? m + plane : m - plane;
            ys = gate[mn * 2 + g] * (float)X[mn * gm.ldx + c];
        }
        G[m * fold + d] = (bf16)(ys + r);
    }
}

// ------------------------------------------------------------------ backward: gate pre-activation gradient
// dpre[m,g] = (1 - gate^2) * sum_{c in g} x[m,c] * (dG[m -+ plane, dest(c)] - dG[m, dest(c)])
__global__ __launch_bounds__(256) void gsm_bwd_gate_kernel(const bf16* __restrict__ dG, long ldg, const bf16* __restrict__ X,
                                                           const float* __restrict__ gate, float* __restrict__ dpre, Geom gm) {
    const int fold = gm.fold, half = fold >> 1;
    const long plane = (long)gm.H * gm.W;
    const long total = gm.M * 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i >> 1; const int g = (int)(i & 1);
        const int t = (int)((m / plane) % gm.T);
        const int tp = g == 0 ? t - 1 : t + 1;                 // frame whose output used y[t]
        const bool has = tp >= 0 && tp < gm.T;
        const long mp = g == 0 ? m - plane : m + plane;
        float acc = 0.f;
        for (int cc = 0; cc < half; ++cc) {
            const int c = g * half + cc, d = dest_of(c, fold);
            float dy = -(float)dG[m * ldg + d];
            if (has) dy += (float)dG[mp * ldg + d];
            acc += dy * (float)X[m * gm.ldx + c];
        }
        const float gt = gate[i];
        dpre[i] = acc * (1.f - gt * gt);
    }
}

// ------------------------------------------------------------------ backward: conv3d weight / bias gradient
// outputs o = tap*fold + c; each block reduces a chunk of rows
__global__ __launch_bounds__(256) void gsm_bwd_wgrad_kernel(const float* __restrict__ dpre, const bf16* __restrict__ X,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ dw, float* __restrict__ db, Geom gm, int rows_per_block) {
    const int fold = gm.fold, half = fold >> 1;
    const long plane = (long)gm.H * gm.W;
    const long m0 = (long)blockIdx.x * rows_per_block;
    const long m1 = m0 + rows_per_block < gm.M ? m0 + rows_per_block : gm.M;
    for (int o = threadIdx.x; o < 27 * fold; o += blockDim.x) {
        const int tap = o / fold, c = o - tap * fold;
        const int g = c / half;
        const int kt = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        const float s = scale[c], b = shift[c];
        float acc = 0.f;
        for (long m = m0; m < m1; ++m) {
            const Pos p = decode(m, gm);
            const int t = p.t + kt - 1, h = p.h + kh - 1, w = p.w + kw - 1;
            if (t < 0 || t >= gm.T || h < 0 || h >= gm.H || w < 0 || w >= gm.W) continue;
            const long ms = m + (long)(kt - 1) * plane + (long)(kh - 1) * gm.W + (kw - 1);
            acc += dpre[m * 2 + g] * fmaxf((float)X[ms * gm.ldx + c] * s + b, 0.f);
        }
        atomicAdd(dw + (long)c * 27 + tap, acc);
    }
    if (threadIdx.x < 2) {
        float acc = 0.f;
        for (long m = m0; m < m1; ++m) acc += dpre[m * 2 + threadIdx.x];
        atomicAdd(db + threadIdx.x, acc);
    }
}

// gradient reaching the BatchNorm3d output (after the ReLU mask) at (m, c)
__device__ __forceinline__ float dxbn_at(long m, int c, int g, const Pos& p, const float* __restrict__ dpre, const float* sw,
                                         float xbn, const Geom& gm, long plane) {
    if (!(xbn > 0.f)) return 0.f;
    float acc = 0.f;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
        const int t = p.t - (kt - 1);
        if (t < 0 || t >= gm.T) continue;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int h = p.h - (kh - 1);
            if (h < 0 || h >= gm.H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int w = p.w - (kw - 1);
                if (w < 0 || w >= gm.W) continue;
                const long mo = m - (long)(kt - 1) * plane - (long)(kh - 1) * gm.W - (kw - 1);
                acc += sw[c * 27 + (kt * 3 + kh) * 3 + kw] * dpre[mo * 2 + g];
            }
        }
    }
    return acc;
}

// s_dy[c] += sum dxbn, s_dyx[c] += sum dxbn * x       (thread per (row, channel), channel fixed per thread)
__global__ __launch_bounds__(256) void gsm_bwd_bnreduce_kernel(const float* __restrict__ dpre, const bf16* __restrict__ X,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ w3d, float* s_dy, float* s_dyx, Geom gm) {
    extern __shared__ float lds[];     // w[fold*27] + acc[2*fold]
    const int fold = gm.fold, half = fold >> 1;
    float* sw = lds; float* sa = lds + fold * 27;
    for (int i = threadIdx.x; i < fold * 27; i += blockDim.x) sw[i] = w3d[i];
    for (int i = threadIdx.x; i < 2 * fold; i += blockDim.x) sa[i] = 0.f;
    __syncthreads();
    const long plane = (long)gm.H * gm.W;
    const long gt = (long)blockIdx.x * blockDim.x + threadIdx.x, TT = (long)gridDim.x * blockDim.x;
    const int c = (int)(gt % fold), g = c / half;
    const float s = scale[c], b = shift[c];
    float a0 = 0.f, a1 = 0.f;
    for (long m = gt / fold; m < gm.M; m += TT / fold) {
        const Pos p = decode(m, gm);
        const float x = (float)X[m * gm.ldx + c];
        const float d = dxbn_at(m, c, g, p, dpre, sw, x * s + b, gm, plane);
        a0 += d; a1 += d * x;
    }
    atomicAdd(&sa[c], a0); atomicAdd(&sa[fold + c], a1);
    __syncthreads();
    for (int i = threadIdx.x; i < fold; i += blockDim.x) { atomicAdd(s_dy + i, sa[i]); atomicAdd(s_dyx + i, sa[fold + i]); }
}

// dXf[m,c] = dG[m,dest] + gate[m,g]*(dG[m -+ plane, dest] - dG[m, dest]) + alpha*dxbn + beta*x + delta
__global__ __launch_bounds__(256) void gsm_bwd_dx_kernel(const bf16* __restrict__ dG, long ldg, const bf16* __restrict__ X,
                                                         const float* __restrict__ gate, const float* __restrict__ dpre,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ w3d, const float* __restrict__ alpha,
                                                         const float* __restrict__ beta, const float* __restrict__ delta,
                                                         bf16* __restrict__ dXf, Geom gm) {
    extern __shared__ float lds[];
    const int fold = gm.fold, half = fold >> 1;
    float* sw = lds;
    for (int i = threadIdx.x; i < fold * 27; i += blockDim.x) sw[i] = w3d[i];
    __syncthreads();
    const long plane = (long)gm.H * gm.W;
    const long total = gm.M * fold;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / fold; const int c = (int)(i - m * fold);
        const int g = c / half, d = dest_of(c, fold);
        const Pos p = decode(m, gm);
        const float x = (float)X[m * gm.ldx + c];
        const float dr = (float)dG[m * ldg + d];
        const int tp = g == 0 ? p.t - 1 : p.t + 1;
        float dyt = -dr;
        if (tp >= 0 && tp < gm.T) dyt += (float)dG[(g == 0 ? m - plane : m + plane) * ldg + d];
        float v = dr + dyt * gate[m * 2 + g];
        const float dbn = dxbn_at(m, c, g, p, dpre, sw, x * scale[c] + shift[c], gm, plane);
        v += alpha[c] * dbn + beta[c] * x + delta[c];
        dXf[m * fold + c] = (bf16)v;
    }
}

// ------------------------------------------------------------------ column statistics of a slice
__global__ __launch_bounds__(256) void col_stats_kernel(const bf16* __restrict__ X, long ldx, int ncols, long M, float* sum, float* sumsq) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 2 * ncols; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const long gt = (long)blockIdx.x * blockDim.x + threadIdx.x, TT = (long)gridDim.x * blockDim.x;
    const int c = (int)(gt % ncols);
    float a = 0.f, b = 0.f;
    for (long m = gt / ncols; m < M; m += TT / ncols) { const float v = (float)X[m * ldx + c]; a += v; b += v * v; }
    atomicAdd(&lds[c], a); atomicAdd(&lds[ncols + c], b);
    __syncthreads();
    for (int i = threadIdx.x; i < ncols; i += blockDim.x) { atomicAdd(sum + i, lds[i]); atomicAdd(sumsq + i, lds[ncols + i]); }
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

int flat_blocks(long total, int tpb, int ncols) {
    // grid*tpb must be a multiple of ncols
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
    STES_CHECK(g->fold >= 4 && g->fold % 4 == 0 && g->fold <= g->ldx, STES_ERR_ARG, "gsm: fold must be a positive multiple of 4, at most the row stride");
    return STES_OK;
}

Geom to_geom(const stes_gsm_geom* g) {
    Geom gm;
    gm.B = g->B; gm.T = g->T; gm.H = g->H; gm.W = g->W; gm.ldx = g->ldx; gm.fold = g->fold;
    gm.M = (long)g->B * g->T * g->H * g->W;
    return gm;
}

}  // namespace

#define S(s) ((hipStream_t)(s))

extern "C" int stes_gsm_gate(const uint16_t* X, const float* scale, const float* shift, const float* w3d, const float* b3d,
                             float* gate, const stes_gsm_geom* g, void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(X && scale && shift && w3d && b3d && gate, STES_ERR_ARG, "stes_gsm_gate: null pointer");
    STES_CHECK(g->fold % 8 == 0 && g->ldx % 4 == 0, STES_ERR_ARG, "stes_gsm_gate: fold % 8, ldx % 4");
    const Geom gm = to_geom(g);
    const int half = gm.fold / 2;
    long blocks = (gm.M + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gsm_gate_kernel, dim3((int)blocks, 2), dim3(256), (size_t)(half * 29) * sizeof(float), S(stream), (const bf16*)X,
                       scale, shift, w3d, b3d, gate, gm);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gsm_shift(const uint16_t* X, const float* gate, uint16_t* G, const stes_gsm_geom* g, void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(X && gate && G, STES_ERR_ARG, "stes_gsm_shift: null pointer");
    const Geom gm = to_geom(g);
    long blocks = (gm.M * gm.fold + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gsm_shift_kernel, dim3((int)blocks), dim3(256), 0, S(stream), (const bf16*)X, gate, (bf16*)G, gm);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gsm_bwd_gate(const uint16_t* dG, int64_t ldg, const uint16_t* X, const float* gate, float* dpre,
                                 const stes_gsm_geom* g, void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(dG && X && gate && dpre && ldg >= g->fold, STES_ERR_ARG, "stes_gsm_bwd_gate: bad argument");
    const Geom gm = to_geom(g);
    long blocks = (gm.M * 2 + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gsm_bwd_gate_kernel, dim3((int)blocks), dim3(256), 0, S(stream), (const bf16*)dG, (long)ldg, (const bf16*)X, gate, dpre, gm);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gsm_bwd_wgrad(const float* dpre, const uint16_t* X, const float* scale, const float* shift, float* dw,
                                  float* db, const stes_gsm_geom* g, void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(dpre && X && scale && shift && dw && db, STES_ERR_ARG, "stes_gsm_bwd_wgrad: null pointer");
    const Geom gm = to_geom(g);
    int rows = (int)((gm.M + 1023) / 1024);
    if (rows < 64) rows = 64;
    const int blocks = stes_div_up(gm.M, rows);
    hipLaunchKernelGGL(gsm_bwd_wgrad_kernel, dim3(blocks), dim3(256), 0, S(stream), dpre, (const bf16*)X, scale, shift, dw, db, gm, rows);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gsm_bwd_bnreduce(const float* dpre, const uint16_t* X, const float* scale, const float* shift, const float* w3d,
                                     float* s_dy, float* s_dyx, const stes_gsm_geom* g, void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(dpre && X && scale && shift && w3d && s_dy && s_dyx, STES_ERR_ARG, "stes_gsm_bwd_bnreduce: null pointer");
    const Geom gm = to_geom(g);
    const int blocks = flat_blocks(gm.M * gm.fold, 256, gm.fold);
    hipLaunchKernelGGL(gsm_bwd_bnreduce_kernel, dim3(blocks), dim3(256), (size_t)(gm.fold * 29) * sizeof(float), S(stream), dpre, (const bf16*)X,
                       scale, shift, w3d, s_dy, s_dyx, gm);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gsm_bwd_dx(const uint16_t* dG, int64_t ldg, const uint16_t* X, const float* gate, const float* dpre,
                               const float* scale, const float* shift, const float* w3d, const float* alpha, const float* beta,
                               const float* delta, uint16_t* dXf, const stes_gsm_geom* g, void* stream) {
    int rc = check_geom(g); if (rc) return rc;
    STES_CHECK(dG && X && gate && dpre && scale && shift && w3d && alpha && beta && delta && dXf, STES_ERR_ARG, "stes_gsm_bwd_dx: null pointer");
    const Geom gm = to_geom(g);
    long blocks = (gm.M * gm.fold + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gsm_bwd_dx_kernel, dim3((int)blocks), dim3(256), (size_t)(gm.fold * 27) * sizeof(float), S(stream), (const bf16*)dG,
                       (long)ldg, (const bf16*)X, gate, dpre, scale, shift, w3d, alpha, beta, delta, (bf16*)dXf, gm);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_col_stats(const uint16_t* X, int64_t ldx, int ncols, int64_t M, float* sum, float* sumsq, void* stream) {
    STES_CHECK(X && sum && sumsq && ncols > 0 && ncols <= ldx && M > 0, STES_ERR_ARG, "stes_col_stats: bad argument");
    const int blocks = flat_blocks(M * ncols, 256, ncols);
    hipLaunchKernelGGL(col_stats_kernel, dim3(blocks), dim3(256), (size_t)(2 * ncols) * sizeof(float), S(stream), (const bf16*)X, (long)ldx,
                       ncols, (long)M, sum, sumsq);
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
