// HBM-bound channel-vector kernels of the RegNet-Y bottleneck: BatchNorm finalisation,
// BN-apply + residual + ReLU, squeeze-excite pooling / FC, BatchNorm backward reductions and
// applications, the block-input gradient merge, global average pooling.
//
// Layout: NHWC bf16 viewed as [rows][C]; every thread owns one 8-channel (16-byte) chunk and walks
// rows with a stride that keeps its chunk fixed, so per-channel reductions stay in registers until
// one LDS + one global atomic per channel per workgroup.
#include "stes_common.h"

namespace {

constexpr int TPB = 256;

__host__ int flat_grid(long rows, int ch, int max_blocks = 2048) {
    // total threads must be a multiple of ch: grid a multiple of ch / gcd(ch, TPB)
    int a = ch, b = TPB;
    while (b) { int t = a % b; a = b; b = t; }
    const int unit = ch / a;
    long want = (rows * ch + TPB - 1) / TPB;
    long g = (want + unit - 1) / unit * unit;
    if (g > max_blocks) g = max_blocks / unit * unit;       // never more than the cap: see resident_blocks
    if (g < unit) g = unit;
    return (int)g;
}

// Workgroups of `kernel` that are resident at once on the whole GPU.  The grid-stride kernels below run with at most this
// many: the workgroups of a second, partial round would each walk their share of the rows alone, at one memory round trip
// per row (measured on the 3x3 weight gradient: 364 us at exactly one resident set, 494 us with 8 % more workgroups).
template <class K>
__host__ int resident_blocks(K kernel, size_t lds = 0) {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, TPB, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return per_cu * (cus > 0 ? cus : 256);
}


// ---------------------------------------------------------------- BN finalize
__global__ void bn_finalize_kernel(const float* sum, const float* sumsq, double count, const float* gamma,
                                   const float* beta, float* rmean, float* rvar, float momentum, float eps,
                                   int training, float* scale, float* shift, float* smean, float* sinv, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean, var;
    if (training) {
        const double m = (double)sum[c] / count;
        double v = (double)sumsq[c] / count - m * m;
        if (v < 0) v = 0;
        mean = (float)m; var = (float)v;
        if (rmean) {
            const double unb = count > 1 ? v * count / (count - 1) : v;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
        }
    } else {
        mean = rmean[c]; var = rvar[c];
    }
    const float inv = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * inv;
    shift[c] = b - mean * g * inv;
    if (smean) { smean[c] = mean; sinv[c] = inv; }
}

// ---------------------------------------------------------------- out = relu(bn(c) + residual)
__global__ __launch_bounds__(TPB) void bn_act_residual_kernel(
    const bf16* __restrict__ c, const float* __restrict__ scale, const float* __restrict__ shift,
    const bf16* __restrict__ res, const float* __restrict__ rscale, const float* __restrict__ rshift,
    bf16* __restrict__ out, float* stat_sum, float* stat_sumsq, int stat_c, bf16* __restrict__ slice_out, long M, int C,
    uint8_t* __restrict__ relu_bits, StesDet det, int tsm_fold, FastDiv fd_hw, FastDiv fd_T) {
    extern __shared__ float lds[];     // [2][stat_c]
    const int CH = C / 8;
    const long gt = (long)blockIdx.x * TPB + threadIdx.x;
    const long TT = (long)gridDim.x * TPB;
    const int chunk = (int)(gt % CH);
    const long rstep = TT / CH;
    const int c0 = chunk * 8;
    float s[8], b[8], rs[8], rb[8];
    load8f(scale + c0, s); load8f(shift + c0, b);
    if (rscale) { load8f(rscale + c0, rs); load8f(rshift + c0, rb); }
    float ssum[8], ssq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ssum[i] = 0.f; ssq[i] = 0.f; }
    for (int i = threadIdx.x; i < 2 * stat_c; i += TPB) lds[i] = 0.f;
    for (long m = gt / CH; m < M; m += rstep) {
        float v[8], r[8];
        load8b(c + m * C + c0, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * s[i] + b[i];
        if (res) {
            load8b(res + m * C + c0, r);
            if (rscale) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += r[i] * rs[i] + rb[i];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += r[i];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) { o[i] = (bf16)fmaxf(v[i], 0.f); v[i] = (float)o[i]; }
        if (c0 < 2 * tsm_fold) {
            // the next block's plain temporal shift (reference model/shift.py:37-52) taken in the store: channels [0, fold) of
            // frame t belong in frame t-1, [fold, 2 fold) in frame t+1; the value that would leave the clip is dropped and its
            // writer puts the zero into the frame at the other end that gets no value (four channels at a time)
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 raw = __builtin_bit_cast(u32x4, o);
            uint32_t nimg, rem, clip, t;
            fdivmod((uint32_t)m, fd_hw, nimg, rem);
            fdivmod(nimg, fd_T, clip, t);
            const long plane = fd_hw.d;
            const int T = (int)fd_T.d;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cc = c0 + 4 * h;
                const int step = cc < tsm_fold ? -1 : (cc < 2 * tsm_fold ? 1 : 0);          // destination frame = t + step
                const int td = (int)t + step;
                const bool inside = td >= 0 && td < T;
                const long md = inside ? m + step * plane : m - (long)step * (T - 1) * plane;
                const u32x2 val = inside ? (u32x2){raw[2 * h], raw[2 * h + 1]} : (u32x2){0u, 0u};
                *reinterpret_cast<u32x2*>(out + md * C + cc) = val;
            }
        } else {
            *reinterpret_cast<bf16x8*>(out + m * C + c0) = o;
        }
        if (relu_bits) {                           // bit i: out[m, c0 + i] > 0 (what the block input gradient masks with)
            unsigned bits = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) bits |= (v[i] > 0.f ? 1u : 0u) << i;
            relu_bits[m * CH + chunk] = (uint8_t)bits;
        }
        if (c0 < stat_c) {
            if (slice_out) *reinterpret_cast<bf16x8*>(slice_out + m * stat_c + c0) = o;
            if (stat_sum) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { ssum[i] += v[i]; ssq[i] += v[i] * v[i]; }
            }
        }
    }
    if (stat_c > 0 && stat_sum) {
        __syncthreads();
        STES_DET_LDS(det, if (c0 < stat_c) {
            _Pragma("unroll") for (int i = 0; i < 8; ++i) { atomicAdd(&lds[c0 + i], ssum[i]); atomicAdd(&lds[stat_c + c0 + i], ssq[i]); }
        });
        __syncthreads();
        det_enter(det);
        for (int i = threadIdx.x; i < stat_c; i += TPB) {
            atomicAdd(stat_sum + i, lds[i]);
            atomicAdd(stat_sumsq + i, lds[stat_c + i]);
        }
        det_leave(det);
    }
}

// ---------------------------------------------------------------- per-image channel pooling
// out[n,ch] = norm * sum_hw act(c[n,hw,ch]) * (mul ? mul[n,hw,ch] : 1);  act = relu(scale*c+shift) or identity
__global__ void chan_pool_kernel(const bf16* __restrict__ c, const float* __restrict__ scale,
                                 const float* __restrict__ shift, const bf16* __restrict__ mul,
                                 float* __restrict__ out, bf16* __restrict__ out_bf, int HW, int C, float norm, int R) {
    extern __shared__ float lds[];     // [R][C]
    const int CH = C / 8;
    const int n = blockIdx.x;
    const int chunk = threadIdx.x % CH, rg = threadIdx.x / CH;
    const int c0 = chunk * 8;
    float s[8], b[8], acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    if (scale) { load8f(scale + c0, s); load8f(shift + c0, b); }
    const bf16* base = c + (long)n * HW * C + c0;
    const bf16* mbase = mul ? mul + (long)n * HW * C + c0 : nullptr;
    // PU rows per trip, their loads requested together (the rows are still added in increasing order: same sums)
    constexpr int PU = 4;
    for (int r0 = rg; r0 < HW; r0 += PU * R) {
        bf16x8 rv[PU], rw[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int r = r0 + u * R;
            if (r < HW) {
                rv[u] = *reinterpret_cast<const bf16x8*>(base + (long)r * C);
                if (mbase) rw[u] = *reinterpret_cast<const bf16x8*>(mbase + (long)r * C);
            }
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            if (r0 + u * R >= HW) break;
            float v[8];
            unpack8(rv[u], v);
            if (scale) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i] * s[i] + b[i], 0.f);
            }
            if (mbase) {
                float w[8];
                unpack8(rw[u], w);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] *= w[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) lds[rg * C + c0 + i] = acc[i];
    __syncthreads();
    for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
        float t = 0.f;
        for (int r = 0; r < R; ++r) t += lds[r * C + ch];
        t *= norm;
        if (out) out[(long)n * C + ch] = t;
        if (out_bf) out_bf[(long)n * C + ch] = (bf16)t;
    }
}

// ---------------------------------------------------------------- squeeze-excite backward pooling
// One pass over (c2, dA3) per image gives everything that depends on them through per-image reductions: with
// mask = [scale*c2 + shift > 0],   S0 = sum mask, S1 = sum dA3*mask, S2 = sum mask*c2, S3 = sum dA3*mask*c2   (over h,w)
//   dgate[n,c]          = sum relu(bn(c2)) * dA3                     = scale*S3 + shift*S1
//   conv2.bn backward   : dy = (dA3*gate + dpool) * mask  =>  sum dy = sum_n gate*S1 + dpool*S0,
//                                                           sum dy*c2 = sum_n gate*S3 + dpool*S2     (se_bwd_bnsums_kernel)
// so the BatchNorm reduction pass over the two tensors is not needed.
__global__ void se_bwd_pool_kernel(const bf16* __restrict__ c, const float* __restrict__ scale, const float* __restrict__ shift,
                                   const bf16* __restrict__ dA, float* __restrict__ S, float* __restrict__ dgate,
                                   const float* __restrict__ gate, bf16* __restrict__ dpre, int N, int HW, int C, int R) {
    extern __shared__ float lds[];     // [R][4][C]
    const int CH = C / 8;
    const int n = blockIdx.x;
    const int chunk = threadIdx.x % CH, rg = threadIdx.x / CH;
    const int c0 = chunk * 8;
    float s[8], b[8], a0[8], a1[8], a2[8], a3[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; a3[i] = 0.f; }
    load8f(scale + c0, s); load8f(shift + c0, b);
    const bf16* cb = c + (long)n * HW * C + c0;
    const bf16* db = dA + (long)n * HW * C + c0;
    // two rows per trip, their four loads requested together (rows are still added in increasing order: same sums)
    constexpr int PU = 2;
    for (int r0 = rg; r0 < HW; r0 += PU * R) {
        bf16x8 rv[PU], rd[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int r = r0 + u * R;
            if (r < HW) {
                rv[u] = *reinterpret_cast<const bf16x8*>(cb + (long)r * C);
                rd[u] = *reinterpret_cast<const bf16x8*>(db + (long)r * C);
            }
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            if (r0 + u * R >= HW) break;
            float v[8], d[8];
            unpack8(rv[u], v);
            unpack8(rd[u], d);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool on = v[i] * s[i] + b[i] > 0.f;
                const float m = on ? 1.f : 0.f, dm = on ? d[i] : 0.f;
                a0[i] += m; a1[i] += dm; a2[i] += m * v[i]; a3[i] += dm * v[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float* l = lds + (long)rg * 4 * C + c0 + i;
        l[0] = a0[i]; l[C] = a1[i]; l[2 * C] = a2[i]; l[3 * C] = a3[i];
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] += lds[((long)r * 4 + k) * C + ch];
#pragma unroll
        for (int k = 0; k < 4; ++k) S[((long)k * N + n) * C + ch] = t[k];
        const float dg = scale[ch] * t[3] + shift[ch] * t[1];
        dgate[(long)n * C + ch] = dg;
        if (dpre) { const float gv = gate[(long)n * C + ch]; dpre[(long)n * C + ch] = (bf16)(dg * gv * (1.f - gv)); }   // sigmoid backward
    }
}

// s_dy[c] += sum_n gate*S1 + dpool*S0;  s_dyr[c] += sum_n gate*S3 + dpool*S2   grid (C/64, image slices), 256 threads =
// 64 channels x 4 image lanes; one contiguous atomic per slice (the caller zeroes s_dy / s_dyr)
__global__ __launch_bounds__(256) void se_bwd_bnsums_kernel(const float* __restrict__ S, const float* __restrict__ gate,
                                                            const float* __restrict__ dpool, float* __restrict__ s_dy,
                                                            float* __restrict__ s_dyr, int N, int C, StesDet det) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + cl;
    float a = 0.f, b = 0.f;
    if (ch < C) {
        for (int n = blockIdx.y * 4 + part; n < N; n += 4 * gridDim.y) {
            const long o = (long)n * C + ch;
            const float g = gate[o], dp = dpool[o];
            a += g * S[(long)N * C + o] + dp * S[o];
            b += g * S[3L * N * C + o] + dp * S[2L * N * C + o];
        }
    }
    red[0][part][cl] = a; red[1][part][cl] = b;
    __syncthreads();
    det_enter(det);
    if (part == 0 && ch < C) {
        atomicAdd(s_dy + ch, red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]);
        atomicAdd(s_dyr + ch, red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]);
    }
    det_leave(det);
}

// ---------------------------------------------------------------- sigmoid backward (squeeze-excite gate)
__global__ void sigmoid_bwd_kernel(const float* __restrict__ d, const float* __restrict__ g, bf16* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float gv = g[i]; out[i] = (bf16)(d[i] * gv * (1.f - gv)); }
}

// ---------------------------------------------------------------- per-clip channel gate over the frame features
// out[b,t,c] = x[b,t,c] * gate[b,c]   (8 channels per thread)
__global__ void clip_gate_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gate, bf16* __restrict__ out, int T, int C8,
                                     long total8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
        const long row = i / C8;
        const int c = (int)(i - row * C8) * 8;
        const float* g = gate + (row / T) * (long)C8 * 8 + c;
        float v[8];
        load8b(x + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= g[j];
        store8b(out + i * 8, v);
    }
}

// dgate[b,c] = sum_t (da[b,t,c] + db[b,t,c]) * x[b,t,c]; grid (ceil(C/256), B), the features before the gate in x
__global__ void clip_gate_bwd_reduce_kernel(const float* __restrict__ da, const float* __restrict__ db, const bf16* __restrict__ x,
                                            float* __restrict__ dgate, int T, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
        const long i = ((long)b * T + t) * C + c;
        acc += (da[i] + (db ? db[i] : 0.f)) * bf2f(x[i]);
    }
    dgate[(long)b * C + c] = acc;
}

// out[b,t,c] = (da + db)[b,t,c] * gate[b,c] + dpool[b,c]   (dpool already carries the 1/T of the mean; out may be da)
__global__ void clip_gate_bwd_apply_kernel(const float* da, const float* __restrict__ db, const float* __restrict__ gate,
                                           const float* __restrict__ dpool, float* out, int T, int C, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / C;
        const long g = (row / T) * C + (i - row * C);
        out[i] = (da[i] + (db ? db[i] : 0.f)) * gate[g] + dpool[g];
    }
}

// ---------------------------------------------------------------- BN backward
struct DySrc {
    const bf16* D; const float* mul; const float* add; FastDiv group;
    const bf16* act_raw; const float* act_scale; const float* act_shift;
};

// dY for 8 channels of row m
__device__ __forceinline__ void compute_dy(const DySrc& d, long m, int C, int c0, const float (&as)[8], const float (&ab)[8],
                                           float (&dy)[8]) {
    load8b(d.D + m * C + c0, dy);
    if (d.mul) {
        float mu[8];
        load8f(d.mul + (long)fdiv((uint32_t)m, d.group) * C + c0, mu);
#pragma unroll
        for (int i = 0; i < 8; ++i) dy[i] *= mu[i];
    }
    if (d.add) {
        float ad[8];
        load8f(d.add + (long)fdiv((uint32_t)m, d.group) * C + c0, ad);
#pragma unroll
        for (int i = 0; i < 8; ++i) dy[i] += ad[i];
    }
    if (d.act_raw) {
        float a[8];
        load8b(d.act_raw + m * C + c0, a);
#pragma unroll
        for (int i = 0; i < 8; ++i) dy[i] = (a[i] * as[i] + ab[i] > 0.f) ? dy[i] : 0.f;
    }
}

__global__ __launch_bounds__(TPB) void bn_bwd_reduce_kernel(const stes_bn_bwd_args p, FastDiv fd_group, StesDet det) {
    extern __shared__ float lds[];     // [3][C]
    const int C = p.C, CH = C / 8;
    const long gt = (long)blockIdx.x * TPB + threadIdx.x;
    const long TT = (long)gridDim.x * TPB;
    const int c0 = (int)(gt % CH) * 8;
    const long rstep = TT / CH;
    DySrc d{(const bf16*)p.D, p.mul, p.add, fd_group, (const bf16*)p.act_raw, p.act_scale, p.act_shift};
    float as[8], ab[8];
    if (d.act_raw) { load8f(p.act_scale + c0, as); load8f(p.act_shift + c0, ab); }
    float s0[8], s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s0[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f; }
    for (int i = threadIdx.x; i < 3 * C; i += TPB) lds[i] = 0.f;
    const bf16* r1 = (const bf16*)p.raw1; const bf16* r2 = (const bf16*)p.raw2;
    for (long m = gt / CH; m < p.M; m += rstep) {
        float dy[8], a[8];
        compute_dy(d, m, C, c0, as, ab, dy);
        load8b(r1 + m * C + c0, a);
#pragma unroll
        for (int i = 0; i < 8; ++i) { s0[i] += dy[i]; s1[i] += dy[i] * a[i]; }
        if (r2) {
            load8b(r2 + m * C + c0, a);
#pragma unroll
            for (int i = 0; i < 8; ++i) s2[i] += dy[i] * a[i];
        }
    }
    __syncthreads();
    STES_DET_LDS(det, _Pragma("unroll") for (int i = 0; i < 8; ++i) {
        atomicAdd(&lds[c0 + i], s0[i]); atomicAdd(&lds[C + c0 + i], s1[i]);
        if (r2) atomicAdd(&lds[2 * C + c0 + i], s2[i]);
    });
    __syncthreads();
    det_enter(det);
    for (int i = threadIdx.x; i < C; i += TPB) {
        atomicAdd(p.s_dy + i, lds[i]);
        atomicAdd(p.s_dyr1 + i, lds[C + i]);
        if (r2) atomicAdd(p.s_dyr2 + i, lds[2 * C + i]);
    }
    det_leave(det);
}

__global__ void bn_bwd_finalize_kernel(const float* s_dy, const float* s_dyr, const float* mean, const float* invstd,
                                       const float* gamma, double count, float* alpha, float* beta, float* delta,
                                       float* dgamma, float* dbeta, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float g = gamma ? gamma[c] : 1.f;
    const float inv = invstd[c], mu = mean[c];
    const float db = s_dy[c];
    const float dg = inv * (s_dyr[c] - mu * db);                // sum dY * xhat
    const float k = g * inv;
    const float inv_n = (float)(1.0 / count);
    alpha[c] = k;
    beta[c] = -k * inv * dg * inv_n;
    delta[c] = -k * db * inv_n + k * inv * mu * dg * inv_n;
    if (dgamma) dgamma[c] += dg;
    if (dbeta) dbeta[c] += db;
}

__global__ __launch_bounds__(TPB) void bn_bwd_apply_kernel(const stes_bn_bwd_args p, FastDiv fd_group) {
    const int C = p.C, CH = C / 8;
    const long gt = (long)blockIdx.x * TPB + threadIdx.x;
    const long TT = (long)gridDim.x * TPB;
    const int c0 = (int)(gt % CH) * 8;
    const long rstep = TT / CH;
    DySrc d{(const bf16*)p.D, p.mul, p.add, fd_group, (const bf16*)p.act_raw, p.act_scale, p.act_shift};
    float as[8], ab[8], al[8], be[8], de[8];
    if (d.act_raw) { load8f(p.act_scale + c0, as); load8f(p.act_shift + c0, ab); }
    load8f(p.alpha + c0, al); load8f(p.beta + c0, be); load8f(p.delta + c0, de);
    const bf16* r1 = (const bf16*)p.raw1;
    bf16* out = (bf16*)p.out;
    for (long m = gt / CH; m < p.M; m += rstep) {
        float dy[8], a[8];
        compute_dy(d, m, C, c0, as, ab, dy);
        load8b(r1 + m * C + c0, a);
#pragma unroll
        for (int i = 0; i < 8; ++i) dy[i] = al[i] * dy[i] + be[i] * a[i] + de[i];
        store8b(out + m * C + c0, dy);
    }
}

// ---------------------------------------------------------------- block input gradient
// out[m,c] = ((c < fold ? dXf[m,c] : dY[m,c]) + residual) * (X[m,c] > 0)
// `out` is the gradient at the output of the layer in front of this block; with raw1 (that layer's raw conv output feeding
// its last BatchNorm; raw2: the raw output of its shortcut conv) the BatchNorm-backward sums of that layer,
// s_dy[c] += sum out, s_dyr1[c] += sum out*raw1, s_dyr2[c] += sum out*raw2, are taken from the rounded values on their way
// out, which saves the separate reduction pass its two reads.
template <bool SUMS>
__global__ __launch_bounds__(TPB) void block_in_grad_kernel(const bf16* __restrict__ dY, long ldy, const bf16* __restrict__ dXf,
                                                            int fold, const bf16* __restrict__ resid, int ms, int Ho, int Wo,
                                                            int Hi, int Wi, const bf16* __restrict__ X, bf16* __restrict__ out,
                                                            long M, int C, FastDiv fd_hw, FastDiv fd_wi,
                                                            const bf16* __restrict__ raw1, const bf16* __restrict__ raw2,
                                                            float* __restrict__ s_dy, float* __restrict__ s_dyr1,
                                                            float* __restrict__ s_dyr2, const uint8_t* __restrict__ x_bits, StesDet det,
                                                            int tsm_fold, FastDiv fd_T) {
    extern __shared__ float lds[];     // [3][C] when SUMS
    const int CH = C / 8;
    const long gt = (long)blockIdx.x * TPB + threadIdx.x;
    const long TT = (long)gridDim.x * TPB;          // a multiple of CH (flat_grid): a thread keeps its 8 channels
    const int c0 = (int)(gt % CH) * 8;
    const long rstep = TT / CH;
    float s0[8], s1[8], s2[8];
    if (SUMS) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { s0[k] = 0.f; s1[k] = 0.f; s2[k] = 0.f; }
        for (int i = threadIdx.x; i < 3 * C; i += TPB) lds[i] = 0.f;
    }
    // Two rows per trip, every load of both rows requested before the first use: a wave keeps ten operand loads in flight
    // instead of five (the loop has no other way to overlap trips: each ends in a store).
    constexpr int U = 2;          // (four rows per trip measured slower: 195 vs 184 us per launch)
    for (long mb = gt / CH; mb < M; mb += U * rstep) {
        bf16x8 rv[U], rr[U], rx[U], ra[U], rb[U];
        unsigned bits[U];
        bool live[U], rok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long m = mb + u * rstep;
            live[u] = m < M;
            rok[u] = false;
            if (!live[u]) continue;
            if (c0 < 2 * tsm_fold) {
                // the adjoint of the plain temporal shift (reference model/shift.py:37-52), taken while dY is read: channels
                // [0, fold) get the gradient of frame t-1, [fold, 2 fold) that of frame t+1, zero beyond the clip; four channels
                // at a time (fold is a multiple of 4, so a piece lies in one segment)
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                uint32_t nimg, rem, clip, t;
                fdivmod((uint32_t)m, fd_hw, nimg, rem);
                fdivmod(nimg, fd_T, clip, t);
                const long plane = (long)Hi * Wi;
                u32x2 half[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = c0 + 4 * h;
                    const int step = c < tsm_fold ? -1 : (c < 2 * tsm_fold ? 1 : 0);
                    const int ts = (int)t + step;
                    half[h] = (u32x2){0u, 0u};
                    if (ts >= 0 && ts < (int)fd_T.d) half[h] = *reinterpret_cast<const u32x2*>(dY + (m + step * plane) * ldy + c);
                }
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                rv[u] = __builtin_bit_cast(bf16x8, (u32x4){half[0][0], half[0][1], half[1][0], half[1][1]});
            } else {
                rv[u] = (c0 < fold) ? *reinterpret_cast<const bf16x8*>(dXf + m * fold + c0) : *reinterpret_cast<const bf16x8*>(dY + m * ldy + c0);
            }
            if (resid) {
                long rm = m; bool ok = true;
                if (ms > 1) {
                    uint32_t n, rem, hi, wi;
                    fdivmod((uint32_t)m, fd_hw, n, rem);
                    fdivmod(rem, fd_wi, hi, wi);
                    ok = ((hi | wi) & 1) == 0 && (int)(hi >> 1) < Ho && (int)(wi >> 1) < Wo;      // ms == 2
                    rm = ((long)n * Ho + (hi >> 1)) * Wo + (wi >> 1);
                }
                rok[u] = ok;
                if (ok) rr[u] = *reinterpret_cast<const bf16x8*>(resid + rm * C + c0);
            }
            if (x_bits) bits[u] = x_bits[m * CH + (c0 >> 3)];        // [M][C/8] bytes, bit k: X[m, c0 + k] > 0 (a sixteenth of X's bytes)
            else rx[u] = *reinterpret_cast<const bf16x8*>(X + m * C + c0);
            if (SUMS) {
                ra[u] = *reinterpret_cast<const bf16x8*>(raw1 + m * C + c0);
                if (raw2) rb[u] = *reinterpret_cast<const bf16x8*>(raw2 + m * C + c0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            const long m = mb + u * rstep;
            float v[8];
            unpack8(rv[u], v);
            if (rok[u]) {
                float r[8];
                unpack8(rr[u], r);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += r[k];
            }
            if (x_bits) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (bits[u] >> k) & 1u ? v[k] : 0.f;
            } else {
                float x[8];
                unpack8(rx[u], x);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = x[k] > 0.f ? v[k] : 0.f;
            }
            store8b(out + m * C + c0, v);
            if (SUMS) {
                float a[8];
                unpack8(ra[u], a);
#pragma unroll
                for (int k = 0; k < 8; ++k) { v[k] = (float)(bf16)v[k]; s0[k] += v[k]; s1[k] += v[k] * a[k]; }
                if (raw2) {
                    unpack8(rb[u], a);
#pragma unroll
                    for (int k = 0; k < 8; ++k) s2[k] += v[k] * a[k];
                }
            }
        }
    }
    if (SUMS) {
        __syncthreads();
        STES_DET_LDS(det, _Pragma("unroll") for (int k = 0; k < 8; ++k) {
            atomicAdd(&lds[c0 + k], s0[k]); atomicAdd(&lds[C + c0 + k], s1[k]);
            if (raw2) atomicAdd(&lds[2 * C + c0 + k], s2[k]);
        });
        __syncthreads();
        det_enter(det);
        for (int i = threadIdx.x; i < C; i += TPB) {
            atomicAdd(s_dy + i, lds[i]);
            atomicAdd(s_dyr1 + i, lds[C + i]);
            if (raw2) atomicAdd(s_dyr2 + i, lds[2 * C + i]);
        }
        det_leave(det);
    }
}

// dZ[m,c] = (da[n,c] + db[n,c]) * inv_hw * (out[m,c] > 0)
__global__ __launch_bounds__(TPB) void pool_bwd_kernel(const float* __restrict__ da, const float* __restrict__ db,
                                                       const bf16* __restrict__ outp, bf16* __restrict__ dZ, int HW, int C,
                                                       float inv_hw, long M, FastDiv fd_ch, FastDiv fd_hw) {
    const int CH = C / 8;
    const uint32_t total = (uint32_t)(M * CH);
    for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < total; i += gridDim.x * TPB) {
        uint32_t m, chk;
        fdivmod(i, fd_ch, m, chk);
        const int c0 = (int)chk * 8;
        const long n = fdiv(m, fd_hw);
        float g[8], o[8];
        load8f(da + n * C + c0, g);
        if (db) {
            float h[8];
            load8f(db + n * C + c0, h);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] += h[k];
        }
        load8b(outp + (long)m * C + c0, o);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = o[k] > 0.f ? g[k] * inv_hw : 0.f;
        store8b(dZ + (long)m * C + c0, g);
    }
}

}  // namespace

#define S(s) ((hipStream_t)(s))

extern "C" int stes_bn_finalize(const float* sum, const float* sumsq, double count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, int training,
                                float* scale, float* shift, float* save_mean, float* save_invstd, int C, void* stream) {
    STES_CHECK(scale && shift && C > 0, STES_ERR_ARG, "stes_bn_finalize: null output");
    STES_CHECK(training ? (sum && sumsq && count > 0) : (running_mean && running_var), STES_ERR_ARG, "stes_bn_finalize: missing statistics");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(stes_div_up(C, 128)), dim3(128), 0, S(stream), sum, sumsq, count, gamma, beta,
                       running_mean, running_var, momentum, eps, training, scale, shift, save_mean, save_invstd, C);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_bn_act_residual(const uint16_t* c, const float* scale, const float* shift, const uint16_t* res,
                                    const float* res_scale, const float* res_shift, uint16_t* out, float* stat_sum,
                                    float* stat_sumsq, int stat_c, uint16_t* slice_out, int64_t M, int C, uint8_t* relu_bits,
                                    void* stream) {
    STES_CHECK(c && scale && shift && out && M > 0, STES_ERR_ARG, "stes_bn_act_residual: null pointer");
    STES_CHECK(C % 8 == 0 && stat_c % 8 == 0 && stat_c <= C, STES_ERR_ARG, "stes_bn_act_residual: C and stat_c must be multiples of 8");
    STES_CHECK((stat_sum == nullptr) == (stat_sumsq == nullptr), STES_ERR_ARG, "stes_bn_act_residual: stats pointers");
    STES_CHECK(stat_c == 0 || stat_sum || slice_out, STES_ERR_ARG, "stes_bn_act_residual: stat_c without an output");
    static const int cap = resident_blocks(bn_act_residual_kernel);
    // (measured per step over caps: 1.90 ms at 1,024 and 768 workgroups, 1.97 at the resident 1,280, 2.09 at 2,048, 2.29 at 512)
    const int grid = flat_grid(M, C / 8, cap < 1024 ? cap : 1024);
    hipLaunchKernelGGL(bn_act_residual_kernel, dim3(grid), dim3(TPB), 2 * stat_c * sizeof(float), S(stream), (const bf16*)c, scale,
                       shift, (const bf16*)res, res_scale, res_shift, (bf16*)out, stat_sum, stat_sumsq, stat_c, (bf16*)slice_out, (long)M, C, relu_bits,
                       stat_c > 0 && stat_sum ? stes_det_take(grid) : StesDet{nullptr, 0u, 0u}, 0, make_fastdiv(1), make_fastdiv(1));
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_bn_act_residual_tsm(const uint16_t* c, const float* scale, const float* shift, const uint16_t* res,
                                        const float* res_scale, const float* res_shift, uint16_t* out, int64_t M, int C,
                                        int tsm_fold, int T, int HW, void* stream) {
    STES_CHECK(c && scale && shift && out && M > 0, STES_ERR_ARG, "stes_bn_act_residual_tsm: null pointer");
    STES_CHECK(C % 8 == 0 && (res_scale == nullptr) == (res_shift == nullptr) && (!res_scale || res), STES_ERR_ARG, "stes_bn_act_residual_tsm: bad argument");
    STES_CHECK(tsm_fold >= 0 && tsm_fold % 4 == 0 && 2 * tsm_fold <= C && T > 0 && HW > 0 && M % ((int64_t)T * HW) == 0 && M <= 0xffffffffL,
               STES_ERR_ARG, "stes_bn_act_residual_tsm: the shifted width must be a multiple of 4, on whole clips");
    static const int cap = resident_blocks(bn_act_residual_kernel);
    const int grid = flat_grid(M, C / 8, cap < 1024 ? cap : 1024);
    hipLaunchKernelGGL(bn_act_residual_kernel, dim3(grid), dim3(TPB), 0, S(stream), (const bf16*)c, scale, shift, (const bf16*)res, res_scale,
                       res_shift, (bf16*)out, (float*)nullptr, (float*)nullptr, 0, (bf16*)nullptr, (long)M, C, (uint8_t*)nullptr,
                       StesDet{nullptr, 0u, 0u}, tsm_fold, make_fastdiv(HW), make_fastdiv(T));
    STES_LAUNCH_CHECK();
    return STES_OK;
}

// out[m][c] = bf16(relu(c[m][c] * scale[c] + shift[c]) * gate[m / HW][c]): the operand of a bottleneck's last 1x1 convolution
// written out once.  Where that convolution runs on the tiled GEMM (768 channels) its operand prologue costs more than this
// pass plus a plain product, and the weight gradient reads the same tensor instead of transforming its operand again.
__global__ __launch_bounds__(TPB) void bn_act_gate_kernel(const bf16* __restrict__ c, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ gate,
                                                          bf16* __restrict__ out, long M, int C, FastDiv fd_hw) {
    const int CH = C / 8;
    const long gt = (long)blockIdx.x * TPB + threadIdx.x, TT = (long)gridDim.x * TPB;
    const int chunk = (int)(gt % CH), c0 = chunk * 8;
    float s[8], b[8];
    load8f(scale + c0, s); load8f(shift + c0, b);
    for (long m = gt / CH; m < M; m += TT / CH) {
        float v[8], g[8];
        load8b(c + m * C + c0, v);
        load8f(gate + (long)fdiv((uint32_t)m, fd_hw) * C + c0, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i] * s[i] + b[i], 0.f) * g[i];
        store8b(out + m * C + c0, v);
    }
}

extern "C" int stes_bn_act_gate(const uint16_t* c, const float* scale, const float* shift, const float* gate, uint16_t* out, int N, int HW,
                                int C, void* stream) {
    STES_CHECK(c && scale && shift && gate && out && N > 0 && HW > 0, STES_ERR_ARG, "stes_bn_act_gate: null pointer");
    STES_CHECK(C % 8 == 0, STES_ERR_ARG, "stes_bn_act_gate: C must be a multiple of 8");
    const long M = (long)N * HW;
    STES_CHECK(M < (1L << 32), STES_ERR_UNSUPPORTED, "stes_bn_act_gate: too many rows");
    static const int cap = resident_blocks(bn_act_gate_kernel);
    const int grid = flat_grid(M, C / 8, cap < 1024 ? cap : 1024);
    hipLaunchKernelGGL(bn_act_gate_kernel, dim3(grid), dim3(TPB), 0, S(stream), (const bf16*)c, scale, shift, gate, (bf16*)out, M, C,
                       make_fastdiv(HW));
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_chan_pool(const uint16_t* c, const float* scale, const float* shift, const uint16_t* mul, float* out,
                              uint16_t* out_bf16, int N, int HW, int C, float norm, void* stream) {
    STES_CHECK(c && (out || out_bf16) && N > 0 && HW > 0, STES_ERR_ARG, "stes_chan_pool: null pointer");
    STES_CHECK(C % 8 == 0 && C / 8 <= TPB, STES_ERR_ARG, "stes_chan_pool: C must be a multiple of 8, at most 2048");
    const int CH = C / 8;
    int R = TPB / CH;
    if (R > HW) R = HW;
    if (R < 1) R = 1;
    hipLaunchKernelGGL(chan_pool_kernel, dim3(N), dim3(CH * R), (size_t)R * C * sizeof(float), S(stream), (const bf16*)c, scale, shift,
                       (const bf16*)mul, out, (bf16*)out_bf16, HW, C, norm, R);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_se_bwd_pool(const uint16_t* c2, const float* scale, const float* shift, const uint16_t* dA3, float* S,
                                float* dgate, const float* gate, uint16_t* dpre, int N, int HW, int C, void* stream) {
    STES_CHECK(c2 && scale && shift && dA3 && S && dgate && N > 0 && HW > 0, STES_ERR_ARG, "stes_se_bwd_pool: null pointer");
    STES_CHECK(!dpre || gate, STES_ERR_ARG, "stes_se_bwd_pool: dpre needs the gate");
    STES_CHECK(C % 8 == 0 && C / 8 <= TPB, STES_ERR_ARG, "stes_se_bwd_pool: C must be a multiple of 8, at most 2048");
    const int CH = C / 8;
    int R = TPB / CH;
    if (R > HW) R = HW;
    if (R < 1) R = 1;
    hipLaunchKernelGGL(se_bwd_pool_kernel, dim3(N), dim3(CH * R), (size_t)R * 4 * C * sizeof(float), S(stream), (const bf16*)c2, scale, shift,
                       (const bf16*)dA3, S, dgate, gate, (bf16*)dpre, N, HW, C, R);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_se_bwd_bnsums(const float* S, const float* gate, const float* dpool, float* s_dy, float* s_dyr, int N, int C,
                                  void* stream) {
    STES_CHECK(S && gate && dpool && s_dy && s_dyr && N > 0 && C > 0, STES_ERR_ARG, "stes_se_bwd_bnsums: null pointer");
    int slices = N / 32; if (slices < 1) slices = 1; if (slices > 64) slices = 64;
    hipLaunchKernelGGL(se_bwd_bnsums_kernel, dim3(stes_div_up(C, 64), slices), dim3(256), 0, S(stream), S, gate, dpool, s_dy, s_dyr, N, C,
                       stes_det_take(stes_div_up(C, 64) * slices));
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_sigmoid_bwd(const float* d, const float* g, uint16_t* out, int64_t n, void* stream) {
    STES_CHECK(d && g && out && n > 0, STES_ERR_ARG, "stes_sigmoid_bwd: bad argument");
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(stes_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, d, g, (bf16*)out, (long)n);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_clip_gate_fwd(const uint16_t* x, const float* gate, uint16_t* out, int B, int T, int C, void* stream) {
    STES_CHECK(x && gate && out && B > 0 && T > 0 && C > 0 && C % 8 == 0, STES_ERR_ARG, "stes_clip_gate_fwd: bad argument");
    const long total8 = (long)B * T * (C / 8);
    long blocks = stes_div_up(total8, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(clip_gate_fwd_kernel, dim3((int)blocks), dim3(256), 0, S(stream), (const bf16*)x, gate, (bf16*)out, T, C / 8, total8);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_clip_gate_bwd(const float* da, const float* db, const uint16_t* x, const float* gate, const float* dpool,
                                  float* dgate, float* out, int B, int T, int C, void* stream) {
    STES_CHECK(da && B > 0 && T > 0 && C > 0, STES_ERR_ARG, "stes_clip_gate_bwd: bad argument");
    STES_CHECK((x && dgate && !out) || (gate && dpool && out && !dgate), STES_ERR_ARG,
               "stes_clip_gate_bwd: pass (x, dgate) for the reduction or (gate, dpool, out) for the apply pass");
    if (dgate) {
        hipLaunchKernelGGL(clip_gate_bwd_reduce_kernel, dim3(stes_div_up(C, 256), B), dim3(256), 0, S(stream), da, db, (const bf16*)x, dgate, T, C);
    } else {
        const long total = (long)B * T * C;
        long blocks = stes_div_up(total, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(clip_gate_bwd_apply_kernel, dim3((int)blocks), dim3(256), 0, S(stream), da, db, gate, dpool, out, T, C, total);
    }
    STES_LAUNCH_CHECK();
    return STES_OK;
}

static int check_bwd(const stes_bn_bwd_args* a, const char* who) {
    STES_CHECK(a && a->D && a->raw1 && a->M > 0, STES_ERR_ARG, who);
    STES_CHECK(a->M <= STES_MAX_ROWS, STES_ERR_UNSUPPORTED, "bn_bwd: more than 2^31 rows");
    STES_CHECK(a->C % 8 == 0, STES_ERR_ARG, "bn_bwd: C must be a multiple of 8");
    STES_CHECK((!a->mul && !a->add) || a->group > 0, STES_ERR_ARG, "bn_bwd: group");
    STES_CHECK(!a->act_raw || (a->act_scale && a->act_shift), STES_ERR_ARG, "bn_bwd: mask affine");
    return STES_OK;
}

extern "C" int stes_bn_bwd_reduce(const stes_bn_bwd_args* a, void* stream) {
    int rc = check_bwd(a, "stes_bn_bwd_reduce: null pointer");
    if (rc) return rc;
    STES_CHECK(a->s_dy && a->s_dyr1 && (!a->raw2 || a->s_dyr2), STES_ERR_ARG, "stes_bn_bwd_reduce: outputs");
    const int grid = flat_grid(a->M, a->C / 8, 1024);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(grid), dim3(TPB), 3 * a->C * sizeof(float), S(stream), *a, make_fastdiv(a->group > 0 ? a->group : 1),
                       stes_det_take(grid));
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_bn_bwd_finalize(const float* s_dy, const float* s_dyr, const float* mean, const float* invstd,
                                    const float* gamma, double count, float* alpha, float* beta, float* delta,
                                    float* dgamma, float* dbeta, int C, void* stream) {
    STES_CHECK(s_dy && s_dyr && mean && invstd && alpha && beta && delta && count > 0, STES_ERR_ARG, "stes_bn_bwd_finalize: null pointer");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(stes_div_up(C, 128)), dim3(128), 0, S(stream), s_dy, s_dyr, mean, invstd, gamma,
                       count, alpha, beta, delta, dgamma, dbeta, C);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_bn_bwd_apply(const stes_bn_bwd_args* a, void* stream) {
    int rc = check_bwd(a, "stes_bn_bwd_apply: null pointer");
    if (rc) return rc;
    STES_CHECK(a->alpha && a->beta && a->delta && a->out, STES_ERR_ARG, "stes_bn_bwd_apply: coefficients/output");
    static const int cap = resident_blocks(bn_bwd_apply_kernel);
    // (measured per step over caps: 1.75 ms at the resident 1,536 workgroups, 2.15 at 1,024, 2.54 at 768, 2.19 at 2,048)
    const int grid = flat_grid(a->M, a->C / 8, cap);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(TPB), 0, S(stream), *a, make_fastdiv(a->group > 0 ? a->group : 1));
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_block_in_grad(const uint16_t* dY, int64_t ldy, const uint16_t* dXf, int fold, const uint16_t* resid,
                                  int map_stride, int Ho, int Wo, int Hi, int Wi, const uint16_t* X, uint16_t* out,
                                  int64_t M, int C, const uint16_t* raw1, const uint16_t* raw2, float* s_dy, float* s_dyr1,
                                  float* s_dyr2, const uint8_t* x_bits, void* stream) {
    return stes_block_in_grad_tsm(dY, ldy, dXf, fold, resid, map_stride, Ho, Wo, Hi, Wi, X, out, M, C, raw1, raw2, s_dy, s_dyr1, s_dyr2,
                                  x_bits, 0, 1, stream);
}

extern "C" int stes_block_in_grad_tsm(const uint16_t* dY, int64_t ldy, const uint16_t* dXf, int fold, const uint16_t* resid,
                                      int map_stride, int Ho, int Wo, int Hi, int Wi, const uint16_t* X, uint16_t* out,
                                      int64_t M, int C, const uint16_t* raw1, const uint16_t* raw2, float* s_dy, float* s_dyr1,
                                      float* s_dyr2, const uint8_t* x_bits, int tsm_fold, int T, void* stream) {
    STES_CHECK(dY && (X || x_bits) && out && M > 0, STES_ERR_ARG, "stes_block_in_grad: null pointer");
    STES_CHECK(tsm_fold >= 0 && tsm_fold % 4 == 0 && 2 * tsm_fold <= C && T > 0 && (tsm_fold == 0 || (fold == 0 && M % ((int64_t)Hi * Wi * T) == 0)),
               STES_ERR_ARG, "stes_block_in_grad_tsm: the shifted width must be a multiple of 4, without a gated-shift slice, on whole clips");
    STES_CHECK(C % 8 == 0 && fold % 8 == 0 && ldy % 8 == 0 && (fold == 0 || dXf), STES_ERR_ARG, "stes_block_in_grad: alignment");
    STES_CHECK(map_stride == 1 || map_stride == 2, STES_ERR_UNSUPPORTED, "stes_block_in_grad: stride must be 1 or 2");
    STES_CHECK(M <= 0xffffffffL, STES_ERR_UNSUPPORTED, "stes_block_in_grad: tensor too large");
    STES_CHECK(!raw1 || (s_dy && s_dyr1 && (!raw2 || s_dyr2)), STES_ERR_ARG, "stes_block_in_grad: sums need their outputs");
    STES_CHECK(raw1 || !raw2, STES_ERR_ARG, "stes_block_in_grad: raw2 without raw1");
    // (with the sums and two rows per trip: 177 us per launch at 512 workgroups, 181 at 768, 187 at 1,024, 196 at 1,280, 227 at
    // 3,072; counts that are not a multiple of the 256 CUs lose 10 us: 192 at 640, 191 at 896)
    const int grid = flat_grid(M, C / 8, raw1 ? 512 : 4096);
    const FastDiv fhw = make_fastdiv(Hi * Wi), fwi = make_fastdiv(Wi);
    if (raw1)
        hipLaunchKernelGGL(block_in_grad_kernel<true>, dim3(grid), dim3(TPB), 3 * C * sizeof(float), S(stream), (const bf16*)dY, (long)ldy,
                           (const bf16*)dXf, fold, (const bf16*)resid, map_stride, Ho, Wo, Hi, Wi, (const bf16*)X, (bf16*)out, (long)M, C,
                           fhw, fwi, (const bf16*)raw1, (const bf16*)raw2, s_dy, s_dyr1, s_dyr2, x_bits, stes_det_take(grid), tsm_fold, make_fastdiv(T));
    else
        hipLaunchKernelGGL(block_in_grad_kernel<false>, dim3(grid), dim3(TPB), 0, S(stream), (const bf16*)dY, (long)ldy, (const bf16*)dXf,
                           fold, (const bf16*)resid, map_stride, Ho, Wo, Hi, Wi, (const bf16*)X, (bf16*)out, (long)M, C, fhw, fwi,
                           (const bf16*)nullptr, (const bf16*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, x_bits, StesDet{nullptr, 0u, 0u},
                           tsm_fold, make_fastdiv(T));
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_pool_bwd(const float* da, const float* db, const uint16_t* out, uint16_t* dZ, int N, int HW, int C,
                             void* stream) {
    STES_CHECK(da && out && dZ && N > 0 && HW > 0 && C % 8 == 0, STES_ERR_ARG, "stes_pool_bwd: bad argument");
    const long M = (long)N * HW;
    long blocks = (M * (C / 8) + TPB - 1) / TPB;
    if (blocks > 4096) blocks = 4096;
    STES_CHECK(M * (C / 8) <= 0xffffffffL, STES_ERR_UNSUPPORTED, "stes_pool_bwd: tensor too large");
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((int)blocks), dim3(TPB), 0, S(stream), da, db, (const bf16*)out, (bf16*)dZ, HW, C,
                       1.0f / HW, M, make_fastdiv(C / 8), make_fastdiv(HW));
    STES_LAUNCH_CHECK();
    return STES_OK;
}
