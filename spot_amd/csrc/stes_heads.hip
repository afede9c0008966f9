// Temporal head and training-step kernels: GRU recurrence (forward / BPTT step), minGRU scan,
// dropout mask, weighted-CE + masked-L1 loss with its gradient, AdamW, fp32 -> bf16 weight casts.
//
// GRU (torch.nn.GRU semantics, gate order r,z,n; reference model/modules.py:24-30): the input
// projection of every time step is one big MFMA GEMM done beforehand (stes_gemm).  The recurrence
// is one launch per time step; a workgroup owns 16 hidden units of one direction for all (<=16)
// batch rows, its 4 waves split K = H for the [B,H] x [H,3*16] product on
// v_mfma_f32_16x16x32_bf16 with the hidden state as the A operand (read straight from the output
// sequence) and the W_hh rows as B, then the gate math runs on the accumulator layout.
#include <mutex>
#include "stes_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// Gate arithmetic of one (batch row, hidden unit), shared by the launch-per-step and the one-launch kernels.  Contraction
// into fused multiply-adds is off here: left to the compiler it depends on the surrounding code, and the two forms of
// the recurrence must round alike (they are tested to agree bit for bit).
struct GruFwdOut { float r, z, n, hn, h; };
__device__ __forceinline__ GruFwdOut gru_gate_fwd(float ir, float iz, float in, float gh_r, float gh_z, float gh_n, float bh_r, float bh_z,
                                                  float bh_n, float hp) {
#pragma clang fp contract(off)
    GruFwdOut o;
    o.hn = gh_n + bh_n;
    o.r = sigmoidf_(ir + gh_r + bh_r);
    o.z = sigmoidf_(iz + gh_z + bh_z);
    o.n = tanhf(in + o.r * o.hn);
    o.h = (1.f - o.z) * o.n + o.z * hp;
    return o;
}
struct GruBwdOut { float dr, dz, dn, dnr, dh_next; };
__device__ __forceinline__ GruBwdOut gru_gate_bwd(float dh, float rg, float zg, float ng, float hn, float hp) {
#pragma clang fp contract(off)
    GruBwdOut o;
    const float dn = dh * (1.f - zg);
    const float dz = dh * (hp - ng);
    o.dn = dn * (1.f - ng * ng);
    o.dr = o.dn * hn * rg * (1.f - rg);
    o.dz = dz * zg * (1.f - zg);
    o.dnr = o.dn * rg;
    o.dh_next = dh * zg;
    return o;
}

// ------------------------------------------------------------------ GRU forward step
// grid (H/16, 2 dirs, batch tiles of 16); block 256
__global__ __launch_bounds__(256) void gru_fwd_step_kernel(const stes_gru_args p, int step) {
    __shared__ float red[4][64][12];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int dir = blockIdx.y, j0 = blockIdx.x * 16, b0 = blockIdx.z * 16;
    const int H = p.H, T = p.T, B = p.B;
    const int t = dir == 0 ? step : T - 1 - step;
    const int tprev = dir == 0 ? t - 1 : t + 1;
    const bf16* y = (const bf16*)p.y;
    const bf16* __restrict__ whh = (const bf16*)p.whh + (long)dir * 3 * H * H;
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (step > 0) {
        const int nk = H / 32;
        const int brow = b0 + fr;
        const bf16* arow = (brow < B) ? y + ((long)brow * T + tprev) * 2 * H + dir * H : nullptr;
        // a step is a few microseconds of pure latency: all loads of six k-steps are issued before the first MFMA needs one
        constexpr int U = 6;
        for (int ks0 = wave; ks0 < nk; ks0 += 4 * U) {
            bf16x8 fa[U], fb[U][3];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ks = ks0 + 4 * u;
                const int k = ks * 32 + fq * 8;
                const bool on = ks < nk;
                fa[u] = (on && arow) ? *reinterpret_cast<const bf16x8*>(arow + k) : zero8();
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    fb[u][g] = on ? *reinterpret_cast<const bf16x8*>(whh + ((long)g * H + j0 + fr) * H + k) : zero8();
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[u], fb[u][g], acc[g], 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][lane][g * 4 + r] = acc[g][r];
    __syncthreads();
    if (wave != 0) return;
    const int j = j0 + fr;
    const float* bhh = p.bhh + (long)dir * 3 * H;
    const float bh_r = bhh[j], bh_z = bhh[H + j], bh_n = bhh[2 * H + j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + fq * 4 + r;
        if (b >= B) continue;
        float gh[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) gh[g] = red[0][lane][g * 4 + r] + red[1][lane][g * 4 + r] + red[2][lane][g * 4 + r] + red[3][lane][g * 4 + r];
        const bf16* gi = (const bf16*)p.gi + (((long)b * T + t) * 2 + dir) * 3 * H;
        const float ir = (float)gi[j], iz = (float)gi[H + j], in = (float)gi[2 * H + j];
        const float hp = step > 0 ? p.hstate[(((long)dir * 2 + ((step - 1) & 1)) * B + b) * H + j] : 0.f;
        const GruFwdOut go = gru_gate_fwd(ir, iz, in, gh[0], gh[1], gh[2], bh_r, bh_z, bh_n, hp);
        const float hn = go.hn, rg = go.r, zg = go.z, ng = go.n, h = go.h;
        p.hstate[(((long)dir * 2 + (step & 1)) * B + b) * H + j] = h;
        ((bf16*)p.y)[((long)b * T + t) * 2 * H + dir * H + j] = (bf16)h;
        if (p.saved) {
            float* sv = p.saved + ((((long)dir * T + t) * B + b) * 5) * H;      // r, z, n, hn, hprev
            sv[j] = rg; sv[H + j] = zg; sv[2 * H + j] = ng; sv[3 * H + j] = hn; sv[4 * H + j] = hp;
            ((bf16*)p.hprev_bf)[(((long)dir * T + t) * B + b) * H + j] = (bf16)hp;
        }
    }
}

// ------------------------------------------------------------------ GRU forward, the whole recurrence in ONE launch
// grid (H/32, 2 dirs), 512 threads; B <= 16, H <= 768.  A workgroup owns 32 hidden units of one direction for all T steps:
// its W_hh rows (3 gates x 32 units x H) stay in registers as MFMA B fragments (wave = unit half x quarter of the k-steps,
// the k-step assignment and every sum in the order of gru_fwd_step_kernel, so the results are the same bit for bit), its
// fp32 hidden state stays in registers too.  Between steps the workgroups of a direction exchange the new hidden state
// (bf16, what the step kernel reads back from y) through a small global buffer without any cache write-back or
// invalidate: a workgroup's 16 x 32 tile is exactly the A-operand tile of one k-step, written as ONE 1 KiB write-through
// (sc1) store instruction of one wave, followed by that wave's vmcnt(0) wait and one agent-scope counter add; consumers
// poll the counter with an sc1 load, pass a workgroup barrier and read the tiles with sc1 loads (L2-served).  Spins are
// bounded.  hx: [dir][parity][H/32 tiles][16 rows][32 units] bf16, cnt: one counter per direction on its own line.
constexpr int GRU_U = 6;                                       // k-steps per wave (H <= 768)
__device__ __forceinline__ bf16x8 load_sc1(const void* ptr) {
    bf16x8 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ void store_sc1(void* ptr, bf16x8 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory"); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// All workgroups of the direction have published step `step - 1` (each adds one per step).  The spin is bounded; when the
// bound is reached (a peer workgroup not resident in time) the wait reports it instead of letting the caller read stale
// tiles: the word at `err` (host memory that the device can write, checked by the next stes_gru_* call) is raised, and the
// return value makes the workgroup poison everything it produces from then on (NaN), so the failure cannot pass unnoticed.
__device__ __forceinline__ bool gru_wait(unsigned* cnt, unsigned target, int limit, unsigned* err) {
    int spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < limit) __builtin_amdgcn_s_sleep(1);
    if (limit > 0 && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    if (err) __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return false;
}

__global__ __launch_bounds__(512) void gru_fwd_persist_kernel(const stes_gru_args p, unsigned* __restrict__ cnt, bf16* __restrict__ hx,
                                                              int spin_limit, unsigned* err) {
    __shared__ float red[8][64][12];
    __shared__ __attribute__((aligned(16))) bf16 tile[16 * 32];
    __shared__ int failed;
    if (threadIdx.x == 0) failed = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int uh = wave >> 2, kq = wave & 3;               // unit half, k-step residue
    const int blk = blockIdx.x, dir = blockIdx.y;
    const int H = p.H, T = p.T, B = p.B, nk = H / 32;
    const int j = blk * 32 + uh * 16 + fr;                 // B operand column / epilogue unit of this lane
    const bf16* __restrict__ whh = (const bf16*)p.whh + (long)dir * 3 * H * H;
    bf16x8 fb[GRU_U][3];
#pragma unroll
    for (int u = 0; u < GRU_U; ++u) {
        const int ks = kq + 4 * u, k = ks * 32 + fq * 8;
#pragma unroll
        for (int g = 0; g < 3; ++g) fb[u][g] = ks < nk ? *reinterpret_cast<const bf16x8*>(whh + ((long)g * H + j) * H + k) : zero8();
    }
    const bool epi = kq == 0;                              // waves 0 and 4 finish the gates of their unit half
    const float* bhh = p.bhh + (long)dir * 3 * H;
    const float bh_r = bhh[j], bh_z = bhh[H + j], bh_n = bhh[2 * H + j];
    unsigned* my_cnt = cnt + dir * 32;
    bf16* hx_dir = hx + (long)dir * 2 * nk * 512;
    float hp[4] = {0.f, 0.f, 0.f, 0.f};
    float gic[4][3];                                       // input projections of the current step, fetched a step ahead
    auto load_gi = [&](int step, float (&dst)[4][3]) {
        const int t = dir == 0 ? step : T - 1 - step;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = fq * 4 + r;
            const uint32_t o = (uint32_t)((((b < B ? b : 0) * T + t) * 2 + dir) * 3 * H + j);      // (element counts < 2^31: checked by the host)
            const bf16* gi = (const bf16*)p.gi;
            dst[r][0] = (float)gi[o]; dst[r][1] = (float)gi[o + (uint32_t)H]; dst[r][2] = (float)gi[o + 2u * (uint32_t)H];
        }
    };
    if (epi) load_gi(0, gic);
    for (int step = 0; step < T; ++step) {
        const int t = dir == 0 ? step : T - 1 - step;
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
            if (threadIdx.x == 0 && !gru_wait(my_cnt, (unsigned)gridDim.x * (unsigned)step, spin_limit, err)) failed = 1;
            __syncthreads();
            const bf16* src = hx_dir + (long)((step - 1) & 1) * nk * 512 + fr * 32 + fq * 8;
            bf16x8 fa[GRU_U];
#pragma unroll
            for (int u = 0; u < GRU_U; ++u) {
                const int ks = kq + 4 * u;
                fa[u] = load_sc1(src + (long)(ks < nk ? ks : 0) * 512);
            }
            wait_vm0();
#pragma unroll
            for (int u = 0; u < GRU_U; ++u)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    if (kq + 4 * u < nk) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[u], fb[u][g], acc[g], 0, 0, 0);
            if (failed) acc[0][0] = acc[0][1] = acc[0][2] = acc[0][3] = __builtin_nanf("");
        }
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][lane][g * 4 + r] = acc[g][r];
        __syncthreads();
        float rg[4], zg[4], ng[4], hn[4], hold[4];
        if (epi) {
            const int w0 = uh * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float gh[3];
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    gh[g] = red[w0][lane][g * 4 + r] + red[w0 + 1][lane][g * 4 + r] + red[w0 + 2][lane][g * 4 + r] + red[w0 + 3][lane][g * 4 + r];
                hold[r] = hp[r];
                const GruFwdOut go = gru_gate_fwd(gic[r][0], gic[r][1], gic[r][2], gh[0], gh[1], gh[2], bh_r, bh_z, bh_n, hold[r]);
                hn[r] = go.hn; rg[r] = go.r; zg[r] = go.z; ng[r] = go.n;
                const float h = go.h;
                hp[r] = h;
                tile[(fq * 4 + r) * 32 + uh * 16 + fr] = (fq * 4 + r < B) ? (bf16)h : (bf16)0.f;
            }
        }
        __syncthreads();
        if (wave == 0) {                                   // publish first: the other workgroups wait for this
            store_sc1(hx_dir + (long)(step & 1) * nk * 512 + (long)blk * 512 + lane * 8, *reinterpret_cast<const bf16x8*>(tile + lane * 8));
            wait_vm0();
            if (lane == 0) __hip_atomic_fetch_add(my_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (epi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = fq * 4 + r;
                if (b >= B) continue;
                ((bf16*)p.y)[(uint32_t)((b * T + t) * 2 * H + dir * H + j)] = (bf16)hp[r];
                if (p.saved) {
                    const uint32_t o = (uint32_t)((((dir * T + t) * B + b) * 5) * H + j);             // r, z, n, hn, hprev
                    const uint32_t Hu = (uint32_t)H;
                    p.saved[o] = rg[r]; p.saved[o + Hu] = zg[r]; p.saved[o + 2 * Hu] = ng[r]; p.saved[o + 3 * Hu] = hn[r]; p.saved[o + 4 * Hu] = hold[r];
                    ((bf16*)p.hprev_bf)[(uint32_t)(((dir * T + t) * B + b) * H + j)] = (bf16)hold[r];
                }
            }
            if (step + 1 < T) load_gi(step + 1, gic);      // arrives while the next step waits for the other workgroups
        }
    }
}

// ------------------------------------------------------------------ GRU backward step
__global__ __launch_bounds__(256) void gru_bwd_step_kernel(const stes_gru_args p, int step) {
    __shared__ float red[4][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int dir = blockIdx.y, j0 = blockIdx.x * 16, b0 = blockIdx.z * 16;
    const int H = p.H, T = p.T, B = p.B;
    // backward walks the forward order in reverse: forward step index s = T-1-step
    const int s = T - 1 - step;
    const int t = dir == 0 ? s : T - 1 - s;
    const int tnext = dir == 0 ? t + 1 : t - 1;          // time whose gate gradients were produced by the previous bwd step
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bf16* dgh_all = (bf16*)p.dgh;                         // [2][T][B][3H]
    if (step > 0) {
        const bf16* __restrict__ whhT = (const bf16*)p.whhT + (long)dir * 3 * H * H;   // [H][3H]
        const int K = 3 * H, nk = K / 32;
        const int brow = b0 + fr;
        const bf16* arow = (brow < B) ? dgh_all + (((long)dir * T + tnext) * B + brow) * K : nullptr;
        // loads of nine k-steps in flight together, three independent accumulator chains (see the forward step)
        constexpr int U = 9;
        f32x4 part[3] = {acc, acc, acc};
        for (int ks0 = wave; ks0 < nk; ks0 += 4 * U) {
            bf16x8 fa[U], fb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ks = ks0 + 4 * u;
                const int k = ks * 32 + fq * 8;
                const bool on = ks < nk;
                fa[u] = (on && arow) ? *reinterpret_cast<const bf16x8*>(arow + k) : zero8();
                fb[u] = on ? *reinterpret_cast<const bf16x8*>(whhT + (long)(j0 + fr) * K + k) : zero8();
            }
#pragma unroll
            for (int u = 0; u < U; ++u) part[u % 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[u], fb[u], part[u % 3], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = part[0][r] + part[1][r] + part[2][r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc[r];
    __syncthreads();
    if (wave != 0) return;
    const int j = j0 + fr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = b0 + fq * 4 + r;
        if (b >= B) continue;
        float dh = red[0][lane][r] + red[1][lane][r] + red[2][lane][r] + red[3][lane][r];
        if (step > 0) dh += p.dhstate[(((long)dir * 2 + ((step - 1) & 1)) * B + b) * H + j];
        float dy = p.dy[((long)b * T + t) * p.ld_dy + dir * H + j];
        if (p.mask) dy *= (float)((const bf16*)p.mask)[((long)b * T + t) * 2 * H + dir * H + j];
        dh += dy;
        const float* sv = p.saved + ((((long)dir * T + t) * B + b) * 5) * H;
        const float rg = sv[j], zg = sv[H + j], ng = sv[2 * H + j], hn = sv[3 * H + j], hp = sv[4 * H + j];
        const GruBwdOut go = gru_gate_bwd(dh, rg, zg, ng, hn, hp);
        p.dhstate[(((long)dir * 2 + (step & 1)) * B + b) * H + j] = go.dh_next;
        bf16* dgi = (bf16*)p.dgi + (((long)b * T + t) * 2 + dir) * 3 * H;
        dgi[j] = (bf16)go.dr; dgi[H + j] = (bf16)go.dz; dgi[2 * H + j] = (bf16)go.dn;
        bf16* dg = dgh_all + (((long)dir * T + t) * B + b) * 3 * H;
        dg[j] = (bf16)go.dr; dg[H + j] = (bf16)go.dz; dg[2 * H + j] = (bf16)go.dnr;
    }
}

// ------------------------------------------------------------------ GRU backward, the whole recurrence in ONE launch
// The counterpart of gru_fwd_persist_kernel: a workgroup owns 32 hidden units of one direction, its W_hh^T rows (K = 3H)
// stay in registers, dh stays in registers, and what travels between workgroups per step is the gate-gradient vector
// (dr, dz, dn * r as bf16: three 16 x 32 tiles per workgroup, the A-operand tiles of k-steps g * H/32 + blk), by the same
// write-through store / counter / sc1-load hand-off.  k-step assignment and summation order are gru_bwd_step_kernel's.
constexpr int GRU_UB = 18;                                     // k-steps per wave: 3H / 32 / 4, H <= 768
__global__ __launch_bounds__(512) void gru_bwd_persist_kernel(const stes_gru_args p, unsigned* __restrict__ cnt, bf16* __restrict__ dx,
                                                              int spin_limit, unsigned* err) {
    __shared__ float red[8][64][4];
    __shared__ __attribute__((aligned(16))) bf16 tile[3][16 * 32];
    __shared__ int failed;
    if (threadIdx.x == 0) failed = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int uh = wave >> 2, kq = wave & 3;
    const int blk = blockIdx.x, dir = blockIdx.y;
    const int H = p.H, T = p.T, B = p.B, K = 3 * H, nk = K / 32, nkh = H / 32;
    const int j = blk * 32 + uh * 16 + fr;
    const bf16* __restrict__ whhT = (const bf16*)p.whhT + (long)dir * 3 * H * H;   // [H][3H]
    bf16x8 fb[GRU_UB];
#pragma unroll
    for (int v = 0; v < GRU_UB; ++v) {
        const int ks = kq + 4 * v;
        fb[v] = ks < nk ? *reinterpret_cast<const bf16x8*>(whhT + (long)j * K + ks * 32 + fq * 8) : zero8();
    }
    const bool epi = kq == 0;
    unsigned* my_cnt = cnt + dir * 32;
    bf16* dx_dir = dx + (long)dir * 2 * nk * 512;
    bf16* dgh_all = (bf16*)p.dgh;                         // [2][T][B][3H]
    float dhp[4] = {0.f, 0.f, 0.f, 0.f};
    // what the gate math of a step reads, fetched a step ahead: r, z, n, hn, hprev, dy (masked)
    float cur[4][6];
    auto prefetch = [&](int step, float (&dst)[4][6]) {
        const int s = T - 1 - step;
        const int t = dir == 0 ? s : T - 1 - s;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = fq * 4 + r < B ? fq * 4 + r : 0;
            const uint32_t o = (uint32_t)((((dir * T + t) * B + b) * 5) * H + j), Hu = (uint32_t)H;   // (element counts < 2^31: checked by the host)
            dst[r][0] = p.saved[o]; dst[r][1] = p.saved[o + Hu]; dst[r][2] = p.saved[o + 2 * Hu]; dst[r][3] = p.saved[o + 3 * Hu]; dst[r][4] = p.saved[o + 4 * Hu];
            float dy = p.dy[(uint32_t)((b * T + t) * (int)p.ld_dy + dir * H + j)];
            if (p.mask) dy *= (float)((const bf16*)p.mask)[(uint32_t)((b * T + t) * 2 * H + dir * H + j)];
            dst[r][5] = dy;
        }
    };
    if (epi) prefetch(0, cur);
    for (int step = 0; step < T; ++step) {
        const int s = T - 1 - step;
        const int t = dir == 0 ? s : T - 1 - s;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
            if (threadIdx.x == 0 && !gru_wait(my_cnt, (unsigned)gridDim.x * (unsigned)step, spin_limit, err)) failed = 1;
            __syncthreads();
            const bf16* src = dx_dir + (long)((step - 1) & 1) * nk * 512 + fr * 32 + fq * 8;
            f32x4 part[3] = {acc, acc, acc};
#pragma unroll
            for (int half_ = 0; half_ < 2; ++half_) {      // two batches of nine tiles: all eighteen at once spill
                bf16x8 fa[GRU_UB / 2];
#pragma unroll
                for (int w = 0; w < GRU_UB / 2; ++w) {
                    const int ks = kq + 4 * (half_ * (GRU_UB / 2) + w);
                    fa[w] = load_sc1(src + (long)(ks < nk ? ks : 0) * 512);
                }
                wait_vm0();
#pragma unroll
                for (int w = 0; w < GRU_UB / 2; ++w) {
                    const int v = half_ * (GRU_UB / 2) + w;
                    if (kq + 4 * v < nk) part[v % 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[w], fb[v], part[v % 3], 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = part[0][r] + part[1][r] + part[2][r];
            if (failed) acc[0] = acc[1] = acc[2] = acc[3] = __builtin_nanf("");
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc[r];
        __syncthreads();
        float drp[4], dzp[4], dnp[4], dnr[4];
        if (epi) {
            const int w0 = uh * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float dh = red[w0][lane][r] + red[w0 + 1][lane][r] + red[w0 + 2][lane][r] + red[w0 + 3][lane][r];
                if (step > 0) dh += dhp[r];
                dh += cur[r][5];
                const GruBwdOut go = gru_gate_bwd(dh, cur[r][0], cur[r][1], cur[r][2], cur[r][3], cur[r][4]);
                dnp[r] = go.dn; drp[r] = go.dr; dzp[r] = go.dz; dnr[r] = go.dnr;
                dhp[r] = go.dh_next;
                const bool on = fq * 4 + r < B;
                const int o = (fq * 4 + r) * 32 + uh * 16 + fr;
                tile[0][o] = on ? (bf16)drp[r] : (bf16)0.f;
                tile[1][o] = on ? (bf16)dzp[r] : (bf16)0.f;
                tile[2][o] = on ? (bf16)dnr[r] : (bf16)0.f;
            }
        }
        __syncthreads();
        if (wave == 0) {                                   // publish first: the other workgroups wait for this
            bf16* dst = dx_dir + (long)(step & 1) * nk * 512 + lane * 8;
#pragma unroll
            for (int g = 0; g < 3; ++g) store_sc1(dst + (long)(g * nkh + blk) * 512, *reinterpret_cast<const bf16x8*>(&tile[g][lane * 8]));
            wait_vm0();
            if (lane == 0) __hip_atomic_fetch_add(my_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (epi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = fq * 4 + r;
                if (b >= B) continue;
                const uint32_t Hu = (uint32_t)H;
                const uint32_t oi = (uint32_t)((((b * T + t) * 2 + dir) * 3) * H + j), og = (uint32_t)((((dir * T + t) * B + b) * 3) * H + j);
                bf16* dgi = (bf16*)p.dgi;
                dgi[oi] = (bf16)drp[r]; dgi[oi + Hu] = (bf16)dzp[r]; dgi[oi + 2 * Hu] = (bf16)dnp[r];
                dgh_all[og] = (bf16)drp[r]; dgh_all[og + Hu] = (bf16)dzp[r]; dgh_all[og + 2 * Hu] = (bf16)dnr[r];
            }
            if (step + 1 < T) prefetch(step + 1, cur);     // arrives while the next step waits for the other workgroups
        }
    }
}

// ------------------------------------------------------------------ minGRU scan (reference model/min_gru.py:76-92)
// pre [B,T,2H] bf16: columns [0,H) gate pre-activation, [H,2H) candidate; h_t = (1-z) h_{t-1} + z * cand; the initial
// state h0 [H] is the reference's frozen per-layer parameter (model/min_gru.py:42,60), zeros unless a checkpoint says otherwise
__global__ void mingru_scan_fwd_kernel(const bf16* __restrict__ pre, const float* __restrict__ h0, bf16* __restrict__ h, int B, int T,
                                       int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    float hp = h0 ? h0[j] : 0.f;
    for (int t = 0; t < T; ++t) {
        const bf16* r = pre + ((long)b * T + t) * 2 * H;
        const float z = sigmoidf_((float)r[j]);
        hp = (1.f - z) * hp + z * (float)r[H + j];
        h[((long)b * T + t) * H + j] = (bf16)hp;
    }
}
// dpre [B,T,2H] bf16 from dh [B,T,ld] f32; hidden states re-read from h (bf16)
__global__ void mingru_scan_bwd_kernel(const bf16* __restrict__ pre, const float* __restrict__ h0, const bf16* __restrict__ h,
                                       const float* __restrict__ dh, long ld_dh, bf16* __restrict__ dpre, int B, int T, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    float carry = 0.f;
    for (int t = T - 1; t >= 0; --t) {
        const bf16* r = pre + ((long)b * T + t) * 2 * H;
        const float z = sigmoidf_((float)r[j]), cand = (float)r[H + j];
        const float hp = t > 0 ? (float)h[((long)b * T + t - 1) * H + j] : (h0 ? h0[j] : 0.f);
        const float g = dh[((long)b * T + t) * ld_dh + j] + carry;
        bf16* o = dpre + ((long)b * T + t) * 2 * H;
        o[j] = (bf16)(g * (cand - hp) * z * (1.f - z));
        o[H + j] = (bf16)(g * z);
        carry = g * (1.f - z);
    }
}

// ------------------------------------------------------------------ dropout
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// out = x * mask, mask = keep ? 1/(1-p) : 0 (mask stored as bf16 for the backward)
__global__ void dropout_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, bf16* __restrict__ mask, long n, float p,
                               uint32_t seed) {
    const float inv = 1.f / (1.f - p);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const uint32_t r = mix32((uint32_t)i * 0x9E3779B9U + seed) ^ mix32((uint32_t)(i >> 32) + seed * 0x85ebca6bU);
        const float u = (r >> 8) * (1.0f / 16777216.0f);
        const float m = u >= p ? inv : 0.f;
        mask[i] = (bf16)m;
        out[i] = (bf16)((float)x[i] * m);
    }
}

// ------------------------------------------------------------------ loss (reference train_e2e_spatial.py:566-593)
// single workgroup; rows = B*T frames
__global__ __launch_bounds__(1024) void loss_kernel(const float* __restrict__ logits, long ldl, const float* __restrict__ loc, long ldloc,
                                                    const int64_t* __restrict__ label, const float* __restrict__ xy, int M, int C,
                                                    float fg_weight, float grad_scale, const float* ext_denoms, float* out_loss,
                                                    bf16* __restrict__ dlogits, bf16* __restrict__ dloc) {
    __shared__ float red[4][32];
    __shared__ float tot[4];
    float wsum = 0.f, msum = 0.f, ce = 0.f, l1 = 0.f;
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        const int yv = (int)label[m];
        const float w = yv == 0 ? 1.f : fg_weight;
        const float* lg = logits + (long)m * ldl;
        float mx = lg[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, lg[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf(lg[c] - mx);
        ce += w * (logf(se) + mx - lg[yv]);
        wsum += w;
        if (loc && yv != 0) {
            msum += 1.f;
            for (int d = 0; d < 2; ++d) l1 += fabsf(sigmoidf_(loc[(long)m * ldloc + d]) - xy[(long)m * 2 + d]);
        }
    }
    float v[4] = {wsum, msum, ce, l1};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = wave_sum(v[k]); if (lane == 0) red[k][wave] = v[k]; }
    __syncthreads();
    if (threadIdx.x < 4) {
        float s = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[threadIdx.x][w];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    // denominators: local, or supplied (sum over all data-parallel ranks)
    const float wden = ext_denoms ? ext_denoms[0] : tot[0];
    const float mden = (ext_denoms ? ext_denoms[1] : tot[1]) + 1e-6f;
    if (threadIdx.x == 0) {
        const float lc = tot[2] / wden, ll = loc ? tot[3] / mden : 0.f;
        out_loss[0] = lc + ll; out_loss[1] = lc; out_loss[2] = ll; out_loss[3] = tot[0]; out_loss[4] = tot[1];
    }
    if (!dlogits) return;
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        const int yv = (int)label[m];
        const float w = (yv == 0 ? 1.f : fg_weight) / wden * grad_scale;
        const float* lg = logits + (long)m * ldl;
        float mx = lg[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, lg[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf(lg[c] - mx);
        const float inv = 1.f / se;
        const int ldd = (C + 7) / 8 * 8;
        for (int c = 0; c < ldd; ++c) {
            float g = 0.f;
            if (c < C) g = w * (__expf(lg[c] - mx) * inv - (c == yv ? 1.f : 0.f));
            dlogits[(long)m * ldd + c] = (bf16)g;
        }
        if (dloc) {
            for (int d = 0; d < 8; ++d) {
                float g = 0.f;
                if (d < 2 && yv != 0) {
                    const float sg = sigmoidf_(loc[(long)m * ldloc + d]);
                    const float df = sg - xy[(long)m * 2 + d];
                    g = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * sg * (1.f - sg) / mden * grad_scale;
                }
                dloc[(long)m * 8 + d] = (bf16)g;
            }
        }
    }
}

// ------------------------------------------------------------------ AdamW (torch.optim.AdamW, used at model/common.py:40)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                             float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt, float grad_mul) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gr = g[i] * grad_mul;
        float pv = p[i] * (1.f - lr * wd);
        const float mv = beta1 * m[i] + (1.f - beta1) * gr;
        const float vv = beta2 * v[i] + (1.f - beta2) * gr * gr;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pv -= (lr / bc1) * mv / denom;
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}

// ------------------------------------------------------------------ fp32 -> bf16 weight copies
// dst[r][c] = src[r][c] (transpose = 0) or dst[c][r] = src[r][c] (transpose = 1); dst rows padded with zeros to ld
// perm_fold > 0: the first perm_fold source COLUMNS are read through the gated shift's channel interleave, i.e. destination
// column c takes source column dest(c) (reference model/shift.py:124-128), so the consumer can keep source channel order
__device__ __forceinline__ int gsm_dest(int c, int fold) {
    const int half = fold >> 1, quarter = fold >> 2;
    const int g = c / half, rem = c - g * half;
    const int i = rem / quarter, j = rem - i * quarter;
    return g * half + 2 * j + i;
}
__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int rows, int cols, int ld, int dst_rows, int transpose,
                                 int perm_fold) {
    const long total = (long)dst_rows * ld;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int dr = (int)(i / ld), dc = (int)(i - (long)dr * ld);
        float v = 0.f;
        if (!transpose) {
            if (dr < rows && dc < cols) v = src[(long)dr * cols + (dc < perm_fold ? gsm_dest(dc, perm_fold) : dc)];
        } else {
            if (dc < rows && dr < cols) v = src[(long)dc * cols + (dr < perm_fold ? gsm_dest(dr, perm_fold) : dr)];
        }
        dst[i] = (bf16)v;
    }
}

// All weight copies of a step in one launch: workgroup -> (table entry, STES_CAST_CHUNK-element chunk) by binary search over
// the entries' first workgroup.  ~140 launches of a few microseconds each otherwise sit at the head of every training step.
// (A workgroup per 256 elements made 120,600 workgroups of the configuration's 16 M parameters: bound by dispatch, 133 us.)
__global__ __launch_bounds__(256) void cast_bf16_multi_kernel(const stes_cast_desc* __restrict__ table, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const stes_cast_desc d = table[lo];
    const long total = (long)d.dst_rows * d.ld;
    if (d.transpose) {
        // Transposed copies go through LDS in tiles of 32 destination rows x 64 destination columns, so that the source is read in
        // 128-byte runs (element-wise, a wave read 64 source rows for 4 bytes each: 28x the bytes of the whole table).  The entry's
        // workgroups share its tiles round-robin.
        __shared__ float tile[64][33];
        const int nb = (int)((total + STES_CAST_CHUNK - 1) / STES_CAST_CHUNK);
        const int tr = (d.dst_rows + 31) >> 5, tc = (d.ld + 63) >> 6;
        for (int t = (int)blockIdx.x - d.block0; t < tr * tc; t += nb) {
            const int r0 = (t / tc) << 5, c0 = (t - (t / tc) * tc) << 6;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = k * 256 + threadIdx.x, sr = idx >> 5, sc = idx & 31;
                const int dcol = c0 + sr, drow = r0 + sc;            // source row = destination column
                float v = 0.f;
                if (dcol < d.rows && drow < d.cols) v = d.src[(long)dcol * d.cols + (drow < d.perm_fold ? gsm_dest(drow, d.perm_fold) : drow)];
                tile[sr][sc] = v;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int idx = k * 256 + threadIdx.x, dr = idx >> 6, dc = idx & 63;
                if (r0 + dr < d.dst_rows && c0 + dc < d.ld) reinterpret_cast<bf16*>(d.dst)[(long)(r0 + dr) * d.ld + c0 + dc] = (bf16)tile[dc][dr];
            }
            __syncthreads();
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < STES_CAST_CHUNK / 256; ++j) {
        const long i = (long)(blockIdx.x - d.block0) * STES_CAST_CHUNK + j * 256 + threadIdx.x;
        if (i >= total) return;
        const int dr = (int)(i / d.ld), dc = (int)(i - (long)dr * d.ld);
        float v = 0.f;
        if (!d.transpose) {
            if (dr < d.rows && dc < d.cols) v = d.src[(long)dr * d.cols + (dc < d.perm_fold ? gsm_dest(dc, d.perm_fold) : dc)];
        } else {
            if (dc < d.rows && dr < d.cols) v = d.src[(long)dc * d.cols + (dr < d.perm_fold ? gsm_dest(dr, d.perm_fold) : dr)];
        }
        reinterpret_cast<bf16*>(d.dst)[i] = (bf16)v;
    }
}

}  // namespace

#define S(s) ((hipStream_t)(s))

static int gru_check(const stes_gru_args* a) {
    STES_CHECK(a && a->gi && a->whh && a->bhh && a->y && a->hstate, STES_ERR_ARG, "stes_gru: null pointer");
    STES_CHECK(a->H % 32 == 0 && a->B > 0 && a->T > 0, STES_ERR_ARG, "stes_gru: H must be a multiple of 32");
    return STES_OK;
}

// floats of workspace the one-launch recurrence needs at hstate (forward) / dhstate (backward): a 256-byte counter block and
// the exchange tiles [2 dirs][2 parities][k-steps][16][32] bf16
static int64_t gru_persist_floats(int H, int backward) { return 64 + (int64_t)(backward ? 3 : 1) * 4 * (H / 32) * 256; }

extern "C" int64_t stes_gru_ws_floats(int B, int H, int backward) {
    const int64_t step = 4L * B * H, one = (B <= 16 && H <= 32 * 4 * GRU_U) ? gru_persist_floats(H, backward) : 0;
    return step > one ? step : one;
}

// Time-out reporting of the one-launch recurrences (gru_wait): one word of pinned host memory that every GPU of the process
// can write.  It is created by the first stes_gru_forward / stes_gru_backward call (not by the entry point that allocates
// nothing, which uses it only once it exists) and read, without any synchronisation, at the start of every stes_gru_* call:
// a raised word means an EARLIER recurrence launch of this process timed out and poisoned its outputs with NaN.
static unsigned* g_gru_err = nullptr;
static std::mutex g_gru_err_mu;
static int g_gru_spin_limit = 1 << 20;

static unsigned* gru_err_word(bool create) {
    std::lock_guard<std::mutex> lock(g_gru_err_mu);
    if (!g_gru_err && create) {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && p) {
            g_gru_err = reinterpret_cast<unsigned*>(p);
            *g_gru_err = 0u;
        }
    }
    return g_gru_err;
}

static int gru_check_earlier_timeout(bool create) {
    unsigned* w = gru_err_word(create);
    if (w && __atomic_load_n(w, __ATOMIC_RELAXED) != 0u) {
        __atomic_store_n(w, 0u, __ATOMIC_RELAXED);
        return stes_set_error(STES_ERR_LAUNCH, "stes_gru: an earlier one-launch recurrence timed out waiting for its peer workgroups; "
                                               "its outputs are NaN (the GPU was shared with work that kept them from running together)");
    }
    return STES_OK;
}

extern "C" int stes_gru_set_spin_limit(int limit) {
    STES_CHECK(limit >= 0, STES_ERR_ARG, "stes_gru_set_spin_limit: negative limit");
    g_gru_spin_limit = limit;
    return STES_OK;
}

// the forward recurrence on the caller's stream: one launch when the workspace allows it, else one launch per step
static int gru_forward_launch(const stes_gru_args& a, int64_t ws_floats, hipStream_t stream) {
    if (a.B <= 16 && a.H <= 32 * 4 * GRU_U && ws_floats >= gru_persist_floats(a.H, 0) && 2L * a.T * a.B * 6 * a.H < (1L << 31)) {
        unsigned* cnt = reinterpret_cast<unsigned*>(a.hstate);
        if (hipMemsetAsync(cnt, 0, 256, stream) != hipSuccess) return stes_set_error(STES_ERR_LAUNCH, "stes_gru_forward: cannot clear the step counters");
        hipLaunchKernelGGL(gru_fwd_persist_kernel, dim3(a.H / 32, 2), dim3(512), 0, stream, a, cnt, reinterpret_cast<bf16*>(a.hstate + 64),
                           g_gru_spin_limit, gru_err_word(false));
    } else {
        const dim3 grid(a.H / 16, 2, stes_div_up(a.B, 16));
        for (int s = 0; s < a.T; ++s) hipLaunchKernelGGL(gru_fwd_step_kernel, grid, dim3(256), 0, stream, a, s);
    }
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_gru_forward(const stes_gru_args* a, void* stream) {
    int rc = gru_check(a); if (rc) return rc;
    STES_CHECK(!a->saved || a->hprev_bf, STES_ERR_ARG, "stes_gru_forward: saved needs hprev_bf");
    rc = gru_check_earlier_timeout(true); if (rc) return rc;
    return gru_forward_launch(*a, a->ws_floats, S(stream));
}

// ------------------------------------------------------------------ GRU forward for callers that cannot allocate
// The fp32 hidden-state ping-pong buffer ([2 dirs][2][B][H], the `hstate` workspace of stes_gru_forward) lives in this
// file-scope device array instead of caller memory.  Every GPU holds its own copy at its own address: the address is
// resolved once per device with hipGetSymbolAddress and only that pointer is ever handed to a kernel.
constexpr long GRU_SCRATCH_FLOATS = 1L << 18;                 // 1 MiB per GPU: 4*B*H floats, i.e. B*H <= 65,536
static __device__ __align__(256) float scratch[GRU_SCRATCH_FLOATS];
constexpr int MAX_DEVICES = 64;
static float* g_scratch_ptr[MAX_DEVICES];                      // per device, resolved on the first call there
static std::mutex g_scratch_mu;

// this device's copy of the array (resolved once per device)
static int gru_scratch_of_device(float** out) {
    int dev = -1;
    STES_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES, STES_ERR_UNSUPPORTED,
               "stes_gru_forward_noalloc: cannot identify the current device");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* sym = nullptr;
        const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !sym) return stes_set_error(STES_ERR_LAUNCH, hipGetErrorString(e));
        g_scratch_ptr[dev] = (float*)sym;
    }
    *out = g_scratch_ptr[dev];
    return STES_OK;
}

extern "C" int stes_gru_noalloc_prepare(void) {
    float* ptr = nullptr;
    return gru_scratch_of_device(&ptr);
}

extern "C" int stes_gru_forward_noalloc(const stes_gru_args* a, void* stream) {
    STES_CHECK(a && a->gi && a->whh && a->bhh && a->y, STES_ERR_ARG, "stes_gru_forward_noalloc: null pointer");
    STES_CHECK(a->H > 0 && a->H % 32 == 0 && a->B > 0 && a->T > 0, STES_ERR_ARG, "stes_gru_forward_noalloc: H must be a positive multiple of 32");
    STES_CHECK(!a->saved || a->hprev_bf, STES_ERR_ARG, "stes_gru_forward_noalloc: saved needs hprev_bf");
    STES_CHECK(4L * a->B * a->H <= GRU_SCRATCH_FLOATS, STES_ERR_UNSUPPORTED,
               "stes_gru_forward_noalloc: B*H exceeds the static scratch (B*H <= 65536); use stes_gru_forward with a caller workspace");
    float* ptr = nullptr;
    {
        const int rc0 = gru_scratch_of_device(&ptr);
        if (rc0) return rc0;
    }
    // what must start at zero (the step counters of the one-launch recurrence) is cleared by a memset node on the stream;
    // the step launches need nothing cleared: step 0 takes h = 0 without reading the buffer, step s reads what step s-1 wrote
    const int rc = gru_check_earlier_timeout(false);
    if (rc) return rc;
    stes_gru_args b = *a;
    b.hstate = ptr;
    return gru_forward_launch(b, GRU_SCRATCH_FLOATS, S(stream));
}

extern "C" int stes_gru_backward(const stes_gru_args* a, void* stream) {
    int rc = gru_check(a); if (rc) return rc;
    STES_CHECK(a->saved && a->whhT && a->dy && a->dgi && a->dgh && a->dhstate && a->ld_dy >= 2 * a->H, STES_ERR_ARG, "stes_gru_backward: null pointer");
    rc = gru_check_earlier_timeout(true); if (rc) return rc;
    if (a->B <= 16 && a->H <= 32 * 4 * GRU_U && a->ws_floats >= gru_persist_floats(a->H, 1) && 2L * a->T * a->B * 6 * a->H < (1L << 31) &&
        (long)a->B * a->T * a->ld_dy < (1L << 31)) {
        unsigned* cnt = reinterpret_cast<unsigned*>(a->dhstate);
        STES_CHECK(hipMemsetAsync(cnt, 0, 256, S(stream)) == hipSuccess, STES_ERR_LAUNCH, "stes_gru_backward: cannot clear the step counters");
        hipLaunchKernelGGL(gru_bwd_persist_kernel, dim3(a->H / 32, 2), dim3(512), 0, S(stream), *a, cnt, reinterpret_cast<bf16*>(a->dhstate + 64),
                           g_gru_spin_limit, gru_err_word(false));
    } else {
        const dim3 grid(a->H / 16, 2, stes_div_up(a->B, 16));
        for (int s = 0; s < a->T; ++s) hipLaunchKernelGGL(gru_bwd_step_kernel, grid, dim3(256), 0, S(stream), *a, s);
    }
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_mingru_scan_fwd(const uint16_t* pre, const float* h0, uint16_t* h, int B, int T, int H, void* stream) {
    STES_CHECK(pre && h && B > 0 && T > 0 && H > 0, STES_ERR_ARG, "stes_mingru_scan_fwd: bad argument");
    hipLaunchKernelGGL(mingru_scan_fwd_kernel, dim3(stes_div_up((long)B * H, 128)), dim3(128), 0, S(stream), (const bf16*)pre, h0, (bf16*)h, B, T, H);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_mingru_scan_bwd(const uint16_t* pre, const float* h0, const uint16_t* h, const float* dh, int64_t ld_dh, uint16_t* dpre,
                                    int B, int T, int H, void* stream) {
    STES_CHECK(pre && h && dh && dpre && ld_dh >= H, STES_ERR_ARG, "stes_mingru_scan_bwd: bad argument");
    hipLaunchKernelGGL(mingru_scan_bwd_kernel, dim3(stes_div_up((long)B * H, 128)), dim3(128), 0, S(stream), (const bf16*)pre, h0, (const bf16*)h,
                       dh, (long)ld_dh, (bf16*)dpre, B, T, H);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_dropout(const uint16_t* x, uint16_t* out, uint16_t* mask, int64_t n, float p, uint32_t seed, void* stream) {
    STES_CHECK(x && out && mask && n > 0 && p >= 0.f && p < 1.f, STES_ERR_ARG, "stes_dropout: bad argument");
    long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(dropout_kernel, dim3((int)blocks), dim3(256), 0, S(stream), (const bf16*)x, (bf16*)out, (bf16*)mask, (long)n, p, seed);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_loss(const float* logits, int64_t ldl, const float* loc, int64_t ldloc, const int64_t* label, const float* xy,
                         int M, int C, float fg_weight, float grad_scale, const float* ext_denoms, float* out_loss,
                         uint16_t* dlogits, uint16_t* dloc, void* stream) {
    STES_CHECK(logits && label && out_loss && M > 0 && C > 0 && ldl >= C, STES_ERR_ARG, "stes_loss: bad argument");
    STES_CHECK(!loc || (xy && ldloc >= 2), STES_ERR_ARG, "stes_loss: location targets");
    hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(1024), 0, S(stream), logits, (long)ldl, loc, (long)ldloc, label, xy, M, C, fg_weight,
                       grad_scale, ext_denoms, out_loss, (bf16*)dlogits, (bf16*)dloc);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int step, float grad_mul, void* stream) {
    STES_CHECK(p && g && m && v && n > 0 && step >= 1, STES_ERR_ARG, "stes_adamw: bad argument");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, S(stream), p, g, m, v, (long)n, lr, beta1, beta2, eps, weight_decay,
                       (float)bc1, (float)sqrt(bc2), grad_mul);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_cast_bf16(const float* src, uint16_t* dst, int rows, int cols, int ld, int dst_rows, int transpose, int perm_fold,
                              void* stream) {
    STES_CHECK(src && dst && rows > 0 && cols > 0, STES_ERR_ARG, "stes_cast_bf16: bad argument");
    STES_CHECK(transpose ? (ld >= rows && dst_rows >= cols) : (ld >= cols && dst_rows >= rows), STES_ERR_ARG, "stes_cast_bf16: destination too small");
    long blocks = ((long)dst_rows * ld + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((int)blocks), dim3(256), 0, S(stream), src, (bf16*)dst, rows, cols, ld, dst_rows, transpose, perm_fold);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

extern "C" int stes_cast_bf16_multi(const stes_cast_desc* table, int n, int total_blocks, void* stream) {
    STES_CHECK(table && n > 0 && total_blocks > 0, STES_ERR_ARG, "stes_cast_bf16_multi: bad argument");
    hipLaunchKernelGGL(cast_bf16_multi_kernel, dim3(total_blocks), dim3(256), 0, S(stream), table, n);
    STES_LAUNCH_CHECK();
    return STES_OK;
}
