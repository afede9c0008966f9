// Shared device helpers for the STES gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/stes.h"

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define STES_WAVE 64

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float f) { return (bf16)f; }

// 8 bf16 (16 bytes) <-> 8 floats
__device__ __forceinline__ void unpack8(const bf16x8& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
__device__ __forceinline__ bf16x8 pack8(const float (&f)[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
    return v;
}
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)0.f;
    return v;
}

__device__ __forceinline__ void load8f(const float* p, float (&f)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[i + 4] = b[i]; }
}
__device__ __forceinline__ void load8b(const bf16* p, float (&f)[8]) {
    unpack8(*reinterpret_cast<const bf16x8*>(p), f);
}
__device__ __forceinline__ void store8b(bf16* p, const float (&f)[8]) {
    *reinterpret_cast<bf16x8*>(p) = pack8(f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Error plumbing shared by every entry point (defined in stes_api.hip).
int stes_set_error(int code, const char* msg);
#define STES_CHECK(cond, code, msg) \
    do { if (!(cond)) return stes_set_error((code), (msg)); } while (0)
#define STES_LAUNCH_CHECK() \
    do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return stes_set_error(STES_ERR_LAUNCH, hipGetErrorString(e_)); } while (0)

static inline int stes_div_up(long a, long b) { return (int)((a + b - 1) / b); }

// Division of a 32-bit index by a run-time constant without the ~30-instruction integer divide (row -> image / pixel
// decoding sits in the inner loop of every NHWC kernel).  Granlund-Montgomery round-up form, exact for all 32-bit x.
struct FastDiv {
    uint32_t magic, shift, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) { f.magic = 0; f.shift = 0; return f; }
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    f.magic = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
    f.shift = s;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t x, const FastDiv& f) {
    if (f.d <= 1) return x;
    const uint32_t t = __umulhi(x, f.magic);
    return (t + ((x - t) >> 1)) >> (f.shift - 1);
}
__device__ __forceinline__ void fdivmod(uint32_t x, const FastDiv& f, uint32_t& q, uint32_t& r) {
    q = fdiv(x, f);
    r = x - q * f.d;
}
#define STES_MAX_ROWS 0x7fffffffL

// ---------------------------------------------------------------- deterministic mode (stes_set_deterministic)
// Every cross-workgroup fp32 accumulation of the library is a float atomic, whose result depends on the order in which the
// workgroups of a launch arrive.  In deterministic mode that order is fixed: the workgroups of a launch commit their sums
// one after the other in block-index order (a ticket word in device memory that the caller provides), and accumulations
// inside a workgroup that use LDS atomics run wave after wave.  The values added are the same as in the normal mode; only
// their order is pinned, so two runs of the same step give the same bits.  It is a test mode: commits are serialised.
struct StesDet {
    unsigned* ticket;     // nullptr: normal mode.  ticket[0]: turn counter, ticket[1]: raised when a wait ran out
    unsigned base;        // value of the turn counter at which block 0 of this launch commits
    unsigned pad;
};
// host: the ticket window of a launch of `nblocks` workgroups ({nullptr, 0} in normal mode); launches that use it must be
// issued in the order of these calls on one stream
StesDet stes_det_take(unsigned nblocks);

#ifdef __HIPCC__
__device__ __forceinline__ unsigned det_block_id() { return blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); }

// All threads of the workgroup call both.  Between them the workgroup issues its global atomics; det_leave makes sure they
// have been performed before the next workgroup's turn starts.
__device__ __forceinline__ void det_enter(const StesDet& d) {
    if (!d.ticket) return;
    if (threadIdx.x == 0) {
        const unsigned turn = d.base + det_block_id();
        int spins = 0;
        while (__hip_atomic_load(d.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != turn) {
            if (++spins > (1 << 24)) { __hip_atomic_fetch_or(d.ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void det_leave(const StesDet& d) {
    if (!d.ticket) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's atomics have been performed
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(d.ticket, d.base + det_block_id() + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// `stmt` (LDS float atomics of every thread): as issued in normal mode, wave after wave in deterministic mode
#define STES_DET_LDS(det, ...)                                                                       \
    do {                                                                                             \
        if (!(det).ticket) { __VA_ARGS__; }                                                               \
        else {                                                                                       \
            const int nw_ = ((int)blockDim.x + 63) >> 6, w_ = (int)threadIdx.x >> 6;                 \
            for (int i_ = 0; i_ < nw_; ++i_) { if (i_ == w_) { __VA_ARGS__; } __syncthreads(); }          \
        }                                                                                            \
    } while (0)
// `stmt` (global float atomics that every wave issues on the same addresses): as issued in normal mode; in deterministic
// mode wave after wave, each wave's atomics performed before the next wave starts
#define STES_DET_WAVES(det, ...)                                                                     \
    do {                                                                                             \
        if (!(det).ticket) { __VA_ARGS__; }                                                               \
        else {                                                                                       \
            const int nw_ = ((int)blockDim.x + 63) >> 6, w_ = (int)threadIdx.x >> 6;                 \
            for (int i_ = 0; i_ < nw_; ++i_) {                                                       \
                if (i_ == w_) { __VA_ARGS__; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }           \
                __syncthreads();                                                                     \
            }                                                                                        \
        }                                                                                            \
    } while (0)
#endif
