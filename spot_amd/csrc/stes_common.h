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
