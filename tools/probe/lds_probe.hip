// LDS read-rate probe (debug aid): time a workgroup-resident loop of ds_read_b128 / ds_read_b64 with several lane->address
// patterns.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/lds_probe tools/probe/lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

template <int W>   // bytes per lane per read: 16 or 8
__global__ __launch_bounds__(256) void probe(int pattern, int stride, int iters, int* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 256) reinterpret_cast<int*>(smem)[i] = i;
    __syncthreads();
    const int p = lane & 15, q = lane >> 4;
    int addr;
    if (pattern == 0) addr = lane * W;                                             // linear
    else if (pattern == 1) addr = p * stride + (q & 1) * 16 + (q >> 1) * stride;   // conv fragment: pixel stride, half, next tap
    else if (pattern == 2) addr = p * stride + q * 16;                             // gemm fragment: row stride, k chunk
    else addr = (lane >> 2) * stride + (lane & 3) * 16;                            // 4 lanes per row
    addr += wave * 32;
    int acc = 0;
    const unsigned base = (unsigned)(size_t)smem;        // LDS byte address of the dynamic segment
    for (int it = 0; it < iters; ++it) {
        i32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned a = base + ((addr + u * 2560) & 65535 & ~(W - 1));
            if (W == 16) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(a));
            else { i32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(a)); v[u][0] = t[0]; v[u][1] = t[1]; v[u][2] = 0; v[u][3] = 0; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u][0] ^ v[u][3];
    }
    if (acc == 0x12345678) out[0] = acc;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
// the 3x3 grouped convolution's fragment loop on a 14 x 14 image (window rows of 16 pixels x 160 bytes): 5 groups of
// 3 fragments, 15 reads then (mode 1) 15 MFMAs
__global__ __launch_bounds__(256, 2) void conv_emul(int mode, int iters, int* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 256) reinterpret_cast<int*>(smem)[i] = 0;
    __syncthreads();
    const int p = lane & 15, q = lane >> 4;
    const unsigned base = (unsigned)(size_t)smem;
    const int ROWB = 16 * 160;
    unsigned tapc[5];
    for (int s = 0; s < 5; ++s) {
        const int t0 = 2 * s, t1 = 2 * s + 1 < 9 ? 2 * s + 1 : 8;
        tapc[s] = (q < 2 ? (t0 / 3) * ROWB + (t0 % 3) * 160 : (t1 / 3) * ROWB + (t1 % 3) * 160) + wave * 32 + (q & 1) * 16;
    }
    unsigned fb[15];
    for (int f = 0; f < 15; ++f) {
        int pi = f * 16 + p; if (pi > 195) pi = 195;
        fb[f] = base + (pi / 14) * ROWB + (pi % 14) * 160;
    }
    bf16x8_t w[5];
    for (int s = 0; s < 5; ++s) for (int j = 0; j < 8; ++j) w[s][j] = (__bf16)(float)(lane + s + j);
    f32x4_t tot = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            bf16x8_t frag[5][3];
#pragma unroll
            for (int s = 0; s < 5; ++s)
#pragma unroll
                for (int u = 0; u < 3; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(frag[s][u]) : "v"(fb[g * 3 + u] + tapc[s]));
            f32x4_t acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#define KS(S, CNT) asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(frag[S][0]), "+v"(frag[S][1]), "+v"(frag[S][2])); \
            if (mode == 1) { _Pragma("unroll") for (int u = 0; u < 3; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[S], frag[S][u], acc[u], 0, 0, 0); } \
            else { _Pragma("unroll") for (int u = 0; u < 3; ++u) acc[u][0] += (float)frag[S][u][0]; } \
            __builtin_amdgcn_sched_barrier(0);
            KS(0, 12) KS(1, 9) KS(2, 6) KS(3, 3) KS(4, 0)
#undef KS
            tot += acc[0] + acc[1] + acc[2];
        }
    }
    if (tot[0] == 123.f) out[0] = 1;
}

int main() {
    int* out; hipMalloc(&out, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    struct { int w, pat, stride; const char* what; } cases[] = {
        {16, 0, 0, "b128 linear"}, {16, 1, 160, "b128 conv stride 160"}, {16, 1, 144, "b128 conv stride 144"}, {16, 1, 128, "b128 conv stride 128"},
        {16, 1, 272, "b128 conv stride 272"}, {16, 2, 160, "b128 gemm stride 160"}, {16, 2, 80, "b128 gemm stride 80"}, {16, 2, 96, "b128 gemm stride 96"},
        {16, 3, 64, "b128 4/row stride 64"}, {16, 3, 80, "b128 4/row stride 80"}, {8, 0, 0, "b64 linear"}, {8, 1, 160, "b64 conv stride 160"}};
    for (auto& c : cases) {
        for (int wgs : {256, 512}) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (c.w == 16) hipLaunchKernelGGL(probe<16>, dim3(wgs), dim3(256), 65536, 0, c.pat, c.stride, iters, out);
                else hipLaunchKernelGGL(probe<8>, dim3(wgs), dim3(256), 65536, 0, c.pat, c.stride, iters, out);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double bytes = (double)wgs * 256 * c.w * 8 * iters;
            const double per_cu_clk = bytes / 256 / (ms * 1e-3) / 2.4e9;
            printf("%-26s wgs/CU %d  %8.3f ms  %6.1f B/clk/CU (at 2.4 GHz)\n", c.what, wgs / 256, ms, per_cu_clk);
        }
    }
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_emul), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int mode = 0; mode < 2; ++mode)
        for (int wgs : {256, 512}) {
            float ms = 0;
            const int it2 = 200;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(conv_emul, dim3(wgs), dim3(256), 65536, 0, mode, it2, out);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double cyc = ms * 1e-3 * 2.4e9 / it2;
            printf("conv emul mode %d (1 = with MFMA) wgs/CU %d: %8.3f ms, %7.0f cycles per image (75 reads, 75 MFMA per wave), LDS %6.1f B/clk/CU\n",
                   mode, wgs / 256, ms, cyc, (double)(wgs / 256) * 4 * 75 * 1024 / cyc);
        }
    return 0;
}
