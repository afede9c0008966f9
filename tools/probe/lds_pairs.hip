// LDS bank behaviour of ds_read_b128 on one CU (debug aid): 16 waves of one workgroup issue the same pattern; cycles per
// wave-instruction at CU level from s_memtime.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/lds_pairs tools/probe/lds_pairs.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;

__global__ __launch_bounds__(1024) void pat(const int* lane_off, long long* out) {
    __shared__ __attribute__((aligned(16))) int smem[16384];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 1024) smem[i] = i;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)smem + (unsigned)lane_off[lane];
    int acc = 0;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < 128; ++it) {
        i32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(addr));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u][0];
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) { out[0] = t1 - t0; out[1] = acc; }
}

static long long* out;
static int* doff;
static double run(const std::vector<int>& off) {
    (void)hipMemcpy(doff, off.data(), 256, hipMemcpyHostToDevice);
    long long h[2];
    hipLaunchKernelGGL(pat, dim3(1), dim3(1024), 0, 0, doff, out);
    (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    return h[0] / (16.0 * 1024.0);
}

int main() {
    (void)hipMalloc(&out, 16);
    (void)hipMalloc(&doff, 256);
    std::vector<int> off(64);
    printf("cycles per wave-instruction (ds_read_b128, 16 waves on one CU)\n");
    printf("lane l at l * stride:\n");
    for (int st = 16; st <= 320; st += 16) {
        for (int l = 0; l < 64; ++l) off[l] = l * st;
        printf("  stride %4d: %6.2f\n", st, run(off));
    }
    printf("fragment pattern: lane (p = l & 15, q = l >> 4) at p * stride + q * 16:\n");
    for (int st : {64, 80, 96, 112, 128, 144, 160, 176, 192, 272, 288}) {
        for (int l = 0; l < 64; ++l) off[l] = (l & 15) * st + (l >> 4) * 16;
        printf("  stride %4d: %6.2f\n", st, run(off));
    }
    printf("conv pattern: p * stride + (q & 1) * 16 + (q >> 1) * stride:\n");
    for (int st : {128, 144, 160, 176, 192, 208, 272, 288}) {
        for (int l = 0; l < 64; ++l) off[l] = (l & 15) * st + ((l >> 4) & 1) * 16 + (l >> 5) * st;
        printf("  stride %4d: %6.2f\n", st, run(off));
    }
    printf("n distinct lanes-worth: first n lanes linear (l * 16), the rest read lane 0's address:\n");
    for (int n : {1, 8, 16, 24, 32, 48, 64}) {
        for (int l = 0; l < 64; ++l) off[l] = l < n ? l * 16 : 0;
        printf("  n %2d: %6.2f\n", n, run(off));
    }
    printf("two quads loaded: lanes 0..n-1 at l * 256 (same banks, different rows), rest linear in another region:\n");
    for (int n : {1, 2, 4, 8, 16}) {
        for (int l = 0; l < 64; ++l) off[l] = l < n ? l * 256 : 16384 + l * 16;
        printf("  n %2d: %6.2f\n", n, run(off));
    }
    printf("which lanes share a pass: lanes a and b at 0 and 256 (same banks), rest broadcast address 4096:\n");
    for (int b : {1, 2, 4, 8, 15, 16, 31, 32, 33, 48, 63}) {
        for (int l = 0; l < 64; ++l) off[l] = 4096;
        off[0] = 0; off[b] = 256;
        printf("  b %2d: %6.2f\n", b, run(off));
    }
    printf("misaligned halves: lane l at l * 32 + 16 vs l * 32:\n");
    for (int sh : {0, 16}) {
        for (int l = 0; l < 64; ++l) off[l] = l * 32 + sh;
        printf("  shift %2d: %6.2f\n", sh, run(off));
    }
    return 0;
}
