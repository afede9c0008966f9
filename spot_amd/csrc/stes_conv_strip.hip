// Grouped 3x3 convolution (group width 16), stride 1, forward, for feature maps whose rows fit one 16-lane strip (at most
// 14 columns: the 14 x 14 and 7 x 7 maps of stages 3 and 4): every input pixel is read once, from global memory straight
// into MFMA operand registers.  No LDS window, no im2col.
//
// Per group the product is taken tap by tap on v_mfma_f32_16x16x16_bf16 with the 16 output channels as rows and 16
// PIXELS OF ONE IMAGE ROW as columns:  P[tap][co, x] = sum_ci W[tap][co, ci] * in[row, x, ci].  The B operand of lane
// (x = lane & 15, q = lane >> 4) is the 4 channels 4q..4q+3 of pixel x: one 8-byte NHWC load.  A wave walks down a strip of
// 16 columns; a source row is multiplied by all nine taps while it is in registers:
//   * the three vertical taps feed three accumulators that belong to three different output rows (the accumulators rotate
//     through the MFMAs' C -> D operands, no copies);
//   * the three horizontal taps give three results that belong to output columns x-1, x, x+1: they are added after a DPP
//     row shift of one lane (the 16 lanes of a DPP row are exactly the 16 pixels).  The outermost lanes of a strip are halo
//     columns whose sums are incomplete and never stored.
// The result of lane (x, q) is 4 consecutive output channels of pixel x: an 8-byte store; the four waves of a workgroup take
// the four groups of a 64-channel block of the same strip, so together they read and write whole 128-byte lines.
// BatchNorm + ReLU of the producing layer is applied to the operand in registers; the statistics of the output (training)
// or the squeeze-excite squeeze (evaluation) are taken from the result registers.
// Measured (1,024 images, 14 x 14 x 320): 72 us against 90 us for the LDS-window kernel of stes_conv.hip, whose operand
// reads cost ten LDS reads per input byte.  The kernel is bound by instruction issue (about 100 instructions per 16-pixel
// row and group), not by memory: on wider maps, for the masked input gradient and for stride 2 the same scheme measured no
// faster than the window kernels (8-byte accesses in 32-byte segments also cap a pure copy at 3.1-3.5 TB/s,
// tools/probe/lane_pattern.hip), so those keep the window kernels.
#include <mutex>
#include "stes_common.h"
#include "stes_conv_strip.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float dpp_from_next(float v) {        // lane j <- lane j + 1 of the same 16-lane row (0 at the end)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_prev(float v) {        // lane j <- lane j - 1 (0 at the start)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {            // sum over the 16 lanes of a DPP row, in every lane
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// A operand (row m = lane & 15, k = 4 * (lane >> 4) + i) of one tap, read from the k-step-pair pack of stes_pack_conv3x3g:
// [G][5][64][8] with lane (m, q) of k-step s holding tap slot 2s + (q >> 1), inner channels (q & 1) * 8 + 0..7.
__device__ __forceinline__ s16x4 tap_frag(const bf16* __restrict__ wp, int g, int slot, int lane) {
    const int m = lane & 15, q = lane >> 4;
    const int packed_lane = ((slot & 1) * 2 + (q >> 1)) * 16 + m;
    return *reinterpret_cast<const s16x4*>(wp + (((long)g * 5 + (slot >> 1)) * 64 + packed_lane) * 8 + 4 * (q & 1));
}

__device__ __forceinline__ void unpack4(u32x2 v, float (&f)[4]) {
    f[0] = __builtin_bit_cast(float, v[0] << 16); f[1] = __builtin_bit_cast(float, v[0] & 0xffff0000u);
    f[2] = __builtin_bit_cast(float, v[1] << 16); f[3] = __builtin_bit_cast(float, v[1] & 0xffff0000u);
}
__device__ __forceinline__ u32x2 pack4(const float (&f)[4]) {
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (bf16)f[i];
    return __builtin_bit_cast(u32x2, o);
}

struct StripGeom {
    int N, Hs, Ws, C;             // images, image size, channels
    int n_ct;                     // column tiles (strips) per image
    int rows, chunk;              // rows of the flat row sequence of a channel block (N * n_ct * H), rows per workgroup
};

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

// What the result registers of one output row go through.
struct Epilogue {
    bool stats, pool;
    float sc[4], sh[4];           // pooling affine
    float ssum[4], ssq[4];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ssum[i] = 0.f; ssq[i] = 0.f; }
    }
    // o: the four channels of this lane's pixel; ok: the lane holds a pixel of the image that this workgroup owns
    __device__ __forceinline__ u32x2 finish(const float (&o)[4], bool ok) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        u32x2 ob;
        ob[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){o[0], o[1]}, bf16x2));
        ob[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){o[2], o[3]}, bf16x2));
        if (pool) {
            float v[4];
            unpack4(ob, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) ssum[i] += ok ? fmaxf(v[i] * sc[i] + sh[i], 0.f) : 0.f;
        } else if (stats) {
            const u32x2 sv = ok ? ob : u32x2{0u, 0u};
            float v[4];
            unpack4(sv, v);
#pragma unroll
            for (int i = 0; i < 4; ++i) { ssum[i] += v[i]; ssq[i] += v[i] * v[i]; }
        }
        return ob;
    }
    // the workgroup's statistics: one contiguous 64-channel atomic per statistic
    __device__ __forceinline__ void commit(float* s0, float* s1, int cb, float (*red)[64], const StesDet& det) {
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = row16_sum(ssum[i]), b = row16_sum(ssq[i]);
            if ((lane & 15) == 0) { red[0][wave * 16 + (lane >> 4) * 4 + i] = a; red[1][wave * 16 + (lane >> 4) * 4 + i] = b; }
        }
        __syncthreads();
        det_enter(det);
        if (tid < 128) atomicAdd((tid < 64 ? s0 : s1) + cb * 64 + (tid & 63), red[tid >> 6][tid & 63]);
        det_leave(det);
    }
    // the image's pooled values (a wave owns its 16 channels of the image: fixed order, no atomics)
    __device__ __forceinline__ void pool_out(bf16* __restrict__ dst, float norm) {
        const int lane = threadIdx.x & 63;
        bf16x4 pv;
#pragma unroll
        for (int i = 0; i < 4; ++i) { pv[i] = (bf16)(row16_sum(ssum[i]) * norm); ssum[i] = 0.f; }
        if ((lane & 15) == 0) *reinterpret_cast<bf16x4*>(dst) = pv;
    }
};

// ------------------------------------------------------------------ the kernel
// Lane j of a strip holds column 14 * tile - 1 + j: 14 output columns per strip.  The rows of all strips of a 64-channel
// block form one flat sequence ((image * tiles + tile) * H + row); a workgroup owns a contiguous range of OUTPUT rows of it
// and walks the source rows that range needs as one stream of jobs: job k multiplies one source row, and its operand was
// requested S1_D - 1 jobs earlier, whatever strip or image that row belongs to, so there is no pipeline bubble between
// strips and the ranges of all workgroups are equal to within one row.
// Memory goes through buffer instructions: descriptor = one image, scalar offset = the row, lane offset = pixel and
// channels; a lane outside the image (or a result the workgroup does not own) carries an offset beyond the descriptor's
// range, which reads zero and stores nothing.  Every job therefore issues the same load and store without a branch and the
// compiler's wait counts leave the whole ring in flight.
constexpr int S1_D = 8;           // ring slots: source rows in flight per wave + 1
constexpr unsigned OOB = 0x80000000u;
enum { EPI_NONE = 0, EPI_STATS = 1, EPI_POOL = 3 };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const void* base, long off, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + off, 0, (int)bytes, 0x00020000);
}

template <bool AFF, int EPI>
__global__ __launch_bounds__(256, 4) void conv3x3g_strip_kernel(const bf16* __restrict__ src, const bf16* __restrict__ wp,
                                                                bf16* __restrict__ dst, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, float* col_sum, float* col_sumsq,
                                                                StripGeom gm, const float* __restrict__ pool_scale,
                                                                const float* __restrict__ pool_shift, bf16* __restrict__ pool_dst,
                                                                float pool_norm, StesDet det) {
    __shared__ float red[2][64];
    constexpr bool POOL = EPI == EPI_POOL;
    constexpr int D = S1_D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int cb = blockIdx.y;
    const int g = cb * 4 + wave;
    const int H = gm.Hs, W = gm.Ws, C = gm.C, n_ct = gm.n_ct;
    const int ch0 = g * 16 + 4 * q;
    const int kb = blockIdx.x * gm.chunk;                 // owned output rows [kb, ke) of the flat row sequence
    if (kb >= gm.rows) return;
    const int ke = kb + gm.chunk < gm.rows ? kb + gm.chunk : gm.rows;
    // Jobs: source rows 0..H of a strip, row H being the zero row below the image, so that every job completes exactly one
    // output row (the one above its source row) and all jobs are alike.  The first job is the source row above the first
    // owned output row (or row 0), the last one the source row below the last owned output row.
    const int strip_b = kb / H, ro_b = kb - strip_b * H;
    const int strip_e = (ke - 1) / H, ro_e = ke - 1 - strip_e * H;
    const int r_first = ro_b > 0 ? ro_b - 1 : 0;
    const int K = (strip_e * (H + 1) + ro_e + 1) - (strip_b * (H + 1) + r_first) + 1;
    // wA[dy + 1][dx + 1]: the weights a source pixel at offset (dy, dx) from the output pixel is multiplied by
    s16x4 wA[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) wA[a][b] = tap_frag(wp, g, a * 3 + b, lane);
    float psc[4], psh[4];
    Epilogue ep;
    ep.stats = EPI == EPI_STATS; ep.pool = POOL;
    ep.clear();
#pragma unroll
    for (int i = 0; i < 4; ++i) { psc[i] = 1.f; psh[i] = 0.f; ep.sc[i] = 1.f; ep.sh[i] = 0.f; }
    if constexpr (AFF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { psc[i] = scale[ch0 + i]; psh[i] = shift[ch0 + i]; }
    }
    if constexpr (POOL) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { ep.sc[i] = pool_scale[ch0 + i]; ep.sh[i] = pool_shift[ch0 + i]; }
    }
    const unsigned rowb = (unsigned)(W * C * 2);          // bytes per image row
    const unsigned imgb = rowb * (unsigned)H;             // bytes per image (below 2 GiB: host check)
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    // a position in the job sequence (wave-uniform): source row r (H = the zero row) of column tile ct of image n
    struct Cur { int r, ct, n; };
    // this lane's byte offset inside an image row for column tile ct (beyond the descriptor for columns outside the image)
    auto lane_off_of = [&](int ct, bool store) {
        const int c = ct * 14 - 1 + j;
        const bool ok = c >= 0 && c < W && (!store || (j >= 1 && j <= 14));
        return ok ? (unsigned)((c * C + ch0) * 2) : OOB;
    };
    Cur cc;
    cc.r = r_first; cc.n = strip_b / n_ct; cc.ct = strip_b - cc.n * n_ct;
    Cur pc = cc;                                          // the job whose operand is requested next
    int pk = 0;
    unsigned poff = lane_off_of(pc.ct, false);
    __amdgpu_buffer_rsrc_t psrc = image_rsrc(src, (long)pc.n * imgb, imgb);
    auto pf_step = [&]() {
        if (pk + 1 < K) {
            ++pk;
            if (++pc.r > H) {
                pc.r = 0;
                if (++pc.ct == n_ct) { pc.ct = 0; ++pc.n; psrc = image_rsrc(src, (long)pc.n * imgb, imgb); }
                poff = lane_off_of(pc.ct, false);
            }
        }
    };
    // the zero row is asked for like any other row, with every lane out of range
    auto pf_src = [&]() { return __builtin_amdgcn_raw_buffer_load_b64(psrc, pc.r == H ? OOB : poff, pc.r * rowb, 0); };
    u32x2 buf[D];
#pragma unroll
    for (int u = 0; u < D; ++u) buf[u] = u32x2{0u, 0u};
#pragma unroll
    for (int u = 0; u < D - 1; ++u) { buf[u] = pf_src(); pf_step(); }
    unsigned coff = lane_off_of(cc.ct, true);
    bool colin = lane_off_of(cc.ct, false) != OOB;
    __amdgpu_buffer_rsrc_t cdst = image_rsrc(dst, (long)cc.n * imgb, imgb);
    // a0: output row r - 1 so far, a1: output row r so far (r = the source row about to be multiplied)
    f32x4 a0[3] = {zero, zero, zero}, a1[3] = {zero, zero, zero};
    int fo = kb - (ro_b > 0 ? 1 : 0) - 1;                 // flat index of the output row the current job completes (row r - 1)
    for (int k0 = 0; k0 < K; k0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            u32x2 braw = buf[u];
            if constexpr (AFF) {
                float f[4];
                unpack4(braw, f);
#pragma unroll
                for (int i = 0; i < 4; ++i) f[i] = fmaxf(f[i] * psc[i] + psh[i], 0.f);
                braw = pack4(f);
                if (!colin || cc.r == H) braw = u32x2{0u, 0u};
            }
            const s16x4 b = __builtin_bit_cast(s16x4, braw);
            buf[(u + D - 1) % D] = pf_src();              // the slot of the job before this one is free
            f32x4 done[3];
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                done[x] = MFMA16(wA[2][x], b, a0[x]);
                a0[x] = MFMA16(wA[1][x], b, a1[x]);
                a1[x] = MFMA16(wA[0][x], b, zero);
            }
            {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = done[1][i] + dpp_from_next(done[2][i]) + dpp_from_prev(done[0][i]);
                const bool own = cc.r >= 1 && fo >= kb && fo < ke;
                const unsigned off = own ? coff : OOB;
                const u32x2 ob = ep.finish(o, off != OOB);
                __builtin_amdgcn_raw_buffer_store_b64(ob, cdst, off, (cc.r - 1) * rowb, 0);
            }
            pf_step();
            ++fo;
            if (++cc.r > H) {
                if constexpr (POOL) {
                    if (cc.ct == n_ct - 1 && fo - 1 < ke) ep.pool_out(pool_dst + (long)cc.n * C + ch0, pool_norm);
                }
                cc.r = 0;
                --fo;
                if (++cc.ct == n_ct) { cc.ct = 0; ++cc.n; cdst = image_rsrc(dst, (long)cc.n * imgb, imgb); }
                coff = lane_off_of(cc.ct, true);
                colin = lane_off_of(cc.ct, false) != OOB;
#pragma unroll
                for (int x = 0; x < 3; ++x) { a0[x] = zero; a1[x] = zero; }
            }
        }
    }
    if constexpr (EPI == EPI_STATS) ep.commit(col_sum, col_sumsq, cb, red, det);
}

template <typename K>
int resident_blocks(K kernel) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, 0) != hipSuccess || occ < 1) occ = 1;
    hipDeviceProp_t prop;
    int dev = 0;
    int cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    return occ * cus;
}

}  // namespace

bool stes_conv_strip_takes(int mode, int stride, int Wi) { return mode == 0 && stride == 1 && Wi <= 14; }

int stes_conv_strip_launch(const uint16_t* src, const uint16_t* wp, uint16_t* dst, const float* scale, const float* shift, float* col_sum,
                           float* col_sumsq, int N, int Hi, int Wi, int C, void* stream, const float* pool_scale, const float* pool_shift,
                           uint16_t* pool_out, float pool_norm) {
    // every per-lane byte offset is taken inside one image row, every row offset inside one image: 32 bits are enough
    STES_CHECK((long)Hi * Wi * C * 2 < (1L << 31), STES_ERR_UNSUPPORTED, "stes_conv3x3g: image beyond 2 GiB");
    const int ncb = C / 64;
    StripGeom gm;
    gm.N = N; gm.C = C; gm.Hs = Hi; gm.Ws = Wi;
    gm.n_ct = stes_div_up(Wi, 14);
    const bool pool = pool_out != nullptr;
    const long rows = (long)N * gm.n_ct * Hi;
    STES_CHECK(rows < (1L << 30), STES_ERR_UNSUPPORTED, "stes_conv3x3g: too many image rows");
    // Training: one resident set of workgroups (the statistics are committed once per workgroup), the flat row sequence of a
    // channel block cut into equal ranges.  With the squeeze a workgroup owns one image (its sums need no atomics).
    static std::mutex mu;
    static int resident[2] = {0, 0};
    int res;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (resident[pool] == 0)
            resident[pool] = pool ? resident_blocks(conv3x3g_strip_kernel<true, EPI_POOL>) : resident_blocks(conv3x3g_strip_kernel<true, EPI_STATS>);
        res = resident[pool];
    }
    const long per = res / ncb > 0 ? res / ncb : 1;
    long chunk = pool ? (long)gm.n_ct * Hi : (rows + per - 1) / per;
    if (chunk < 1) chunk = 1;
    gm.rows = (int)rows; gm.chunk = (int)chunk;
    const dim3 grid((unsigned)((rows + chunk - 1) / chunk), (unsigned)ncb), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bf16* s = (const bf16*)src; const bf16* w = (const bf16*)wp; bf16* d = (bf16*)dst;
    const int epi = pool ? EPI_POOL : (col_sum ? EPI_STATS : EPI_NONE);
    const StesDet det = epi == EPI_STATS ? stes_det_take(grid.x * grid.y) : StesDet{nullptr, 0u, 0u};
#define STRIP_LAUNCH(AF, EP) hipLaunchKernelGGL((conv3x3g_strip_kernel<AF, EP>), grid, block, 0, st, s, w, d, scale, shift, col_sum, \
        col_sumsq, gm, pool_scale, pool_shift, (bf16*)pool_out, pool_norm, det)
    if (scale) { if (epi == EPI_POOL) STRIP_LAUNCH(true, EPI_POOL); else if (epi == EPI_STATS) STRIP_LAUNCH(true, EPI_STATS); else STRIP_LAUNCH(true, EPI_NONE); }
    else { if (epi == EPI_POOL) STRIP_LAUNCH(false, EPI_POOL); else if (epi == EPI_STATS) STRIP_LAUNCH(false, EPI_STATS); else STRIP_LAUNCH(false, EPI_NONE); }
#undef STRIP_LAUNCH
    STES_LAUNCH_CHECK();
    return STES_OK;
}
