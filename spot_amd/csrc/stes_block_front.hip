// Evaluation forward of the front half of a bottleneck of stages 1 and 2 (RegNet-Y 800MF s1.b1: 112 x 112 x 32 -> 64 channels
// -> 56 x 56; s2.b1: 56 x 56 x 64 -> 128 channels -> 28 x 28; s2.b2 / s2.b3: 28 x 28 x 128 -> 128 channels, stride 1; s3.b1:
// 28 x 28 x 128 -> 320 channels -> 14 x 14) in one kernel: conv1 (1x1, with the gated shift's result
// G substituted for the first `fold` input channels, model/shift.py:157-161) -> bn1 + ReLU -> grouped 3x3 conv2 (stride 2 or 1)
// -> raw c2 and the squeeze-excite squeeze mean_hw relu(bn2(c2)) (timm Bottleneck as the reference builds it,
// train_e2e_spatial.py:305-314; BatchNorms in evaluation mode).  conv1 runs at the block's INPUT resolution, so its output is
// the largest tensor of the network (1.6 GB per 1,024 frames in s1.b1): as separate kernels it is written once and read once
// only to be reduced four to one by the strided convolution.  Here it never leaves the chip.
//   One workgroup (4 waves) walks an image's output rows.  Per output row ro (described for stride 2; with stride 1 a step
//   brings one new row and the lanes' pixels are one apart):
//   (a) the two new conv1 rows 2ro and 2ro+1: pixels x channels products on MFMA (weights in registers for the whole launch,
//       pixel fragments straight from memory, fetched one output row ahead), rounded to bf16 as the kernel sequence stores
//       them, bn1 + ReLU, into a ring of four row slots in LDS (row 2ro-1 is there from the step before; row -1 is zeros);
//   (b) the strided convolution of output row ro from the three slots: lane = output column, nine K = 16 MFMAs per
//       (16-channel group, 16-column fragment) on 8-byte LDS reads (pixel stride padded by 8 bytes: two pixels apart is four
//       banks apart), result to a staging row and out of it as whole 16-byte chunks; the squeeze is summed on the way.
// Same roundings as stes_gemm + stes_conv3x3g_fwd_pool.
#include "stes_common.h"

namespace {

typedef short f_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned f_u32x2 __attribute__((ext_vector_type(2)));
#define BF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float bf_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}
// A operand of one tap from the k-step-pair pack of stes_pack_conv3x3g ([G][5][64][8])
__device__ __forceinline__ f_s16x4 bf_tap_frag(const bf16* __restrict__ wp, int g, int slot, int lane) {
    const int m = lane & 15, q = lane >> 4;
    const int packed_lane = ((slot & 1) * 2 + (q >> 1)) * 16 + m;
    return *reinterpret_cast<const f_s16x4*>(wp + (((long)g * 5 + (slot >> 1)) * 64 + packed_lane) * 8 + 4 * (q & 1));
}

// STRIDE: conv2's stride (2: the first block of a stage, 1: the others); NSPLIT: conv1's output channels are dealt to this many
// wave groups (2 where a row has too few pixel tiles to occupy four waves); WLDS: conv1's weights live in LDS in fragment order
// instead of registers (s3.b1: 80 KB), and the ring has three slots instead of four (the two rows a step brings replace the two
// oldest ones, which the barrier after the convolution has released)
template <int CIN, int CMID, int WI, int STRIDE, int NSPLIT, bool WLDS>
struct FrontShape {
    static constexpr int NT = CMID / 16, NTW = NT / NSPLIT, KS = CIN / 32, WO = WI / STRIDE, CT = (WI + 15) / 16, CF = (WO + 15) / 16;
    // ring: elements per pixel; the padding puts two lanes' pixels (STRIDE pixels apart) four banks apart for the 8-byte reads
    static constexpr int PXS = CMID + (STRIDE == 2 ? 4 : 8);
    static constexpr int SLOT = (WI + 2) * PXS;            // a row slot with a zero pixel either side
    static constexpr int NSLOT = WLDS ? 3 : 4;
    static constexpr int WEL = WLDS ? NT * KS * 64 * 8 : 0; // elements of the weight image in LDS
    static constexpr int SPX = CMID + 8;                   // staging row: elements per pixel (16-byte chunks stay aligned)
    static constexpr int GPW = CMID / 64;                  // 16-channel groups per wave
    static_assert(CMID % 64 == 0 && (!WLDS || STRIDE == 2), "groups per wave; three slots need two new rows per step");
    static constexpr int UNITS = STRIDE * CT * NSPLIT, UPW = (UNITS + 3) / 4;      // (pixel tile, channel share) units of a step
    static constexpr size_t LDS = (size_t)(NSLOT * SLOT + WO * SPX + WEL) * 2 + 4 * CMID * sizeof(float);
    static_assert(NT % NSPLIT == 0 && (NSPLIT == 1 || NSPLIT == 2), "channel shares");
};

template <int CIN, int CMID, int WI, int STRIDE, int NSPLIT, bool WLDS>
__global__ __launch_bounds__(256, (WLDS ? 1 : 2)) void block_front_kernel(const stes_block_front_args p) {
    typedef FrontShape<CIN, CMID, WI, STRIDE, NSPLIT, WLDS> S;
    extern __shared__ __attribute__((aligned(16))) unsigned char bf_smem[];
    bf16* ring = reinterpret_cast<bf16*>(bf_smem);                          // [NSLOT][SLOT]
    bf16* stage = ring + S::NSLOT * S::SLOT;                                // [WO][SPX]
    bf16* wl = stage + S::WO * S::SPX;                                      // [NT][KS][64 lanes][8] when WLDS
    float* cst = reinterpret_cast<float*>(wl + S::WEL);
    float* s1 = cst, *b1 = cst + CMID, *s2 = cst + 2 * CMID, *b2 = cst + 3 * CMID;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int HI = p.H, HO = HI / STRIDE, fold = p.fold;
    const bf16* __restrict__ X = (const bf16*)p.X;
    const bf16* __restrict__ G = (const bf16*)p.G;
    for (int i = tid; i < CMID; i += 256) { s1[i] = p.s1[i]; b1[i] = p.b1[i]; s2[i] = p.s2[i]; b2[i] = p.b2[i]; }
    for (int i = tid; i < S::NSLOT * S::SLOT / 2; i += 256) reinterpret_cast<unsigned*>(ring)[i] = 0u;     // borders stay zero for good
    // conv1's weights of this wave's channel share: fragment (j, ks) = output channels 16 (nt0 + j).., input channels 32ks..
    const int nt0 = (wave % NSPLIT) * S::NTW;
    bf16x8 wf[WLDS ? 1 : S::NTW][WLDS ? 1 : S::KS];
    if (WLDS) {
        // fragment (j, ks) as 64 consecutive 16-byte pieces, one per lane: every wave reads its operands with conflict-free b128s
        for (int i = tid; i < S::NT * S::KS * 64; i += 256) {
            const int l = i & 63, f = i >> 6, j = f / S::KS, ks = f - j * S::KS;
            *reinterpret_cast<bf16x8*>(wl + (long)i * 8) =
                *reinterpret_cast<const bf16x8*>((const bf16*)p.W1 + (long)(16 * j + (l & 15)) * CIN + 32 * ks + 8 * (l >> 4));
        }
    } else {
#pragma unroll
        for (int j = 0; j < S::NTW; ++j)
#pragma unroll
            for (int ks = 0; ks < S::KS; ++ks)
                wf[WLDS ? 0 : j][WLDS ? 0 : ks] = *reinterpret_cast<const bf16x8*>((const bf16*)p.W1 + (long)(16 * (nt0 + j) + fr) * CIN + 32 * ks + 8 * fq);
    }
    // conv2's taps of this wave's groups
    f_s16x4 wA[S::GPW][3][3];
#pragma unroll
    for (int gi = 0; gi < S::GPW; ++gi)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) wA[gi][a][b] = bf_tap_frag((const bf16*)p.wp2, wave + 4 * gi, a * 3 + b, lane);
    // bn1's affine map for the channels this lane holds after conv1 (in registers where the wave's share is at most 64 channels)
    constexpr bool BN1_REGS = S::NTW <= 4;
    float sc1[BN1_REGS ? S::NTW : 1][4], sh1[BN1_REGS ? S::NTW : 1][4];
    if (BN1_REGS) {
#pragma unroll
        for (int j = 0; j < S::NTW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc1[j][e] = p.s1[16 * (nt0 + j) + 4 * fq + e]; sh1[j][e] = p.b1[16 * (nt0 + j) + 4 * fq + e]; }
    }
    // bn2's affine map for the channels this lane holds after conv2
    float sc2[S::GPW][4], sh2[S::GPW][4];
#pragma unroll
    for (int gi = 0; gi < S::GPW; ++gi)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc2[gi][e] = p.s2[(wave + 4 * gi) * 16 + 4 * fq + e]; sh2[gi][e] = p.b2[(wave + 4 * gi) * 16 + 4 * fq + e]; }
    __syncthreads();

    for (int n = blockIdx.x; n < p.N; n += gridDim.x) {
        const long img = (long)n * HI * WI;                                 // first pixel of the image
        // a pixel fragment of conv1's input: row r, column tile ct, k-step ks (the first `fold` channels come from G)
        auto load_a = [&](int r, int ct, int ks) -> bf16x8 {
            const int px = min(16 * ct + fr, WI - 1), k = 32 * ks + 8 * fq;
            const long pix = img + (long)(STRIDE == 1 ? min(r, HI - 1) : r) * WI + px;
            const bf16* src = k < fold ? G + pix * fold + k : X + pix * CIN + k;
            return *reinterpret_cast<const bf16x8*>(src);
        };
        // the conv1 rows a step brings: stride 2: rows 2ro and 2ro+1; stride 1: row ro+1 (row 0 comes from a step -1 that has no
        // convolution; the row below the image is zeros)
        const int ro_first = STRIDE == 2 ? 0 : -1;
        auto unit_row = [&](int ro, int u) { return STRIDE == 2 ? 2 * ro + (u / NSPLIT) / S::CT : ro + 1; };
        bf16x8 af[S::UPW][S::KS];
#pragma unroll
        for (int i = 0; i < S::UPW; ++i) {
            const int u = wave + 4 * i;
            if (u < S::UNITS) {
#pragma unroll
                for (int ks = 0; ks < S::KS; ++ks) af[i][ks] = load_a(unit_row(ro_first, u), (u / NSPLIT) % S::CT, ks);
            }
        }
        // row -1 of the image: zeros (the slot is reused by rows 3, 7, ..)
        for (int i = tid; i < WI * (CMID / 2); i += 256) {
            const int px = i / (CMID / 2), c2 = i - px * (CMID / 2);
            reinterpret_cast<unsigned*>(ring + (px + 1) * S::PXS)[c2] = 0u;
        }
        float psum[S::GPW][4];
#pragma unroll
        for (int gi = 0; gi < S::GPW; ++gi)
#pragma unroll
            for (int i = 0; i < 4; ++i) psum[gi][i] = 0.f;
        __syncthreads();

#pragma unroll 1
        for (int ro = ro_first; ro <= HO; ++ro) {
            // the staged output row of the step before leaves as whole chunks
            if (ro > 0) {
                bf16* dst = (bf16*)p.c2 + ((long)n * HO + (ro - 1)) * S::WO * CMID;
                for (int idx = tid; idx < S::WO * (CMID / 8); idx += 256) {
                    const int co = idx / (CMID / 8), c0 = (idx - co * (CMID / 8)) * 8;
                    *reinterpret_cast<bf16x8*>(dst + (long)co * CMID + c0) = *reinterpret_cast<const bf16x8*>(stage + co * S::SPX + c0);
                }
            }
            if (ro == HO) break;
            // (a) the step's new conv1 rows into their ring slots (row r lives in slot (r + 1) mod NSLOT; a row below the image is zeros)
#pragma unroll
            for (int i = 0; i < S::UPW; ++i) {
                const int u = wave + 4 * i;
                if (u >= S::UNITS) continue;
                const int r = unit_row(ro, u), ct = (u / NSPLIT) % S::CT;
                f32x4 acc[S::NTW];
#pragma unroll
                for (int j = 0; j < S::NTW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < S::KS; ++ks)
#pragma unroll
                    for (int j = 0; j < S::NTW; ++j) {
                        const bf16x8 w = WLDS ? *reinterpret_cast<const bf16x8*>(wl + ((long)((nt0 + j) * S::KS + ks) * 64 + lane) * 8) : wf[WLDS ? 0 : j][WLDS ? 0 : ks];
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, af[i][ks], acc[j], 0, 0, 0);
                    }
                const int px = 16 * ct + fr;
                if (px < WI) {
                    bf16* dstp = ring + ((r + 1) % S::NSLOT) * S::SLOT + (px + 1) * S::PXS + 16 * nt0 + 4 * fq;
                    const bool inside = STRIDE == 2 || r < HI;
#pragma unroll
                    for (int j = 0; j < S::NTW; ++j) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int ch = 16 * (nt0 + j) + 4 * fq + e;
                            const float sc = BN1_REGS ? sc1[BN1_REGS ? j : 0][e] : s1[ch], sh = BN1_REGS ? sh1[BN1_REGS ? j : 0][e] : b1[ch];
                            o[e] = inside ? (bf16)fmaxf((float)(bf16)acc[j][e] * sc + sh, 0.f) : (bf16)0.f;
                        }
                        *reinterpret_cast<bf16x4*>(dstp + 16 * j) = o;
                    }
                }
            }
            // the next step's pixel fragments are requested now and arrive during (b)
            if (ro + 1 < HO) {
#pragma unroll
                for (int i = 0; i < S::UPW; ++i) {
                    const int u = wave + 4 * i;
                    if (u < S::UNITS) {
#pragma unroll
                        for (int ks = 0; ks < S::KS; ++ks) af[i][ks] = load_a(unit_row(ro + 1, u), (u / NSPLIT) % S::CT, ks);
                    }
                }
            }
            __syncthreads();
            if (ro < 0) continue;                                           // (stride 1: the step that only brings row 0)
            // (b) output row ro: source rows STRIDE ro - 1 + a = slots (STRIDE ro + a) mod NSLOT
#pragma unroll
            for (int gi = 0; gi < S::GPW; ++gi) {
                const int g = wave + 4 * gi, ch0 = g * 16 + 4 * fq;
                f32x4 acc[S::CF];
#pragma unroll
                for (int f = 0; f < S::CF; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const bf16* row = ring + ((STRIDE * ro + a) % S::NSLOT) * S::SLOT + ch0;
#pragma unroll
                    for (int b = 0; b < 3; ++b)
#pragma unroll
                        for (int f = 0; f < S::CF; ++f) {
                            const int pxi = min(STRIDE * (16 * f + fr) + b, WI + 1);      // ring pixel index of column STRIDE co + b - 1
                            const f_u32x2 raw = *reinterpret_cast<const f_u32x2*>(row + pxi * S::PXS);
                            acc[f] = BF_MFMA16(wA[gi][a][b], __builtin_bit_cast(f_s16x4, raw), acc[f]);
                        }
                }
#pragma unroll
                for (int f = 0; f < S::CF; ++f) {
                    const int co = 16 * f + fr;
                    if (co < S::WO) {
                        bf16x4 ob;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ob[e] = (bf16)acc[f][e];
                        *reinterpret_cast<bf16x4*>(stage + co * S::SPX + ch0) = ob;
#pragma unroll
                        for (int e = 0; e < 4; ++e) psum[gi][e] += fmaxf((float)ob[e] * sc2[gi][e] + sh2[gi][e], 0.f);
                    }
                }
            }
            __syncthreads();
        }
        // the squeeze of the image: mean over its output pixels, rounded like the kernel sequence's bf16 vector
        const float inv = 1.0f / (float)(HO * S::WO);
#pragma unroll
        for (int gi = 0; gi < S::GPW; ++gi)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = bf_row16_sum(psum[gi][e]) * inv;
                if (fr == 0) ((bf16*)p.pooled)[(long)n * CMID + (wave + 4 * gi) * 16 + 4 * fq + e] = (bf16)s;
            }
    }
}

template <int CIN, int CMID, int WI, int STRIDE, int NSPLIT, bool WLDS = false>
int front_launch(const stes_block_front_args* a, void* stream) {
    typedef FrontShape<CIN, CMID, WI, STRIDE, NSPLIT, WLDS> S;
    static const bool big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(&block_front_kernel<CIN, CMID, WI, STRIDE, NSPLIT, WLDS>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    STES_CHECK(big_lds, STES_ERR_UNSUPPORTED, "stes_block_front_eval: cannot raise the dynamic LDS limit");
    static int resident = 0;
    if (!resident) {
        int per_cu = 0, dev = 0, cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, block_front_kernel<CIN, CMID, WI, STRIDE, NSPLIT, WLDS>, 256, S::LDS) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        resident = per_cu * (cus < 1 ? 256 : cus);
    }
    const int grid = a->N < resident ? a->N : resident;
    hipLaunchKernelGGL((block_front_kernel<CIN, CMID, WI, STRIDE, NSPLIT, WLDS>), dim3(grid), dim3(256), S::LDS, (hipStream_t)stream, *a);
    STES_LAUNCH_CHECK();
    return STES_OK;
}

}  // namespace

extern "C" int stes_block_front_eval(const stes_block_front_args* a, void* stream) {
    STES_CHECK(a && a->X && a->W1 && a->s1 && a->b1 && a->wp2 && a->s2 && a->b2 && a->c2 && a->pooled, STES_ERR_ARG,
               "stes_block_front_eval: null pointer");
    STES_CHECK(a->N > 0 && a->H > 0 && (a->stride == 1 || (a->stride == 2 && a->H % 2 == 0)), STES_ERR_ARG, "stes_block_front_eval: bad geometry");
    STES_CHECK(a->fold >= 0 && a->fold % 8 == 0 && a->fold <= 32 && a->fold <= a->Cin && (a->fold == 0 || a->G), STES_ERR_ARG,
               "stes_block_front_eval: fold must be a multiple of 8 up to 32 (with G given)");
    if (a->stride == 2 && a->Cin == 32 && a->Cmid == 64 && a->W == 112) return front_launch<32, 64, 112, 2, 1>(a, stream);
    if (a->stride == 2 && a->Cin == 64 && a->Cmid == 128 && a->W == 56) return front_launch<64, 128, 56, 2, 1>(a, stream);
    if (a->stride == 1 && a->Cin == 128 && a->Cmid == 128 && a->W == 28) return front_launch<128, 128, 28, 1, 2>(a, stream);
    if (a->stride == 2 && a->Cin == 128 && a->Cmid == 320 && a->W == 28) return front_launch<128, 320, 28, 2, 1, true>(a, stream);
    return stes_set_error(STES_ERR_UNSUPPORTED, "stes_block_front_eval: built for 112 x 112 x 32 -> 64, 56 x 56 x 64 -> 128 and 28 x 28 x 128 -> 320 channels "
                                                "with stride 2 and 28 x 28 x 128 -> 128 channels with stride 1");
}
