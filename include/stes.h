/* libstes_hip: C-ABI of the MI355X (gfx950) kernels behind the STES hot path.
 *
 * The reference (hoangqnguyen/spot) has no FFI of its own: its hot path is
 * PyTorch module code.  Each entry point below replaces the arithmetic of the
 * reference lines cited beside it; the Python host (spot_amd/) binds them with
 * ctypes and mirrors the reference's module API on top.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless it says "host"; the caller owns all
 *    memory; no entry point allocates, frees or synchronises;
 *  - `stream` is a hipStream_t passed as void*; launches are asynchronous;
 *  - return 0 on success, a negative STES_ERR_* otherwise (text via stes_last_error);
 *  - activations are NHWC ("channels last") bf16, viewed as [rows = n*h*w][channels];
 *  - bf16 values are passed as raw 16-bit words (uint16_t).
 */
#ifndef STES_H
#define STES_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define STES_OK 0
#define STES_ERR_ARG (-1)
#define STES_ERR_LAUNCH (-2)
#define STES_ERR_UNSUPPORTED (-3)

int stes_version(void);
const char* stes_last_error(void);

/* ---------------------------------------------------------------- GEMM ----
 * C[M,N] = act( pro(A)[M,K] * W[N,K]^T + bias )      (1x1 convolutions, Linear layers)
 * pro(A)[m,k] = relu?( A[m,k] * a_scale[k] + a_shift[k] ) * a_gate[m / gate_group][k]
 * A's columns [0,k0) may come from a second, compact tensor A0 (the gated-shift output).
 * Optional row mapping for a stride-s 1x1 conv: output pixel (n,ho,wo) reads input pixel
 * (n, ho*s, wo*s) of an Hi x Wi image.
 * Optional per-column sum / sum of squares of the bf16-rounded output (BatchNorm batch
 * statistics), accumulated with atomics into col_sum / col_sumsq (caller zeroes them).
 * Replaces: timm ConvNormAct 1x1 convs of the RegNet-Y bottleneck (constructed at
 * train_e2e_spatial.py:305-314), nn.Linear in model/common.py:86-92, model/modules.py:11-17,
 * the GRU input projection inside nn.GRU (model/modules.py:24-30).                      */
typedef struct {
    const uint16_t* A;  int64_t lda;
    const uint16_t* A0; int64_t lda0; int32_t k0;
    const float* a_scale; const float* a_shift; int32_t a_relu;
    const float* a_gate; int32_t gate_group;
    int32_t map_stride, Ho, Wo, Hi, Wi;
    const uint16_t* W;  int64_t ldw;
    const float* bias;  int32_t act;          /* 0 none, 1 relu */
    void* C; int64_t ldc; int32_t c_f32;      /* output bf16 (0) or f32 (1) */
    float* col_sum; float* col_sumsq;
    int64_t M; int32_t N; int32_t K;
} stes_gemm_args;
int stes_gemm(const stes_gemm_args* a, void* stream);

/* dW[N1,N2] += P[M,N1]^T * pro(Q)[M,N2]   (f32 atomics; caller zeroes dW)
 * dbias[N1] += column sums of P (optional).  pro() and the row mapping apply to Q only.
 * Replaces: autograd weight gradients of the same layers.                               */
typedef struct {
    const uint16_t* P;  int64_t ldp;
    const uint16_t* Q;  int64_t ldq;
    const uint16_t* Q0; int64_t ldq0; int32_t k0;
    const float* q_scale; const float* q_shift; int32_t q_relu;
    const float* q_gate; int32_t gate_group;
    int32_t map_stride, Ho, Wo, Hi, Wi;
    float* dW; int64_t lddw;
    float* dbias;
    int64_t M; int32_t N1; int32_t N2;
} stes_wgrad_args;
int stes_wgrad(const stes_wgrad_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif
