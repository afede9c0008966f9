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
 *  - bf16 values are passed as raw 16-bit words (uint16_t);
 *  - one device per process: occupancy figures and the raised dynamic-LDS limits of the kernels are looked up once, on
 *    the device that is current at a kernel's first launch (the one exception, stes_gru_forward_noalloc, keeps its
 *    scratch address per device).  Drive a second GPU from a second process.
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

/* Deterministic mode (a test mode; no counterpart in the reference, whose float atomics inside cuDNN/MIOpen are not under
 * its control either).  Every cross-workgroup fp32 sum of this library (BatchNorm statistics, BatchNorm-backward sums,
 * weight gradients, bias gradients) is accumulated with float atomics, so its last bits depend on the order in which the
 * workgroups of a launch arrive.  With a ticket (256 bytes of device memory, cleared by this call on `stream`) the
 * workgroups of every such launch commit in block-index order and LDS accumulations run wave after wave: the same values
 * are added in a fixed order, and two runs of the same work give the same bits.  Commits are serialised, so launches are
 * slower.  While the mode is on, drive the library from one stream.  ticket[1] != 0 afterwards means a wait for a turn ran
 * out (the order was not kept).  NULL turns the mode off. */
int stes_set_deterministic(uint32_t* ticket, void* stream);

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
    const float* bias;  int32_t act;          /* 0 none, 1 relu, 2 sigmoid */
    void* C; int64_t ldc; int32_t c_f32;      /* output bf16 (0) or f32 (1) */
    float* col_sum; float* col_sumsq;
    const uint16_t* relu_mask; int64_t ldmask; /* optional: C[m,n] = 0 where relu_mask[m,n] <= 0 (ReLU backward) */
    uint16_t* C0; int64_t ldc0; int32_t n_split; /* optional: columns [0,n_split) go to the compact bf16 tensor C0 instead */
    int64_t M; int32_t N; int32_t K;
    float out_scale;                          /* C = act(out_scale * pro(A) W^T + bias); 0 means 1 */
    /* BatchNorm-backward prologue (instead of the ones above): with a_raw [M,K] (leading dimension lda; the raw tensor under
     * a BatchNorm whose output gradient is A), pro(A)[m,k] = bf16(bn_alpha[k]*A[m,k] + bn_beta[k]*a_raw[m,k] + bn_delta[k]),
     * i.e. the tensor stes_bn_bwd_apply would write, which is also stored to a_out [M,K] (lda_out) when a_out is given.
     * Streaming shapes only (N in {32,64,128,160,320} or a multiple of 128 up to 1024, K % 32 == 0, K <= 320, M >= 8192,
     * bf16 output); else UNSUPPORTED.                                                                                    */
    const uint16_t* a_raw; const float* bn_alpha; const float* bn_beta; const float* bn_delta;
    uint16_t* a_out; int64_t lda_out;
    /* Weight gradient of the same layer in the same launch (optional, with the BatchNorm-backward prologue only):
     * wg_dw[K, N] (f32, leading dimension wg_lddw, atomics; caller zeroes) += pro(A)[M,K]^T * Q[M,N], Q's columns [0,wg_k0) taken
     * from the compact tensor wg_q0 (as stes_wgrad's Q0/k0) and the columns [0,wg_perm_fold) of wg_dw permuted like
     * stes_wgrad's perm_fold.  pro(A) then never travels through memory: no a_out, no stes_wgrad launch.  With
     * wg_map_stride > 1 row m of Q is the input pixel (n, ho*s, wo*s) of an Hi x Wi image (the Ho/Wo/Hi/Wi fields above; the
     * weight gradient of a strided 1x1 shortcut conv, whose input gradient is computed at the output resolution).  Built for
     * K = 64, N = 32 (stage 1's first 1x1 conv, 12.8 M rows at the named configuration, and its shortcut conv); other shapes:
     * STES_ERR_UNSUPPORTED.                                                                                                 */
    const uint16_t* wg_q; int64_t wg_ldq; const uint16_t* wg_q0; int64_t wg_ldq0; int32_t wg_k0; int32_t wg_perm_fold;
    float* wg_dw; int64_t wg_lddw; int32_t wg_map_stride;
    /* Block-tail epilogue (optional; streaming shapes, bf16 output, no statistics, no split output): with e_scale / e_shift [N]
     * the value stored at C[m,n] is relu(e_scale[n]*c + e_shift[n] + r) with c the bf16-rounded product and r = e_res[m,n]
     * (bf16, leading dimension e_ldres), or e_rscale[n]*e_res[m,n] + e_rshift[n] when e_rscale is given, or nothing when
     * e_res is NULL (the stem: BatchNorm + ReLU only); columns
     * [0, e_slice_n) of the result are also copied to the compact tensor e_slice [M][e_slice_n].  This is
     * stes_bn_act_residual applied to this product without the product's trip through memory, for evaluation, where the
     * BatchNorm over the product is a fixed affine map (timm Bottleneck: conv3.bn + shortcut + act3; stem bn + act).
     * Built for (N, K) = (64, 64), (128, 128), (320, 320) with the BatchNorm + ReLU (+ gate) prologue and (32, 32) without.  */
    const float* e_scale; const float* e_shift; const uint16_t* e_res; int64_t e_ldres; const float* e_rscale; const float* e_rshift;
    uint16_t* e_slice; int32_t e_slice_n;
} stes_gemm_args;
int stes_gemm(const stes_gemm_args* a, void* stream);

/* Evaluation forward of the front half of a bottleneck of stages 1 and 2 of RegNet-Y 800MF (stride 2: 112 x 112 x 32 -> 64 channels
 * -> 56 x 56, 56 x 56 x 64 -> 128 channels -> 28 x 28 and 28 x 28 x 128 -> 320 channels -> 14 x 14; stride 1: 28 x 28 x 128 -> 128 channels) in ONE kernel: conv1 (1x1, the
 * gated shift's result G in place of the first `fold` input channels, reference model/shift.py:157-161; fold may be 0) -> bn1 +
 * ReLU -> grouped 3x3 conv2 with the given stride -> raw c2 [N*(H/stride)*(W/stride)][Cmid] and the squeeze-excite squeeze pooled [N][Cmid] = mean relu(bn2(c2)) (timm Bottleneck,
 * train_e2e_spatial.py:305-314; evaluation-mode BatchNorms).  conv1's output, at the block's input resolution the largest tensor
 * of the network, stays on chip.  Same roundings and outputs as stes_gemm followed by stes_conv3x3g_fwd_pool.
 * X [N*H*W][Cin]; G [N*H*W][fold]; W1 [Cmid][Cin] bf16 (fold-permuted columns as stes_gemm takes them); wp2: stes_pack_conv3x3g's
 * forward pack. */
typedef struct {
    const uint16_t* X; const uint16_t* G; int32_t fold;
    const uint16_t* W1; const float* s1; const float* b1;
    const uint16_t* wp2; const float* s2; const float* b2;
    uint16_t* c2; uint16_t* pooled;
    int32_t N, H, W, Cin, Cmid, stride;
} stes_block_front_args;
int stes_block_front_eval(const stes_block_front_args* a, void* stream);

/* Evaluation forward of one stage-3 bottleneck of RegNet-Y 800MF (14 x 14 x 320, stride 1, identity shortcut, squeeze-excite
 * with 80 hidden units, gated shift on the first 80 input channels) in ONE kernel, an image's activations staying in LDS
 * between the three convolutions: timm Bottleneck conv1 -> bn/act -> grouped conv2 -> bn/act -> SE -> conv3 -> bn -> + shortcut
 * -> act as the reference builds it (train_e2e_spatial.py:305-314), with model/shift.py:157-161's substitution of the gated
 * shift's result G for the first `fold` channels of conv1's input.  BatchNorms are the evaluation-mode affine maps
 * (scale, shift as stes_bn_finalize gives them with training == 0).  Same roundings as the kernel sequence
 * stes_gemm / stes_conv3x3g_fwd_pool / stes_gemm_small2 / stes_gemm with the block-tail epilogue.
 * X: block input [N*H*W][C]; G: [N*H*W][fold] (fold = 80, or 0 and G = NULL for a block without a gated shift); W1, W3: [C][C] bf16 (W1 with the fold-permuted columns stes_gemm takes);
 * wp2: stes_pack_conv3x3g's forward pack; Wse1 [rd][C], Wse2 [C][rd] bf16 with fp32 biases; out: [N*H*W][C];
 * slice (optional): the first slice_n channels of out as [N*H*W][slice_n] (the next block's gated-shift slice). */
typedef struct {
    const uint16_t* X; const uint16_t* G;
    const uint16_t* W1; const float* s1; const float* b1;
    const uint16_t* wp2; const float* s2; const float* b2;
    const uint16_t* Wse1; const float* bse1; const uint16_t* Wse2; const float* bse2;
    const uint16_t* W3; const float* s3; const float* b3;
    uint16_t* out; uint16_t* slice; int32_t slice_n;
    int32_t N, H, W, C, rd, fold;
} stes_block_eval_args;
int stes_block_eval_s3(const stes_block_eval_args* a, void* stream);

/* Two chained small products in one launch, for the squeeze-excite block's pair of 1x1 convolutions on the pooled vectors
 * (timm SEModule fc1 -> ReLU -> fc2 -> sigmoid as the reference builds it, train_e2e_spatial.py:305-314) and their
 * input-gradient pair in backward:
 *   H[M,N1] = act1(A[M,K1] W1[N1,K1]^T + bias1), zeroed where relu_mask1 <= 0     (bf16, also written to memory)
 *   C[M,N2] = act2(H W2[N2,N1]^T + bias2) * out_scale                             (fp32 or bf16)
 * act: 0 none, 1 ReLU, 2 sigmoid.  A workgroup owns 16 rows of both products, H stays in LDS in between; the same
 * roundings as two stes_gemm calls (H is rounded to bf16 before the second product).
 * K1, N1 multiples of 8; N1 <= 256, K1 <= 1024, N2 <= 1024.  Replaces two launches of a few microseconds each that sit in a
 * dependent chain between two large kernels of every block. */
typedef struct {
    const uint16_t* A; int64_t lda; int64_t M; int32_t K1;
    const uint16_t* W1; int32_t ldw1; int32_t N1; const float* bias1; int32_t act1;
    const uint16_t* relu_mask1; int64_t ldmask1;
    uint16_t* H; int64_t ldh;
    const uint16_t* W2; int32_t ldw2; int32_t N2; const float* bias2; int32_t act2; float out_scale;
    void* C; int64_t ldc; int32_t c_f32;
} stes_gemm2_args;
int stes_gemm_small2(const stes_gemm2_args* a, void* stream);

/* dW[N1,N2] += P[M,N1]^T * pro(Q)[M,N2]   (f32 atomics; caller zeroes dW)
 * dbias[N1] += column sums of P (optional; of the prologue's result when P has one).  pro() and the row mapping apply to Q only.
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
    int32_t perm_fold;    /* > 0: output column c < perm_fold is stored at column dest(c) (gated-shift interleave) */
    int64_t M; int32_t N1; int32_t N2;
    /* optional BatchNorm-backward prologue on P (64-wide tile only: N1 <= 64 or N2 <= 64): with p_raw [M,N1] (ldp), the raw
     * tensor under the BatchNorm whose output gradient is P, the operand is bf16(bn_alpha*P + bn_beta*p_raw + bn_delta) --
     * the tensor stes_bn_bwd_apply would have written.  For layers with no input gradient (the stem).                   */
    const uint16_t* p_raw; const float* bn_alpha; const float* bn_beta; const float* bn_delta;
    /* Optional workspace of ws_floats floats (contents irrelevant on entry, garbage on return).  The 320 x 320 products of
     * stage 3 (M a multiple of 32, at least 8192 rows, no dbias) run on a kernel that owns whole 320-column strips per row
     * chunk; with stes_wgrad_ws_floats() floats every row chunk writes its partial result to its own slab and a second
     * launch adds the slabs to dW, instead of 128 chunks adding 400 KiB each by fp32 atomics (40 % of that kernel's time). */
    float* ws; int64_t ws_floats;
} stes_wgrad_args;
int stes_wgrad(const stes_wgrad_args* a, void* stream);
int64_t stes_wgrad_ws_floats(void);

/* ---------------------------------------------------- BatchNorm / block glue ----
 * Training-mode BatchNorm2d/3d is split around the convolutions: the producing kernel
 * accumulates per-channel sum / sum of squares, stes_bn_finalize turns them into
 * scale = gamma*invstd, shift = beta - mean*scale (and updates the running statistics with
 * the unbiased variance, momentum as torch.nn.BatchNorm), and the consuming kernel applies
 * scale/shift(+ReLU) while it loads.  training = 0 uses the running statistics instead.
 * Replaces: timm BatchNormAct2d inside every ConvNormAct of regnety_008 and nn.BatchNorm3d of
 * _GSM (model/shift.py:94,108).                                                         */
int stes_bn_finalize(const float* sum, const float* sumsq, double count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, int training,
                     float* scale, float* shift, float* save_mean, float* save_invstd, int C, void* stream);

/* out = relu(scale*c + shift + r), r = res_scale*res + res_shift | res | 0; optional sum / sum of
 * squares of out's first stat_c channels (the next block's gated-shift BatchNorm3d input) and a compact
 * copy slice_out [M][stat_c] of those channels.  relu_bits (optional, [M][C/8] bytes): bit i of byte (m, j) = out[m, 8j+i] > 0,
 * the ReLU mask the backward pass needs at a sixteenth of out's bytes (stes_block_in_grad's x_bits).
 * Replaces: conv3.bn + shortcut add + act3 of the timm Bottleneck, and stem bn + ReLU.   */
int stes_bn_act_residual(const uint16_t* c, const float* scale, const float* shift, const uint16_t* res,
                         const float* res_scale, const float* res_shift, uint16_t* out, float* stat_sum,
                         float* stat_sumsq, int stat_c, uint16_t* slice_out, int64_t M, int C, uint8_t* relu_bits,
                         void* stream);
/* The same for a block whose successor applies the plain temporal shift (TemporalShift, reference model/shift.py:37-66, which
 * shifts its input in place): the shift is taken in the store, so out is the shifted tensor and no shift pass follows.
 * Rows = clips x T frames x HW pixels; tsm_fold % 4 == 0. */
int stes_bn_act_residual_tsm(const uint16_t* c, const float* scale, const float* shift, const uint16_t* res,
                             const float* res_scale, const float* res_shift, uint16_t* out, int64_t M, int C,
                             int tsm_fold, int T, int HW, void* stream);

/* out[n,ch] = norm * sum_hw act(c[n,hw,ch]) * (mul ? mul[n,hw,ch] : 1), act = relu(scale*c+shift)
 * when scale is given.  Squeeze of timm SEModule, its backward, and the head's global pool.   */
/* out = bf16(relu(scale * c + shift) * gate[row / HW]): the squeeze-excited activation in front of a bottleneck's last 1x1
 * convolution (timm Bottleneck: conv2's bn + act, then SEModule's x * gate) written out, for the layers where stes_gemm's
 * operand prologue is slower than this pass plus a plain product (the 768-channel blocks); same arithmetic and rounding as
 * that prologue.  c, out: [N*HW][C] bf16; gate: [N][C] fp32. */
int stes_bn_act_gate(const uint16_t* c, const float* scale, const float* shift, const float* gate, uint16_t* out, int N, int HW,
                     int C, void* stream);
int stes_chan_pool(const uint16_t* c, const float* scale, const float* shift, const uint16_t* mul, float* out,
                   uint16_t* out_bf16, int N, int HW, int C, float norm, void* stream);

/* Squeeze-excite (timm SEModule inside regnety_008's Bottleneck: mean over H,W -> fc1 + ReLU -> fc2 -> sigmoid) runs on
 * stes_chan_pool + two stes_gemm calls (bias + ReLU / sigmoid epilogues); its backward on stes_se_bwd_pool (which also
 * applies the sigmoid's derivative) + two stes_gemm (relu_mask, out_scale = 1/HW) + two stes_wgrad calls.
 * stes_sigmoid_bwd: out[i] = bf16(d[i] * g[i] * (1 - g[i])), the gradient at a sigmoid's input, on its own.      */
int stes_sigmoid_bwd(const float* d, const float* g, uint16_t* out, int64_t n, void* stream);
/* Channel attention over a clip (reference model/modules.py:86-105 ChannelAttention, applied to the frame features at
 * train_e2e_spatial.py:437-438): mean over the T frames (stes_chan_pool with HW = T) -> two bias-free stes_gemm calls
 * (ReLU / sigmoid epilogues) -> stes_clip_gate_fwd: out[b,t,c] = x[b,t,c] * gate[b,c]   (x, out bf16 [B*T][C]; gate fp32).
 * Backward, two calls of stes_clip_gate_bwd around the gate MLP's backward (stes_sigmoid_bwd, stes_gemm, stes_wgrad):
 *   reduction (x, dgate given; gate, dpool, out NULL):  dgate[b,c] = sum_t (da + db)[b,t,c] * x[b,t,c]
 *   apply (gate, dpool, out given; x, dgate NULL):      out[b,t,c] = (da + db)[b,t,c] * gate[b,c] + dpool[b,c]
 * da, db (db may be NULL), out fp32 [B*T][C]; out may alias da; dpool already scaled by 1/T.             */
int stes_clip_gate_fwd(const uint16_t* x, const float* gate, uint16_t* out, int B, int T, int C, void* stream);
int stes_clip_gate_bwd(const float* da, const float* db, const uint16_t* x, const float* gate, const float* dpool,
                       float* dgate, float* out, int B, int T, int C, void* stream);
/* One pass over (c2, dA3 = gradient at the gated activation) per image: S [4][N][C] = sums over H,W of mask, dA3*mask,
 * mask*c2, dA3*mask*c2 with mask = [scale*c2 + shift > 0]; dgate [N][C] = gradient at the SE gate.  With the SE backward's
 * dpool, stes_se_bwd_bnsums turns S into the two BatchNorm-backward sums of conv2.bn (what stes_bn_bwd_reduce would give
 * for dy = (dA3*gate + dpool)*mask; accumulated with atomics, the caller zeroes them), so that reduction pass over the
 * tensors is skipped.  With gate and dpre (both or neither) it also writes dpre [N][C] = bf16(dgate * gate * (1 - gate)), the
 * gradient at the sigmoid's input (what stes_sigmoid_bwd would give).                                               */
int stes_se_bwd_pool(const uint16_t* c2, const float* scale, const float* shift, const uint16_t* dA3, float* S,
                     float* dgate, const float* gate, uint16_t* dpre, int N, int HW, int C, void* stream);
int stes_se_bwd_bnsums(const float* S, const float* gate, const float* dpool, float* s_dy, float* s_dyr, int N, int C,
                       void* stream);

/* BatchNorm backward.  The incoming gradient is formed on the fly:
 *   dY = (D * mul[m/group] + add[m/group]) * [act_scale*act_raw + act_shift > 0]
 * (each factor optional).  reduce: s_dy += sum dY, s_dyr1 += sum dY*raw1, s_dyr2 += sum dY*raw2.
 * finalize: alpha/beta/delta such that d(raw) = alpha*dY + beta*raw + delta, and
 * dgamma += sum dY*xhat, dbeta += sum dY.  apply: out = alpha*dY + beta*raw1 + delta.     */
typedef struct {
    const uint16_t* D; const float* mul; const float* add; int32_t group;
    const uint16_t* act_raw; const float* act_scale; const float* act_shift;
    const uint16_t* raw1; const uint16_t* raw2;
    float* s_dy; float* s_dyr1; float* s_dyr2;
    const float* alpha; const float* beta; const float* delta; uint16_t* out;
    int64_t M; int32_t C;
} stes_bn_bwd_args;
int stes_bn_bwd_reduce(const stes_bn_bwd_args* a, void* stream);
int stes_bn_bwd_finalize(const float* s_dy, const float* s_dyr, const float* mean, const float* invstd,
                         const float* gamma, double count, float* alpha, float* beta, float* delta,
                         float* dgamma, float* dbeta, int C, void* stream);
int stes_bn_bwd_apply(const stes_bn_bwd_args* a, void* stream);

/* Gradient w.r.t. a bottleneck's input: out = ((c<fold ? dXf : dY) + residual) * (X > 0); the
 * residual is either same-shape (identity shortcut) or the stride-s shortcut conv's input
 * gradient given at output resolution and scattered to the even pixels.
 * `out` is the gradient at the output of the layer in front; with raw1 [M][C] (the raw conv output under that layer's last
 * BatchNorm; raw2: the raw output of its shortcut conv, or NULL) the BatchNorm-backward sums of that layer are accumulated on
 * the way (s_dy += sum out, s_dyr1 += sum out*raw1, s_dyr2 += sum out*raw2; atomics, caller zeroes) and replace its
 * stes_bn_bwd_reduce pass.  x_bits (optional): the mask (X > 0) as written by stes_bn_act_residual's relu_bits, read
 * instead of X (which may then be NULL).                                                   */
int stes_block_in_grad(const uint16_t* dY, int64_t ldy, const uint16_t* dXf, int fold, const uint16_t* resid,
                       int map_stride, int Ho, int Wo, int Hi, int Wi, const uint16_t* X, uint16_t* out,
                       int64_t M, int C, const uint16_t* raw1, const uint16_t* raw2, float* s_dy, float* s_dyr1,
                       float* s_dyr2, const uint8_t* x_bits, void* stream);
/* The same with the adjoint of the plain temporal shift (TemporalShift, reference model/shift.py:37-52) applied to dY while it is
 * read: channels [0, tsm_fold) take frame t-1's gradient, [tsm_fold, 2 tsm_fold) frame t+1's, zero beyond a clip of T frames
 * (rows = clips x T x Hi x Wi).  Replaces stes_tsm_shift(..., backward = 1) in front of stes_block_in_grad.  tsm_fold % 4 == 0,
 * fold == 0. */
int stes_block_in_grad_tsm(const uint16_t* dY, int64_t ldy, const uint16_t* dXf, int fold, const uint16_t* resid,
                           int map_stride, int Ho, int Wo, int Hi, int Wi, const uint16_t* X, uint16_t* out,
                           int64_t M, int C, const uint16_t* raw1, const uint16_t* raw2, float* s_dy, float* s_dyr1,
                           float* s_dyr2, const uint8_t* x_bits, int tsm_fold, int T, void* stream);
/* Backward of global average pooling + the last block's ReLU: dZ = (da+db)/HW * (out>0). */
int stes_pool_bwd(const float* da, const float* db, const uint16_t* out, uint16_t* dZ, int N, int HW, int C, void* stream);

/* ---------------------------------------------------- grouped 3x3 convolution ----
 * timm regnety_008 conv2: 3x3, padding 1, stride 1|2, groups = C/16, no bias.  in/out NHWC bf16.
 * Weights are re-packed from the checkpoint layout [C][16][3][3] fp32 into MFMA fragment order
 * by stes_pack_conv3x3g (one buffer for forward, one for the input gradient; G*5*512 bf16 each; the
 * input-gradient order depends on the stride: stride 2 groups the taps by output-pixel parity).
 * forward applies relu(in_scale*x + in_shift) (the previous BatchNorm+ReLU) while loading and
 * accumulates the output's per-channel sum / sum of squares.                              */
int stes_pack_conv3x3g(const float* w, uint16_t* wp_fwd, uint16_t* wp_dgrad, int G, int stride, void* stream);
int stes_conv3x3g_fwd(const uint16_t* in, const uint16_t* wp_fwd, uint16_t* out, const float* in_scale,
                      const float* in_shift, float* col_sum, float* col_sumsq, int N, int Hi, int Wi, int Ho, int Wo,
                      int stride, int C, void* stream);
/* Forward for evaluation, with the squeeze of the squeeze-excite block behind it (timm SEModule: x.mean((2, 3)) of the
 * activated conv2 output) taken on the way out: pool_out[n][c] (bf16) = mean over the image of
 * relu(pool_scale[c]*out + pool_shift[c]), pool_scale / pool_shift being conv2's BatchNorm folded from its running statistics.
 * Equals stes_conv3x3g_fwd followed by stes_chan_pool up to the order of the fp32 sum (fixed, independent of the batch);
 * images too large for the LDS window kernel: STES_ERR_UNSUPPORTED (run the two separately).                               */
int stes_conv3x3g_fwd_pool(const uint16_t* in, const uint16_t* wp_fwd, uint16_t* out, const float* in_scale,
                           const float* in_shift, const float* pool_scale, const float* pool_shift, uint16_t* pool_out,
                           int N, int Hi, int Wi, int Ho, int Wo, int stride, int C, void* stream);
/* Input gradient.  With mask_raw (the raw conv1 output [N*Hi*Wi][C] whose BatchNorm+ReLU fed this convolution) the result is
 * multiplied by [mask_scale*raw + mask_shift > 0] and the BatchNorm-backward sums s_dy[c] += sum din, s_dyr[c] += sum din*raw
 * are accumulated (atomics; caller zeroes), replacing a stes_bn_bwd_reduce pass.                                         */
int stes_conv3x3g_dgrad(const uint16_t* dout, const uint16_t* wp_dgrad, uint16_t* din, const uint16_t* mask_raw,
                        const float* mask_scale, const float* mask_shift, float* s_dy, float* s_dyr, int N, int Hi, int Wi,
                        int Ho, int Wo, int stride, int C, void* stream);
/* dw [C][16][3][3] fp32 += (atomics; caller zeroes) */
int stes_conv3x3g_wgrad(const uint16_t* dout, const uint16_t* in, const float* in_scale, const float* in_shift, float* dw,
                        int N, int Hi, int Wi, int Ho, int Wo, int stride, int C, void* stream);
/* Stem 3x3 stride-2 conv as im2col + stes_gemm: x NCHW fp32 -> patches [N*Ho*Wo][32] bf16,
 * column k = ci*9 + kh*3 + kw (timm stem.conv weight order), columns >= 9*Cin zero.       */
int stes_im2col_stem(const float* x, uint16_t* patches, int N, int Cin, int H, int W, void* stream);
/* The same patches from decoded uint8 frames x [B*T_valid][Cin][H][W] (what torchvision.io.read_image returns, reference
 * dataset/frame.py:29): value = (u/255 - mean[ci]) / std[ci] in fp32, i.e. `.float()/255` (:29) followed by Normalize (:124;
 * inference_on_mp4.py:523-524) -- the patches equal those of stes_im2col_stem on the loader's fp32 tensor bit for bit.  Clips
 * of T_valid frames are written as clips of T frames; frames t >= T_valid are the end padding of reference
 * train_e2e_spatial.py:427-430 (zeros AFTER normalisation).  mean/std: HOST arrays of Cin floats.                          */
int stes_im2col_stem_u8(const uint8_t* x, uint16_t* patches, const float* mean, const float* std, int B, int T, int T_valid,
                        int Cin, int H, int W, void* stream);

/* The stem convolution in one pass, without the round trip of its patches through memory (the 3x3 stride-2 convolution of
 * timm's RegNet stem as the reference builds it, model/shift.py / model/modules.py via timm.create_model at
 * train_e2e_spatial.py:230-300; the normalisation of uint8 frames follows dataset/frame.py:29,:124).
 * x: [B*T][Cin][H][W] fp32 (x_is_u8 == 0, T_valid == T), or [B*T_valid][Cin][H][W] uint8 laid out as B clips of T frames
 * whose missing frames are zero after normalisation, with HOST mean/std of Cin floats.  w_bf: [32][32] bf16, row = output
 * channel, column k = ci*9 + kh*3 + kw (zero from Cin*9 up).
 * Training form (e_scale == NULL): out = the raw output [B*T*Ho*Wo][32], patches (required) = what stes_im2col_stem gives,
 * col_sum/col_sumsq (optional, 32 floats each) += column statistics of the rounded output.
 * Evaluation form: out = relu(e_scale * c + e_shift); e_slice (optional) = its first e_slice_n (multiple of 8) channels as
 * [B*T*Ho*Wo][e_slice_n]; patches is not written. */
int stes_stem_conv(const void* x, int x_is_u8, const float* mean, const float* stdv, const uint16_t* w_bf, uint16_t* patches,
                   uint16_t* out, float* col_sum, float* col_sumsq, const float* e_scale, const float* e_shift,
                   uint16_t* e_slice, int e_slice_n, int B, int T, int T_valid, int Cin, int H, int W, void* stream);

/* ---------------------------------------------------- clip augmentation ----
 * The training loader's random augmentation (reference dataset/frame.py:78-140, applied per clip with one set of random
 * parameters, :296-323) on decoded uint8 frames x [n_clips*T][3][Hs][Ws]: ONE resampling through a per-clip homography (the
 * composition of the loader's crop / resized crop / flip / zoom-out / perspective / rotation / affine / Resize choices; zeros
 * outside the source like torchvision's fill=0), then, each only when its flag is set, hue shift, saturation, brightness,
 * contrast (against the frame's mean grey value), 5x5 Gaussian blur with reflected borders, and Normalize.
 * params [n_clips][STES_AUGMENT_PARAMS] f32 (device):
 *   [0..8]  row-major 3x3 map from output coordinates (pixel index + 0.5) to source coordinates (pixel index + 0.5)
 *   [9]     != 0: nearest-neighbour instead of bilinear
 *   [10,11] hue flag, shift (fraction of the colour circle)        [12,13] saturation flag, factor
 *   [14,15] brightness flag, factor                                [16,17] contrast flag, factor
 *   [18]    blur flag, [19..23] the five normalised Gaussian taps  [24..26] mean, [27..29] std of Normalize
 * Output: exactly one of out_f32 [n][3][Ho][Wo] = (v - mean)/std (what the loader hands to the model; feeds stes_im2col_stem)
 * or out_u8 = round(255 v) (feeds stes_im2col_stem_u8).  ws: stes_augment_ws_floats(n_clips*T, Ho, Wo) floats of workspace.
 * Results do not depend on scheduling (the frame mean is added up in a fixed order).                                        */
#define STES_AUGMENT_PARAMS 32
int64_t stes_augment_ws_floats(int n_frames, int Ho, int Wo);
int stes_augment_u8(const uint8_t* x, float* out_f32, uint8_t* out_u8, const float* params, float* ws, int64_t ws_floats,
                    int n_clips, int T, int Hs, int Ws, int Ho, int Wo, void* stream);

/* ---------------------------------------------------- temporal shift ----
 * Gated Shift Module, reference model/shift.py:83-161 (_GSM.forward :103-130, GatedShift.forward :157-161).
 * Xc is the COMPACT copy [B*T*H*W][fold] (NHWC bf16) of the first `fold` channels of the block input.  All tensors keep
 * the SOURCE channel order; the reference's within-group interleave dest(c) (:124-128) is applied by the consumer
 * (stes_cast_bf16 / stes_wgrad `perm_fold`).  fold must be 8 or a multiple of 16 (regnety_008: 8,16,32,80,192).
 *   stes_gsm_prep  : conv3D weight fp32 [2][fold/2][27] -> bf16 [2][27][fold/2] (layout of the gate kernel's scalar loads)
 *   stes_gsm_gate  : gate[m,2] = tanh(conv3D(relu(scale*x+shift)))    (scale/shift = BatchNorm3d via stes_bn_finalize)
 *   stes_gsm_shift : G[m,c] = y[t+1|t-1] + (x - y), y = gate*x
 *   stes_gsm_bwd_gate : dpre[m,2], gradient at the conv3D output (tanh' applied), from dG (source order)
 *   stes_gsm_bwd_wgrad: dw [2][fold/2][27], db [2]  += (atomics)
 *   stes_gsm_dxbn     : gradient at the BatchNorm3d output, ReLU mask applied, bf16 [M][fold]
 *                       (then stes_bn_bwd_reduce / _finalize with D = dxbn, raw1 = Xc)
 *   stes_gsm_bwd_dx   : dXf = dG + gate*(dG[t-+1] - dG) + alpha*dxbn + beta*x + delta
 *   stes_gsm_forward  : gate AND G in one kernel (_GSM.forward :103-130 whole): the gate conv runs on MFMA (channels first,
 *                       the 3x3 spatial taps as a gather sum in LDS), the shift follows in the same workgroup.  fold >= 32.
 *   stes_gsm_bwd_conv : backward of the gate conv in one kernel: dxbn (as stes_gsm_dxbn), the conv3D weight/bias gradient
 *                       (as stes_gsm_bwd_wgrad) and the BatchNorm3d-backward sums s_dy += sum dxbn, s_dyr += sum dxbn * x
 *                       (as stes_bn_bwd_reduce on dxbn, Xc).  Slices of 8 and 16 channels: dxbn and the sums are one kernel
 *                       and the conv3D gradients a second one behind it, skipped when dw and db are NULL (a caller that runs
 *                       stes_gsm_bwd_wgrad on another stream); ws is not used.  fold >= 32: dw and db are required; ws: NULL,
 *                       or stes_gsm_bwd_conv_ws_floats(fold)
 *                       floats that are ZERO on entry (they are zero again on return): the per-workgroup partial sums then go
 *                       to STES_GSM_WS_REPLICAS copies and a second small launch adds them up -- a thousand workgroups adding
 *                       into the same few KiB serialise in the memory-side atomic units (6x the kernel's time without ws).  */
#define STES_GSM_WS_REPLICAS 64
typedef struct { int32_t B, T, H, W, fold; } stes_gsm_geom;
int stes_gsm_prep(const float* w3d, uint16_t* wt, int fold, void* stream);
int stes_gsm_gate(const uint16_t* Xc, const float* scale, const float* shift, const uint16_t* wt, const float* b3d,
                  float* gate, const stes_gsm_geom* g, void* stream);
int stes_gsm_shift(const uint16_t* Xc, const float* gate, uint16_t* G, const stes_gsm_geom* g, void* stream);
int stes_gsm_bwd_gate(const uint16_t* dG, const uint16_t* Xc, const float* gate, float* dpre,
                      const stes_gsm_geom* g, void* stream);
int stes_gsm_bwd_wgrad(const float* dpre, const uint16_t* Xc, const float* scale, const float* shift, float* dw,
                       float* db, const stes_gsm_geom* g, void* stream);
int stes_gsm_dxbn(const float* dpre, const uint16_t* Xc, const float* scale, const float* shift, const float* w3d,
                  uint16_t* dxbn, const stes_gsm_geom* g, void* stream);
int stes_gsm_bwd_dx(const uint16_t* dG, const uint16_t* Xc, const float* gate, const uint16_t* dxbn,
                    const float* alpha, const float* beta, const float* delta, uint16_t* dXf, const stes_gsm_geom* g,
                    void* stream);
int stes_gsm_forward(const uint16_t* Xc, const float* scale, const float* shift, const uint16_t* wt, const float* b3d,
                     float* gate, uint16_t* G, const stes_gsm_geom* g, void* stream);
int stes_gsm_bwd_conv(const float* dpre, const uint16_t* Xc, const float* scale, const float* shift, const float* w3d,
                      uint16_t* dxbn, float* s_dy, float* s_dyr, float* dw, float* db, float* ws, const stes_gsm_geom* g,
                      void* stream);
int stes_gsm_bwd_conv_ws_floats(int fold);
/* per-column sum / sum of squares of X[:, :ncols] (atomics) and/or a compact copy [M][ncols] of that slice */
int stes_col_stats(const uint16_t* X, int64_t ldx, int ncols, int64_t M, float* sum, float* sumsq, uint16_t* copy,
                   void* stream);
/* TemporalShift.shift, reference model/shift.py:37-52 (out of place; backward = adjoint shift). */
int stes_tsm_shift(const uint16_t* X, uint16_t* out, int fold, int C, int B, int T, int H, int W, int backward, void* stream);
/* The same shift in place, as the reference's module does it (model/shift.py:55-66: the block input itself is shifted, so the
 * shortcut and the ReLU mask of the layer in front see the shifted tensor).  Only the 2 * fold channels that move are touched.
 * fold and C multiples of 4. */
int stes_tsm_shift_inplace(uint16_t* X, int fold, int C, int B, int T, int H, int W, int backward, void* stream);

/* ---------------------------------------------------- temporal head, loss, optimiser ----
 * Bidirectional GRU layer (torch.nn.GRU inside GRUPrediction, reference model/modules.py:20-36).
 * gi = x W_ih^T + b_ih for both directions comes from stes_gemm ([B,T,2,3H] bf16).  With `saved` != NULL forward keeps
 * what the backward needs; backward leaves dgi (for dW_ih, db_ih, dx) and dgh (for dW_hh, db_hh with hprev_bf).
 * hstate / dhstate are workspaces of ws_floats floats (0 means 4*B*H).  With at least stes_gru_ws_floats(B, H, backward)
 * floats, B <= 16 and H <= 768 the whole recurrence is ONE launch: every workgroup keeps its W_hh rows in registers for
 * all T steps and the hidden state travels between workgroups through the workspace (write-through stores, a step
 * counter, no cache write-back); otherwise it is T launches.  Both forms give the same results bit for bit.     */
typedef struct {
    const uint16_t* gi; const uint16_t* whh; const uint16_t* whhT; const float* bhh;
    uint16_t* y; float* hstate; float* saved; uint16_t* hprev_bf;
    const float* dy; int64_t ld_dy; const uint16_t* mask;
    uint16_t* dgi; uint16_t* dgh; float* dhstate;
    int32_t B, T, H;
    int64_t ws_floats;
} stes_gru_args;
int stes_gru_forward(const stes_gru_args* a, void* stream);
int stes_gru_backward(const stes_gru_args* a, void* stream);
int64_t stes_gru_ws_floats(int B, int H, int backward);
/* The one-launch recurrences wait for their peer workgroups with a bounded spin.  A wait that runs out (the workgroups of
 * the launch were not resident together, e.g. on a GPU shared with other work) does not go unnoticed: the workgroup fills
 * everything it produces from then on with NaN and raises a word of pinned host memory, and the NEXT stes_gru_* call of the
 * process returns STES_ERR_LAUNCH (once) instead of launching.  stes_gru_set_spin_limit changes the bound (polls of about
 * 64 cycles each, default 2^20; 0 makes every wait run out, satisfied or not: a diagnostic aid). */
int stes_gru_set_spin_limit(int limit);
/* stes_gru_forward for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and results as stes_gru_forward,
 * except that a->hstate is ignored (may be NULL): the fp32 hidden-state ping-pong buffer is a 1 MiB array inside the
 * library, one copy per GPU, so B*H <= 65536 (else STES_ERR_UNSUPPORTED).  Calls no hipMalloc/hipFree.  All calls on one
 * GPU share that array: calls that overlap on one GPU (two streams, two threads) race; calls on one stream do not. */
int stes_gru_forward_noalloc(const stes_gru_args* a, void* stream);
/* Resolves the address of that array on the current device (what the first stes_gru_forward_noalloc call on a device
 * does otherwise).  Address resolution is not a capturable stream operation: call this once per device before a HIP
 * graph capture whose first use of stes_gru_forward_noalloc would be inside the capture.  No counterpart in the reference. */
int stes_gru_noalloc_prepare(void);
/* MinGRU recurrence, reference model/min_gru.py:76-92; pre = [gate pre-activation | candidate]; h0 [H] is the layer's
 * frozen initial-state parameter (:42,60; NULL = zeros). */
int stes_mingru_scan_fwd(const uint16_t* pre, const float* h0, uint16_t* h, int B, int T, int H, void* stream);
int stes_mingru_scan_bwd(const uint16_t* pre, const float* h0, const uint16_t* h, const float* dh, int64_t ld_dh, uint16_t* dpre,
                         int B, int T, int H, void* stream);
/* nn.Dropout of GRUPrediction (model/modules.py:32): out = x*mask, mask in {0, 1/(1-p)} from a counter hash. */
int stes_dropout(const uint16_t* x, uint16_t* out, uint16_t* mask, int64_t n, float p, uint32_t seed, void* stream);
/* Weighted-mean cross entropy + masked L1 on sigmoid(loc) (train_e2e_spatial.py:566-593) and their
 * gradients (bf16, rows padded to 8 columns).  out_loss = {sum, cls, loc, sum of class weights,
 * number of event frames}.  ext_denoms (optional) replaces the two denominators, e.g. by their
 * sums over data-parallel ranks.                                                          */
int stes_loss(const float* logits, int64_t ldl, const float* loc, int64_t ldloc, const int64_t* label, const float* xy,
              int M, int C, float fg_weight, float grad_scale, const float* ext_denoms, float* out_loss,
              uint16_t* dlogits, uint16_t* dloc, void* stream);
/* torch.optim.AdamW update over a flat fp32 range (model/common.py:40, step() :61-75). */
int stes_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
               float weight_decay, int step, float grad_mul, void* stream);
/* fp32 [rows][cols] -> bf16 [dst_rows][ld], optionally transposed, zero padded; perm_fold > 0 reads the first
 * perm_fold source columns through the gated-shift channel interleave (see the temporal shift section). */
int stes_cast_bf16(const float* src, uint16_t* dst, int rows, int cols, int ld, int dst_rows, int transpose, int perm_fold,
                   void* stream);
/* The same for a whole table of copies in ONE launch.  `table` is DEVICE memory holding n entries; entry e owns the
 * workgroups [block0, block0 + ceil(dst_rows*ld / STES_CAST_CHUNK)) and the entries are sorted by block0 (entry 0 has
 * block0 = 0); total_blocks is the sum.  Field meaning as in stes_cast_bf16.                                            */
#define STES_CAST_CHUNK 2048
typedef struct {
    const float* src; uint16_t* dst;
    int32_t rows, cols, ld, dst_rows, transpose, perm_fold;
    int32_t block0, reserved;
} stes_cast_desc;
int stes_cast_bf16_multi(const stes_cast_desc* table, int n, int total_blocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif
