// Grouped 3x3 convolution with stride 2, forward (stes_conv_ring.hip): an image's input rows walk through a ring of row slots in
// LDS and every output row is convolved from three of them; called by stes_conv3x3g_fwd for the map widths it is built for.
#pragma once
#include <stdint.h>

bool stes_conv_ring_takes(int mode, int stride, int Hi, int Wi, bool pooled);
// Returns STES_OK or a negative error code.
// With pool_out: the squeeze-excite squeeze of every image (mean of relu(pool_scale * out + pool_shift), bf16 [N][C]) instead of the sums.
int stes_conv_ring_launch(const uint16_t* in, const uint16_t* wp, uint16_t* out, const float* scale, const float* shift, float* col_sum,
                          float* col_sumsq, const float* pool_scale, const float* pool_shift, uint16_t* pool_out, float pool_norm, int N,
                          int Hi, int Wi, int C, void* stream);

// The input gradient of the same convolution (Hi x Wi = the gradient's map), with the ReLU mask and BatchNorm sums of
// stes_conv3x3g_dgrad when mask_raw is given.
bool stes_dgrad_ring_takes(int mode, int stride, int Hi, int Wi);
int stes_dgrad_ring_launch(const uint16_t* dy, const uint16_t* wp, uint16_t* dx, const uint16_t* mask_raw, const float* mscale,
                           const float* mshift, float* s_dy, float* s_dyr, int N, int Hi, int Wi, int C, void* stream);

// Stride 1, forward (mode 0: in -> out with the input transform and the output's sums) and input gradient (mode 1: in = dY,
// out = dX, with the mask and the BatchNorm backward sums when mask_raw is given), maps 28, 14 or 7 wide of any height.
bool stes_conv_ring1_takes(int mode, int stride, int Wi, bool pooled);
int stes_conv_ring1_launch(int mode, const uint16_t* in, const uint16_t* wp, uint16_t* out, const uint16_t* mask_raw, const float* scale,
                           const float* shift, float* sum_a, float* sum_b, const float* pool_scale, const float* pool_shift,
                           uint16_t* pool_out, float pool_norm, int N, int H, int W, int C, void* stream);

// The weight gradient at stride 1 on maps 28, 14 or 7 wide (dw [C][16][3][3] fp32, added to): each input pixel fetched once.
bool stes_wgrad_ring_takes(int stride, int Wi);
int stes_wgrad_ring_launch(const uint16_t* dout, const uint16_t* in, const float* scale, const float* shift, float* dw, int N, int H, int W,
                           int C, void* stream);
