// Column-strip form of the stride-1 forward grouped 3x3 convolution (stes_conv_strip.hip), called by the entry points in
// stes_conv.hip for the feature maps it takes.
#pragma once
#include <stdint.h>

bool stes_conv_strip_takes(int mode, int stride, int Wi);
// Hi x Wi: the image (input and output have the same size).  Returns STES_OK or a negative error code.
int stes_conv_strip_launch(const uint16_t* src, const uint16_t* wp, uint16_t* dst, const float* scale, const float* shift, float* col_sum,
                           float* col_sumsq, int N, int Hi, int Wi, int C, void* stream, const float* pool_scale, const float* pool_shift,
                           uint16_t* pool_out, float pool_norm);
