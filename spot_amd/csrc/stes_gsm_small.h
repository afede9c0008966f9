// Gated shift on the narrow channel slices of the high-resolution stages (fold 8 and 16): forward in one kernel, and the
// gradient at the BatchNorm3d output with its backward sums in one kernel (stes_gsm_small.hip); called by the entry points
// in stes_gsm.hip.
#pragma once
#include <stdint.h>

// Returns STES_OK or a negative error code (STES_ERR_UNSUPPORTED when a frame row does not fit the workgroup's tile).
int stes_gsm_small_forward(const uint16_t* Xc, const float* scale, const float* shift, const uint16_t* wt, const float* b3d, float* gate,
                           uint16_t* G, int B, int T, int H, int W, int fold, void* stream);
int stes_gsm_small_dxbn(const float* dpre, const uint16_t* Xc, const float* scale, const float* shift, const float* w3d, uint16_t* dxbn,
                        float* s_dy, float* s_dyr, int B, int T, int H, int W, int fold, void* stream);
