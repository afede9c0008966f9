// Streaming GEMM (stes_gemm_stream.hip) as seen by the stes_gemm dispatcher.
#pragma once
#include "stes_common.h"

struct StreamDivs { FastDiv hw, wo, gate; };

// Column slice (16*NI) the streaming kernel would use for this product, 0 when it does not take it.
int stes_stream_slice(const stes_gemm_args* a);
// Launches the streaming kernel when the product is one of its shapes: 1 launched, 0 not taken.
int stes_gemm_stream(const stes_gemm_args* a, void* stream);
