// Internal launchers of the fsvit gfx950 kernels.  Every kernel source that touches 16-bit activations is compiled twice
// (fsvit_common.h): the same declarations exist in namespace fsvit (bf16) and namespace fsvit_f16 (fp16).
#pragma once
#include <hip/hip_runtime.h>
#include "conv_gemm.h"

#define FSVIT_DECL_NS fsvit
#include "kernels_decl.inc"
#undef FSVIT_DECL_NS
#define FSVIT_DECL_NS fsvit_f16
#include "kernels_decl.inc"
#undef FSVIT_DECL_NS
