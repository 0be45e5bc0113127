// Internal launch API of the gfx950 kernels (host side). Everything here takes raw device pointers and an
// explicit hipStream_t; no torch types. The argument structs are the public C-ABI structs (include/internnav_amd.h);
// always value-initialise them (`GemmArgs p{};`) - zero means "absent"/"default" for every field.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/internnav_amd.h"

using GemmArgs = ina_gemm_args;
using AttnArgs = ina_attn_args;
using NormArgs = ina_norm_args;

int ina_launch_gemm(const GemmArgs& p, hipStream_t stream);
int ina_launch_attention(const AttnArgs& p, hipStream_t stream);
int ina_launch_norm(const NormArgs& p, hipStream_t stream);
