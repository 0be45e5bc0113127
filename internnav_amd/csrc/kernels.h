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
using PatchifyArgs = ina_patchify_args;
using Embed3Args = ina_embed3_args;
using Head3Args = ina_head3_args;
using SeqpoolArgs = ina_seqpool_args;
using SelectArgs = ina_select_args;
using PoolActArgs = ina_pool_act_args;
using GatherArgs = ina_gather_args;
using RopeArgs = ina_rope_args;
using MropeTableArgs = ina_mrope_table_args;
using ArgmaxArgs = ina_argmax_args;
using DitAttnArgs = ina_dit_attn_args;
using ResizeU8Args = ina_resize_u8_args;
using QwenPatchifyArgs = ina_qwen_patchify_args;
using U8LutArgs = ina_u8_lut_args;
using ResizeF32Args = ina_resize_f32_args;
using DitRowchainArgs = ina_dit_rowchain_args;
using GnMishArgs = ina_gn_mish_args;
using PadRowsArgs = ina_pad_rows_args;
using DdimStepArgs = ina_ddim_step_args;
using EwArgs = ina_ew_args;
using ColsumArgs = ina_colsum_args;
using NormBwdArgs = ina_norm_bwd_args;
using TransposeArgs = ina_transpose_args;
using SparseRowsArgs = ina_sparse_rows_args;
using SmallLinearArgs = ina_small_linear_args;
using MseArgs = ina_mse_args;
using AdamwArgs = ina_adamw_args;
using GemmNnArgs = ina_gemm_nn_args;
using AttnBwdArgs = ina_attn_bwd_args;

int ina_launch_gemm(const GemmArgs& p, hipStream_t stream);
int ina_plan_gemm(const GemmArgs& p_in, GemmArgs& p, int& kernel);   // validation + kernel selection only (no launch)
int ina_launch_gemm_skinny_prenorm(const GemmArgs& p, hipStream_t stream);   // M <= 16 with the input RMSNorm fused in front (gemm_skinny.hip)
int ina_launch_gemm_glds(const GemmArgs& p, hipStream_t stream, int cfg);  // direct-to-LDS staged large-K path
int ina_launch_gemm_skinny_fused(const GemmArgs& p, hipStream_t stream);  // M <= 64 weight-streaming, workgroup owns its columns for all K, fused epilogue
int ina_launch_gemm_rowpanel(const GemmArgs& p, hipStream_t stream, int cfg);  // gemm_rowpanel.hip: K = 384, register-resident row panels (cfg 34 / 35)
bool ina_gemm_rowpanel_contract(const GemmArgs& p);
bool ina_gemm_w4_contract(const GemmArgs& p);
int ina_launch_gemm_preshuffle(const void* W, void* Wp, int N, int K, long ldw, hipStream_t stream);   // gemm_w4.hip: W -> MFMA fragment order (cfg 40)
int ina_launch_gemm_w4(const GemmArgs& p, hipStream_t stream, int cfg);        // gemm_w4.hip: 256 x 256 tile on four waves of 128 x 128 (cfg 39 / 40)
int ina_launch_attention(const AttnArgs& p, hipStream_t stream);
bool ina_attention_wide_eligible(const AttnArgs& p);                   // attention_wide.hip: long dense shapes (32 query rows per wave) - the automatic rule
bool ina_attention_wide_contract(const AttnArgs& p);                   // what that kernel can run at all
int ina_launch_attention_wide(const AttnArgs& p, hipStream_t stream);
int ina_launch_norm(const NormArgs& p, hipStream_t stream);
int ina_launch_patchify(const PatchifyArgs& p, hipStream_t stream);
int ina_launch_embed3(const Embed3Args& p, hipStream_t stream);
int ina_launch_head3(const Head3Args& p, hipStream_t stream);
int ina_launch_seqpool(const SeqpoolArgs& p, hipStream_t stream);
int ina_launch_select(const SelectArgs& p, hipStream_t stream);
int ina_launch_pool_act(const PoolActArgs& p, hipStream_t stream);
int ina_launch_gather(const GatherArgs& p, hipStream_t stream);
int ina_launch_rope(const RopeArgs& p, hipStream_t stream);
int ina_launch_mrope_table(const MropeTableArgs& p, hipStream_t stream);
int ina_launch_argmax(const ArgmaxArgs& p, hipStream_t stream);
int ina_launch_resize_u8(const ResizeU8Args& p, hipStream_t stream);            // one axis of PIL's 8-bit bicubic resample
int ina_launch_qwen_patchify_u8(const QwenPatchifyArgs& p, hipStream_t stream);  // HF Qwen2-VL rescale + normalize + patchify from bytes
int ina_launch_u8_lut(const U8LutArgs& p, hipStream_t stream);
int ina_launch_resize_f32(const ResizeF32Args& p, hipStream_t stream);          // one axis of PIL's float ("F" mode) bicubic resample
int ina_launch_dit_attention(const DitAttnArgs& p, hipStream_t stream);  // q/k-LayerNorm + self-attention + gated cross-attention of a NextDiT block
int ina_launch_dit_rowchain(const DitRowchainArgs& p, hipStream_t stream);  // GEMM + gated rmsnorm + residual + next pre-norm + next GEMM of a NextDiT block, one launch
int ina_launch_gn_mish(const GnMishArgs& p, hipStream_t stream);      // GroupNorm + Mish (+ FiLM, + residual) of a ConditionalUnet1D block
int ina_launch_pad_rows(const PadRowsArgs& p, hipStream_t stream);
int ina_launch_ddim_step(const DdimStepArgs& p, hipStream_t stream);
// SFT step (train.hip, attention_bwd.hip)
int ina_launch_ew(const EwArgs& p, hipStream_t stream);
int ina_launch_colsum(const ColsumArgs& p, hipStream_t stream);
int ina_launch_norm_bwd(const NormBwdArgs& p, hipStream_t stream);
int ina_launch_transpose(const TransposeArgs& p, hipStream_t stream);
int ina_launch_sparse_rows(const SparseRowsArgs& p, hipStream_t stream);
int ina_launch_small_linear(const SmallLinearArgs& p, hipStream_t stream);
int ina_launch_mse(const MseArgs& p, hipStream_t stream);
int ina_launch_adamw(const AdamwArgs& p, hipStream_t stream);
int ina_launch_gemm_nn(const GemmNnArgs& p, hipStream_t stream);
int ina_launch_gemm_dw(const ina_gemm_dw_args& p, hipStream_t stream);
int ina_launch_attention_bwd(const AttnBwdArgs& p, hipStream_t stream);
