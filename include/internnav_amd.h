/* internnav_amd - C ABI of the MI355X (gfx950) InternVLA-N1 / NavDP policy engine.
 *
 * The reference (InternRobotics/InternNav) is 100 % Python: its hot path reaches the GPU only through
 * torch / transformers / diffusers / flash-attn, so there is no reference FFI to mirror.
 * This header is the boundary a maintainer binds instead (ctypes stub in INTEGRATION.md): plain pointers,
 * sizes and a hipStream_t; the caller (PyTorch-ROCm) owns every tensor, the library owns only its workspace.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; ina_last_error() gives the message (thread-local).
 *  - all device pointers are HBM addresses on the current HIP device; `stream` is a hipStream_t passed as void*.
 *  - bf16 = bfloat16 (uint16 storage); "f32" params (bias, norm weights, gates) are float32.
 *  - functions are stream-ordered and never synchronise the device.
 */
#ifndef INTERNNAV_AMD_H
#define INTERNNAV_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INA_ABI_VERSION 1

/* activation codes (GEMM epilogue) */
#define INA_ACT_NONE_C 0
#define INA_ACT_GELU_ERF_C 1
#define INA_ACT_GELU_TANH_C 2
#define INA_ACT_RELU_C 3
#define INA_ACT_SILU_C 4
/* dtype codes */
#define INA_BF16 0
#define INA_F32 1

int ina_abi_version(void);
const char* ina_last_error(void);
/* Fails (non-zero) unless a gfx950 device is current; fills name[0..n) with the arch string. */
int ina_device_check(char* name, int n);

/* ---- C[M,N] = epilogue(A[M,K] . W[N,K]^T): replaces every nn.Linear / patch-embed conv on the path
 *      (reference: torch.nn.Linear call sites, e.g. dinov2_layers/attention.py:46-48, mlp.py:27-29,
 *       navdp.py:94-100, internvla_n1_arch.py:129-134; transformers Qwen2.5-VL q/k/v/o/mlp projections). */
typedef struct ina_gemm_args {
    const void* A;          /* bf16 [M,K], row stride lda (elements) */
    const void* W;          /* bf16 [N,K], row stride ldw */
    void* C;                /* bf16|f32 [M,N] (N/2 columns in glu mode), row stride ldc */
    const float* bias;      /* f32 [N] or NULL */
    const float* colscale;  /* f32 [N] or NULL (LayerScale gamma) */
    const float* rowscale;  /* f32 [M / rowscale_div] or NULL */
    const void* R;          /* residual [M,N] or NULL, dtype res_dtype, row stride ldr */
    int32_t M, N, K;
    int32_t lda, ldw, ldc, ldr;
    int32_t act;            /* INA_ACT_* */
    int32_t out_dtype;      /* INA_BF16 | INA_F32 */
    int32_t res_dtype;
    int32_t glu;            /* 1: W rows interleaved [gate16|up16], C = act(gate) * up */
    int32_t rowscale_div;   /* 0 means 1 */
    int32_t batch;          /* 0 means 1; grid.y batches with the element strides below */
    int64_t strideA, strideW, strideC, strideR;
    int32_t force_cfg;      /* 0 = auto tile selection */
    int32_t _pad;
} ina_gemm_args;
int ina_gemm_bf16(const ina_gemm_args* args, void* stream);

/* ---- flash-style attention forward (bf16, fp32 softmax): replaces flash_attn / SDPA / nn.MultiheadAttention
 *      (reference call sites: dinov2_layers/attention.py:49-62, navdp.py:57-66,192, navdp_backbone.py:77,148,
 *       nextdit_traj.py:147-165; transformers Qwen2.5-VL vision/text attention). Strides in elements. */
typedef struct ina_attn_args {
    const void* Q;
    const void* K;
    const void* V;
    void* O;
    int64_t q_bs, q_rs, q_hs;
    int64_t k_bs, k_rs, k_hs;
    int64_t v_bs, v_rs, v_hs;
    int64_t o_bs, o_rs, o_hs;
    int32_t B, H, Hkv;
    int32_t Lq, Lk;
    int32_t D;              /* 48 | 64 | 80 | 128 */
    int32_t causal;
    int32_t kv_start;       /* keys < kv_start masked */
    int32_t kv_bdiv;        /* K/V batch = b / kv_bdiv (0 means 1) */
    float scale;
    const int32_t* cu_q;    /* int32 [B+1] varlen offsets or NULL */
    const int32_t* cu_k;
    const float* head_gate; /* f32 [H] or NULL: O *= tanh(gate[h]) */
    int32_t accumulate;     /* 1: O += result */
    int32_t _pad;
} ina_attn_args;
int ina_attention_bf16(const ina_attn_args* args, void* stream);

/* ---- LayerNorm / RMSNorm (+ residual-in, + NextDiT modulation / tanh gate); reference: nn.LayerNorm call sites
 *      (dinov2_layers/block.py:83-87, navdp.py:78,193), diffusers RMSNorm / LuminaRMSNormZero (nextdit_traj.py:109-119). */
typedef struct ina_norm_args {
    const void* X;
    const void* R;
    void* Y;
    void* S;
    const float* gamma;
    const float* beta;
    const float* mod_scale;
    const float* gate;
    const void* G;
    int32_t rows, C;
    int32_t ldx, ldr, ldy, ldg;
    int32_t mod_div, mod_ld;
    int32_t rms;
    float eps;
} ina_norm_args;
int ina_norm_bf16(const ina_norm_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INTERNNAV_AMD_H */
