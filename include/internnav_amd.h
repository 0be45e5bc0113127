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

#define INA_ABI_VERSION 8

/* activation codes (GEMM epilogue) */
#define INA_ACT_NONE_C 0
#define INA_ACT_GELU_ERF_C 1
#define INA_ACT_GELU_TANH_C 2
#define INA_ACT_RELU_C 3
#define INA_ACT_SILU_C 4
#define INA_ACT_MISH_C 5
/* dtype codes */
#define INA_BF16 0
#define INA_F32 1

int ina_abi_version(void);
const char* ina_last_error(void);
/* Fails (non-zero) unless a gfx950 device is current; fills name[0..n) with the arch string. */
int ina_device_check(char* name, int n);
/* sizeof() of the k-th argument struct below (0 gemm, 1 attn, 2 norm, 3 patchify, 4 embed3, 5 head3, 6 seqpool, 7 select, 8 pool_act, 9 gather, 10 rope, 11 mrope_table, 12 argmax, 13 dit_attn, 14 resize_u8, 15 qwen_patchify, 16 u8_lut, 17 resize_f32, 18 gn_mish, 19 pad_rows, 20 ddim_step, 21 ew, 22 colsum, 23 norm_bwd, 24 transpose, 25 sparse_rows, 26 small_linear, 27 mse, 28 adamw, 29 gemm_nn, 30 attn_bwd, 31 dit_rowchain):
 * lets a binding verify its struct mirrors against the compiled layout. */
int ina_struct_size(int k);
/* Per-launch timing for the benchmark's roofline line: while enabled every launch is bracketed by a hipEvent pair on its
 * stream and its algorithmic FLOPs / bytes are tallied per kernel class (0 tiled MFMA gemm, 1 attention, 2 norm, 3 elementwise,
 * 4 weight-streaming skinny gemm (M <= 64, HBM-bound)).
 * ina_prof_enable(0|1) also clears the tally; ina_prof_read synchronises on the recorded events. Eager launches only. */
/* Library-owned scratch (split-K partials of the skinny GEMM, flash-decoding partials) lives in numbered slots [0, 8). Launches
 * use the slot current at issue time (default 0) and a captured graph keeps it: graphs that may replay concurrently on different
 * streams must be captured under different slots.
 * The current slot is per host thread. A buffer that a captured graph has seen is never freed: if a later launch under the same
 * slot needs more, a new buffer serves new launches and the old one is retired but stays allocated (ina_workspace_retired() counts
 * them), so graphs captured earlier keep replaying into valid memory. */
int ina_set_workspace_slot(int slot);
int ina_workspace_retired(void);
int ina_prof_enable(int on);
int ina_prof_read(int kind, double* ms_total, int64_t* launches, double* flops, double* bytes);
/* the same tally restricted to one kernel of the class: sub = GEMM tile config id (18 = gemm_bf16_pp_kernel<256,256,4>, 21 = <192,256,4>,
 * 22 = gemm_bf16_glds_kernel<128,128,2,2,1>, 33 = gemm_bf16_glds_kernel<256,256,4,4,2>, 11 / 14 / 26 / 27 other LDS-DMA tiles, 1-5 gemm_bf16_nt_kernel tiles, 34 / 35 = gemm_bf16_rowpanel_kernel<8|4 waves>, 39 = gemm_bf16_w4_kernel<256> (four-wave 256x256 tile), 40 = gemm_bf16_w4p_kernel<256> (the same tile on fragment-ordered weights), 42 = dit_rowchain_kernel, 43 = gemm_dw_kernel) */
int ina_prof_read_sub(int kind, int sub, double* ms_total, int64_t* launches, double* flops, double* bytes);

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
    int32_t force_cfg;      /* 0 = auto tile selection; -1 = auto for a launch that shares the device with another stream's GEMMs (tile quantisation not charged); > 0 = that tile config */
    int32_t group_m;        /* tile order of the LDS-DMA kernels: 0 = auto, 1 = row-major, n > 1 = groups of n row-tiles (L2 locality) */
    /* fused input RMSNorm (M <= 16, the single-token decode passes of the LLM: transformers Qwen2RMSNorm in front of q/k/v and gate/up):
     * C = epilogue(bf16(A * rsqrt(mean(A^2) + norm_eps) * norm_gamma) . W^T) with A of dtype a_dtype - the separate norm launch and its
     * bf16 round trip disappear (every workgroup normalises the <= 16 rows into LDS while its first weight tiles are in flight) */
    const float* norm_gamma; /* f32 [K] or NULL (no fused norm; A is bf16) */
    float norm_eps;
    int32_t a_dtype;        /* dtype of A when norm_gamma is set: INA_BF16 | INA_F32 */
    /* LayerNorm statistics of the produced rows (row-panel kernels only: K = 384, plain epilogue, N a multiple of 384, M >= 16384 rows): seg_stats
     * f32 [M, N / 384, 2] receives (mean, 1 / sqrt(var + seg_eps)) of every 384-wide segment of C's fp32 row - what ina_dit_attention(stats=)
     * consumes (the fused q1|k1|v1|q2 projection of a NextDiT block). Refused, not ignored, where another kernel would run. */
    float* seg_stats;
    float seg_eps;
    int32_t _pad_seg;
    const void* Wp;         /* NULL, or the same W in MFMA fragment order (ina_gemm_preshuffle; N % 16 == 0, K % 32 == 0): tile config 40 takes its
                             * B fragments from it straight into registers (selected where cfg 39 would run; bit-equal results) */
} ina_gemm_args;
int ina_gemm_bf16(const ina_gemm_args* args, void* stream);
/* W bf16 [N,K] (row stride ldw) -> Wp bf16 [N*K]: fragment (n / 16, k / 32) is one contiguous KiB, element (n, k) at
 * ((n / 16) * (K / 32) + k / 32) * 512 + ((k % 32) / 8 * 16 + n % 16) * 8 + k % 8. Done once per weight at load time. */
int ina_gemm_preshuffle(const void* W, void* Wp, int32_t N, int32_t K, int64_t ldw, void* stream);
/* Which kernel ina_gemm_bf16 would run for these arguments - validation and tile selection only, nothing is launched and no GPU is needed
 * (the selection is host arithmetic on M / N / K, the epilogue and force_cfg): *kernel = 1-5 register-staged tiles, 11-27 / 33 LDS-DMA tiles
 * (18 = 256x256 ping-pong, 21 = 192x256 ping-pong, 22 / 26 / 27 single-buffer tiles of the d = 384 heads, 33 = 256x256 with 16 waves),
 * 34 / 35 row-panel kernels (K = 384), 39 / 40 the 256x256 tile on four waves of 128x128 (39 is selected for wide no-residual K = 2048 .. 4096 GEMMs, 40 instead of it when Wp is given),
 * 30 = weight streaming with the fused input RMSNorm, 32 = weight streaming (M <= 64). Returns non-zero (and sets ina_last_error)
 * exactly when ina_gemm_bf16 would reject the arguments. */
int ina_gemm_select(const ina_gemm_args* args, int* kernel);

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
    const int32_t* k_len;   /* int32 [B / kv_bdiv] or NULL: valid keys per K/V batch (dense mode; Lk is then the maximum) */
    int32_t accumulate;     /* 1: O += result */
    /* training only - attention-probability dropout of nn.MultiheadAttention(dropout=p) in train mode (dense layouts, no decode path):
     * P[b,h,q,k] is kept iff ina_hash(drop_seed, ((b*H + h)*Lq + q)*Lk + k) >= drop_thresh and scaled by drop_scale = 1 / (1 - p);
     * drop_thresh = p * 2^32, 0 = off. The backward kernel regenerates the same mask from the same numbers. */
    uint32_t drop_seed;
    uint32_t drop_thresh;
    float drop_scale;
    int32_t kernel;         /* (decode shapes - few query rows against a long dense KV - have their own kernels: 0 = one launch (d = 128: eight waves,
                             * K fragments straight from global memory; otherwise four waves), 1 = the split + combine pair, 3 = the four-wave one-launch kernel.)
                             * 0 = automatic (long dense shapes run the 32-rows-per-wave kernel of attention_wide.hip, everything else the
                             * 16-rows-per-wave kernel), 1 = the 16-rows-per-wave kernel, 2 = the 32-rows-per-wave kernel (an error outside its
                             * contract: d 64 / 80 / 128, Lq and Lk >= 128 - or packed d-80 sequences of <= 64 tokens, the Qwen ViT windows - no head gate /
                             * accumulate / dropout). Same result up to the bf16
                             * rounding of P and O; parity tests pin both and compare them. */
    int32_t _pad0;
    const uint32_t* drop_salt; /* training, optional: one device word ADDED to drop_seed when the kernel starts - a launch sequence captured in a
                                * hipGraph draws fresh masks on every replay (the host bumps the word between replays); NULL = drop_seed alone */
    /* single-token decoder passes (d = 128, causal, 256 <= Lk <= 1024, Lq <= 8: the one-launch decode kernel): the rotary embedding of the NEW tokens
     * and their KV-cache append inside the attention launch - what ina_rope_bf16 with KV set does in a launch of its own. Q holds the UN-rotated
     * query heads; k_new / v_new the un-rotated key / raw value heads of the Lq new tokens (element strides kn_bs per sequence, kn_rs per token,
     * kn_hs per kv head - the q|k|v projection buffer); the kernel rotates q in registers, writes rotate(k_new) and v_new into rows
     * len_k - Lq .. len_k - 1 of K / V (the cache) and then attends. Same arithmetic and rounding as the separate launch. NULL: Q / K / V as given. */
    const float* rope_cos;     /* f32 [B * Lq, 128] or NULL */
    const float* rope_sin;
    const void* k_new;
    const void* v_new;
    int64_t kn_bs, kn_rs, kn_hs;
} ina_attn_args;
int ina_attention_bf16(const ina_attn_args* args, void* stream);

/* One-level row map: logical row r -> physical row (r / seg_len) * seg_stride + off + r % seg_len (seg_len 0 = identity).
 * Lets norm / embed kernels drop the ViT cls token, concatenate token groups and scatter into [env, slot] buffers
 * (reference: torch.cat call sites navdp_backbone.py:187,281; navdp_policy.py:162-164; navdp.py:182-185). */
typedef struct ina_rowmap {
    int32_t seg_len, seg_stride, off, _pad;
} ina_rowmap;

/* ---- LayerNorm / RMSNorm (+ NextDiT modulation / tanh gate / base add / positional table); reference: nn.LayerNorm call
 *      sites (dinov2_layers/block.py:83-87, navdp.py:78,193), diffusers RMSNorm / LuminaRMSNormZero (nextdit_traj.py:109-119,146,172-176).
 *      t = norm(X[in_map(r)]) * gamma + beta ; t *= 1 + mod_scale[r/mod_div] ; t *= tanh(gate[r/mod_div]) ; t += G[r] ; t += P[r % p_mod]
 *      -> Y[out_map(r)] (bf16) and/or Y32[out_map(r)] (f32).
 *      Chained pre-norm (Y2 != NULL): Y2[out_map(r)] = norm(t) * gamma2 * (1 + mod_scale2[r/mod_div]) (bf16; same rms / eps) - the
 *      next sub-block's modulated norm of the residual row this call just produced (LuminaNextDiTBlock: norm2 -> ffn_norm1,
 *      ffn_norm2 -> next block's norm1). */
typedef struct ina_norm_args {
    const void* X;          /* bf16|f32 (x_dtype) rows of C, row stride ldx */
    void* Y;                /* bf16 out or NULL */
    void* Y32;              /* f32 out or NULL */
    const float* gamma;     /* f32 [C] or NULL */
    const float* beta;      /* f32 [C] or NULL */
    const float* mod_scale; /* f32 [rows/mod_div, mod_ld] or NULL */
    const float* gate;      /* f32 [rows/mod_div, mod_ld] or NULL */
    const void* G;          /* bf16|f32 (g_dtype) [rows, ldg] base added after gating, or NULL */
    const float* P;         /* f32 [p_mod, C] table or NULL */
    ina_rowmap in_map, out_map;
    int32_t rows, C;
    int32_t ldx, ldy, ldy32, ldg;
    int32_t x_dtype, g_dtype;
    int32_t mod_div, mod_ld;
    int32_t p_mod, rms;
    float eps;
    int32_t ldy2;
    void* Y2;               /* bf16 chained output or NULL */
    const float* gamma2;    /* f32 [C] or NULL */
    const float* mod_scale2;/* f32 [rows/mod_div, mod_ld] or NULL */
} ina_norm_args;
int ina_norm_bf16(const ina_norm_args* args, void* stream);

/* ---- patchify: NHWC frames -> im2col rows of the ViT patch-embed conv, fused with the input normalisation.
 *      out[(i*gh + py)*gw + px, c*ps*ps + y*ps + x] = (img[i, py*ps+y, px*ps+x, c] - mean[c]) * inv_std[c]   (bf16),
 *      columns >= 3*ps*ps are zero padding. C == 1 replicates the channel 3x (depth frames).
 *      reference: navdp_backbone.py:155-181,258-279 + dinov2_layers/patch_embed.py:151-164 (Conv2d 14x14 stride 14). */
typedef struct ina_patchify_args {
    const void* img;        /* f32|bf16 (in_dtype) [n, H, W, C] */
    void* out;              /* bf16 [n * gh * gw, ldo] */
    float mean[3];
    float inv_std[3];
    int32_t n, H, W, C;
    int32_t ps, ldo, in_dtype, _pad;
} ina_patchify_args;
int ina_patchify(const ina_patchify_args* args, void* stream);

/* ---- embed3: Y[out_map(r)] = W[:, 0:3] . X[r, 0:3] + b + P[r % p_mod]   (nn.Linear(3, C) + positional table; X NULL = table fill)
 *      reference: input_embed / point_encoder / action_encoder (navdp_policy.py:160,165; navdp.py:178,190; internvla_n1.py:402-410). */
typedef struct ina_embed3_args {
    const float* X;         /* f32 [rows, 3] or NULL */
    const float* W;         /* f32 [C, 3] */
    const float* b;         /* f32 [C] or NULL */
    const float* P;         /* f32 [p_mod, C] or NULL */
    void* Y;                /* bf16|f32 (out_dtype) */
    ina_rowmap out_map;
    int32_t rows, C, ldy, p_mod;
    int32_t out_dtype, x_div;   /* x row = r / x_div (0 means 1): broadcast one vector to x_div consecutive rows */
} ina_embed3_args;
int ina_embed3(const ina_embed3_args* args, void* stream);

/* ---- head3: final norm + Linear(C, 3) + sampler update, one wave per row.
 *      e = W . (norm(X[r]) * gamma + beta) * (1 + mod_scale[r / mod_div]) ... + b
 *      mode 0: eps_out[r] = e ; mode 1 (DDPM, diffusers DDPMScheduler.step): x0 = clamp((s - c1 e) c0, +-clip),
 *      s <- c2 x0 + c3 s + c4 noise[r] ; mode 2 (FlowMatch Euler): s <- s + c0 e.
 *      reference: navdp_policy.py:167-169,312-315; navdp.py:193-195,247-250; internvla_n1.py:418-431. */
typedef struct ina_head3_args {
    const void* X;          /* bf16|f32 (x_dtype) [rows, ldx] */
    const float* gamma;     /* f32 [C] or NULL */
    const float* beta;      /* f32 [C] or NULL */
    const float* mod_scale; /* f32 [rows/mod_div, mod_ld] or NULL */
    const float* W;         /* f32 [3, C] */
    const float* b;         /* f32 [3] */
    float* sample;          /* f32 [rows, 3], updated in place (modes 1, 2) */
    const float* noise;     /* f32 [rows, 3] or NULL */
    float* eps_out;         /* f32 [rows, 3] or NULL */
    float coef[5];
    float clip;
    float eps;
    int32_t rows, C, ldx, x_dtype;
    int32_t mode, mod_div, mod_ld;
} ina_head3_args;
int ina_head3(const ina_head3_args* args, void* stream);

/* ---- seqpool_head: out[s] = w . mean_t( norm(X[s*T + t]) * gamma + beta ) + b   (critic head, navdp_policy.py:183-184) */
typedef struct ina_seqpool_args {
    const void* X;
    const float* gamma;
    const float* beta;
    const float* w;         /* f32 [C] */
    const float* b;         /* f32 [1] */
    float* out;             /* f32 [nseq] */
    float eps;
    int32_t nseq, T, C, ldx, x_dtype, _pad;
} ina_seqpool_args;
int ina_seqpool_head(const ina_seqpool_args* args, void* stream);

/* ---- pool_act: Y[s, :] = act( mean_{t < T} X[s*T + t, :] + P[s % p_mod, :] )   (T = 1: bias + activation)
 *      reference: masked mean of the caption features + SiLU(temb) (diffusers LuminaCombinedTimestepCaptionEmbedding,
 *      LuminaRMSNormZero as used by nextdit_traj.py:109-119, 355). */
typedef struct ina_pool_act_args {
    const void* X;          /* bf16|f32 (x_dtype) [nseq*T, ldx] */
    const float* P;         /* f32 [p_mod, C] or NULL */
    void* Y;                /* bf16|f32 (out_dtype) [nseq, ldy] */
    int32_t nseq, T, C, ldx, ldy, p_mod;
    int32_t x_dtype, out_dtype, act, _pad;
} ina_pool_act_args;
int ina_pool_act(const ina_pool_act_args* args, void* stream);

/* ---- gather_rows: Y[dst ? dst[r] : r, :] = X[src ? src[r] : r, :]   (bit copy of `row_bytes` per row)
 *      reference: embed_tokens lookup, masked_scatter of image embeds, latent_queries rows (internvla_n1.py:129-172),
 *      the ViT window permutation and its inverse (transformers modeling_qwen2_5_vl.py:434-466). */
typedef struct ina_gather_args {
    const void* X;
    void* Y;
    const int32_t* src;     /* int32 [rows] or NULL */
    const int32_t* dst;     /* int32 [rows] or NULL */
    int64_t ldx_bytes, ldy_bytes;
    int32_t rows, row_bytes; /* row_bytes multiple of 16 */
} ina_gather_args;
int ina_gather_rows(const ina_gather_args* args, void* stream);

/* ---- rope: in-place rotary embedding of `heads` heads of width D starting at column `col0` of each row:
 *      x' = x * cos + rotate_half(x) * sin with per-row tables cos/sin f32 [*, D] (row tab ? tab[r] : r).
 *      reference: apply_rotary_pos_emb_vision / apply_multimodal_rotary_pos_emb (transformers modeling_qwen2_5_vl.py:160-171,557-599). */
typedef struct ina_rope_args {
    void* X;                /* bf16 rows, row stride ldx (elements) */
    const float* cos;
    const float* sin;
    const int32_t* tab;     /* int32 [rows] table row per logical row, or NULL */
    ina_rowmap map;         /* logical row -> physical row of X */
    int32_t rows, heads, D, ldx, col0, _pad;
    /* optional fused KV-cache append (decoder layers: q|k|v rows -> rotate q in place, write rotated k and the raw v of each row into
     * the cache row kv_dst[r]): heads [kv_head0, heads) are the key heads, v_heads value heads follow them in X; the cache row holds
     * [k heads | v heads] contiguously. KV == NULL: plain in-place rope on all heads. */
    void* KV;               /* bf16 cache rows, row stride ldkv, or NULL */
    const int32_t* kv_dst;  /* int32 [rows] cache row per logical row */
    int32_t kv_head0, v_heads, ldkv, _pad2;
} ina_rope_args;
int ina_rope_bf16(const ina_rope_args* args, void* stream);

/* ---- mrope_table: cos/sin [n, D] from the 3-D (t, h, w) position ids of Qwen2.5-VL: frequency f = j % (D/2) uses axis
 *      axis_of[f] (mrope_section [16, 24, 24] -> 0 x16, 1 x24, 2 x24); angle = pos[axis][i] * inv_freq[f].
 *      reference: Qwen2_5_VLRotaryEmbedding.forward + mrope interleave (modeling_qwen2_5_vl.py:525-538,582-590); position ids
 *      from internnav/dataset/rope2d.py:6 (get_rope_index_25). */
typedef struct ina_mrope_table_args {
    const int32_t* pos;     /* int32 [3, n] */
    const float* inv_freq;  /* f32 [D/2] */
    const int32_t* axis_of; /* int32 [D/2] */
    float* cos;             /* f32 [n, D] */
    float* sin;
    int32_t n, D;
} ina_mrope_table_args;
int ina_mrope_table(const ina_mrope_table_args* args, void* stream);

/* ---- argmax_rows: out[r] = argmax_j X[r, j] (first maximum), greedy decoding (HF generate do_sample=False) */
typedef struct ina_argmax_args {
    const float* X;         /* f32 [rows, ldx] */
    int32_t* out;           /* int32 [rows] */
    int32_t rows, n, ldx, _pad;
} ina_argmax_args;
int ina_argmax_rows(const ina_argmax_args* args, void* stream);

/* ---- select_traj: per env, rank the S samples by critic value; neg = the k lowest (ascending), pos = the k highest
 *      (descending); trajectories are cumsum_t(sample * scale).  reference: navdp_policy.py:317-320. */
typedef struct ina_select_args {
    const float* critic;    /* f32 [B, S] */
    const float* sample;    /* f32 [B, S, T, 3] */
    float* neg;             /* f32 [B, k, T, 3] */
    float* pos;             /* f32 [B, k, T, 3] */
    float scale;
    int32_t B, S, T, k, _pad;
} ina_select_args;
int ina_select_traj(const ina_select_args* args, void* stream);

/* ---- frame pre-processing on the device, bit-exact with the host path of the reference:
 *      resize_u8: one axis of PIL's 8-bit ImagingResample (Pillow libImaging/Resample.c): tensor [outer, n_in, inner] -> [outer, n_out,
 *      inner], out = clip8((2^21 + sum_x in[xmin + x] * coefs[xx][x]) >> 22); bounds / coefs are PIL's precompute_coeffs +
 *      normalize_coeffs_8bpc tables (internnav_amd/preprocess.py builds them). Image.resize = W pass, then H pass.
 *      reference: internvla_n1_policy.py:105-116, internvla_n1_agent.py:309-320, HF Qwen2VLImageProcessor.resize. */
typedef struct ina_resize_u8_args {
    const void* in;         /* u8 [outer, n_in, inner] */
    void* out;              /* u8 [outer, n_out, inner] */
    const int32_t* bounds;  /* int32 [n_out, 2] = (xmin, count) */
    const int32_t* coefs;   /* int32 [n_out, ksize] 22-bit fixed point */
    int32_t outer, n_in, n_out, inner, ksize, _pad;
} ina_resize_u8_args;
int ina_resize_u8(const ina_resize_u8_args* args, void* stream);

/*      resize_f32: the same for PIL mode "F" images (the depth frames, internvla_n1_agent.py:313,319): double coefficients, double
 *      accumulation in tap order, one rounding to float (ImagingResampleHorizontal_32bpc / Vertical_32bpc). */
typedef struct ina_resize_f32_args {
    const void* in;         /* f32 [outer, n_in, inner] */
    void* out;              /* f32 [outer, n_out, inner] */
    const int32_t* bounds;  /* int32 [n_out, 2] */
    const double* coefs;    /* f64 [n_out, ksize] */
    int32_t outer, n_in, n_out, inner, ksize, _pad;
} ina_resize_f32_args;
int ina_resize_f32(const ina_resize_f32_args* args, void* stream);

/*      qwen_patchify_u8: HF Qwen2VLImageProcessor rescale + normalize (as a 3 x 256 fp32 table computed with the processor's own
 *      arithmetic) + patchify: rows (grid_h/merge, grid_w/merge, merge, merge), columns (C, tdup copies, ps, ps); bf16 out.
 *      reference: transformers image_processing_qwen2_vl.py (_preprocess / patchify), called at internvla_n1_policy.py:163-165. */
typedef struct ina_qwen_patchify_args {
    const void* img;        /* u8 [n, H, W, 3] */
    void* out;              /* bf16 [n * (H/ps) * (W/ps), ldo] */
    const float* lut;       /* f32 [3, 256] */
    int32_t n, H, W, ps, merge, tdup, ldo, _pad;
} ina_qwen_patchify_args;
int ina_qwen_patchify_u8(const ina_qwen_patchify_args* args, void* stream);

/*      u8_lut: out[i] = bf16(lut[in[i]])  (np.array(img) / 255.0 of the System-1 frames, internvla_n1_agent.py:309-317). */
typedef struct ina_u8_lut_args {
    const void* in;         /* u8 [n] */
    void* out;              /* bf16 [n] */
    const float* lut;       /* f32 [256] */
    int64_t n;
} ina_u8_lut_args;
int ina_u8_lut(const ina_u8_lut_args* args, void* stream);

/* ---- dit_rowchain (round 5): everything of a NextDiT block between two attention stages that is local to a row, in one launch:
 *          P  = A . W1^T (bf16)                                     attn2.to_out (K1 = 384) / feed_forward.linear_2 (K1 = 1024)
 *          X += tanh(gate[r/mod_div]) * rmsnorm(P) * gamma1         norm2 / ffn_norm2 + gate + residual (fp32, in place)
 *          H  = rmsnorm(X) * gamma2 * (1 + mod_scale2[r/mod_div])   ffn_norm1 / the next block's norm1 + adaLN scale
 *          C2 = H . W2^T  (glu2: silu(H . Wg^T) * (H . Wu^T), W2 rows interleaved [gate16 | up16])
 *      replaces GEMM + norm launch + GEMM of diffusers' LuminaNextDiTBlock.forward (diffusers==0.33.1) as wired by
 *      nextdit_traj.py:121-178: the projection and H stay in registers. W2 == NULL: no second GEMM (the last block); H != NULL
 *      additionally writes H (bf16) to memory. Built pairs: (K1 = 384, glu2 = 1) and (K1 = 1024, glu2 = 0); N = 384 fixed.
 *      M and mod_div must be multiples of the 128-row panel. */
typedef struct ina_dit_rowchain_args {
    const void* A;          /* bf16 [M,K1], row stride lda */
    const void* W1;         /* bf16 [384,K1], row stride ldw1 */
    const float* gamma1;    /* f32 [384] RMSNorm weight on the projection */
    const float* gate;      /* f32 [M/mod_div, mod_ld] (tanh applied) or NULL */
    float* X;               /* f32 [M,384] residual stream, updated in place, row stride ldx */
    const float* gamma2;    /* f32 [384] or NULL */
    const float* mod_scale2;/* f32 [M/mod_div, mod_ld] or NULL */
    void* H;                /* bf16 [M,384] or NULL, row stride ldh */
    const void* W2;         /* bf16 [N2,384] or NULL, row stride ldw2 */
    void* C2;               /* bf16 [M,N2] (N2/2 columns with glu2), row stride ldc2 */
    int32_t M, K1, N2;
    int32_t lda, ldw1, ldx, ldh, ldw2, ldc2;
    int32_t glu2;
    int32_t mod_div, mod_ld;
    float eps;
    int32_t _reserved;      /* 0 */
    float* seg_stats;       /* f32 [M][N2/384][mean, rstd] or NULL (plain second GEMM, N2 % 384 == 0): LayerNorm statistics (eps seg_eps) of every
                             * 384-wide segment of the C2 rows, from the fp32 accumulators - what ina_dit_attn_args.stats consumes */
    float seg_eps;
    int32_t _pad;
} ina_dit_rowchain_args;
int ina_dit_rowchain(const ina_dit_rowchain_args* args, void* stream);

/* ---- dit_attention: the attention stage of one NextDiT block in one launch:
 *          O = SDPA(LN(q1), LN(k1), v1) + tanh(head_gate[h]) * SDPA(LN(q2), K2, V2)
 *      X rows hold the fused projection [q1 | k1 | v1 | q2], each heads*64 wide; LN = LayerNorm over the whole segment
 *      ("layer_norm_across_heads"); self-attention inside each T-token sequence, cross-attention against the Lz condition rows of
 *      env = seq / seq_per_env.  reference: diffusers LuminaAttnProcessor2_0 + LuminaNextDiTBlock gate (diffusers==0.33.1,
 *      requirements/internvla_n1.txt:3) as wired by nextdit_traj.py:121-188.
 *      V2T is the transposed, key-permuted image of the condition V the kernel consumes; a call with V2T_src != NULL (re)builds it
 *      from V2T_src [env][Lz][heads*64] (strides v2_bs / v2_rs) first, and with X == NULL does only that. */
typedef struct ina_dit_attn_args {
    const void* X;          /* bf16 [nseq*T, 4*heads*64], row stride ldx */
    void* O;                /* bf16 [nseq*T, heads*64], row stride ldo */
    const float* g_q1; const float* b_q1;   /* f32 [heads*64] LayerNorm weight / bias of q1, k1, q2 */
    const float* g_k1; const float* b_k1;
    const float* g_q2; const float* b_q2;
    const void* K2;         /* bf16 condition keys: element (env, row, h, d) at env*k2_bs + row*k2_rs + h*64 + d */
    void* V2T;              /* bf16 [envs, heads, 64, 64] */
    const void* V2T_src;    /* bf16 condition values or NULL */
    const float* head_gate; /* f32 [heads] (tanh applied) or NULL */
    int64_t k2_bs, k2_rs, v2_bs, v2_rs;
    int32_t nseq, T, heads, seq_per_env, Lz, ldx, ldo;
    float scale, eps;
    int32_t stats_ld;       /* floats per row of stats (>= 8) */
    const float* stats;     /* f32 [nseq*T][4 segments][mean, rstd] of the X rows as ina_dit_rowchain_args.seg_stats writes them, or NULL:
                             * the kernel then computes the LayerNorm statistics itself (a second pass over three segments) */
} ina_dit_attn_args;
int ina_dit_attention(const ina_dit_attn_args* args, void* stream);

/* ---- diffusion-policy ConditionalUnet1D head (vendored: internnav/model/encoder/diffusion_policy/model/diffusion/conditional_unet1d.py:14-241,
 *      conv1d_components.py:7-40). Activations are channels-last, sequence-padded: row (b, t) of a level = b * (T + 2 pad) + pad + t.
 *      The Conv1d / ConvTranspose1d themselves run on ina_gemm_bf16 over overlapping row windows (implicit GEMM); these three entry
 *      points are what remains: gn_mish = GroupNorm -> Mish [-> FiLM] [-> + residual] of a Conv1dBlock / ConditionalResidualBlock1D,
 *      pad_rows = zero the pad rows after a strided convolution (+ optional bias on the valid rows), ddim_step = DDIMScheduler.step
 *      (diffusers, eta 0, epsilon prediction, clip_sample, use_clipped_model_output 0|1) on the fp32 sample + refresh of the bf16 network input. */
typedef struct ina_gn_mish_args {
    const void* X;          /* bf16 (f32 when x_f32) conv output, row (b, t) at b * in_seq_stride + t, row stride ldx */
    void* Y;                /* bf16 padded output [seqs, T + 2 pad, C], row stride ldy (pad rows are zeroed) */
    const void* R;          /* bf16 residual in the padded indexing of Y, row stride ldr, or NULL */
    const float* gamma; const float* beta;      /* f32 [C] GroupNorm affine */
    const float* film_env;  /* f32 [envs, film_ld]: this block's [scale C | bias C] at column film_off, or NULL */
    const float* film_step; /* f32 [film_ld]: the timestep part, added to film_env */
    int32_t seqs, T, C, groups, pad, in_seq_stride, ldx, ldy, ldr, seq_per_env, film_ld, film_off;
    float eps;
    int32_t x_f32;
} ina_gn_mish_args;
int ina_gn_mish(const ina_gn_mish_args* args, void* stream);

typedef struct ina_pad_rows_args {
    void* X;                /* bf16 [seqs, T + 2 pad, C] */
    const float* bias;      /* f32 [C] added to the valid rows, or NULL */
    int32_t seqs, T, pad, C, ldx, _pad;
} ina_pad_rows_args;
int ina_pad_rows(const ina_pad_rows_args* args, void* stream);

typedef struct ina_ddim_step_args {
    const float* eps;       /* f32 predicted noise in the padded row indexing, row stride lde */
    float* sample;          /* f32 [seqs * T, D], updated in place */
    void* Xin;              /* bf16 padded network input [seqs, T + 2 pad, ldx], channels [0, D) refreshed */
    int32_t seqs, T, D, pad, lde, ldx;
    float inv_sqrt_a, sqrt_b, sqrt_ap, sqrt_bp, clip;   /* 1/sqrt(abar_t), sqrt(1-abar_t), sqrt(abar_prev), sqrt(1-abar_prev), clip range (0 = off) */
    int32_t use_clipped_model_output;   /* diffusers DDIMScheduler.step(use_clipped_model_output=): 0 (the default, what
                                           diffusion_unet_lowdim_policy.py:87-91 runs: step(model_output, t, trajectory, **{})) keeps the
                                           network's eps in the direction term; 1 re-derives eps from the clipped x0 */
} ina_ddim_step_args;
int ina_ddim_step(const ina_ddim_step_args* args, void* stream);

/* ======================================================================================================================
 * SFT step (BASELINE config #5): backward and optimiser kernels. The reference has no FFI here either:
 * these replace torch autograd of InternVLAN1ForCausalLM.forward(labels=...) (internvla_n1.py:222-286) and the HF Trainer's
 * adamw_torch + clip_grad_norm_ (train_dual_system.sh:72-77). Backward GEMMs go through ina_gemm_bf16 on transposed copies.
 * Tensors with a *_dt field are bf16 (0) or f32 (1).
 * ====================================================================================================================== */
#define INA_ACT_TANH_C 6
#define INA_EW_AFFINE 0     /* Y = A * f(S[r / s_div]) + B + tab[r % tab_mod]   (f: 0 s, 1 1+s, 2 tanh s; S, B, tab optional) */
#define INA_EW_ACT_FWD 1    /* Y = act(A) */
#define INA_EW_ACT_BWD 2    /* Y = B * act'(A)            (A = pre-activation, B = dy) */
#define INA_EW_GLU_FWD 3    /* Y = silu(A) * B */
#define INA_EW_GLU_BWD 4    /* Y = D * B * silu'(A), Y2 = D * silu(A)   (D = dy) */
#define INA_EW_DROPOUT 5    /* Y = A * keep / (1 - p): keep iff ina_hash(drop_seed, r * C + c) >= drop_thresh (nn.Dropout, train mode; the same
                             * call on dy is its backward) */
typedef struct ina_ew_args {
    const void* A; const void* B; const void* D; const void* S;
    void* Y; void* Y2;
    const float* tab;       /* f32 [tab_mod, C] or NULL */
    int32_t op, rows, C;
    int32_t a_dt, b_dt, d_dt, s_dt, y_dt, y2_dt;
    int32_t lda, ldb, ldd, lds, ldy, ldy2;
    int32_t s_div, s_f, tab_mod, act, accumulate; /* accumulate: Y += */
    uint32_t drop_seed, drop_thresh;
    float drop_scale;
    const uint32_t* drop_salt; /* optional device word added to drop_seed (see ina_attn_args.drop_salt) */
} ina_ew_args;
int ina_ew(const ina_ew_args* args, void* stream);

/* out[g, c] (+)= scale * sum_{r in group g} X[r, c] * X2[r, c]  (bias / gain / LayerScale / modulation gradients, squared norms).
 * x_cs / x2_cs: column stride of X / X2 (1 = dense, 0 = broadcast one value per row); out_cs: element stride between output columns. */
typedef struct ina_colsum_args {
    const void* X; const void* X2;  /* X2 optional */
    float* out;
    float* partial;         /* f32 scratch [groups * chunks * C] with chunks = ceil(group_rows / chunk), chunk = 32 rows up to 2048 rows per group, 64 up to 4096,
                             * 128 up to 8192, 256 beyond; needed when a group has more than 32 rows */
    int64_t partial_elems;
    int32_t rows, C, group_rows;    /* group_rows 0 = one group of all rows */
    int32_t x_dt, x2_dt, ldx, ldx2, x_cs, x2_cs, ldo, out_cs;
    int32_t accumulate;
    float scale;            /* 0 means 1 */
} ina_colsum_args;
int ina_colsum(const ina_colsum_args* args, void* stream);

/* backward of y = norm(x) * gamma (+ beta), LayerNorm or RMSNorm over the last dim; DX (+)=; XHAT (bf16, optional) for dgamma */
typedef struct ina_norm_bwd_args {
    const void* X; const void* DY; const float* gamma;
    void* DX; void* XHAT;
    int32_t rows, C, x_dt, dy_dt, dx_dt, ldx, lddy, lddx, ldxh, rms, accumulate;
    float eps;
} ina_norm_bwd_args;
int ina_norm_bwd(const ina_norm_bwd_args* args, void* stream);

/* Y[c, r] = bf16(X[r, c]); Y rows are ldy >= rows long (multiple of 8 for the GEMM), the tail is zero-filled */
typedef struct ina_transpose_args {
    const void* X; void* Y;
    int32_t rows, cols, x_dt, ldx, ldy, _pad;
} ina_transpose_args;
int ina_transpose(const ina_transpose_args* args, void* stream);

/* out[t, :] (+)= sum_j coef[t, j] * in[idx[t, j], :]  (idx < 0 = unused tap): DINOv2 bicubic pos-embed interpolation
 * (dinov2.py:180-211) and, with the transposed tap table, its gradient */
typedef struct ina_sparse_rows_args {
    const float* in; float* out; const int32_t* idx; const float* coef;
    int32_t n_out, C, taps, accumulate;
} ina_sparse_rows_args;
int ina_sparse_rows(const ina_sparse_rows_args* args, void* stream);

/* Y[r, n] = sum_k X[r, k] * W[n * w_ns + k * w_ks] + bias[n] + tab[r % tab_mod, n]: nn.Linear(3, 384) / nn.Linear(384, 3)
 * (internvla_n1_arch.py:129-131) forward and input gradient */
typedef struct ina_small_linear_args {
    const void* X; const float* W; const float* bias; const float* tab; void* Y;
    int32_t rows, N, K, x_dt, y_dt, ldx, ldy, w_ns, w_ks, tab_mod;
} ina_small_linear_args;
int ina_small_linear(const ina_small_linear_args* args, void* stream);

/* masked flow-matching MSE of internvla_n1.py:283-286 and its gradient: pred [nseq * T, D] (row stride ldp), mask [nseq] */
typedef struct ina_mse_args {
    const void* pred; const float* target; const float* mask; float* loss; void* dpred;
    int32_t nseq, T, D, pred_dt, dpred_dt, ldp, lddp;
    float loss_scale;
} ina_mse_args;
int ina_mse_masked(const ina_mse_args* args, void* stream);

/* fused AdamW (torch.optim.AdamW update order) on a flat f32 buffer + clip_grad_norm_(max_norm) + 1 / world averaging + bf16 copy */
typedef struct ina_adamw_args {
    float* p; float* g; float* m; float* v;
    void* p_bf16;               /* bf16 [n] working copy or NULL */
    const float* sumsq_parts;   /* f32 [n_parts]: partial sums of g^2 (ina_colsum with X2 = X), or NULL when max_norm == 0 */
    float* norm_out;            /* f32 [1]: the total gradient norm (after grad_scale), or NULL */
    int64_t n;
    int32_t n_parts, zero_grad;
    float lr, beta1, beta2, eps, wd, bc1, bc2, max_norm, grad_scale;   /* bc = 1 - beta^step */
    int32_t _pad;
} ina_adamw_args;
int ina_adamw(const ina_adamw_args* args, void* stream);

/* partial[(split * 4 + w) * MR + m, k] = sum_{n in slice} X[m, n] * W[n, k]  (MR = 8 for M <= 8, else 16): dX of a frozen nn.Linear
 * whose weight W [N, K] stays in its stored layout; reduce the splits * 4 slots with ina_colsum. M <= 16. */
typedef struct ina_gemm_nn_args {
    const void* X; const void* W; float* partial;
    int64_t partial_elems;
    int32_t M, N, K, ldx, ldw, splits;
} ina_gemm_nn_args;
int ina_gemm_nn_bf16(const ina_gemm_nn_args* args, void* stream);

/* dW[n, k] += sum_r DY[r, n] * X[r, k] and, with db, db[n] += sum_r DY[r, n]: weight / bias gradient of nn.Linear from the row-major operands of
 * the backward tape (DY [rows, N] f32 or bf16 - rounded to bf16 for the MFMAs, the bias sum takes the unrounded values; X [rows, K] bf16), one launch
 * instead of two transposes + a GEMM + two column-sum launches (two launches with `splits` row ranges: long reductions over few tiles). N, K, lddy, ldx
 * multiples of 8, ldw of 4. */
typedef struct ina_gemm_dw_args {
    const void* DY; const void* X;
    float* dW;                  /* f32 [N, ldw], accumulated into */
    float* db;                  /* f32 [N], accumulated into, or NULL */
    float* partial;             /* splits > 1: f32 scratch [splits * N * (K + 1)] (per-range tiles, then per-range bias sums) */
    int64_t partial_elems;
    int32_t rows, N, K, dy_dt, lddy, ldx, ldw;
    int32_t splits;             /* <= 1: one launch, every tile walks all rows; 2 .. 64: the rows in that many ranges (grid.z) + a launch that adds them in order */
} ina_gemm_dw_args;
int ina_gemm_dw(const ina_gemm_dw_args* args, void* stream);

/* backward of ina_attention_bf16 (dense layouts only): f describes the forward call (Q, K, V, O and their strides);
 * dO has O's strides. dQ pass writes lse / delta [B, H, Lq] f32, the dK / dV pass (optional) reads them. dK / dV are indexed by the
 * QUERY head (sum the heads of a GQA group afterwards) and hold the key rows [kv_row0, Lk); kv_row0 = -1: the last Lq key rows of every
 * sequence (k_len - Lq .. k_len: the query rows' own keys in a cached causal pass). */
typedef struct ina_attn_bwd_args {
    ina_attn_args f;
    const void* dO; void* dQ; void* dK; void* dV;
    float* lse; float* delta;
    int64_t dq_bs, dq_rs, dq_hs, dkv_bs, dkv_rs, dkv_hs;
    int32_t kv_row0;
    int32_t nsplit;         /* > 1: split the keys of the dQ pass over nsplit workgroups per query tile (few query rows, long key axis):
                             * dQ is then accumulated with f32 atomics into dq32 (dense [B, Lq, H, D], zeroed by the caller; dQ is not written) */
    float* part;            /* f32 scratch [B, H, nsplit, 2, Lq] (nsplit > 1) */
    float* dq32;
    int32_t stage, _pad;    /* internal (set by the launcher) */
} ina_attn_bwd_args;
int ina_attention_bwd_bf16(const ina_attn_bwd_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INTERNNAV_AMD_H */
