"""Algorithmic FLOP counts of the System-1 policies, recomputed from the config ("don't hand-copy").

Convention: 2*M*N*K per GEMM, 4*Lq*Lk*d per attention head (QK^T + PV) over the UNMASKED keys; work that
is invariant over samples / sampler steps (condition K/V, RGB-D tokens, goal embedding) is counted once per environment.
These are the numerators of `roofline.achieved` in bench.py; the as-executed counts come from the library's own
per-launch tally (ina_prof_read).
"""
from __future__ import annotations


def vit_s_flops(tokens: int = 257, d: int = 384, depth: int = 12, heads: int = 6, patch_k: int = 588) -> float:
    """DINOv2 ViT-S/14 on one 224x224 frame (256 patches + cls)."""
    per_block = 2 * tokens * d * 3 * d + 2 * tokens * d * d + 2 * 2 * tokens * d * 4 * d + 4 * tokens * tokens * d
    return 2 * (tokens - 1) * patch_k * d + depth * per_block


def former_flops(n_query: int, n_tokens: int, d: int = 384, ffn: int = 2048, layers: int = 2, out_dim: int = 384) -> float:
    per = (2 * n_query * d * 3 * d + 4 * n_query * n_query * d + 2 * n_query * d * d      # self-attention
           + 2 * n_query * d * d + 2 * n_tokens * d * 2 * d + 4 * n_query * n_tokens * d + 2 * n_query * d * d  # cross
           + 2 * 2 * n_query * d * ffn)
    return layers * per + 2 * n_query * d * out_dim


def denoiser_pass_flops(S: int, T: int, Lc_visible: int, causal: bool, d: int = 384, depth: int = 16) -> float:
    """one pass of the 16-layer pre-LN decoder over the S sample sequences of ONE env (T tokens each)."""
    rows = S * T
    self_keys = (T + 1) / 2.0 if causal else float(T)
    per = (2 * rows * d * 3 * d + 4 * rows * self_keys * d + 2 * rows * d * d
           + 2 * rows * d * d + 4 * rows * Lc_visible * d + 2 * rows * d * d
           + 2 * 2 * rows * d * 4 * d)
    return depth * per + 2 * 2 * rows * 3 * d


def navdpnet_flops_per_env(cfg) -> dict:
    """NavDPNet.predict_pointgoal_batch_action_vel (BASELINE config #2) - algorithmic FLOPs per environment step."""
    M, T, S, D = cfg["memory_size"], cfg["predict_size"], cfg["sample_num"], cfg["token_dim"]
    K, depth = cfg["num_train_timesteps"], cfg["temporal_depth"]
    Lc = M * 16 + 4
    vit = (M + 1) * vit_s_flops()
    former = former_flops(M * 16, (M + 1) * 256, out_dim=D)
    cond_kv = depth * 2 * Lc * D * 2 * D
    denoise = K * denoiser_pass_flops(S, T, Lc, causal=True, d=D, depth=depth)
    critic = denoiser_pass_flops(S, T, Lc - 4, causal=False, d=D, depth=depth)
    total = vit + former + cond_kv + denoise + critic
    return dict(vit=vit, former=former, cond_kv=cond_kv, denoise=denoise, critic=critic, total=total)


def n1_navdp_flops_per_env(cfg, n_query: int = 4) -> dict:
    """NavDP_Policy_DPT_CriticSum_DAT.predict_pointgoal_action_async (InternVLA-N1 navdp_async S1 call)."""
    M, T, S, D, V = cfg["memory_size"], cfg["predict_size"], cfg["sample_num"], cfg["token_dim"], cfg["vlm_token_dim"]
    K, depth = cfg["num_train_timesteps"], cfg["temporal_depth"]
    Lc = M * 16 + 2
    vit = 2 * M * vit_s_flops()
    former = former_flops(M * 16, 2 * M * 256, out_dim=D)
    goal = 2 * n_query * (V * V // 4 + (V // 4) * (V // 8) + (V // 8) * D) + 2 * n_query * D * 2 * D + 2 * D * D
    cond_kv = depth * 2 * Lc * D * 2 * D
    denoise = K * denoiser_pass_flops(S, T, Lc, causal=True, d=D, depth=depth)
    total = vit + former + goal + cond_kv + denoise
    return dict(vit=vit, former=former, goal=goal, cond_kv=cond_kv, denoise=denoise, total=total)


# ---------------------------------------------------------------------------------------------------- InternVLA-N1 dual system
def qwen_vit_flops(grids, cfg) -> float:
    """Qwen2.5-VL vision tower on a list of (t, h, w) patch grids: patch embed, window / per-image attention, SwiGLU, merger."""
    D, I, O, hd = cfg["v_hidden"], cfg["v_inter"], cfg["v_out"], cfg["v_hidden"] // cfg["v_heads"]
    from .qwen_vl import vision_window_permutation

    total = 0.0
    for g in grids:
        n = g[0] * g[1] * g[2]
        _, cu = vision_window_permutation([g], 2, cfg["v_window"], cfg["v_patch"])
        win_keys = float(((cu[1:] - cu[:-1]).astype("float64") ** 2).sum())   # sum over windows of len^2
        n_full = len(cfg["v_fullatt"])
        lin = 2 * n * D * 3 * D + 2 * n * D * D + 3 * 2 * n * D * I
        attn = 4 * D * (n_full * float(n) * n + (cfg["v_depth"] - n_full) * win_keys)
        total += 2 * n * 1176 * D + cfg["v_depth"] * lin + attn + 2 * (n // 4) * (4 * D) * (4 * D) + 2 * (n // 4) * (4 * D) * O
    return total


def llm_flops(new_tokens: int, ctx_start: int, cfg, lm_head_rows: int = 0) -> float:
    """decoder stack over `new_tokens` tokens per sequence appended at context length ctx_start (causal), + lm_head rows."""
    H, TI, nh, nkv = cfg["t_hidden"], cfg["t_inter"], cfg["t_heads"], cfg["t_kv_heads"]
    hd = H // nh
    lin = 2 * new_tokens * (H * (nh + 2 * nkv) * hd + H * H + 3 * H * TI)
    keys = sum(ctx_start + i + 1 for i in range(new_tokens))
    attn = 4 * keys * H
    return cfg["t_layers"] * (lin + attn) + 2 * lm_head_rows * H * cfg["vocab"]


def s2_call_flops(S: int, grids, n_decode: int, cfg, prefix_len: int = 0) -> dict:
    """one System-2 call per env (pixel-goal answer): ViT + prefill + n_decode greedy tokens + N_QUERY latent queries on the cache
    (16.46 TFLOP for 4 frames, S = 920, n_dec = 8). `grids` = the images the call ENCODES; prefix_len = prompt tokens
    whose K/V come from the prefix cache (they cost no GEMM FLOPs; the other tokens still attend to them)."""
    vit = qwen_vit_flops(grids, cfg)
    prefill = llm_flops(S - prefix_len, prefix_len, cfg, lm_head_rows=1)
    decode = sum(llm_flops(1, S + j, cfg, lm_head_rows=1) for j in range(max(n_decode - 1, 0)))
    lat = llm_flops(1 + cfg["n_query"], S + max(n_decode - 1, 0), cfg)
    return dict(vit=vit, prefill=prefill, decode=decode, latents=lat, total=vit + prefill + decode + lat)


def nextdit_s1_flops_per_env(cfg) -> dict:
    """generate_traj, nextdit_async (DualVLN): 2 ViT-S frames, MemoryEncoder, QFormer, cond K/V once, 10 x 12-layer DiT on 32 x 32 tokens."""
    D, L, S, T, nl, ffn = cfg["dit_dim"], cfg["latent_dim"], cfg["sample_num"], cfg["predict_size"], cfg["dit_layers"], cfg["dit_ffn"]
    nm, Lz, nq = cfg["memory_frames"] * 256, 32 + cfg["n_query"], cfg["n_query"]
    vit = cfg["memory_frames"] * vit_s_flops()
    mem = 3 * (2 * nm * D * 3 * D + 4 * nm * nm * D + 2 * nm * D * D + 2 * 2 * nm * D * 2048)
    qf = 3 * (2 * 32 * L * 3 * L + 4 * 32 * 32 * L + 2 * 32 * L * L + 2 * 32 * L * L + 2 * nm * L * 2 * L + 4 * 32 * nm * L + 2 * 32 * L * L + 2 * 2 * 32 * L * 2048)
    cond = 2 * nq * (cfg["vlm_token_dim"] * L + L * L) + 2 * Lz * (L * D + D * D) + nl * 2 * Lz * D * 2 * D
    rows = S * T
    per_layer = 2 * rows * D * 4 * D + 4 * rows * T * D + 4 * rows * Lz * D + 2 * rows * D * D + 3 * 2 * rows * D * ffn
    dit = cfg["num_inference_steps"] * (nl * per_layer + 2 * 2 * rows * 3 * D)
    total = vit + mem + qf + cond + dit
    return dict(vit=vit, memory_encoder=mem, qformer=qf, cond=cond, dit=dit, total=total)


def unet1d_flops_per_env(cfg) -> dict:
    """ConditionalUnet1D + DDIM head per env: num_inference_steps x one forward over sample_num sequences (2 M N K per convolution as an
    implicit GEMM, K = taps x C_in; FiLM projections: condition half once per call)."""
    T, S, k = cfg["predict_size"], cfg["sample_num"], cfg["kernel_size"]
    d = list(cfg["down_dims"])
    cond = cfg["dsed"] + cfg["global_cond_dim"]

    def conv(rows, co, ci, taps):
        return 2.0 * rows * co * ci * taps

    def res(rows, ci, co):
        return conv(rows, co, ci, k) + conv(rows, co, co, k) + (conv(rows, co, ci, 1) if ci != co else 0.0)

    Ts = [T, T // 2, T // 4]
    f = res(Ts[0], cfg["input_dim"], d[0]) + res(Ts[0], d[0], d[0]) + conv(Ts[1], d[0], d[0], 3)
    f += res(Ts[1], d[0], d[1]) + res(Ts[1], d[1], d[1]) + conv(Ts[2], d[1], d[1], 3)
    f += res(Ts[2], d[1], d[2]) + res(Ts[2], d[2], d[2]) + 2 * res(Ts[2], d[2], d[2])
    f += res(Ts[2], 2 * d[2], d[1]) + res(Ts[2], d[1], d[1]) + conv(Ts[1], d[1], d[1], 4)        # ConvTranspose1d(4, 2, 1): 4 taps over T/4 inputs
    f += res(Ts[1], 2 * d[1], d[0]) + res(Ts[1], d[0], d[0]) + conv(Ts[0], d[0], d[0], 4)
    f += conv(Ts[0], d[0], d[0], k) + conv(Ts[0], cfg["input_dim"], d[0], 1)
    film = 2.0 * cfg["global_cond_dim"] * 2 * (2 * d[0] + 2 * d[1] + 2 * d[2] + 2 * d[2] + 2 * d[1] + 2 * d[0])
    total = cfg["num_inference_steps"] * S * f + film
    return dict(forward_per_sequence=f, film=film, total=total)
