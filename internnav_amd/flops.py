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
