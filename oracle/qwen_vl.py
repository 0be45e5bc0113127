"""Oracle restatement of the InternVLA-N1 System-2 VLM (Qwen2.5-VL vision tower + LLM + the reference's glue).  TEST INFRA ONLY.

The arithmetic lives in a third-party dependency, transformers==4.51.0 (requirements/internvla_n1.txt:7), which the
reference subclasses (internvla_n1.py:31-56). This image has transformers 5.x - same math, different packaging - whose
modules pin this restatement in `oracle/make_golden.py` (Qwen2_5_VisionTransformerPretrainedModel, Qwen2_5_VLTextModel).
In-tree reference code followed here:
  internnav/model/basemodel/internvla_n1/internvla_n1.py:128-172   embed lookup, image-embed scatter, latent_queries rows
  internnav/model/basemodel/internvla_n1/internvla_n1.py:185,206-220  m-rope index, decoder stack, lm_head on every position
  internnav/model/basemodel/internvla_n1/internvla_n1.py:320-347   generate_latents (full re-run with N_QUERY traj tokens)
  internnav/dataset/rope2d.py:6-180                                get_rope_index_25 (vendored 3-D rope index)
  internnav/model/basemodel/internvla_n1/internvla_n1_policy.py:169-176  greedy generate(max_new_tokens, do_sample=False)
transformers pieces restated (file = models/qwen2_5_vl/modeling_qwen2_5_vl.py of the pinned package):
  Qwen2_5_VisionPatchEmbed, Qwen2_5_VLVisionBlock (RMSNorm, qkv+bias, 2-D rope, window / full attention, SwiGLU+bias),
  Qwen2_5_VLPatchMerger, window index / cu_seqlens, Qwen2_5_VLDecoderLayer (q/k/v bias, m-rope [16,24,24], GQA, SwiGLU).

State-dict keys are those of a reference checkpoint (4.51 layout): visual.*, model.embed_tokens, model.layers.*, model.norm,
model.latent_queries, lm_head.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .nn_ref import rms_norm, sdpa

IMAGE_TOKEN_INDEX = 151655   # internvla_n1.py:19
TRAJ_TOKEN_INDEX = 151667    # internvla_n1.py:18
VISION_START = 151652


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


# ----------------------------------------------------------------------------------------------------- vision tower
def vision_position_ids(grid_thw, merge: int = 2):
    """(h, w) index of every patch, block-major over merge x merge cells (the order of pixel_values rows)."""
    out = []
    for t, h, w in grid_thw:
        hp = torch.arange(h).unsqueeze(1).expand(h, w)
        wp = torch.arange(w).unsqueeze(0).expand(h, w)
        shape = (h // merge, merge, w // merge, merge)
        hp = hp.reshape(shape).transpose(1, 2).flatten()
        wp = wp.reshape(shape).transpose(1, 2).flatten()
        out.append(torch.stack([hp, wp], dim=-1).repeat(t, 1))
    return torch.cat(out, dim=0)


def vision_window_index(grid_thw, merge: int = 2, window_size: int = 112, patch: int = 14):
    """window permutation over merged cells + cumulative window lengths in PATCH units (transformers get_window_index)."""
    win = window_size // merge // patch
    index, cu, base = [], [0], 0
    for t, h, w in grid_thw:
        gh, gw = h // merge, w // merge
        idx = torch.arange(t * gh * gw).reshape(t, gh, gw)
        ph, pw = win - gh % win, win - gw % win
        nh, nw = (gh + ph) // win, (gw + pw) // win
        padded = F.pad(idx, (0, pw, 0, ph), "constant", -100).reshape(t, nh, win, nw, win).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, win, win)
        lens = (padded != -100).sum([2, 3]).reshape(-1)
        flat = padded.reshape(-1)
        index.append(flat[flat != -100] + base)
        cu.extend((lens.cumsum(0) * merge * merge + cu[-1]).tolist())
        base += t * gh * gw
    cu = torch.unique_consecutive(torch.tensor(cu, dtype=torch.int32))
    return torch.cat(index), cu


def vision_tower(pixel_values, grid_thw, sd, cfg, p="visual.", tap=None):
    """Qwen2_5_VisionTransformerPretrainedModel.forward -> merged image embeds [sum(h*w)/4, out_hidden] in token order.
    tap(i, x): called with the residual stream [N, D] (window order) after block i (drift reports of the full-depth fixture)."""
    D, H, merge = cfg["v_hidden"], cfg["v_heads"], 2
    hd = D // H
    grid = [tuple(int(v) for v in g) for g in grid_thw]
    x = F.linear(pixel_values.float(), sd[p + "patch_embed.proj.weight"].reshape(D, -1))
    N = x.shape[0]
    win_idx, cu_win = vision_window_index(grid, merge, cfg["v_window"], cfg["v_patch"])
    x = x.reshape(N // 4, 4, D)[win_idx].reshape(N, D)
    pos = vision_position_ids(grid, merge)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float32) / (hd // 2)))
    rot = (pos.unsqueeze(-1).float() * inv).flatten(1)                     # [N, hd/2]
    rot = rot.reshape(N // 4, 4, -1)[win_idx].reshape(N, -1)
    emb = torch.cat((rot, rot), dim=-1)
    cos, sin = emb.cos().unsqueeze(1), emb.sin().unsqueeze(1)              # [N, 1, hd]
    cu_full = torch.tensor([0] + [t * h * w for t, h, w in grid], dtype=torch.int32).cumsum(0)
    for i in range(cfg["v_depth"]):
        b = f"{p}blocks.{i}."
        cu = cu_full if i in cfg["v_fullatt"] else cu_win
        y = rms_norm(x, sd[b + "norm1.weight"], 1e-6)
        qkv = F.linear(y, sd[b + "attn.qkv.weight"], sd[b + "attn.qkv.bias"]).reshape(N, 3, H, hd)
        q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        o = torch.empty(N, H, hd)
        for s, e in zip(cu[:-1].tolist(), cu[1:].tolist()):
            o[s:e] = sdpa(q[s:e].transpose(0, 1), k[s:e].transpose(0, 1), v[s:e].transpose(0, 1)).transpose(0, 1)
        x = x + F.linear(o.reshape(N, D), sd[b + "attn.proj.weight"], sd[b + "attn.proj.bias"])
        y = rms_norm(x, sd[b + "norm2.weight"], 1e-6)
        g = F.silu(F.linear(y, sd[b + "mlp.gate_proj.weight"], sd[b + "mlp.gate_proj.bias"]))
        u = F.linear(y, sd[b + "mlp.up_proj.weight"], sd[b + "mlp.up_proj.bias"])
        x = x + F.linear(g * u, sd[b + "mlp.down_proj.weight"], sd[b + "mlp.down_proj.bias"])
        if tap is not None:
            tap(i, x)
    m = p + "merger."
    y = rms_norm(x, sd[m + "ln_q.weight"], 1e-6).reshape(N // 4, 4 * D)
    y = F.linear(F.gelu(F.linear(y, sd[m + "mlp.0.weight"], sd[m + "mlp.0.bias"])), sd[m + "mlp.2.weight"], sd[m + "mlp.2.bias"])
    return y[torch.argsort(win_idx)]


# ----------------------------------------------------------------------------------------------------- rope index
def rope_index(input_ids, grid_thw, image_token_id=IMAGE_TOKEN_INDEX, vision_start_id=VISION_START, merge: int = 2):
    """get_rope_index_25 for still images without padding (rope2d.py:69-158): position ids [3, B, S] and rope deltas [B]."""
    B, S = input_ids.shape
    out = torch.zeros(3, B, S, dtype=torch.long)
    deltas = []
    img = 0
    for b in range(B):
        toks = input_ids[b].tolist()
        n_img = sum(1 for i, t in enumerate(toks[:-1]) if t == vision_start_id and toks[i + 1] == image_token_id)
        chunks, st = [], 0
        for _ in range(n_img):
            ed = toks.index(image_token_id, st)
            t, h, w = (int(v) for v in grid_thw[img])
            img += 1
            gh, gw = h // merge, w // merge
            base = (chunks[-1].max().item() + 1) if chunks else 0
            text_len = ed - st
            chunks.append(torch.arange(text_len).view(1, -1).expand(3, -1) + base)
            ti = torch.zeros(t * gh * gw, dtype=torch.long)  # second_per_grid_t = 0 for images (rope2d.py:112)
            hi = torch.arange(gh).view(1, -1, 1).expand(t, -1, gw).flatten()
            wi = torch.arange(gw).view(1, 1, -1).expand(t, gh, -1).flatten()
            chunks.append(torch.stack([ti, hi, wi]) + text_len + base)
            st = ed + t * gh * gw
        if st < len(toks):
            base = (chunks[-1].max().item() + 1) if chunks else 0
            chunks.append(torch.arange(len(toks) - st).view(1, -1).expand(3, -1) + base)
        pos = torch.cat(chunks, dim=1)
        out[:, b] = pos
        deltas.append(int(pos.max().item()) + 1 - S)
    return out, torch.tensor(deltas)


def mrope_cos_sin(position_ids, hd: int = 128, theta: float = 1e6, section=(16, 24, 24)):
    """Qwen2_5_VLRotaryEmbedding + the mrope_section interleave: cos/sin [B, S, hd] (fp32)."""
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    freqs = position_ids[..., None].float() * inv                           # [3, B, S, hd/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos(), emb.sin()
    sec = list(section) * 2
    cos = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1)
    sin = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1)
    return cos, sin


# ----------------------------------------------------------------------------------------------------- text model
def decoder_stack(x, position_ids, sd, cfg, p="model.", tap=None, cache=None):
    """Qwen2_5_VLTextModel on inputs_embeds x [B, S, H] with a causal mask -> final-norm hidden states [B, S, H].
    tap(i, x): called with the residual stream after layer i. cache: a list (one entry per layer, initially None) of (k, v) of the
    tokens already processed - the S new tokens attend to them and are appended (HF `use_cache=True`; same result as re-running
    the whole sequence in exact arithmetic, which tests/test_oracle_golden.py checks on the reduced configuration)."""
    B, S, Hd = x.shape
    nh, nkv = cfg["t_heads"], cfg["t_kv_heads"]
    hd = Hd // nh
    cos, sin = mrope_cos_sin(position_ids, hd, cfg["rope_theta"])
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    past = cache[0][0].shape[2] if cache is not None and cache[0] is not None else 0
    mask = torch.triu(torch.full((S, past + S), float("-inf")), diagonal=past + 1)
    for i in range(cfg["t_layers"]):
        b = f"{p}layers.{i}."
        y = rms_norm(x, sd[b + "input_layernorm.weight"], 1e-6)
        q = F.linear(y, sd[b + "self_attn.q_proj.weight"], sd[b + "self_attn.q_proj.bias"]).view(B, S, nh, hd).transpose(1, 2)
        k = F.linear(y, sd[b + "self_attn.k_proj.weight"], sd[b + "self_attn.k_proj.bias"]).view(B, S, nkv, hd).transpose(1, 2)
        v = F.linear(y, sd[b + "self_attn.v_proj.weight"], sd[b + "self_attn.v_proj.bias"]).view(B, S, nkv, hd).transpose(1, 2)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        if cache is not None:
            if cache[i] is not None:
                k, v = torch.cat([cache[i][0], k], dim=2), torch.cat([cache[i][1], v], dim=2)
            cache[i] = (k, v)
        k = k.repeat_interleave(nh // nkv, dim=1)
        v = v.repeat_interleave(nh // nkv, dim=1)
        o = sdpa(q, k, v, mask).transpose(1, 2).reshape(B, S, Hd)
        x = x + F.linear(o, sd[b + "self_attn.o_proj.weight"])
        y = rms_norm(x, sd[b + "post_attention_layernorm.weight"], 1e-6)
        x = x + F.linear(F.silu(F.linear(y, sd[b + "mlp.gate_proj.weight"])) * F.linear(y, sd[b + "mlp.up_proj.weight"]),
                         sd[b + "mlp.down_proj.weight"])
        if tap is not None:
            tap(i, x)
    return rms_norm(x, sd[p + "norm.weight"], 1e-6)


def input_embeds(input_ids, image_embeds, sd, cfg):
    """internvla_n1.py:128-172: token embeddings, image rows <- vision tower output, TRAJ rows <- latent_queries."""
    x = sd["model.embed_tokens.weight"][input_ids].clone()
    if image_embeds is not None:
        x[input_ids == cfg["image_token_id"]] = image_embeds.to(x.dtype)
    traj = input_ids == cfg["traj_token_id"]
    if traj.any():
        x[traj] = sd["model.latent_queries"].repeat(input_ids.shape[0], 1, 1).view(-1, x.shape[-1])
    return x


def forward_logits(sd, cfg, input_ids, pixel_values, grid_thw):
    """InternVLAN1ForCausalLM.forward, inference branch: logits [B, S, V] (lm_head on every position, internvla_n1.py:220)."""
    img = vision_tower(pixel_values, grid_thw, sd, cfg) if pixel_values is not None else None
    x = input_embeds(input_ids, img, sd, cfg)
    pos, _ = rope_index(input_ids, grid_thw, cfg["image_token_id"], cfg["vision_start_id"])
    h = decoder_stack(x, pos, sd, cfg)
    return F.linear(h, sd["lm_head.weight"]), h


def generate(sd, cfg, input_ids, pixel_values, grid_thw, max_new_tokens: int, eos_token_id=None):
    """greedy HF generate (internvla_n1_policy.py:169-176) by full recomputation (exactly the cached result in exact arithmetic).
    Returns the generated sequences [B, S + n] (all rows advance together; a row that has emitted EOS keeps emitting EOS)."""
    ids = input_ids.clone()
    done = torch.zeros(ids.shape[0], dtype=torch.bool)
    for _ in range(max_new_tokens):
        logits, _ = forward_logits(sd, cfg, ids, pixel_values, grid_thw)
        nxt = logits[:, -1].argmax(-1)
        if eos_token_id is not None:
            nxt = torch.where(done, torch.full_like(nxt, eos_token_id), nxt)
            done |= nxt == eos_token_id
        ids = torch.cat([ids, nxt[:, None]], dim=1)
        if eos_token_id is not None and bool(done.all()):
            break
    return ids


def generate_latents(sd, cfg, output_ids, pixel_values, grid_thw):
    """generate_latents (internvla_n1.py:320-347): append N_QUERY TRAJ tokens, full forward, last-layer hidden of those rows."""
    nq = cfg["n_query"]
    ids = torch.cat([output_ids, torch.full((output_ids.shape[0], nq), cfg["traj_token_id"], dtype=output_ids.dtype)], dim=1)
    _, h = forward_logits(sd, cfg, ids, pixel_values, grid_thw)
    return h[:, -nq:, :]
