"""InternVLA-N1 System-2 (Qwen2.5-VL vision tower + 7B LLM + latent trajectory queries) on the gfx950 op library.

Drop-in for the model surface the reference's callers use:
    model.generate(input_ids, pixel_values, image_grid_thw, max_new_tokens, do_sample=False).sequences
    model.generate_latents(output_ids, pixel_values, image_grid_thw) -> [B, N_QUERY, 3584]
(/root/reference/internnav/model/basemodel/internvla_n1/internvla_n1.py:58-220, 320-347; call sites
internvla_n1_policy.py:169-191, habitat_vln_evaluator.py:418-448), batched over environments (the reference is batch 1).

Arithmetic = transformers Qwen2.5-VL (un-vendored dependency of the reference; restated and pinned in oracle/qwen_vl.py):
  vision: patch-embed GEMM (K = 1176), window permutation, 32 x {RMSNorm, qkv+bias GEMM, 2-D rope, varlen window / per-image
          attention (16 heads x 80), proj GEMM + residual, RMSNorm, SwiGLU GEMM pair + bias}, merger (RMSNorm, 5120->5120 GELU ->3584)
  text:   embedding gather, image-embed scatter, latent_queries rows, m-rope tables from 3-D position ids, 28 x {RMSNorm,
          fused q|k|v GEMM + bias, m-rope, KV-cache append, causal GQA attention (28q / 4kv x 128), o_proj + residual, RMSNorm,
          SwiGLU GEMM pair + residual}, final RMSNorm, lm_head on the LAST position only, greedy argmax on the device.
Work the reference wastes and this engine removes without changing results (SURVEY.md 7 "hard parts"):
  * lm_head on every prompt position (internvla_n1.py:220) -> last position only (the only row generate() reads);
  * generate_latents re-runs ViT + the whole prompt (internvla_n1.py:330-344) -> the N_QUERY latent tokens run against the KV
    cache left by generate(): causal attention makes the prefix K/V identical;
  * per-token host syncs of HF generate -> a fixed number of decode steps on the device, EOS handled after the fact.

HBM layout (B sequences of S tokens, cache capacity S_max, H = 3584):
  x f32 [B*S, H] residual stream | h, att bf16 [B*S, H] | qkv bf16 [B*S, 4608] = q(28x128) | k(4x128) | v(4x128)
  ff bf16 [B*S, 18944] | kv[l] bf16 [B*S_max, 1024] = k | v per cached token | cos/sin f32 [B*S, 128] m-rope tables
  vision: xv f32 [Np, 1280] | hv/attv bf16 [Np, 1280] | qkvv bf16 [Np, 3840] | ffv bf16 [Np, 3456] (3420 zero-padded) | emb bf16 [Np/4, H]
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .runtime import CapacityError


# ---------------------------------------------------------------------------------------------------- host index logic (integers only)
def vision_window_permutation(grids: Sequence[Tuple[int, int, int]], merge: int = 2, window: int = 112, patch: int = 14):
    """Window order of the merged 2x2 cells and the window boundaries in patch units (transformers get_window_index)."""
    win = window // merge // patch
    idx_all, cu, base = [], [0], 0
    for t, h, w in grids:
        gh, gw = h // merge, w // merge
        idx = np.arange(t * gh * gw).reshape(t, gh, gw)
        ph, pw = win - gh % win, win - gw % win
        nh, nw = (gh + ph) // win, (gw + pw) // win
        pad = np.pad(idx, ((0, 0), (0, ph), (0, pw)), constant_values=-100)
        pad = pad.reshape(t, nh, win, nw, win).transpose(0, 1, 3, 2, 4).reshape(t, nh * nw, win * win)
        lens = (pad != -100).sum(-1).reshape(-1)
        flat = pad.reshape(-1)
        idx_all.append(flat[flat != -100] + base)
        for n in lens.tolist():
            if n:
                cu.append(cu[-1] + n * merge * merge)
        base += t * gh * gw
    return np.concatenate(idx_all), np.asarray(cu, dtype=np.int32)


def vision_rope_tables(grids, hd: int = 80, merge: int = 2):
    """cos/sin [Np, hd] of the 2-D vision rope in pixel_values (block-major) order (transformers rot_pos_emb)."""
    out = []
    for t, h, w in grids:
        hp = np.broadcast_to(np.arange(h)[:, None], (h, w)).reshape(h // merge, merge, w // merge, merge).transpose(0, 2, 1, 3).reshape(-1)
        wp = np.broadcast_to(np.arange(w)[None, :], (h, w)).reshape(h // merge, merge, w // merge, merge).transpose(0, 2, 1, 3).reshape(-1)
        out.append(np.tile(np.stack([hp, wp], -1), (t, 1)))
    pos = torch.from_numpy(np.concatenate(out)).float()
    dim = hd // 2
    inv = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    rot = (pos.unsqueeze(-1) * inv).flatten(1)
    emb = torch.cat((rot, rot), dim=-1)
    return emb.cos(), emb.sin()


def rope_index(input_ids: np.ndarray, grids, image_token_id: int, vision_start_id: int, merge: int = 2):
    """3-D (t,h,w) position ids of get_rope_index_25 (reference internnav/dataset/rope2d.py:69-158) for still images, no padding."""
    B, S = input_ids.shape
    pos = np.zeros((3, B, S), dtype=np.int64)
    deltas = np.zeros(B, dtype=np.int64)
    img = 0
    for b in range(B):
        toks = input_ids[b]
        starts = np.nonzero((toks[:-1] == vision_start_id) & (toks[1:] == image_token_id))[0]
        chunks, st, nxt = [], 0, 0
        for s0 in starts:
            ed = int(s0) + 1
            t, h, w = grids[img]
            img += 1
            gh, gw = h // merge, w // merge
            text_len = ed - st
            chunks.append(np.broadcast_to(np.arange(text_len)[None], (3, text_len)) + nxt)
            ti = np.zeros(t * gh * gw, dtype=np.int64)
            hi = np.broadcast_to(np.arange(gh)[None, :, None], (t, gh, gw)).reshape(-1)
            wi = np.broadcast_to(np.arange(gw)[None, None, :], (t, gh, gw)).reshape(-1)
            vis = np.stack([ti, hi, wi]) + text_len + nxt
            chunks.append(vis)
            nxt = int(vis.max()) + 1
            st = ed + t * gh * gw
        if st < S:
            chunks.append(np.broadcast_to(np.arange(S - st)[None], (3, S - st)) + nxt)
        p = np.concatenate(chunks, axis=1)
        pos[:, b] = p
        deltas[b] = int(p.max()) + 1 - S
    return pos, deltas


def _interleave16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    n, k = a.shape
    return torch.stack([a.view(n // 16, 16, k), b.view(n // 16, 16, k)], dim=1).reshape(2 * n, k).contiguous()


def _pad_rows(t: torch.Tensor, n: int) -> torch.Tensor:
    return t if t.shape[0] == n else torch.cat([t, t.new_zeros((n - t.shape[0],) + tuple(t.shape[1:]))], 0)


class QwenVLEngine:
    """weights: mapping key -> tensor (a state dict, or a lazy provider that materialises tensors on the device)."""

    def __init__(self, weights, cfg: dict, device="cuda:0", max_seqs: int = 16, max_seq_len: int = 1024, max_patches: int = 16 * 3136,
                 frag_weights: Optional[bool] = None):
        dev = torch.device(device)
        bf, f32 = torch.bfloat16, torch.float32
        self.cfg, self.device = cfg, dev
        self.B_max, self.S_max, self.Np_max = max_seqs, max_seq_len, max_patches
        self.fuse_decode_norm = True   # decode passes: RMSNorm fused into the q|k|v and gate|up weight-streaming GEMMs
        # prefill of >= 2 plain prompts (no cached prefix / embeddings) as two half batches on two streams (fork / join inside the launch
        # sequence, graph capturable): every op of the tower and of the decoder stack works on row ranges, so the halves use disjoint
        # slices of the same buffers and cache slots and the results are the same; the tails of one half's GEMM launches (tile
        # quantisation over 256 CUs) are filled by the other half's workgroups (119.6 -> 115.5 ms at 7 prompts, profiles/r03q_*)
        self.split_prefill = True
        self.split_serial = False     # True: the two half micro-batches of split_prefill on ONE stream (per-launch event timing, bench.py)
        self._side = None
        self.fuse_decode_rope = True   # single-token passes: rotary embedding + KV-cache append inside the attention launch instead of a launch of their own
        self.tap = None   # debug / parity hook: tap(kind, index, residual_stream) after every ViT block ("vit") and decoder layer ("llm"); eager runs only
        D, I = cfg["v_hidden"], cfg["v_inter"]
        Ip = (I + 63) // 64 * 64   # SwiGLU width padded with zero rows/cols: GLU tiles need N % 32 == 0, the LDS-DMA GEMM K % 64 == 0
        self.vD, self.vI, self.vH, self.vhd = D, Ip, cfg["v_heads"], D // cfg["v_heads"]
        H, TI = cfg["t_hidden"], cfg["t_inter"]
        self.H, self.TI, self.nh, self.nkv = H, TI, cfg["t_heads"], cfg["t_kv_heads"]
        self.hd = H // self.nh
        self.qkv_w = (self.nh + 2 * self.nkv) * self.hd
        self.kv_w = 2 * self.nkv * self.hd
        W = weights

        def w(k):
            return W[k].to(device=dev, dtype=bf).contiguous()

        def f(k):
            return W[k].to(device=dev, dtype=f32).contiguous()

        # ---- vision tower
        self.v_patch = W["visual.patch_embed.proj.weight"].to(device=dev, dtype=bf).reshape(D, -1).contiguous()
        self.v_blocks = []
        for i in range(cfg["v_depth"]):
            b = f"visual.blocks.{i}."
            gw_, uw = _pad_rows(W[b + "mlp.gate_proj.weight"].to(dev).float(), Ip), _pad_rows(W[b + "mlp.up_proj.weight"].to(dev).float(), Ip)
            gb, ub = _pad_rows(W[b + "mlp.gate_proj.bias"].to(dev).float(), Ip), _pad_rows(W[b + "mlp.up_proj.bias"].to(dev).float(), Ip)
            dw = W[b + "mlp.down_proj.weight"].to(dev).float()
            dw = torch.cat([dw, dw.new_zeros(D, Ip - I)], 1) if Ip != I else dw
            self.v_blocks.append(dict(
                n1=f(b + "norm1.weight"), n2=f(b + "norm2.weight"), qkv_w=w(b + "attn.qkv.weight"), qkv_b=f(b + "attn.qkv.bias"),
                proj_w=w(b + "attn.proj.weight"), proj_b=f(b + "attn.proj.bias"),
                gu_w=_interleave16(gw_, uw).to(bf), gu_b=_interleave16(gb[:, None], ub[:, None]).reshape(-1).contiguous(),
                down_w=dw.to(bf).contiguous(), down_b=f(b + "mlp.down_proj.bias"), full=i in cfg["v_fullatt"]))
        self.m_ln = f("visual.merger.ln_q.weight")
        self.m0 = (w("visual.merger.mlp.0.weight"), f("visual.merger.mlp.0.bias"))
        self.m2 = (w("visual.merger.mlp.2.weight"), f("visual.merger.mlp.2.bias"))
        Np = max_patches
        self.pv_perm = torch.empty(Np, 1176, dtype=bf, device=dev)
        self.xv = torch.empty(Np, D, dtype=f32, device=dev)
        self.hv = torch.empty(Np, D, dtype=bf, device=dev)
        self.attv = torch.empty(Np, D, dtype=bf, device=dev)
        self.qkvv = torch.empty(Np, 3 * D, dtype=bf, device=dev)
        self.ffv = torch.empty(Np, Ip, dtype=bf, device=dev)
        self.mh = torch.empty(Np // 4, 4 * D, dtype=bf, device=dev)
        self.emb = torch.empty(Np // 4, cfg["v_out"], dtype=bf, device=dev)
        self.emb_tok = torch.empty(Np // 4, cfg["v_out"], dtype=bf, device=dev)   # the same rows in image-token order (what a per-frame cache keeps)
        self.v_cos = torch.empty(Np, self.vhd, dtype=f32, device=dev)
        self.v_sin = torch.empty(Np, self.vhd, dtype=f32, device=dev)
        # ---- text model
        self.embed = w("model.embed_tokens.weight")
        self.layers = []
        for i in range(cfg["t_layers"]):
            b = f"model.layers.{i}."
            qkvw = torch.cat([W[b + "self_attn.q_proj.weight"].to(dev), W[b + "self_attn.k_proj.weight"].to(dev), W[b + "self_attn.v_proj.weight"].to(dev)], 0)
            qkvb = torch.cat([W[b + "self_attn.q_proj.bias"].to(dev), W[b + "self_attn.k_proj.bias"].to(dev), W[b + "self_attn.v_proj.bias"].to(dev)], 0)
            self.layers.append(dict(
                n1=f(b + "input_layernorm.weight"), n2=f(b + "post_attention_layernorm.weight"),
                qkv_w=qkvw.to(bf).contiguous(), qkv_b=qkvb.to(f32).contiguous(), o_w=w(b + "self_attn.o_proj.weight"),
                gu_w=_interleave16(W[b + "mlp.gate_proj.weight"].to(dev), W[b + "mlp.up_proj.weight"].to(dev)).to(bf),
                down_w=w(b + "mlp.down_proj.weight"),
                kv=torch.empty(max_seqs * max_seq_len, self.kv_w, dtype=bf, device=dev)))
        # the three wide decoder weights a second time in MFMA fragment order (ops.gemm_preshuffle, + 12 GB at the 7B geometry): the prefill GEMMs
        # that run the four-wave 256 x 256 tile then fetch their B fragments straight from global memory into registers (tile config 40,
        # gemm_w4.hip: half the LDS-DMA pieces and fragment reads per stage; bit-equal); the single-token passes keep streaming the row-major copy
        self.frag_weights = torch.device(dev).type == "cuda" if frag_weights is None else bool(frag_weights)
        if self.frag_weights:
            self.refresh_frag_weights()
        self.norm_w = f("model.norm.weight")
        self.lm_head = w("lm_head.weight")
        self.latent_q = W["model.latent_queries"].to(device=dev, dtype=bf).reshape(-1, H).contiguous()
        rows = max_seqs * max_seq_len
        self.x_in = torch.empty(rows, H, dtype=bf, device=dev)
        self.x = torch.empty(rows, H, dtype=f32, device=dev)
        self.h = torch.empty(rows, H, dtype=bf, device=dev)
        self.att = torch.empty(rows, H, dtype=bf, device=dev)
        self.qkv = torch.empty(rows, self.qkv_w, dtype=bf, device=dev)
        self.ff = torch.empty(rows, TI, dtype=bf, device=dev)
        self.cos = torch.empty(rows, self.hd, dtype=f32, device=dev)
        self.sin = torch.empty(rows, self.hd, dtype=f32, device=dev)
        self.hl = torch.empty(max_seqs * 8, H, dtype=bf, device=dev)
        self.xl = torch.empty(max_seqs, H, dtype=f32, device=dev)
        self.logits = torch.empty(max_seqs, cfg["vocab"], dtype=f32, device=dev)
        self.next_tok = torch.empty(max_seqs, dtype=torch.int32, device=dev)
        half = self.hd // 2
        self.inv_freq = (1.0 / (cfg["rope_theta"] ** (torch.arange(0, self.hd, 2, dtype=f32) / self.hd))).to(dev)
        axis = np.concatenate([np.full(16, 0), np.full(24, 1), np.full(24, 2)]).astype(np.int32)  # mrope_section [16, 24, 24]
        assert axis.size == half
        self.axis_of = torch.from_numpy(axis).to(dev)

    _BUFFERS = ("pv_perm", "xv", "hv", "attv", "qkvv", "ffv", "mh", "emb", "emb_tok", "v_cos", "v_sin", "x_in", "x", "h", "att", "qkv", "ff", "cos", "sin",
                "hl", "xl", "logits", "next_tok")

    def twin(self) -> "QwenVLEngine":
        """a second engine over the SAME weight tensors (nothing is copied; `latent_q` stays one shared parameter) with its own activation
        buffers and KV cache: two calls can then be in flight on two streams - the SFT trainer prefills the frozen prefix of micro-batch
        i + 1 in the twin while micro-batch i's latent-query rows still read this engine's cache (trainer.prefetch)."""
        import copy

        t = copy.copy(self)
        for n in self._BUFFERS:
            setattr(t, n, torch.empty_like(getattr(self, n)))
        t.layers = [dict(L, kv=torch.empty_like(L["kv"])) for L in self.layers]
        t._side = None
        return t

    # ------------------------------------------------------------------------------------------------ vision tower
    def plan_vision(self, grids: List[Tuple[int, int, int]]) -> dict:
        """host side of the vision tower for a list of image grids: window permutation, cu_seqlens, rope tables (uploaded once)."""
        dev, hd = self.device, self.vhd
        win_idx, cu_win = vision_window_permutation(grids, 2, self.cfg["v_window"], self.cfg["v_patch"])
        perm = (win_idx[:, None] * 4 + np.arange(4)[None]).reshape(-1).astype(np.int32)
        cu_full = np.concatenate([[0], np.cumsum([t * h * w for t, h, w in grids])]).astype(np.int32)
        cos, sin = vision_rope_tables(grids, hd)
        Np = int(cu_full[-1])
        if Np > self.Np_max:
            raise CapacityError(f"{Np} image patches exceed the engine's max_patches={self.Np_max}")
        return dict(Np=Np, perm=torch.from_numpy(perm).to(dev), cos=cos[perm].contiguous().to(dev), sin=sin[perm].contiguous().to(dev),
                    cu_win=torch.from_numpy(cu_win).to(dev), cu_full=torch.from_numpy(cu_full).to(dev),
                    max_win=int(np.diff(cu_win).max()), max_full=int(np.diff(cu_full).max()), inv=np.argsort(win_idx).astype(np.int32))

    def run_vision(self, vp: dict, pixel_values: torch.Tensor) -> torch.Tensor:
        """launch sequence of the vision tower (graph capturable): pixel_values bf16 [Np, 1176] -> embeds bf16 [Np/4, 3584], WINDOW order."""
        D, Hh, hd, Np = self.vD, self.vH, self.vhd, vp["Np"]
        assert pixel_values.shape[0] == Np and pixel_values.dtype == torch.bfloat16
        p0 = vp.get("p0", 0)                                      # first patch row of this launch sequence in the engine's buffers (split prefill)
        xp, x, h, att, qkv, ff = (t[p0:p0 + Np] for t in (self.pv_perm, self.xv, self.hv, self.attv, self.qkvv, self.ffv))
        ops.gather_rows(pixel_values, xp, src=vp["perm"])
        ops.linear(xp, self.v_patch, out=x)
        q3 = qkv.view(Np, 3, Hh, hd)
        for bi, blk in enumerate(self.v_blocks):
            ops.norm(x, blk["n1"], None, eps=1e-6, rms=True, out=h)
            ops.linear(h, blk["qkv_w"], bias=blk["qkv_b"], out=qkv)
            ops.rope(qkv, vp["cos"], vp["sin"], heads=2 * Hh, D=hd, col0=0, rows=Np)
            cu, mx = (vp["cu_full"], vp["max_full"]) if blk["full"] else (vp["cu_win"], vp["max_win"])
            ops.attention(q3[:, 0], q3[:, 1], q3[:, 2], cu_q=cu, cu_k=cu, max_q=mx, max_k=mx, out=att.view(Np, Hh, hd))
            ops.linear(att, blk["proj_w"], bias=blk["proj_b"], residual=x, out=x)
            ops.norm(x, blk["n2"], None, eps=1e-6, rms=True, out=h)
            ops.linear(h, blk["gu_w"], bias=blk["gu_b"], act="silu", glu=True, out=ff)
            ops.linear(ff, blk["down_w"], bias=blk["down_b"], residual=x, out=x)
            if self.tap is not None:
                self.tap("vit", bi, x)
        ops.norm(x, self.m_ln, None, eps=1e-6, rms=True, out=h)
        mh, emb = self.mh[p0 // 4: (p0 + Np) // 4], self.emb[p0 // 4: (p0 + Np) // 4]
        ops.linear(h.view(Np // 4, 4 * D), self.m0[0], bias=self.m0[1], act="gelu", out=mh)
        ops.linear(mh, self.m2[0], bias=self.m2[1], out=emb)
        return emb

    def vision(self, pixel_values: torch.Tensor, grids: List[Tuple[int, int, int]]):
        """pixel_values bf16 [Np, 1176] (HF processor patch layout) -> (embeds in WINDOW order, inv) with inv[k] = embed row of the
        k-th image token in sequence order."""
        vp = self.plan_vision(grids)
        return self.run_vision(vp, pixel_values), vp["inv"]

    def refresh_frag_weights(self):
        """(re)build the fragment-ordered copies of the wide decoder weights from the row-major ones - at load, and after anything that
        replaces or updates a layer weight in place (checkpoint reload, LoRA merge): the copies are not views"""
        self.frag_weights = True
        for L in self.layers:
            for k in ("qkv_w", "gu_w", "down_w"):
                n_, k_ = L[k].shape
                if n_ % 16 == 0 and k_ % 32 == 0:
                    L[k + "f"] = ops.gemm_preshuffle(L[k], out=L.get(k + "f"))
                else:
                    L[k + "f"] = None

    def drop_frag_weights(self):
        """release the fragment-ordered copies (12 GB at the 7B geometry): the prefill GEMMs fall back to tile config 39 / 18, same bits"""
        self.frag_weights = False
        for L in self.layers:
            for k in ("qkv_wf", "gu_wf", "down_wf"):
                L.pop(k, None)

    # ------------------------------------------------------------------------------------------------ text model
    def _phase(self, B: int, S: int, pos3: np.ndarray, cache_pos0, k_len: Optional[np.ndarray] = None, b0: int = 0) -> dict:
        """host side of one pass of the decoder stack over B x S new tokens: 3-D position ids, cache rows, key lengths.
        b0: first cache slot / sequence of the pass (a half batch of the split prefill runs sequences b0 .. b0 + B on buffer rows b0 * S ..)."""
        dev = self.device
        pos0 = np.broadcast_to(np.asarray(cache_pos0, dtype=np.int64).reshape(-1, 1), (B, 1))
        rows = ((b0 + np.arange(B))[:, None] * self.S_max + pos0 + np.arange(S)[None]).reshape(-1).astype(np.int32)
        if int(pos0.max()) + S > self.S_max:
            raise CapacityError(f"KV cache capacity exceeded: {int(pos0.max()) + S} tokens > max_seq_len={self.S_max}")
        ph = dict(B=B, S=S, pos=torch.from_numpy(np.ascontiguousarray(np.broadcast_to(pos3, (3, B, S)).reshape(3, B * S)).astype(np.int32)).to(dev),
                  rows=torch.from_numpy(rows).to(dev), Lk=int(pos0.max()) + S, k_len=None, b0=b0, r0=b0 * S)
        if k_len is not None:
            ph["k_len"] = torch.from_numpy(np.asarray(k_len, dtype=np.int32)).to(dev)
        # the new tokens of every sequence are its LAST S keys (what the attention launch assumes when it appends them itself, ina_attn_args.rope_cos)
        ph["new_are_last"] = bool(np.all((np.asarray(k_len).reshape(-1) if k_len is not None else ph["Lk"]) == pos0[:, 0] + S))
        return ph

    def _layers(self, ph: dict):
        """28 decoder layers over the phase's rows (x_in holds their bf16 input embeddings); K/V land at ph['rows'] of the cache."""
        H, nh, nkv, hd, Smax = self.H, self.nh, self.nkv, self.hd, self.S_max
        B, S = ph["B"], ph["S"]
        rows, r0, b0 = B * S, ph.get("r0", 0), ph.get("b0", 0)
        x, h, att, qkv, ff, x_in, cos, sin = (t[r0:r0 + rows] for t in (self.x, self.h, self.att, self.qkv, self.ff, self.x_in, self.cos, self.sin))
        ops.mrope_table(ph["pos"], self.inv_freq, self.axis_of, cos, sin)
        q4 = qkv[:, : nh * hd].view(B, S, nh, hd)
        # single-token decode passes (<= 16 rows): the two RMSNorms of a layer run inside the weight-streaming GEMMs that consume them
        # (ina_gemm_bf16 norm_gamma) and the rotary embedding + KV-cache append of the new token inside the attention launch
        # (ina_attn_args.rope_cos): 5 launches per layer - q|k|v, attention, o, gate|up, down - in a chain that is launch / latency bound
        single = rows <= 16 and self.tap is None
        fused_norm = single and self.fuse_decode_norm
        fuse_rope = single and self.fuse_decode_rope and ph.get("new_are_last", False) and ops.attention_rope_ok(S, ph["Lk"], nh, nkv, hd)
        if fuse_rope:
            kn4 = qkv[:, nh * hd:(nh + nkv) * hd].view(B, S, nkv, hd)
            vn4 = qkv[:, (nh + nkv) * hd:].view(B, S, nkv, hd)
        for li, L in enumerate(self.layers):
            src = x_in if li == 0 else x
            wf = (lambda k, L=L: L.get(k)) if (self.frag_weights and rows > 64) else (lambda k: None)
            if fused_norm:
                ops.linear(src, L["qkv_w"], bias=L["qkv_b"], out=qkv, prenorm=(L["n1"], 1e-6))
            else:
                ops.norm(src, L["n1"], None, eps=1e-6, rms=True, out=h)
                ops.linear(h, L["qkv_w"], bias=L["qkv_b"], out=qkv, w_frag=wf("qkv_wf"))
            if not fuse_rope:
                # m-rope on q (in place) and k, and the KV-cache append (rotated k | v -> cache row of every token) in ONE launch
                ops.rope(qkv, cos, sin, heads=nh + nkv, D=hd, col0=0, rows=rows, kv_out=L["kv"], kv_dst=ph["rows"], kv_head0=nh, v_heads=nkv)
            kv4 = L["kv"].view(self.B_max, Smax, 2, nkv, hd)[b0:b0 + B, : ph["Lk"]]
            ops.attention(q4, kv4[:, :, 0], kv4[:, :, 1], causal=True, out=att.view(B, S, nh, hd), k_len=ph["k_len"],
                          rope=(cos, sin, kn4, vn4) if fuse_rope else None)
            ops.linear(att, L["o_w"], residual=src, out=x)
            if fused_norm:
                ops.linear(x, L["gu_w"], act="silu", glu=True, out=ff, prenorm=(L["n2"], 1e-6))
            else:
                ops.norm(x, L["n2"], None, eps=1e-6, rms=True, out=h)
                ops.linear(h, L["gu_w"], act="silu", glu=True, out=ff, w_frag=wf("gu_wf"))
            ops.linear(ff, L["down_w"], residual=x, out=x, w_frag=wf("down_wf"))
            if self.tap is not None:
                self.tap("llm", li, x)

    def _last_logits(self, B: int, S: int, row_in_seq, rows_idx: Optional[torch.Tensor] = None):
        """final RMSNorm + lm_head on ONE row per sequence, greedy argmax on the device. row_in_seq: the same row for every sequence,
        or (ragged batches) rows_idx int32 [B] = absolute row of each sequence's last real token."""
        if rows_idx is not None:
            ops.gather_rows(self.x[: B * S], self.xl[:B], src=rows_idx)
            ops.norm(self.xl[:B], self.norm_w, None, eps=1e-6, rms=True, out=self.hl[:B], rows=B)
        else:
            ops.norm(self.x[: B * S], self.norm_w, None, eps=1e-6, rms=True, out=self.hl[:B], rows=B, in_map=(1, S, row_in_seq))
        ops.linear(self.hl[:B], self.lm_head, out=self.logits[:B])     # (152064 columns: one wave per 16-column tile owns all of K)
        ops.argmax_rows(self.logits[:B], self.next_tok[:B])

    # ---- plan / run: all host work up front, then a pure launch sequence (hipGraph capturable)
    def plan(self, input_ids, image_grid_thw, n_decode: int = 0, with_latents: bool = False, cached_embeds: Optional[list] = None,
             prefix_len=0) -> dict:
        """Host-side plan of one S2 call for B equal-length prompts: embedding / scatter indices, vision plan, position ids and cache
        rows of the prefill, of every decode step and of the latent-query pass (fixed-length answers of n_decode tokens).
        cached_embeds: one entry per image (prompt order over the batch), None = run the vision tower on it, else its merged embeddings
        bf16 [h*w/4, H] in token order from an earlier call (per-frame ViT cache, SURVEY.md 8f-1: the tower attends per image, so a
        frame's embeddings do not depend on what else is in the batch). pixel_values then holds the patches of the fresh images only.
        prefix_len (prefix-KV reuse, SURVEY.md 7 / DESIGN 8.5): the first prefix_len tokens of EVERY sequence are not run - their K/V of
        all layers are already in the cache slots of this batch (`import_prefix_kv`, from an earlier call whose prompt started with the
        same tokens: system prompt + instruction + first history frame between the System-2 calls of an episode). With a causal mask
        the K/V of a token depend on the tokens before it only, so the suffix sees exactly what a full prefill would have cached.
        Images whose tokens lie inside the prefix are not encoded (pixel_values holds the patches of the other images); the prefix
        must end on an image boundary. prefix_len may be an int or one value per sequence (0 = nothing cached for that sequence): the
        call then runs a right-padded rectangle of max(S - prefix_len) tokens per sequence behind each sequence's own prefix."""
        cfg, dev = self.cfg, self.device
        ids = (input_ids.cpu().numpy() if isinstance(input_ids, torch.Tensor) else np.asarray(input_ids)).astype(np.int64)
        B, S = ids.shape
        if B > self.B_max or S > self.S_max:
            raise CapacityError(f"System-2 batch of {B} x {S} tokens exceeds the engine's max_seqs={self.B_max} / max_seq_len={self.S_max}")
        pl = np.broadcast_to(np.asarray(prefix_len, dtype=np.int64).reshape(-1), (B,)).copy()
        assert int(pl.min()) >= 0 and int(pl.max()) < S, f"prefix_len={pl.tolist()} must leave at least one token of the {S}-token prompts to run"
        # the planned decode / latent passes read every sequence's first token at row S_run - 1 of its rectangle: only uniform prefixes put
        # the last real token there (ragged prefixes decode through the eager `decode()`, which carries per-sequence row indices) - ADVICE r3
        assert (n_decode == 0 and not with_latents) or int(pl.min()) == int(pl.max()), \
            f"plan(n_decode={n_decode}, with_latents={with_latents}) needs ONE prefix length for all sequences, got {pl.tolist()}: use prefill() + decode()"
        grids = [tuple(int(v) for v in g) for g in (image_grid_thw.tolist() if image_grid_thw is not None else [])]
        grids_all = grids
        Sr = S - int(pl.min())                                    # tokens per sequence this call runs (rectangle width)
        run = np.zeros((B, Sr), dtype=np.int64)                   # row b = ids[b, pl[b]:] right-padded
        for b in range(B):
            run[b, : S - pl[b]] = ids[b, pl[b]:]
        flat = run.reshape(-1)
        prefix_len = pl
        P = dict(B=B, S=S, S_run=Sr, prefix_len=prefix_len, n_decode=n_decode, ids=torch.from_numpy(flat.astype(np.int32)).to(dev), vision=None,
                 cached=[], fresh_tokens=[])
        img_pos = np.nonzero(flat == cfg["image_token_id"])[0].astype(np.int32)
        if grids:
            ntok_all = [t * h * w // 4 for t, h, w in grids]
            # images inside the prefix: their tokens are cached K/V, nothing of them is needed (neither pixels nor embeddings)
            in_prefix, img_seq, k = [], [], 0
            for b in range(B):
                ipos = np.nonzero(ids[b] == cfg["image_token_id"])[0]
                cum = 0
                while cum < ipos.size:
                    assert k < len(grids), "image count of the prompts and image_grid_thw disagree"
                    first, last = int(ipos[cum]), int(ipos[cum + ntok_all[k] - 1])
                    assert last < pl[b] or first >= pl[b], "the cached prefix must end on an image boundary"
                    in_prefix.append(last < pl[b])
                    img_seq.append(b)
                    cum += ntok_all[k]
                    k += 1
            assert k == len(grids), "image count of the prompts and image_grid_thw disagree"
            keep = [k for k, ins in enumerate(in_prefix) if not ins]
            P["images_run"] = keep
            cached_all = list(cached_embeds) if cached_embeds is not None else [None] * len(grids)
            assert len(cached_all) == len(grids)
            grids = [grids_all[k] for k in keep]
            cached_embeds = [cached_all[k] for k in keep]
            ntok = [ntok_all[k] for k in keep]
            assert img_pos.size == sum(ntok), f"Image features and image tokens do not match: tokens: {img_pos.size}, features {sum(ntok)}"
            off = np.concatenate([[0], np.cumsum(ntok)])
            cached = list(cached_embeds)
            fresh = [k for k, c in enumerate(cached) if c is None]
            if fresh:
                vp = self.plan_vision([grids[k] for k in fresh])
                dst = np.concatenate([img_pos[off[k]:off[k + 1]] for k in fresh])
                P["vision"], P["img_src"], P["img_dst"] = vp, torch.from_numpy(vp["inv"]).to(dev), torch.from_numpy(dst).to(dev)
                P["_split_info"] = (img_seq, ntok, off, img_pos) if (len(fresh) == len(grids_all) and int(pl.max()) == 0) else None
                o = 0
                for k in fresh:
                    P["fresh_tokens"].append((keep[k], o, o + ntok[k]))      # rows of emb_tok that hold image keep[k] (index over ALL images) after run_prefill
                    o += ntok[k]
            for k, c in enumerate(cached):
                if c is not None:
                    assert c.shape == (ntok[k], self.H) and c.dtype == torch.bfloat16, f"cached embeds of image {k}: {tuple(c.shape)} vs grid {grids[k]}"
                    P["cached"].append((c, torch.from_numpy(img_pos[off[k]:off[k + 1]]).to(dev)))
        traj_pos = np.nonzero(flat == cfg["traj_token_id"])[0].astype(np.int32)
        if traj_pos.size:
            nq = self.latent_q.shape[0]
            P["traj_src"] = torch.from_numpy((np.arange(traj_pos.size) % nq).astype(np.int32)).to(dev)
            P["traj_dst"] = torch.from_numpy(traj_pos).to(dev)
        pos3, _ = rope_index(ids, grids_all, cfg["image_token_id"], cfg["vision_start_id"])
        if int(pl.max()) == 0:
            P["prefill"] = self._phase(B, Sr, pos3, 0)
            info = P.pop("_split_info", None)
            if self.split_prefill and B >= 2 and info is not None:
                # two half batches (sequences [0, Ba) and [Ba, B)) with their own vision plans: same buffers, disjoint row ranges
                img_seq, ntok, off, img_pos = info
                P["split"] = []
                Ba = (B + 1) // 2
                for b0, b1 in ((0, Ba), (Ba, B)):
                    ks = [k for k in range(len(grids_all)) if b0 <= img_seq[k] < b1]
                    if not ks or ks != list(range(ks[0], ks[-1] + 1)):
                        P.pop("split")
                        break
                    vp = self.plan_vision([grids_all[k] for k in ks])
                    vp["p0"] = int(sum(t * h * w for t, h, w in grids_all[: ks[0]]))
                    P["split"].append(dict(vision=vp, img_src=torch.from_numpy(vp["inv"]).to(dev), tok0=int(off[ks[0]]), n_tok=int(off[ks[-1] + 1] - off[ks[0]]),
                                           img_dst=torch.from_numpy(np.concatenate([img_pos[off[k]:off[k + 1]] for k in ks])).to(dev),
                                           prefill=self._phase(b1 - b0, Sr, pos3[:, b0:b1], 0, b0=b0)))
        P.pop("_split_info", None)
        if int(pl.max()) != 0:
            pos_run = np.zeros((3, B, Sr), dtype=np.int64)
            for b in range(B):
                n = S - pl[b]
                pos_run[:, b, :n] = pos3[:, b, pl[b]:]
                pos_run[:, b, n:] = pos3[:, b, -1:]                # pad rows: any position (never attended by real tokens, causal)
            uniform = bool((pl == pl[0]).all())
            # per-sequence prefixes: query i of sequence b sees keys <= pl[b] + i (the kernels align the causal mask to each k_len)
            P["prefill"] = self._phase(B, Sr, pos_run, pl, k_len=None if uniform else pl + Sr)
        nxt = pos3[:, :, -1].max(axis=0) + 1                      # text position of the first generated token, per sequence
        P["next_pos"] = nxt
        P["decode"] = [self._phase(B, 1, (nxt + j)[None, :, None], S + j) for j in range(max(n_decode - 1, 0))]
        if with_latents:
            nq = self.latent_q.shape[0]
            cur = S + max(n_decode - 1, 0)                        # cached tokens after the decode steps
            m = 1 if n_decode > 0 else 0                          # the last sampled token has no K/V yet
            p = (nxt + (cur - S))[None, :, None] + np.arange(m + nq)[None, None, :]
            P["latents"] = self._phase(B, m + nq, p, cur)
            P["lat_m"] = m
        return P

    def run_prefill(self, P: dict, pixel_values: Optional[torch.Tensor]):
        ops.gather_rows(self.embed, self.x_in, src=P["ids"])
        if self.split_prefill and P.get("split") and self.tap is None:
            if "traj_src" in P:
                ops.gather_rows(self.latent_q, self.x_in, src=P["traj_src"], dst=P["traj_dst"])
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
            self._side.wait_stream(main)

            def half(s):
                vp = s["vision"]
                emb = self.run_vision(vp, pixel_values[vp["p0"]: vp["p0"] + vp["Np"]])
                tok = self.emb_tok[s["tok0"]: s["tok0"] + s["n_tok"]]
                ops.gather_rows(emb, tok, src=s["img_src"])                   # window order -> image-token order (kept for the frame cache)
                ops.gather_rows(tok, self.x_in, dst=s["img_dst"])
                self._layers(s["prefill"])

            # both halves' GEMMs run beside each other: tiles are picked without the quantisation charge (ops.shared_tail; decoder chain of
            # 3 + 3 envs 61.8 vs 64.2 ms, 4 + 3 envs 73.4 vs 74.5 ms, vision chain 28.0 vs 28.7 ms: profiles/r03x_native_chain_partitions.log)
            with ops.shared_tail():
                if self.split_serial:        # profiling: the SAME launches (half batches, shared-tail tile selection) back to back on one stream
                    half(P["split"][1])
                else:
                    with torch.cuda.stream(self._side):
                        half(P["split"][1])
                half(P["split"][0])
            main.wait_stream(self._side)
            return
        if P["vision"] is not None:
            emb = self.run_vision(P["vision"], pixel_values)
            n = P["img_src"].shape[0]
            ops.gather_rows(emb, self.emb_tok[:n], src=P["img_src"])          # window order -> image-token order (kept for the frame cache)
            ops.gather_rows(self.emb_tok[:n], self.x_in, dst=P["img_dst"])
        for c, dst in P["cached"]:
            ops.gather_rows(c, self.x_in, dst=dst)
        if "traj_src" in P:
            ops.gather_rows(self.latent_q, self.x_in, src=P["traj_src"], dst=P["traj_dst"])
        self._layers(P["prefill"])

    def fresh_image_embeds(self, P: dict) -> Dict[int, torch.Tensor]:
        """image index -> a copy of its merged embeddings bf16 [h*w/4, H] (token order) as computed by the last run_prefill(P):
        what a caller stores in its per-frame cache and hands back through plan(..., cached_embeds=...)."""
        return {k: self.emb_tok[a:b].clone() for k, a, b in P["fresh_tokens"]}

    def run_decode(self, P: dict, tokens_out: torch.Tensor, j0: int = 0, j1: Optional[int] = None):
        """n_decode greedy tokens: the first from the prompt's last position, then n_decode - 1 single-token passes.
        [j0, j1): the tokens this call produces (default: all) - the chain can be issued (and captured) in pieces, e.g. to let another stream
        start between two passes; the pieces hand over through the engine's buffers (next token, KV cache) only."""
        B, S, n = P["B"], P["S_run"], P["n_decode"]               # rows of x: the tokens this call ran (all of them without a cached prefix)
        j1 = n if j1 is None else min(j1, n)
        if n == 0 or j0 >= j1:
            return
        if j0 == 0:
            self._last_logits(B, S, S - 1)
        for j in range(j0, j1):
            tokens_out[:, j].copy_(self.next_tok[:B])
            if j == n - 1:
                break
            ops.gather_rows(self.embed, self.x_in, src=self.next_tok[:B], rows=B)
            self._layers(P["decode"][j])
            self._last_logits(B, 1, 0)

    def run_latents(self, P: dict, out: torch.Tensor):
        """N_QUERY latent queries (behind the last sampled token) against the KV cache -> out bf16 [B, N_QUERY, H]."""
        B, H, m = P["B"], self.H, P["lat_m"]
        nq = self.latent_q.shape[0]
        x3 = self.x_in[: B * (m + nq)].view(B, m + nq, H)
        if m:
            tmp = self.hl[:B]
            ops.gather_rows(self.embed, tmp, src=self.next_tok[:B], rows=B)
            x3[:, 0].copy_(tmp)
        x3[:, m:].copy_(self.latent_q.view(1, nq, H).expand(B, nq, H))
        self._layers(P["latents"])
        ops.norm(self.x[: B * (m + nq)], self.norm_w, None, eps=1e-6, rms=True, out=out.view(B * nq, H), rows=B * nq, in_map=(nq, m + nq, m))

    def run_s2(self, P: dict, pixel_values, tokens_out: torch.Tensor, latents_out: Optional[torch.Tensor]):
        """whole System-2 call as one launch sequence: ViT + prefill + n_decode greedy tokens (+ latent queries)."""
        self.run_prefill(P, pixel_values)
        self.run_decode(P, tokens_out)
        if latents_out is not None:
            self.run_latents(P, latents_out)

    # ---- prefix-KV reuse: the K/V of a prompt prefix leave / re-enter the batch's cache slots (device copies, no arithmetic)
    def images_in_prefix(self, input_ids, image_grid_thw, prefix_len) -> List[bool]:
        """per image of the batch (prompt order): do all of its tokens lie inside its sequence's cached prefix?"""
        ids = (input_ids.cpu().numpy() if isinstance(input_ids, torch.Tensor) else np.asarray(input_ids)).astype(np.int64)
        pl = np.broadcast_to(np.asarray(prefix_len, dtype=np.int64).reshape(-1), (ids.shape[0],))
        ntok = [int(t * h * w) // 4 for t, h, w in (image_grid_thw.tolist() if image_grid_thw is not None else [])]
        out, k = [], 0
        for b in range(ids.shape[0]):
            ipos = np.nonzero(ids[b] == self.cfg["image_token_id"])[0]
            cum = 0
            while cum < ipos.size:
                out.append(int(ipos[cum + ntok[k] - 1]) < pl[b])
                cum += ntok[k]
                k += 1
        return out

    def export_prefix_kv(self, seq: int, n_tokens: int) -> torch.Tensor:
        """K/V of the first n_tokens cached tokens of batch slot `seq`, all layers: bf16 [layers, n_tokens, 2 * kv_heads * head_dim]
        (a copy: what a caller keeps per environment between the System-2 calls of an episode)."""
        assert 0 <= seq < self.B_max and 0 < n_tokens <= self.S_max
        return torch.stack([L["kv"].view(self.B_max, self.S_max, self.kv_w)[seq, :n_tokens] for L in self.layers])

    def import_prefix_kv_batch(self, kv: torch.Tensor):
        """kv bf16 [m, layers, n, kv_w]: the prefixes of m environments into batch slots 0 .. m-1 (one strided copy per layer)."""
        m, nl, n, w = kv.shape
        assert nl == len(self.layers) and w == self.kv_w and m <= self.B_max and n <= self.S_max and kv.dtype == torch.bfloat16
        for li, L in enumerate(self.layers):
            L["kv"].view(self.B_max, self.S_max, self.kv_w)[:m, :n].copy_(kv[:, li])

    def import_prefix_kv(self, seq: int, kv: torch.Tensor):
        """put an exported prefix back into batch slot `seq` (before plan(..., prefix_len=kv.shape[1]) / prefill(..., prefix_len=))."""
        n = kv.shape[1]
        assert kv.shape == (len(self.layers), n, self.kv_w) and kv.dtype == torch.bfloat16 and n <= self.S_max
        for li, L in enumerate(self.layers):
            L["kv"].view(self.B_max, self.S_max, self.kv_w)[seq, :n].copy_(kv[li])

    # ---- eager, stateful API (used by the policy layer: answers have data-dependent lengths)
    def prefill(self, input_ids: torch.Tensor, pixel_values: Optional[torch.Tensor], image_grid_thw, cached_embeds: Optional[list] = None,
                seq_lens=None, prefix_len: int = 0) -> dict:
        """seq_lens [B] (optional): RAGGED batch - input_ids is right-padded to a common length S and sequence b has seq_lens[b] real
        tokens. Causal attention keeps every real position independent of the padding behind it, so the prefill runs on the padded
        rectangle; the first token is read at each sequence's own last position and the decode / latent passes append at per-sequence
        cache positions with per-sequence key lengths (the pad rows' K/V are never attended and get overwritten)."""
        P = self.plan(input_ids, image_grid_thw, cached_embeds=cached_embeds, prefix_len=prefix_len)
        self.run_prefill(P, pixel_values)
        return self.prefill_state(P, input_ids, image_grid_thw, seq_lens)

    def prefill_state(self, P: dict, input_ids, image_grid_thw, seq_lens=None) -> dict:
        """the host-side state `prefill` returns for a plan that has been run (decode / latent passes continue from it): a function of the
        prompt GEOMETRY only (lengths, image-token positions), not of the token values - a captured prefill can be replayed on new tokens
        and pixels of the same geometry and continue from a copy of this state (trainer: graphed frozen prefix)."""
        st = dict(B=P["B"], S=P["S"], S_run=P["S_run"], next_pos=P["next_pos"].copy(), plan=P)
        if seq_lens is not None:
            lens = np.asarray(seq_lens, dtype=np.int64)
            assert lens.shape == (P["B"],) and int(lens.max()) <= P["S"] and int(lens.min()) >= 1
            if bool((lens != P["S"]).any()):
                assert bool((lens > P["prefix_len"]).all()), "every sequence must have tokens behind its cached prefix"
                ids = (input_ids.cpu().numpy() if isinstance(input_ids, torch.Tensor) else np.asarray(input_ids)).astype(np.int64)
                grids = [tuple(int(v) for v in g) for g in (image_grid_thw.tolist() if image_grid_thw is not None else [])]
                pos3, _ = rope_index(ids, grids, self.cfg["image_token_id"], self.cfg["vision_start_id"])
                st["next_pos"] = np.asarray([int(pos3[:, b, : lens[b]].max()) + 1 for b in range(P["B"])], dtype=np.int64)
                st["lens"] = lens
        return st

    def decode(self, state: dict, n_steps: int) -> torch.Tensor:
        """n greedy steps after prefill (or after a previous decode); returns int32 [B, n] generated tokens (device). The last
        returned token is sampled but not yet run through the layers (its K/V are not cached)."""
        B, S = state["B"], state["S"]
        Sr = state.get("S_run", S)                                # rows per sequence in x (prompt minus a cached prefix)
        out = torch.empty(B, n_steps, dtype=torch.int32, device=self.device)
        lens = state.get("lens")
        if "cur" not in state:
            if lens is None and bool((state["plan"]["prefix_len"] == state["plan"]["prefix_len"][0]).all()):
                self._last_logits(B, Sr, Sr - 1)
                state["cur"] = S
            elif lens is None:       # equal-length prompts behind prefixes of different lengths: each sequence's last real row
                rows = torch.from_numpy((np.arange(B) * Sr + S - state["plan"]["prefix_len"] - 1).astype(np.int32)).to(self.device)
                self._last_logits(B, Sr, None, rows_idx=rows)
                state["cur"] = S
            else:
                rows = torch.from_numpy((np.arange(B) * Sr + lens - state["plan"]["prefix_len"] - 1).astype(np.int32)).to(self.device)
                self._last_logits(B, Sr, None, rows_idx=rows)
                state["cur"] = lens.copy()
        for j in range(n_steps):
            out[:, j].copy_(self.next_tok[:B])
            if j == n_steps - 1:
                break
            cur = state["cur"]
            ops.gather_rows(self.embed, self.x_in, src=self.next_tok[:B], rows=B)
            if lens is None:
                self._layers(self._phase(B, 1, state["next_pos"][None, :, None], cur))
            else:
                self._layers(self._phase(B, 1, state["next_pos"][None, :, None], cur, k_len=cur + 1))
            self._last_logits(B, 1, 0)
            state["cur"] = cur + 1
            state["next_pos"] = state["next_pos"] + 1
        return out

    def latents(self, state: dict, tail_tokens: Optional[torch.Tensor], seq_lens: Optional[np.ndarray] = None) -> torch.Tensor:
        """N_QUERY latent trajectory queries against the cached prefix (generate_latents without the re-run).
        tail_tokens int32 [B, m]: tokens to run in front of the queries because their K/V are not cached yet (the last sampled
        token). seq_lens [B]: number of cached tokens to keep per sequence (prompt + answer up to where it ended); the m + N_QUERY
        new tokens are placed right behind them, K/V cached beyond that point (post-EOS decode steps) are ignored / overwritten."""
        B, H = state["B"], self.H
        nq = self.latent_q.shape[0]
        m = 0 if tail_tokens is None else tail_tokens.shape[1]
        rows = B * (m + nq)
        cur = state.get("cur", state.get("lens", state["S"]))
        x3 = self.x_in[:rows].view(B, m + nq, H)
        if m:
            tmp = self.hl[: B * m]
            ops.gather_rows(self.embed, tmp, src=tail_tokens.reshape(-1).contiguous(), rows=B * m)
            x3[:, :m].copy_(tmp.view(B, m, H))
        x3[:, m:].copy_(self.latent_q.view(1, nq, H).expand(B, nq, H))
        if seq_lens is None and np.ndim(cur) > 0:
            seq_lens = cur                                            # ragged batch: keep everything cached so far, per sequence
        if seq_lens is None:
            p = state["next_pos"][None, :, None] + np.arange(m + nq)[None, None, :]
            ph = self._phase(B, m + nq, p, cur)
        else:
            seq_lens = np.asarray(seq_lens, dtype=np.int64)
            start_pos = state["next_pos"] - (cur - seq_lens)          # text positions advance by one per cached token
            p = start_pos[None, :, None] + np.arange(m + nq)[None, None, :]
            ph = self._phase(B, m + nq, p, seq_lens, k_len=seq_lens + m + nq)
        self._layers(ph)
        out = torch.empty(B, nq, H, dtype=torch.bfloat16, device=self.device)
        ops.norm(self.x[:rows], self.norm_w, None, eps=1e-6, rms=True, out=out.view(B * nq, H), rows=B * nq, in_map=(nq, m + nq, m))
        return out

    # ------------------------------------------------------------------------------------------------ HF-style surface
    def generate(self, input_ids, pixel_values=None, image_grid_thw=None, max_new_tokens: int = 8, eos_token_id=None, **_):
        """greedy generate: returns sequences int64 [B, S + n] (a row that reached EOS keeps EOS), n = max_new_tokens."""
        state = self.prefill(input_ids, pixel_values, image_grid_thw)
        toks = self.decode(state, max_new_tokens).cpu().long()
        eos = self.cfg["eos_token_id"] if eos_token_id is None else eos_token_id
        for b in range(toks.shape[0]):
            hit = (toks[b] == eos).nonzero()
            if hit.numel():
                toks[b, int(hit[0]) + 1:] = eos
        self._state = state
        return torch.cat([input_ids.cpu().long(), toks], dim=1)

    def generate_latents(self, output_ids, pixel_values=None, image_grid_thw=None, cached_embeds: Optional[list] = None):
        """reference signature (internvla_n1.py:320): full prefill over output_ids + N_QUERY latent queries (no cache reuse)."""
        cfg = self.cfg
        nq = self.latent_q.shape[0]
        ids = torch.cat([output_ids.cpu().long(), torch.full((output_ids.shape[0], nq), cfg["traj_token_id"], dtype=torch.long)], dim=1)
        P = self.plan(ids, image_grid_thw, cached_embeds=cached_embeds)
        self.run_prefill(P, pixel_values)
        B, S = P["B"], P["S"]
        out = torch.empty(B, nq, self.H, dtype=torch.bfloat16, device=self.device)
        ops.norm(self.x[: B * S], self.norm_w, None, eps=1e-6, rms=True, out=out.view(B * nq, self.H), rows=B * nq, in_map=(nq, S, S - nq))
        return out
