"""Gradient of the SFT loss w.r.t. `latent_queries` through the FROZEN Qwen2.5-VL decoder.

In the reference the trajectory tokens sit at `t_s_pos` of the training sequence, their embedding rows are overwritten with the
trainable `latent_queries` (internvla_n1.py:166-172), the whole sequence runs through the LLM and autograd walks all S rows of all
28 layers back (internvla_n1_trainer.py:116 `latent_queries.requires_grad = True` with every LLM weight frozen).
With causal attention only the N_QUERY rows themselves depend on `latent_queries`: every earlier row - and its K / V - is a constant,
and later rows never reach the loss (the loss reads hidden_states[t_s_pos : t_s_pos + n_query], :224-227). So the backward pass here
touches B * n_query rows per layer: the forward is the engine's cached latent-query pass (prefix K/V from the prefill of the same
step = the activation checkpoint), the backward streams each frozen weight matrix once in its stored layout (ina_gemm_nn_bf16, HBM
bound, no transposed copies of 7.6 B parameters) and differentiates attention against the KV cache (ina_attention_bwd_bf16, GQA,
causal, only the query rows' own K / V receive gradients).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from . import train_ops as T
from .qwen_vl import QwenVLEngine

BF, F32 = torch.bfloat16, torch.float32


class LatentQueryGrad:
    def __init__(self, engine: QwenVLEngine):
        self.e = engine
        self._ctx = None

    def forward(self, state: dict) -> torch.Tensor:
        """the N_QUERY latent queries right behind the cached tokens of `state` (QwenVLEngine.prefill of the tokens before t_s_pos)
        -> final-norm hidden states bf16 [B, N_QUERY, H]; keeps what backward() needs. Same arithmetic as QwenVLEngine.latents except
        that the SwiGLU gate / up pre-activations are materialised (bf16) for the derivative."""
        e = self.e
        B, H, nh, nkv, hd, TI = state["B"], e.H, e.nh, e.nkv, e.hd, e.TI
        nq = e.latent_q.shape[0]
        R = B * nq
        assert R <= 16, "the weight-streaming dX kernel handles up to 16 rows (B * n_query): split the batch"
        cur = state.get("cur", state.get("lens", state["S"]))
        if np.ndim(cur) > 0:
            seq_lens = np.asarray(cur, dtype=np.int64)
            p = state["next_pos"][None, :, None] + np.arange(nq)[None, None, :]
            ph = e._phase(B, nq, p, seq_lens, k_len=seq_lens + nq)
        else:
            p = state["next_pos"][None, :, None] + np.arange(nq)[None, None, :]
            ph = e._phase(B, nq, p, cur)
        ops.mrope_table(ph["pos"], e.inv_freq, e.axis_of, e.cos, e.sin)
        cos, sin = e.cos[:R].clone(), e.sin[:R].clone()
        src = e.latent_q.view(1, nq, H).expand(B, nq, H).reshape(R, H).contiguous()
        saves = []
        for L in e.layers:
            h = ops.norm(src, L["n1"], None, eps=1e-6, rms=True)
            qkv = ops.linear(h, L["qkv_w"], bias=L["qkv_b"])
            ops.rope(qkv, cos, sin, heads=nh + nkv, D=hd, col0=0, rows=R, kv_out=L["kv"], kv_dst=ph["rows"], kv_head0=nh, v_heads=nkv)
            kv4 = L["kv"].view(e.B_max, e.S_max, 2, nkv, hd)[:B, : ph["Lk"]]
            att = ops.attention(qkv[:, : nh * hd].view(B, nq, nh, hd), kv4[:, :, 0], kv4[:, :, 1], causal=True, k_len=ph["k_len"])
            x1 = ops.linear(att.view(R, H), L["o_w"], residual=src, out_dtype=F32)
            h2 = ops.norm(x1, L["n2"], None, eps=1e-6, rms=True)
            gu = ops.linear(h2, L["gu_w"])                           # [R, 2*TI], columns interleaved [gate16 | up16] like the weight rows
            g2 = gu.view(-1, 32)
            ff = T.glu_fwd(g2[:, :16], g2[:, 16:]).view(R, TI)
            x2 = ops.linear(ff, L["down_w"], residual=x1, out_dtype=F32)
            saves.append((src, qkv, att, x1, gu))
            src = x2
        hq = ops.norm(src, e.norm_w, None, eps=1e-6, rms=True)
        self._ctx = dict(B=B, nq=nq, ph=ph, cos=cos, neg_sin=-sin, saves=saves, x_final=src)
        return hq.view(B, nq, H)

    def backward(self, d_hidden: torch.Tensor) -> torch.Tensor:
        """d loss / d hidden states [B, N_QUERY, H] -> d loss / d latent_queries f32 [N_QUERY, H] (summed over the batch)."""
        e, c = self.e, self._ctx
        assert c is not None, "forward() first"
        B, nq, ph = c["B"], c["nq"], c["ph"]
        H, nh, nkv, hd = e.H, e.nh, e.nkv, e.hd
        R, G = B * nq, nh // nkv
        dx, _ = T.norm_bwd(c["x_final"], d_hidden.reshape(R, H).contiguous(), e.norm_w, 1e-6, True, dx_dtype=F32)
        for L, (src, qkv, att, x1, gu) in zip(reversed(e.layers), reversed(c["saves"])):
            d_ff = T.gemm_nn(T.affine(dx, out_dtype=BF), L["down_w"], out_dtype=BF)                  # [R, TI]
            d_gu = torch.empty_like(gu)
            g2, d2 = gu.view(-1, 32), d_gu.view(-1, 32)
            T.glu_bwd(g2[:, :16], g2[:, 16:], d_ff.view(-1, 16), da=d2[:, :16], db=d2[:, 16:])
            d_h2 = T.gemm_nn(d_gu, L["gu_w"])                                                        # f32 [R, H]
            T.norm_bwd(x1, d_h2, L["n2"], 1e-6, True, dx=dx, accumulate=True)                        # dx is now d loss / d x1
            d_att = T.gemm_nn(T.affine(dx, out_dtype=BF), L["o_w"], out_dtype=BF)                    # [R, H]
            dqkv = torch.empty_like(qkv)
            kv4 = L["kv"].view(e.B_max, e.S_max, 2, nkv, hd)[:B, : ph["Lk"]]
            q4 = qkv[:, : nh * hd].view(B, nq, nh, hd)
            _, dk, dv = T.attention_bwd(q4, kv4[:, :, 0], kv4[:, :, 1], att, d_att.view(B, nq, nh, hd), causal=True, k_len=ph["k_len"],
                                        dq=dqkv[:, : nh * hd].view(B, nq, nh, hd), kv_row0=-1)
            dqkv[:, nh * hd: (nh + nkv) * hd] = dk.float().view(R, nkv, G, hd).sum(2).view(R, nkv * hd)   # sum the query heads of a GQA group
            dqkv[:, (nh + nkv) * hd:] = dv.float().view(R, nkv, G, hd).sum(2).view(R, nkv * hd)
            ops.rope(dqkv, c["cos"], c["neg_sin"], heads=nh + nkv, D=hd, col0=0, rows=R)              # transpose of the rotation
            d_h = T.gemm_nn(dqkv, L["qkv_w"])
            T.norm_bwd(src, d_h, L["n1"], 1e-6, True, dx=dx, accumulate=True)                        # d loss / d (layer input)
        return dx.view(B, nq, H).sum(0)
