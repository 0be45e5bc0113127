"""InternVLA-N1 DualVLN System-1 (`nextdit_async`) on the gfx950 op library - batched over environments.

Mirrors the 'nextdit' + 'async' branch of `InternVLAN1ForCausalLM.generate_traj`
(/root/reference/internnav/model/basemodel/internvla_n1/internvla_n1.py:349-432): cond_projector, DINOv2 ViT-S on the two
look-down frames, MemoryEncoder, QFormer (internvla_n1_arch.py:76-118), then 10 flow-matching Euler steps of the 12-layer
NextDiT (nextdit_traj.py:121-178, 299-368) over 32 sampled trajectories per environment.

Hoisted out of the sampler loop without changing results:
  * condition tokens, caption_projection, pooled caption embedding and the cross-attention K/V of every DiT layer depend only
    on the environment: computed once per call, shared by the 32 samples through the attention kernel's kv_bdiv broadcast;
  * the timestep embedding is input independent: tabulated per sampler step at load;
  * classifier-free guidance with guidance_scale == 1.0 evaluates u + 1.0*(c - u): the null-condition half has zero weight and
    is not computed (c is returned; equal up to one fp32 rounding of the reference's expression);
  * norm_out.linear_2 and action_decoder are two consecutive Linear maps with no nonlinearity between them: composed at load
    into one [3, 384] head (exact algebra, fp32) evaluated in the fused norm + head + Euler-update kernel;
  * the 12 adaLN modulation projections + norm_out.linear_1 share their input SiLU(temb): one concatenated GEMM per step.

HBM layout (B envs, S = 32 samples, T = 32 tokens, C = 384):
  z      bf16 [B*36, 768]   condition tokens per env: 32 QFormer memory tokens | 4 projected VLM latents
  enc    bf16 [B*36, 384]   caption_projection(z);  kv2[l] bf16 [B*36, 768] per-layer cross-attention K|V
  x      f32  [B*S*T, 384]  DiT residual stream;  qkvq bf16 [B*S*T, 1536] = q1|k1|v1|q2;  ff bf16 [B*S*T, dit_ffn]
  mod    f32  [B, 12*1536 + 384] adaLN scale/gate vectors of all layers + norm_out scale, refreshed per step
  sample f32  [B*S*T, 3]    trajectories being integrated (updated in place)
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from . import ops
from .navdp import IMAGENET_MEAN, IMAGENET_STD, _DecoderLayer, _SeqWorkspace, decoder_layer_postnorm
from .vit_s import DinoV2Encoder, VitWorkspace


def _interleave16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """rows of the GLU GEMM weight: 16-row blocks [gate16 | up16] (layout expected by ina_gemm_bf16 glu mode)."""
    n, k = a.shape
    return torch.stack([a.view(n // 16, 16, k), b.view(n // 16, 16, k)], dim=1).reshape(2 * n, k).contiguous()


class NextDiTSystem1:
    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: dict, device="cuda:0", max_envs: int = 64, use_async: bool = True, row_chain: bool = True):
        """use_async: the 'async' System-1 types condition the DiT on [32 memory tokens of the two look-down frames | projected VLM latents]
        (internvla_n1.py:364-381); without it ('nextdit': :382-383) the condition is the n_query projected latents alone - no DINOv2,
        MemoryEncoder or QFormer weights are read."""
        dev = torch.device(device)
        bf, f32 = torch.bfloat16, torch.float32
        sd = state_dict
        self.cfg, self.device, self.b_max = cfg, dev, max_envs
        # everything of a block between two attention stages that is local to a row as TWO launches (csrc/dit_rowchain.hip):
        # [attn2.to_out + norm2 / gate / residual + ffn_norm1 + linear_1/3 SwiGLU] and [linear_2 + ffn_norm2 / gate / residual + the next block's
        # norm1 + its fused q1|k1|v1|q2 projection]; the bf16 projection and the pre-normed GEMM operand stay in registers. Used from 16 k rows
        # (16 envs) on - below that the one-workgroup-per-128-rows grid leaves the chip empty, as for the row-panel GEMMs it is built from.
        # The chain is built for dim 384 and the two FFN widths the reference's block can have (1536 under its pinned diffusers 0.33.1, 1024
        # under <= 0.32: synthetic.lumina_ffn_width); its 128-row panels must not straddle two environments (S * T % 128 == 0). Any other
        # geometry takes the GEMM + chained-norm launches below - same results, more launches.
        self.row_chain = (bool(row_chain) and cfg["dit_dim"] == 384 and cfg["dit_ffn"] in (1024, 1536)
                          and (cfg["sample_num"] * cfg["predict_size"]) % 128 == 0)
        self.chain_min_rows = 16384
        # The two chain launches are selected separately (the unselected half runs its GEMM / norm launches). The second one pays inside the policy
        # step at FFN 1024 (+ 0.9 %) and costs at FFN 1536 (- 1.0 %: its first GEMM re-reads the 32 x 1536 SwiGLU panel once per column tile, and its
        # long workgroups hold the CUs the concurrent decode chain wants) - profiles/r06i_rowchain_halves_in_step.txt
        self.chain_a, self.chain_b = True, cfg["dit_ffn"] <= 1024
        # the chain's second launch also hands the attention stage the LayerNorm statistics of the projection rows it writes (blocks 1 ...):
        # dit_attention then reads every row once and runs without its statistics pass / barrier
        self.chain_stats = True
        D, L, S, T = cfg["dit_dim"], cfg["latent_dim"], cfg["sample_num"], cfg["predict_size"]
        self.D, self.L, self.S, self.T, self.nq = D, L, S, T, cfg["n_query"]
        self.Fr = cfg["memory_frames"]
        self.use_async = bool(use_async)
        self.Lz = (32 if self.use_async else 0) + self.nq
        self.nl, self.nh = cfg["dit_layers"], cfg["dit_heads"]

        def w(k):
            return sd[k].to(device=dev, dtype=bf).contiguous()

        def f(k):
            return sd[k].to(device=dev, dtype=f32).contiguous()

        # ---- condition path
        self.cp = [(w("cond_projector.0.weight"), f("cond_projector.0.bias")), (w("cond_projector.2.weight"), f("cond_projector.2.bias"))]
        nm = self.Fr * 256
        self.nm = nm
        if self.use_async:
            self.vit = DinoV2Encoder(sd, "rgb_model.", dev)
            self.vit_ws = VitWorkspace(max_envs * self.Fr, dev)
            self.mem_pos = sd["memory_encoder.memory_pos"][:nm].to(device=dev, dtype=f32).contiguous()
        self.me_layers = []
        for i in range(3 if self.use_async else 0):
            p = f"memory_encoder.encoder.layers.{i}"
            self.me_layers.append(dict(sa_w=w(p + ".self_attn.in_proj_weight"), sa_b=f(p + ".self_attn.in_proj_bias"),
                                       sa_ow=w(p + ".self_attn.out_proj.weight"), sa_ob=f(p + ".self_attn.out_proj.bias"),
                                       l1w=w(p + ".linear1.weight"), l1b=f(p + ".linear1.bias"), l2w=w(p + ".linear2.weight"),
                                       l2b=f(p + ".linear2.bias"), n1=(f(p + ".norm1.weight"), f(p + ".norm1.bias")),
                                       n2=(f(p + ".norm2.weight"), f(p + ".norm2.bias"))))
        if self.use_async:
            self.me_ws = _SeqWorkspace(max_envs * nm, D, 2048, dev)
            self.memcat = torch.empty(max_envs * nm, L, dtype=bf, device=dev)      # [feat | memory_feat] per token
            self.q_layers = [_DecoderLayer(sd, f"rgb_resampler.decoder.layers.{i}", dev, L) for i in range(3)]
            self.q_init = (sd["rgb_resampler.query_tokens"].float() + sd["rgb_resampler.query_pos"].float()).to(dev).contiguous()
            self.q_ws = _SeqWorkspace(max_envs * 32, L, 2048, dev)
            self.q_kv = torch.empty(max_envs * nm, 2 * L, dtype=bf, device=dev)
        self.cp_h = torch.empty(max_envs * self.nq, L, dtype=bf, device=dev)
        self.z = torch.empty(max_envs * self.Lz, L, dtype=bf, device=dev)
        # ---- DiT
        p = "traj_dit.model."
        self.cap = [(w(p + "caption_projection.linear_1.weight"), f(p + "caption_projection.linear_1.bias")),
                    (w(p + "caption_projection.linear_2.weight"), f(p + "caption_projection.linear_2.bias"))]
        self.cap_h = torch.empty(max_envs * self.Lz, D, dtype=bf, device=dev)
        self.enc = torch.empty(max_envs * self.Lz, D, dtype=bf, device=dev)
        self.enc_n = torch.empty(max_envs * self.Lz, D, dtype=bf, device=dev)
        self.ce_ln = (f(p + "time_caption_embed.caption_embedder.0.weight"), f(p + "time_caption_embed.caption_embedder.0.bias"))
        self.ce_w, self.ce_b = w(p + "time_caption_embed.caption_embedder.1.weight"), f(p + "time_caption_embed.caption_embedder.1.bias")
        self.pool = torch.empty(max_envs, D, dtype=f32, device=dev)
        self.pool_n = torch.empty(max_envs, D, dtype=bf, device=dev)
        self.cap_emb = torch.empty(max_envs, D, dtype=f32, device=dev)
        self.silu_temb = torch.empty(max_envs, D, dtype=bf, device=dev)
        # sampler schedule: sigmas = linspace(1, 1/n, n) (+ 0), timesteps = sigma * 1000 cast to long (internvla_n1.py:396-397,419)
        n = cfg["num_inference_steps"]
        sig = np.linspace(1.0, 1.0 / n, n).astype(np.float32)
        self.sigmas = np.concatenate([sig, np.zeros(1, np.float32)])
        ts = torch.from_numpy(sig * np.float32(1000.0)).to(torch.long).float()
        half = 128
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=f32) / half)
        tfreq = torch.cat([torch.cos(ts[:, None] * freqs[None]), torch.sin(ts[:, None] * freqs[None])], dim=-1)  # [n, 256]
        te1w, te1b = sd[p + "time_caption_embed.timestep_embedder.linear_1.weight"].float().cpu(), sd[p + "time_caption_embed.timestep_embedder.linear_1.bias"].float().cpu()
        te2w, te2b = sd[p + "time_caption_embed.timestep_embedder.linear_2.weight"].float().cpu(), sd[p + "time_caption_embed.timestep_embedder.linear_2.bias"].float().cpu()
        h = tfreq @ te1w.t() + te1b
        self.time_emb = ((h * torch.sigmoid(h)) @ te2w.t() + te2b).to(dev).contiguous()  # [n, 384] input independent (host, fp32)
        mods_w, mods_b = [], []
        self.layers = []
        for i in range(self.nl):
            b = f"{p}layers.{i}"
            mods_w.append(sd[b + ".norm1.linear.weight"])
            mods_b.append(sd[b + ".norm1.linear.bias"])
            wq = torch.cat([sd[b + ".attn1.to_q.weight"], sd[b + ".attn1.to_k.weight"], sd[b + ".attn1.to_v.weight"], sd[b + ".attn2.to_q.weight"]], 0)
            wkv2 = torch.cat([sd[b + ".attn2.to_k.weight"], sd[b + ".attn2.to_v.weight"]], 0)
            self.layers.append(dict(
                n1=f(b + ".norm1.norm.weight"), wq=wq.to(device=dev, dtype=bf).contiguous(),
                q1n=(f(b + ".attn1.norm_q.weight"), f(b + ".attn1.norm_q.bias")), k1n=(f(b + ".attn1.norm_k.weight"), f(b + ".attn1.norm_k.bias")),
                q2n=(f(b + ".attn2.norm_q.weight"), f(b + ".attn2.norm_q.bias")), k2n=(f(b + ".attn2.norm_k.weight"), f(b + ".attn2.norm_k.bias")),
                ctx=f(b + ".norm1_context.weight"), wkv2=wkv2.to(device=dev, dtype=bf).contiguous(), gate=f(b + ".gate"),
                wo=w(b + ".attn2.to_out.0.weight"), n2=f(b + ".norm2.weight"), fn1=f(b + ".ffn_norm1.weight"), fn2=f(b + ".ffn_norm2.weight"),
                w13=_interleave16(sd[b + ".feed_forward.linear_1.weight"].float(), sd[b + ".feed_forward.linear_3.weight"].float()).to(device=dev, dtype=bf),
                w2=w(b + ".feed_forward.linear_2.weight"),
                kv2=torch.empty(max_envs * self.Lz, 2 * D, dtype=bf, device=dev),
                v2t=torch.empty(max_envs, self.nh, 64, 64, dtype=bf, device=dev)))
        mods_w.append(sd[p + "norm_out.linear_1.weight"])
        mods_b.append(sd[p + "norm_out.linear_1.bias"])
        self.mod_w = torch.cat(mods_w, 0).to(device=dev, dtype=bf).contiguous()    # [12*1536 + 384, 384]
        self.mod_b = torch.cat(mods_b, 0).to(device=dev, dtype=f32).contiguous()
        self.mod = torch.empty(max_envs, self.mod_w.shape[0], dtype=f32, device=dev)
        w2, b2 = sd[p + "norm_out.linear_2.weight"].float().cpu(), sd[p + "norm_out.linear_2.bias"].float().cpu()
        wd, bd = sd["action_decoder.weight"].float().cpu(), sd["action_decoder.bias"].float().cpu()   # load-time constant folding on the host
        self.head_w = (wd @ w2).to(dev).contiguous()            # [3, 384]
        self.head_b = (wd @ b2 + bd).to(dev).contiguous()       # [3]
        self.ae_w, self.ae_b = f("action_encoder.weight"), f("action_encoder.bias")
        half = D // 2
        ex = -torch.arange(half, dtype=f32) * (torch.log(torch.tensor(10000.0)) / half)
        fr = torch.arange(T, dtype=f32).unsqueeze(-1) * ex.exp()
        self.pos_tab = torch.cat([torch.sin(fr), torch.cos(fr)], dim=-1).to(dev).contiguous()  # [T, 384]
        rows = max_envs * S * T
        self.x = torch.empty(rows, D, dtype=f32, device=dev)
        self.h = torch.empty(rows, D, dtype=bf, device=dev)
        self.att = torch.empty(rows, D, dtype=bf, device=dev)
        self.qkvq = torch.empty(rows, 4 * D, dtype=bf, device=dev)
        self.qstats = torch.empty(rows, 4, 2, dtype=torch.float32, device=dev)
        self.proj = torch.empty(rows, D, dtype=bf, device=dev)   # wo / w2 outputs feed an RMSNorm, not the residual stream: bf16 halves their traffic
        self.ff = torch.empty(rows, cfg["dit_ffn"], dtype=bf, device=dev)
        self.sample = torch.empty(rows, 3, dtype=f32, device=dev)

    # ------------------------------------------------------------------------------------------------ condition
    def _memory_encoder(self, B: int):
        """MemoryEncoder (internvla_n1_arch.py:76-94): 3 post-LN nn.TransformerEncoderLayer (6 heads, ReLU, ffn 2048)."""
        ws, nm, D = self.me_ws, self.nm, self.D
        rows = B * nm
        x, h, att, qkv, ff = ws.x[:rows], ws.h[:rows], ws.att[:rows], ws.qkv[:rows], ws.ff[:rows]
        q5 = qkv.view(B, nm, 3, 6, D // 6)
        for i, Lr in enumerate(self.me_layers):
            ops.linear(h, Lr["sa_w"], bias=Lr["sa_b"], out=qkv)
            ops.attention(q5[:, :, 0], q5[:, :, 1], q5[:, :, 2], out=att.view(B, nm, 6, D // 6))
            ops.linear(att, Lr["sa_ow"], bias=Lr["sa_ob"], residual=x, out=x)
            ops.norm(x, Lr["n1"][0], Lr["n1"][1], eps=1e-5, out=h, out32=x)
            ops.linear(h, Lr["l1w"], bias=Lr["l1b"], act="relu", out=ff)
            ops.linear(ff, Lr["l2w"], bias=Lr["l2b"], residual=x, out=x)
            last = i == len(self.me_layers) - 1
            # the last LayerNorm also lands in the right half of the [feat | memory_feat] buffer the QFormer attends to
            ops.norm(x, Lr["n2"][0], Lr["n2"][1], eps=1e-5, out=self.memcat[:rows, D:] if last else h, out32=None if last else x)

    def _cond_set(self, null: bool = False) -> dict:
        """buffers of everything the DiT derives from ONE set of condition tokens z: caption projection, pooled caption embedding, per-layer
        cross-attention K|V (+ the transposed V image) and the adaLN modulation vectors. Set 0 = the real condition (buffers allocated in
        __init__); the null set (all-zero tokens, the unconditional half of classifier-free guidance, internvla_n1.py:384-385) is created
        on first use."""
        if not null:
            return dict(z=self.z, cap_h=self.cap_h, enc=self.enc, enc_n=self.enc_n, pool=self.pool, pool_n=self.pool_n, cap_emb=self.cap_emb,
                        silu_temb=self.silu_temb, mod=self.mod, kv2=[Lr["kv2"] for Lr in self.layers], v2t=[Lr["v2t"] for Lr in self.layers])
        if getattr(self, "_null", None) is None:
            e = torch.empty_like
            self._null = dict(z=torch.zeros_like(self.z), cap_h=e(self.cap_h), enc=e(self.enc), enc_n=e(self.enc_n), pool=e(self.pool),
                              pool_n=e(self.pool_n), cap_emb=e(self.cap_emb), silu_temb=e(self.silu_temb), mod=e(self.mod),
                              kv2=[e(Lr["kv2"]) for Lr in self.layers], v2t=[e(Lr["v2t"]) for Lr in self.layers])
            self.x_u = torch.empty_like(self.x)                       # second residual stream + its prediction buffers (guidance != 1)
            self.pred = [torch.empty_like(self.sample), torch.empty_like(self.sample)]
        return self._null

    def encode_images(self, B: int, images_dp: torch.Tensor):
        """the part of the condition that depends on the look-down frames only (internvla_n1.py:366-380: DINOv2, MemoryEncoder, QFormer -> the
        32 memory tokens of z). A caller that has the frames before the VLM latents can launch it early and call
        generate_traj(..., images_encoded=True) once the latents exist (bench.py --s1-early-images: measured slower inside the two-stream
        step, kept as an experiment switch)."""
        assert self.use_async
        D, L, nm, Lz = self.D, self.L, self.nm, self.Lz
        z3 = self.z[: B * Lz].view(B, Lz, L)
        # DINOv2 on the look-down frames: tokens -> left half of memcat, tokens + memory_pos -> MemoryEncoder stream
        mrows = B * nm
        self.vit.forward(images_dp.reshape(B * self.Fr, 224, 224, 3), self.vit_ws, self.memcat[:mrows, :D], mean=IMAGENET_MEAN, std=IMAGENET_STD,
                         extra_outputs=[(self.me_ws.h[:mrows], self.me_ws.x[:mrows], None, self.mem_pos)])
        self._memory_encoder(B)
        # QFormer (internvla_n1_arch.py:97-118): 32 learned queries attend to [feat | memory_feat]
        qrows = B * 32
        ops.embed3(None, None, None, out=self.q_ws.x[:qrows], pos=self.q_init, rows=qrows)
        ops.embed3(None, None, None, out=self.q_ws.h[:qrows], pos=self.q_init, rows=qrows)
        for Lr in self.q_layers:
            ops.linear(self.memcat[:mrows], Lr.ca_kvw, bias=Lr.ca_kvb, out=self.q_kv[:mrows])
            decoder_layer_postnorm(Lr, self.q_ws, B, 32, self.q_kv[:mrows], B, nm, 12, "relu")
        z3[:, :32, :].copy_(self.q_ws.h[:qrows].view(B, 32, L))  # data movement only: memory tokens into the condition buffer

    def encode_condition(self, B: int, traj_latents: torch.Tensor, images_dp: torch.Tensor, cs: dict = None, images_encoded: bool = False):
        """internvla_n1.py:364-383 -> z [B, Lz, 768] (async: 32 memory tokens | n_query projected latents; otherwise the latents alone)
        and everything of the DiT that only depends on it."""
        D, L, nq, nm, Lz = self.D, self.L, self.nq, self.nm, self.Lz
        cs = cs or self._cond_set()
        z3 = self.z[: B * Lz].view(B, Lz, L)
        rows = B * nq
        ops.linear(traj_latents.reshape(rows, -1), self.cp[0][0], bias=self.cp[0][1], act="gelu_tanh", out=self.cp_h[:rows])
        ops.linear(self.cp_h[:rows].view(B, nq, L), self.cp[1][0], bias=self.cp[1][1], out=z3[:, Lz - nq:, :], batched=True)
        if self.use_async and not images_encoded:
            self.encode_images(B, images_dp)
        self._derive_condition(B, cs)

    def _derive_condition(self, B: int, cs: dict):
        """caption_projection, pooled caption embedding (nextdit_traj.py:340-341; all-ones mask -> plain mean) and the per-layer
        cross-attention K/V (attn2.to_k/to_v on norm1_context(enc), norm_k across heads) of the condition tokens cs['z']."""
        D, Lz = self.D, self.Lz
        zr = B * Lz
        ops.linear(cs["z"][:zr], self.cap[0][0], bias=self.cap[0][1], act="gelu_tanh", out=cs["cap_h"][:zr])
        ops.linear(cs["cap_h"][:zr], self.cap[1][0], bias=self.cap[1][1], out=cs["enc"][:zr])
        ops.pool_act(cs["enc"][:zr], cs["pool"][:B], T=Lz)
        ops.norm(cs["pool"][:B], self.ce_ln[0], self.ce_ln[1], eps=1e-5, out=cs["pool_n"][:B])
        ops.linear(cs["pool_n"][:B], self.ce_w, bias=self.ce_b, out=cs["cap_emb"][:B])
        for l, Lr in enumerate(self.layers):
            ops.norm(cs["enc"][:zr], Lr["ctx"], None, eps=1e-5, rms=True, out=cs["enc_n"][:zr])
            kv = cs["kv2"][l][:zr]
            ops.linear(cs["enc_n"][:zr], Lr["wkv2"], out=kv)
            kk = kv.view(zr * 2, D)
            ops.norm(kk, Lr["k2n"][0], Lr["k2n"][1], eps=1e-5, out=kk, rows=zr, in_map=(1, 2, 0), out_map=(1, 2, 0))
            ops.dit_v2t(kv.view(B, Lz, 2, self.nh, D // self.nh), self.nh, cs["v2t"][l])   # V image the fused attention stage consumes

    # ------------------------------------------------------------------------------------------------ DiT
    def _dit_layer(self, l: int, B: int, cs: dict = None, x_buf: torch.Tensor = None):
        Lr, D, S, T, Lz, nh = self.layers[l], self.D, self.S, self.T, self.Lz, self.nh
        cs = cs or self._cond_set()
        rows, nseq, hd = B * S * T, B * S, D // self.nh
        x, h, att, qkvq, proj, ff = (self.x if x_buf is None else x_buf)[:rows], self.h[:rows], self.att[:rows], self.qkvq[:rows], self.proj[:rows], self.ff[:rows]
        mod = cs["mod"]
        m = mod[:B, l * 4 * D:(l + 1) * 4 * D]
        scale_msa, gate_msa, scale_mlp, gate_mlp = m[:, :D], m[:, D:2 * D], m[:, 2 * D:3 * D], m[:, 3 * D:]
        chain = self.row_chain and rows >= self.chain_min_rows
        chain_a, chain_b = chain and self.chain_a, chain and self.chain_b
        # whoever produces the fused q1|k1|v1|q2 projection of >= 16 k rows - the second chain launch or the row-panel GEMM - also hands the
        # attention stage the LayerNorm statistics of its rows (dit_attention then reads every row once, without a statistics pass / barrier)
        stats = self.qstats[:rows] if (chain and self.chain_stats and D == 384) else None
        if l == 0 or not chain_b:   # later blocks of the chain get their projection from the previous block's second chain launch
            if l == 0:              # (unchained: later blocks get their pre-norm from the previous block's ffn_norm2 launch)
                ops.norm(x, Lr["n1"], None, eps=1e-5, rms=True, mod_scale=scale_msa, mod_div=S * T, out=h)
            ops.linear(h, Lr["wq"], out=qkvq, seg_stats=(stats, 1e-5) if stats is not None else None)
        # LayerNorm across heads on q1 / k1 / q2 + self-attention inside each sample's T tokens + gated cross-attention against the
        # env's condition rows (shared by its S samples): one launch, the projection row is read once
        kv5 = cs["kv2"][l][: B * Lz].view(B, Lz, 2, nh, hd)
        ops.dit_attention(qkvq, att, (Lr["q1n"], Lr["k1n"], Lr["q2n"]), kv5, cs["v2t"][l], Lr["gate"], T=T, seq_per_env=S, heads=nh, eps=1e-5, stats=stats)
        last = l + 1 >= self.nl
        nxt = None if last else mod[:B, (l + 1) * 4 * D:(l + 1) * 4 * D + D]
        if chain_a:
            # attn2.to_out -> x += tanh(gate_msa) * norm2(.) -> ffn_norm1(x) * (1 + scale_mlp) -> linear_1/3 + SiLU gate
            ops.dit_rowchain(att, Lr["wo"], Lr["n2"], x, gate=gate_msa, gamma2=Lr["fn1"], mod_scale2=scale_mlp, w2=Lr["w13"], c2=ff, glu2=True,
                             mod_div=S * T, eps=1e-5)
        else:
            ops.linear(att, Lr["wo"], out=proj)
            # x += tanh(gate) * norm2(attn) and, chained in the same launch on the fresh row, h = ffn_norm1(x) * (1 + scale_mlp)
            ops.norm(proj, Lr["n2"], None, eps=1e-5, rms=True, gate=gate_msa, base=x, mod_div=S * T, out32=x,
                     out2=h, gamma2=Lr["fn1"], mod_scale2=scale_mlp)
            ops.linear(h, Lr["w13"], act="silu", glu=True, out=ff)
        if chain_b:
            # linear_2 -> x += tanh(gate_mlp) * ffn_norm2(.) -> the next block's norm1(x) * (1 + scale_msa) -> its q1|k1|v1|q2 projection
            if last:
                ops.dit_rowchain(ff, Lr["w2"], Lr["fn2"], x, gate=gate_mlp, mod_div=S * T, eps=1e-5)
            else:
                ops.dit_rowchain(ff, Lr["w2"], Lr["fn2"], x, gate=gate_mlp, gamma2=self.layers[l + 1]["n1"], mod_scale2=nxt,
                                 w2=self.layers[l + 1]["wq"], c2=qkvq, mod_div=S * T, eps=1e-5, seg_stats=stats, seg_eps=1e-5)
            return
        # linear_2, then x += tanh(gate) * ffn_norm2(ffn) and the next block's norm1(x) * (1 + scale_msa)
        ops.linear(ff, Lr["w2"], out=proj)
        if last:
            ops.norm(proj, Lr["fn2"], None, eps=1e-5, rms=True, gate=gate_mlp, base=x, mod_div=S * T, out32=x)
        else:
            ops.norm(proj, Lr["fn2"], None, eps=1e-5, rms=True, gate=gate_mlp, base=x, mod_div=S * T, out32=x,
                     out2=h, gamma2=self.layers[l + 1]["n1"], mod_scale2=nxt)

    def generate_traj(self, traj_latents: torch.Tensor, images_dp: torch.Tensor, x_init: torch.Tensor, guidance_scale: float = 1.0,
                      images_encoded: bool = False) -> torch.Tensor:
        """traj_latents bf16 [B,n_query,3584]; images_dp bf16|f32 [B,2,224,224,3] in 0..1 (ignored without 'async'); x_init f32 [B,S,T,3]
        (the initial noise the reference draws with randn_tensor)  ->  latents f32 [B,S,T,3] (x4-scaled waypoint increments).
        guidance_scale: classifier-free guidance weight of internvla_n1.py:386-387,425-427 - noise = u + g (c - u) with u the prediction
        under the all-zero condition. g == 1 returns c (the null half has zero weight and is not computed); g != 1 runs the DiT twice
        per step (conditional and null condition set) and combines the two predictions in the Euler update."""
        B = traj_latents.shape[0]
        assert B <= self.b_max and traj_latents.dtype == torch.bfloat16
        D, S, T = self.D, self.S, self.T
        rows = B * S * T
        cs = self._cond_set()
        self.encode_condition(B, traj_latents, images_dp, cs, images_encoded=images_encoded)
        cfg_on = float(guidance_scale) != 1.0
        sets = [(cs, self.x)]
        if cfg_on:
            from . import train_ops as Tr

            null = self._cond_set(null=True)
            self._derive_condition(B, null)           # input independent (z = 0): constants of the weights, recomputed per call for simplicity
            sets.append((null, self.x_u))
        sample = self.sample[:rows]
        sample.copy_(x_init.reshape(rows, 3))
        nmod = self.nl * 4 * D
        if cfg_on:       # per-step weights of the two predictions: dt * g for the conditional, dt * (1 - g) for the null one  [steps, 2, 3]
            dts = torch.from_numpy(self.sigmas[1:] - self.sigmas[:-1]).float()
            g = float(guidance_scale)
            coefs = torch.stack([dts * g, dts * (1.0 - g)], 1)[:, :, None].expand(-1, -1, 3).contiguous().to(self.device)
        for i in range(self.cfg["num_inference_steps"]):
            dt = float(self.sigmas[i + 1] - self.sigmas[i])
            for k, (c, xb) in enumerate(sets):
                ops.pool_act(c["cap_emb"][:B], c["silu_temb"][:B], T=1, pos=self.time_emb[i:i + 1], act="silu")
                ops.linear(c["silu_temb"][:B], self.mod_w, bias=self.mod_b, out=c["mod"][:B])
                ops.embed3(sample, self.ae_w, self.ae_b, out=xb[:rows], pos=self.pos_tab)
                for l in range(self.nl):
                    self._dit_layer(l, B, c, xb)
                if cfg_on:    # prediction only (mode 0); the update needs both halves
                    ops.head3(xb[:rows], self.head_w, self.head_b, None, None, eps=1e-6, mode=0, eps_out=self.pred[k][:rows],
                              mod_scale=c["mod"][:B, nmod:], mod_div=S * T)
                else:
                    ops.head3(xb[:rows], self.head_w, self.head_b, None, None, eps=1e-6, mode=2, sample=sample, coef=(dt, 0, 0, 0, 0),
                              mod_scale=c["mod"][:B, nmod:], mod_div=S * T)
            if cfg_on:
                # sample += dt * (u + g (c - u)) = dt g * c + dt (1 - g) * u   (FlowMatchEulerDiscreteScheduler.step on the guided prediction)
                for pred, wgt in ((self.pred[0], coefs[i, 0:1]), (self.pred[1], coefs[i, 1:2])):
                    Tr.affine(pred[:rows], scale=wgt, s_div=rows, out=sample, accumulate=True)
        return sample.view(B, S, T, 3)
