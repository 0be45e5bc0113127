"""Oracle of the SFT loss of the `nextdit_async` System-1 (BASELINE config #5).  TEST INFRASTRUCTURE ONLY.

Restates the `labels is not None` branch of InternVLAN1ForCausalLM.forward
  internnav/model/basemodel/internvla_n1/internvla_n1.py:222-286   (trajectory hidden states -> flow-matching MSE)
  internnav/model/basemodel/internvla_n1/internvla_n1_arch.py:189-198  get_sigmas
on the fp32 functional modules of oracle/nextdit.py / oracle/dinov2.py; torch autograd of this function is the reference gradient.
The noise and the time-step indices the reference draws inside forward (torch.randn / torch.rand, :261-264) are arguments here.
FlowMatchEulerDiscreteScheduler() (diffusers 0.33.1, un-vendored) default state: timesteps = [1000, 999, ..., 1], sigmas = t / 1000.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import dinov2
from .nextdit import RESNET_MEAN, RESNET_STD, memory_encoder, qformer, sinusoidal_positional_encoding, traj_dit
from .nn_ref import linear


def nextdit_sft_loss(sd, hidden_q, traj_images, traj_poses, video_frame_num, noise, t_index, num_train_timesteps=1000, use_async=True):
    """hidden_q [B, n_query, 3584] (hidden_states[b, t_s_pos[b] : t_s_pos[b] + n_query]); traj_images [B, T, 224, 224, 3] in 0..1;
    traj_poses [B, T, 32, 3]; video_frame_num [B]; noise [B*T, 32, 3]; t_index long [B*T]. Returns the scalar loss.
    use_async False = system1 'nextdit' (:256-258): the condition is the projected trajectory hidden states alone - no DINOv2, MemoryEncoder
    or QFormer (those modules do not exist in such a model, internvla_n1_arch.py:126-141); traj_images only supplies B and T."""
    B, Tn = traj_images.shape[:2]
    ths = hidden_q.unsqueeze(1).repeat(1, Tn, 1, 1).flatten(0, 1)                                  # :229
    loss_mask = torch.arange(Tn).expand(B, Tn) < video_frame_num.unsqueeze(1)                      # :230-232
    lat = linear(F.gelu(linear(ths.float(), sd, "cond_projector.0"), approximate="tanh"), sd, "cond_projector.2")
    if use_async:
        cur = traj_images.flatten(0, 1)
        goal = traj_images[:, 0:1].repeat(1, Tn, 1, 1, 1).flatten(0, 1)
        bsz = cur.size(0)
        images_dp = torch.stack([goal, cur], dim=1).permute(0, 1, 4, 2, 3)                         # :239
        norm = (images_dp.float() - RESNET_MEAN.view(1, 1, 3, 1, 1)) / RESNET_STD.view(1, 1, 3, 1, 1)
        feat = dinov2.forward_tokens(norm.flatten(0, 1), sd, "rgb_model.").unflatten(0, (bsz, -1)) # :242-246
        mem = memory_encoder(feat.flatten(1, 2), sd)                                               # :248-250
        tokens = qformer(torch.cat([feat.flatten(1, 2), mem], dim=-1), sd)                         # :251-252
        latents = torch.cat([tokens, lat], dim=1)                                                  # :255
    else:
        latents = lat                                                                              # :257-258
    x = traj_poses.flatten(0, 1).float()
    timesteps = (num_train_timesteps - t_index).float()
    sig = (timesteps / num_train_timesteps).view(-1, 1, 1)
    noisy = (1 - sig) * x + sig * noise                                                            # :270
    feats = linear(noisy, sd, "action_encoder") + sinusoidal_positional_encoding(x.shape[1], 384)  # :271-274
    pred = linear(traj_dit(feats, timesteps, latents, sd), sd, "action_decoder")                   # :276-281
    target = noise - x
    loss = F.mse_loss(pred.float(), target.float(), reduction="none")
    mask = loss_mask.flatten(0, 1)[:, None, None]
    return (loss * mask).sum() / mask.sum() / (loss.shape[1] * loss.shape[2])                      # :283-286


def navdp_sft_loss(sd, hidden_q, traj_images, traj_depths, traj_poses, video_frame_num, noise, timesteps, cfg):
    """navdp_async branch (internvla_n1.py:287-303) = NavDP_Policy_DPT_CriticSum_DAT.forward_vlm_traj (internvla_n1/navdp.py:291-312) with
    sample_noise's draws (:163-175) as arguments: noise [B*T, P, 3], timesteps long [B*T] in [0, num_train_timesteps).
    hidden_q [B, n_query, 3584]; traj_images [B, T, 224, 224, 3] in 0..1; traj_depths [B, T, 224, 224] metres; traj_poses [B, T, P, 3]."""
    from . import navdp as o_n
    from .nn_ref import causal_mask, layer_norm, sinusoidal_pos_emb
    from .schedulers import DDPMScheduler

    B, Tn = traj_images.shape[:2]
    ths = hidden_q.unsqueeze(1).repeat(1, Tn, 1, 1).flatten(0, 1)
    loss_mask = torch.arange(Tn).expand(B, Tn) < video_frame_num.unsqueeze(1)
    cur, cur_d = traj_images.flatten(0, 1), traj_depths.flatten(0, 1)
    goal = traj_images[:, 0:1].repeat(1, Tn, 1, 1, 1).flatten(0, 1)
    goal_d = traj_depths[:, 0:1].repeat(1, Tn, 1, 1).flatten(0, 1)
    images_dp = torch.stack([goal, cur], dim=1)
    depths_dp = torch.stack([goal_d, cur_d], dim=1).unsqueeze(-1)
    h = ths.float()
    h = F.relu(linear(h, sd, "vlm_embed_mlp.0"))
    h = F.relu(linear(h, sd, "vlm_embed_mlp.2"))
    h = linear(h, sd, "vlm_embed_mlp.4")
    vlm_embed = o_n.token_compressor(h, sd)                                                        # [N, 1, D]
    actions = traj_poses.flatten(0, 1).float()
    sch = DDPMScheduler(num_train_timesteps=cfg["num_train_timesteps"])
    noisy = sch.add_noise(actions, noise, timesteps)
    time_embeds = sinusoidal_pos_emb(timesteps, cfg["token_dim"]).unsqueeze(1)
    a = linear(noisy, sd, "input_embed")
    rgbd = o_n.dat_rgbd_backbone(images_dp, depths_dp, sd)
    cond = torch.cat([time_embeds, vlm_embed, rgbd], dim=1)
    cond = cond + sd["cond_pos_embed"][:, : cond.shape[1]]
    x = a + sd["out_pos_embed"][:, : cfg["predict_size"], :]
    x = o_n._decoder(x, cond, sd, cfg["temporal_depth"], cfg["heads"], tgt_mask=causal_mask(x.shape[1]))
    pred = linear(layer_norm(x, sd, "layernorm", 1e-5), sd, "action_head")
    loss = (pred - noise).square()
    mask = loss_mask.flatten(0, 1)[:, None, None]
    return (loss * mask).sum() / mask.sum() / (loss.shape[1] * loss.shape[2])


# ----------------------------------------------------------------------------------------------------- latent queries through the frozen LLM
class _LayerView:
    """state-dict view that presents decoder layer `i` as layer 0 (so oracle.qwen_vl.decoder_stack can run ONE layer)."""

    def __init__(self, sd, i, p="model."):
        self.sd, self.i, self.p = sd, i, p

    def __getitem__(self, k):
        pre = f"{self.p}layers.0."
        return self.sd[f"{self.p}layers.{self.i}." + k[len(pre):]] if k.startswith(pre) else self.sd[k]


class LatentQueryOracle:
    """d loss / d latent_queries through the frozen decoder at FULL depth without holding the whole autograd graph.

    The reference appends N_QUERY TRAJ tokens, overwrites their embeddings with `latent_queries` and back-propagates through all S rows
    of all layers (internvla_n1.py:166-172, 222-227; internvla_n1_trainer.py:116). With a causal mask no earlier row depends on
    `latent_queries` and no later row exists, so the gradient flows through the N_QUERY rows only: `forward` runs the prefix once
    without grad on a KV cache (oracle.qwen_vl.decoder_stack(cache=), HF use_cache semantics), then the query rows layer by layer;
    `backward` differentiates one layer at a time (torch.autograd.grad on a re-run of that layer) - the same numbers as autograd over
    the whole sequence (tests/test_oracle_golden.py checks that on the reduced configuration), at O(1 layer) memory.
    input_ids [1, L]: one unpadded sequence WITHOUT the TRAJ tokens. tap(i, x): residual stream of the query rows after layer i."""

    def __init__(self, sd, cfg, input_ids, pixel_values, grid_thw, tap=None):
        from . import qwen_vl as o_q
        from .nn_ref import rms_norm

        assert input_ids.shape[0] == 1
        self.sd, self.cfg, self.o_q, self.rms_norm = sd, cfg, o_q, rms_norm
        nq, Ln = cfg["n_query"], cfg["t_layers"]
        with torch.no_grad():
            emb = o_q.vision_tower(pixel_values, grid_thw, sd, cfg) if pixel_values is not None else None
            x = o_q.input_embeds(input_ids, emb, sd, cfg)
            pos, _ = o_q.rope_index(input_ids, grid_thw, cfg["image_token_id"], cfg["vision_start_id"])
            self.cache = [None] * Ln
            o_q.decoder_stack(x, pos, sd, cfg, cache=self.cache)
            pq = (pos[:, :, -1].max(0).values + 1)[None, :, None] + torch.arange(nq)[None, None, :]
            self.pq = pq.expand(3, 1, nq)
            self.xs = [sd["model.latent_queries"].reshape(1, nq, -1).float().clone()]
            for i in range(Ln):
                self.xs.append(self._layer(i, self.xs[-1]))
                if tap is not None:
                    tap(i, self.xs[-1])
            self.hidden = rms_norm(self.xs[-1], sd["model.norm.weight"], 1e-6)      # [1, N_QUERY, H]

    def _layer(self, i, x_in):
        got = []
        self.o_q.decoder_stack(x_in, self.pq, _LayerView(self.sd, i), dict(self.cfg, t_layers=1), tap=lambda _, y: got.append(y), cache=[self.cache[i]])
        return got[0]

    def backward(self, d_hidden):
        """d loss / d hidden [1, N_QUERY, H] -> d loss / d latent_queries f32 [N_QUERY, H]."""
        xf = self.xs[-1].detach().requires_grad_(True)
        h = self.rms_norm(xf, self.sd["model.norm.weight"], 1e-6)
        (g,) = torch.autograd.grad(h, xf, d_hidden.to(h.dtype))
        for i in reversed(range(self.cfg["t_layers"])):
            x_in = self.xs[i].detach().requires_grad_(True)
            (g,) = torch.autograd.grad(self._layer(i, x_in), x_in, g.to(x_in.dtype))
        return g[0].float()
