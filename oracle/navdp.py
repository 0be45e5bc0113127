"""Oracle restatement of the NavDP System-1 policies.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows
  internnav/model/encoder/navdp_backbone.py:205-286   RGBDBackbone                      (NavDPNet, config #2)
  internnav/model/encoder/navdp_backbone.py:102-202   DAT_RGBD_Patch_Backbone           (InternVLA-N1 navdp_async head)
  internnav/model/encoder/navdp_backbone.py:60-99     TokenCompressor
  internnav/model/basemodel/navdp/navdp_policy.py:159-185, 302-321   NavDPNet.predict_noise / predict_critic /
                                                                     predict_pointgoal_batch_action_vel
  internnav/model/basemodel/internvla_n1/navdp.py:177-253            NavDP_Policy_DPT_CriticSum_DAT.predict_noise /
                                                                     predict_pointgoal_action_async

The reference only works for one environment per call (navdp_policy.py:165 repeats the condition by 32*B;
navdp.py:228-231 truncates to the first row): the batched oracle is a loop of batch-1 reference semantics over envs
(SURVEY.md fact 2). All sampler noise is an explicit input.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import dinov2
from .nn_ref import causal_mask, decoder_layer, layer_norm, linear, mha, sinusoidal_pos_emb
from .schedulers import DDPMScheduler

IMAGENET_MEAN = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
IMAGENET_STD = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
# DAT_RGBD_Patch_Backbone creates its constants with dtype=input_dtype, "bf16" by default and never overridden by
# build_navdp (navdp_backbone.py:119-127, internvla_n1_arch.py:13): the reference therefore normalises with the
# bf16-ROUNDED mean/std (0.484375, 0.455078125, 0.40625 / 0.228515625, 0.2236328125, 0.224609375).
IMAGENET_MEAN_BF16 = IMAGENET_MEAN.to(torch.bfloat16).float()
IMAGENET_STD_BF16 = IMAGENET_STD.to(torch.bfloat16).float()


# ----------------------------------------------------------------------------------------------- RGB-D tokenisers
def _rgb_tokens(images, sd, p, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """images [B,T,H,W,3] in 0..1 -> ImageNet-normalised -> ViT-S tokens [B, T*256, 384] (navdp_backbone.py:258-266)."""
    B, T = images.shape[:2]
    x = images.float().permute(0, 1, 4, 2, 3).reshape(-1, 3, images.shape[2], images.shape[3])
    x = (x - mean) / std
    return dinov2.forward_tokens(x, sd, p + "rgb_model.").reshape(B, T * 256, -1)


def _depth_tokens(depths, sd, p):
    """depths [B,T,H,W,1] metres -> replicated to 3 channels, NOT normalised (navdp_backbone.py:274-279)."""
    B, T = depths.shape[:2]
    x = depths.float().permute(0, 1, 4, 2, 3).reshape(-1, 1, depths.shape[2], depths.shape[3])
    x = torch.cat([x, x, x], dim=1)
    return dinov2.forward_tokens(x, sd, p + "depth_model.").reshape(B, T * 256, -1)


def _former(tokens, query, sd, p):
    """former_net = nn.TransformerDecoder(nn.TransformerDecoderLayer(384, 8, batch_first=True), 2): post-LN, ReLU, ffn 2048."""
    x = query
    for i in range(2):
        x = decoder_layer(x, tokens, sd, f"{p}former_net.layers.{i}", 8, norm_first=False, act="relu")
    return linear(x, sd, p + "project_layer")


def rgbd_backbone(images, depths, sd, p="rgbd_encoder."):
    """RGBDBackbone.forward (navdp_backbone.py:248-286). images [B,M,224,224,3], depths [B,1,224,224,1] -> [B, M*16, token_dim].
    former_pe / former_query are LearnablePositionalEncoding tables indexed by arange (the query is pe(zeros) = the table)."""
    tok = torch.cat((_rgb_tokens(images, sd, p), _depth_tokens(depths, sd, p)), dim=1)
    L = tok.shape[1]
    tok = tok + sd[p + "former_pe.position_embedding.weight"][:L]
    nq = sd[p + "former_query.position_embedding.weight"].shape[0]
    q = sd[p + "former_query.position_embedding.weight"][:nq].unsqueeze(0).expand(tok.shape[0], -1, -1)
    return _former(tok, q, sd, p)


def dat_rgbd_backbone(images, depths, sd, p="rgbd_encoder."):
    """DAT_RGBD_Patch_Backbone.forward, version > 0 (navdp_backbone.py:151-202): images/depths [B,M,224,224,3|1] ->
    [B, M*16, 384]; former_pe has (2M)*256 rows, former_query M*16 rows (nn.Embedding tables indexed by arange)."""
    tok = torch.cat((_rgb_tokens(images, sd, p, IMAGENET_MEAN_BF16, IMAGENET_STD_BF16), _depth_tokens(depths, sd, p)), dim=1)
    tok = tok + sd[p + "former_pe.weight"][: tok.shape[1]]
    q = sd[p + "former_query.weight"].unsqueeze(0).expand(tok.shape[0], -1, -1)
    return _former(tok, q, sd, p)


def token_compressor(x, sd, p="goal_compressor."):
    """TokenCompressor.forward (navdp_backbone.py:79-99) with target_length queries, no padding mask."""
    B, L, _ = x.shape
    x = x + sd[p + "token_positional_encoding.position_embedding.weight"][:L]
    q = sd[p + "target_embedding.weight"]
    q = (q + sd[p + "query_positional_encoding.position_embedding.weight"][: q.shape[0]]).unsqueeze(0).expand(B, -1, -1)
    return mha(q, x, x, sd, p + "cross_attention", 8)


# ----------------------------------------------------------------------------------------------- denoiser / critic
def _decoder(tgt, mem, sd, nlayers, nhead, tgt_mask=None, memory_mask=None, p="decoder."):
    x = tgt
    for i in range(nlayers):
        x = decoder_layer(x, mem, sd, f"{p}layers.{i}", nhead, norm_first=True, act="gelu", tgt_mask=tgt_mask,
                          memory_mask=memory_mask)
    return x


def navdpnet_predict_noise(sd, last_actions, t, goal_embed, rgbd_embed, cfg):
    """NavDPNet.predict_noise (navdp_policy.py:159-170) for ONE env: last_actions [S,T,3], goal_embed [1,1,D], rgbd [1,M*16,D]."""
    D = cfg["token_dim"]
    a = linear(last_actions, sd, "input_embed")
    te = sinusoidal_pos_emb(torch.tensor([float(t)]), D).unsqueeze(1)
    cond = torch.cat([te, goal_embed, goal_embed, goal_embed, rgbd_embed], dim=1)
    cond = cond + sd["cond_pos_embed.position_embedding.weight"][: cond.shape[1]]
    cond = cond.repeat(a.shape[0], 1, 1)
    x = a + sd["out_pos_embed.position_embedding.weight"][: a.shape[1]]
    x = _decoder(x, cond, sd, cfg["temporal_depth"], cfg["heads"], tgt_mask=causal_mask(a.shape[1]))
    return linear(layer_norm(x, sd, "layernorm", 1e-5), sd, "action_head")


def navdpnet_predict_critic(sd, traj, rgbd_embed, cfg):
    """NavDPNet.predict_critic (navdp_policy.py:172-185): no causal mask, memory_mask hides the 4 time/goal slots."""
    S, T = traj.shape[:2]
    rg = rgbd_embed.repeat(S, 1, 1)
    ng = torch.zeros_like(rg[:, 0:1])
    a = linear(traj, sd, "input_embed")
    a = a + sd["out_pos_embed.position_embedding.weight"][:T]
    cond = torch.cat([ng, ng, ng, ng, rg], dim=1)
    cond = cond + sd["cond_pos_embed.position_embedding.weight"][: cond.shape[1]]
    mm = torch.zeros(T, cond.shape[1])
    mm[:, 0:4] = float("-inf")
    x = _decoder(a, cond, sd, cfg["temporal_depth"], cfg["heads"], memory_mask=mm)
    x = layer_norm(x, sd, "layernorm", 1e-5)
    return linear(x.mean(dim=1), sd, "critic_head")[:, 0]


def navdpnet_pointgoal(sd, goal_point, images, depths, x_init, step_noise, cfg, return_all=False):
    """NavDPNet.predict_pointgoal_batch_action_vel (navdp_policy.py:302-321), looped over envs; goal_point None = its zero-goal sibling
    predict_nogoal_batch_action_vel (:323-339).
    goal_point [B,3]; images [B,M,224,224,3] (0..1); depths [B,1,224,224,1]; x_init [B,S,T,3]; step_noise [K,B,S,T,3]
    (K = num_train_timesteps; the entry of the last step, t = 0, is unused). Returns negative / positive trajectories
    [B,8,T,3] (+ final samples [B,S,T,3] and critic values [B,S])."""
    B = images.shape[0]
    K = cfg["num_train_timesteps"]
    sch = DDPMScheduler(num_train_timesteps=K)
    sch.set_timesteps(K)
    rgbd = rgbd_backbone(images, depths, sd)
    if goal_point is None:       # predict_nogoal_batch_action_vel (navdp_policy.py:323-339): nogoal_embed = zeros_like(rgbd_embed[:, 0:1])
        goal = torch.zeros_like(rgbd[:, 0:1])
    else:
        goal = linear(goal_point.float(), sd, "point_encoder").unsqueeze(1)
    neg, pos, finals, critics = [], [], [], []
    for b in range(B):
        x = x_init[b].float()
        for i, t in enumerate(sch.timesteps.tolist()):
            eps = navdpnet_predict_noise(sd, x, t, goal[b:b + 1], rgbd[b:b + 1], cfg)
            x = sch.step(eps, t, x, noise=step_noise[i, b].float()).prev_sample
        c = navdpnet_predict_critic(sd, x, rgbd[b:b + 1], cfg)
        traj = torch.cumsum(x / 4.0, dim=1)
        neg.append(traj[c.argsort()[0:8]])
        pos.append(traj[(-c).argsort()[0:8]])
        finals.append(x)
        critics.append(c)
    out = (torch.stack(neg), torch.stack(pos))
    if return_all:
        out = out + (torch.stack(finals), torch.stack(critics), rgbd)
    return out


# ----------------------------------------------------------------------------------------------- InternVLA-N1 NavDP head
def n1_navdp_predict_noise(sd, last_actions, t, goal_embed, rgbd_embed, cfg):
    """NavDP_Policy_DPT_CriticSum_DAT.predict_noise (internvla_n1/navdp.py:177-195)."""
    D = cfg["token_dim"]
    a = linear(last_actions, sd, "input_embed")
    te = sinusoidal_pos_emb(torch.tensor([float(t)]), D).unsqueeze(1)
    cond = torch.cat([te, goal_embed, rgbd_embed], dim=1)
    cond = cond + sd["cond_pos_embed"][:, : cond.shape[1]]
    cond = cond.repeat(a.shape[0], 1, 1)
    x = a + sd["out_pos_embed"][:, : a.shape[1]]
    x = _decoder(x, cond, sd, cfg["temporal_depth"], cfg["heads"], tgt_mask=causal_mask(a.shape[1]))
    return linear(layer_norm(x, sd, "layernorm", 1e-5), sd, "action_head")


def n1_navdp_async(sd, vlm_tokens, images, depths, x_init, step_noise, cfg, return_all=False):
    """predict_pointgoal_action_async (internvla_n1/navdp.py:197-253), looped over envs.
    vlm_tokens [B,n_query,3584]; images [B,2,224,224,3]; depths [B,2,224,224,1]; x_init [B,S,T,3];
    step_noise [K,B,S,T,3] -> samples [B,S,T,3]."""
    B = vlm_tokens.shape[0]
    K = cfg["num_train_timesteps"]
    sch = DDPMScheduler(num_train_timesteps=K)
    sch.set_timesteps(K)
    h = vlm_tokens.float()
    h = F.relu(linear(h, sd, "vlm_embed_mlp.0"))
    h = F.relu(linear(h, sd, "vlm_embed_mlp.2"))
    h = linear(h, sd, "vlm_embed_mlp.4")
    goal = token_compressor(h, sd)
    rgbd = dat_rgbd_backbone(images, depths, sd)
    outs = []
    for b in range(B):
        x = x_init[b].float()
        for i, t in enumerate(sch.timesteps.tolist()):
            eps = n1_navdp_predict_noise(sd, x, t, goal[b:b + 1], rgbd[b:b + 1], cfg)
            x = sch.step(eps, t, x, noise=step_noise[i, b].float()).prev_sample
        outs.append(x)
    out = torch.stack(outs)
    return (out, goal, rgbd) if return_all else out


def n1_navdp_plain(sd, vlm_tokens, x_init, step_noise, cfg):
    """predict_pointgoal_action (internvla_n1/navdp.py:255-289), the non-async 'navdp' System-1: vlm_embed = mean over the tokens of
    vlm_embed_mlp(vlm_tokens), condition [time, vlm_embed] + cond_pos_embed[:, :2] (predict_noise with rgbd_embed=None, :186-187).
    Looped over envs (the reference keeps only sample 0 of a batch, :265-268). vlm_tokens [B,n_query,3584] -> samples [B,S,T,3]."""
    B = vlm_tokens.shape[0]
    K = cfg["num_train_timesteps"]
    sch = DDPMScheduler(num_train_timesteps=K)
    sch.set_timesteps(K)
    h = vlm_tokens.float()
    h = F.relu(linear(h, sd, "vlm_embed_mlp.0"))
    h = F.relu(linear(h, sd, "vlm_embed_mlp.2"))
    goal = linear(h, sd, "vlm_embed_mlp.4").mean(dim=1, keepdim=True)
    empty = goal[:, :0]
    outs = []
    for b in range(B):
        x = x_init[b].float()
        for i, t in enumerate(sch.timesteps.tolist()):
            eps = n1_navdp_predict_noise(sd, x, t, goal[b:b + 1], empty[b:b + 1], cfg)
            x = sch.step(eps, t, x, noise=step_noise[i, b].float()).prev_sample
        outs.append(x)
    return torch.stack(outs)
