"""Generate tests/golden/*.pt by executing the REAL reference modules (from /root/reference) on CPU.  TEST INFRASTRUCTURE.

Run in the build container only:  python -m oracle.make_golden [--only NAME]
For each unit the reference module is constructed under the stubs of `oracle/ref_loader.py`, loaded with the seeded
state-dict of `oracle/weights.py` (strict key/shape check of our tables against the reference constructor), run on the
seeded inputs, and its outputs are saved in fp32. The oracle restatement is evaluated on the same inputs and the
max-abs difference is printed and stored next to the outputs (`oracle_max_abs_diff`), so the fixture both pins the
oracle and documents how tightly. `tests/test_oracle_golden.py` re-checks oracle-vs-fixture on every CPU test run.
"""
from __future__ import annotations

import argparse
from pathlib import Path

import numpy as np
import torch

from . import dinov2 as o_dino
from . import navdp as o_navdp
from . import ref_loader as R
from . import weights as W

GOLD = Path(__file__).resolve().parent.parent / "tests" / "golden"


def _load_strict(module, sd, allow_missing_prefixes=(), skip_buffers=()):
    ref_sd = module.state_dict()
    for k, v in sd.items():
        assert k in ref_sd, f"our key {k} does not exist in the reference module"
        assert tuple(ref_sd[k].shape) == tuple(v.shape), f"{k}: ours {tuple(v.shape)} vs reference {tuple(ref_sd[k].shape)}"
    missing = [k for k in ref_sd if k not in sd and not k.startswith(tuple(allow_missing_prefixes)) and k not in skip_buffers]
    assert not missing, f"reference parameters without a counterpart in oracle/weights.py: {missing[:8]}"
    module.load_state_dict({k: v.to(ref_sd[k].dtype) for k, v in sd.items()}, strict=False)
    return module.float().eval()


def gold_dinov2():
    sd = W.materialize(W.dinov2_vits_spec(), seed=3)
    vit = _load_strict(R.dinov2_vits(), sd)
    g = torch.Generator().manual_seed(7)
    img = torch.randn(2, 3, 224, 224, generator=g)
    with torch.no_grad():
        ref = vit.get_intermediate_layers(img)[0]
        mine = o_dino.forward_tokens(img, sd)
    return dict(seed=3, img_seed=7, tokens=ref, oracle_max_abs_diff=(ref - mine).abs().max().item())


class _Inject:
    """Feed explicit noise into the reference's sampler loop: the initial torch.randn and every scheduler.step."""

    def __init__(self, mod, sched, x_init, step_noise):
        self.queue = [n for n in step_noise]
        self.x_init = x_init
        orig = sched.step

        def step(model_output, timestep, sample, **kw):
            return orig(model_output, timestep, sample, noise=self.queue.pop(0))

        sched.step = step
        self.mod = mod


def gold_navdpnet(B=2):
    torch_load = torch.load
    torch.load = lambda *a, **k: {}
    try:
        npm = R.navdp_policy_module()
        cfg = W.NAVDPNET_CFG
        il = dict(image_size=224, memory_size=cfg["memory_size"], predict_size=cfg["predict_size"], pixel_channel=4,
                  temporal_depth=cfg["temporal_depth"], heads=cfg["heads"], channels=3, dropout=0.1,
                  token_dim=cfg["token_dim"], scratch=False, finetune=False)
        net = npm.NavDPNet(npm.NavDPModelConfig(model_cfg={"model": {}, "local_rank": 0, "il": il}))
    finally:
        torch.load = torch_load
    sd = W.navdpnet_state_dict(seed=0)
    net = _load_strict(net, sd, allow_missing_prefixes=("pixel_encoder.", "image_encoder.", "pixel_aux_head.", "image_aux_head."))
    net._device = torch.device("cpu")
    net.cond_critic_mask = net.cond_critic_mask.float()
    inp = W.navdpnet_inputs(B, seed=0)
    negs, poss, rgbds = [], [], []
    with torch.no_grad():
        for b in range(B):
            _Inject(npm, net.noise_scheduler, inp["x_init"][b], inp["step_noise"][:, b])
            # the reference draws the initial noise with torch.randn (navdp_policy.py:308): patch it for this call
            real_randn = torch.randn
            torch.randn = lambda *a, **k: inp["x_init"][b].clone()
            try:
                neg, pos = net.predict_pointgoal_batch_action_vel(inp["goal"][b:b + 1].numpy(), inp["images"][b:b + 1],
                                                                  inp["depths"][b:b + 1])
            finally:
                torch.randn = real_randn
            negs.append(neg)
            poss.append(pos)
            rgbds.append(net.rgbd_encoder(inp["images"][b:b + 1], inp["depths"][b:b + 1]))
            net.noise_scheduler.step = net.noise_scheduler.__class__.step.__get__(net.noise_scheduler)
        neg, pos, rgbd = torch.stack(negs), torch.stack(poss), torch.cat(rgbds)
        o_neg, o_pos, o_fin, o_cr, o_rgbd = o_navdp.navdpnet_pointgoal(sd, inp["goal"], inp["images"], inp["depths"],
                                                                     inp["x_init"], inp["step_noise"], cfg, return_all=True)
    d = max((neg - o_neg).abs().max().item(), (pos - o_pos).abs().max().item(), (rgbd - o_rgbd).abs().max().item())
    # the reference returns only the ranked trajectories; the final samples and critic values (continuous quantities the
    # ranking is a discontinuous function of) are stored from the oracle, which the lines above pin to the reference.
    return dict(B=B, seed=0, negative=neg, positive=pos, rgbd_embed=rgbd, oracle_final=o_fin, oracle_critic=o_cr,
                oracle_max_abs_diff=d)


def gold_navdpnet_nogoal(B=2):
    """the reference's own `NavDPNet.predict_nogoal_batch_action_vel` (navdp_policy.py:323-339), one env per call as it executes, with the
    same weights / frames / injected noise as the point-goal fixture (its own file: navdpnet.pt stays byte-identical)."""
    torch_load = torch.load
    torch.load = lambda *a, **k: {}
    try:
        npm = R.navdp_policy_module()
        cfg = W.NAVDPNET_CFG
        il = dict(image_size=224, memory_size=cfg["memory_size"], predict_size=cfg["predict_size"], pixel_channel=4,
                  temporal_depth=cfg["temporal_depth"], heads=cfg["heads"], channels=3, dropout=0.1,
                  token_dim=cfg["token_dim"], scratch=False, finetune=False)
        net = npm.NavDPNet(npm.NavDPModelConfig(model_cfg={"model": {}, "local_rank": 0, "il": il}))
    finally:
        torch.load = torch_load
    sd = W.navdpnet_state_dict(seed=0)
    net = _load_strict(net, sd, allow_missing_prefixes=("pixel_encoder.", "image_encoder.", "pixel_aux_head.", "image_aux_head."))
    net._device = torch.device("cpu")
    net.cond_critic_mask = net.cond_critic_mask.float()
    inp = W.navdpnet_inputs(B, seed=0)
    negs, poss = [], []
    with torch.no_grad():
        for b in range(B):
            _Inject(npm, net.noise_scheduler, inp["x_init"][b], inp["step_noise"][:, b])
            real_randn = torch.randn
            torch.randn = lambda *a, **k: inp["x_init"][b].clone()       # the initial noise of navdp_policy.py:328
            try:
                neg, pos = net.predict_nogoal_batch_action_vel(inp["images"][b:b + 1], inp["depths"][b:b + 1])
            finally:
                torch.randn = real_randn
            negs.append(neg)
            poss.append(pos)
            net.noise_scheduler.step = net.noise_scheduler.__class__.step.__get__(net.noise_scheduler)
        neg, pos = torch.stack(negs), torch.stack(poss)
        o_neg, o_pos, o_fin, o_cr, _ = o_navdp.navdpnet_pointgoal(sd, None, inp["images"], inp["depths"], inp["x_init"], inp["step_noise"], cfg,
                                                                  return_all=True)
    d = max((neg - o_neg).abs().max().item(), (pos - o_pos).abs().max().item())
    return dict(B=B, seed=0, negative=neg, positive=pos, oracle_final=o_fin, oracle_critic=o_cr, oracle_max_abs_diff=d)


def gold_n1_navdp(B=2):
    torch_load = torch.load
    torch.load = lambda *a, **k: {}
    try:
        n1 = R.n1_navdp_module()
        cfg = W.N1_NAVDP_CFG
        m = n1.NavDP_Policy_DPT_CriticSum_DAT(memory_size=cfg["memory_size"], navdp_version=0.1, input_dtype="fp32")
    finally:
        torch.load = torch_load
    sd = W.n1_navdp_state_dict(seed=1)
    m = _load_strict(m, sd, allow_missing_prefixes=("point_encoder.", "critic_head.", "pg_embed_mlp.", "pg_pred_mlp.",
                                                    "decoder_layer."),
                     skip_buffers=("goal_compressor.positional_encoding.pe",))
    # the RGB-D backbone keeps its default input_dtype="bf16" constants (bf16-rounded ImageNet mean/std), as in the
    # reference N1 build (internvla_n1_arch.py:13); only the arithmetic is lifted to fp32 for the fixture
    m.rgbd_encoder.input_dtype = torch.float32
    m.rgbd_encoder.preprocess_mean = m.rgbd_encoder.preprocess_mean.float()
    m.rgbd_encoder.preprocess_std = m.rgbd_encoder.preprocess_std.float()
    m.tgt_mask = m.tgt_mask.float()
    inp = W.n1_navdp_inputs(B, seed=1)
    outs = []
    with torch.no_grad():
        for b in range(B):
            _Inject(n1, m.noise_scheduler, inp["x_init"][b], inp["step_noise"][:, b])
            real_randn = torch.randn
            # the reference draws the initial noise with torch.randn (navdp.py:242): patch it for the duration of the call
            torch.randn = lambda *a, **k: inp["x_init"][b].clone()
            try:
                outs.append(m.predict_pointgoal_action_async(inp["vlm_tokens"][b:b + 1], inp["images"][b:b + 1], inp["depths"][b:b + 1]))
            finally:
                torch.randn = real_randn
            m.noise_scheduler.step = m.noise_scheduler.__class__.step.__get__(m.noise_scheduler)
        ref = torch.stack(outs)
        mine = o_navdp.n1_navdp_async(sd, inp["vlm_tokens"], inp["images"], inp["depths"], inp["x_init"], inp["step_noise"], cfg)
        # the non-async 'navdp' System-1 type: predict_pointgoal_action(vlm_tokens) (internvla_n1/navdp.py:255-289), same weights
        plain = []
        for b in range(B):
            _Inject(n1, m.noise_scheduler, inp["x_init"][b], inp["step_noise"][:, b])
            real_randn = torch.randn
            torch.randn = lambda *a, **k: inp["x_init"][b].clone()
            try:
                plain.append(m.predict_pointgoal_action(inp["vlm_tokens"][b:b + 1], vlm_mask=None))
            finally:
                torch.randn = real_randn
            m.noise_scheduler.step = m.noise_scheduler.__class__.step.__get__(m.noise_scheduler)
        plain = torch.stack(plain)
        mine_plain = o_navdp.n1_navdp_plain(sd, inp["vlm_tokens"], inp["x_init"], inp["step_noise"], cfg)
    d = max((ref - mine).abs().max().item(), (plain - mine_plain).abs().max().item())
    return dict(B=B, seed=1, trajectories=ref, trajectories_plain=plain, oracle_max_abs_diff=d)


def _ffn_variant(variant: str):
    """(engine / weight config, context manager selecting the matching LuminaFeedForward convention of the fake diffusers package) for one of
    the two FFN widths the reference's in-tree block can have: 'ffn1536' = diffusers 0.33.1 as pinned, 'ffn1024' = diffusers <= 0.32."""
    from . import diffusers_blocks as blk

    cfg = W.N1_NEXTDIT_VARIANTS[variant]
    return cfg, blk.ffn_convention(legacy_two_thirds=(variant == "ffn1024"))


def _check_ffn_width(m, cfg):
    """the reference's own constructor must have produced the width the weight table assumes"""
    got = m.traj_dit.model.layers[0].feed_forward.linear_1.weight.shape[0]
    assert got == cfg["dit_ffn"], (got, cfg["dit_ffn"])


def gold_n1_nextdit(B=2, variant="ffn1536"):
    """The reference's System-1 component modules (NextDiTCrossAttn incl. the in-tree LuminaNextDiTBlock wiring, MemoryEncoder,
    QFormer, DINOv2) driven by a line-by-line transcription of generate_traj's nextdit_async branch (internvla_n1.py:359-432):
    `InternVLAN1ForCausalLM` itself cannot be constructed here (it subclasses the transformers-4.51 Qwen2.5-VL layout)."""
    import numpy as np
    import torch.nn as nn

    from . import nextdit as o_nd

    nd, arch = R.nextdit_module(), R.n1_arch_module()
    cfg, convention = _ffn_variant(variant)
    sd = W.n1_nextdit_state_dict(seed=4, cfg=cfg)

    class S1(nn.Module):  # attribute names of InternVLAN1MetaModel.__init__ (internvla_n1_arch.py:127-141)
        def __init__(self):
            super().__init__()
            self.traj_dit = nd.NextDiTCrossAttn(nd.NextDiTCrossAttnConfig(latent_embedding_size=768))
            self.action_encoder = nn.Linear(3, 384, bias=True)
            self.pos_encoding = arch.SinusoidalPositionalEncoding(384)
            self.action_decoder = nn.Linear(384, 3, bias=True)
            self.cond_projector = nn.Sequential(nn.Linear(3584, 768), nn.GELU(approximate="tanh"), nn.Linear(768, 768))
            self.rgb_model = R.dinov2_vits()
            self.memory_encoder = arch.MemoryEncoder()
            self.rgb_resampler = arch.QFormer()

    with convention:
        m = S1()
    _check_ffn_width(m, cfg)
    m = _load_strict(m, sd, allow_missing_prefixes=("traj_dit.model.patch_embedder.", "rgb_resampler.visual_proj."))
    inp = W.n1_nextdit_inputs(B, seed=4)
    from .schedulers import FlowMatchEulerDiscreteScheduler

    mean = torch.FloatTensor([0.485, 0.456, 0.406]).view(1, 1, 3, 1, 1)
    std = torch.FloatTensor([0.229, 0.224, 0.225]).view(1, 1, 3, 1, 1)
    def run_reference(guidance_scale, use_async):
        """generate_traj (internvla_n1.py:359-432) transcribed line by line on the reference modules, one env at a time."""
        outs = []
        for b in range(B):
            scheduler = FlowMatchEulerDiscreteScheduler()
            traj_latents = m.cond_projector(inp["traj_latents"][b:b + 1])
            if use_async:
                images_dp = inp["images"][b:b + 1].permute(0, 1, 4, 2, 3)
                images_dp_norm = (images_dp - mean) / std
                feat = m.rgb_model.get_intermediate_layers(images_dp_norm.flatten(0, 1))[0].unflatten(dim=0, sizes=(1, -1))
                memory_feat = m.memory_encoder(feat.flatten(1, 2))
                memory_feat = torch.cat([feat.flatten(1, 2), memory_feat], dim=-1)
                memory_tokens = m.rgb_resampler(memory_feat)
                hidden_states = torch.cat([memory_tokens, traj_latents], dim=1)
            else:
                hidden_states = traj_latents
            hidden_states_input = torch.cat([torch.zeros_like(hidden_states), hidden_states], 0)
            latents = inp["x_init"][b].clone()
            sigmas = np.linspace(1.0, 1 / 10, 10)
            scheduler.set_timesteps(10, sigmas=sigmas)
            hidden_states_input = hidden_states_input.repeat_interleave(32, dim=0)
            for t in scheduler.timesteps:
                latent_features = m.action_encoder(latents)
                pos_ids = torch.arange(latent_features.shape[1]).reshape(1, -1).repeat(1, 1)
                latent_features += m.pos_encoding(pos_ids)
                latent_model_input = latent_features.repeat(2, 1, 1)
                noise_pred = m.traj_dit(x=latent_model_input, timestep=t.unsqueeze(0).expand(latent_model_input.shape[0]).to(torch.long),
                                        z_latents=hidden_states_input)
                noise_pred = m.action_decoder(noise_pred)
                noise_pred_uncond, noise_pred = noise_pred.chunk(2)
                noise_pred = noise_pred_uncond + guidance_scale * (noise_pred - noise_pred_uncond)
                latents = scheduler.step(noise_pred, t, latents).prev_sample
            outs.append(latents)
        return torch.stack(outs)

    with torch.no_grad():
        # the other branches of generate_traj: classifier-free guidance with a weight != 1 (:386-387,425-427) and the plain 'nextdit'
        # System-1 type whose condition is the projected latents alone (:382-383)
        variants = {}
        for name, (gs, asy) in (("cfg_2p5", (2.5, True)), ("plain", (1.0, False)), ("plain_cfg_0p5", (0.5, False))):
            r = run_reference(gs, asy)
            o = o_nd.generate_traj(sd, inp["traj_latents"], inp["images"], inp["x_init"], guidance_scale=gs, use_async=asy)
            variants[name] = dict(guidance_scale=gs, use_async=asy, latents=r, oracle_max_abs_diff=(r - o).abs().max().item())
        outs = list(run_reference(1.0, True))
        ref = torch.stack(outs)
        mine = o_nd.generate_traj(sd, inp["traj_latents"], inp["images"], inp["x_init"])
    worst = max([(ref - mine).abs().max().item()] + [v["oracle_max_abs_diff"] for v in variants.values()])
    return dict(B=B, seed=4, dit_ffn=cfg["dit_ffn"], latents=ref, variants=variants, oracle_max_abs_diff=worst)


def gold_qwen(B=2, n_img=2):
    """System-2: the installed transformers Qwen2.5-VL vision tower and text model (the reference's un-vendored arithmetic,
    transformers==4.51.0 pinned / 5.x installed), the reference's vendored get_rope_index_25, and the reference's glue
    (internvla_n1.py:128-220, 320-347) transcribed onto them: logits of a prefill, greedy tokens, generate_latents."""
    import importlib.util

    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLTextConfig, Qwen2_5_VLVisionConfig
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VisionTransformerPretrainedModel, Qwen2_5_VLTextModel

    from . import qwen_vl as o_q

    cfg = W.QWEN_TEST_CFG
    sd = W.qwen_state_dict(seed=6, cfg=cfg)
    vc = Qwen2_5_VLVisionConfig(depth=cfg["v_depth"], hidden_size=cfg["v_hidden"], intermediate_size=cfg["v_inter"], num_heads=cfg["v_heads"],
                                out_hidden_size=cfg["v_out"], fullatt_block_indexes=list(cfg["v_fullatt"]), window_size=cfg["v_window"])
    vc._attn_implementation = "eager"
    vit = Qwen2_5_VisionTransformerPretrainedModel(vc).float().eval()
    vit.load_state_dict({k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}, strict=True)
    tc = Qwen2_5_VLTextConfig(vocab_size=cfg["vocab"], hidden_size=cfg["t_hidden"], intermediate_size=cfg["t_inter"],
                              num_hidden_layers=cfg["t_layers"], num_attention_heads=cfg["t_heads"], num_key_value_heads=cfg["t_kv_heads"],
                              rms_norm_eps=1e-6, rope_parameters={"rope_type": "default", "rope_theta": cfg["rope_theta"], "mrope_section": [16, 24, 24]},
                              max_position_embeddings=32768, pad_token_id=0)
    tc._attn_implementation = "eager"
    llm = Qwen2_5_VLTextModel(tc).float().eval()
    llm.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.") and k != "model.latent_queries"}, strict=True)
    spec = importlib.util.spec_from_file_location("ref_rope2d", str(R.REF / "internnav" / "dataset" / "rope2d.py"))
    rope2d = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rope2d)
    # input seed 9: every greedy step of both sequences has a top-2 margin >= 0.13 logits (seed 6 had a 0.004 near-tie on the first token of
    # sequence 1, which made exact-token tests depend on fp contraction choices of unrelated kernels)
    inp = W.qwen_inputs(B, n_img, seed=9, cfg=cfg)
    ids, pv, grid = inp["input_ids"], inp["pixel_values"], inp["grid_thw"]

    def ref_rope(i):  # the reference's vendored implementation with the test config's token ids patched in
        src = (R.REF / "internnav" / "dataset" / "rope2d.py").read_text()
        src = src.replace("image_token_id = 151655", f"image_token_id = {cfg['image_token_id']}").replace(
            "vision_start_token_id = 151652", f"vision_start_token_id = {cfg['vision_start_id']}")
        ns = {}
        exec(compile(src, "rope2d_patched", "exec"), ns)
        return ns["get_rope_index_25"](2, i, grid)

    def ref_forward(i):
        with torch.no_grad():
            emb = vit(pv, grid_thw=grid).pooler_output
            x = llm.embed_tokens(i)
            x = x.masked_scatter((i == cfg["image_token_id"]).unsqueeze(-1).expand_as(x), emb)
            traj = i == cfg["traj_token_id"]
            if traj.any():
                x[traj] = sd["model.latent_queries"].repeat(i.shape[0], 1, 1).view(-1, x.shape[-1])
            pos, _ = ref_rope(i)
            h = llm(inputs_embeds=x, position_ids=pos, use_cache=False).last_hidden_state
            return torch.nn.functional.linear(h, sd["lm_head.weight"]), h, emb, pos

    logits, h, emb, pos = ref_forward(ids)
    gen = ids.clone()
    margins = []
    for _ in range(3):
        lg, _, _, _ = ref_forward(gen)
        t2 = lg[:, -1].topk(2, dim=-1).values
        margins.append(t2[:, 0] - t2[:, 1])
        gen = torch.cat([gen, lg[:, -1].argmax(-1)[:, None]], dim=1)
    assert float(torch.stack(margins).min()) > 0.05, "pick an input seed without greedy near-ties"
    ids_q = torch.cat([gen, torch.full((B, cfg["n_query"]), cfg["traj_token_id"], dtype=torch.long)], dim=1)
    _, hq, _, _ = ref_forward(ids_q)
    latents = hq[:, -cfg["n_query"]:].clone()
    with torch.no_grad():
        o_emb = o_q.vision_tower(pv, grid, sd, cfg)
        o_pos, _ = o_q.rope_index(ids, grid, cfg["image_token_id"], cfg["vision_start_id"])
        o_logits, _ = o_q.forward_logits(sd, cfg, ids, pv, grid)
        o_gen = o_q.generate(sd, cfg, ids, pv, grid, 3)
        o_lat = o_q.generate_latents(sd, cfg, o_gen, pv, grid)
    assert torch.equal(o_pos, pos), "rope_index differs from the reference's get_rope_index_25"
    assert torch.equal(o_gen, gen), "greedy tokens differ"
    d = max((o_emb - emb).abs().max().item(), (o_logits - logits).abs().max().item(), (o_lat - latents).abs().max().item())
    # fixtures stay small: every 8th row of the image embeds, last-position logits only
    return dict(B=B, n_img=n_img, seed=6, input_seed=9, image_embeds=emb[::8].clone(), embed_row_stride=8, position_ids=pos.to(torch.int32),
                last_logits=logits[:, -1].clone(), generated=gen, latents=latents, margins=torch.stack(margins, 1), oracle_max_abs_diff=d)


def gold_unet1d(B=2):
    """vendored diffusion-policy ConditionalUnet1D (conditional_unet1d.py:69-241) executed as-is: one noise prediction and a 10-step DDIM
    sampling loop (scheduler = the restated diffusers DDIMScheduler, the package being absent) on seeded weights / inputs."""
    import importlib

    from . import unet1d as o_u
    from .schedulers import DDIMScheduler

    R.setup()
    mod = importlib.import_module("diffusion_policy.model.diffusion.conditional_unet1d")
    cfg = W.UNET1D_CFG
    net = mod.ConditionalUnet1D(input_dim=cfg["input_dim"], global_cond_dim=cfg["global_cond_dim"], diffusion_step_embed_dim=cfg["dsed"],
                                down_dims=list(cfg["down_dims"]), kernel_size=cfg["kernel_size"], n_groups=cfg["n_groups"], cond_predict_scale=True)
    sd = W.materialize(W.unet1d_spec(cfg), seed=4)
    net = _load_strict(net, sd)
    inp = W.unet1d_inputs(B, seed=4, cfg=cfg)
    S, T, D = cfg["sample_num"], cfg["predict_size"], cfg["input_dim"]
    g = inp["global_cond"].repeat_interleave(S, dim=0)
    x0 = inp["x_init"].reshape(B * S, T, D).clone()
    sch = DDIMScheduler(num_train_timesteps=cfg["num_train_timesteps"])
    sch.set_timesteps(cfg["num_inference_steps"])
    out, d, n_clipped = {}, 0.0, 0
    with torch.no_grad():
        eps0 = net(x0, int(sch.timesteps[0]), global_cond=g)
        # the loop of diffusion_unet_lowdim_policy.py:77-91 with its **kwargs empty (the shipped behaviour: use_clipped_model_output absent
        # = False) and, as a second fixture, with use_clipped_model_output=True
        for flag in (False, True):
            x = x0.clone()
            for t in sch.timesteps.tolist():
                r = sch.step(net(x, t, global_cond=g), t, x, **({"use_clipped_model_output": True} if flag else {}))
                n_clipped += int((r.pred_original_sample.abs() >= 1.0).sum()) if not flag else 0
                x = r.prev_sample
            out[flag] = x.reshape(B, S, T, D).clone()
            mine = o_u.ddim_sample(sd, inp["global_cond"], inp["x_init"], cfg["num_train_timesteps"], cfg["num_inference_steps"], use_clipped_model_output=flag)
            d = max(d, (out[flag] - mine).abs().max().item())
        mine_eps = o_u.unet_forward(sd, inp["x_init"].reshape(B * S, T, D), int(sch.timesteps[0]), g)
    d = max(d, (eps0 - mine_eps).abs().max().item())
    assert n_clipped > 0 and (out[False] - out[True]).abs().max().item() > 1e-3, "the fixture must exercise the clip (the two step variants differ only there)"
    # yardstick: the same loops under bf16 autocast (the precision the vendored policy runs in). With the default step the network's own
    # eps stays in the update even where x0 was clipped, so the 10-step recursion is far more sensitive to rounding than the re-derived
    # variant (which contracts onto the clip bounds): bf16 PyTorch is ~1e-2 from fp32 there, ~2e-3 with use_clipped_model_output=True
    yard = {}
    for flag in (False, True):
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            o16 = o_u.ddim_sample(sd, inp["global_cond"], inp["x_init"], cfg["num_train_timesteps"], cfg["num_inference_steps"], use_clipped_model_output=flag).float()
        e = (o16 - out[flag]).abs()
        yard[flag] = dict(mean=e.mean().item(), max=e.max().item())
    return dict(seed=4, B=B, eps0=eps0.reshape(B, S, T, D).clone(), samples=out[False], samples_use_clipped_model_output=out[True],
                bf16_autocast_err=yard[False], bf16_autocast_err_use_clipped_model_output=yard[True],
                clipped_x0_elements=n_clipped, timesteps=sch.timesteps.clone(), oracle_max_abs_diff=d)


def gold_vln_utils():
    """The reference's own host post-processing (internnav/model/utils/vln_utils.py) on seeded trajectories: the integer action
    lists are the known answers for internnav_amd.policy.traj_to_actions / chunk_token / split_and_clean (bit-exact)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_vln_utils", str(R.REF / "internnav" / "model" / "utils" / "vln_utils.py"))
    vu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vu)
    g = torch.Generator().manual_seed(11)
    cases = []
    for i in range(12):
        base = torch.tensor([[1.2 - 0.15 * i, 0.35 * ((i % 5) - 2), 0.1 * ((i % 3) - 1)]]) * (1 + 0.2 * (i % 4))
        traj = (base.view(1, 1, 3) * torch.linspace(1.0, 0.3, 32).view(1, 32, 1) + 0.05 * torch.randn(32, 32, 3, generator=g)).contiguous()
        inp = traj.clone()
        acts = vu.traj_to_actions(inp)
        chunks = vu.chunk_token(traj[0] / 4.0)
        cases.append(dict(traj=traj, actions=[int(a) for a in acts], mutated=inp, chunk=[int(a) for a in chunks]))
    texts = ["go to <image>\n the door. you can see <image>.", "<image><image> a \n b", "no image here"]
    return dict(cases=cases, texts=texts, split=[vu.split_and_clean(t) for t in texts], oracle_max_abs_diff=0.0)


def gold_qwen_lookdown():
    """S2 with the un-resized look-down frame the reference feeds (476x644 -> 34x46 patches, ragged 112-pixel windows: 17x23 merged
    cells = 9 full + 6 half + ... windows) next to a 28x28 frame, B = 1: transformers vision tower + text model, same glue as gold_qwen."""
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLTextConfig, Qwen2_5_VLVisionConfig
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VisionTransformerPretrainedModel, Qwen2_5_VLTextModel

    from . import qwen_vl as o_q

    cfg = W.QWEN_TEST_CFG
    sd = W.qwen_state_dict(seed=6, cfg=cfg)
    vc = Qwen2_5_VLVisionConfig(depth=cfg["v_depth"], hidden_size=cfg["v_hidden"], intermediate_size=cfg["v_inter"], num_heads=cfg["v_heads"],
                                out_hidden_size=cfg["v_out"], fullatt_block_indexes=list(cfg["v_fullatt"]), window_size=cfg["v_window"])
    vc._attn_implementation = "eager"
    vit = Qwen2_5_VisionTransformerPretrainedModel(vc).float().eval()
    vit.load_state_dict({k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}, strict=True)
    tc = Qwen2_5_VLTextConfig(vocab_size=cfg["vocab"], hidden_size=cfg["t_hidden"], intermediate_size=cfg["t_inter"],
                              num_hidden_layers=cfg["t_layers"], num_attention_heads=cfg["t_heads"], num_key_value_heads=cfg["t_kv_heads"],
                              rms_norm_eps=1e-6, rope_parameters={"rope_type": "default", "rope_theta": cfg["rope_theta"], "mrope_section": [16, 24, 24]},
                              max_position_embeddings=32768, pad_token_id=0)
    tc._attn_implementation = "eager"
    llm = Qwen2_5_VLTextModel(tc).float().eval()
    llm.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.") and k != "model.latent_queries"}, strict=True)
    inp = lookdown_inputs(cfg)
    ids, pv, grid = inp["input_ids"], inp["pixel_values"], inp["grid_thw"]
    src = (R.REF / "internnav" / "dataset" / "rope2d.py").read_text()
    src = src.replace("image_token_id = 151655", f"image_token_id = {cfg['image_token_id']}").replace(
        "vision_start_token_id = 151652", f"vision_start_token_id = {cfg['vision_start_id']}")
    ns = {}
    exec(compile(src, "rope2d_patched", "exec"), ns)
    with torch.no_grad():
        emb = vit(pv, grid_thw=grid).pooler_output
        x = llm.embed_tokens(ids)
        x = x.masked_scatter((ids == cfg["image_token_id"]).unsqueeze(-1).expand_as(x), emb)
        pos, _ = ns["get_rope_index_25"](2, ids, grid)
        h = llm(inputs_embeds=x, position_ids=pos, use_cache=False).last_hidden_state
        logits = torch.nn.functional.linear(h[:, -1], sd["lm_head.weight"])
        o_emb = o_q.vision_tower(pv, grid, sd, cfg)
        o_pos, _ = o_q.rope_index(ids, grid, cfg["image_token_id"], cfg["vision_start_id"])
        o_logits, _ = o_q.forward_logits(sd, cfg, ids, pv, grid)
    assert torch.equal(o_pos, pos)
    d = max((o_emb - emb).abs().max().item(), (o_logits[:, -1] - logits).abs().max().item())
    return dict(seed=6, image_embeds=emb[::8].clone(), embed_row_stride=8, position_ids=pos.to(torch.int32), last_logits=logits,
                oracle_max_abs_diff=d)


def lookdown_inputs(cfg):
    return W.qwen_lookdown_inputs(cfg)


def gold_preprocess():
    """Frame pre-processing as the reference's host path runs it: PIL itself (Image.resize, bicubic 8-bit) and the installed
    transformers Qwen2VLImageProcessorPil on seeded camera-sized frames (small geometry so the fixture stays small):
    policy resize 80x60 -> 56x56 (internvla_n1_policy.py:105-116), processor (smart_resize -> 56x56, rescale, normalize, patchify),
    look-down pair resize to 32x32 then / 255.0 (internvla_n1_agent.py:309-317), plus an up-scaling and a non-square case."""
    from PIL import Image
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil

    from oracle import preprocess as o_pp

    rng = np.random.default_rng(2024)
    frames = rng.integers(0, 256, (3, 60, 80, 3), dtype=np.uint8)
    frames[1] = (np.linspace(0, 255, 60 * 80 * 3).reshape(60, 80, 3) % 256).astype(np.uint8)       # smooth ramp: exercises the rounding
    rw = rh = 56
    resized = np.stack([np.array(Image.fromarray(f).resize((rw, rh))) for f in frames])
    proc = Qwen2VLImageProcessorPil()
    hf = proc(images=[Image.fromarray(f) for f in resized], return_tensors="np")
    s1 = np.stack([np.array(Image.fromarray(f).resize((32, 32))) / 255.0 for f in frames])
    extra = [(frames[0], 100, 37), (frames[2][:33, :17], 40, 33)]                                    # (image, w, h): up-scale, identity height
    extra_out = [np.array(Image.fromarray(np.ascontiguousarray(im)).resize((w, h))) for im, w, h in extra]
    depth = (rng.random((2, 60, 80), dtype=np.float32) * 0.8).astype(np.float32)                      # metres / 10, some beyond the 5 m clip
    s1d = []
    for d in depth:
        x = np.array(Image.fromarray(d).resize((32, 32))) * 10.0
        x[x > 5.0] = 5.0
        s1d.append(x)
    s1d = np.stack(s1d)
    pv, grid = o_pp.qwen_pixel_values(frames, rw, rh)
    diffs = [np.abs(o_pp.pil_resize(f, rw, rh).astype(int) - r.astype(int)).max() for f, r in zip(frames, resized)]
    diffs += [np.abs(o_pp.pil_resize(np.ascontiguousarray(im), w, h).astype(int) - o.astype(int)).max() for (im, w, h), o in zip(extra, extra_out)]
    diffs += [float(np.abs(pv - hf["pixel_values"]).max()), float(np.abs(o_pp.s1_frames(frames, 32) - s1).max())]
    diffs += [float(np.abs(o_pp.s1_depth(depth, 32) - s1d).max())]
    assert np.array_equal(o_pp.s1_depth(depth, 32), s1d)
    assert (grid == hf["image_grid_thw"]).all()
    return dict(frames=torch.from_numpy(frames), resize_w=rw, resize_h=rh, resized=torch.from_numpy(resized),
                pixel_values=torch.from_numpy(hf["pixel_values"]), image_grid_thw=torch.from_numpy(np.asarray(hf["image_grid_thw"])),
                s1_size=32, s1=torch.from_numpy(s1), depth=torch.from_numpy(depth), s1_depth=torch.from_numpy(s1d),
                extra=[dict(image=torch.from_numpy(np.ascontiguousarray(im)), w=w, h=h, out=torch.from_numpy(o)) for (im, w, h), o in zip(extra, extra_out)],
                pillow=__import__("PIL").__version__, transformers=__import__("transformers").__version__,
                oracle_max_abs_diff=float(max(diffs)))


def _count_dropout_sites(module, fn):
    """how many times a forward of `fn` in train() mode passes an ACTIVE dropout: calls of nn.Dropout modules with p > 0 and of
    nn.MultiheadAttention modules built with dropout > 0 (attention-probability dropout). The SFT tape must draw one mask per site."""
    import torch.nn as nn

    counts = {"dropout": 0, "attention": 0}
    hooks = []
    for sub_m in module.modules():
        if isinstance(sub_m, nn.Dropout) and sub_m.p > 0:
            hooks.append(sub_m.register_forward_hook(lambda *_: counts.__setitem__("dropout", counts["dropout"] + 1)))
        elif isinstance(sub_m, nn.MultiheadAttention) and sub_m.dropout > 0:
            hooks.append(sub_m.register_forward_hook(lambda *_: counts.__setitem__("attention", counts["attention"] + 1)))
    was = module.training
    module.train()
    try:
        with torch.no_grad():
            fn()
    finally:
        module.train(was)
        for h in hooks:
            h.remove()
    return counts


def gold_sft(B=1, T=2, variant="ffn1536"):
    """SFT loss of the nextdit_async branch (internvla_n1.py:222-286) through the reference's own System-1 modules under autograd
    (dropout off: .eval()), transcribed line by line; the noise / time-step draws of :261-264 are seeded inputs. Saves the loss, the
    gradient w.r.t. the trajectory hidden states and, per parameter, the gradient norm + 32 sampled entries."""
    import torch.nn as nn
    import torch.nn.functional as F

    from . import sft as o_sft
    from .schedulers import FlowMatchEulerDiscreteScheduler

    nd, arch = R.nextdit_module(), R.n1_arch_module()
    cfg, convention = _ffn_variant(variant)
    sd = W.n1_nextdit_state_dict(seed=6, cfg=cfg)

    class S1(nn.Module):
        def __init__(self):
            super().__init__()
            self.traj_dit = nd.NextDiTCrossAttn(nd.NextDiTCrossAttnConfig(latent_embedding_size=768))
            self.noise_scheduler = FlowMatchEulerDiscreteScheduler()
            self.action_encoder = nn.Linear(3, 384, bias=True)
            self.pos_encoding = arch.SinusoidalPositionalEncoding(384)
            self.action_decoder = nn.Linear(384, 3, bias=True)
            self.cond_projector = nn.Sequential(nn.Linear(3584, 768), nn.GELU(approximate="tanh"), nn.Linear(768, 768))
            self.rgb_model = R.dinov2_vits()
            self.memory_encoder = arch.MemoryEncoder()
            self.rgb_resampler = arch.QFormer()

    with convention:
        m = S1()
    _check_ffn_width(m, cfg)
    m = _load_strict(m, sd, allow_missing_prefixes=("traj_dit.model.patch_embedder.", "rgb_resampler.visual_proj."))
    g = torch.Generator().manual_seed(6)
    hidden_q = torch.randn(B, 4, 3584, generator=g).requires_grad_(True)
    traj_images = torch.rand(B, T, 224, 224, 3, generator=g)
    traj_poses = torch.randn(B, T, 32, 3, generator=g)
    video_frame_num = torch.tensor([T] * (B - 1) + [max(1, T - 1)])
    noise = torch.randn(B * T, 32, 3, generator=g)
    indices = torch.randint(0, 1000, (B * T,), generator=g)
    mean = torch.FloatTensor([0.485, 0.456, 0.406]).view(1, 1, 3, 1, 1)
    std = torch.FloatTensor([0.229, 0.224, 0.225]).view(1, 1, 3, 1, 1)

    traj_hidden_states = hidden_q.unsqueeze(1).repeat(1, traj_poses.size(1), 1, 1).flatten(0, 1)
    loss_mask = torch.arange(traj_images.size(1)).expand(traj_images.size(0), traj_images.size(1)) < video_frame_num.unsqueeze(1)
    cur_images = traj_images.flatten(0, 1)
    pix_goal_images = traj_images[:, 0:1].repeat(1, traj_images.size(1), 1, 1, 1).flatten(0, 1)
    bsz = cur_images.size(0)
    images_dp = torch.stack([pix_goal_images, cur_images], dim=1).permute(0, 1, 4, 2, 3)
    images_dp_norm = (images_dp - mean) / std
    images_dp_feat = m.rgb_model.get_intermediate_layers(images_dp_norm.flatten(0, 1))[0].unflatten(dim=0, sizes=(bsz, -1))
    memory_feat = m.memory_encoder(images_dp_feat.flatten(1, 2))
    memory_feat = torch.cat([images_dp_feat.flatten(1, 2), memory_feat], dim=-1)
    memory_tokens = m.rgb_resampler(memory_feat)
    ths = m.cond_projector(traj_hidden_states)
    latents = torch.cat([memory_tokens, ths], dim=1)
    relative_poses = traj_poses.flatten(0, 1)
    timesteps = m.noise_scheduler.timesteps[indices]
    schedule_timesteps = m.noise_scheduler.timesteps
    step_indices = [(schedule_timesteps == t).nonzero().item() for t in timesteps]      # get_sigmas, internvla_n1_arch.py:189-198
    sigmas = m.noise_scheduler.sigmas[step_indices].flatten()
    while len(sigmas.shape) < relative_poses.dim():
        sigmas = sigmas.unsqueeze(-1)
    noisy_trajectory = (1 - sigmas) * relative_poses + sigmas * noise
    action_features = m.action_encoder(noisy_trajectory)
    pos_ids = torch.arange(relative_poses.shape[1]).reshape(1, -1).repeat(bsz, 1)
    action_features = action_features + m.pos_encoding(pos_ids)
    noise_pred = m.traj_dit(x=action_features, timestep=timesteps, z_latents=latents)
    noise_pred = m.action_decoder(noise_pred)
    target = noise - relative_poses
    loss = F.mse_loss(noise_pred.float(), target.float(), reduction="none")
    mask = loss_mask.flatten(0, 1)[:, None, None]
    loss = (loss * mask).sum() / mask.sum() / (loss.shape[1] * loss.shape[2])
    loss.backward()
    ref_grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}

    sd_o = {k: v.float().clone().requires_grad_(True) for k, v in sd.items()}
    hq_o = hidden_q.detach().clone().requires_grad_(True)
    loss_o = o_sft.nextdit_sft_loss(sd_o, hq_o, traj_images, traj_poses, video_frame_num, noise, indices)
    loss_o.backward()
    worst = abs(loss_o.item() - loss.item())
    samples = {}
    gscale = max(g.abs().max().item() for g in ref_grads.values())
    for k, gr in ref_grads.items():
        assert k in sd_o and sd_o[k].grad is not None, f"reference parameter {k} has a gradient, the oracle's has none"
        scale = gr.abs().max().item()
        rel_k = (gr - sd_o[k].grad).abs().max().item() / max(scale, 1e-12)
        if scale < 1e-6 * gscale:        # identically-zero gradients (norm_k.bias: softmax is invariant to a bias on every key): fp32 noise
            rel_k = 0.0
        if rel_k > 1e-3:
            print(f"  [sft] {k}: rel {rel_k:.3e} scale {scale:.3e}")
        worst = max(worst, rel_k)
        flat = gr.flatten()
        pick = torch.linspace(0, flat.numel() - 1, min(32, flat.numel())).long()
        samples[k] = dict(norm=gr.norm().item(), idx=pick, val=flat[pick].clone())
    worst = max(worst, ((hidden_q.grad - hq_o.grad).abs().max() / hidden_q.grad.abs().max()).item())
    no_grad = sorted(k for k, p in m.named_parameters() if p.grad is None)
    sites = _count_dropout_sites(m, lambda: m.traj_dit(x=action_features, timestep=timesteps, z_latents=torch.cat(
        [m.rgb_resampler(torch.cat([images_dp_feat.flatten(1, 2), m.memory_encoder(images_dp_feat.flatten(1, 2))], dim=-1)), ths], dim=1)))
    return dict(B=B, T=T, seed=6, weights_seed=6, dit_ffn=cfg["dit_ffn"], dropout_sites=sites, loss=loss.item(), d_hidden=hidden_q.grad.clone(), grads=samples, params_without_grad=no_grad,
                inputs=dict(hidden_q=hidden_q.detach(), traj_images=traj_images, traj_poses=traj_poses, video_frame_num=video_frame_num,
                            noise=noise, t_index=indices),
                oracle_max_abs_diff=worst)


ASYNC_ONLY_PREFIXES = ("rgb_model.", "memory_encoder.", "rgb_resampler.")


def gold_sft_plain(B=2, T=2):
    """SFT loss of the NON-async `nextdit` branch (internvla_n1.py:234,256-286: condition = cond_projector(trajectory hidden states) alone)
    through the reference's own NextDiTCrossAttn under autograd, transcribed line by line. The module set is the one
    InternVLAN1MetaModel builds for system1 = 'nextdit' (internvla_n1_arch.py:126-137): no rgb_model / memory_encoder / rgb_resampler."""
    import torch.nn as nn
    import torch.nn.functional as F

    from . import sft as o_sft
    from .schedulers import FlowMatchEulerDiscreteScheduler

    nd, arch = R.nextdit_module(), R.n1_arch_module()
    sd = {k: v for k, v in W.n1_nextdit_state_dict(seed=8).items() if not k.startswith(ASYNC_ONLY_PREFIXES)}

    class S1(nn.Module):
        def __init__(self):
            super().__init__()
            self.traj_dit = nd.NextDiTCrossAttn(nd.NextDiTCrossAttnConfig(latent_embedding_size=768))
            self.noise_scheduler = FlowMatchEulerDiscreteScheduler()
            self.action_encoder = nn.Linear(3, 384, bias=True)
            self.pos_encoding = arch.SinusoidalPositionalEncoding(384)
            self.action_decoder = nn.Linear(384, 3, bias=True)
            self.cond_projector = nn.Sequential(nn.Linear(3584, 768), nn.GELU(approximate="tanh"), nn.Linear(768, 768))

    m = _load_strict(S1(), sd, allow_missing_prefixes=("traj_dit.model.patch_embedder.",))
    g = torch.Generator().manual_seed(8)
    hidden_q = torch.randn(B, 4, 3584, generator=g).requires_grad_(True)
    traj_images = torch.rand(B, T, 224, 224, 3, generator=g)            # unused by this branch beyond its [B, T] shape
    traj_poses = torch.randn(B, T, 32, 3, generator=g)
    video_frame_num = torch.tensor([T] * (B - 1) + [max(1, T - 1)])
    noise = torch.randn(B * T, 32, 3, generator=g)
    indices = torch.randint(0, 1000, (B * T,), generator=g)

    traj_hidden_states = hidden_q.unsqueeze(1).repeat(1, traj_poses.size(1), 1, 1).flatten(0, 1)
    loss_mask = torch.arange(traj_images.size(1)).expand(traj_images.size(0), traj_images.size(1)) < video_frame_num.unsqueeze(1)
    traj_hidden_states = m.cond_projector(traj_hidden_states)
    latents = traj_hidden_states
    relative_poses = traj_poses.flatten(0, 1)
    bsz = relative_poses.shape[0]
    timesteps = m.noise_scheduler.timesteps[indices]
    schedule_timesteps = m.noise_scheduler.timesteps
    step_indices = [(schedule_timesteps == t).nonzero().item() for t in timesteps]      # get_sigmas, internvla_n1_arch.py:189-198
    sigmas = m.noise_scheduler.sigmas[step_indices].flatten()
    while len(sigmas.shape) < relative_poses.dim():
        sigmas = sigmas.unsqueeze(-1)
    noisy_trajectory = (1 - sigmas) * relative_poses + sigmas * noise
    action_features = m.action_encoder(noisy_trajectory)
    pos_ids = torch.arange(relative_poses.shape[1]).reshape(1, -1).repeat(bsz, 1)
    action_features = action_features + m.pos_encoding(pos_ids)
    noise_pred = m.traj_dit(x=action_features, timestep=timesteps, z_latents=latents)
    noise_pred = m.action_decoder(noise_pred)
    target = noise - relative_poses
    loss = F.mse_loss(noise_pred.float(), target.float(), reduction="none")
    mask = loss_mask.flatten(0, 1)[:, None, None]
    loss = (loss * mask).sum() / mask.sum() / (loss.shape[1] * loss.shape[2])
    loss.backward()
    ref_grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}

    sd_o = {k: v.float().clone().requires_grad_(True) for k, v in sd.items()}
    hq_o = hidden_q.detach().clone().requires_grad_(True)
    loss_o = o_sft.nextdit_sft_loss(sd_o, hq_o, traj_images, traj_poses, video_frame_num, noise, indices, use_async=False)
    loss_o.backward()
    worst = abs(loss_o.item() - loss.item())
    samples = {}
    gscale = max(g_.abs().max().item() for g_ in ref_grads.values())
    for k, gr in ref_grads.items():
        assert k in sd_o and sd_o[k].grad is not None, f"reference parameter {k} has a gradient, the oracle's has none"
        scale = gr.abs().max().item()
        rel_k = (gr - sd_o[k].grad).abs().max().item() / max(scale, 1e-12)
        if scale < 1e-6 * gscale:
            rel_k = 0.0
        if rel_k > 1e-3:
            print(f"  [sft_plain] {k}: rel {rel_k:.3e} scale {scale:.3e}")
        worst = max(worst, rel_k)
        flat = gr.flatten()
        pick = torch.linspace(0, flat.numel() - 1, min(32, flat.numel())).long()
        samples[k] = dict(norm=gr.norm().item(), idx=pick, val=flat[pick].clone())
    worst = max(worst, ((hidden_q.grad - hq_o.grad).abs().max() / hidden_q.grad.abs().max()).item())
    no_grad = sorted(k for k, p in m.named_parameters() if p.grad is None)
    sites = _count_dropout_sites(m, lambda: m.traj_dit(x=action_features, timestep=timesteps, z_latents=latents))
    return dict(B=B, T=T, seed=8, weights_seed=8, dropout_sites=sites, loss=loss.item(), d_hidden=hidden_q.grad.clone(), grads=samples, params_without_grad=no_grad,
                inputs=dict(hidden_q=hidden_q.detach(), traj_images=traj_images[:, :, :1, :1].clone(), traj_poses=traj_poses, video_frame_num=video_frame_num,
                            noise=noise, t_index=indices),
                oracle_max_abs_diff=worst)


def gold_sft_navdp(B=1, T=2):
    """SFT loss of the navdp_async branch (internvla_n1.py:287-303) through the reference's own NavDP_Policy_DPT_CriticSum_DAT.forward_vlm_traj
    (internvla_n1/navdp.py:291-312) under autograd, dropout off; sample_noise's random draws (:163-175) are replaced by seeded inputs fed
    through the module's own time_emb / noise_scheduler.add_noise / input_embed."""
    from . import sft as o_sft

    torch_load = torch.load
    torch.load = lambda *a, **k: {}
    try:
        n1 = R.n1_navdp_module()
        cfg = W.N1_NAVDP_CFG
        m = n1.NavDP_Policy_DPT_CriticSum_DAT(memory_size=cfg["memory_size"], navdp_version=0.1, input_dtype="fp32")
    finally:
        torch.load = torch_load
    sd = W.n1_navdp_state_dict(seed=7)
    m = _load_strict(m, sd, allow_missing_prefixes=("point_encoder.", "critic_head.", "pg_embed_mlp.", "pg_pred_mlp.", "decoder_layer."),
                     skip_buffers=("goal_compressor.positional_encoding.pe",))
    m.rgbd_encoder.input_dtype = torch.float32
    m.rgbd_encoder.preprocess_mean = m.rgbd_encoder.preprocess_mean.float()
    m.rgbd_encoder.preprocess_std = m.rgbd_encoder.preprocess_std.float()
    m.tgt_mask = m.tgt_mask.float()
    g = torch.Generator().manual_seed(7)
    hidden_q = torch.randn(B, 4, 3584, generator=g).requires_grad_(True)
    traj_images = torch.rand(B, T, 224, 224, 3, generator=g)
    traj_depths = torch.rand(B, T, 224, 224, generator=g) * 5.0
    traj_poses = torch.randn(B, T, 32, 3, generator=g)
    video_frame_num = torch.tensor([T] * (B - 1) + [max(1, T - 1)])
    noise = torch.randn(B * T, 32, 3, generator=g)
    timesteps = torch.randint(0, cfg["num_train_timesteps"], (B * T,), generator=g)

    def sample_noise(action):
        time_embeds = m.time_emb(timesteps).unsqueeze(1).float()
        noisy_action = m.noise_scheduler.add_noise(action, noise, timesteps)
        return noise, time_embeds, m.input_embed(noisy_action)
    m.sample_noise = sample_noise
    traj_hidden_states = hidden_q.unsqueeze(1).repeat(1, traj_poses.size(1), 1, 1).flatten(0, 1)
    loss_mask = torch.arange(traj_images.size(1)).expand(traj_images.size(0), traj_images.size(1)) < video_frame_num.unsqueeze(1)
    cur_images, cur_depths = traj_images.flatten(0, 1), traj_depths.flatten(0, 1)
    pix_goal_images = traj_images[:, 0:1].repeat(1, traj_images.size(1), 1, 1, 1).flatten(0, 1)
    pix_goal_depths = traj_depths[:, 0:1].repeat(1, traj_depths.size(1), 1, 1).flatten(0, 1)
    images_dp = torch.stack([pix_goal_images, cur_images], dim=1)
    depths_dp = torch.stack([pix_goal_depths, cur_depths], dim=1).unsqueeze(-1)
    pred_pg, nz = m.forward_vlm_traj(traj_hidden_states, images_dp, depths_dp, tensor_label_actions=traj_poses)
    pg_action_loss = (pred_pg - nz).square()
    mask = loss_mask.flatten(0, 1)[:, None, None]
    loss = (pg_action_loss * mask).sum() / mask.sum() / (pg_action_loss.shape[1] * pg_action_loss.shape[2])
    loss.backward()
    ref_grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}

    sd_o = {k: v.float().clone().requires_grad_(True) for k, v in sd.items()}
    hq_o = hidden_q.detach().clone().requires_grad_(True)
    loss_o = o_sft.navdp_sft_loss(sd_o, hq_o, traj_images, traj_depths, traj_poses, video_frame_num, noise, timesteps, cfg)
    loss_o.backward()
    worst = abs(loss_o.item() - loss.item())
    gscale = max(x.abs().max().item() for x in ref_grads.values())
    samples = {}
    for k, gr in ref_grads.items():
        assert k in sd_o and sd_o[k].grad is not None, f"reference parameter {k} has a gradient, the oracle's has none"
        scale = gr.abs().max().item()
        rel_k = 0.0 if scale < 1e-6 * gscale else (gr - sd_o[k].grad).abs().max().item() / scale
        if rel_k > 1e-3:
            print(f"  [sft_navdp] {k}: rel {rel_k:.3e} scale {scale:.3e}")
        worst = max(worst, rel_k)
        flat = gr.flatten()
        pick = torch.linspace(0, flat.numel() - 1, min(32, flat.numel())).long()
        samples[k] = dict(norm=gr.norm().item(), idx=pick, val=flat[pick].clone())
    worst = max(worst, ((hidden_q.grad - hq_o.grad).abs().max() / hidden_q.grad.abs().max()).item())
    sites = _count_dropout_sites(m, lambda: m.forward_vlm_traj(traj_hidden_states, images_dp, depths_dp, tensor_label_actions=traj_poses))
    return dict(B=B, T=T, weights_seed=7, dropout_sites=sites, loss=loss.item(), d_hidden=hidden_q.grad.clone(), grads=samples,
                inputs=dict(hidden_q=hidden_q.detach(), traj_images=traj_images, traj_depths=traj_depths, traj_poses=traj_poses,
                            video_frame_num=video_frame_num, noise=noise, timesteps=timesteps),
                oracle_max_abs_diff=worst)


def _variant(fn, variant):
    def run():
        return fn(variant=variant)
    return run


# the NextDiT fixtures exist for both FFN widths the reference's block can have (oracle/diffusers_blocks.py: LEGACY_TWO_THIRDS): the default
# files are the pinned diffusers==0.33.1 reading (1536), *_ffn1024 the diffusers <= 0.32 one
UNITS = {"n1_nextdit_ffn1024": _variant(gold_n1_nextdit, "ffn1024"), "sft_ffn1024": _variant(gold_sft, "ffn1024"),
         "sft_navdp": gold_sft_navdp, "sft": gold_sft, "sft_nextdit_plain": gold_sft_plain, "unet1d": gold_unet1d, "preprocess": gold_preprocess, "vln_utils": gold_vln_utils, "qwen_lookdown": gold_qwen_lookdown, "dinov2": gold_dinov2, "n1_nextdit": gold_n1_nextdit, "qwen": gold_qwen, "navdpnet": gold_navdpnet, "navdpnet_nogoal": gold_navdpnet_nogoal, "n1_navdp": gold_n1_navdp}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    GOLD.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(8)
    for name, fn in UNITS.items():
        if a.only and a.only != name:
            continue
        out = fn()
        print(f"[golden] {name}: oracle vs reference max|diff| = {out['oracle_max_abs_diff']:.3e}", flush=True)
        torch.save(out, GOLD / f"{name}.pt")


if __name__ == "__main__":
    main()
