"""CPU: the oracle restatement against the golden fixtures produced by the REAL reference modules (oracle/make_golden.py).

This is what pins the oracle (the reference ships no known-answer vectors of its own). The fixtures hold
fp32 outputs of the reference's classes run in the build container on the seeded weights/inputs of oracle/weights.py.
"""
from pathlib import Path

import pytest
import torch

from oracle import dinov2 as o_dino
from oracle import navdp as o_navdp
from oracle import weights as W
from oracle.schedulers import DDPMScheduler, FlowMatchEulerDiscreteScheduler

GOLD = Path(__file__).resolve().parent / "golden"


def _load(name):
    return torch.load(GOLD / f"{name}.pt", weights_only=True)


def test_dinov2_matches_reference():
    gold = _load("dinov2")
    sd = W.materialize(W.dinov2_vits_spec(), seed=gold["seed"])
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(gold["img_seed"]))
    with torch.no_grad():
        out = o_dino.forward_tokens(img, sd)
    assert out.shape == gold["tokens"].shape
    assert (out - gold["tokens"]).abs().max().item() < 1e-4


def test_navdpnet_matches_reference():
    """NavDPNet.predict_pointgoal_batch_action_vel of the reference (batch-1 calls) == batched oracle, same injected noise."""
    gold = _load("navdpnet")
    B = gold["B"]
    sd = W.navdpnet_state_dict(seed=gold["seed"])
    inp = W.navdpnet_inputs(B, seed=gold["seed"])
    with torch.no_grad():
        neg, pos, fin, critic, rgbd = o_navdp.navdpnet_pointgoal(sd, inp["goal"], inp["images"], inp["depths"], inp["x_init"],
                                                                 inp["step_noise"], W.NAVDPNET_CFG, return_all=True)
    assert (rgbd - gold["rgbd_embed"]).abs().max().item() < 1e-4
    assert (neg - gold["negative"]).abs().max().item() < 1e-4
    assert (pos - gold["positive"]).abs().max().item() < 1e-4


def test_navdpnet_nogoal_matches_reference():
    """NavDPNet.predict_nogoal_batch_action_vel of the reference (navdp_policy.py:323-339, batch-1 calls) == the oracle with goal None; the
    zero goal is a different policy output than the point goal of the sibling fixture."""
    gold, gold_pg = _load("navdpnet_nogoal"), _load("navdpnet")
    B = gold["B"]
    sd = W.navdpnet_state_dict(seed=gold["seed"])
    inp = W.navdpnet_inputs(B, seed=gold["seed"])
    with torch.no_grad():
        neg, pos, fin, critic, _ = o_navdp.navdpnet_pointgoal(sd, None, inp["images"], inp["depths"], inp["x_init"], inp["step_noise"],
                                                              W.NAVDPNET_CFG, return_all=True)
    assert (neg - gold["negative"]).abs().max().item() < 1e-4
    assert (pos - gold["positive"]).abs().max().item() < 1e-4
    assert (fin - gold["oracle_final"]).abs().max().item() < 1e-4
    assert (gold["oracle_final"] - gold_pg["oracle_final"]).abs().max().item() > 1e-2


def test_n1_navdp_head_matches_reference():
    gold = _load("n1_navdp")
    B = gold["B"]
    sd = W.n1_navdp_state_dict(seed=gold["seed"])
    inp = W.n1_navdp_inputs(B, seed=gold["seed"])
    with torch.no_grad():
        out = o_navdp.n1_navdp_async(sd, inp["vlm_tokens"], inp["images"], inp["depths"], inp["x_init"], inp["step_noise"],
                                     W.N1_NAVDP_CFG)
    assert (out - gold["trajectories"]).abs().max().item() < 1e-4


def test_ddpm_scheduler_known_properties():
    """diffusers is absent (parity unpinned): check the restated DDPM against closed-form properties of the algorithm."""
    for K in (10, 20):
        s = DDPMScheduler(num_train_timesteps=K)
        s.set_timesteps(K)
        assert s.timesteps.tolist() == list(range(K - 1, -1, -1))
        assert torch.all(s.betas > 0) and torch.all(s.betas <= 0.999)
        assert abs(float(s.alphas_cumprod[0]) - (1 - float(s.betas[0]))) < 1e-7
        # t = 0: no noise, x_prev = clipped x0 prediction
        x = torch.randn(4, 3)
        e = torch.randn(4, 3)
        out = s.step(e, 0, x).prev_sample
        a0 = s.alphas_cumprod[0]
        assert torch.allclose(out, ((x - (1 - a0).sqrt() * e) / a0.sqrt()).clamp(-1, 1), atol=1e-6)
        # posterior mean coefficients sum to the DDPM identity when eps is the true noise and no clipping is active
        x0 = torch.rand(4, 3) * 0.5
        t = K // 2
        noise = torch.randn(4, 3)
        xt = s.add_noise(x0, noise, torch.tensor([t]))
        _, _, c0, ct, _ = s.coefficients(t)
        mean = s.step(noise, t, xt, noise=torch.zeros(4, 3)).prev_sample
        assert torch.allclose(mean, c0 * x0 + ct * xt, atol=1e-5)


def test_flow_match_euler_scheduler():
    import numpy as np

    s = FlowMatchEulerDiscreteScheduler()
    s.set_timesteps(10, sigmas=np.linspace(1.0, 0.1, 10))
    assert torch.allclose(s.timesteps, torch.linspace(1000, 100, 10))
    x = torch.ones(2, 3)
    for t in s.timesteps:
        x = s.step(torch.ones(2, 3), t, x).prev_sample
    assert torch.allclose(x, torch.zeros(2, 3), atol=1e-6)  # integrates v = 1 from sigma 1 to 0


FFN_FIXTURES = [("", W.N1_NEXTDIT_CFG), ("_ffn1024", W.N1_NEXTDIT_CFG_FFN1024)]
"""the NextDiT fixtures exist for both FFN widths the reference's in-tree block can have: 1536 (LuminaFeedForward of the pinned
diffusers 0.33.1) and 1024 (diffusers <= 0.32) - oracle/diffusers_blocks.py: LEGACY_TWO_THIRDS"""


def test_lumina_ffn_width_conventions():
    from internnav_amd import synthetic
    from oracle import diffusers_blocks as blk

    assert blk.LuminaFeedForward(384, 4 * 384, 256, None).linear_1.weight.shape == (1536, 384)          # 0.33.1 (pinned): inner_dim as given
    with blk.ffn_convention(legacy_two_thirds=True):
        assert blk.LuminaFeedForward(384, 4 * 384, 256, None).linear_1.weight.shape == (1024, 384)      # <= 0.32: int(2 * inner_dim / 3)
    assert blk.LEGACY_TWO_THIRDS is False
    assert synthetic.lumina_ffn_width(384) == 1536 and synthetic.lumina_ffn_width(384, legacy_two_thirds=True) == 1024
    assert W.N1_NEXTDIT_CFG["dit_ffn"] == 1536 and W.N1_NEXTDIT_CFG_FFN1024["dit_ffn"] == 1024
    for cfg in (W.N1_NEXTDIT_CFG, W.N1_NEXTDIT_CFG_FFN1024):      # geometry read back off the tensor shapes (what from_pretrained does)
        spec = synthetic.n1_nextdit_spec(cfg)
        shapes = {k: torch.empty(v[0], device="meta") for k, v in spec.items()}
        assert synthetic.n1_nextdit_cfg_from_weights(shapes) == cfg


@pytest.mark.parametrize("suffix,cfg", FFN_FIXTURES)
def test_n1_nextdit_matches_reference(suffix, cfg):
    """generate_traj (nextdit_async) driven through the reference's own NextDiT / MemoryEncoder / QFormer / DINOv2 modules."""
    from oracle import nextdit as o_nd

    gold = _load("n1_nextdit" + suffix)
    assert gold["dit_ffn"] == cfg["dit_ffn"]
    B = gold["B"]
    sd = W.n1_nextdit_state_dict(seed=gold["seed"], cfg=cfg)
    inp = W.n1_nextdit_inputs(B, seed=gold["seed"])
    with torch.no_grad():
        out = o_nd.generate_traj(sd, inp["traj_latents"], inp["images"], inp["x_init"])
    assert (out - gold["latents"]).abs().max().item() < 1e-4


def test_qwen_s2_matches_transformers_and_reference_rope_index():
    """System-2 oracle vs the fixture from the installed transformers Qwen2.5-VL modules + the reference's get_rope_index_25."""
    from oracle import qwen_vl as o_q

    gold = _load("qwen")
    cfg = W.QWEN_TEST_CFG
    sd = W.qwen_state_dict(seed=gold["seed"], cfg=cfg)
    inp = W.qwen_inputs(gold["B"], gold["n_img"], seed=gold.get("input_seed", gold["seed"]), cfg=cfg)
    with torch.no_grad():
        pos, _ = o_q.rope_index(inp["input_ids"], inp["grid_thw"], cfg["image_token_id"], cfg["vision_start_id"])
        assert torch.equal(pos, gold["position_ids"].long())
        emb = o_q.vision_tower(inp["pixel_values"], inp["grid_thw"], sd, cfg)
        assert (emb[:: gold["embed_row_stride"]] - gold["image_embeds"]).abs().max().item() < 1e-3
        logits, _ = o_q.forward_logits(sd, cfg, inp["input_ids"], inp["pixel_values"], inp["grid_thw"])
        assert (logits[:, -1] - gold["last_logits"]).abs().max().item() < 1e-3
        lat = o_q.generate_latents(sd, cfg, gold["generated"], inp["pixel_values"], inp["grid_thw"])
        assert (lat - gold["latents"]).abs().max().item() < 1e-3


def test_unet1d_matches_vendored_reference():
    """diffusion-policy ConditionalUnet1D (vendored, conditional_unet1d.py:69-241) executed by oracle/make_golden.py: one noise prediction and
    the 10-step DDIM loop == the oracle restatement on the same seeded weights / inputs."""
    from oracle import unet1d as o_u

    gold = _load("unet1d")
    cfg = W.UNET1D_CFG
    sd = W.materialize(W.unet1d_spec(cfg), seed=gold["seed"])
    inp = W.unet1d_inputs(gold["B"], seed=gold["seed"], cfg=cfg)
    B, S, T, D = inp["x_init"].shape
    with torch.no_grad():
        eps = o_u.unet_forward(sd, inp["x_init"].reshape(B * S, T, D), int(gold["timesteps"][0]), inp["global_cond"].repeat_interleave(S, dim=0))
        out = o_u.ddim_sample(sd, inp["global_cond"], inp["x_init"], cfg["num_train_timesteps"], cfg["num_inference_steps"])
        out_c = o_u.ddim_sample(sd, inp["global_cond"], inp["x_init"], cfg["num_train_timesteps"], cfg["num_inference_steps"], use_clipped_model_output=True)
    assert (eps.reshape(B, S, T, D) - gold["eps0"]).abs().max().item() < 1e-4
    assert (out - gold["samples"]).abs().max().item() < 1e-4
    assert (out_c - gold["samples_use_clipped_model_output"]).abs().max().item() < 1e-4
    assert gold["clipped_x0_elements"] > 0 and (gold["samples"] - gold["samples_use_clipped_model_output"]).abs().max().item() > 1e-3   # the clip is exercised
    assert float(out.abs().max()) <= 1.0 + 1e-6       # clip_sample: the last step (alpha_prev = 1) returns the clipped x0


def test_ddim_scheduler_known_properties():
    """diffusers is absent (parity unpinned): closed-form properties of the restated DDIM step (eta = 0)."""
    from oracle.schedulers import DDIMScheduler

    sch = DDIMScheduler(num_train_timesteps=100)
    sch.set_timesteps(10)
    assert sch.timesteps.tolist() == [90, 80, 70, 60, 50, 40, 30, 20, 10, 0]
    g = torch.Generator().manual_seed(0)
    x0 = torch.rand(4, 8, 3, generator=g) * 1.6 - 0.8
    eps = torch.randn(4, 8, 3, generator=g)
    for t in (90, 40):
        xt = sch.add_noise(x0, eps, torch.tensor(t))
        prev = sch.step(eps, t, xt).prev_sample                    # the true noise -> exactly the forward process at t - 10
        assert torch.allclose(prev, sch.add_noise(x0, eps, torch.tensor(t - 10)), atol=1e-5)
    xt = sch.add_noise(x0, eps, torch.tensor(0))
    assert torch.allclose(sch.step(eps, 0, xt).prev_sample, x0, atol=1e-5)     # set_alpha_to_one: the last step returns x0
    # diffusers' default (use_clipped_model_output=False, what diffusion_unet_lowdim_policy.py:87-91 runs): when the clip is active the
    # direction term keeps the model's eps; True re-derives it from the clipped x0. Both written out from DDIMScheduler.step's formulas.
    big = 3.0 * torch.randn(4, 8, 3, generator=g)
    a_t, a_p = sch.alphas_cumprod[40], sch.alphas_cumprod[30]
    x0c = ((big - (1 - a_t).sqrt() * eps) / a_t.sqrt()).clamp(-1, 1)
    assert (x0c.abs() == 1).any()
    assert torch.allclose(sch.step(eps, 40, big).prev_sample, a_p.sqrt() * x0c + (1 - a_p).sqrt() * eps, atol=1e-5)
    eps_c = (big - a_t.sqrt() * x0c) / (1 - a_t).sqrt()
    assert torch.allclose(sch.step(eps, 40, big, use_clipped_model_output=True).prev_sample, a_p.sqrt() * x0c + (1 - a_p).sqrt() * eps_c, atol=1e-5)


@pytest.mark.parametrize("suffix,cfg", FFN_FIXTURES)
def test_sft_loss_and_gradients_match_reference_autograd(suffix, cfg):
    """SFT loss of the nextdit_async branch (internvla_n1.py:222-286): torch autograd of the oracle restatement against the fixture made
    by back-propagating through the reference's own NextDiT / MemoryEncoder / QFormer / DINOv2 modules (oracle/make_golden.py gold_sft)."""
    from oracle import sft as o_sft

    gold = _load("sft" + suffix)
    assert gold["dit_ffn"] == cfg["dit_ffn"]
    sd = {k: v.float().clone().requires_grad_(True) for k, v in W.n1_nextdit_state_dict(seed=gold["weights_seed"], cfg=cfg).items()}
    inp = gold["inputs"]
    hq = inp["hidden_q"].clone().requires_grad_(True)
    loss = o_sft.nextdit_sft_loss(sd, hq, inp["traj_images"], inp["traj_poses"], inp["video_frame_num"], inp["noise"], inp["t_index"])
    loss.backward()
    assert abs(loss.item() - gold["loss"]) < 1e-5 * abs(gold["loss"])
    assert ((hq.grad - gold["d_hidden"]).abs().max() / gold["d_hidden"].abs().max()).item() < 1e-4
    gscale = max(g["norm"] for g in gold["grads"].values())
    for k, g in gold["grads"].items():
        mine = sd[k].grad
        assert mine is not None, k
        if g["norm"] < 1e-6 * gscale:          # identically zero (norm_k.bias)
            assert mine.norm().item() < 1e-5 * gscale, k
            continue
        assert abs(mine.norm().item() - g["norm"]) < 1e-4 * g["norm"], k
        assert (mine.flatten()[g["idx"]] - g["val"]).abs().max().item() < 1e-4 * max(g["val"].abs().max().item(), g["norm"] / mine.numel() ** 0.5), k
    # tensors the loss never touches in the reference: the same ones get no gradient here
    for k in gold["params_without_grad"]:
        if k in sd:
            assert sd[k].grad is None or float(sd[k].grad.abs().max()) == 0.0, k


def test_plain_nextdit_sft_loss_matches_reference_autograd():
    """system1 = 'nextdit' (internvla_n1.py:256-258): oracle autograd vs the fixture back-propagated through the reference's own modules."""
    from oracle import sft as o_sft

    gold = _load("sft_nextdit_plain")
    sd = {k: v.float().clone().requires_grad_(True) for k, v in W.n1_nextdit_state_dict(seed=gold["weights_seed"]).items()}
    inp = gold["inputs"]
    hq = inp["hidden_q"].clone().requires_grad_(True)
    loss = o_sft.nextdit_sft_loss(sd, hq, inp["traj_images"], inp["traj_poses"], inp["video_frame_num"], inp["noise"], inp["t_index"], use_async=False)
    loss.backward()
    assert abs(loss.item() - gold["loss"]) < 1e-5 * abs(gold["loss"])
    assert ((hq.grad - gold["d_hidden"]).abs().max() / gold["d_hidden"].abs().max()).item() < 1e-4
    gscale = max(g["norm"] for g in gold["grads"].values())
    for k, g in gold["grads"].items():
        mine = sd[k].grad
        assert mine is not None, k
        if g["norm"] < 1e-6 * gscale:
            assert mine.norm().item() < 1e-5 * gscale, k
            continue
        assert abs(mine.norm().item() - g["norm"]) < 1e-4 * g["norm"], k
        assert (mine.flatten()[g["idx"]] - g["val"]).abs().max().item() < 1e-4 * max(g["val"].abs().max().item(), g["norm"] / mine.numel() ** 0.5), k
    for k in sd:                         # the async-only modules never reach this loss
        if k.startswith(("rgb_model.", "memory_encoder.", "rgb_resampler.")):
            assert sd[k].grad is None, k


def test_flow_match_scheduler_default_state():
    """FlowMatchEulerDiscreteScheduler() as constructed (what the SFT loss indexes): timesteps 1000..1, sigmas = t / 1000."""
    s = FlowMatchEulerDiscreteScheduler()
    assert s.timesteps.shape == (1000,) and s.timesteps[0].item() == 1000.0 and s.timesteps[-1].item() == 1.0
    assert torch.allclose(s.sigmas, s.timesteps / 1000.0)


def test_navdp_sft_loss_and_gradients_match_reference_autograd():
    """navdp_async SFT branch: autograd of oracle/sft.navdp_sft_loss vs the fixture made by back-propagating through the reference's own
    NavDP_Policy_DPT_CriticSum_DAT.forward_vlm_traj (oracle/make_golden.py gold_sft_navdp)."""
    from oracle import sft as o_sft

    gold = _load("sft_navdp")
    sd = {k: v.float().clone().requires_grad_(True) for k, v in W.n1_navdp_state_dict(seed=gold["weights_seed"]).items()}
    inp = gold["inputs"]
    hq = inp["hidden_q"].clone().requires_grad_(True)
    loss = o_sft.navdp_sft_loss(sd, hq, inp["traj_images"], inp["traj_depths"], inp["traj_poses"], inp["video_frame_num"], inp["noise"],
                                inp["timesteps"], W.N1_NAVDP_CFG)
    loss.backward()
    assert abs(loss.item() - gold["loss"]) < 1e-5 * abs(gold["loss"])
    assert ((hq.grad - gold["d_hidden"]).abs().max() / gold["d_hidden"].abs().max()).item() < 1e-4
    gscale = max(g["norm"] for g in gold["grads"].values())
    for k, g in gold["grads"].items():
        mine = sd[k].grad
        assert mine is not None, k
        if g["norm"] < 1e-6 * gscale:
            assert mine.norm().item() < 1e-5 * gscale, k
            continue
        assert abs(mine.norm().item() - g["norm"]) < 1e-4 * g["norm"], k
        assert (mine.flatten()[g["idx"]] - g["val"]).abs().max().item() < 1e-4 * max(g["val"].abs().max().item(), g["norm"] / mine.numel() ** 0.5), k


def test_latent_query_oracle_cached_equals_whole_sequence_autograd():
    """oracle/sft.LatentQueryOracle (prefix on a KV cache without grad, query rows differentiated one layer at a time - what
    oracle/make_golden_sft_full.py runs at 28 layers) gives the gradient torch autograd computes over the WHOLE sequence, i.e. the
    reference's computation (internvla_n1.py:166-172, 222-227 with every LLM weight frozen), on the reduced configuration."""
    from oracle import qwen_vl as o_q
    from oracle import sft as o_sft

    cfg = W.QWEN_TEST_CFG
    sd = {k: v.float() for k, v in W.qwen_state_dict(seed=11, cfg=cfg).items()}
    inp = W.qwen_inputs(1, 1, seed=9, cfg=cfg)
    G = torch.randn(1, cfg["n_query"], cfg["t_hidden"], generator=torch.Generator().manual_seed(5))
    lq = sd["model.latent_queries"].clone().requires_grad_(True)
    sd_w = dict(sd)
    sd_w["model.latent_queries"] = lq
    h = o_q.generate_latents(sd_w, cfg, inp["input_ids"], inp["pixel_values"], inp["grid_thw"])
    (h * G).sum().backward()
    orc = o_sft.LatentQueryOracle(sd, cfg, inp["input_ids"], inp["pixel_values"], inp["grid_thw"])
    g = orc.backward(G)
    assert ((orc.hidden - h.detach()).abs().max() / h.detach().abs().max()).item() < 1e-5
    assert ((g - lq.grad.reshape(g.shape)).norm() / lq.grad.norm()).item() < 1e-5
