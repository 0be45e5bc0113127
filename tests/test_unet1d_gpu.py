"""GPU parity of the diffusion-policy ConditionalUnet1D + DDIM head (internnav_amd.unet1d) against the fixture produced by
the VENDORED reference module (tests/golden/unet1d.pt, oracle/make_golden.py) and, at the bench batch, against the per-env CPU oracle.
Every convolution runs as an implicit GEMM over overlapping row windows of the padded channels-last buffers; bf16 activations between
the blocks, fp32 GroupNorm statistics and DDIM state. Tolerance: samples live in [-1, 1] (clip_sample): mean |err| <= 2e-3."""
from pathlib import Path

import pytest
import torch

from oracle import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_gn_mish_film_residual_op(built_lib):
    from internnav_amd import ops

    g = torch.Generator().manual_seed(3)
    for C, T, pad, dt in ((256, 32, 8, torch.bfloat16), (512, 16, 4, torch.float32), (1024, 8, 2, torch.float32), (512, 8, 2, torch.bfloat16)):
        seqs, Tp = 6, T + 2 * pad
        x = torch.randn(seqs * Tp, C, generator=g).to(dt).to(DEV)
        res = torch.randn(seqs * Tp, C, generator=g).to(torch.bfloat16).to(DEV)
        gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
        film_env = torch.randn(3, 4 * C, generator=g).to(DEV)
        film_step = torch.randn(4 * C, generator=g).to(DEV)
        out = torch.full((seqs * Tp, C), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.gn_mish(x, out, gamma, beta, seqs, T, pad, Tp, 8, residual=res, film_env=film_env, film_step=film_step, film_off=2 * C, seq_per_env=2)
        xv = x.float().view(seqs, Tp, C)[:, :T]                                   # conv-output indexing: valid rows first
        y = torch.nn.functional.group_norm(xv.transpose(1, 2), 8, gamma, beta, eps=1e-5).transpose(1, 2)
        y = y * torch.tanh(torch.nn.functional.softplus(y))
        e = (film_env + film_step)[:, 2 * C:].repeat_interleave(2, dim=0)
        y = y * e[:, None, :C] + e[:, None, C:]
        y = y + res.float().view(seqs, Tp, C)[:, pad:pad + T]
        o = out.float().view(seqs, Tp, C)
        assert torch.equal(o[:, :pad], torch.zeros_like(o[:, :pad])) and torch.equal(o[:, pad + T:], torch.zeros_like(o[:, pad + T:]))
        assert torch.allclose(o[:, pad:pad + T], y, atol=3e-2, rtol=2e-2), (C, T, (o[:, pad:pad + T] - y).abs().max().item())


def test_unet1d_ddim_vs_vendored_reference_fixture(built_lib):
    from internnav_amd.unet1d import UNet1DHead

    gold = torch.load(Path(__file__).resolve().parent / "golden" / "unet1d.pt", weights_only=True)
    cfg = W.UNET1D_CFG
    sd = W.materialize(W.unet1d_spec(cfg), seed=gold["seed"])
    inp = W.unet1d_inputs(gold["B"], seed=gold["seed"], cfg=cfg)
    B = gold["B"]
    eng = UNet1DHead(sd, cfg, DEV, max_envs=B)
    out = eng.sample_traj(inp["global_cond"].to(DEV), inp["x_init"].to(DEV)).float().cpu()
    d = (out - gold["samples"]).abs()
    y = gold["bf16_autocast_err"]
    print(f"unet1d ddim samples: mean|err| {d.mean():.3e} max|err| {d.max():.3e} (range [-1, 1]); bf16-autocast PyTorch vs fp32: mean {y['mean']:.3e} max {y['max']:.3e}")
    # diffusers' default step keeps the network's eps where x0 was clipped: the recursion does not contract onto the clip bounds and bf16
    # PyTorch itself sits at ~1.4e-2 from fp32 on this fixture. Bar: not further from fp32 than bf16 PyTorch is.
    assert d.mean().item() <= y["mean"] and d.max().item() <= max(1.5 * y["max"], 1e-1)
    assert eng.sched["timesteps"] == gold["timesteps"].tolist()
    # the other value of diffusers' step(use_clipped_model_output=) has its own fixture
    out_c = eng.sample_traj(inp["global_cond"].to(DEV), inp["x_init"].to(DEV), use_clipped_model_output=True).float().cpu()
    dc = (out_c - gold["samples_use_clipped_model_output"]).abs()
    yc = gold["bf16_autocast_err_use_clipped_model_output"]
    print(f"unet1d ddim samples (use_clipped_model_output): mean|err| {dc.mean():.3e} max|err| {dc.max():.3e}; bf16-autocast PyTorch mean {yc['mean']:.3e}")
    assert dc.mean().item() <= max(yc["mean"], 2e-3) and dc.max().item() <= max(1.5 * yc["max"], 1e-1)
    assert (out - out_c).abs().max().item() > 1e-2           # the fixture clips: the two variants are different trajectories
    # first noise prediction alone (the network without the sampler)
    nseq = B * eng.S
    eng.sample[: nseq * eng.T].copy_(inp["x_init"].reshape(-1, 3).to(DEV))
    eng.xin.zero_()
    eng.xin[: nseq * eng.Tp[0]].view(nseq, eng.Tp[0], 8)[:, 8:8 + eng.T, :3].copy_(inp["x_init"].reshape(nseq, eng.T, 3).to(DEV))
    eng._forward(nseq, 0)
    eps = eng.eps[: nseq * eng.Tp[0]].view(nseq, eng.Tp[0], 4)[:, 8:8 + eng.T, :3].float().cpu().reshape(B, eng.S, eng.T, 3)
    de = (eps - gold["eps0"]).abs()
    print(f"unet1d eps(t={eng.sched['timesteps'][0]}): mean|err| {de.mean():.3e} max|err| {de.max():.3e} ref rms {gold['eps0'].pow(2).mean().sqrt():.3f}")
    assert de.mean().item() < 1e-2 * gold["eps0"].pow(2).mean().sqrt().item()


def test_unet1d_b64_vs_per_env_oracle_and_timing(built_lib):
    """64 envs x 32 samples in one call (2048 sequences): envs 0 / 63 vs the per-env CPU oracle; prints the call time."""
    from internnav_amd.unet1d import UNet1DHead
    from oracle import unet1d as o_u

    cfg = W.UNET1D_CFG
    sd = W.materialize(W.unet1d_spec(cfg), seed=9)
    B = 64
    inp = W.unet1d_inputs(B, seed=9, cfg=cfg)
    eng = UNet1DHead(sd, cfg, DEV, max_envs=B)
    gc, xi = inp["global_cond"].to(DEV), inp["x_init"].to(DEV)
    out = eng.sample_traj(gc, xi).float().cpu()
    for b in (0, 63):
        with torch.no_grad():
            ref = o_u.ddim_sample(sd, inp["global_cond"][b:b + 1], inp["x_init"][b:b + 1], cfg["num_train_timesteps"], cfg["num_inference_steps"])
            with torch.autocast("cpu", dtype=torch.bfloat16):      # yardstick: the same env in bf16 PyTorch (see the fixture test above)
                r16 = o_u.ddim_sample(sd, inp["global_cond"][b:b + 1], inp["x_init"][b:b + 1], cfg["num_train_timesteps"], cfg["num_inference_steps"]).float()
        d, y = (out[b] - ref[0]).abs(), (r16[0] - ref[0]).abs()
        print(f"unet1d B=64 env {b}: mean|err| {d.mean():.3e} max|err| {d.max():.3e}; bf16-autocast PyTorch mean {y.mean():.3e} max {y.max():.3e}")
        assert d.mean().item() <= 1.25 * y.mean().item() and d.max().item() <= max(1.5 * y.max().item(), 1e-1)
    torch.cuda.synchronize()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        eng.sample_traj(gc, xi)
    z.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(z) / 3
    print(f"unet1d head, 64 envs x 32 samples x 10 DDIM steps (eager launches): {ms:.1f} ms per call = {64e3 / ms:.0f} policy steps/s, {23.8 / ms * 1e3:.0f} TF/s")
