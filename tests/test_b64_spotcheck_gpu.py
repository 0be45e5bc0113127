"""GPU parity at the BASELINE batch size: envs 0 / 31 / 63 of ONE 64-env call of each System-1 engine against the per-env CPU oracle
(the reference's own semantics: one env per call). Batch size changes the GEMM tile selection, so B = 64 is a different code path from
the B = 2 fixtures. The oracle runs on the host cores of the GPU box (a few seconds per env)."""
import pytest
import torch

from oracle import navdp as o_navdp
from oracle import nextdit as o_nextdit
from oracle import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ENVS = (0, 31, 63)


def test_navdpnet_b64_vs_per_env_oracle(built_lib):
    """BASELINE config #2 at its own batch: NavDPNet, 64 envs, 10 DDPM steps, 32 samples + critic."""
    from internnav_amd.navdp import NavDPNet

    B, cfg = 64, W.NAVDPNET_CFG
    sd = W.navdpnet_state_dict(seed=21)
    inp = W.navdpnet_inputs(B, seed=21)
    net = NavDPNet(sd, cfg, DEV, max_envs=B)
    d = {k: v.to(DEV) for k, v in inp.items()}
    neg, pos = net.predict_pointgoal_batch_action_vel(d["goal"], d["images"], d["depths"], d["x_init"], d["step_noise"])
    S, T = net.S, net.T
    fin = net.sample[: B * S * T].view(B, S, T, 3).float().cpu()
    cr = net.critic[: B * S].view(B, S).float().cpu()
    for b in ENVS:
        with torch.no_grad():
            o_neg, o_pos, o_fin, o_cr, _ = o_navdp.navdpnet_pointgoal(sd, inp["goal"][b:b + 1], inp["images"][b:b + 1], inp["depths"][b:b + 1],
                                                                      inp["x_init"][b:b + 1], inp["step_noise"][:, b:b + 1], cfg, return_all=True)
        e = (fin[b] - o_fin[0]).abs()
        ec = (cr[b] - o_cr[0]).abs()
        # yardstick: the same oracle under bf16 autocast = the precision the reference itself runs at. Ten DDPM steps with clipping amplify
        # last-bit differences in a few elements: over the 64 envs of this batch bf16 PyTorch's own max|err| against fp32 has a median of
        # 4.7e-2 and exceeds 5e-2 for 29 of them (profiles/r03E_navdp_bf16_yardstick_cpu.log), env 31 sits at 4.63e-2 - where the engine
        # measures 4.65e-2. The max bound therefore never asks for more than 1.25x the reference's own precision on the same env.
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            _, _, y_fin, _, _ = o_navdp.navdpnet_pointgoal(sd, inp["goal"][b:b + 1], inp["images"][b:b + 1], inp["depths"][b:b + 1],
                                                           inp["x_init"][b:b + 1], inp["step_noise"][:, b:b + 1], cfg, return_all=True)
        ey = (y_fin[0].float() - o_fin[0]).abs()
        print(f"NavDPNet B=64 env {b}: samples mean|err| {e.mean():.3e} max {e.max():.3e} (bf16 PyTorch: mean {ey.mean():.3e} max {ey.max():.3e}); "
              f"critic max|err| {ec.max():.3e} (range {o_cr.abs().max():.2f})")
        assert e.mean().item() < 1.5e-3     # measured 1.09e-3 at B = 64 (8.2e-4 on the B = 2 fixture): 10 DDPM steps with clip
        assert e.max().item() < max(5e-2, 1.25 * ey.max().item())
        assert ec.max().item() < 5e-2 * max(1.0, o_cr.abs().max().item())
        order = o_cr[0].argsort()
        if (o_cr[0][order[8]] - o_cr[0][order[7]]) > 2 * ec.max():
            assert set(cr[b].argsort()[:8].tolist()) == set(order[:8].tolist())
        if torch.equal(cr[b].argsort()[:8], order[:8]):
            assert (neg[b].cpu() - o_neg[0]).abs().max().item() < 1e-1


def test_nextdit_b64_vs_per_env_oracle(built_lib):
    """DualVLN System-1 at 64 envs per call (the bench's batch): 10 flow-matching steps, 32 samples."""
    from internnav_amd.nextdit import NextDiTSystem1

    B, cfg = 64, W.N1_NEXTDIT_CFG
    sd = W.n1_nextdit_state_dict(seed=22)
    inp = W.n1_nextdit_inputs(B, seed=22)
    eng = NextDiTSystem1(sd, cfg, DEV, max_envs=B)
    out = eng.generate_traj(inp["traj_latents"].to(DEV, torch.bfloat16), inp["images"].to(DEV, torch.bfloat16), inp["x_init"].to(DEV)).float().cpu()
    for b in ENVS:
        with torch.no_grad():
            ref = o_nextdit.generate_traj(sd, inp["traj_latents"][b:b + 1], inp["images"][b:b + 1], inp["x_init"][b:b + 1])
        ref = ref.reshape(out[b].shape)
        e = (out[b] - ref).abs()
        scale = max(1.0, ref.abs().max().item())
        print(f"NextDiT B=64 env {b}: latents mean|err| {e.mean():.3e} max {e.max():.3e} (ref max {ref.abs().max():.2f})")
        assert e.mean().item() < 1e-3 * scale and e.max().item() < 5e-2 * scale
