"""GPU parity of the NavDP System-1 engines (internnav_amd.navdp) against the golden outputs of the REAL reference
modules (tests/golden, produced by oracle/make_golden.py) and against the CPU oracle on further seeded cases.

Tolerance: the engines compute in bf16 on MFMA with fp32 accumulation and an fp32 residual stream; the reference
fixtures are fp32. BASELINE.json asks for 1e-3 "bf16 tolerance" on waypoints: we check the waypoint error normalised by
the waypoint range (samples live in [-1, 1]) - mean abs error <= 1e-3 and a max-abs bound that covers bf16 operand
rounding (2^-9 relative per operand) amplified through 12 ViT + 2 former + 16 decoder layers x 10-20 sampler steps.
"""
import pytest
import torch

from oracle import dinov2 as o_dino
from oracle import navdp as o_navdp
from oracle import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _stats(out, ref):
    d = (out.float().cpu() - ref.float().cpu()).abs()
    return d.mean().item(), d.max().item(), ref.abs().max().item()


def _assert_sampler_output(out, ref, what, max_bound=1.5e-1):
    """waypoint increments in [-1, 1] after 10-20 clipped sampler steps: mean |err| inside the north-star 1e-3, 99th percentile inside 1e-2,
    and a max bound of 1.5e-1. A tight bound on the MAX is the wrong statistic for a chaotic recursion - bf16-autocast PyTorch's own max |err|
    on such a batch has a median of 4.6e-2 over 64 envs and a 90th percentile of 1.4e-1 (profiles/r03E_navdp_bf16_yardstick_cpu.log), and
    any change of summation order moves single elements (r04: 4.0e-2 -> 5.5e-2 on this fixture with the 32-rows-per-wave attention kernel at
    an unchanged mean). The distribution-level statement is tests/test_b64_distribution_gpu.py."""
    d = (out.float().cpu() - ref.float().cpu()).abs().flatten()
    m, p99, mx = d.mean().item(), torch.quantile(d[: 1 << 24], 0.99).item(), d.max().item()
    print(f"{what}: mean|err| {m:.3e} p99 {p99:.3e} max|err| {mx:.3e} ref max {ref.abs().max().item():.2f}")
    assert m < 1e-3 and p99 < 1e-2 and mx < max_bound, (what, m, p99, mx)


def _gold(name):
    from pathlib import Path

    return torch.load(Path(__file__).resolve().parent / "golden" / f"{name}.pt", weights_only=True)


def test_dinov2_encoder_vs_reference_fixture(built_lib):
    from internnav_amd.vit_s import DinoV2Encoder, VitWorkspace

    gold = _gold("dinov2")
    sd = W.materialize(W.dinov2_vits_spec(), seed=gold["seed"])
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(gold["img_seed"]))
    enc = DinoV2Encoder(sd, "", DEV)
    ws = VitWorkspace(2, DEV)
    out = torch.empty(2 * 256, 384, dtype=torch.bfloat16, device=DEV)
    enc.forward(img.permute(0, 2, 3, 1).contiguous().to(DEV), ws, out)
    mean, mx, ref = _stats(out.view(2, 256, 384), gold["tokens"])
    print(f"dinov2: mean|err| {mean:.3e} max|err| {mx:.3e} ref max {ref:.2f}")
    assert mean < 5e-3 and mx < 5e-2     # measured 2.4e-3 / 2.2e-2 on O(1) tokens after 12 blocks (was 6-10x slack)


def test_navdpnet_vs_reference_fixture(built_lib):
    """BASELINE config #2 path: NavDPNet point-goal, 10 DDPM steps, 32 samples/env, critic ranking - B = 2 envs in one call."""
    from internnav_amd.navdp import NavDPNet

    gold = _gold("navdpnet")
    B = gold["B"]
    sd = W.navdpnet_state_dict(seed=gold["seed"])
    inp = {k: v.to(DEV) for k, v in W.navdpnet_inputs(B, seed=gold["seed"]).items()}
    net = NavDPNet(sd, W.NAVDPNET_CFG, DEV, max_envs=B)
    neg, pos = net.predict_pointgoal_batch_action_vel(inp["goal"], inp["images"], inp["depths"], inp["x_init"], inp["step_noise"])
    torch.cuda.synchronize()
    rg = net.cond[: B * net.Lc].view(B, net.Lc, 384)[:, 4:] .float().cpu() - net.cond_pos[4:].cpu()
    m, mx, ref = _stats(rg, gold["rgbd_embed"])
    print(f"rgbd_embed: mean|err| {m:.3e} max|err| {mx:.3e} ref max {ref:.2f}")
    assert m < 2e-2
    S, T = net.S, net.T
    # continuous quantities: the 32 denoised samples per env (waypoint increments in [-1, 1]) and their critic values
    fin = net.sample[: B * S * T].view(B, S, T, 3)
    # this B = 2 fixture is the per-fixture regression guard: measured max 4.0e-2 (16-row attention) / 5.5e-2 (32-row kernel,
    # deferred maximum), so 8e-2 attributes a regression of the wide / deferred path while leaving room for a summation-order change
    _assert_sampler_output(fin, gold["oracle_final"], "final samples", max_bound=8e-2)
    cr = net.critic[: B * S].view(B, S).float().cpu()
    m, mx, ref = _stats(cr, gold["oracle_critic"])
    print(f"critic: mean|err| {m:.3e} max|err| {mx:.3e} ref max {ref:.2f}")
    assert mx < 5e-2 * max(ref, 1.0)
    # ranking (discontinuous): wherever the reference's top-8 / bottom-8 boundary is separated by more than twice our
    # critic error, the selected sets must be identical and the trajectories must match to waypoint tolerance.
    gc = gold["oracle_critic"]
    for b in range(B):
        order = gc[b].argsort()
        for name, out, idx_ref, gap in (("negative", neg, order[:8], gc[b][order[8]] - gc[b][order[7]]),
                                        ("positive", pos, order.flip(0)[:8], gc[b][order[-8]] - gc[b][order[-9]])):
            mine = cr[b].argsort()[:8] if name == "negative" else (-cr[b]).argsort()[:8]
            if gap > 2 * mx:
                assert set(mine.tolist()) == set(idx_ref.tolist()), f"env {b} {name}: selected set differs"
            if torch.equal(mine, idx_ref):
                m2, mx2, _ = _stats(out[b], gold[name][b])
                print(f"env {b} {name}: same ranking, trajectory mean|err| {m2:.3e} max|err| {mx2:.3e}")
                assert m2 < 5e-3 and mx2 < 1e-1  # cumulative sums of 24 waypoints
    # our own selection is exactly consistent with our own critic values and samples
    traj = torch.cumsum(fin.float().cpu() / 4.0, dim=2)
    for b in range(B):
        assert torch.allclose(neg[b].cpu(), traj[b][cr[b].argsort()[:8]], atol=1e-5)
        assert torch.allclose(pos[b].cpu(), traj[b][(-cr[b]).argsort()[:8]], atol=1e-5)


def test_navdpnet_nogoal_vs_reference_fixture(built_lib):
    """the zero-goal sibling of the config-#2 entry point (navdp_policy.py:323-339): fixture from the reference's own
    `predict_nogoal_batch_action_vel`; same sampler / critic / ranking checks as the point-goal test, B = 2 envs in one call."""
    from internnav_amd.navdp import NavDPNet

    gold = _gold("navdpnet_nogoal")
    B = gold["B"]
    sd = W.navdpnet_state_dict(seed=gold["seed"])
    inp = {k: v.to(DEV) for k, v in W.navdpnet_inputs(B, seed=gold["seed"]).items()}
    net = NavDPNet(sd, W.NAVDPNET_CFG, DEV, max_envs=B)
    neg, pos = net.predict_nogoal_batch_action_vel(inp["images"], inp["depths"], inp["x_init"], inp["step_noise"])
    torch.cuda.synchronize()
    S, T = net.S, net.T
    fin = net.sample[: B * S * T].view(B, S, T, 3)
    _assert_sampler_output(fin, gold["oracle_final"], "nogoal final samples", max_bound=8e-2)
    cr = net.critic[: B * S].view(B, S).float().cpu()
    m, mx, ref = _stats(cr, gold["oracle_critic"])
    print(f"nogoal critic: mean|err| {m:.3e} max|err| {mx:.3e} ref max {ref:.2f}")
    assert mx < 5e-2 * max(ref, 1.0)
    gc = gold["oracle_critic"]
    for b in range(B):
        order = gc[b].argsort()
        for name, out, idx_ref, gap in (("negative", neg, order[:8], gc[b][order[8]] - gc[b][order[7]]),
                                        ("positive", pos, order.flip(0)[:8], gc[b][order[-8]] - gc[b][order[-9]])):
            mine = cr[b].argsort()[:8] if name == "negative" else (-cr[b]).argsort()[:8]
            if gap > 2 * mx:
                assert set(mine.tolist()) == set(idx_ref.tolist()), f"env {b} {name}: selected set differs"
            if torch.equal(mine, idx_ref):
                m2, mx2, _ = _stats(out[b], gold[name][b])
                assert m2 < 5e-3 and mx2 < 1e-1
    # and the point-goal call on the same engine afterwards is unaffected by the table-filled goal slots
    neg2, _ = net.predict_pointgoal_batch_action_vel(inp["goal"], inp["images"], inp["depths"], inp["x_init"], inp["step_noise"])
    torch.cuda.synchronize()
    _assert_sampler_output(net.sample[: B * S * T].view(B, S, T, 3), _gold("navdpnet")["oracle_final"], "point goal after nogoal", max_bound=8e-2)


def test_n1_navdp_head_vs_reference_fixture(built_lib):
    from internnav_amd.navdp import NavDPPolicyDAT

    gold = _gold("n1_navdp")
    B = gold["B"]
    sd = W.n1_navdp_state_dict(seed=gold["seed"])
    inp = W.n1_navdp_inputs(B, seed=gold["seed"])
    net = NavDPPolicyDAT(sd, W.N1_NAVDP_CFG, DEV, max_envs=B)
    out = net.predict_pointgoal_action_async(inp["vlm_tokens"].to(DEV, torch.bfloat16), inp["images"].to(DEV), inp["depths"].to(DEV),
                                             inp["x_init"].to(DEV), inp["step_noise"].to(DEV))
    _assert_sampler_output(out, gold["trajectories"], "n1 navdp trajectories")


def test_n1_navdp_plain_head_vs_reference_fixture(built_lib):
    """the non-async 'navdp' System-1 type: NavDP_Policy_DPT_CriticSum_DAT.predict_pointgoal_action (internvla_n1/navdp.py:255-289), fixture
    from the reference module itself; the engine is built from a state dict WITHOUT the RGB-D / goal-compressor modules."""
    from internnav_amd.navdp import NavDPPolicyDAT

    gold = _gold("n1_navdp")
    B = gold["B"]
    sd = {k: v for k, v in W.n1_navdp_state_dict(seed=gold["seed"]).items() if not k.startswith(("rgbd_encoder.", "goal_compressor."))}
    inp = W.n1_navdp_inputs(B, seed=gold["seed"])
    net = NavDPPolicyDAT(sd, W.N1_NAVDP_CFG, DEV, max_envs=B, use_async=False)
    out = net.predict_pointgoal_action(inp["vlm_tokens"].to(DEV, torch.bfloat16), inp["x_init"].to(DEV), inp["step_noise"].to(DEV))
    _assert_sampler_output(out, gold["trajectories_plain"], "n1 navdp (non-async) trajectories")
    assert (gold["trajectories_plain"] - gold["trajectories"]).abs().max().item() > 1e-2


def test_navdpnet_batch_invariance(built_lib):
    """env b of a B = 3 call == the same env run alone (the reference's batch-1 semantics hold per env in the batched engine)."""
    from internnav_amd.navdp import NavDPNet

    sd = W.navdpnet_state_dict(seed=5)
    inp = {k: v.to(DEV) for k, v in W.navdpnet_inputs(3, seed=5).items()}
    net = NavDPNet(sd, W.NAVDPNET_CFG, DEV, max_envs=3)
    S, T = net.S, net.T
    neg3, pos3 = net.predict_pointgoal_batch_action_vel(inp["goal"], inp["images"], inp["depths"], inp["x_init"], inp["step_noise"])
    neg3, pos3 = neg3.clone(), pos3.clone()
    fin3 = net.sample[: 3 * S * T].view(3, S, T, 3).clone()
    cr3 = net.critic[: 3 * S].view(3, S).clone()
    b = 1
    neg1, pos1 = net.predict_pointgoal_batch_action_vel(inp["goal"][b:b + 1].contiguous(), inp["images"][b:b + 1].contiguous(),
                                                        inp["depths"][b:b + 1].contiguous(), inp["x_init"][b:b + 1].contiguous(),
                                                        inp["step_noise"][:, b:b + 1].contiguous())
    fin1, cr1 = net.sample[: S * T].view(S, T, 3), net.critic[:S]
    # tile-kernel selection depends on the row count, so the two runs differ in fp32 accumulation order: the continuous quantities
    # (denoised samples, critic values) must agree to rounding level; the ranked outputs are compared whenever the ranking agrees.
    assert (fin3[b] - fin1).abs().max().item() < 2e-2 and (cr3[b] - cr1).abs().max().item() < 2e-2
    if torch.equal(cr3[b].argsort()[:8], cr1.argsort()[:8]):
        assert (neg3[b] - neg1[0]).abs().max().item() < 5e-2
    if torch.equal((-cr3[b]).argsort()[:8], (-cr1).argsort()[:8]):
        assert (pos3[b] - pos1[0]).abs().max().item() < 5e-2


def test_navdpnet_from_pretrained_reference_loader(built_lib, tmp_path):
    """get_policy('NavDP_Policy') -> NavDPNet.from_pretrained(path, config=NavDPModelConfig(model_cfg={'il': ..., 'local_rank': 0})) as the
    reference loads it (navdp_policy.py:36-64): a state-dict file on disk, hyper-parameters from model_cfg['il']; same outputs as the
    engine built from the in-memory state dict."""
    import internnav_amd

    cls, cfg_cls = internnav_amd.get_policy("NavDP_Policy"), internnav_amd.get_config("NavDP_Policy")
    sd = W.navdpnet_state_dict(seed=3)
    torch.save(sd, tmp_path / "navdp.ckpt")
    il = dict(image_size=224, memory_size=8, predict_size=24, temporal_depth=16, heads=8, token_dim=384, pixel_channel=4, channels=3, dropout=0.1,
              scratch=False, finetune=False)
    net = cls.from_pretrained(str(tmp_path / "navdp.ckpt"), config=cfg_cls(model_cfg={"model": {}, "il": il, "local_rank": 0}), max_envs=2)
    ref = cls(sd, W.NAVDPNET_CFG, DEV, max_envs=2)
    inp = {k: v.to(DEV) for k, v in W.navdpnet_inputs(2, seed=3).items()}
    a = net.eval().predict_pointgoal_batch_action_vel(inp["goal"], inp["images"], inp["depths"], inp["x_init"], inp["step_noise"])
    b = ref.predict_pointgoal_batch_action_vel(inp["goal"], inp["images"], inp["depths"], inp["x_init"], inp["step_noise"])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    with pytest.raises(KeyError):
        torch.save({k: v for k, v in sd.items() if not k.startswith("critic_head")}, tmp_path / "bad.ckpt")
        cls.from_pretrained(str(tmp_path / "bad.ckpt"), config=cfg_cls(model_cfg={"il": il, "local_rank": 0}))
