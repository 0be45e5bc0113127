"""SFT step, `navdp_async` branch (internvla_n1.py:287-303 -> NavDP_Policy_DPT_CriticSum_DAT.forward_vlm_traj): loss and every gradient of
the HIP tape against torch autograd of the fp32 oracle (oracle/sft.navdp_sft_loss, pinned by tests/golden/sft_navdp.pt to autograd through
the reference's own NavDP module), bf16-autocast autograd of the same functions as the yardstick."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def _inputs(B, T, seed=1):
    g = torch.Generator().manual_seed(seed)
    return dict(hidden_q=torch.randn(B, 4, 3584, generator=g).bfloat16().float(), traj_images=torch.rand(B, T, 224, 224, 3, generator=g),
                traj_depths=torch.rand(B, T, 224, 224, generator=g) * 5.0, traj_poses=torch.randn(B, T, 32, 3, generator=g),
                video_frame_num=torch.tensor([T] + [max(1, T - 1)] * (B - 1)), noise=torch.randn(B * T, 32, 3, generator=g),
                timesteps=torch.randint(0, 20, (B * T,), generator=g))


def _oracle(sd0, inp, cfg, autocast):
    from oracle import sft as O

    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    hq = inp["hidden_q"].clone().requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        loss = O.navdp_sft_loss(sd, hq, inp["traj_images"], inp["traj_depths"], inp["traj_poses"], inp["video_frame_num"], inp["noise"],
                                inp["timesteps"], cfg)
    loss.backward()
    return loss.item(), hq.grad.float(), {k: v.grad.float() for k, v in sd.items() if v.grad is not None}


def test_navdp_sft_loss_and_gradients(built_lib):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from internnav_amd import sft as E
    from internnav_amd import synthetic as S

    dev = torch.device("cuda:0")
    cfg = S.N1_NAVDP_CFG
    sd0 = {k: v.float() for k, v in S.materialize(S.n1_navdp_spec(), 3).items()}
    inp = _inputs(2, 2)
    l32, dh32, g32 = _oracle(sd0, inp, cfg, False)
    l16, dh16, g16 = _oracle(sd0, inp, cfg, True)
    head = E.NavDPSftHead(sd0, dev, cfg)
    loss, dh = head.loss_and_grads(inp["hidden_q"].to(dev), inp["traj_images"].to(dev), inp["traj_depths"].to(dev), inp["traj_poses"],
                                   inp["video_frame_num"], inp["noise"], inp["timesteps"])
    assert abs(loss.item() - l32) <= max(2 * abs(l16 - l32), 3e-3 * abs(l32)), (loss.item(), l32, l16)
    e, y = _rel(dh.float().cpu().view_as(dh32), dh32), _rel(dh16, dh32)
    print(f"loss {loss.item():.5f} / {l32:.5f} (bf16 {l16:.5f}); d hidden: engine {e:.3e} vs bf16 {y:.3e}")
    assert e <= 1.25 * y + 1e-3
    scale = max(g.norm().item() for g in g32.values())
    errs, yards, bad = [], [], []
    for k, ref in g32.items():
        if "rgb_model" in k:
            assert k not in head.P.index and k in head.F.index           # frozen (internvla_n1_trainer.py:119-120): no gradient, no moments
            continue
        assert k in head.P.index, k
        got = head.P.grad(k).cpu().view_as(ref)
        if ref.norm().item() < 1e-6 * scale:
            assert got.norm().item() < 1e-4 * scale, k
            continue
        e, y = _rel(got, ref), _rel(g16[k], ref)
        errs.append(e)
        yards.append(y)
        if e > 1.5 * y + 1e-3:
            bad.append((k, e, y))
    print(f"{len(errs)} parameter gradients: engine mean {sum(errs) / len(errs):.3e}, bf16 PyTorch mean {sum(yards) / len(yards):.3e}")
    # Round 4: with the bf16 ImageNet constants of DAT_RGBD_Patch_Backbone corrected (two hand-typed entries were wrong: the RGB tokens sat 2x
    # further from fp32 than bf16 PyTorch's, and all 16 decoder layers re-read them) and fp32 post-LN streams in the former, this branch is
    # where the NextDiT branch is: closer to fp32 than bf16-autocast PyTorch (CPU replica of the tape: gradients 5.2e-3 vs 6.6e-3 on average,
    # worst single tensor 1.10x its yardstick; on the GPU two of 523 tensors - LayerNorm biases, column sums over all rows - reach 1.4x).
    # Bound: never above 1.5x the yardstick per tensor, not above it on average.
    assert not bad, bad[:10]
    assert sum(errs) / len(errs) <= sum(yards) / len(yards)


def test_navdp_sft_per_layer_drift_table(built_lib):
    """Where the navdp_async branch sits against its yardstick, layer by layer: the residual stream of the 16-layer
    decoder after every layer and its GRADIENT on the way back, engine vs fp32 oracle next to bf16-autocast PyTorch vs the same oracle.
    Written to gpurun_out/sft_navdp_drift.txt. Asserted (round 4): at EVERY layer the engine is not further from fp32 than bf16-autocast
    PyTorch, forward stream and stream gradient alike (measured 0.67-0.77x / 0.75-0.84x)."""
    import os
    from pathlib import Path

    import oracle.navdp as ON
    import oracle.sft as O
    from internnav_amd import sft as E
    from internnav_amd import synthetic as S

    dev = torch.device("cuda:0")
    cfg = S.N1_NAVDP_CFG
    sd0 = {k: v.float() for k, v in S.materialize(S.n1_navdp_spec(), 3).items()}
    inp = _inputs(2, 2)
    rec = {}
    orig = ON.decoder_layer

    def tapped(x, mem, sd, p, *a, **k):
        y = orig(x, mem, sd, p, *a, **k)
        if p.startswith("decoder.layers."):
            y.retain_grad()
            rec[f"layer{p.split('.')[-1]}"] = y
        return y

    def run(autocast):
        rec.clear()
        ON.decoder_layer = tapped
        try:
            sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
            with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
                loss = O.navdp_sft_loss(sd, inp["hidden_q"].clone().requires_grad_(True), inp["traj_images"], inp["traj_depths"], inp["traj_poses"],
                                        inp["video_frame_num"], inp["noise"], inp["timesteps"], cfg)
            loss.backward()
        finally:
            ON.decoder_layer = orig
        return {k: (v.detach().float().clone(), v.grad.float().clone()) for k, v in rec.items()}

    r32, r16 = run(False), run(True)
    head = E.NavDPSftHead(sd0, dev, cfg)
    head.taps = []
    head.loss_and_grads(inp["hidden_q"].to(dev), inp["traj_images"].to(dev), inp["traj_depths"].to(dev), inp["traj_poses"], inp["video_frame_num"],
                        inp["noise"], inp["timesteps"])
    fwd, bwd = dict(head.taps), dict(head.grad_taps)
    lines = ["# navdp_async SFT: decoder residual stream per layer, engine vs fp32 oracle | bf16-autocast PyTorch vs fp32 oracle (relative L2)",
             "# layer | forward: engine  autocast  ratio | gradient of the stream: engine  autocast  ratio"]
    worst_f = worst_b = 0.0
    for i in range(cfg["temporal_depth"]):
        n = f"layer{i}"
        ef, yf = _rel(fwd[n].cpu().view_as(r32[n][0]), r32[n][0]), _rel(r16[n][0], r32[n][0])
        eb, yb = _rel(bwd[n].cpu().view_as(r32[n][1]), r32[n][1]), _rel(r16[n][1], r32[n][1])
        worst_f, worst_b = max(worst_f, ef / yf), max(worst_b, eb / yb)
        lines.append(f"{n:8s} {ef:.3e} {yf:.3e} {ef / yf:5.2f} | {eb:.3e} {yb:.3e} {eb / yb:5.2f}")
    out = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parent.parent)) / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        (out / "sft_navdp_drift.txt").write_text("\n".join(lines) + "\n")
    except OSError:
        pass
    print("\n".join(lines))
    assert worst_f <= 1.0 and worst_b <= 1.0, (worst_f, worst_b)
