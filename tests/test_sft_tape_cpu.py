"""CPU: the WIRING of the SFT tape (internnav_amd/sft.py) with torch stand-ins for the HIP kernel wrappers (tests/_cpu_kernels.py): every
forward op and every backward closure of the NextDiT and NavDP loss graphs, against torch autograd of the fp32 oracle. The kernels themselves
and the real step are tested on the GPU (tests/test_train_ops_gpu.py, tests/test_sft_gpu.py, tests/test_sft_navdp_gpu.py)."""
import pytest
import torch

from tests import _cpu_kernels as K


def _rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def _check(head, ref_grads, skip=(), tol=4e-2):
    scale = max(g.norm().item() for g in ref_grads.values())
    worst = ("", 0.0)
    n = 0
    for k, ref in ref_grads.items():
        if any(s in k for s in skip):
            continue
        assert k in head.P.index, k
        got = head.P.grad(k).view_as(ref)
        if ref.norm().item() < 1e-6 * scale:
            assert got.norm().item() < 1e-4 * scale, k
            continue
        e = _rel(got, ref)
        n += 1
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < tol, worst          # bf16 activations between the stand-in ops, like the engine
    return n


def test_nextdit_tape_wiring(monkeypatch):
    from internnav_amd import sft as E
    from internnav_amd import synthetic as S
    from oracle import sft as O

    K.install(monkeypatch)
    sd0 = {k: v.float() for k, v in S.materialize(S.n1_nextdit_spec(), 3).items()}
    g = torch.Generator().manual_seed(0)
    B, T = 1, 2
    inp = dict(hq=torch.randn(B, 4, 3584, generator=g).bfloat16().float(), img=torch.rand(B, T, 224, 224, 3, generator=g),
               poses=torch.randn(B, T, 32, 3, generator=g), vfn=torch.tensor([T]), noise=torch.randn(B * T, 32, 3, generator=g),
               ti=torch.randint(0, 1000, (B * T,), generator=g))
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    hq = inp["hq"].clone().requires_grad_(True)
    loss = O.nextdit_sft_loss(sd, hq, inp["img"], inp["poses"], inp["vfn"], inp["noise"], inp["ti"])
    loss.backward()
    head = E.NextDiTSftHead(sd0, "cpu")
    l2, dh = head.loss_and_grads(inp["hq"], inp["img"], inp["poses"], inp["vfn"], inp["noise"], inp["ti"])
    assert abs(l2.item() - loss.item()) < 1e-2 * abs(loss.item())
    assert _rel(dh.float().view_as(hq.grad), hq.grad) < 4e-2
    assert _check(head, {k: v.grad for k, v in sd.items() if v.grad is not None}) > 550
    # gradient accumulation: two half-weighted passes over the same micro-batch add up to the full gradient (HF Trainer's loss / k)
    full = head.P.g32.clone()
    head.P.zero_grad()
    for _ in range(2):
        l_half, dh_half = head.loss_and_grads(inp["hq"], inp["img"], inp["poses"], inp["vfn"], inp["noise"], inp["ti"], loss_scale=0.5)
    assert abs(l_half.item() - l2.item()) < 1e-6 and _rel(dh_half.float(), 0.5 * dh.float()) < 2e-2      # the reported loss is unscaled
    assert _rel(head.P.g32, full) < 1e-2
    # one fused optimiser step through the store moves every tensor that had a gradient and resets the gradients
    before = head.P.p32.clone()
    head.P.adamw_step(1e-3)
    assert float(head.P.g32.abs().max()) == 0.0 and float((head.P.p32 - before).abs().max()) > 0 and torch.equal(head.P.p16, head.P.p32.bfloat16())


def test_plain_nextdit_tape_wiring_against_the_reference_fixture(monkeypatch):
    """system1 = 'nextdit' (internvla_n1.py:256-258, VERDICT r4 missing #1): the condition is the projected trajectory hidden states alone.
    The tape (stand-in kernels) against tests/golden/sft_nextdit_plain.pt = loss / gradients back-propagated through the REFERENCE's own
    NextDiTCrossAttn for a model without rgb_model / memory_encoder / rgb_resampler; the trainable store holds none of those tensors."""
    from pathlib import Path

    from internnav_amd import sft as E
    from internnav_amd import synthetic as S

    K.install(monkeypatch)
    gold = torch.load(Path(__file__).resolve().parent / "golden" / "sft_nextdit_plain.pt", weights_only=True)
    sd0 = {k: v.float() for k, v in S.materialize(S.n1_nextdit_spec(), gold["weights_seed"]).items()}     # the async modules are present in the dict, and ignored
    inp = gold["inputs"]
    head = E.NextDiTSftHead(sd0, "cpu", use_async=False, dropout=0.1)
    assert not any(k.startswith(E.S1_ASYNC_ONLY_PREFIXES) for k in head.P.index) and head.dino is None
    loss, dh = head.loss_and_grads(inp["hidden_q"], inp["traj_images"], inp["traj_poses"], inp["video_frame_num"], inp["noise"], inp["t_index"])
    assert head.last_dropout_sites == gold["dropout_sites"] == {"dropout": 0, "attention": 0}          # no nn.Transformer layer in this branch
    assert abs(loss.item() - gold["loss"]) < 1e-2 * abs(gold["loss"])
    assert _rel(dh.float().view_as(gold["d_hidden"]), gold["d_hidden"]) < 4e-2
    gscale = max(g["norm"] for g in gold["grads"].values())
    n = 0
    for k, g in gold["grads"].items():
        assert k in head.P.index, k
        mine = head.P.grad(k).flatten()
        if g["norm"] < 1e-6 * gscale:
            assert mine.norm().item() < 1e-4 * gscale, k
            continue
        assert abs(mine.norm().item() - g["norm"]) < 4e-2 * g["norm"], (k, mine.norm().item(), g["norm"])
        n += 1
    assert n > 150 and set(head.P.index) >= set(gold["grads"])


def test_navdp_tape_wiring(monkeypatch):
    from internnav_amd import sft as E
    from internnav_amd import synthetic as S
    from oracle import sft as O

    K.install(monkeypatch)
    cfg = dict(S.N1_NAVDP_CFG, temporal_depth=2)         # two of the 16 decoder layers: the wiring is the same, the CPU time is not
    spec = {k: v for k, v in S.n1_navdp_spec().items() if not k.startswith("decoder.layers.") or int(k.split(".")[2]) < 2}
    sd0 = {k: v.float() for k, v in S.materialize(spec, 3).items()}
    g = torch.Generator().manual_seed(1)
    B, T = 1, 2
    inp = dict(hq=torch.randn(B, 4, 3584, generator=g).bfloat16().float(), img=torch.rand(B, T, 224, 224, 3, generator=g),
               dep=torch.rand(B, T, 224, 224, generator=g) * 5.0, poses=torch.randn(B, T, 32, 3, generator=g), vfn=torch.tensor([T]),
               noise=torch.randn(B * T, 32, 3, generator=g), ts=torch.randint(0, 20, (B * T,), generator=g))
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    hq = inp["hq"].clone().requires_grad_(True)
    loss = O.navdp_sft_loss(sd, hq, inp["img"], inp["dep"], inp["poses"], inp["vfn"], inp["noise"], inp["ts"], cfg)
    loss.backward()
    head = E.NavDPSftHead(sd0, "cpu", cfg)
    l2, dh = head.loss_and_grads(inp["hq"], inp["img"], inp["dep"], inp["poses"], inp["vfn"], inp["noise"], inp["ts"])
    assert abs(l2.item() - loss.item()) < 1e-2 * abs(loss.item())
    assert _rel(dh.float().view_as(hq.grad), hq.grad) < 8e-2     # bf16 PyTorch itself shows 3-5 % here (three ReLU / bf16 layers from 3584 to 384)
    assert not any("rgb_model" in k for k in head.P.index) and any("rgb_model" in k for k in head.F.index)
    assert _check(head, {k: v.grad for k, v in sd.items() if v.grad is not None}, skip=("rgb_model",), tol=8e-2) > 150


def test_dropout_sites_match_the_reference_modules(monkeypatch):
    """train mode: the tape draws one mask per ACTIVE dropout site of the reference's modules - nn.Dropout with p > 0 and
    nn.MultiheadAttention built with dropout > 0 - counted by forward hooks on the reference modules in train() mode
    (oracle/make_golden.py _count_dropout_sites -> tests/golden/sft.pt / sft_navdp.pt). ADVICE r2: TokenCompressor.cross_attention is an
    nn.MultiheadAttention with the default dropout 0.0 (encoder/navdp_backbone.py:77) and must NOT get a mask."""
    from pathlib import Path

    from internnav_amd import sft as E
    from internnav_amd import synthetic as S

    K.install(monkeypatch)
    monkeypatch.setattr(K, "COUNT_ONLY", True)
    gold = Path(__file__).resolve().parent / "golden"
    ref_nd = torch.load(gold / "sft.pt", weights_only=True)["dropout_sites"]
    ref_nv = torch.load(gold / "sft_navdp.pt", weights_only=True)["dropout_sites"]
    g = torch.Generator().manual_seed(0)
    B, T = 1, 1
    hq, img = torch.randn(B, 4, 3584, generator=g), torch.rand(B, T, 224, 224, 3, generator=g)
    poses, noise = torch.randn(B, T, 32, 3, generator=g), torch.randn(B * T, 32, 3, generator=g)
    head = E.NextDiTSftHead({k: v.float() for k, v in S.materialize(S.n1_nextdit_spec(), 3).items()}, "cpu", dropout=0.1)
    head.loss_and_grads(hq, img, poses, torch.tensor([T]), noise, torch.tensor([500]), seed=1)
    assert head.last_dropout_sites == ref_nd == {"dropout": 21, "attention": 9}
    depth = 2                                            # 2 of the 16 decoder layers (CPU time); each layer = 2 attention + 4 nn.Dropout sites
    cfg = dict(S.N1_NAVDP_CFG, temporal_depth=depth)
    spec = {k: v for k, v in S.n1_navdp_spec().items() if not k.startswith("decoder.layers.") or int(k.split(".")[2]) < depth}
    head = E.NavDPSftHead({k: v.float() for k, v in S.materialize(spec, 3).items()}, "cpu", cfg, dropout=0.1)
    head.loss_and_grads(hq, img, torch.rand(B, T, 224, 224, generator=g) * 5.0, poses, torch.tensor([T]), noise, torch.tensor([7]), seed=1)
    skipped = S.N1_NAVDP_CFG["temporal_depth"] - depth
    assert head.last_dropout_sites == {"dropout": ref_nv["dropout"] - 4 * skipped, "attention": ref_nv["attention"] - 2 * skipped}
    # eval-mode gradients (what the parity tests compare) draw nothing
    head = E.NextDiTSftHead({k: v.float() for k, v in S.materialize(S.n1_nextdit_spec(), 3).items()}, "cpu", dropout=0.0)
    head.loss_and_grads(hq, img, poses, torch.tensor([T]), noise, torch.tensor([500]))
    assert head.last_dropout_sites == {"dropout": 0, "attention": 0}
