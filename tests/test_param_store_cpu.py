"""CPU: the flat parameter store of the SFT step (host layout logic only - no kernels): alignment, views, padding, frozen stores."""
import torch


def test_flat_store_layout_and_views():
    from internnav_amd.sft import ParamStore

    g = torch.Generator().manual_seed(0)
    tensors = {"a.weight": torch.randn(384, 3, generator=g), "a.bias": torch.randn(384, generator=g), "conv": torch.randn(8, 3, 14, 14, generator=g),
               "latent_queries": torch.randn(1, 4, 64, generator=g)}
    P = ParamStore(tensors, "cpu")
    assert P.numel % 1024 == 0 and P.p32.numel() == P.g32.numel() == P.m.numel() == P.v.numel() == P.p16.numel() == P.numel
    prev_end = 0
    for k, t in tensors.items():
        off, shape = P.index[k]
        assert off % 8 == 0 and off >= prev_end and shape == tuple(t.shape)        # 16-byte aligned bf16 views for the GEMM operands
        prev_end = off + t.numel()
        assert torch.equal(P.w32(k), t) and torch.equal(P.w16(k), t.bfloat16())
        assert P.w32(k).data_ptr() == P.p32.data_ptr() + 4 * off and P.grad(k).shape == t.shape
        assert P.w32(k).is_contiguous() and P.w16(k).is_contiguous()
    # the gaps and the tail are zero: they take part in the fused update and in the gradient norm without changing either
    used = torch.zeros(P.numel, dtype=torch.bool)
    for k, t in tensors.items():
        used[P.index[k][0]: P.index[k][0] + t.numel()] = True
    assert float(P.p32[~used].abs().sum()) == 0.0 and float(P.g32.abs().sum()) == 0.0
    P.grad("a.bias").add_(1.0)
    assert float(P.g32.sum()) == 384.0
    P.zero_grad()
    assert float(P.g32.abs().sum()) == 0.0
    sd = P.state_dict()
    assert set(sd) == set(tensors) and all(torch.equal(sd[k], tensors[k]) for k in tensors)


def test_frozen_store_has_no_optimizer_state():
    from internnav_amd.sft import ParamStore, _Params

    t = {"rgb_model.w": torch.ones(16, 8)}
    F = ParamStore(t, "cpu", trainable=False)
    assert not hasattr(F, "g32") and not hasattr(F, "m") and torch.equal(F.w16("rgb_model.w"), torch.ones(16, 8, dtype=torch.bfloat16))
    P = ParamStore({"head.w": torch.zeros(8, 8)}, "cpu")
    names = _Params(P, F)
    assert names.trains("head.w") and not names.trains("rgb_model.w")
    assert names.w32("rgb_model.w").data_ptr() == F.p32.data_ptr() and names.grad("head.w").data_ptr() == P.g32.data_ptr()


def test_checkpoint_resume_roundtrip(tmp_path):
    """trainer.save_checkpoint / load_checkpoint: master weights, Adam moments and the optimiser / schedule / mask counters survive, the
    engine's latent queries follow, and a mismatching trainable set is refused."""
    import pytest

    from internnav_amd.sft import ParamStore
    from internnav_amd.trainer import LQ, InternVLAN1SftTrainer

    def make(seed):
        g = torch.Generator().manual_seed(seed)
        P = ParamStore({"traj_dit.w": torch.randn(16, 8, generator=g), "cond_projector.0.bias": torch.randn(24, generator=g),
                        LQ: torch.randn(1, 4, 32, generator=g)}, "cpu")
        tr = object.__new__(InternVLAN1SftTrainer)
        tr.P, tr.system1, tr.step_idx, tr.micro_idx = P, "nextdit_async", 0, 0

        class _E:
            latent_q = torch.zeros(4, 32, dtype=torch.bfloat16)
        tr.engine = _E()
        return tr

    a = make(1)
    a.P.m.normal_()
    a.P.v.uniform_()
    a.P.step_count, a.step_idx, a.micro_idx = 7, 7, 21
    f = tmp_path / "ck.pt"
    a.save_checkpoint(str(f))
    b = make(2)
    b.load_checkpoint(str(f))
    for k in a.P.index:
        assert torch.equal(a.P.w32(k), b.P.w32(k)) and torch.equal(a.P._view(a.P.m, k), b.P._view(b.P.m, k)) and torch.equal(a.P._view(a.P.v, k), b.P._view(b.P.v, k))
    assert torch.equal(b.P.p16, b.P.p32.bfloat16()) and (b.P.step_count, b.step_idx, b.micro_idx) == (7, 7, 21)
    assert torch.equal(b.engine.latent_q, a.P.w16(LQ).view(4, 32))
    sd = a.state_dict()
    assert set(sd) == {"model.traj_dit.w", "model.cond_projector.0.bias", "model.latent_queries"}
    b.system1 = "navdp_async"
    with pytest.raises(ValueError):
        b.load_checkpoint(str(f))
    c = make(3)
    c.P = ParamStore({"other": torch.zeros(8)}, "cpu")
    with pytest.raises(KeyError):
        c.load_checkpoint(str(f))
