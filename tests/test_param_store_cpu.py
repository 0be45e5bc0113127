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
