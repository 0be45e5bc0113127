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
        tr.rank, tr.world, tr.device = 0, 1, torch.device("cpu")
        tr.gen_dev, tr.gen_cpu = torch.Generator().manual_seed(seed), torch.Generator().manual_seed(seed + 100)

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
    # the noise / time-step streams continue where the saved run stood (the global torch RNG was neither saved nor restored)
    assert torch.equal(torch.randn(5, generator=a.gen_dev), torch.randn(5, generator=b.gen_dev))
    assert torch.equal(torch.rand(5, generator=a.gen_cpu), torch.rand(5, generator=b.gen_cpu))
    sd = a.state_dict()
    assert set(sd) == {"model.traj_dit.w", "model.cond_projector.0.bias", "model.latent_queries"}
    b.system1 = "navdp_async"
    with pytest.raises(ValueError):
        b.load_checkpoint(str(f))
    c = make(3)
    c.P = ParamStore({"other": torch.zeros(8)}, "cpu")
    with pytest.raises(KeyError):
        c.load_checkpoint(str(f))


def test_trainer_from_pretrained_checkpoint_layout(tmp_path, monkeypatch):
    """InternVLAN1SftTrainer.from_pretrained on an on-disk checkpoint in the reference's layout: the Qwen2.5-VL keys go to the (here stubbed)
    engine, every System-1 tensor + latent_queries lands in the trainable store with the checkpoint's values, and state_dict() maps the
    names back. (The engine itself needs the GPU; its checkpoint loading is covered by tests/test_agent_gpu.py.)"""
    from internnav_amd import synthetic as S
    from internnav_amd import trainer as TR

    cfg = dict(S.QWEN_TEST_CFG, v_depth=1, v_fullatt=(0,), t_layers=1, vocab=64, image_token_id=50, traj_token_id=51, vision_start_id=52,
               vision_end_id=53, eos_token_id=54)
    sd_disk = S.write_checkpoint(tmp_path / "ck", cfg, "nextdit_async", seed=5, shards=3)

    class _Engine:
        def __init__(self, weights, qcfg, device, max_seqs, max_seq_len, max_patches):
            assert qcfg["t_layers"] == 1 and qcfg["vocab"] == 64 and "model.layers.0.self_attn.q_proj.weight" in weights
            self.latent_q = weights["model.latent_queries"].reshape(-1, qcfg["t_hidden"]).to(torch.bfloat16)
            self.args = (max_seqs, max_seq_len, max_patches)

    monkeypatch.setattr(TR, "QwenVLEngine", _Engine)
    tr = TR.InternVLAN1SftTrainer.from_pretrained(tmp_path / "ck", device="cpu", max_seqs=2, max_seq_len=256, max_patches=1568, total_steps=10, dropout=0.0)
    assert tr.engine.args == (2, 256, 1568) and tr.system1 == "nextdit_async"
    s1_keys = {k for k in sd_disk if k.startswith("model.") and k[6:].startswith(("action_", "traj_dit", "cond_projector", "memory_encoder", "rgb_resampler", "rgb_model"))
               and not k.endswith("mask_token")}
    got = tr.state_dict()
    assert set(got) == s1_keys | {"model.latent_queries"}
    for k in got:
        assert torch.equal(got[k], sd_disk[k].float().view_as(got[k])), k
    assert not any(k.startswith(("layers.", "embed_tokens", "norm.")) for k in tr.P.index)        # nothing of the frozen LLM is trainable
