"""GPU parity of the SFT step (BASELINE config #5) at its OWN depth: 32 ViT blocks + 28 decoder layers, micro-batch 2,
10 frames + text = 2 108 prompt tokens (ragged: the second sample is 21 tokens shorter) - the `bench_sft.py` shape.

Fixture tests/golden/sft_full.pt (oracle/make_golden_sft_full.py, hash-seeded weights = the same bits on this GPU): fp32 oracle loss,
hidden states of the latent-query rows, their residual stream after EVERY decoder layer, d loss / d latent_queries through the 28 frozen
layers, sampled entries of every System-1 parameter gradient - and the same quantities under bf16 autocast (the precision the reference
trains in) as the yardstick. Bars: the engine is not further from fp32 than bf16 PyTorch is (x1.25 slack on gradients, as in the
2-layer tests), at every layer of the drift table."""
import os
from pathlib import Path

import pytest
import torch

from oracle import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = Path(__file__).resolve().parent / "golden" / "sft_full.pt"


def _rel(a, b):
    return ((a - b).norm() / b.norm()).item()


def _report(lines):
    out = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parent.parent)) / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        (out / "sft_full_drift.txt").write_text("\n".join(lines) + "\n")
    except OSError:
        pass
    print("\n".join(lines))


def test_training_step_at_full_depth(built_lib):
    from internnav_amd.qwen_vl import QwenVLEngine
    from internnav_amd.trainer import LQ, InternVLAN1SftTrainer
    from oracle.make_golden_sft_full import make_batch

    gold = torch.load(GOLD, weights_only=True)
    cfg = W.QWEN_N1_CFG
    inp, lens, traj_images, traj_poses, vfn, noise, t_index = make_batch(cfg)
    assert lens == gold["lens"] and inp["input_ids"].shape[1] == 2108
    B, nq = gold["B"], cfg["n_query"]
    hw = W.HashWeights(W.qwen_spec(cfg), gold["seed"], DEV)
    eng = QwenVLEngine(hw, cfg, DEV, max_seqs=B, max_seq_len=2176, max_patches=inp["pixel_values"].shape[0])
    sd_s = {k: v.float() for k, v in W.materialize(W.n1_nextdit_spec(), gold["seed"]).items()}
    tr = InternVLAN1SftTrainer(eng, sd_s, DEV, total_steps=100, dropout=0.0)          # eval-mode gradients: what the oracle computes
    S = inp["input_ids"].shape[1]
    ids = torch.zeros(B, S + nq, dtype=torch.long)
    for b in range(B):                                    # collator layout: the <traj> tokens right behind each sample's own tokens
        ids[b, : lens[b]] = inp["input_ids"][b, : lens[b]]
        ids[b, lens[b]: lens[b] + nq] = cfg["traj_token_id"]
    batch = dict(input_ids=ids, t_s_pos=lens, pixel_values=inp["pixel_values"], image_grid_thw=inp["grid_thw"],
                 traj_images=traj_images, traj_poses=traj_poses, video_frame_num=vfn)
    loss = tr.forward_backward(batch, noise, t_index)
    torch.cuda.synchronize()
    # ---- per-layer drift of the latent-query rows (the saved layer inputs of LatentQueryGrad: layer i+1's input = stream after layer i)
    ctx = tr.lq._ctx
    streams = [ctx["saves"][i + 1][0] for i in range(cfg["t_layers"] - 1)] + [ctx["x_final"]]
    lines = ["# SFT step at full depth: latent-query rows, engine vs fp32 oracle next to bf16-autocast PyTorch vs the same oracle",
             "# layer | fp32 rms | engine rel | bf16 PyTorch rel | ratio"]
    worst = 0.0
    for i, x in enumerate(streams):
        e = _rel(x.float().cpu().view(B, nq, -1), gold["stream"][i])
        y = float(gold["bf16"]["stream_rel"][i])
        worst = max(worst, e / y)
        lines.append(f"llm {i:2d} {float(gold['stream_rms'][i]):7.3f} {e:.3e} {y:.3e} {e / y:.2f}")
    from internnav_amd import ops

    hidden = ops.norm(ctx["x_final"], eng.norm_w, None, eps=1e-6, rms=True).float().cpu().view(B, nq, -1)
    e_h, y_h = _rel(hidden, gold["hidden"]), gold["bf16"]["hidden_rel"]
    glq = tr.P.grad(LQ).float().cpu().view(nq, -1)
    e_g, y_g = _rel(glq, gold["d_lq"]), gold["bf16"]["d_lq_rel"]
    errs, yard = [], []
    gmax = max(g["norm"] for g in gold["grads"].values())
    for k, g in gold["grads"].items():
        if g["norm"] < 1e-6 * gmax:
            continue
        mine = tr.P.grad(k).float().cpu().flatten()[g["idx"]]
        errs.append(_rel(mine, g["val"]))
        yard.append(g["bf16_rel"])
    m_e, m_y = sum(errs) / len(errs), sum(yard) / len(yard)
    lines += [f"hidden states of the query rows: engine {e_h:.3e} vs bf16 PyTorch {y_h:.3e}",
              f"loss: engine {loss.item():.6f} fp32 {gold['loss']:.6f} bf16 PyTorch {gold['bf16']['loss']:.6f}",
              f"d latent_queries (through 28 frozen layers): engine {e_g:.3e} vs bf16 PyTorch {y_g:.3e}",
              f"System-1 parameter gradients ({len(errs)} tensors, 64 sampled entries each): engine mean rel {m_e:.3e} vs bf16 PyTorch {m_y:.3e}"]
    _report(lines)
    assert worst <= 1.0, f"latent-query rows drift further from fp32 than bf16 PyTorch at some layer (worst ratio {worst:.2f})"
    assert e_h <= y_h
    assert abs(loss.item() - gold["loss"]) <= max(2 * abs(gold["bf16"]["loss"] - gold["loss"]), 3e-3 * abs(gold["loss"]))
    assert e_g <= 1.25 * y_g + 1e-3
    assert m_e <= 1.25 * m_y
    # the optimiser step runs at this size too (flat store, fused clip + AdamW) and moves latent_queries in the engine
    tr.step_idx = 5
    before = eng.latent_q.clone()
    tr.reduce_gradients()
    tr.optimizer_step()
    assert not torch.equal(before, eng.latent_q) and float(tr.P.g32.abs().max()) == 0.0
