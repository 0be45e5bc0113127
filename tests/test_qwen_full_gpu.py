"""GPU parity of the System-2 engine at the FULL BASELINE configuration (32 ViT blocks, 28 decoder layers, vocabulary 152064,
4 x (1,28,28) images + 128 text tokens = S 920, the 7-env micro-batch of bench.py) with a per-layer drift report.

Fixture: tests/golden/qwen_full.pt (oracle/make_golden_full.py): fp32 oracle samples of the residual stream after EVERY ViT
block and decoder layer, last-position logits, 8 greedy tokens, the 4 latent queries - and, per layer, the error of the
reference's own bf16 path (installed transformers modules in bfloat16, the precision the reference runs in) against that fp32
result. The bar (north_star "1e-3 bf16 tolerance", read as: not worse than bf16 PyTorch): at EVERY layer the engine's error
against fp32 must not exceed the bf16-PyTorch error against fp32. Weights come from `HashWeights` (integer hash, identical
bits on the CPU that made the fixture and on this GPU; checked by checksum)."""
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = Path(__file__).resolve().parent / "golden" / "qwen_full.pt"


def _report(lines):
    out = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parent.parent)) / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        with open(out / "qwen_full_drift.txt", "a") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass
    print("\n".join(lines))


@pytest.fixture(scope="module")
def full(built_lib):
    from internnav_amd.qwen_vl import QwenVLEngine

    gold = torch.load(GOLD, weights_only=True)
    cfg = W.QWEN_N1_CFG
    hw = W.HashWeights(W.qwen_spec(cfg), gold["seed"], DEV)
    for k, v in gold["weight_check"].items():   # the GPU-side hash draws the same bits the CPU fixture was made with
        assert int(hw[k].view(torch.int16).to(torch.int64).sum()) == v, f"hash weights differ between CPU and GPU for {k}"
    inp = W.qwen_inputs(gold["B"], gold["n_img"], seed=gold["seed"], cfg=cfg, n_text=gold["n_text"], n_tail=gold["n_tail"])
    assert inp["input_ids"].shape == (gold["B"], gold["S"])
    eng = QwenVLEngine(hw, cfg, DEV, max_seqs=gold["B"], max_seq_len=1024, max_patches=inp["pixel_values"].shape[0])
    return gold, cfg, inp, eng


def _err(a, ref):
    d = (a - ref).abs()
    return d.mean().item(), d.max().item(), ((a - ref).pow(2).sum() / ref.pow(2).sum()).sqrt().item()


def test_full_depth_drift_logits_tokens_latents(full):
    gold, cfg, inp, eng = full
    B, S = gold["B"], gold["S"]
    pv = inp["pixel_values"].to(DEV, torch.bfloat16)
    vit_rows, vit_cols = gold["vit_rows"].to(DEV), gold["vit_cols"].to(DEV)
    llm_rows = (torch.arange(B)[:, None] * S + gold["llm_rows"]).to(DEV)
    llm_cols = gold["llm_cols"].to(DEV)
    taps = {"vit": [], "llm": []}

    def tap(kind, i, x):
        rows, cols = (vit_rows, vit_cols) if kind == "vit" else (llm_rows, llm_cols)
        taps[kind].append(x[rows.reshape(-1)][:, cols].reshape(rows.shape[0], rows.shape[1], cols.numel()).float().cpu())

    eng.tap = tap
    state = eng.prefill(inp["input_ids"], pv, inp["grid_thw"])
    eng.tap = None
    eng._last_logits(B, S, S - 1)
    logits = eng.logits[:B].float().cpu()
    assert len(taps["vit"]) == cfg["v_depth"] and len(taps["llm"]) == cfg["t_layers"]

    lines = ["# engine vs fp32 oracle, next to the reference's bf16 PyTorch path vs the same fp32 oracle (per layer, all 7 envs' samples)",
             "layer   fp32_rms   engine_mean|err|  engine_max|err|  engine_rel   bf16torch_mean|err|  bf16torch_max|err|  bf16torch_rel"]
    worst = 0.0
    for kind, ref_h, ref_rms, yard in (("vit", gold["vit_h"], gold["vit_rms"], gold["bf16_vit"]), ("llm", gold["llm_h"], gold["llm_rms"], gold["bf16_llm"])):
        for i, t in enumerate(taps[kind]):
            m, mx, rel = _err(t, ref_h[i])
            lines.append(f"{kind} {i:2d} {ref_rms[i].mean():9.3f}   {m:.3e}   {mx:.3e}   {rel:.3e}   {yard['mean'][i]:.3e}   {yard['max'][i]:.3e}   {yard['rel'][i]:.3e}")
            worst = max(worst, m / float(yard["mean"][i]))
    ld = (logits[:, gold["voc_idx"]] - gold["logits_samp"]).abs()
    std = gold["logit_std"].mean().item()
    lines.append(f"last-position logits (every 37th of 152064): engine mean|err| {ld.mean():.3e} max {ld.max():.3e}  |  bf16 PyTorch mean {float(gold['bf16_logits']['mean']):.3e} "
                 f"max {float(gold['bf16_logits']['max']):.3e}  |  logit std {std:.3f} -> engine mean|err| / std = {ld.mean() / std:.2e}")
    top = gold["logits_top"]
    td = (logits.gather(1, top["indices"]) - top["values"]).abs()
    lines.append(f"top-32 logits per env: engine mean|err| {td.mean():.3e} max {td.max():.3e}; worst engine/bf16torch mean-error ratio over all layers {worst:.2f}")
    _report(lines)
    for kind, ref_h, yard in (("vit", gold["vit_h"], gold["bf16_vit"]), ("llm", gold["llm_h"], gold["bf16_llm"])):
        for i, t in enumerate(taps[kind]):
            m, mx, rel = _err(t, ref_h[i])
            assert m <= float(yard["mean"][i]) and rel <= float(yard["rel"][i]), f"{kind} layer {i}: engine error {m:.3e} exceeds the bf16 PyTorch path's {float(yard['mean'][i]):.3e}"
    assert ld.mean().item() <= float(gold["bf16_logits"]["mean"])
    assert ld.mean().item() <= 5e-3 * std      # absolute reading of the tolerance at 60 layers of depth (measured value printed above)

    # greedy tokens (the reference parses the decoded ids with a regex, internvla_n1_policy.py:169-186: they must be the same tokens). There is ONE
    # decode path in the engine (no kernel switch can change them: tests/test_qwen_gpu.py::test_single_token_pass_variants_agree); against the fp32
    # oracle a token may differ ONLY at a step whose fp32 top-2 margin is below 0.04 logits - twice the largest logit error ANY bf16 arithmetic shows
    # here (the engine's max is 2.0e-2, the reference's own bf16 path 6.7e-2) - and a sequence stops being comparable after such a step.
    NEAR_TIE = 0.04
    toks = eng.decode(state, gold["n_decode"]).cpu().long()
    ref_t, margins = gold["tokens"], gold["margins"]
    n_cmp, alive = 0, [True] * B
    for j in range(gold["n_decode"]):
        for b in range(B):
            if not alive[b]:
                continue
            if toks[b, j] == ref_t[b, j]:
                n_cmp += 1
            else:
                assert margins[b, j] < NEAR_TIE, f"env {b} token {j}: {toks[b, j]} != {ref_t[b, j]} although the fp32 margin is {margins[b, j]:.3f}"
                alive[b] = False
    clear = [b for b in range(B) if bool((margins[b] >= NEAR_TIE).all())]
    lines = [f"greedy tokens: {n_cmp} of {B * gold['n_decode']} compared equal; envs still identical after {gold['n_decode']} tokens: {sum(alive)} of {B}; "
             f"envs without a near-tie (all margins >= {NEAR_TIE}): {clear} - all identical: {all(alive[b] for b in clear)}",
             f"engine {toks.tolist()}", f"fp32   {ref_t.tolist()}"]
    # the fixture: envs 3, 5, 6 never come closer than 0.04 logits to a tie over the 8 steps - they must be identical; the other four carry one or two
    # sub-0.02 steps each (random-weight logits over 152064 entries). Of the 56 tokens at most those behind a near-tie flip may be lost.
    assert len(clear) >= 3 and all(alive[b] for b in clear)
    near = int((margins < NEAR_TIE).sum())
    assert sum(alive) >= B - near and n_cmp >= len(clear) * gold["n_decode"]
    # the 4 latent queries on the KV cache generate() left (reference: full re-run, internvla_n1.py:320-347)
    lat = eng.latents(state, toks[:, -1:].to(DEV, torch.int32).contiguous()).float().cpu()
    keep = torch.tensor(alive)
    m, mx, rel = _err(lat[keep], gold["latents"][keep])
    lines.append(f"latent queries (envs with identical tokens): engine mean|err| {m:.3e} max {mx:.3e} rel {rel:.3e}  |  bf16 PyTorch mean {float(gold['bf16_latents']['mean']):.3e} "
                 f"max {float(gold['bf16_latents']['max']):.3e} rel {float(gold['bf16_latents']['rel']):.3e}")
    _report(lines)
    assert m <= float(gold["bf16_latents"]["mean"]) and rel <= float(gold["bf16_latents"]["rel"])


def test_full_config_image_embeds(full):
    gold, cfg, inp, eng = full
    grids = [tuple(g) for g in inp["grid_thw"].tolist()]
    emb, inv = eng.vision(inp["pixel_values"].to(DEV, torch.bfloat16), grids)
    tok = emb.float().cpu()[torch.from_numpy(inv).long()]
    rows, cols = gold["emb_rows"], gold["llm_cols"]
    mine = tok[rows.reshape(-1)][:, cols].reshape(rows.shape[0], rows.shape[1], -1)
    m, mx, rel = _err(mine, gold["emb"])
    _report([f"merged image embeds: engine mean|err| {m:.3e} max {mx:.3e} rel {rel:.3e}  |  bf16 PyTorch mean {float(gold['bf16_emb']['mean']):.3e} rel {float(gold['bf16_emb']['rel']):.3e}  (rms {gold['emb_rms'].mean():.3f})"])
    assert m <= float(gold["bf16_emb"]["mean"]) and rel <= float(gold["bf16_emb"]["rel"])
