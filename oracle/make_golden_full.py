"""Full-configuration System-2 parity fixture with a per-layer drift report.  TEST INFRASTRUCTURE (build container only).

    python -m oracle.make_golden_full            # -> tests/golden/qwen_full.pt   (about 25 min on 8 cores, 45 GB RAM)

What runs: the BASELINE configuration of the System-2 call - Qwen2.5-VL-7B dims: 32 ViT blocks, 28 decoder
layers, vocabulary 152064 - on the 7-env micro-batch of `bench.py` (4 frames of 28x28 patches + 128 text tokens per env,
S = 920), with the hash-seeded weights of `internnav_amd.synthetic.HashWeights` (bit-identical on the CPU here and on the GPU in
tests/test_qwen_full_gpu.py):
  1. the fp32 oracle (oracle/qwen_vl.py, pinned against the transformers modules on the reduced configuration by make_golden.py):
     vision tower, prefill, 8 greedy tokens on a KV cache, the 4 latent queries; the residual stream is sampled after EVERY ViT
     block and EVERY decoder layer (16 rows x 128 columns per env) together with its RMS;
  2. the reference's own arithmetic in its own precision: the installed transformers Qwen2_5_VisionTransformerPretrainedModel /
     Qwen2_5_VLTextModel in **bf16** (what `InternVLAN1ForCausalLM.from_pretrained(torch_dtype=bfloat16)` executes,
     internvla_n1_policy.py:33-38; sdpa instead of flash-attn) with the reference's glue (internvla_n1.py:128-220, 320-347) and
     the reference's vendored get_rope_index_25, sampled at the same places. Its error against (1) is stored per layer:
     that is the yardstick "1e-3 bf16 tolerance" of north_star is read against - the HIP engine must not be further from the
     fp32 result than the reference's bf16 PyTorch path is, at every layer (asserted by the GPU test).
The fixture keeps samples only (about 4 MB).
"""
from __future__ import annotations

import argparse
import gc
import time
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

from . import qwen_vl as o_q
from . import ref_loader as R
from . import weights as W

GOLD = Path(__file__).resolve().parent.parent / "tests" / "golden"
SEED, B, N_IMG, N_TEXT, N_TAIL, N_DEC = 11, 7, 4, 96, 32, 8
N_ROWS, N_COLS = 16, 128


class CachedWeights:
    """HashWeights materialised once as bf16 in RAM (15 GB); the fp32 oracle reads each tensor through .float()."""

    def __init__(self, spec, seed, outliers=False):
        t0 = time.time()
        hw = (W.OutlierHashWeights if outliers else W.HashWeights)(spec, seed, "cpu")
        self.bf16 = {k: hw[k] for k in spec}
        print(f"[weights] {sum(v.numel() for v in self.bf16.values()) / 1e9:.2f} B parameters drawn in {time.time() - t0:.0f} s", flush=True)

    def __getitem__(self, k):
        return self.bf16[k].float()

    def get(self, k, default=None):
        return self.bf16[k].float() if k in self.bf16 else default

    def __contains__(self, k):
        return k in self.bf16


def sample_plan(cfg, inp):
    """which rows / columns of the per-layer residual streams the fixture keeps (fixed by SEED)."""
    g = torch.Generator().manual_seed(SEED + 1)
    S = inp["input_ids"].shape[1]
    npatch = inp["pixel_values"].shape[0] // B
    llm_rows = torch.stack([torch.cat([torch.randperm(S - 1, generator=g)[: N_ROWS - 1].sort().values, torch.tensor([S - 1])]) for _ in range(B)])
    vit_rows = torch.stack([b * npatch + torch.randperm(npatch, generator=g)[:N_ROWS].sort().values for b in range(B)])
    emb_rows = torch.stack([b * (npatch // 4) + torch.randperm(npatch // 4, generator=g)[:N_ROWS].sort().values for b in range(B)])
    llm_cols = torch.arange(N_COLS) * (cfg["t_hidden"] // N_COLS) + 3
    vit_cols = torch.arange(N_COLS) * (cfg["v_hidden"] // N_COLS) + 1
    voc_idx = torch.arange(0, cfg["vocab"], 37)
    return dict(llm_rows=llm_rows, vit_rows=vit_rows, emb_rows=emb_rows, llm_cols=llm_cols, vit_cols=vit_cols, voc_idx=voc_idx)


def _samp(x2d, rows, cols):
    return x2d[rows.reshape(-1)][:, cols].reshape(rows.shape[0], rows.shape[1], cols.numel()).float().clone()


def _rms_per_env(x2d):
    return x2d.float().reshape(B, -1).pow(2).mean(-1).sqrt()


def run_fp32(sd, cfg, inp, sp):
    ids, pv, grid = inp["input_ids"], inp["pixel_values"], inp["grid_thw"]
    S = ids.shape[1]
    out = dict(vit_h=[], vit_rms=[], llm_h=[], llm_rms=[])
    t0 = time.time()

    def vtap(i, x):
        out["vit_h"].append(_samp(x, sp["vit_rows"], sp["vit_cols"]))
        out["vit_rms"].append(_rms_per_env(x))
        print(f"[fp32] vit block {i} done {time.time() - t0:.0f}s rms {out['vit_rms'][-1][0]:.3f}", flush=True)

    def ltap(i, x):
        out["llm_h"].append(_samp(x.reshape(B * S, -1), torch.arange(B)[:, None] * S + sp["llm_rows"], sp["llm_cols"]))
        out["llm_rms"].append(_rms_per_env(x.reshape(B * S, -1)))
        print(f"[fp32] llm layer {i} done {time.time() - t0:.0f}s rms {out['llm_rms'][-1][0]:.3f}", flush=True)

    with torch.no_grad():
        emb = o_q.vision_tower(pv, grid, sd, cfg, tap=vtap)
        out["emb"] = _samp(emb, sp["emb_rows"], sp["llm_cols"])
        out["emb_rms"] = _rms_per_env(emb)
        x = o_q.input_embeds(ids, emb, sd, cfg)
        pos, _ = o_q.rope_index(ids, grid, cfg["image_token_id"], cfg["vision_start_id"])
        cache = [None] * cfg["t_layers"]
        h = o_q.decoder_stack(x, pos, sd, cfg, tap=ltap, cache=cache)
        lm = sd["lm_head.weight"]
        logits = F.linear(h[:, -1], lm)
        out["last_logits_full"] = logits.clone()
        nxt_pos = pos[:, :, -1].max(0).values + 1
        toks, margins = [], []
        for j in range(N_DEC):
            top2 = logits.topk(2, dim=-1)
            toks.append(top2.indices[:, 0])
            margins.append(top2.values[:, 0] - top2.values[:, 1])
            print(f"[fp32] token {j}: {toks[-1].tolist()} margin {[round(float(m), 3) for m in margins[-1]]} {time.time() - t0:.0f}s", flush=True)
            xe = sd["model.embed_tokens.weight"][toks[-1]][:, None, :]
            p3 = (nxt_pos + j)[None, :, None].expand(3, B, 1)
            if j == N_DEC - 1:   # the last sampled token runs together with the latent queries (generate_latents appends them, internvla_n1.py:325-329)
                nq = cfg["n_query"]
                xq = torch.cat([xe, sd["model.latent_queries"].reshape(1, nq, -1).expand(B, nq, -1)], dim=1)
                pq = (nxt_pos + j)[None, :, None] + torch.arange(nq + 1)[None, None, :]
                hq = o_q.decoder_stack(xq, pq.expand(3, B, nq + 1), sd, cfg, cache=cache)
                out["latents"] = hq[:, 1:].clone()
            else:
                hj = o_q.decoder_stack(xe, p3, sd, cfg, cache=cache)
                logits = F.linear(hj[:, -1], lm)
        out["tokens"] = torch.stack(toks, 1)
        out["margins"] = torch.stack(margins, 1)
    out["position_ids"] = pos.to(torch.int32)
    for k in ("vit_h", "vit_rms", "llm_h", "llm_rms"):
        out[k] = torch.stack(out[k])
    return out


def run_bf16_transformers(wc, cfg, inp, sp, tokens):
    """the reference's bf16 path on the installed transformers modules (hooks sample every block / layer)."""
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLTextConfig, Qwen2_5_VLVisionConfig
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import (Qwen2_5_VisionRotaryEmbedding, Qwen2_5_VisionTransformerPretrainedModel,
                                                                    Qwen2_5_VLRotaryEmbedding, Qwen2_5_VLTextModel)

    bf = torch.bfloat16
    vc = Qwen2_5_VLVisionConfig(depth=cfg["v_depth"], hidden_size=cfg["v_hidden"], intermediate_size=cfg["v_inter"], num_heads=cfg["v_heads"],
                                out_hidden_size=cfg["v_out"], fullatt_block_indexes=list(cfg["v_fullatt"]), window_size=cfg["v_window"])
    vc._attn_implementation = "sdpa"
    tc = Qwen2_5_VLTextConfig(vocab_size=cfg["vocab"], hidden_size=cfg["t_hidden"], intermediate_size=cfg["t_inter"],
                              num_hidden_layers=cfg["t_layers"], num_attention_heads=cfg["t_heads"], num_key_value_heads=cfg["t_kv_heads"],
                              rms_norm_eps=1e-6, rope_parameters={"rope_type": "default", "rope_theta": cfg["rope_theta"], "mrope_section": [16, 24, 24]},
                              max_position_embeddings=32768, pad_token_id=0)
    tc._attn_implementation = "sdpa"
    with torch.device("meta"):
        vit = Qwen2_5_VisionTransformerPretrainedModel(vc)
        llm = Qwen2_5_VLTextModel(tc)
    vit, llm = vit.to_empty(device="cpu"), llm.to_empty(device="cpu")
    vit.rotary_pos_emb = Qwen2_5_VisionRotaryEmbedding(cfg["v_hidden"] // cfg["v_heads"] // 2)
    llm.rotary_emb = Qwen2_5_VLRotaryEmbedding(config=tc)
    vit.load_state_dict({k[len("visual."):]: v for k, v in wc.bf16.items() if k.startswith("visual.")}, strict=True, assign=True)
    llm.load_state_dict({k[len("model."):]: v for k, v in wc.bf16.items() if k.startswith("model.") and k != "model.latent_queries"},
                        strict=True, assign=True)
    vit, llm = vit.eval(), llm.eval()
    assert next(llm.parameters()).dtype == bf and next(vit.parameters()).dtype == bf
    src = (R.REF / "internnav" / "dataset" / "rope2d.py").read_text()
    src = src.replace("image_token_id = 151655", f"image_token_id = {cfg['image_token_id']}").replace(
        "vision_start_token_id = 151652", f"vision_start_token_id = {cfg['vision_start_id']}")   # no-ops at the full configuration
    ns = {}
    exec(compile(src, "rope2d_ref", "exec"), ns)
    ids, pv, grid = inp["input_ids"], inp["pixel_values"].to(bf), inp["grid_thw"]
    S = ids.shape[1]
    out = dict(vit_h=[], llm_h=[])
    t0 = time.time()
    hooks = []
    def vhook(m, a, o, i):
        out["vit_h"].append(_samp(o if isinstance(o, torch.Tensor) else o[0], sp["vit_rows"], sp["vit_cols"]))
        print(f"[bf16] vit block {i} {time.time() - t0:.0f}s", flush=True)

    for i, blk in enumerate(vit.blocks):
        hooks.append(blk.register_forward_hook(lambda m, a, o, i=i: vhook(m, a, o, i)))
    state = dict(on=True)

    def lhook(m, a, o, i):
        if state["on"]:
            x = o if isinstance(o, torch.Tensor) else o[0]
            out["llm_h"].append(_samp(x.reshape(B * S, -1), torch.arange(B)[:, None] * S + sp["llm_rows"], sp["llm_cols"]))
            print(f"[bf16] llm layer {i} {time.time() - t0:.0f}s", flush=True)

    for i, lyr in enumerate(llm.layers):
        hooks.append(lyr.register_forward_hook(lambda m, a, o, i=i: lhook(m, a, o, i)))
    lq = wc.bf16["model.latent_queries"]
    lm = wc.bf16["lm_head.weight"]

    def forward(i):
        emb = vit(pv, grid_thw=grid).pooler_output
        x = llm.embed_tokens(i)
        x = x.masked_scatter((i == cfg["image_token_id"]).unsqueeze(-1).expand_as(x), emb)
        traj = i == cfg["traj_token_id"]
        if traj.any():
            x[traj] = lq.repeat(i.shape[0], 1, 1).view(-1, x.shape[-1])
        pos, _ = ns["get_rope_index_25"](2, i, grid)
        h = llm(inputs_embeds=x, position_ids=pos, use_cache=False).last_hidden_state
        return h, emb, pos

    with torch.no_grad():
        h, emb, pos = forward(ids)
        out["emb"] = _samp(emb, sp["emb_rows"], sp["llm_cols"])
        out["last_logits_full"] = F.linear(h[:, -1], lm).float()
        out["position_ids"] = pos.to(torch.int32)
        # generate_latents as the reference runs it: ViT + the whole sequence again with N_QUERY traj tokens appended (internvla_n1.py:320-347)
        state["on"] = False
        for hk in hooks[: len(vit.blocks)]:
            hk.remove()
        nq = cfg["n_query"]
        ids_q = torch.cat([ids, tokens, torch.full((B, nq), cfg["traj_token_id"], dtype=torch.long)], dim=1)
        hq, _, _ = forward(ids_q)
        out["latents"] = hq[:, -nq:].float().clone()
        print(f"[bf16] latents done {time.time() - t0:.0f}s", flush=True)
    out["vit_h"], out["llm_h"] = torch.stack(out["vit_h"]), torch.stack(out["llm_h"])
    return out


def _err(a, ref):
    d = (a - ref).abs()
    return dict(mean=d.flatten(1).mean(1) if d.dim() > 1 else d.mean(), max=d.flatten(1).max(1).values if d.dim() > 1 else d.max(),
                rel=((a - ref).flatten(1).pow(2).sum(1) / ref.flatten(1).pow(2).sum(1)).sqrt())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(GOLD / "qwen_full.pt"))
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--reduced", action="store_true", help="dry run of this script on the 2+2-layer test configuration")
    ap.add_argument("--outliers", action="store_true", help="massive-activation channels in the decoder's residual stream (synthetic.OutlierHashWeights) "
                    "-> tests/golden/qwen_full_outliers.pt")
    ap.add_argument("--envs", type=int, default=7, help="sequences in the batch (the outlier fixture uses 3: CPU time)")
    a = ap.parse_args()
    globals()["B"] = a.envs
    if a.outliers and a.out == str(GOLD / "qwen_full.pt"):
        a.out = str(GOLD / "qwen_full_outliers.pt")
    torch.set_num_threads(a.threads)
    cfg = W.QWEN_TEST_CFG if a.reduced else W.QWEN_N1_CFG
    inp = W.qwen_inputs(B, N_IMG, seed=SEED, cfg=cfg, n_text=N_TEXT, n_tail=N_TAIL)
    assert inp["input_ids"].shape == (B, 920)
    assert all(c in (torch.arange(N_COLS) * (cfg["t_hidden"] // N_COLS) + 3).tolist() for c in W.OUTLIER_CHANNELS)
    sp = sample_plan(cfg, inp)
    wc = CachedWeights(W.qwen_spec(cfg), SEED, outliers=a.outliers)
    check = {k: int(wc.bf16[k].view(torch.int16).to(torch.int64).sum()) for k in
             ("model.layers.0.self_attn.q_proj.weight", "model.layers.0.mlp.down_proj.weight", "model.layers.1.input_layernorm.weight", "model.norm.weight",
              f"model.layers.{cfg['t_layers'] - 1}.mlp.down_proj.weight",
              f"visual.blocks.{cfg['v_depth'] - 1}.mlp.up_proj.weight", "lm_head.weight", "model.embed_tokens.weight", "visual.merger.mlp.2.bias")}
    f32 = run_fp32(wc, cfg, inp, sp)
    gc.collect()
    b16 = run_bf16_transformers(wc, cfg, inp, sp, f32["tokens"])
    assert torch.equal(b16["position_ids"], f32["position_ids"]), "oracle rope_index != reference get_rope_index_25"
    fx = dict(seed=SEED, outliers=bool(a.outliers), outlier_channels=torch.tensor(W.OUTLIER_CHANNELS if a.outliers else ()), B=B, n_img=N_IMG, n_text=N_TEXT, n_tail=N_TAIL, S=920, n_decode=N_DEC, weight_check=check, **sp,
              position_ids=f32["position_ids"],
              vit_h=f32["vit_h"], vit_rms=f32["vit_rms"], llm_h=f32["llm_h"], llm_rms=f32["llm_rms"], emb=f32["emb"], emb_rms=f32["emb_rms"],
              tokens=f32["tokens"], margins=f32["margins"], latents=f32["latents"],
              logit_std=f32["last_logits_full"].std(dim=-1), logits_top=f32["last_logits_full"].topk(32, dim=-1),
              logits_samp=f32["last_logits_full"][:, sp["voc_idx"]].clone())
    fx["logits_top"] = dict(values=fx["logits_top"].values.clone(), indices=fx["logits_top"].indices.clone())
    # the reference's bf16 PyTorch path vs fp32, per layer (over all envs' samples): the yardstick of the GPU test
    fx["bf16_vit"] = {k: torch.stack([_err(b16["vit_h"][i].reshape(1, -1), f32["vit_h"][i].reshape(1, -1))[k][0] for i in range(cfg["v_depth"])])
                      for k in ("mean", "max", "rel")}
    fx["bf16_llm"] = {k: torch.stack([_err(b16["llm_h"][i].reshape(1, -1), f32["llm_h"][i].reshape(1, -1))[k][0] for i in range(cfg["t_layers"])])
                      for k in ("mean", "max", "rel")}
    fx["bf16_emb"] = {k: v[0] for k, v in _err(b16["emb"].reshape(1, -1), f32["emb"].reshape(1, -1)).items()}
    fx["bf16_logits"] = {k: v[0] for k, v in _err(b16["last_logits_full"][:, sp["voc_idx"]].reshape(1, -1), fx["logits_samp"].reshape(1, -1)).items()}
    fx["bf16_latents"] = {k: v[0] for k, v in _err(b16["latents"].reshape(1, -1), f32["latents"].reshape(1, -1)).items()}
    fx["bf16_tokens0"] = b16["last_logits_full"].argmax(-1)
    if a.outliers:
        # the same per-layer yardstick over the NON-outlier columns of the samples only (the outlier columns alone carry the bf16 path's
        # 2-4-unit roundings of a bf16 residual stream and would hide everything else in a mean), and over the outlier columns alone
        oc = torch.isin(sp["llm_cols"], fx["outlier_channels"])
        assert int(oc.sum()) == len(W.OUTLIER_CHANNELS), "every outlier channel must be one of the sampled columns"
        for name, m in (("bf16_llm_rest", ~oc), ("bf16_llm_outl", oc)):
            fx[name] = {k: torch.stack([_err(b16["llm_h"][i][..., m].reshape(1, -1), f32["llm_h"][i][..., m].reshape(1, -1))[k][0]
                                        for i in range(cfg["t_layers"])]) for k in ("mean", "max", "rel")}
        fx["outlier_abs_mean"] = torch.stack([f32["llm_h"][i][..., oc].abs().mean() for i in range(cfg["t_layers"])])
        fx["rest_abs_mean"] = torch.stack([f32["llm_h"][i][..., ~oc].abs().mean() for i in range(cfg["t_layers"])])
        print("outlier |x| per layer", [round(float(v), 1) for v in fx["outlier_abs_mean"]], "rest", [round(float(v), 2) for v in fx["rest_abs_mean"]])
    torch.save(fx, a.out)
    print("wrote", a.out, Path(a.out).stat().st_size / 1e6, "MB")
    print("layer | fp32 rms | bf16-PyTorch mean|err| max|err| rel")
    for i in range(cfg["v_depth"]):
        print(f"vit {i:2d} {f32['vit_rms'][i].mean():9.3f} {fx['bf16_vit']['mean'][i]:.3e} {fx['bf16_vit']['max'][i]:.3e} {fx['bf16_vit']['rel'][i]:.3e}")
    for i in range(cfg["t_layers"]):
        print(f"llm {i:2d} {f32['llm_rms'][i].mean():9.3f} {fx['bf16_llm']['mean'][i]:.3e} {fx['bf16_llm']['max'][i]:.3e} {fx['bf16_llm']['rel'][i]:.3e}")
    print("logits", {k: float(v) for k, v in fx["bf16_logits"].items()}, "std", fx["logit_std"].tolist())
    print("latents", {k: float(v) for k, v in fx["bf16_latents"].items()})
    print("bf16 first tokens", fx["bf16_tokens0"].tolist(), "fp32", f32["tokens"][:, 0].tolist())


if __name__ == "__main__":
    main()
