"""SFT step at the FULL depth of config #5 - tests/golden/sft_full.pt.  TEST INFRASTRUCTURE (build container only).

    python -m oracle.make_golden_sft_full          # about 40 min on 8 cores, < 40 GB RAM

VERDICT r2 "SFT at depth": every earlier SFT parity test ran 2 ViT blocks + 2 decoder layers. This fixture is the `bench_sft.py`
shape: Qwen2.5-VL-7B dims (32 ViT blocks, 28 decoder layers), micro-batch 2, 10 frames + text = 2 108 prompt tokens (the second
sample 21 tokens shorter: ragged batch), T sub-goals per sample, hash-seeded weights (`synthetic.HashWeights`, bit-identical on the
GPU). Computed here:
  1. fp32 oracle: prefix on a KV cache, latent-query rows through 28 layers (residual stream kept after EVERY layer), the
     nextdit_async SFT loss of oracle/sft.py on those hidden states, its autograd gradient for every System-1 parameter, and
     d loss / d latent_queries back through the 28 frozen layers (oracle/sft.latent_query_grad_cached);
  2. the same under bf16 autocast = the yardstick (what the reference's bf16 training computes), error per layer stored.
"""
from __future__ import annotations

import argparse
import gc
import time
from pathlib import Path

import torch

from . import sft as o_sft
from . import weights as W
from .make_golden_full import CachedWeights

GOLD = Path(__file__).resolve().parent.parent / "tests" / "golden"
SEED, B, FRAMES, T, N_TEXT, N_TAIL, SHORT = 13, 2, 10, 4, 98, 30, 21


def make_batch(cfg, reduced=False):
    """the collator-style batch of the fixture (shared with tests/test_sft_full_gpu.py through this function's arguments only)."""
    frames = 2 if reduced else FRAMES
    inp = W.qwen_inputs(B, frames, seed=SEED, cfg=cfg, n_text=N_TEXT, n_tail=N_TAIL)
    S = inp["input_ids"].shape[1]
    lens = [S, S - SHORT]
    g = torch.Generator().manual_seed(SEED)
    traj_images = torch.rand(B, T, 224, 224, 3, generator=g)
    traj_poses = torch.randn(B, T, 32, 3, generator=g)
    vfn = torch.tensor([T, T - 1])
    noise = torch.randn(B * T, 32, 3, generator=g)
    t_index = torch.randint(0, 1000, (B * T,), generator=g)
    return inp, lens, traj_images, traj_poses, vfn, noise, t_index


def run(wc, sd_s0, cfg, batch, autocast, log):
    inp, lens, traj_images, traj_poses, vfn, noise, t_index = batch
    per_pv, per_g = inp["pixel_values"].shape[0] // B, inp["grid_thw"].shape[0] // B
    tag = "bf16" if autocast else "fp32"
    t0 = time.time()
    sd_s = {k: v.clone().requires_grad_(True) for k, v in sd_s0.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
        passes, streams = [], []
        for b in range(B):                     # the oracle runs unpadded sequences one by one
            L = lens[b]
            st = []
            passes.append(o_sft.LatentQueryOracle(wc, cfg, inp["input_ids"][b:b + 1, :L], inp["pixel_values"][b * per_pv:(b + 1) * per_pv],
                                                  inp["grid_thw"][b * per_g:(b + 1) * per_g], tap=lambda i, x, st=st: st.append(x[0].float().clone())))
            streams.append(torch.stack(st))
            log(f"{tag} sample {b}: prefix of {L} tokens + query rows through {cfg['t_layers']} layers {time.time() - t0:.0f} s")
        hidden = torch.cat([p.hidden.float() for p in passes]).detach().requires_grad_(True)
        loss = o_sft.nextdit_sft_loss(sd_s, hidden, traj_images, traj_poses, vfn, noise, t_index)
        loss.backward()
        d_lq = 0
        for b in range(B):
            d_lq = d_lq + passes[b].backward(hidden.grad[b:b + 1])
            log(f"{tag} sample {b}: backward through {cfg['t_layers']} layers {time.time() - t0:.0f} s")
    grads = {k: v.grad.clone() for k, v in sd_s.items() if v.grad is not None}
    return dict(loss=loss.item(), hidden=hidden.detach(), d_hidden=hidden.grad.clone(), d_lq=d_lq, grads=grads,
                stream=torch.stack(streams, 1))                                                # stream [layers, B, nq, H]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(GOLD / "sft_full.pt"))
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--reduced", action="store_true", help="dry run on the 2+2-layer test configuration")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    cfg = W.QWEN_TEST_CFG if a.reduced else W.QWEN_N1_CFG
    batch = make_batch(cfg, a.reduced)
    wc = CachedWeights(W.qwen_spec(cfg), SEED)
    sd_s = {k: v.float() for k, v in W.materialize(W.n1_nextdit_spec(), SEED).items()}
    log = lambda m: print("[sft_full]", m, flush=True)  # noqa: E731
    f32 = run(wc, sd_s, cfg, batch, False, log)
    gc.collect()
    b16 = run(wc, sd_s, cfg, batch, True, log)
    rel = lambda x, y: ((x - y).norm() / y.norm()).item()  # noqa: E731
    gmax = max(g.norm().item() for g in f32["grads"].values())
    picks = {}
    for k, g in f32["grads"].items():
        flat = g.flatten()
        idx = torch.linspace(0, flat.numel() - 1, min(64, flat.numel())).long()
        picks[k] = dict(norm=g.norm().item(), idx=idx, val=flat[idx].clone(),
                        bf16_rel=(rel(b16["grads"][k], g) if g.norm().item() > 1e-6 * gmax else 0.0))
    L = cfg["t_layers"]
    fx = dict(seed=SEED, B=B, frames=2 if a.reduced else FRAMES, T=T, n_text=N_TEXT, n_tail=N_TAIL, short=SHORT, lens=batch[1],
              loss=f32["loss"], hidden=f32["hidden"], d_hidden=f32["d_hidden"], d_lq=f32["d_lq"], stream=f32["stream"], grads=picks,
              bf16=dict(loss=b16["loss"], hidden_rel=rel(b16["hidden"], f32["hidden"]), d_lq_rel=rel(b16["d_lq"], f32["d_lq"]),
                        stream_rel=torch.tensor([rel(b16["stream"][i], f32["stream"][i]) for i in range(L)]),
                        stream_mean=torch.tensor([(b16["stream"][i] - f32["stream"][i]).abs().mean().item() for i in range(L)])),
              stream_rms=torch.tensor([f32["stream"][i].pow(2).mean().sqrt().item() for i in range(L)]))
    torch.save(fx, a.out)
    print("wrote", a.out, Path(a.out).stat().st_size / 1e6, "MB")
    print(f"loss fp32 {f32['loss']:.6f} bf16 {b16['loss']:.6f}; hidden rel {fx['bf16']['hidden_rel']:.3e}; d latent_queries rel {fx['bf16']['d_lq_rel']:.3e}")
    for i in range(L):
        print(f"layer {i:2d} rms {fx['stream_rms'][i]:.3f} bf16 rel {fx['bf16']['stream_rel'][i]:.3e} mean|err| {fx['bf16']['stream_mean'][i]:.3e}")


if __name__ == "__main__":
    main()
