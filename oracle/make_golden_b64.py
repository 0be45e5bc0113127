"""TEST INFRASTRUCTURE: per-env yardstick of the 64-env System-1 parity tests (tests/test_b64_distribution_gpu.py).

For EVERY env of the seeded 64-env batches of tests/test_b64_spotcheck_gpu.py (NavDPNet seed 21, NextDiT seed 22) and of an N1 NavDP head
batch (seed 23) this runs, one env per call as the reference executes (navdp_policy.py:165, internvla_n1/navdp.py:228-231):
  * the fp32 CPU oracle                      -> the reference output the engine is compared with on the GPU box (no oracle run there),
  * the same oracle under bf16 autocast      -> the error the reference's own precision has on that env (mean, 99th percentile, max).
Written to tests/golden/s1_b64_yardstick.pt (fp32 outputs stored as float16-free fp32 tensors, ~3 MB).
    python -m oracle.make_golden_b64 [threads] [section ...]      sections: navdpnet nextdit nextdit_ffn1024 n1_navdp (default: all;
                                                                   named sections are recomputed into the existing file)
The NextDiT section exists for both FFN widths of the reference's block (oracle/diffusers_blocks.py: LEGACY_TWO_THIRDS): `nextdit` = 1536
(diffusers 0.33.1 as pinned), `nextdit_ffn1024` = the diffusers <= 0.32 convention.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import torch

from . import navdp as o_navdp
from . import nextdit as o_nextdit
from . import weights as W

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "s1_b64_yardstick.pt"
B = 64


def _stats(y, ref):
    e = (y.float() - ref).abs().flatten()
    return [e.mean().item(), torch.quantile(e, 0.99).item(), e.max().item()]


def _nextdit(cfg, t0, tag):
    sd, inp = W.n1_nextdit_state_dict(seed=22, cfg=cfg), W.n1_nextdit_inputs(B, seed=22, cfg=cfg)
    fin, ys = [], []
    for b in range(B):
        a = (sd, inp["traj_latents"][b:b + 1], inp["images"][b:b + 1], inp["x_init"][b:b + 1])
        with torch.no_grad():
            f32 = o_nextdit.generate_traj(*a)
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            f16 = o_nextdit.generate_traj(*a)
        fin.append(f32.float().reshape(inp["x_init"].shape[1:])), ys.append(_stats(f16.reshape(fin[-1].shape), fin[-1]))
        print(f"{tag} env {b}: bf16 mean {ys[-1][0]:.3e} p99 {ys[-1][1]:.3e} max {ys[-1][2]:.3e}  [{time.time() - t0:.0f}s]", flush=True)
    return dict(seed=22, dit_ffn=cfg["dit_ffn"], latents=torch.stack(fin), yard=torch.tensor(ys))


def main():
    torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
    only = set(sys.argv[2:])
    out = torch.load(OUT, weights_only=False) if only and OUT.exists() else {"B": B}
    t0 = time.time()
    if not only or "navdpnet" in only:
        out["navdpnet"] = _navdpnet(t0)
    for name, cfg in (("nextdit", W.N1_NEXTDIT_CFG), ("nextdit_ffn1024", W.N1_NEXTDIT_CFG_FFN1024)):
        if not only or name in only:
            out[name] = _nextdit(cfg, t0, name)
    if not only or "n1_navdp" in only:
        out["n1_navdp"] = _n1_navdp(t0)
    torch.save(out, OUT)
    print("wrote", OUT, OUT.stat().st_size, "bytes")


def _navdpnet(t0):
    # ---- NavDPNet (BASELINE config #2)
    cfg = W.NAVDPNET_CFG
    sd, inp = W.navdpnet_state_dict(seed=21), W.navdpnet_inputs(B, seed=21)
    fin, crit, ys, yc = [], [], [], []
    for b in range(B):
        a = (sd, inp["goal"][b:b + 1], inp["images"][b:b + 1], inp["depths"][b:b + 1], inp["x_init"][b:b + 1], inp["step_noise"][:, b:b + 1], cfg)
        with torch.no_grad():
            _, _, f32, c32, _ = o_navdp.navdpnet_pointgoal(*a, return_all=True)
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            _, _, f16, c16, _ = o_navdp.navdpnet_pointgoal(*a, return_all=True)
        fin.append(f32[0].float()), crit.append(c32[0].float()), ys.append(_stats(f16[0], f32[0])), yc.append(_stats(c16[0], c32[0]))
        print(f"navdpnet env {b}: bf16 mean {ys[-1][0]:.3e} p99 {ys[-1][1]:.3e} max {ys[-1][2]:.3e}  [{time.time() - t0:.0f}s]", flush=True)
    return dict(seed=21, samples=torch.stack(fin), critic=torch.stack(crit), yard=torch.tensor(ys), yard_critic=torch.tensor(yc))


def _n1_navdp(t0):
    # ---- N1 NavDP head (navdp_async)
    cfg = W.N1_NAVDP_CFG
    sd, inp = W.n1_navdp_state_dict(seed=23), W.n1_navdp_inputs(B, seed=23)
    fin, ys = [], []
    for b in range(B):
        a = (sd, inp["vlm_tokens"][b:b + 1], inp["images"][b:b + 1], inp["depths"][b:b + 1], inp["x_init"][b:b + 1], inp["step_noise"][:, b:b + 1], cfg)
        with torch.no_grad():
            f32 = o_navdp.n1_navdp_async(*a)
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            f16 = o_navdp.n1_navdp_async(*a)
        fin.append(f32[0].float()), ys.append(_stats(f16[0], f32[0]))
        print(f"n1_navdp env {b}: bf16 mean {ys[-1][0]:.3e} p99 {ys[-1][1]:.3e} max {ys[-1][2]:.3e}  [{time.time() - t0:.0f}s]", flush=True)
    return dict(seed=23, trajectories=torch.stack(fin), yard=torch.tensor(ys))


if __name__ == "__main__":
    main()
