"""System-1 call (64 / 57 envs, graph) by variant of the row-local part of the NextDiT blocks: round-4 launches (GEMM, norm, GEMM), the round-5 row
chain (csrc/dit_rowchain.hip, 128- and 256-row panels), the older fused row-norm / FFN kernels. Usage: python tools/bench_s1_variants.py [envs ...]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from internnav_amd import runtime, synthetic  # noqa: E402
from internnav_amd.nextdit import NextDiTSystem1  # noqa: E402

dev = torch.device("cuda:0")
cfg = synthetic.N1_NEXTDIT_CFG
sd = synthetic.n1_nextdit_state_dict(seed=0)
for B in ([int(x) for x in sys.argv[1:]] or [64, 57]):
  inp = synthetic.n1_nextdit_inputs(B, seed=0)
  lat, img, x0 = inp["traj_latents"].to(dev, torch.bfloat16), inp["images"].to(dev), inp["x_init"].to(dev)
  ref = None
  for name, kw in (("unfused", dict(row_chain=False)), ("row_chain", dict(row_chain=True))):
    chain_stats = kw.pop("chain_stats", True)
    eng = NextDiTSystem1(sd, cfg, dev, max_envs=B, **kw)
    eng.chain_stats = chain_stats
    out = eng.generate_traj(lat, img, x0).clone()
    torch.cuda.synchronize()
    if ref is None:
        ref = out
    g = runtime.GraphedCall(lambda: eng.generate_traj(lat, img, x0), {})
    g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name:22s} S1 call over {B} envs (graph): {ms:7.2f} ms   max|diff| vs unfused {(out - ref).abs().max().item():.3e}", flush=True)
    del eng, g
