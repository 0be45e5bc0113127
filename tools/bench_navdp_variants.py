"""NavDPNet (config #2) and N1 NavDP head call over 64 envs (graph): the library's tile choice vs the round-3 choice (tiled kernels) for the
wide K = 384 GEMMs that now run on the row-panel kernels (biased q|k|v and GELU FFN projections of the 16-layer decoder). Usage: python tools/bench_navdp_variants.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from internnav_amd import ops, runtime, synthetic  # noqa: E402
from internnav_amd.navdp import NavDPNet  # noqa: E402

dev = torch.device("cuda:0")
B = 64
cfg = synthetic.NAVDPNET_CFG
net = NavDPNet(synthetic.navdpnet_state_dict(seed=0), cfg, dev, max_envs=B)
inp = {k: v.to(dev) for k, v in synthetic.navdpnet_inputs(B, seed=0).items()}
orig = ops.linear


def tiled(x, w, *a, **k):       # the round-3 selection: 128 x 256 single-buffer tiles for the wide K = 384 GEMMs
    if w.shape[1] == 384 and w.shape[0] >= 1024 and x.numel() // 384 >= 16384 and not k.get("force_cfg"):
        k["force_cfg"] = 26
    return orig(x, w, *a, **k)


ref = None
for name, fn in (("row-panel (library choice)", orig), ("tiled cfg 26 (round 3)", tiled), ("row-panel again", orig)):
    ops.linear = fn
    import internnav_amd.navdp as nv
    nv.ops.linear = fn
    call = lambda: net.predict_pointgoal_batch_action_vel(inp["goal"], inp["images"], inp["depths"], inp["x_init"], inp["step_noise"])  # noqa: E731
    call()
    torch.cuda.synchronize()
    out = net.sample.clone()
    if ref is None:
        ref = out
    g = runtime.GraphedCall(call, {})
    g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{name:28s} NavDPNet call over {B} envs (graph): {ms:7.2f} ms = {B / ms * 1e3:6.1f} policy steps/s   max|diff| {(out - ref).abs().max().item():.3e}", flush=True)
    del g
ops.linear = orig
