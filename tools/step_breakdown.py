"""Per-phase timing of one n1_dual bench step on the GPU box: S2 graph (6/7 envs), S1 graph (64 envs), D2H + host post-processing."""
import sys, time
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

a = SimpleNamespace(envs=64)
dev = torch.device("cuda:0")
wl = bench.N1Dual(a, dev, 0)
wl.capture()


def t(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for m, s in wl.s2.items():
    print(f"S2 graph, {m} envs: {t(s['graph']):.1f} ms")
print(f"S1 graph, 64 envs: {t(wl.s1_graph):.1f} ms")
traj = wl.s1_graph()
torch.cuda.synchronize()
t0 = time.perf_counter(); tc = traj.cpu(); t1 = time.perf_counter()
for b in range(64):
    [x for x in wl.traj_to_actions(tc[b]) if x != 0][:4]
t2 = time.perf_counter()
print(f"D2H {1e3*(t1-t0):.2f} ms, traj_to_actions x64 {1e3*(t2-t1):.2f} ms")
s1 = wl.model.s1
print(f"S1 encode_condition: {t(lambda: s1.encode_condition(64, wl.latent_table, wl.images_dp)):.1f} ms (eager)")
