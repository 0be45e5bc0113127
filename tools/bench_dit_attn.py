"""Fused NextDiT attention stage on the bench shape (64 envs x 32 samples x 32 tokens). Usage: python tools/bench_dit_attn.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from internnav_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
envs, S, T, Lz, heads, D = 64, 32, 32, 36, 6, 384
x = torch.randn(envs * S * T, 4 * D, device=dev).to(torch.bfloat16)
kv2 = torch.randn(envs, Lz, 2, heads, 64, device=dev).to(torch.bfloat16)
norms = [(torch.ones(D, device=dev), torch.zeros(D, device=dev)) for _ in range(3)]
gate = torch.randn(heads, device=dev)
v2t = torch.empty(envs, heads, 64, 64, dtype=torch.bfloat16, device=dev)
ops.dit_v2t(kv2, heads, v2t)
out = torch.empty(envs * S * T, D, dtype=torch.bfloat16, device=dev)
fn = lambda: ops.dit_attention(x, out, norms, kv2, v2t, gate, T=T, seq_per_env=S, heads=heads)
for _ in range(5):
    fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50):
    fn()
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) / 50 * 1e3
mb = envs * S * T * D * 2 * 5 / 1e6
print(f"dit_attention {us:.1f} us   {mb / us / 1e3:.2f} TB/s algorithmic ({mb:.0f} MB)")
