import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from internnav_amd import ops
dev = torch.device("cuda:0")
def run(B, L, H, Hkv, D, causal, n=5):
    q = torch.randn(B, L, H, D, device=dev).to(torch.bfloat16)
    k = torch.randn(B, L, Hkv, D, device=dev).to(torch.bfloat16)
    v = torch.randn(B, L, Hkv, D, device=dev).to(torch.bfloat16)
    o = torch.empty_like(q)
    for _ in range(n):
        ops.attention(q, k, v, causal=causal, out=o)
    torch.cuda.synchronize()
run(7, 920, 28, 4, 128, True)
run(7, 920, 28, 4, 128, False)
run(28, 784, 16, 16, 80, False)
print("ok")
