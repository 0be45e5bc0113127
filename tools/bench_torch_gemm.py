"""What the vendor library reaches on the same GEMM shapes (torch.matmul -> hipBLASLt / rocBLAS): a ceiling check for the hand-written kernels.
NOT used by the engine (measurement only). Usage: python tools/bench_torch_gemm.py"""
import torch

dev = "cuda:0"
shapes = [(8192, 8192, 8192), (6440, 4608, 3584), (6440, 37888, 3584), (6440, 3584, 18944), (21952, 3840, 1280), (65536, 1536, 384), (65536, 2048, 384), (65536, 384, 1024)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        torch.matmul(a, w.t())
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10 * 1e-3
    print(f"{M:6d} {N:6d} {K:6d}  torch.matmul {2.0 * M * N * K / t * 1e-12:7.1f} TF/s  ({t * 1e6:7.1f} us)", flush=True)
