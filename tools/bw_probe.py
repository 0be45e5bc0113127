import torch, time
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e-3
for mb in (50, 201, 800, 3200):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev); b = torch.empty_like(a)
    tz = t(lambda: a.zero_()); tc = t(lambda: b.copy_(a))
    s = torch.empty(n // 2, dtype=torch.float32, device=dev)
    tr = t(lambda: s.sum())
    print(f"{mb} MB: write-only {mb/1024/tz/1e3*1.048576:.2f} TB/s  copy(r+w) {2*mb/1024/tc/1e3*1.048576:.2f} TB/s  read-only(sum f32) {mb/1024/tr/1e3*1.048576:.2f} TB/s")
