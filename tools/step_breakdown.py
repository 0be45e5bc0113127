"""Per-phase timing of one n1_dual bench step on the GPU box: System-2 prefill / decode graphs (6/7 envs), System-1 graphs
(64 / 57 / 7 envs), the decode || System-1 overlap, D2H + host post-processing."""
import sys, time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

kw = {}
argv = sys.argv[1:]
for i, x in enumerate(argv):        # e.g. `python tools/step_breakdown.py --s1-tail-envs 0 --no-row-chain --dit-ffn 1024`
    if x in ("--dit-ffn",):
        kw[x[2:].replace("-", "_")] = int(argv[i + 1])
    elif x in ("--no-row-chain", "--no-fuse-decode-rope", "--no-frag-weights", "--no-split-prefill"):
        kw[x[2:].replace("-", "_")] = True
a = bench.default_args(**kw)
print("# variant:", kw or "default")
dev = torch.device("cuda:0")
wl = bench.N1Dual(a, dev, 0)
wl.capture()


def t(fn, n=5):
    fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for m, s in wl.s2.items():
    print(f"S2 whole graph, {m} envs: {t(s['graph']):.1f} ms   prefill graph {t(wl.gP[m]):.1f} ms   decode+latents graph {t(wl.gD[m]):.1f} ms")
print(f"S1 graph, 64 envs: {t(wl.s1_graph):.1f} ms")
for nA, g in wl.gA.items():
    print(f"S1 graph, {nA} envs{' (without the look-down encoder pass)' if wl.merge_images else ''}: {t(g):.1f} ms")
for (nA, m_), g in wl.gAimg.items():
    print(f"look-down encoder pass (DINOv2, MemoryEncoder, QFormer) over {nA} + {m_} envs: {t(g):.1f} ms")
for m, g in wl.gB.items():
    print(f"S1 graph (small engine{', from the projected latents on' if wl.merge_images else ''}), {m} envs: {t(g):.1f} ms")
m = max(wl.mb)
X = 0
nA = 64 - m - X
mB = m + X


def side_call():
    if wl.merge_images:
        wl.gAimg[(nA, mB)]()
    wl.gA[nA]()


def both():
    wl.side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(wl.side):
        side_call()
    wl.gD[m]()
    torch.cuda.current_stream().wait_stream(wl.side)


print(f"decode+latents ({m} envs) || S1 ({nA} envs): {t(both):.1f} ms")


def both3():
    wl.side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(wl.side):
        side_call()
    wl.gD[m]()
    wl.gB[mB]()
    torch.cuda.current_stream().wait_stream(wl.side)


print(f"(decode+latents, S1 of {mB} envs) || S1 ({nA} envs): {t(both3):.1f} ms")
# timeline of the concurrent phase: when does each chain end?
ev = {k: torch.cuda.Event(enable_timing=True) for k in ("start", "dec", "s1b", "s1a")}
torch.cuda.synchronize()
main = torch.cuda.current_stream()
ev["start"].record(main)
wl.side.wait_stream(main)
with torch.cuda.stream(wl.side):
    side_call()
    ev["s1a"].record(wl.side)
wl.gD[m]()
ev["dec"].record(main)
wl.gB[mB]()
ev["s1b"].record(main)
torch.cuda.synchronize()
print("concurrent phase timeline: decode+latents ends at %.1f ms, S1(small) at %.1f ms (main stream); S1(%d envs, side stream) at %.1f ms"
      % (ev["start"].elapsed_time(ev["dec"]), ev["start"].elapsed_time(ev["s1b"]), nA, ev["start"].elapsed_time(ev["s1a"])))
print(f"full step (bench schedule): {t(lambda: wl.step(0)):.1f} ms")
traj = wl.s1_graph()
torch.cuda.synchronize()
t0 = time.perf_counter(); tc = traj.cpu(); t1 = time.perf_counter()
for b in range(64):
    [x for x in wl.traj_to_actions(tc[b]) if x != 0][:4]
t2 = time.perf_counter()
print(f"D2H {1e3*(t1-t0):.2f} ms, traj_to_actions x64 {1e3*(t2-t1):.2f} ms")
s1 = wl.model.s1
print(f"S1 encode_condition: {t(lambda: s1.encode_condition(64, wl.latent_table, wl.images_dp)):.1f} ms (eager)")
