"""Per-kernel durations inside the concurrent phase of the last graph-replayed step (everything that starts after the step's last tiled prefill
GEMM), grouped by hardware queue and kernel name - to compare with the same kernels' durations in a call that runs alone
(profiles/*_s1_3calls_kernel_stats.txt, *_s2_2calls_kernel_stats.txt). Input: a kernel dump of tools/dump_kernels.py taken from
`rocprofv3 --kernel-trace -- python tools/profile_step.py`. Usage: python tools/phase_kernels_in_step.py step.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = float(r["start_us"]), float(r["end_us"])
rows.sort(key=lambda r: r["s"])
t0 = max(r["e"] for r in rows if "gemm_bf16_pp" in r["name"] or "gemm_bf16_w4" in r["name"])
phase = [r for r in rows if r["s"] >= t0]
print(f"# concurrent phase of the last step: {len(phase)} kernels over {(max(r['e'] for r in phase) - t0) / 1e3:.1f} ms")
by = collections.defaultdict(list)
for r in phase:
    by[(r["queue_id"], r["name"][:90])].append(r)
qs = sorted({k[0] for k in by})
for q in qs:
    ks = [(k, v) for k, v in by.items() if k[0] == q]
    tot = sum(r["e"] - r["s"] for _, v in ks for r in v)
    first, last = min(r["s"] for _, v in ks for r in v), max(r["e"] for _, v in ks for r in v)
    print(f"queue {q}: {sum(len(v) for _, v in ks)} kernels, kernel time {tot / 1e3:.1f} ms, from {(first - t0) / 1e3:.1f} to {(last - t0) / 1e3:.1f} ms")
    for (qq, name), v in sorted(ks, key=lambda kv: -sum(r["e"] - r["s"] for r in kv[1]))[:12]:
        d = [r["e"] - r["s"] for r in v]
        print(f"    {name:92s} n {len(v):5d}  avg {sum(d) / len(d):8.1f} us  total {sum(d) / 1e3:7.2f} ms")
