"""The decode chain inside the step vs alone, per kernel, from two kernel dumps of tools/dump_kernels.py:
  step.csv  = last steps of tools/profile_step.py under `rocprofv3 --kernel-trace` (graph-replayed timed steps)
  alone.csv = tools/profile_phases.py s2 N (eager System-2 calls on one stream)
Prints span / kernel time / gaps of the last decode chain of each, and per kernel (name, grid) its count and average duration.
Usage: python tools/decode_chain_in_step.py step.csv alone.csv"""
import collections
import csv
import sys


def load(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = float(r["start_us"]), float(r["end_us"])
    return rows


def last_chain(rows, queue=None):
    q = rows if queue is None else [r for r in rows if r["queue_id"] == queue]
    last_tiled = max(i for i, r in enumerate(q) if "gemm_bf16_pp" in r["name"] or "gemm_bf16_w4" in r["name"])
    chain = q[last_tiled + 1:]
    cut = next((i for i, r in enumerate(chain) if "dit_attn" in r["name"] or "rowpanel" in r["name"] or "embed3" in r["name"] or "rowchain" in r["name"]), len(chain))
    return chain[:cut]


def report(label, dec):
    gaps = [max(0.0, dec[k + 1]["s"] - dec[k]["e"]) for k in range(len(dec) - 1)]
    print(f"{label}: {len(dec)} kernels, span {(dec[-1]['e'] - dec[0]['s']) / 1e3:.2f} ms, kernel time {sum(r['e'] - r['s'] for r in dec) / 1e3:.2f} ms, gaps {sum(gaps) / 1e3:.2f} ms")
    by = collections.defaultdict(list)
    for r in dec:
        by[(r["name"][:58], r["grid_x"])].append(r["e"] - r["s"])
    return by


def main():
    step, alone = load(sys.argv[1]), load(sys.argv[2])
    main_q = collections.Counter(r["queue_id"] for r in step if "gemm_skinny" in r["name"]).most_common(1)[0][0]
    a = report("inside the step (beside System-1, graph replay)", last_chain(step, main_q))
    b = report("alone (eager, one stream)", last_chain(alone))
    print(f"{'kernel':58s} {'grid':>8s} {'count':>6s} {'alone avg us':>13s} {'in-step avg us':>15s} {'ratio':>6s} {'in-step total ms':>17s}")
    for k, v in sorted(a.items(), key=lambda kv: -sum(kv[1])):
        w = b.get(k)
        al = sum(w) / len(w) if w else float("nan")
        st = sum(v) / len(v)
        print(f"{k[0]:58s} {k[1]:>8s} {len(v):6d} {al:13.1f} {st:15.1f} {st / al if w else float('nan'):6.2f} {sum(v) / 1e3:17.2f}")


if __name__ == "__main__":
    main()
