"""CPU: the two multi-process workers of tests/test_rccl_gpu.py (per-step action all-gather + padded metric gather; the SFT trainer's
flat-bucket ZeRO-2 vs all-reduce equivalence) executed here with backend `gloo`, world size 2 - the RCCL tests themselves need >= 2
GPUs and have never had a box to run on, so at least the very same worker code runs on every CPU round."""
import socket

import torch

from tests import test_rccl_gpu as R


def _run(target, world=2):
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=target, args=(r, world, port, q, "gloo")) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in ps:
        p.join(60)
    return res


def test_action_all_gather_worker_on_gloo():
    world = 2
    res = _run(R._worker, world)
    base = torch.arange(64 * 4, dtype=torch.int32).view(64, 4) % 4
    for r, g, m in res:
        g = torch.tensor(g)
        assert g.shape == (world, 64, 4)
        for k in range(world):
            exp = base.clone()
            exp[:, 0] = k
            assert torch.equal(g[k], exp)
        assert m == [float(v + 10 * k) for k in range(world) for v in range(3 + k)]


def test_sft_zero2_worker_on_gloo():
    res = _run(R._sft_worker, 2)
    for r, head, diff, n_ar, n_z2, dlq in res:
        assert head == res[0][1]
        assert diff <= 1e-6 and dlq == 0.0
        assert abs(n_ar - n_z2) <= 1e-4 * n_ar and n_ar == res[0][3]
