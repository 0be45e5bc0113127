"""CPU: host logic of the SFT trainer - learning-rate schedule against transformers, ZeRO-2 shard arithmetic, and the data-parallel
reduction of the flat gradient bucket over a world_size-2 gloo group."""
import os
import socket

import pytest
import torch


def test_cosine_with_min_lr_matches_transformers():
    from transformers.optimization import get_cosine_with_min_lr_schedule_with_warmup

    from internnav_amd.trainer import cosine_with_min_lr

    for total, warm in ((1000, 3), (37, 1), (200, 0)):
        p = [torch.nn.Parameter(torch.zeros(1))]
        opt = torch.optim.SGD(p, lr=1e-4)
        sch = get_cosine_with_min_lr_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total, min_lr=1e-5)
        for s in range(total):
            assert abs(opt.param_groups[0]["lr"] - cosine_with_min_lr(s, total, warm, 1e-4, 1e-5)) < 1e-12, (total, warm, s)
            opt.step()
            sch.step()


def test_shard_bounds_partition_the_flat_buffer():
    from internnav_amd.trainer import shard_bounds

    for numel, world in ((10240, 3), (2048, 8), (90634240, 8), (1024, 1)):
        cover = 0
        prev = 0
        for r in range(world):
            lo, hi = shard_bounds(numel, world, r)
            assert lo == prev and lo % 1024 == 0 and hi % 1024 == 0 and hi >= lo
            cover += hi - lo
            prev = hi
        assert cover == numel and prev == numel
        assert all(shard_bounds(numel, world, r)[1] - shard_bounds(numel, world, r)[0] <= shard_bounds(numel, world, 0)[1] for r in range(world))


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from internnav_amd.trainer import InternVLAN1SftTrainer

    class _P:
        numel = 4096
        g32 = torch.arange(4096, dtype=torch.float32) * (rank + 1)

    tr = object.__new__(InternVLAN1SftTrainer)
    tr.P, tr.world, tr.rank, tr.pg, tr.zero2, tr.device = _P(), world, rank, None, False, torch.device("cpu")
    tr.reduce_gradients()
    ok = torch.equal(tr.P.g32, torch.arange(4096, dtype=torch.float32) * sum(r + 1 for r in range(world)))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_all_reduce_world2_gloo():
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
