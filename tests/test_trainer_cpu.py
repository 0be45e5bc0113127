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


def _cpu_adamw(p, g, m, v, lr, beta1, beta2, eps, wd, step, p_bf16=None, sumsq_parts=None, max_norm=0.0, grad_scale=1.0, norm_out=None, zero_grad=False):
    """torch restatement of ina_adamw's contract (csrc/train.hip adamw_kernel) for the CPU-only plumbing test below - TEST CODE."""
    total = float(sumsq_parts.sum().sqrt()) * grad_scale if sumsq_parts is not None else 0.0
    clip = min(1.0, max_norm / (total + 1e-6)) if max_norm > 0 else 1.0
    gg = g * (grad_scale * clip)
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(gg, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
    p.addcdiv_(m, v.sqrt() / (1 - beta2 ** step) ** 0.5 + eps, value=-lr / (1 - beta1 ** step))
    if p_bf16 is not None:
        p_bf16.copy_(p)
    if norm_out is not None:
        norm_out.fill_(total)
    if zero_grad:
        g.zero_()


def _zero2_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from internnav_amd import train_ops as T
    from internnav_amd.sft import ParamStore
    from internnav_amd.trainer import InternVLAN1SftTrainer, shard_bounds

    T.adamw = _cpu_adamw                                                     # the GPU kernels are not under test here: the collectives are
    T.sumsq_parts = lambda flat, width=1024: flat.view(-1, width).pow(2).sum(0)
    results = []
    for zero2 in (False, True):
        g = torch.Generator().manual_seed(0)
        # 3 blocks of 1024 after padding: with world 2 the shards are 2 + 1 blocks - the uneven case
        P = ParamStore({"a": torch.randn(300, 7, generator=g), "b": torch.randn(500, generator=g), "latent_queries": torch.randn(1, 4, 16, generator=g)}, "cpu")
        tr = object.__new__(InternVLAN1SftTrainer)
        tr.P, tr.world, tr.rank, tr.pg, tr.zero2, tr.device = P, world, rank, None, zero2, torch.device("cpu")
        tr.total_steps, tr.lr, tr.min_lr, tr.warmup_steps, tr.wd, tr.max_norm, tr.betas, tr.eps = 100, 1e-2, 1e-3, 0, 0.01, 1.0, (0.9, 0.999), 1e-8
        tr.grad_norm, tr.step_idx = torch.zeros(1), 0
        if zero2:
            P.shard_moments(*shard_bounds(P.numel, world, rank))            # what the constructor does under ZeRO-2
            assert P.m.numel() == shard_bounds(P.numel, world, rank)[1] - shard_bounds(P.numel, world, rank)[0] < P.numel

        class _E:
            latent_q = torch.zeros(4, 16, dtype=torch.bfloat16)
        tr.engine = _E()
        used = 300 * 7 + 4 + 500 + 4 + 64          # entries incl. alignment gaps (gaps get gradient 0 like in a real step)
        for step in range(3):
            gg = torch.Generator().manual_seed(100 * step + rank)
            P.g32.zero_()
            for k in P.index:
                P.grad(k).copy_(torch.randn(P.grad(k).shape, generator=gg) * (5.0 if step == 1 else 0.05))
            tr.reduce_gradients()
            tr.optimizer_step()
        results.append((P.p32.clone(), P.p16.clone(), tr.grad_norm.item(), tr.engine.latent_q.clone(), float(P.g32.abs().max())))
    (p_ar, p16_ar, n_ar, lq_ar, g_ar), (p_z2, p16_z2, n_z2, lq_z2, g_z2) = results
    q.put((rank, p_ar[:64].tolist(), float((p_ar - p_z2).abs().max()), bool(torch.equal(p16_ar, p16_z2)), n_ar, n_z2, bool(torch.equal(lq_ar, lq_z2)), g_ar, g_z2))
    dist.barrier()
    dist.destroy_process_group()


def test_zero2_equals_all_reduce_world2_gloo():
    """reduce-scatter -> sharded clip + update -> all-gather of master and working weights lands on the all-reduce (replicated update) result,
    identical on both ranks; uneven shards (3 blocks over 2 ranks). The update arithmetic is a torch stand-in for the HIP kernel
    (tests/test_train_ops_gpu.py::test_adamw_matches_torch covers the kernel itself)."""
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_zero2_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert len(res) == 2
    for rank, head, diff, same16, n_ar, n_z2, same_lq, g_ar, g_z2 in res:
        assert head == res[0][1]
        assert diff <= 1e-7 and same16 and same_lq
        assert abs(n_ar - n_z2) <= 1e-5 * n_ar and abs(n_ar - res[0][4]) <= 1e-6 * n_ar
        assert g_ar == 0.0 and g_z2 == 0.0


def _zero2_ckpt_worker(rank, world, port, q, tmp):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from internnav_amd import train_ops as T
    from internnav_amd.sft import ParamStore
    from internnav_amd.trainer import InternVLAN1SftTrainer, shard_bounds

    T.adamw = _cpu_adamw
    T.sumsq_parts = lambda flat, width=1024: flat.view(-1, width).pow(2).sum(0)

    def make(seed):
        g = torch.Generator().manual_seed(seed)
        P = ParamStore({"a": torch.randn(300, 7, generator=g), "b": torch.randn(500, generator=g), "latent_queries": torch.randn(1, 4, 16, generator=g)}, "cpu")
        tr = object.__new__(InternVLAN1SftTrainer)
        tr.P, tr.world, tr.rank, tr.pg, tr.zero2, tr.device, tr.system1 = P, world, rank, None, True, torch.device("cpu"), "nextdit_async"
        tr.total_steps, tr.lr, tr.min_lr, tr.warmup_steps, tr.wd, tr.max_norm, tr.betas, tr.eps = 100, 1e-2, 1e-3, 0, 0.01, 1.0, (0.9, 0.999), 1e-8
        tr.grad_norm, tr.step_idx, tr.micro_idx, tr.seed = torch.zeros(1), 0, 0, 0
        tr.gen_dev, tr.gen_cpu = torch.Generator().manual_seed(rank), torch.Generator().manual_seed(rank + 50)

        class _E:
            latent_q = torch.zeros(4, 16, dtype=torch.bfloat16)
        tr.engine = _E()
        P.shard_moments(*shard_bounds(P.numel, world, rank))
        return tr

    def step(tr, k):
        gg = torch.Generator().manual_seed(100 * k + rank)
        tr.P.g32.zero_()
        for name in tr.P.index:
            tr.P.grad(name).copy_(torch.randn(tr.P.grad(name).shape, generator=gg) * 0.05)
        tr.reduce_gradients()
        tr.optimizer_step()

    a = make(0)                      # uninterrupted: 4 steps
    for k in range(4):
        step(a, k)
    b = make(0)                      # 2 steps, checkpoint (collective), resume in a fresh trainer with other initial weights, 2 more steps
    for k in range(2):
        step(b, k)
    path = os.path.join(tmp, f"ck_rank{rank}.pt")
    b.save_checkpoint(path)
    dist.barrier()                   # rank 1 reads rank 0's file below
    ck = torch.load(path, weights_only=True)
    lo1, hi1 = shard_bounds(b.P.numel, world, 1)
    # the file of EVERY rank holds the moments of EVERY shard (rank 0's file used to lack rank 1's slice): compare with the owners' values
    full_m = b._gather_flat(b.P.m)
    owned_by_1 = float(full_m[lo1:hi1].abs().sum())
    in_file = sum(float(ck["store"]["exp_avg"][k].abs().sum()) for k in b.P.index)
    c = make(7)
    c.load_checkpoint(os.path.join(tmp, "ck_rank0.pt") if rank == 1 else path)      # rank 1 resumes from RANK 0's file
    for k in range(2, 4):
        step(c, k)
    # rng streams after a resume: a rank that loaded ITS OWN file continues its stream (next draw == the uninterrupted trainer's next draw);
    # a rank that loaded another rank's file starts a FRESH stream keyed by (seed, rank, step) - not a replay of the draws from step 0
    # (what a constructor-seeded generator would produce) and the same again on a second resume from that file
    fresh = make(7)
    first_ever = torch.rand(4, generator=fresh.gen_cpu)
    nxt = torch.rand(4, generator=c.gen_cpu)
    if rank == 0:
        rng_ok = torch.equal(nxt, torch.rand(4, generator=b.gen_cpu))
    else:
        d = make(9)
        d.load_checkpoint(os.path.join(tmp, "ck_rank0.pt"))
        rng_ok = (not torch.equal(nxt, first_ever)) and torch.equal(nxt, torch.rand(4, generator=d.gen_cpu))
    q.put((rank, float((a.P.p32 - c.P.p32).abs().max()), bool(torch.equal(a.P.m, c.P.m) and torch.equal(a.P.v, c.P.v)), owned_by_1 > 0,
           abs(in_file - float(full_m.abs().sum())) <= 1e-4 * in_file, c.step_idx, c.P.step_count, bool(rng_ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_zero2_checkpoint_resume_world2_gloo(tmp_path):
    """under ZeRO-2 each rank updates the Adam moments of its own shard only. save_checkpoint all-gathers them, so any rank's
    file is complete, and a run resumed from it (even from the OTHER rank's file) continues exactly like an uninterrupted run."""
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_zero2_ckpt_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert len(res) == 2
    for rank, diff, same_moments, other_shard_nonzero, file_complete, step_idx, step_count, rng_ok in res:
        assert diff == 0.0 and same_moments and other_shard_nonzero and file_complete and step_idx == 4 and step_count == 4
        assert rng_ok, f"rank {rank}: generator streams after the resume"
