"""GPU. The multi-rank tests need >= 2 devices (skipped on the 1-GPU test box); the one-rank test runs everywhere: the per-step action exchange over the `nccl` backend (= RCCL over xGMI on
ROCm), one process per GPU as bench.py / the evaluator launch it. The gathered tensor must be bit-equal to what the gloo CPU test of
the same helper produces (tests/test_host_logic.py::test_data_parallel_helpers_gloo_world2): integer action ids, rank-major."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _device(rank, backend):
    """one GPU per rank under RCCL; the gloo twin of these workers (tests/test_rccl_workers_gloo_cpu.py) runs the same code on the CPU."""
    if backend == "nccl":
        torch.cuda.set_device(rank)
        return torch.device("cuda", rank)
    return torch.device("cpu")


def _worker(rank, world, port, q, backend="nccl"):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from internnav_amd import dist as D

    dev = _device(rank, backend)
    r, _, w = D.init_distributed(backend, device=dev)
    acts = (torch.arange(64 * 4, dtype=torch.int32, device=dev).view(64, 4) % 4) + 0 * r
    acts[:, 0] = r
    g = D.all_gather_actions(acts)
    m = D.all_gather_metrics(torch.arange(3 + r, dtype=torch.float32, device=dev) + 10 * r)
    q.put((r, g.cpu().tolist(), m.cpu().tolist()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs on one node (RCCL over xGMI)")
def test_action_all_gather_over_rccl():
    import torch.multiprocessing as mp

    world = min(torch.cuda.device_count(), 8)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in ps:
        p.join(120)
    base = (torch.arange(64 * 4, dtype=torch.int32).view(64, 4) % 4)
    for r, g, m in res:
        g = torch.tensor(g)
        assert g.shape == (world, 64, 4)
        for k in range(world):
            exp = base.clone()
            exp[:, 0] = k
            assert torch.equal(g[k], exp)                      # rank-major, bit-equal integers on every rank
        assert m == [float(v + 10 * k) for k in range(world) for v in range(3 + k)]


def _sft_worker(rank, world, port, q, backend="nccl"):
    """the SFT trainer's flat-bucket reduction + fused AdamW: ZeRO-2 (reduce-scatter, sharded update, all-gather) must land on exactly the
    all-reduce (replicated update) weights, and every rank must hold the same ones."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    dev = _device(rank, backend)
    dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {"rank": rank, "world_size": world}))
    from internnav_amd.sft import ParamStore
    from internnav_amd.trainer import InternVLAN1SftTrainer, shard_bounds

    if backend != "nccl":            # CPU twin: torch stand-ins for the two HIP kernels of the optimiser (the collectives are under test)
        from internnav_amd import train_ops as T
        from tests import _cpu_kernels as K

        T.adamw, T.sumsq_parts = K.adamw, K.sumsq_parts

    out = []
    for zero2 in (False, True):
        g = torch.Generator().manual_seed(0)
        P = ParamStore({"a": torch.randn(3000, 7, generator=g), "b": torch.randn(5000, generator=g), "latent_queries": torch.randn(1, 4, 64, generator=g)}, dev)
        tr = object.__new__(InternVLAN1SftTrainer)
        tr.P, tr.world, tr.rank, tr.pg, tr.zero2, tr.device = P, world, rank, None, zero2, dev
        tr.total_steps, tr.lr, tr.min_lr, tr.warmup_steps, tr.wd, tr.max_norm, tr.betas, tr.eps = 100, 1e-2, 1e-3, 0, 0.01, 1.0, (0.9, 0.999), 1e-8
        tr.grad_norm, tr.step_idx = torch.zeros(1, device=dev), 0
        if zero2:
            P.shard_moments(*shard_bounds(P.numel, world, rank))          # as the trainer's constructor does under ZeRO-2

        class _E:
            latent_q = torch.zeros(4, 64, dtype=torch.bfloat16, device=dev)
        tr.engine = _E()
        for step in range(3):
            gg = torch.Generator().manual_seed(100 * step + rank)
            P.g32[: 3000 * 7 + 5000 + 8].copy_(torch.randn(3000 * 7 + 5000 + 8, generator=gg))
            tr.reduce_gradients()
            tr.optimizer_step()
        out.append((P.p32.cpu().clone(), tr.grad_norm.item(), tr.engine.latent_q.float().cpu().clone()))
    q.put((rank, out[0][0].tolist()[:64], float((out[0][0] - out[1][0]).abs().max()), out[0][1], out[1][1], float((out[0][2] - out[1][2]).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs on one node (RCCL over xGMI)")
def test_sft_flat_bucket_zero2_equals_all_reduce_over_rccl():
    import torch.multiprocessing as mp

    world = min(torch.cuda.device_count(), 8)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_sft_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in ps:
        p.join(120)
    for r, head, diff, n_ar, n_z2, dlq in res:
        assert head == res[0][1]                       # identical weights on every rank
        assert diff <= 1e-6 and dlq == 0.0             # sharded update == replicated update
        assert abs(n_ar - n_z2) <= 1e-4 * n_ar and n_ar == res[0][3]


def _worker_one(port, q):
    """a ONE-rank process group on the `nccl` backend: the helpers short-circuit at world 1, so the collectives are called directly here"""
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    acts = (torch.arange(64 * 4, dtype=torch.int32, device=dev).view(64, 4) % 4)
    out = torch.full_like(acts, -1)
    dist.all_gather_into_tensor(out, acts)                       # the per-step exchange of bench.py / dist.all_gather_actions
    t = torch.tensor([3.0], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                     # the max-over-ranks timing of bench.py
    dist.barrier()
    torch.cuda.synchronize()
    q.put((str(dist.get_backend()), dist.get_world_size(), out.cpu().tolist(), float(t.item())))
    dist.destroy_process_group()


def test_rccl_communicator_with_one_rank_on_the_test_gpu():
    """the single test GPU can still LOAD RCCL: a one-rank `nccl` process group (communicator creation), the per-step
    all_gather_into_tensor of the action table, the timing all-reduce, a barrier, teardown - so a broken librccl / IPC environment shows up
    here and not first at the driver's 8-GPU run. (No scaling claim: one rank.)"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_one, args=(31000 + (os.getpid() % 2000), q))
    p.start()
    backend, world, g, t = q.get(timeout=300)
    p.join(120)
    assert p.exitcode == 0 and backend == "nccl" and world == 1 and t == 3.0
    assert torch.equal(torch.tensor(g, dtype=torch.int32), torch.arange(64 * 4, dtype=torch.int32).view(64, 4) % 4)
