"""GPU, >= 2 devices only (skipped on the 1-GPU test box): the per-step action exchange over the `nccl` backend (= RCCL over xGMI on
ROCm), one process per GPU as bench.py / the evaluator launch it. The gathered tensor must be bit-equal to what the gloo CPU test of
the same helper produces (tests/test_host_logic.py::test_data_parallel_helpers_gloo_world2): integer action ids, rank-major."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from internnav_amd import dist as D

    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    r, _, w = D.init_distributed("nccl", device=dev)
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
