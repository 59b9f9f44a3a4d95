"""world_size=2 gloo test of the multi-GPU path's control logic (no data-path collective
exists: SURVEY.md section 8e)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lmcache_amd.distributed import max_over_ranks, shard_chunks, sum_over_ranks
    nchunks = 67
    mine = shard_chunks(nchunks, rank, world)
    # every chunk is owned exactly once: gather the ownership sets
    owned = [None] * world
    dist.all_gather_object(owned, mine)
    flat = sorted(i for part in owned for i in part)
    assert flat == list(range(nchunks))
    assert abs(len(mine) - nchunks / world) < 1
    dev = torch.device("cpu")
    elapsed = 0.010 * (rank + 1)
    mx = max_over_ranks(elapsed, dev)
    units = sum_over_ranks(float(len(mine)), dev)
    assert abs(mx - 0.010 * world) < 1e-12 and units == nchunks
    dist.barrier()
    q.put((rank, mx, units))
    dist.destroy_process_group()


def test_sharding_and_timing_reduce_world2():
    world, port = 2, 29533 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert [g[0] for g in got] == [0, 1]
    assert all(abs(g[1] - 0.020) < 1e-12 and g[2] == 67 for g in got)


def test_owner_rank_is_stable():
    from lmcache_amd.distributed import owner_rank, whole_job_rate
    k = "vllm@meta-llama/Llama-3.1-8B-Instruct@8@3@" + "ab" * 32
    assert owner_rank(k, 8) == owner_rank(k, 8) and 0 <= owner_rank(k, 8) < 8
    counts = [0] * 8
    for i in range(800):
        counts[owner_rank(k + str(i), 8)] += 1
    assert min(counts) > 50
    assert whole_job_rate([2.0, 2.0], 0.5) == 8.0
