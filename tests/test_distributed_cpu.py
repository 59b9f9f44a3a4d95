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


def _exchange_worker(rank, world, port, q, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":  # RCCL: one GPU per rank (tests/test_gpu_exchange.py)
        torch.cuda.set_device(rank % torch.cuda.device_count())
    dist.init_process_group(backend, rank=rank, world_size=world)
    from lmcache_amd.storage_backend.connector.xgmi_exchange import XgmiShardStore
    store = XgmiShardStore()
    assert store.device.type == ("cuda" if backend == "nccl" else "cpu") and store.world == world

    def blob_of(key):  # deterministic, ragged sizes
        g = torch.Generator().manual_seed(sum(key.encode()))
        n = 1000 + (sum(key.encode()) * 37) % 5000
        return torch.randint(0, 256, (n,), dtype=torch.uint8, generator=g).to(store.device)

    keys = [f"vllm@m@{world}@{r}@{i:04x}" for r in range(world) for i in range(6)]
    mine = [k for k in keys if k.split("@")[3] == str(rank)]
    owned_now = store.exchange_put([(k, blob_of(k)) for k in mine])
    assert owned_now == sum(1 for k in keys if store.owner(k) == rank)
    assert sorted(store.list_local()) == sorted(k for k in keys if store.owner(k) == rank)
    # every rank asks for a different mix: all keys rotated by rank, plus two misses
    want = keys[rank:] + keys[:rank]
    want.insert(3, "vllm@m@missing@a")
    want.append("vllm@m@missing@b")
    got = store.exchange_get(want)
    assert len(got) == len(want)
    for k, t in zip(want, got):
        if "missing" in k:
            assert t is None
        else:
            assert t is not None and torch.equal(t, blob_of(k)), k
    # an empty request from one rank must not wedge the others
    got2 = store.exchange_get([] if rank == 0 else keys[:2])
    assert (got2 == []) if rank == 0 else all(torch.equal(t, blob_of(k)) for k, t in zip(keys[:2], got2))
    # byte-like inputs are accepted too
    store.exchange_put([(f"bytes@{rank}", bytes([rank, 1, 2, 3]))])
    back = store.exchange_get([f"bytes@{r}" for r in range(world)])
    assert [bytes(t.tolist()) for t in back] == [bytes([r, 1, 2, 3]) for r in range(world)]
    dist.barrier()
    q.put(rank)
    dist.destroy_process_group()


def _run_world(target, world, port, *extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    return sorted(q.get(timeout=5) for _ in range(world))


def test_shard_exchange_world2():
    assert _run_world(_exchange_worker, 2, 31533 + os.getpid() % 2000) == [0, 1]


def test_shard_exchange_world3():
    assert _run_world(_exchange_worker, 3, 33533 + os.getpid() % 2000) == [0, 1, 2]


def test_bench_launch_plumbing_world2_stub():
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one rank per GPU), with
    LMC_BENCH_STUB=1: gloo instead of RCCL and a sleep instead of the kernel, so that process-group init, the
    barrier / synchronize bracket, the max-over-ranks reduction, the whole-job value and the teardown are all
    exercised without a GPU -- the first real 8-GPU run must not die in launch plumbing."""
    import json
    import subprocess
    port = 29700 + os.getpid() % 2000
    env = dict(os.environ, LMC_BENCH_STUB="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "4", "--warmup", "1"], env=env, capture_output=True, text=True,
                         timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 prints ONE JSON line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["higher_is_better"] is True and r["unit"] == "GB/s" and r["vs_baseline"] is None
    # whole-job value: both ranks' bytes over the max-over-ranks time of exactly 4 steps (2 ms sleeps)
    raw = 32 * 2 * 16384 * 8 * 128 * 2
    assert abs(r["value"] - 2 * raw * 4 / (r["ms_per_step"] * 4 / 1e3) / 1e9) / r["value"] < 1e-3
    assert 2.0 <= r["ms_per_step"] < 50.0
    assert r["rccl_ranks_seen"] == 2


def test_bench_launch_plumbing_world8_strong_stub():
    """The same rehearsal at the size of the node the driver's scaling run uses -- 8 ranks -- with SURVEY.md 8e's own
    split (--scaling strong: ONE context, rank r takes chunks r, r + 8, ...): every rank of the process group is seen
    by the all_reduce, and value is the context's bytes over the max-over-ranks time."""
    import json
    import subprocess
    port = 27700 + os.getpid() % 2000
    env = dict(os.environ, LMC_BENCH_STUB="1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "8", "--steps", "3", "--warmup", "1", "--scaling", "strong"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["scaling"] == "strong" and r["rccl_ranks_seen"] == 8
    assert "chunks i mod 8 == rank (8 chunks on rank 0)" in r["config"]["sharding"]
    raw = 32 * 2 * 16384 * 8 * 128 * 2
    assert abs(r["value"] - raw * 3 / (r["ms_per_step"] * 3 / 1e3) / 1e9) / r["value"] < 1e-3


def test_shard_exchange_world8():
    assert _run_world(_exchange_worker, 8, 35533 + os.getpid() % 2000) == list(range(8))


def test_numa_topology_helpers(tmp_path):
    """GPU -> NUMA node -> CPU list, read from sysfs (bind_to_gpu_numa pins a rank's pinned arenas next to its GPU)."""
    from lmcache_amd.distributed import gpu_numa_node, numa_cpus
    dev = tmp_path / "pci" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "node" / "node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("64-67,192-193\n")
    assert gpu_numa_node("0000:C1:00.0", str(tmp_path / "pci")) == 1
    assert gpu_numa_node("0000:ff:00.0", str(tmp_path / "pci")) is None
    assert numa_cpus(1, str(tmp_path / "node")) == [64, 65, 66, 67, 192, 193]
    assert numa_cpus(7, str(tmp_path / "node")) == []
    (dev / "numa_node").write_text("-1\n")
    assert gpu_numa_node("0000:c1:00.0", str(tmp_path / "pci")) is None
