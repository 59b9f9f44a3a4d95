"""xgmi:// connector: the RemoteConnector contract, key ownership across ranks and
the shared directory, with the arenas in shared-memory files (CPU, runs anywhere) -- and, on a GPU box, the same
two-process exchange through real HIP IPC handles (both ranks on the one device)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _name(tag):
    return f"t{tag}{os.getpid()}"


def test_remote_connector_contract_single_rank():
    """exists / get / set / list / close as the reference's connectors behave (tests/test_connectors.py style):
    a missing key is None / False, set then get returns the bytes, overwriting replaces them."""
    from lmcache_amd.storage_backend.connector import CreateConnector
    from lmcache_amd.storage_backend.connector.xgmi_connector import XgmiConnector
    os.environ["LMC_XGMI_ARENA_MB"] = "8"
    name = _name("one")
    conn = CreateConnector(f"xgmi://{name}:1") if not torch.cuda.is_available() else XgmiConnector(name, 1, device="cpu")
    try:
        assert isinstance(conn, XgmiConnector) and conn.world == 1 and conn.rank == 0
        assert not conn.exists("a") and conn.get("a") is None and conn.list() == []
        blobs = {f"vllm@model@1@0@{i:064x}": os.urandom(1000 + 37 * i) for i in range(20)}
        for k, b in blobs.items():
            conn.set(k, b)
        for k, b in blobs.items():
            assert conn.exists(k) and conn.get(k) == b
        assert sorted(conn.list()) == sorted(blobs)
        k0 = next(iter(blobs))
        conn.set(k0, b"short")                      # overwrite in place
        assert conn.get(k0) == b"short"
        big = os.urandom(5000)
        conn.set(k0, big)                           # overwrite that needs a new region
        assert conn.get(k0) == big and len(conn.list()) == 20
        head, size = conn.peek(k0, 128)
        assert head == big[:128] and size == 5000
        t = conn.get_device(k0)
        assert t.dtype == torch.uint8 and bytes(t.cpu().numpy().tobytes()) == big
        with pytest.raises(RuntimeError):           # no eviction: a full arena says so
            conn.set("huge", bytes(9 << 20))
        with pytest.raises(ValueError):
            conn.set("k" * 300, b"x")
    finally:
        conn.close()
        conn.unlink()


def test_restart_retires_the_records_of_the_old_arena():
    """The directory file outlives the arenas.  A rank that comes back (here: close + reopen without unlink) creates
    a new arena: the keys its old arena held must read as misses, not as garbage of the new arena, the bump pointer
    starts again, and new stores of the same keys work."""
    from lmcache_amd.storage_backend.connector.xgmi_connector import XgmiConnector
    os.environ["LMC_XGMI_ARENA_MB"] = "2"
    name = _name("gen")
    conn = XgmiConnector(name, 1, device="cpu")
    try:
        keys = [f"vllm@m@1@0@{i:064x}" for i in range(8)]
        for k in keys:
            conn.set(k, k.encode() * 50)
        assert all(conn.exists(k) for k in keys)
        conn.close()
        conn = XgmiConnector(name, 1, device="cpu")          # same store name, stale directory, NEW arena
        assert not any(conn.exists(k) for k in keys) and conn.list() == []
        assert all(conn.get(k) is None for k in keys)
        for k in keys:                                        # skip-existing logic upstream now stores them again
            conn.set(k, k.encode() * 60)
        assert all(conn.get(k) == k.encode() * 60 for k in keys)
        for _ in range(40):                                   # 40 x 8 x ~4 KB would overflow 2 MiB if the old bump pointer had survived
            conn.close()
            conn = XgmiConnector(name, 1, device="cpu")
            for k in keys:
                conn.set(k, os.urandom(4000))
    finally:
        conn.close()
        conn.unlink()


def test_failed_and_abandoned_writes_do_not_wedge_a_key():
    """a key whose first writer fails between reserving its slot and publishing it, or dies there, must
    not stay "being written" for ever.  (1) a copy that raises retires the pending record: the same key can be set at
    once; (2) a pending record whose writer is no longer alive (or is older than the timeout) is taken over; (3) setting a
    published key again with the same number of bytes -- keys are content hashes -- takes no new extent; (4) retired slots
    are reused by later insertions."""
    import struct
    from lmcache_amd.storage_backend.connector import xgmi_connector as xc
    os.environ["LMC_XGMI_ARENA_MB"] = "2"
    conn = xc.XgmiConnector(_name("pend"), 1, device="cpu", nslots=64)
    try:
        used = lambda: struct.unpack_from("<Q", conn._dir, xc._USED_OFF)[0]
        # (1) the copy fails
        class Boom(RuntimeError):
            pass
        real = conn._arena
        def broken(owner, gen):
            raise Boom("no peer access")
        conn._arena = broken
        with pytest.raises(Boom):
            conn.set("k1", b"x" * 100)
        conn._arena = real
        assert not conn.exists("k1")
        conn.set("k1", b"y" * 100)
        assert conn.get("k1") == b"y" * 100
        # (2) a writer that died while its record was pending: stamp of a pid that does not exist
        with conn._locked():
            res = conn._reserve("k2", 50)
            assert res is not None
            off = conn._rec_off(res[2])
            state, owner, offset, size, cap, klen, gen = xc._REC.unpack_from(conn._dir, off)
            assert state == xc._PENDING
            xc._REC.pack_into(conn._dir, off, state, owner, offset, (0x3fffff << 32) | (size & 0xffffffff), cap, klen, gen)
            assert conn._reserve("k2", 50) is not None      # dead writer: taken over
        with conn._locked():                                 # a live writer (this process, just now) is respected
            assert conn._reserve("k2", 50) is None
        assert not conn.exists("k2") and "k2" not in conn.list()
        with conn._locked():                                 # ... until it has held the key for too long
            off = conn._rec_off(conn._find("k2")[1])
            f = list(xc._REC.unpack_from(conn._dir, off))
            f[3] = (os.getpid() << 32) | ((int(__import__("time").time()) - 10 * xc._PENDING_TIMEOUT_S) & 0xffffffff)
            xc._REC.pack_into(conn._dir, off, *f)
        conn.set("k2", b"z" * 50)
        assert conn.get("k2") == b"z" * 50
        # (3) the same bytes again: nothing is allocated
        before = used()
        conn.set("k2", b"z" * 50)
        assert used() == before and conn.get("k2") == b"z" * 50
        conn.set("k2", b"w" * 51)                            # another size: a fresh extent, the record flips to it
        assert used() > before and conn.get("k2") == b"w" * 51
        # (4) a retired slot is taken by the next key that probes over it
        slot_k1 = conn._find("k1")[1]
        with conn._locked():
            struct.pack_into("<I", conn._dir, conn._rec_off(slot_k1), xc._DEAD)
        assert not conn.exists("k1")
        conn.set("k1", b"again")
        assert conn._find("k1")[1] == slot_k1 and conn.get("k1") == b"again"
    finally:
        conn.close()
        conn.unlink()


def test_a_late_publish_never_lands_on_another_keys_record():
    """a slow writer whose pending slot was taken over, retired and re-used for another key must not
    publish its extent over that key's record."""
    import struct
    from lmcache_amd.storage_backend.connector import xgmi_connector as xc
    os.environ["LMC_XGMI_ARENA_MB"] = "2"
    conn = xc.XgmiConnector(_name("late"), 1, device="cpu", nslots=64)
    try:
        with conn._locked():
            owner, offset, slot, gen, cap = conn._reserve("slow", 64)   # the slow writer reserves ...
            conn._abandon(slot, offset)                                   # ... is given up on (as a take-over + failure would)
        # the retired slot goes to another key if that key probes over it: force it by writing the record by hand
        conn.set("other", b"o" * 32)
        oslot = conn._find("other")[1]
        with conn._locked():
            rec_other = bytes(conn._dir[conn._rec_off(oslot):conn._rec_off(oslot) + xc._REC.size + 5])
            # the slow writer wakes up and publishes with ITS slot number, which now holds `other`
            assert conn._publish(oslot, offset, 64, cap, gen, "slow") is False
            assert bytes(conn._dir[conn._rec_off(oslot):conn._rec_off(oslot) + xc._REC.size + 5]) == rec_other
            # its own retired slot: not a pending reservation of this extent any more -> dropped as well
            assert conn._publish(slot, offset, 64, cap, gen, "slow") is False
        assert conn.get("other") == b"o" * 32 and not conn.exists("slow")
        # the normal path still publishes: pending reservation of exactly this extent, and an overwrite of a published key
        conn.set("slow", b"s" * 64)
        assert conn.get("slow") == b"s" * 64
        conn.set("slow", b"t" * 65)
        assert conn.get("slow") == b"t" * 65
    finally:
        conn.close()
        conn.unlink()


def test_no_pickle_in_the_connectors():
    """Peers exchange fixed-layout binary records (directory, arena exports, SPMD exchange): nothing in the connector
    package unpickles bytes another process wrote."""
    d = os.path.join(ROOT, "lmcache_amd", "storage_backend", "connector")
    for f in os.listdir(d):
        if f.endswith(".py"):
            src = open(os.path.join(d, f)).read()
            assert "import pickle" not in src and "pickle.load" not in src and "pickle.dump" not in src, f


def _rank(rank, world, name, device, q):
    sys.path.insert(0, ROOT)
    os.environ["LMC_XGMI_ARENA_MB"] = "16"
    from lmcache_amd.distributed import owner_rank
    from lmcache_amd.storage_backend.connector.xgmi_connector import XgmiConnector
    try:
        if device == "cuda:rank":                  # one GPU per rank when the box has them, else both ranks on GPU 0
            device = f"cuda:{rank % torch.cuda.device_count()}"
        if device != "cpu":
            torch.cuda.set_device(torch.device(device))
        conn = XgmiConnector(name, world, rank=rank, device=device)
        mine = {f"vllm@m@{world}@{rank}@{i:04x}": bytes([rank * 16 + (i % 16)]) * (3000 + 100 * i) for i in range(12)}
        for i, (k, b) in enumerate(mine.items()):
            if device == "cpu" or i % 2:
                conn.set(k, b)
        if device != "cpu":                       # the device form for every key (overwrites the ones set above)
            for k, b in mine.items():
                conn.set_device(k, torch.frombuffer(bytearray(b), dtype=torch.uint8).to(device))
        # wait until the other rank has published everything
        import time
        other = {f"vllm@m@{world}@{1 - rank}@{i:04x}": bytes([(1 - rank) * 16 + (i % 16)]) * (3000 + 100 * i) for i in range(12)}
        deadline = time.time() + 60
        while not all(conn.exists(k) for k in other):
            assert time.time() < deadline, "peer never published"
            time.sleep(0.01)
        owners = set()
        for k, b in list(mine.items()) + list(other.items()):
            assert conn.get(k) == b, k
            d = conn.get_device(k)
            assert bytes(d.cpu().numpy().tobytes()) == b
            owners.add(owner_rank(k, world))
        assert owners == {0, 1}, "keys should spread over both ranks' arenas"
        assert len(conn.list()) == 24
        q.put((rank, "ok"))
        # keep the arena alive until the peer is done reading it
        deadline = time.time() + 60
        while not os.path.exists(f"/dev/shm/{name}.done{1 - rank}"):
            open(f"/dev/shm/{name}.done{rank}", "w").close()
            assert time.time() < deadline
            time.sleep(0.01)
        open(f"/dev/shm/{name}.done{rank}", "w").close()
        time.sleep(0.2)
        conn.close()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail: " + traceback.format_exc()))


def _two_ranks(device):
    name = _name("two" + device[:3])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 2, name, device, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    for f in os.listdir("/dev/shm"):
        if f.startswith(f"lmc_xgmi_{name}") or f.startswith(f"{name}.done"):
            os.unlink(os.path.join("/dev/shm", f))
    assert got == [(0, "ok"), (1, "ok")], got


def test_two_ranks_share_one_store_cpu():
    """Two processes = two ranks: each publishes its chunks (half of them land in the OTHER rank's arena, by key
    ownership) and reads the peer's; the directory is the only thing they share besides the arenas."""
    _two_ranks("cpu")


@pytest.mark.gpu
def test_two_ranks_share_one_store_hip_ipc():
    """The same exchange with the arenas in HBM, mapped across the two processes through HIP IPC handles (both ranks
    on the one GPU of the test box): peer writes on set, peer reads on get."""
    _two_ranks("cuda:0")


@pytest.mark.gpu
def test_two_ranks_two_devices_hip_ipc():
    """One GPU per rank: the peer arena is another device's HBM, reached over xGMI through the IPC mapping."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the one-GPU form of this test is test_two_ranks_share_one_store_hip_ipc)")
    _two_ranks("cuda:rank")
