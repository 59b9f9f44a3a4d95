"""XgmiConnector -- `xgmi://<name>:<world>`: encoded KV chunks shared between the vLLM instances of ONE node,
resident in the HBM of the GPUs and moved over xGMI (BASELINE configs[2]).

What it replaces: LMCServerConnector + the lmcache_server process (lmcache/storage_backend/connector/
lm_connector.py:15-84, lmcache/server/__main__.py:29-104): a TCP round trip per chunk through a Python
recv loop into the server's host dict.  Same five-method RemoteConnector interface
(connector/base_connector.py:11-70: exists / get / set / list / close), so LMCRemoteBackend and the engine
use it unchanged through CreateConnector.

MI355X-native form -- no server process, no peer Python in the data path:

  * every rank (one process per GPU) owns one HBM arena and the shard of keys with owner_rank(key) == rank
    (lmcache_amd/distributed.py).  The arena is exported once through HIP IPC (torch's CUDA-IPC reduction,
    dmabuf handles: HSA_ENABLE_IPC_MODE_LEGACY=0) and mapped by the peers on first use;
  * `set` of a key owned by rank o is a copy INTO o's arena (a peer write over xGMI), `get` a copy OUT of it
    (a peer read): hipMemcpy between mapped device pointers, the owner's process does not take part;
  * the control plane is a fixed-size directory in POSIX shared memory (/dev/shm/lmc_xgmi_<name>.dir):
    key -> (owner, offset, size, arena generation), open addressing on sha256(key), mutations under flock; a
    record is the counterpart of the reference's 158-byte ClientMetaMessage (lmcache/protocol.py:45-47).  An
    arena is exported as a fixed-layout binary record of its IPC handles (no pickled Python objects anywhere:
    a peer parses bytes, it never executes what it reads), in a file created 0600 / O_EXCL / O_NOFOLLOW whose
    owner is checked before it is read;
  * the directory outlives processes, the arenas do not: every arena carries a GENERATION.  A rank that creates
    a new arena bumps its generation in the directory header, resets its bump pointer and retires every record
    it owned; records carry the generation of the arena their bytes live in, and a peer that meets a newer
    generation than the mapping it holds maps the new export -- a restarted rank (or a whole restarted job on a
    stale /dev/shm) never serves bytes of an arena that is gone;
  * live bytes are never overwritten: `set` of a key that is present copies into a FRESH extent of the owner's
    arena and flips the record to it when the bytes have landed (a reader that looked the key up before keeps
    reading the old extent, which nobody touches); a second writer of a key whose first writer has not
    published yet is a no-op (keys are content hashes, cache_engine.py:58-96: the bytes would be the same);
  * `set_device` / `get_device` / `peek` are the zero-host-hop forms LMCPipelinedRemoteBackend uses when the
    connector has them: blobs go from the encode arena to the owner's HBM and from there to the decode arena
    without touching host memory (the bytes methods of the interface bounce through the host by definition).

No eviction, like the reference's server (server_storage_backend/local_backend.py:69): a full arena raises.
With device="cpu" the arenas are shared-memory files -- that is how the ownership / directory logic is tested
by two processes without GPUs (tests/test_xgmi_connector.py); on a GPU box the same test runs two processes on
one device through real HIP IPC handles.
"""
import fcntl
import hashlib
import mmap
import os
import stat
import struct
import time
from typing import Dict, List, Optional, Tuple

import torch

from lmcache_amd.distributed import owner_rank
from lmcache_amd.logging import init_logger
from lmcache_amd.storage_backend.connector.base_connector import RemoteConnector

logger = init_logger(__name__)

_MAGIC = 0x32494D58  # "XMI2": directory layout 2 (per-rank generations; records carry one)
_HDR = struct.Struct("<IIIIQ")        # magic, world, nslots, record bytes, arena bytes per rank
_HDR_BYTES = 64
_MAX_RANKS = 64
_USED_OFF = _HDR_BYTES                # u64 [64]: bump pointer of every rank's arena
_GEN_OFF = _HDR_BYTES + 8 * _MAX_RANKS  # u64 [64]: generation of every rank's arena (0 = never created)
_RECS_OFF = _HDR_BYTES + 16 * _MAX_RANKS
_REC = struct.Struct("<IIQQQII")      # state, owner, offset, size, capacity, key length, arena generation  (+ key bytes)
_REC_BYTES = 256                      # one directory record: 40 B of fields + up to 216 B of key
_KEY_MAX = _REC_BYTES - _REC.size
_EMPTY, _FULL, _DEAD, _PENDING = 0, 1, 2, 3  # _DEAD: a retired record (its arena is gone, or its writer failed): probing
                                      # walks past it and an insertion may take it.  _PENDING: a new key whose first
                                      # bytes are on their way -- a miss for readers; its `size` field holds the
                                      # writer's stamp (pid << 32 | seconds) so that a writer that died can be replaced
_PENDING_TIMEOUT_S = 20
SHM_DIR = "/dev/shm"

# export record of a CUDA arena: magic, version, generation, device, storage bytes, storage offset, ref-counter
# offset, event-sync flag, lengths of the three handle blobs that follow (what torch's _share_cuda_ returns)
_XIPC_MAGIC = 0x43504958  # "XIPC"
_XIPC = struct.Struct("<IIQiIQQQBxHHH")
_XIPC_MAX_BLOB = 256


def _write_private(path: str, data: bytes) -> None:
    """Create `path` atomically with mode 0600; refuses to follow links or reuse somebody else's temp file."""
    tmp = f"{path}.tmp{os.getpid()}"
    try:
        os.unlink(tmp)
    except OSError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
    try:
        os.write(fd, data)
    finally:
        os.close(fd)
    os.replace(tmp, path)


def _read_private(path: str, max_bytes: int) -> Optional[bytes]:
    """Contents of a regular file that belongs to this user, or None."""
    try:
        fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
    except OSError:
        return None
    try:
        st = os.fstat(fd)
        if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or st.st_size > max_bytes:
            raise PermissionError(f"{path}: not a private regular file of this user")
        return os.read(fd, max_bytes)
    finally:
        os.close(fd)


def _r256(n: int) -> int:
    return (n + 255) & ~255


# arenas this process owns, by (store name, rank): two connectors of one process on the same store and rank (two
# engines in one process) are two handles of ONE arena -- a second allocation would orphan what the first one holds
_OWN_ARENAS: Dict[Tuple[str, int], list] = {}


class XgmiConnector(RemoteConnector):
    def __init__(self, name: str, world: int, rank: Optional[int] = None, device: Optional[str] = None,
                 arena_bytes: Optional[int] = None, nslots: int = 1 << 15):
        self.name = name
        self.world = max(1, int(world) or int(os.environ.get("WORLD_SIZE", "1")))
        self.rank = int(os.environ.get("RANK", "0")) % self.world if rank is None else int(rank)
        if not (0 <= self.rank < self.world):
            raise ValueError(f"rank {self.rank} outside world {self.world}")
        if device is None:
            device = f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        if arena_bytes is None:
            arena_bytes = int(os.environ.get("LMC_XGMI_ARENA_MB", "4096" if self.device.type == "cuda" else "64")) << 20
        self._base = os.path.join(SHM_DIR, f"lmc_xgmi_{name}")
        self._peers: Dict[int, torch.Tensor] = {}
        self._peer_gen: Dict[int, int] = {}   # generation of the mapping held in _peers
        self._keep = []
        # ---- directory (created by whoever comes first, under the lock of a sidecar file) -------------------
        self._lockf = open(self._base + ".lock", "a+b")
        with self._locked():
            path = self._base + ".dir"
            fresh = not os.path.exists(path) or os.path.getsize(path) == 0
            if not fresh:  # a directory of an older layout (a stale /dev/shm) is replaced, never interpreted
                with open(path, "rb") as f:
                    head = f.read(_HDR.size)
                fresh = len(head) < _HDR.size or _HDR.unpack(head)[0] != _MAGIC
            if fresh:
                fd = os.open(path + ".new", os.O_WRONLY | os.O_CREAT | os.O_TRUNC | os.O_NOFOLLOW, 0o600)
                with os.fdopen(fd, "wb") as f:
                    f.truncate(_RECS_OFF + nslots * _REC_BYTES)
                    f.seek(0)
                    f.write(_HDR.pack(_MAGIC, self.world, nslots, _REC_BYTES, arena_bytes))
                os.replace(path + ".new", path)
            self._dirf = open(path, "r+b")
            self._dir = mmap.mmap(self._dirf.fileno(), 0)
            magic, w, self.nslots, rec, self.arena_bytes = _HDR.unpack_from(self._dir, 0)
            if magic != _MAGIC or w != self.world or rec != _REC_BYTES or self.world > _MAX_RANKS:
                raise ValueError(f"xgmi://{name}: directory belongs to another world ({w} ranks) or version")
        # ---- this rank's arena -------------------------------------------------------------------------------
        own = _OWN_ARENAS.get((name, self.rank))
        if own is not None:
            own[1] += 1
            self._peers[self.rank] = own[0]
            self._peer_gen[self.rank] = own[2]
            self._closed = False
            return
        # a NEW arena: the directory may still describe an older one of this rank (a restart): next generation, bump
        # pointer back to zero, every record this rank owned retired -- before the export becomes visible
        with self._locked():
            gen = struct.unpack_from("<Q", self._dir, _GEN_OFF + 8 * self.rank)[0] + 1
            struct.pack_into("<Q", self._dir, _GEN_OFF + 8 * self.rank, gen)
            struct.pack_into("<Q", self._dir, _USED_OFF + 8 * self.rank, 0)
            for slot in range(self.nslots):
                off = self._rec_off(slot)
                state, owner = struct.unpack_from("<II", self._dir, off)
                if state in (_FULL, _PENDING) and owner == self.rank:
                    struct.pack_into("<I", self._dir, off, _DEAD)
        if self.device.type == "cuda":
            arena = torch.empty(self.arena_bytes, dtype=torch.uint8, device=self.device)
            (dev_index, handle, nbytes, offset, ref_handle, ref_offset, ev_handle, ev_sync) = \
                arena.untyped_storage()._share_cuda_()
            blobs = [bytes(handle), bytes(ref_handle), bytes(ev_handle or b"")]
            if any(len(b) > _XIPC_MAX_BLOB for b in blobs):
                raise RuntimeError("xgmi://: an IPC handle is longer than the export record allows")
            rec = _XIPC.pack(_XIPC_MAGIC, 1, gen, int(dev_index), 0, int(nbytes), int(offset), int(ref_offset),
                             1 if ev_sync else 0, *(len(b) for b in blobs)) + b"".join(blobs)
            _write_private(self._arena_path(self.rank), rec)
        else:
            path = self._arena_path(self.rank) + f".g{gen}"  # one file per generation: an old mapping stays what it was
            fd = os.open(path + ".tmp", os.O_WRONLY | os.O_CREAT | os.O_TRUNC | os.O_NOFOLLOW, 0o600)
            with os.fdopen(fd, "wb") as f:
                f.truncate(self.arena_bytes)
            os.replace(path + ".tmp", path)  # peers only ever see a fully sized file
            f = open(path, "r+b")
            mm = mmap.mmap(f.fileno(), 0)
            self._keep += [f, mm]
            arena = torch.frombuffer(mm, dtype=torch.uint8)
        self._peers[self.rank] = arena
        self._peer_gen[self.rank] = gen
        _OWN_ARENAS[(name, self.rank)] = [arena, 1, gen]
        self._closed = False

    # ------------------------------------------------------------------ plumbing
    def _arena_path(self, r: int) -> str:
        return f"{self._base}.arena{r}"

    class _Lock:
        def __init__(self, f):
            self.f = f

        def __enter__(self):
            fcntl.flock(self.f, fcntl.LOCK_EX)

        def __exit__(self, *a):
            fcntl.flock(self.f, fcntl.LOCK_UN)

    def _locked(self):
        return XgmiConnector._Lock(self._lockf)

    def _arena(self, r: int, gen: int) -> torch.Tensor:
        """Generation `gen` of rank r's arena as a tensor this process can address (mapped on first use, mapped
        again when the rank has created a newer arena since)."""
        a = self._peers.get(r)
        if a is not None and self._peer_gen.get(r) == gen:
            return a
        deadline = time.time() + 30.0
        while True:
            a = self._map_arena(r, gen)
            if a is not None:
                break
            if time.time() > deadline:
                raise RuntimeError(f"xgmi://{self.name}: rank {r} never exported generation {gen} of its arena")
            time.sleep(0.01)
        self._peers[r] = a
        self._peer_gen[r] = gen
        return a

    def _map_arena(self, r: int, gen: int) -> Optional[torch.Tensor]:
        if self.device.type != "cuda":
            path = self._arena_path(r) + f".g{gen}"
            if not os.path.exists(path):
                return None
            f = open(path, "r+b")
            mm = mmap.mmap(f.fileno(), 0)
            self._keep += [f, mm]
            return torch.frombuffer(mm, dtype=torch.uint8)
        raw = _read_private(self._arena_path(r), _XIPC.size + 3 * _XIPC_MAX_BLOB)
        if raw is None or len(raw) < _XIPC.size:
            return None
        magic, ver, fgen, dev_index, _, nbytes, offset, ref_offset, ev_sync, l0, l1, l2 = _XIPC.unpack_from(raw, 0)
        if magic != _XIPC_MAGIC or ver != 1 or max(l0, l1, l2) > _XIPC_MAX_BLOB or len(raw) != _XIPC.size + l0 + l1 + l2:
            raise RuntimeError(f"xgmi://{self.name}: malformed arena export of rank {r}")
        if fgen != gen:
            return None  # the export of another generation: the rank is about to publish the one asked for
        if nbytes < self.arena_bytes or not (0 <= dev_index < torch.cuda.device_count()):
            raise RuntimeError(f"xgmi://{self.name}: arena export of rank {r} does not describe this store")
        p = _XIPC.size
        handle, ref_handle, ev_handle = raw[p:p + l0], raw[p + l0:p + l0 + l1], raw[p + l0 + l1:p + l0 + l1 + l2]
        # HIP IPC: the peer's device memory, addressable from this process (torch's CUDA-IPC storage, opened from
        # plain bytes -- the counterpart of rebuild_cuda_tensor without unpickling anything)
        torch.cuda._lazy_init()
        storage = torch.UntypedStorage._new_shared_cuda(dev_index, handle, nbytes, offset, ref_handle, ref_offset,
                                                        ev_handle, bool(ev_sync))
        return torch.empty(0, dtype=torch.uint8, device=f"cuda:{dev_index}").set_(storage, 0, (self.arena_bytes,), (1,))

    def _slot_of(self, key: str) -> Tuple[int, bytes]:
        kb = key.encode("utf-8")
        if len(kb) > _KEY_MAX:
            raise ValueError(f"key longer than {_KEY_MAX} bytes")
        return int.from_bytes(hashlib.sha256(kb).digest()[8:16], "little") % self.nslots, kb

    def _rec_off(self, slot: int) -> int:
        return _RECS_OFF + slot * _REC_BYTES

    def _find(self, key: str) -> Tuple[Optional[tuple], int]:
        """(record fields or None, slot index where the key is / would go).  The fields are (owner, offset, size,
        capacity, generation, state); an insertion goes to the first retired slot of the probe sequence if there is one,
        else to the empty slot that ends it.  Caller holds the lock."""
        slot, kb = self._slot_of(key)
        first_dead = None
        for _ in range(self.nslots):
            off = self._rec_off(slot)
            state, owner, offset, size, cap, klen, gen = _REC.unpack_from(self._dir, off)
            if state == _EMPTY:
                return None, slot if first_dead is None else first_dead
            if state == _DEAD and first_dead is None:
                first_dead = slot
            if state in (_FULL, _PENDING) and klen == len(kb) and self._dir[off + _REC.size:off + _REC.size + klen] == kb:
                return (owner, offset, size, cap, gen, state), slot
            slot = (slot + 1) % self.nslots
        if first_dead is not None:
            return None, first_dead
        raise RuntimeError(f"xgmi://{self.name}: directory full ({self.nslots} keys)")

    @staticmethod
    def _stamp() -> int:
        return (os.getpid() << 32) | (int(time.time()) & 0xffffffff)

    @staticmethod
    def _stamp_is_stale(stamp: int) -> bool:
        """The writer behind a pending record is gone (no such process on this node) or has held it for too long."""
        pid, t = stamp >> 32, stamp & 0xffffffff
        if ((int(time.time()) & 0xffffffff) - t) & 0xffffffff > _PENDING_TIMEOUT_S:
            return True
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return True
        except OSError:
            pass
        return False

    def _reserve(self, key: str, nbytes: int) -> Optional[Tuple[int, int, int, int, int]]:
        """A fresh extent of nbytes for `key` -> (owner, offset, slot, generation, capacity); the record points to it
        (and the key becomes visible, or its new bytes do) with _publish once the bytes are in place.  None when there
        is nothing to write: another live writer holds the key unpublished, or the key is published with exactly this
        many bytes (keys are content hashes, cache_engine.py:58-96: the bytes would be the same).  A pending record
        whose writer died, or has not published for _PENDING_TIMEOUT_S, is taken over.  Caller holds the lock."""
        rec, slot = self._find(key)
        if rec is not None:
            if rec[5] == _PENDING and not self._stamp_is_stale(rec[2]):
                return None  # being written right now
            if rec[5] == _FULL and rec[2] == nbytes:
                cur = struct.unpack_from("<Q", self._dir, _GEN_OFF + 8 * rec[0])[0] & 0xffffffff
                if rec[4] == cur:
                    return None  # already there
        owner = owner_rank(key, self.world) if rec is None else rec[0]
        gen = struct.unpack_from("<Q", self._dir, _GEN_OFF + 8 * owner)[0]
        if gen == 0:
            raise LookupError(owner)  # the owner has not created its arena yet: set_device waits for it
        boff = _USED_OFF + 8 * owner
        (used,) = struct.unpack_from("<Q", self._dir, boff)
        cap = _r256(max(nbytes, 1))
        if used + cap > self.arena_bytes:
            raise RuntimeError(f"xgmi://{self.name}: the arena of rank {owner} is full ({self.arena_bytes >> 20} MiB; "
                               f"LMC_XGMI_ARENA_MB sizes it)")
        struct.pack_into("<Q", self._dir, boff, used + cap)
        if rec is None or rec[5] == _PENDING:  # a new key takes its slot now, pending (a miss for readers)
            kb = key.encode("utf-8")
            off = self._rec_off(slot)
            self._dir[off + _REC.size:off + _REC.size + len(kb)] = kb
            _REC.pack_into(self._dir, off, _PENDING, owner, used, self._stamp(), cap, len(kb), gen & 0xffffffff)
        return owner, used, slot, gen, cap

    def _abandon(self, slot: int, offset: int) -> None:
        """A write that failed between _reserve and _publish: a pending record (still ours: same extent) is retired so
        that the key can be written again at once; a published record still points to its old bytes and stays."""
        off = self._rec_off(slot)
        state, _, roff, _, _, _, _ = _REC.unpack_from(self._dir, off)
        if state == _PENDING and roff == offset:
            struct.pack_into("<I", self._dir, off, _DEAD)

    def _publish(self, slot: int, offset: int, nbytes: int, cap: int, gen: int, key: str) -> bool:
        """Make the extent visible -- if the slot is still this write's: a pending reservation of exactly this extent,
        or the key's own published record (an overwrite).  A slow writer whose reservation was taken over after
        _PENDING_TIMEOUT_S, retired and re-used for ANOTHER key must not stamp its extent over that key's record
        (ADVICE r04): its write is dropped (the bytes stay unreferenced in the arena).  Caller holds the lock."""
        off = self._rec_off(slot)
        state, owner, roff, _, _, klen, _ = _REC.unpack_from(self._dir, off)
        kb = key.encode("utf-8")
        same_key = klen == len(kb) and self._dir[off + _REC.size:off + _REC.size + klen] == kb
        if not same_key or not ((state == _PENDING and roff == offset) or state == _FULL):
            return False
        _REC.pack_into(self._dir, off, _FULL, owner, offset, nbytes, cap, klen, gen & 0xffffffff)
        return True

    def _lookup(self, key: str) -> Optional[tuple]:
        with self._locked():
            rec, _ = self._find(key)
        if rec is None or rec[5] != _FULL or rec[2] == 0:
            return None
        # a record of an arena generation that is gone (its owner restarted and has not retired it yet) is a miss
        with self._locked():
            cur = struct.unpack_from("<Q", self._dir, _GEN_OFF + 8 * rec[0])[0] & 0xffffffff
        return rec[:5] if rec[4] == cur else None

    # ------------------------------------------------------------------ RemoteConnector
    def exists(self, key: str) -> bool:
        return self._lookup(key) is not None

    def set(self, key: str, obj: bytes) -> None:
        src = torch.frombuffer(bytearray(obj), dtype=torch.uint8)
        self.set_device(key, src)

    def get(self, key: str) -> Optional[bytes]:
        rec = self._lookup(key)
        if rec is None:
            return None
        owner, offset, size, _, gen = rec
        return self._arena(owner, gen)[offset:offset + size].cpu().numpy().tobytes()

    def list(self) -> List[str]:
        out = []
        with self._locked():
            for slot in range(self.nslots):
                off = self._rec_off(slot)
                state, _, _, size, _, klen, _ = _REC.unpack_from(self._dir, off)
                if state == _FULL and size:
                    out.append(bytes(self._dir[off + _REC.size:off + _REC.size + klen]).decode("utf-8"))
        return out

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        own = _OWN_ARENAS.get((self.name, self.rank))
        if own is not None:
            own[1] -= 1
            if own[1] <= 0:
                del _OWN_ARENAS[(self.name, self.rank)]
        self._peers.clear()
        self._peer_gen.clear()
        try:
            self._dir.close()
            self._dirf.close()
            self._lockf.close()
        except Exception:
            pass
        for x in self._keep:
            try:
                x.close()
            except Exception:
                pass
        self._keep = []

    def unlink(self) -> None:
        """Remove the shared files of this store (the last user of a name calls it; mapped segments stay valid)."""
        import glob
        paths = [self._base + ".dir", self._base + ".lock"]
        for r in range(self.world):
            paths += [self._arena_path(r)] + glob.glob(self._arena_path(r) + ".g*")
        for p in paths:
            try:
                os.unlink(p)
            except OSError:
                pass

    # ------------------------------------------------------------------ zero-host-hop forms
    def set_device(self, key: str, blob: torch.Tensor) -> None:
        """Store a uint8 tensor (any device): one copy into the owner's arena -- a peer write over xGMI when the
        owner is another GPU -- and the directory entry is published once the bytes have landed."""
        blob = blob.reshape(-1)
        n = blob.numel()
        deadline = time.time() + 30.0
        while True:
            try:
                with self._locked():
                    res = self._reserve(key, n)
                break
            except LookupError as e:  # a peer that is still starting up
                if time.time() > deadline:
                    raise RuntimeError(f"xgmi://{self.name}: rank {e.args[0]} never created its arena")
                time.sleep(0.01)
        if res is None:
            return  # the key is there, or another rank is writing it right now: content hashes, the bytes would be the same
        owner, offset, slot, gen, cap = res
        try:
            dst = self._arena(owner, gen)[offset:offset + n]
            dst.copy_(blob)
            if dst.is_cuda:  # the bytes must have landed before the entry becomes visible to the other ranks
                torch.cuda.current_stream(self.device).synchronize()
                if dst.device != self.device:
                    torch.cuda.current_stream(dst.device).synchronize()
        except BaseException:
            with self._locked():
                self._abandon(slot, offset)  # the key is not left "being written" for ever
            raise
        with self._locked():
            self._publish(slot, offset, n, cap, gen, key)

    def get_device(self, key: str) -> Optional[torch.Tensor]:
        """The blob as a uint8 tensor on THIS rank's device: a view of the own arena, or a copy out of the
        owner's (a peer read over xGMI)."""
        rec = self._lookup(key)
        if rec is None:
            return None
        owner, offset, size, _, gen = rec
        src = self._arena(owner, gen)[offset:offset + size]
        if owner == self.rank:
            return src
        out = torch.empty(size, dtype=torch.uint8, device=self.device)
        out.copy_(src)
        return out

    def peek(self, key: str, nbytes: int) -> Optional[Tuple[bytes, int]]:
        """(first nbytes of the blob, its total size) -- enough for lmc_blob_info without moving the blob."""
        rec = self._lookup(key)
        if rec is None:
            return None
        owner, offset, size, _, gen = rec
        n = min(nbytes, size)
        return self._arena(owner, gen)[offset:offset + n].cpu().numpy().tobytes(), size
