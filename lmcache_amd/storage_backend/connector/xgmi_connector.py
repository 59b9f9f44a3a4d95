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
    key -> (owner, offset, size), open addressing on sha256(key), mutations under flock; a record is the
    counterpart of the reference's 158-byte ClientMetaMessage (lmcache/protocol.py:45-47) -- no pickled
    Python objects anywhere;
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
import pickle
import struct
import time
from typing import Dict, List, Optional, Tuple

import torch

from lmcache_amd.distributed import owner_rank
from lmcache_amd.logging import init_logger
from lmcache_amd.storage_backend.connector.base_connector import RemoteConnector

logger = init_logger(__name__)

_MAGIC = 0x494D4758  # "XGMI"
_HDR = struct.Struct("<IIIIQ")        # magic, world, nslots, record bytes, arena bytes per rank
_HDR_BYTES = 64
_REC = struct.Struct("<IIQQQI")       # state, owner, offset, size, capacity, key length   (+ key bytes)
_REC_BYTES = 256                      # one directory record: 36 B of fields + up to 220 B of key
_KEY_MAX = _REC_BYTES - _REC.size
_EMPTY, _FULL = 0, 1
SHM_DIR = "/dev/shm"


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
        self._keep = []
        # ---- directory (created by whoever comes first, under the lock of a sidecar file) -------------------
        self._lockf = open(self._base + ".lock", "a+b")
        with self._locked():
            path = self._base + ".dir"
            fresh = not os.path.exists(path) or os.path.getsize(path) == 0
            if fresh:
                with open(path, "wb") as f:
                    f.truncate(_HDR_BYTES + 8 * 64 + nslots * _REC_BYTES)
                    f.seek(0)
                    f.write(_HDR.pack(_MAGIC, self.world, nslots, _REC_BYTES, arena_bytes))
            self._dirf = open(path, "r+b")
            self._dir = mmap.mmap(self._dirf.fileno(), 0)
            magic, w, self.nslots, rec, self.arena_bytes = _HDR.unpack_from(self._dir, 0)
            if magic != _MAGIC or w != self.world or rec != _REC_BYTES:
                raise ValueError(f"xgmi://{name}: directory belongs to another world ({w} ranks) or version")
        # ---- this rank's arena -------------------------------------------------------------------------------
        own = _OWN_ARENAS.get((name, self.rank))
        if own is not None:
            own[1] += 1
            self._peers[self.rank] = own[0]
            self._closed = False
            return
        if self.device.type == "cuda":
            arena = torch.empty(self.arena_bytes, dtype=torch.uint8, device=self.device)
            from torch.multiprocessing.reductions import reduce_tensor
            with open(self._arena_path(self.rank) + ".tmp", "wb") as f:
                pickle.dump(reduce_tensor(arena), f)
            os.replace(self._arena_path(self.rank) + ".tmp", self._arena_path(self.rank))
        else:
            path = self._arena_path(self.rank)
            with open(path + ".tmp", "wb") as f:
                f.truncate(self.arena_bytes)
            os.replace(path + ".tmp", path)  # peers only ever see a fully sized file
            f = open(path, "r+b")
            mm = mmap.mmap(f.fileno(), 0)
            self._keep += [f, mm]
            arena = torch.frombuffer(mm, dtype=torch.uint8)
        self._peers[self.rank] = arena
        _OWN_ARENAS[(name, self.rank)] = [arena, 1]
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

    def _arena(self, r: int) -> torch.Tensor:
        """Rank r's arena as a tensor this process can address (mapped on first use)."""
        a = self._peers.get(r)
        if a is not None:
            return a
        path = self._arena_path(r)
        deadline = time.time() + 30.0
        while not os.path.exists(path):  # the peer has not exported its arena yet
            if time.time() > deadline:
                raise RuntimeError(f"xgmi://{self.name}: rank {r} never exported its arena")
            time.sleep(0.01)
        if self.device.type == "cuda":
            with open(path, "rb") as f:
                fn, args = pickle.load(f)
            a = fn(*args)  # HIP IPC: the peer's device memory, addressable from this process
        else:
            f = open(path, "r+b")
            mm = mmap.mmap(f.fileno(), 0)
            self._keep += [f, mm]
            a = torch.frombuffer(mm, dtype=torch.uint8)
        self._peers[r] = a
        return a

    def _slot_of(self, key: str) -> Tuple[int, bytes]:
        kb = key.encode("utf-8")
        if len(kb) > _KEY_MAX:
            raise ValueError(f"key longer than {_KEY_MAX} bytes")
        return int.from_bytes(hashlib.sha256(kb).digest()[8:16], "little") % self.nslots, kb

    def _rec_off(self, slot: int) -> int:
        return _HDR_BYTES + 8 * 64 + slot * _REC_BYTES

    def _find(self, key: str) -> Tuple[Optional[tuple], int]:
        """(record fields or None, slot index where the key is / would go).  Caller holds the lock."""
        slot, kb = self._slot_of(key)
        for _ in range(self.nslots):
            off = self._rec_off(slot)
            state, owner, offset, size, cap, klen = _REC.unpack_from(self._dir, off)
            if state == _EMPTY:
                return None, slot
            if klen == len(kb) and self._dir[off + _REC.size:off + _REC.size + klen] == kb:
                return (owner, offset, size, cap), slot
            slot = (slot + 1) % self.nslots
        raise RuntimeError(f"xgmi://{self.name}: directory full ({self.nslots} keys)")

    def _reserve(self, key: str, nbytes: int) -> Tuple[int, int, int]:
        """Directory entry for `key` with room for nbytes -> (owner, offset, slot); the record is published
        (state FULL, size) by _publish once the bytes are in place.  Caller holds the lock."""
        rec, slot = self._find(key)
        if rec is not None and rec[3] >= nbytes:  # overwrite in place: a miss until the new bytes are published
            off = self._rec_off(slot)
            _REC.pack_into(self._dir, off, _FULL, rec[0], rec[1], 0, rec[3], len(key.encode("utf-8")))
            return rec[0], rec[1], slot
        owner = owner_rank(key, self.world)
        boff = _HDR_BYTES + 8 * owner
        (used,) = struct.unpack_from("<Q", self._dir, boff)
        cap = _r256(max(nbytes, 1))
        if used + cap > self.arena_bytes:
            raise RuntimeError(f"xgmi://{self.name}: the arena of rank {owner} is full ({self.arena_bytes >> 20} MiB; "
                               f"LMC_XGMI_ARENA_MB sizes it)")
        struct.pack_into("<Q", self._dir, boff, used + cap)
        kb = key.encode("utf-8")
        off = self._rec_off(slot)
        # the key takes its slot now (size 0 reads as a miss) and becomes visible with _publish
        self._dir[off + _REC.size:off + _REC.size + len(kb)] = kb
        _REC.pack_into(self._dir, off, _FULL, owner, used, 0, cap, len(kb))
        return owner, used, slot

    def _publish(self, slot: int, nbytes: int) -> None:
        off = self._rec_off(slot)
        _, owner, offset, _, cap, klen = _REC.unpack_from(self._dir, off)
        _REC.pack_into(self._dir, off, _FULL, owner, offset, nbytes, cap, klen)

    def _lookup(self, key: str) -> Optional[tuple]:
        with self._locked():
            rec, _ = self._find(key)
        if rec is None or rec[2] == 0:
            return None
        return rec

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
        owner, offset, size, _ = rec
        return self._arena(owner)[offset:offset + size].cpu().numpy().tobytes()

    def list(self) -> List[str]:
        out = []
        with self._locked():
            for slot in range(self.nslots):
                off = self._rec_off(slot)
                state, _, _, size, _, klen = _REC.unpack_from(self._dir, off)
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
        for p in [self._base + ".dir", self._base + ".lock"] + [self._arena_path(r) for r in range(self.world)]:
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
        with self._locked():
            owner, offset, slot = self._reserve(key, n)
        dst = self._arena(owner)[offset:offset + n]
        dst.copy_(blob)
        if dst.is_cuda:  # the bytes must have landed before the entry becomes visible to the other ranks
            torch.cuda.current_stream(self.device).synchronize()
            if dst.device != self.device:
                torch.cuda.current_stream(dst.device).synchronize()
        with self._locked():
            self._publish(slot, n)

    def get_device(self, key: str) -> Optional[torch.Tensor]:
        """The blob as a uint8 tensor on THIS rank's device: a view of the own arena, or a copy out of the
        owner's (a peer read over xGMI)."""
        rec = self._lookup(key)
        if rec is None:
            return None
        owner, offset, size, _ = rec
        src = self._arena(owner)[offset:offset + size]
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
        owner, offset, size, _ = rec
        n = min(nbytes, size)
        return self._arena(owner)[offset:offset + n].cpu().numpy().tobytes(), size
