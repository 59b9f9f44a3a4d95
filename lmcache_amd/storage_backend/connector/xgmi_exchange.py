"""Sharded exchange of encoded KV chunks between GPUs of one node.

What it stands in for: two vLLM instances sharing KV through one `lmcache_server` over TCP
(lmcache/storage_backend/connector/lm_connector.py:15-84, lmcache/server/__main__.py:29-104,
README.md:41-59) -- BASELINE config 3.  MI355X-native form: every rank (one process per GPU) owns the
shard of keys with `owner_rank(key) == rank` and keeps those blobs in its own memory (HBM for the RCCL
path); moving blobs between instances is ONE exchange step of point-to-point sends/receives grouped into
a single `batch_isend_irecv` (ncclGroupStart/End underneath on the "nccl" = RCCL backend), so on an
8-GPU node all seven xGMI links of a GPU carry traffic at once.  There is no all-reduce on this path.

The exchange is SPMD: every rank of the group calls `exchange_put` / `exchange_get` at the same point
(e.g. a prefill instance and a decode instance at a hand-over, or N replicas at a scheduling tick).
Control metadata (who wants which key, which sizes) travels as FIXED-SIZE records in uint8 tensors -- 256 bytes
per (key, three integers), the counterpart of the reference's 158-byte ClientMetaMessage
(lmcache/protocol.py:45-47) -- with one `all_gather` of the counts and one of the padded record arrays: no
pickled Python objects on the wire; payloads travel as uint8 tensors -- device tensors when the group's backend is nccl/RCCL, CPU tensors under gloo (which is
how the logic is tested without GPUs, tests/test_distributed_cpu.py).

Blobs are opaque here (the bytes `CacheGenSerializer.to_bytes` / `lmc_encode_chunks` produce), exactly as
they are opaque to the reference's server (server_storage_backend/local_backend.py:69).
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from lmcache_amd.distributed import owner_rank


class XgmiShardStore:
    def __init__(self, group: Optional[dist.ProcessGroup] = None, device: Optional[torch.device] = None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("XgmiShardStore needs an initialised torch.distributed process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        backend = dist.get_backend(group)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        self.device = device
        self.shard: Dict[str, torch.Tensor] = {}  # key string -> uint8 blob owned by this rank

    # ------------------------------------------------------------------ local view
    def owner(self, key: str) -> int:
        return owner_rank(key, self.world)

    def exists_local(self, key: str) -> bool:
        return key in self.shard

    def list_local(self) -> List[str]:
        return list(self.shard.keys())

    def _as_blob(self, b) -> torch.Tensor:
        if isinstance(b, torch.Tensor):
            t = b.reshape(-1)
            if t.dtype != torch.uint8:
                t = t.view(torch.uint8)
            return t.to(self.device).contiguous()
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(self.device)

    # ------------------------------------------------------------------ control plane: fixed-size records
    _REC = 256  # bytes per record: key (<= 224 bytes, zero padded) | key length | three int64

    def _gather_records(self, recs: Sequence[Tuple[str, int, int, int]]) -> List[List[Tuple[str, int, int, int]]]:
        """all_gather of every rank's list of (key, a, b, c) records -> one list per rank."""
        import struct
        n = len(recs)
        cnt = torch.tensor([n], dtype=torch.int64, device=self.device)
        counts = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(self.world)]
        dist.all_gather(counts, cnt, group=self.group)
        counts = [int(c.item()) for c in counts]
        mx = max(counts + [1])
        buf = bytearray(mx * self._REC)
        for i, (k, a, b, c) in enumerate(recs):
            kb = k.encode("utf-8")
            if len(kb) > 224:
                raise ValueError("key longer than 224 bytes")
            off = i * self._REC
            buf[off:off + len(kb)] = kb
            struct.pack_into("<qqqq", buf, off + 224, len(kb), a, b, c)
        mine = torch.frombuffer(buf, dtype=torch.uint8).to(self.device)
        parts = [torch.empty(mx * self._REC, dtype=torch.uint8, device=self.device) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        out = []
        for r in range(self.world):
            raw = parts[r].cpu().numpy().tobytes()
            lst = []
            for i in range(counts[r]):
                off = i * self._REC
                kl, a, b, c = struct.unpack_from("<qqqq", raw, off + 224)
                lst.append((raw[off:off + kl].decode("utf-8"), a, b, c))
            out.append(lst)
        return out

    # ------------------------------------------------------------------ collective: route blobs to their owners
    def exchange_put(self, items: Sequence[Tuple[str, object]]) -> int:
        """Every rank contributes (key, blob) pairs; each blob ends up in its owner's shard.
        Returns the number of blobs this rank now owns from this call."""
        mine = [(k, self._as_blob(b)) for k, b in items]
        all_meta = [[(k, n, o) for k, n, o, _ in lst]
                    for lst in self._gather_records([(k, int(t.numel()), self.owner(k), 0) for k, t in mine])]
        ops, recvs = [], []
        for k, t in mine:  # my blobs that live elsewhere
            o = self.owner(k)
            if o == self.rank:
                self.shard[k] = t
            else:
                ops.append(dist.P2POp(dist.isend, t, o, self.group))
        got = sum(1 for k, _ in mine if self.owner(k) == self.rank)
        for src in range(self.world):  # blobs other ranks hold for me, in their list order
            if src == self.rank:
                continue
            for k, n, o in all_meta[src]:
                if o == self.rank:
                    buf = torch.empty(n, dtype=torch.uint8, device=self.device)
                    ops.append(dist.P2POp(dist.irecv, buf, src, self.group))
                    recvs.append((k, buf))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for k, buf in recvs:
            self.shard[k] = buf
        return got + len(recvs)

    # ------------------------------------------------------------------ collective: fetch blobs from their owners
    def exchange_get(self, keys: Sequence[str]) -> List[Optional[torch.Tensor]]:
        """Every rank asks for its own list of keys; returns the blobs (None for a miss, never raises on a
        miss -- the contract of LMCBackendInterface.get, abstract_backend.py:47-63)."""
        want = list(keys)
        all_want = [[k for k, _, _, _ in lst] for lst in self._gather_records([(k, 0, 0, 0) for k in want])]
        # what I can serve: (requester, position in its list, size or -1)
        serve = []
        for r in range(self.world):
            for pos, k in enumerate(all_want[r]):
                if self.owner(k) == self.rank:
                    t = self.shard.get(k)
                    serve.append((r, pos, -1 if t is None else int(t.numel())))
        all_serve = [[(r, pos, n) for _, r, pos, n in lst]
                     for lst in self._gather_records([("", r, pos, n) for r, pos, n in serve])]
        out: List[Optional[torch.Tensor]] = [None] * len(want)
        ops = []
        for r, pos, n in serve:  # sends, in my serve order
            if n >= 0 and r != self.rank:
                ops.append(dist.P2POp(dist.isend, self.shard[all_want[r][pos]], r, self.group))
            elif n >= 0:
                out[pos] = self.shard[all_want[r][pos]]
        for o in range(self.world):  # matching receives, in each owner's serve order
            if o == self.rank:
                continue
            for r, pos, n in all_serve[o]:
                if r == self.rank and n >= 0:
                    buf = torch.empty(n, dtype=torch.uint8, device=self.device)
                    ops.append(dist.P2POp(dist.irecv, buf, o, self.group))
                    out[pos] = buf
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return out
