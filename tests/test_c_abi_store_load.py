"""lmc_store_chunks / lmc_load_chunks (include/lmc_hip.h): the host-DRAM legs of store and retrieve through the C ABI
alone -- ctypes and raw pointers, none of the Python sequencing of serde/cachegen_device.py.  What a non-Python
binder of liblmc_hip.so gets where the reference has LMCLocalBackend.put / get
(lmcache/storage_backend/local_backend.py:82-100, 128-144)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def nat():
    from lmcache_amd import native
    native.build()
    return native


@pytest.fixture(scope="module")
def oracle():
    from oracle import lmc_oracle
    lmc_oracle.build()
    return lmc_oracle


def _bits(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("shape", [(4, 2148, 8, 128, 1), (2, 700, 3, 128, 0), (3, 256, 8, 128, 3),
                                   (3, 745, 8, 128, 2, 100), (2, 520, 2, 64, 0, 128), (2, 130, 8, 128, 1, 40)],
                         ids=["9chunks_tail_1layer_ranges", "C384_whole", "1chunk_3layer_ranges",
                              "chunks_of_100_tail_45", "chunks_of_128_C128", "chunks_of_40_tail_10"])
def test_store_then_load_through_the_c_abi_only(nat, oracle, shape):
    L, T, H, D, lpr = shape[:5]
    cs = shape[5] if len(shape) > 5 else 256   # chunk lengths below 256 take the counts model too (round 5)
    n = (T + cs - 1) // cs
    g = torch.Generator().manual_seed(T)
    kv = torch.randn(L, 2, T, H, D, generator=g).to(torch.bfloat16)
    kv_d = kv.to(DEV)
    bins = [32 if l < max(1, L // 3) else 16 for l in range(L)] + [32 if l < 1 else 16 for l in range(L)]
    ctx = nat.get_context(0)
    bound = nat.r16(nat.blob_bound(L, cs, H, D))
    arena = nat.PinnedBuffer(n * bound)
    meta = nat.PinnedBuffer(8 * (n + 1) + 8 * n + 4 * n + 64)     # offsets | blob pointers | sizes | status
    o_ptrs, o_sizes, o_status = 8 * (n + 1), 8 * (n + 1) + 8 * n, 8 * (n + 1) + 12 * n
    offs = meta.tensor[:o_ptrs].view(torch.int64)
    ptrs = meta.tensor[o_ptrs:o_sizes].view(torch.int64)
    sizes = meta.tensor[o_sizes:o_status].view(torch.int32)
    status = meta.tensor[o_status:o_status + 4].view(torch.int32)
    status[0] = 0
    p_offs, p_ptrs, p_sizes, p_status = meta.ptr, meta.ptr + o_ptrs, meta.ptr + o_sizes, meta.ptr + o_status
    st = torch.cuda.Stream(device=DEV)
    lay = nat.KVLayout.from_chunk(kv_d, "vllm")
    # ---- store: one call, returns without waiting; everything is valid once the stream has drained -----------
    ctx.store_chunks(lay, 0, T, cs, bins, arena.ptr, arena.nbytes, p_offs, p_sizes, stream=st.cuda_stream, status_ptr=p_status)
    st.synchronize()
    assert int(status[0]) == 0
    offs_l, sizes_l = offs.tolist(), sizes.tolist()
    assert offs_l[0] == 0 and offs_l[n] == sum(nat.r16(s) for s in sizes_l)
    for i in range(n):
        t0, t1 = i * cs, min(T, (i + 1) * cs)
        blob = ctypes.string_at(arena.ptr + offs_l[i], sizes_l[i])
        b, code = oracle.torch_to_bits(kv[:, :, t0:t1].reshape(L, 2, t1 - t0, H * D))
        assert blob == oracle.encode_blob(b, code, H, D, np.array(bins, np.int32)), f"chunk {i}"
        assert offs_l[i + 1] == offs_l[i] + nat.r16(sizes_l[i])
    # ---- load: gather + decode range by range; the events say when a range of layers is complete --------------
    for i in range(n):
        ptrs[i] = arena.ptr + offs_l[i]
    out = torch.zeros_like(kv_d)
    nranges = (L + lpr - 1) // lpr if lpr else 1
    ev_handles = (ctypes.c_void_p * nranges)()
    for r in range(nranges):
        h = ctypes.c_void_p()
        nat.check(nat.lib().lmc_event_create(ctypes.byref(h), 0), "lmc_event_create")
        ev_handles[r] = h
    ctx.load_chunks(p_ptrs, p_sizes, n, nat.KVLayout.from_chunk(out, "vllm"), 0, cs, lpr,
                    ctypes.cast(ev_handles, ctypes.c_void_p).value, stream=st.cuda_stream, status_ptr=p_status)
    for r in range(nranges):  # the last event implies the earlier ones; every one of them fires
        nat.check(nat.lib().lmc_event_synchronize(ev_handles[r]), "lmc_event_synchronize")
    st.synchronize()
    assert int(status[0]) == 0
    for i in range(n):
        t0, t1 = i * cs, min(T, (i + 1) * cs)
        b, code = oracle.torch_to_bits(kv[:, :, t0:t1].reshape(L, 2, t1 - t0, H * D))
        ref = oracle.decode_blob(oracle.encode_blob(b, code, H, D, np.array(bins, np.int32)), oracle.BF16)
        assert np.array_equal(_bits(out[:, :, t0:t1]).reshape(L, 2, t1 - t0, H * D), ref), f"chunk {i}"
    for r in range(nranges):
        nat.lib().lmc_event_destroy(ev_handles[r])
    # ---- an arena that is too small: flagged, sizes of the blobs that did not fit read 0, nothing written past it
    status[0] = 0
    small = offs_l[n] - 16
    ctx.store_chunks(lay, 0, T, cs, bins, arena.ptr, small, p_offs, p_sizes, stream=st.cuda_stream, status_ptr=p_status)
    st.synchronize()
    assert int(status[0]) & 32 and sizes.tolist()[-1] == 0 and all(s > 0 for s in sizes.tolist()[:-1])
    arena.free()
    meta.free()


@pytest.mark.parametrize("shape", [(4, 2148, 8, 128, 1, 5), (2, 700, 3, 128, 0, 0), (3, 512, 8, 128, 2, 1),
                                   (3, 745, 8, 128, 2, 3, 100), (2, 300, 2, 64, 1, 2, 40)],
                         ids=["9chunks_tail_1layer_ranges_prefix5", "C384_whole", "2chunks_prefix1",
                              "chunks_of_100_tail_45_prefix3", "chunks_of_40_tail_20_C128_prefix2"])
def test_pack_store_extract_load_through_the_c_abi_only(nat, oracle, shape):
    """lmc_store_pack / lmc_pack_extract / lmc_load_pack: the layer-major form of the pinned tier.  Every chunk
    extracted from the pack is the oracle's blob byte for byte; loading the whole pack, and a prefix of its chunks,
    range by range gives the oracle's decode; a pack region that is too small is flagged and leaves no pack."""
    L, T, H, D, lpr, prefix = shape[:6]
    cs = shape[6] if len(shape) > 6 else 256
    n = (T + cs - 1) // cs
    g = torch.Generator().manual_seed(T + 1)
    kv = torch.randn(L, 2, T, H, D, generator=g).to(torch.bfloat16)
    kv_d = kv.to(DEV)
    bins = [32 if l < max(1, L // 3) else 16 for l in range(L)] + [32 if l < 1 else 16 for l in range(L)]
    ctx = nat.get_context(0)
    cap = nat.pack_bound(n, L, cs, H, D)
    region = nat.PinnedBuffer(cap)
    meta = nat.PinnedBuffer(4 * n + 64)
    sizes = meta.tensor[:4 * n].view(torch.int32)
    status = meta.tensor[4 * n:4 * n + 4].view(torch.int32)
    status[0] = 0
    st = torch.cuda.Stream(device=DEV)
    lay = nat.KVLayout.from_chunk(kv_d, "vllm")
    ctx.store_pack(lay, 0, T, cs, bins, region.ptr, cap, meta.ptr, stream=st.cuda_stream, status_ptr=meta.ptr + 4 * n)
    st.synchronize()
    assert int(status[0]) == 0
    h = nat.pack_info(region.ptr, cap)
    assert (h.nchunks, h.num_layers, h.num_heads, h.head_size, h.chunk_tokens, h.ntokens) == (n, L, H, D, cs, T)
    blobs = []
    for i in range(n):
        t0, t1 = i * cs, min(T, (i + 1) * cs)
        b, code = oracle.torch_to_bits(kv[:, :, t0:t1].reshape(L, 2, t1 - t0, H * D))
        blobs.append(oracle.encode_blob(b, code, H, D, np.array(bins, np.int32)))
        got = nat.pack_extract(region.ptr, h.total_bytes, i)
        assert got == blobs[i], f"chunk {i}"
        assert int(sizes[i]) == len(blobs[i])
    # ... and the pack as a whole is the oracle's restatement of the layout, byte for byte
    assert ctypes.string_at(region.ptr, h.total_bytes) == oracle.pack_from_blobs(blobs, cs)
    # the pack holds the blobs' bytes once, plus its header, table and slot padding
    # (a ragged last chunk has shorter static sections than its slot)
    assert h.total_bytes <= sum(len(b) for b in blobs) + 256 + 8 * (2 * L * n + 1) + 16 + 16 * n + 4 * L * cs + 64
    refs = [oracle.decode_blob(b, oracle.BF16) for b in blobs]

    def load(m, c0=0):
        out = torch.zeros_like(kv_d)
        nranges = (L + lpr - 1) // lpr if lpr else 1
        ev = (ctypes.c_void_p * nranges)()
        for r in range(nranges):
            e = ctypes.c_void_p()
            nat.check(nat.lib().lmc_event_create(ctypes.byref(e), 0), "lmc_event_create")
            ev[r] = e
        status[0] = 0
        ctx.load_pack(region.ptr, h.total_bytes, c0, m, nat.KVLayout.from_chunk(out, "vllm"), c0 * cs, lpr,
                      ctypes.cast(ev, ctypes.c_void_p).value, stream=st.cuda_stream, status_ptr=meta.ptr + 4 * n)
        for r in range(nranges):
            nat.check(nat.lib().lmc_event_synchronize(ev[r]), "lmc_event_synchronize")
            nat.lib().lmc_event_destroy(ev[r])
        st.synchronize()
        assert int(status[0]) == 0
        mm = m or n - c0
        for i in range(n):
            t0, t1 = i * cs, min(T, (i + 1) * cs)
            got = _bits(out[:, :, t0:t1]).reshape(L, 2, t1 - t0, H * D)
            if c0 <= i < c0 + mm:
                assert np.array_equal(got, refs[i]), f"chunk {i} of [{c0}, {c0 + mm})"
            else:
                assert not got.any(), f"chunk {i} is outside the run and must stay untouched"

    load(0)
    if prefix:
        load(prefix)
    if n > 2:  # a run in the middle, and the tail from chunk 1 on
        load(n - 2, 1)
        load(0, 1)
    # a corrupt offset table is refused on the host, before anything is queued
    tab = ctypes.cast(region.ptr + h.off_table, ctypes.POINTER(ctypes.c_uint64))
    keep = tab[1]
    tab[1] = keep + 16 if n * 2 * L > 1 else keep
    tab[2 * L * n] += 16
    with pytest.raises(nat.NativeError):
        ctx.load_pack(region.ptr, h.total_bytes, 0, 0, nat.KVLayout.from_chunk(torch.zeros_like(kv_d), "vllm"), 0, lpr, None,
                      stream=st.cuda_stream, status_ptr=meta.ptr + 4 * n)
    # a region that is too small: flagged, and what it holds is not a pack
    status[0] = 0
    ctx.store_pack(lay, 0, T, cs, bins, region.ptr, int(h.total_bytes) - 16, meta.ptr, stream=st.cuda_stream,
                   status_ptr=meta.ptr + 4 * n)
    st.synchronize()
    assert int(status[0]) & 32
    with pytest.raises(nat.NativeError):
        nat.pack_info(region.ptr, cap)
    region.free()
    meta.free()


@pytest.mark.parametrize("shape", [(4, 2048, 8, 128, 256, 4, True), (3, 1024, 4, 128, 128, 3, True), (4, 2048, 8, 128, 256, 16, True),
                                   (4, 2148, 8, 128, 256, 4, True), (2, 512, 8, 128, 256, 4, False)],
                         ids=["8planes_4parts", "6planes_3parts_chunks_of_128", "more_parts_than_items", "ragged_tail_is_one_part",
                              "small_job_two_kernels_is_one_part"])
def test_pack_stored_in_parts_is_the_same_pack(nat, oracle, shape):
    """lmc_store_pack_parts (round 6): the encode launched in plane ranges, each range packed into a DEVICE region as soon
    as it is coded.  After part r its bytes lie at their final place (part_info: offset, bytes) and its event has been
    recorded; [0, off_streams) is final after the last part.  Assembled from exactly those pieces, the pack is the
    oracle's restatement of the layout byte for byte -- the same bytes lmc_store_pack writes in one piece."""
    L, T, H, D, cs, nparts, fused = shape
    n = (T + cs - 1) // cs
    g = torch.Generator().manual_seed(T + L)
    kv = torch.randn(L, 2, T, H, D, generator=g).to(torch.bfloat16)
    kv_d = kv.to(DEV)
    bins = [32 if l < max(1, L // 3) else 16 for l in range(L)] + [32 if l < 1 else 16 for l in range(L)]
    ctx = nat.get_context(0)
    cap = nat.pack_bound(n, L, cs, H, D)
    dev = torch.zeros(cap, dtype=torch.uint8, device=DEV)
    host = nat.PinnedBuffer(cap)
    n4 = (4 * n + 7) & ~7  # (the part words are uint64)
    meta = nat.PinnedBuffer(n4 + 64 + 16 * 16)
    sizes = meta.tensor[:4 * n].view(torch.int32)
    status = meta.tensor[n4:n4 + 4].view(torch.int32)
    info = meta.tensor[n4 + 64:n4 + 64 + 16 * nparts].view(torch.int64)
    status[0] = 0
    st = torch.cuda.Stream(device=DEV)
    cp = torch.cuda.Stream(device=DEV)
    lay = nat.KVLayout.from_chunk(kv_d, "vllm")
    events = [nat.NativeEvent() for _ in range(nparts)]
    ctx.set_encode_path("fused" if fused else "auto")
    try:
        ctx.store_pack_parts(lay, 0, T, cs, bins, dev.data_ptr(), cap, meta.ptr, nparts, meta.ptr + n4 + 64, events,
                             stream=st.cuda_stream, status_ptr=meta.ptr + n4)
    finally:
        ctx.set_encode_path("auto")
    off_streams = nat.pack_off_streams(n, L, cs, H, D)
    moved, nonempty = 0, 0
    for r, ev in enumerate(events):  # what CacheGenDeviceCodec.finish_pack does: one DMA per part as its event fires
        ev.synchronize()
        off, nb = int(info[2 * r]), int(info[2 * r + 1])
        assert nb == 0 or off == moved, "the parts are consecutive pieces of the streams region"
        if nb:
            nat.memcpy_async(host.ptr + off_streams + off, dev.data_ptr() + off_streams + off, nb, "d2h", cp.cuda_stream)
            nonempty += 1
        moved += nb
    assert int(status[0]) == 0
    nat.memcpy_async(host.ptr, dev.data_ptr(), off_streams, "d2h", cp.cuda_stream)
    cp.synchronize()
    st.synchronize()
    split = fused and T % cs == 0
    # (a part packs the planes coded so far but the newest one, whose end its successor writes: a first range of a single
    # plane ships nothing, and ranges there are no items for stay empty)
    assert (min(nparts, 2 * L) - 1 <= nonempty <= min(nparts, 2 * L)) if split else nonempty == 1
    h = nat.pack_info(host.ptr, cap)
    assert h.total_bytes == off_streams + moved
    blobs = []
    for i in range(n):
        t0, t1 = i * cs, min(T, (i + 1) * cs)
        b, code = oracle.torch_to_bits(kv[:, :, t0:t1].reshape(L, 2, t1 - t0, H * D))
        blobs.append(oracle.encode_blob(b, code, H, D, np.array(bins, np.int32)))
        assert nat.pack_extract(host.ptr, h.total_bytes, i) == blobs[i], f"chunk {i}"
        assert int(sizes[i]) == len(blobs[i])
    assert ctypes.string_at(host.ptr, h.total_bytes) == oracle.pack_from_blobs(blobs, cs)
    # a region that is too small: flagged, every part reads "no bytes" from the failing one on, and there is no pack header
    status[0] = 0
    ctx.set_encode_path("fused" if fused else "auto")
    try:
        ctx.store_pack_parts(lay, 0, T, cs, bins, dev.data_ptr(), int(h.total_bytes) - 16, meta.ptr, nparts,
                             meta.ptr + n4 + 64, events, stream=st.cuda_stream, status_ptr=meta.ptr + n4)
    finally:
        ctx.set_encode_path("auto")
    st.synchronize()
    assert int(status[0]) & 32
    assert int(info[2 * (len(events) - 1) + 1]) == 0 or not split
    hdr = dev[:256].cpu().numpy().view(np.uint32)
    assert hdr[0] == 0, "a failed pack must not look like one"
    host.free()
    meta.free()
