"""CPU-only tests of the host side: config, keys, hash index, retrieve planning,
factories, blob header parsing and the C-ABI export list.  No GPU compute."""
import json
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
from lmcache_amd.utils import CacheEngineKey

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bare_engine(chunk_size, fmt="vllm"):
    """An engine without a backend: the index logic is pure Python."""
    from lmcache_amd.cache_engine import LMCacheEngine
    e = LMCacheEngine.__new__(LMCacheEngine)
    e.chunk_size = chunk_size
    e.metadata = LMCacheEngineMetadata("test_model", 3, 123, fmt, "half")
    return e


def test_key_string_roundtrip_and_golden(golden_dir):
    with open(os.path.join(golden_dir, "hash_chain.json")) as f:
        gold = json.load(f)
    key = CacheEngineKey(*gold["key_fields"])
    assert key.to_string() == gold["key_string"]
    assert CacheEngineKey.from_string(gold["key_string"]) == key
    assert hash(key) == hash(CacheEngineKey.from_string(key.to_string()))
    with pytest.raises(ValueError):
        CacheEngineKey.from_string("a@b@c")
    assert len(key.to_string()) <= 150  # lm:// header limit (protocol.py:4,28-30)


def test_prefix_hash_matches_reference_and_oracle(golden_dir, oracle):
    with open(os.path.join(golden_dir, "hash_chain.json")) as f:
        gold = json.load(f)
    for case in gold["cases"]:
        e = _bare_engine(case["chunk_size"])
        toks = torch.tensor(case["tokens"], dtype=torch.int64)
        got = e._prefix_hash(e._chunk_tokens(toks))
        assert got == case["hashes"], case["name"]                       # reference-generated
        assert got == oracle.prefix_hash(toks.numpy(), case["chunk_size"])  # independent C SHA-256
        assert e._prefix_hash(e._chunk_tokens(toks), 1) == case["hashes"][1:]
        # the engine's one-conversion form gives the same chain
        assert e._prefix_hashes_of(toks) == case["hashes"] and e._prefix_hashes_of(toks, 1) == case["hashes"][1:]
    # dtype dependence noted in SURVEY.md 8(c): int32 tokens hash differently
    e = _bare_engine(256)
    t64 = torch.arange(300, dtype=torch.int64)
    assert e._prefix_hash(e._chunk_tokens(t64)) != e._prefix_hash(e._chunk_tokens(t64.to(torch.int32)))


def test_retrieve_planning_matches_reference_semantics(golden_dir):
    """ret_mask / hit length for prefix, diverging, extended and suffix-masked queries, as produced by the
    reference engine (tests/golden/engine_semantics.json)."""
    from lmcache_amd.cache_engine import LMCacheEngine
    with open(os.path.join(golden_dir, "engine_semantics.json")) as f:
        gold = json.load(f)
    for rec in gold:
        cs = rec["chunk_size"]
        e = _bare_engine(cs, rec["fmt"])
        stored = set(e._prefix_hash(e._chunk_tokens(torch.tensor(rec["tokens"]))))
        for q in rec["queries"]:
            toks = torch.tensor(q["tokens"])
            skip = q["skip"] or 0
            hashes = e._prefix_hash(e._chunk_tokens(toks), skip // cs)
            hits = 0
            for h in hashes:
                if h not in stored:
                    break
                hits += 1
            _, extra, nret = LMCacheEngine.plan_retrieve(len(toks), cs, skip, hits)
            assert nret == q["ret_tokens"], q["name"]
            mask = torch.ones(len(toks), dtype=torch.bool)
            mask[:skip] = False
            if nret == 0:
                mask[:] = False
            else:
                mask[skip + nret:] = False
            assert mask.to(torch.int8).tolist() == q["ret_mask"], q["name"]


def test_config_loaders(tmp_path):
    c = LMCacheEngineConfig.from_defaults()
    assert (c.chunk_size, c.local_device, c.remote_url, c.remote_serde) == (256, "cuda", "redis://localhost:6379", "torch")
    assert LMCacheEngineConfig.from_legacy(backend="cpu").local_device == "cpu"
    assert LMCacheEngineConfig.from_legacy(backend="file://local_disk/").local_device == "local_disk/"
    c = LMCacheEngineConfig.from_legacy(backend="lm://localhost:65000")
    assert c.local_device is None and c.remote_url == "lm://localhost:65000"
    # positional construction stays the reference's six fields
    c = LMCacheEngineConfig(128, "cpu", None, "cachegen", False, True)
    assert c.local_serde is None and c.save_decode_cache
    p = tmp_path / "a.yaml"
    p.write_text("chunk_size: 64\nlocal_device: cpu\nremote_url: null\nlocal_serde: cachegen\n")
    c = LMCacheEngineConfig.from_file(str(p))
    assert (c.chunk_size, c.local_device, c.remote_url, c.remote_serde, c.local_serde) == (64, "cpu", None, "torch", "cachegen")
    p.write_text("local_device: file://somewhere/\nremote_url: lm://h:1\n")
    c = LMCacheEngineConfig.from_file(str(p))
    assert c.local_device == "somewhere/" and c.remote_url == "lm://h:1" and c.chunk_size == 256
    p.write_text("local_device: tape\n")
    with pytest.raises(ValueError):
        LMCacheEngineConfig.from_file(str(p))
    p.write_text("remote_url: nonsense\n")
    with pytest.raises(ValueError):
        LMCacheEngineConfig.from_file(str(p))


def test_cachegen_tables(oracle):
    from lmcache_amd.storage_backend.serde.cachegen_basics import CacheGenConfig
    # the five models the reference knows, against tables generated by importing the reference
    # (oracle/gen_golden.py gen_bins -> tests/golden/bins.json): product and oracle restate them identically
    import json
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "bins.json")))
    assert len(golden) == 5
    for name, g in golden.items():
        cfg = CacheGenConfig.from_model_name(name)
        assert cfg.key_bins() == g["key_bins"] and cfg.value_bins() == g["value_bins"], name
        bins, nl = oracle.cachegen_bins(name)
        assert bins.tolist() == g["key_bins"] + g["value_bins"] and nl == len(g["key_bins"])
    for name in ("mistralai/Mistral-7B-Instruct-v0.2", "meta-llama/Llama-3.1-8B-Instruct", "THUDM/glm-4-9b-chat"):
        cfg = CacheGenConfig.from_model_name(name)
        bins, nl = oracle.cachegen_bins(name)
        assert cfg.plane_bins(nl) == bins.tolist()
        assert cfg["key_first_bins"] == 32 and cfg.key_third_layers == nl
    assert len(CacheGenConfig.from_model_name("Llama-3-70B").key_bins()) == 80
    with pytest.raises(ValueError):
        CacheGenConfig.from_model_name("not/a-model")


def test_factories_reject_bad_configs():
    from lmcache_amd.storage_backend import CreateStorageBackend
    from lmcache_amd.storage_backend.connector import CreateConnector
    from lmcache_amd.storage_backend.serde import CreateSerde
    meta = LMCacheEngineMetadata("test_model", 1, 0, "vllm", "half")
    with pytest.raises(ValueError):
        CreateStorageBackend(LMCacheEngineConfig(256, None, None, "torch", False, False), meta)
    with pytest.raises(ValueError):
        CreateSerde("bogus", LMCacheEngineConfig.from_defaults(), meta)
    with pytest.raises(ValueError):  # unknown model for cachegen (cachegen_basics.py:77-78)
        CreateSerde("cachegen", LMCacheEngineConfig.from_defaults(), meta)
    with pytest.raises(ValueError):
        CreateConnector("notaurl")
    s, d = CreateSerde("torch", LMCacheEngineConfig.from_defaults(), meta)
    t = torch.rand(2, 2, 5, 3, 8).to(torch.bfloat16)
    assert torch.equal(d.from_bytes(s.to_bytes(t)), t)
    c = CreateConnector("mem://unit:1")
    c.set("k", b"v")
    assert c.exists("k") and c.get("k") == b"v" and c.list() == ["k"] and not c.exists("x")


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads and exports exactly what include/lmc_hip.h declares."""
    from lmcache_amd import native
    hdr = open(os.path.join(ROOT, "include", "lmc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lmc_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"lmc_r16", "lmc_blob_layout", "lmc_group_cap_bytes", "lmc_blob_bound"}  # static inline (lmc_format.h)
    declared -= {n for n in declared if n not in native.SYMBOLS and re.search(r"static inline[^;{]*\b" + n + r"\s*\(",
                 open(os.path.join(ROOT, "include", "lmc_format.h")).read())}
    assert declared == set(native.SYMBOLS), declared ^ set(native.SYMBOLS)
    L = native.lib()  # raises if any symbol is missing
    out = subprocess.check_output(["nm", "-D", "--defined-only", native.SO_PATH], text=True)
    exported = set(re.findall(r" T (lmc_[a-z0-9_]+)", out))
    assert declared <= exported
    assert L.lmc_abi_version() == 6
    assert L.lmc_strerror(-1).decode().startswith("invalid")


def test_blob_header_parse_on_host(oracle):
    """lmc_blob_info / CacheGenEncoderOutput.from_bytes on an oracle-made blob (tests/test_serde.py:59-62 analogue)."""
    from lmcache_amd import native
    from lmcache_amd.storage_backend.serde.cachegen_basics import CacheGenEncoderOutput
    torch.manual_seed(1)
    L, T, H, D = 2, 20, 8, 128
    kv = torch.rand(L, 2, T, H * D).to(torch.bfloat16)
    bits, code = oracle.torch_to_bits(kv)
    bins = np.array([32, 16, 32, 16], np.int32)
    blob = oracle.encode_blob(bits, code, H, D, bins)
    h = native.blob_info(blob)
    assert (h.num_layers, h.ntokens, h.num_heads, h.head_size, h.total_bytes) == (L, T, H, D, len(blob))
    assert native.blob_bound(L, T, H, D) == oracle.blob_bound(L, T, H, D)
    out = CacheGenEncoderOutput.from_bytes(blob)
    assert out.num_heads == 8 and out.head_size == 128
    assert out.bins == bins.tolist()
    sym, scale = oracle.quantize(bits, code, bins)
    assert np.array_equal(out.cdf.numpy().view(np.uint16), oracle.cdf(sym))
    assert np.array_equal(out.max_tensors_key.view(torch.int16).numpy().view(np.uint16)[..., 0], scale[:L])
    assert out.data_chunks[0].ntokens == T
    assert int(out.data_chunks[0].bytestream_lengths.sum()) <= h.stream_bytes
    with pytest.raises(native.NativeError):
        native.blob_info(b"\0" * 200)
    with pytest.raises(native.NativeError):
        native.blob_info(blob[:1000])  # truncated: total_bytes > len


def test_product_never_imports_the_oracle():
    """The product path must not route through oracle/ (checked statically)."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lmcache_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "lmc_oracle" in src or "liblmc_oracle" in src:
                    bad.append(fn)
    assert not bad, bad


def test_pipelined_remote_backend_keeps_one_result_per_key():
    """Host logic of LMCPipelinedRemoteBackend (row f3) with the lossless torch serde on CPU tensors:
    the factory picks it for pipelined_backend=True, batched_get lines results up with the keys (the
    reference's result_list drops misses, remote_backend.py:224-243), async puts land, close() joins."""
    import torch
    from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
    from lmcache_amd.storage_backend import CreateStorageBackend
    from lmcache_amd.storage_backend.remote_backend import LMCPipelinedRemoteBackend, LMCRemoteBackend
    from lmcache_amd.utils import CacheEngineKey
    meta = LMCacheEngineMetadata("test_model", 1, 0, "vllm", "half")
    plain = CreateStorageBackend(LMCacheEngineConfig.from_legacy(backend="mem://hostpipe:7", remote_serde="torch"), meta)
    assert type(plain) is LMCRemoteBackend
    plain.close()
    be = CreateStorageBackend(LMCacheEngineConfig.from_legacy(backend="mem://hostpipe:7", remote_serde="torch",
                                                              pipelined_backend=True), meta)
    try:
        assert isinstance(be, LMCPipelinedRemoteBackend) and not be.supports_kv_layout
        be.dst_device = "cpu"
        keys = [CacheEngineKey("vllm", "test_model", 1, 0, f"{i:064x}") for i in range(6)]
        vals = [torch.full((2, 2, 4, 1, 8), float(i), dtype=torch.bfloat16) for i in range(6)]
        for i in (0, 1, 3):
            be.put(keys[i], vals[i], blocking=True)
        be.put(keys[5], vals[5], blocking=False)
        import time
        for _ in range(200):
            if be.contains(keys[5]):
                break
            time.sleep(0.01)
        got = be.batched_get(iter(keys))
        assert len(got) == 6
        assert [g is None for g in got] == [False, False, True, False, True, False]
        for i in (0, 1, 3, 5):
            assert torch.equal(got[i], vals[i])
        assert be.batched_get(iter([])) == []
    finally:
        be.close()
    assert be._fetcher is None


@pytest.mark.parametrize("T", [256, 255, 300, 1])
def test_blob_counts_section_rebuilds_the_reference_cdf(oracle, T):
    """The blob stores symbol COUNTS (format v6: bit-sliced in the head of every group stream; for T <= 256 a count of
    256 saturates to 255) and the CDF is a function of them: the numpy rebuild of CacheGenEncoderOutput.cdf and the
    oracle's C rebuild must both give the CDF computed from the symbols, including channels whose every token carries
    the same symbol."""
    from lmcache_amd import native
    from lmcache_amd.storage_backend.serde.cachegen_basics import CacheGenEncoderOutput
    torch.manual_seed(T)
    L, H, D = 2, 2, 64
    kv = torch.randn(L, 2, T, H * D).to(torch.bfloat16)
    kv[:, :, :, 5] = 0          # constant channel: count == T on one symbol (the middle one)
    kv[0, 0, :, 7] = kv[0, 0, :, :].abs().amax(dim=-1)  # always the row max: top symbol every token
    bits, code = oracle.torch_to_bits(kv)
    bins = np.array([32, 16, 32, 16], np.int32)
    blob = oracle.encode_blob(bits, code, H, D, bins)
    h = native.blob_info(blob)
    assert h.version == 6 and h.model == (1 if 2 <= T <= 256 else 0)  # round 5: the counts model codes 2 .. 256 tokens
    assert h.off_streams - h.off_gdir == native.r16(8 * 2 * L * ((H * D + 63) // 64))
    assert native.blob_static_bytes(L, T, H, D) == h.off_streams
    sym, _ = oracle.quantize(bits, code, bins)
    want = oracle.cdf(sym)
    assert np.array_equal(CacheGenEncoderOutput.from_bytes(blob).cdf.numpy().view(np.uint16), want)
    assert np.array_equal(oracle.blob_cdf(blob), want)
    assert np.array_equal(oracle.decode_blob_symbols(blob), sym)
    if T == 256:  # the saturated count is really there: channel 5 of plane 0 = lane 5 of its first stream
        _, cnt, _ = oracle.stream_head(blob, 0)
        assert cnt[:, 5].max() == 255 and cnt[:, 5].sum() == 255


def test_divmod_small_is_exact_for_every_row_length():
    """The kernels split e into (e / R, e % R) with one float multiply (lmc_device.h divmod_small); R is the
    number of counts per channel, 3 .. 31, e < 64 * R."""
    for R in range(2, 32):
        e = np.arange(64 * R, dtype=np.uint32)
        q = ((e.astype(np.float32) + np.float32(0.5)) * (np.float32(1.0) / np.float32(R))).astype(np.uint32)
        assert np.array_equal(q, e // R), R


def test_layer_range_schedules():
    """retrieve_layerwise's layers_per_launch: a size, or a schedule whose last entry repeats."""
    from lmcache_amd.storage_backend.serde.cachegen_device import layer_ranges
    assert layer_ranges(32, None) == [(0, 32)] and layer_ranges(32, 0) == [(0, 32)]
    assert layer_ranges(32, 8) == [(0, 8), (8, 16), (16, 24), (24, 32)]
    assert layer_ranges(32, (2, 6, 24)) == [(0, 2), (2, 8), (8, 32)]
    assert layer_ranges(32, (2, 2, 4, 8, 16)) == [(0, 2), (2, 4), (4, 8), (8, 16), (16, 32)]
    assert layer_ranges(5, (2,)) == [(0, 2), (2, 4), (4, 5)] and layer_ranges(3, 7) == [(0, 3)]
    for L in (1, 7, 80):
        for sched in (1, 3, (1, 2), (4, 1), (100,)):
            r = layer_ranges(L, sched)
            assert r[0][0] == 0 and r[-1][1] == L and all(a[1] == b[0] for a, b in zip(r, r[1:]))


def test_encode_path_names_follow_the_header():
    """native.ENCODE_PATHS (what Context.set_encode_path takes) == the LMC_ENCODE_PATH_* constants of lmc_hip.h."""
    from lmcache_amd import native
    hdr = open(os.path.join(ROOT, "include", "lmc_hip.h")).read()
    consts = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"#define LMC_ENCODE_PATH_(\w+)\s+(\d+)", hdr)}
    assert consts == native.ENCODE_PATHS


def test_pack_info_and_extract_on_the_host():
    """lmc_pack_info / lmc_pack_extract are host-only (no GPU): a pack built by the oracle's restatement of the layout
    from oracle blobs (a ragged last chunk included) checks out, every chunk comes back byte for byte, and damaged
    headers / tables are refused."""
    import ctypes
    import numpy as np
    from lmcache_amd import native
    from oracle import lmc_oracle as oracle
    oracle.build()
    L, H, D, cs, T = 2, 3, 128, 256, 600
    rng = np.random.default_rng(11)
    kv = (rng.standard_normal((L, 2, T, H * D)).astype(np.float32))
    bins = np.array([32, 16, 17, 16], np.int32)
    blobs = []
    for t0 in range(0, T, cs):
        bits = np.ascontiguousarray((kv[:, :, t0:t0 + cs].view(np.uint32) >> 16).astype(np.uint16))  # truncated bf16: any bits do
        blobs.append(oracle.encode_blob(bits, oracle.BF16, H, D, bins))
    pack = oracle.pack_from_blobs(blobs, cs)
    buf = ctypes.create_string_buffer(pack, len(pack))
    ptr = ctypes.addressof(buf)
    assert ptr % 16 == 0
    h = native.pack_info(ptr, len(pack))
    assert (h.nchunks, h.num_layers, h.num_heads, h.head_size, h.chunk_tokens, h.ntokens) == (3, L, H, D, cs, T)
    assert h.total_bytes == len(pack)
    for i, b in enumerate(blobs):
        assert native.pack_extract(ptr, len(pack), i) == b
    with pytest.raises(native.NativeError):
        native.pack_extract(ptr, len(pack), 3)
    # damage: magic, a table entry out of order, a total that does not match, a truncated buffer
    for off, val in ((0, b"\x00"), (256 + 8 * 2, b"\xff\xff\xff\xff\xff\xff\xff\x7f"), (72, None)):
        bad = bytearray(pack)
        if val is None:
            bad[off] ^= 0x10  # total_bytes off by 16
        else:
            bad[off:off + len(val)] = val
        bb = ctypes.create_string_buffer(bytes(bad), len(bad))
        with pytest.raises(native.NativeError):
            native.pack_info(ctypes.addressof(bb), len(bad))
    with pytest.raises(native.NativeError):
        native.pack_info(ptr, len(pack) - 16)


def test_pack_cap_covers_the_least_compressible_input():
    """serde/cachegen_device.pack_cap sizes the region a pack is written to from the coder's bound (T * log2(symbols)
    + 48 bits per lane).  Uniform noise over a wide range is the least compressible input the quantiser can see
    (every symbol about equally likely); packs of it, and of a ragged job, must fit."""
    import numpy as np
    from lmcache_amd.storage_backend.serde.cachegen_device import pack_cap
    from oracle import lmc_oracle as oracle
    oracle.build()
    L, H, D, cs = 2, 3, 128, 256
    bins = [32, 17, 16, 32]
    rng = np.random.default_rng(3)
    for T in (512, 300):
        x = rng.uniform(-1.0, 1.0, (L, 2, T, H * D)).astype(np.float32)
        bits = np.ascontiguousarray((x.view(np.uint32) >> 16).astype(np.uint16))  # bf16 bits (truncated): uniform symbols
        blobs = [oracle.encode_blob(np.ascontiguousarray(bits[:, :, t0:t0 + cs]), oracle.BF16, H, D, np.array(bins, np.int32))
                 for t0 in range(0, T, cs)]
        pack = oracle.pack_from_blobs(blobs, cs)
        cap = pack_cap(len(blobs), L, cs, H, D, bins)
        assert len(pack) <= cap, (T, len(pack), cap)
        if T % cs == 0:  # (a ragged last chunk is sized like a full one)
            assert cap < 1.35 * len(pack), "the bound should stay close to the least compressible case"


def test_disk_backend_reads_and_writes_the_reference_files(golden_dir, tmp_path):
    """LMCLocalDiskBackend, raw mode (lmcache/storage_backend/local_backend.py:163-310): the file a chunk is stored in
    -- name and safetensors payload -- is the reference's.  tests/golden/disk holds a file the reference itself wrote
    (oracle/gen_golden.py: gen_disk)."""
    import shutil
    from safetensors import safe_open
    from lmcache_amd.storage_backend import CreateStorageBackend
    from lmcache_amd.storage_backend.local_backend import LMCLocalDiskBackend
    rec = json.load(open(os.path.join(golden_dir, "disk", "disk.json")))
    want = torch.from_numpy(np.array(rec["bits"], np.uint16).view(np.int16)).view(torch.bfloat16).reshape(rec["shape"])
    d = str(tmp_path) + "/"
    cfg = LMCacheEngineConfig.from_legacy(chunk_size=16, backend="file://" + d)
    be = CreateStorageBackend(cfg, LMCacheEngineMetadata("test_model", 1, 0, "vllm", "half"))
    assert isinstance(be, LMCLocalDiskBackend)
    be.dst_device = "cpu"  # "cuda" is hard-coded, as in the reference (:203)
    key = CacheEngineKey.from_string(rec["key"])
    try:
        assert not be.contains(key) and be.get(key) is None  # a miss is None, never an exception
        # our file == the reference's file, byte for byte
        be.put(key, want, blocking=True)
        assert be.contains(key)
        assert os.listdir(d) == [rec["file"]]
        assert open(d + rec["file"], "rb").read() == open(os.path.join(golden_dir, "disk", rec["file"]), "rb").read()
        assert torch.equal(be.get(key), want)
        # the reference's file is read
        shutil.copy(os.path.join(golden_dir, "disk", rec["file"]), d + rec["file"])
        got = be.get(key)
        assert got.dtype == torch.bfloat16 and torch.equal(got, want)
        # the queued put (:254-275): visible once the worker has written the file
        key2 = CacheEngineKey("vllm", "test_model", 1, 0, "cd" * 32)
        be.put(key2, want * 2, blocking=False)
        be.close()
        assert be.contains(key2) and torch.equal(be.get(key2), want * 2)
        with safe_open(be._key_to_path(key2), framework="pt", device="cpu") as f:
            assert list(f.keys()) == ["kv_chunk"]
        assert not [n for n in os.listdir(d) if n.endswith(".tmp")]
    finally:
        be.close()
    # the reference's hybrid backend has no disk tier (hybrid_backend.py:29)
    with pytest.raises(ValueError):
        CreateStorageBackend(LMCacheEngineConfig(16, d, "mem://x:1", "torch", False, False),
                             LMCacheEngineMetadata("test_model", 1, 0, "vllm", "half"))


def test_scratch_of_the_hot_kernels_is_what_design_md_says():
    """round 4's commit log said "without spills" about a kernel that spilled 68 bytes.  native.build()
    keeps the compiler's own resource remarks of the build that produced the library (kernel_resources.json); the hot
    kernels' scratch is pinned here to what DESIGN.md section 4 states: none in the quantiser and the decoder; in the two
    64-VGPR coders a few long-lived values of the stream prologue (lane index, a 64-bit column pointer: reloaded per
    stream, never inside a token loop) -- a bound, so that a regression is a test failure, not a sentence."""
    import json
    from lmcache_amd import native
    native.build()
    if not os.path.exists(native.RESOURCES_PATH):
        pytest.skip("the library was not compiled here (prebuilt): no kernel_resources.json to hold against DESIGN.md")
    res = json.load(open(native.RESOURCES_PATH))

    def one(prefix):
        hits = {k: v for k, v in res.items() if prefix in k}  # (mangled or demangled names, whichever the build could write)
        assert hits, prefix
        return hits
    def one_of(prefixes):
        for p in prefixes:
            hits = {k: v for k, v in res.items() if p in k}
            if hits:
                return hits
        raise AssertionError(prefixes)
    for name, v in one("k_encode_fused").items():
        # (the instances for a paged source -- last template argument true: scalar slot loads, k_fused.h fused_tok_off --
        # hold a few more scalars across phase A)
        paged_src = "true>" in name or "ELb1EE" in name  # (demangled / mangled)
        assert v["VGPRs"] <= 64 and v["Occupancy"] == 8 and v["ScratchSize"] <= (40 if paged_src else 32), (name, v)
    for name, v in one_of(("k_cdf_encodeILb1ELb1ELi8ELb1", "k_cdf_encode<true, true, 8, true>")).items():
        assert v["VGPRs"] <= 64 and v["Occupancy"] == 8 and v["ScratchSize"] <= 96, (name, v)
    for name, v in one("k_decode").items():
        assert v["ScratchSize"] == 0 and v["Occupancy"] == 8, (name, v)
        # the decoder parks its ring's in-flight block in AGPR a0 behind the compiler's back: sound only at exactly 60
        # VGPRs (a0 = physical register 60 of 64) and no AGPR of the compiler's own (k_decode.h, native._check_decoder_registers)
        assert v["VGPRs"] == 60 and v["AGPRs"] == 0, (name, v)
    for name, v in one("k_quantize").items():
        assert v["ScratchSize"] == 0, (name, v)
    for name, v in one("k_encode_fused").items():
        assert v["LDS Size"] * 4 <= 160 * 1024, (name, v)   # four workgroups per CU


def test_prefix_hash_chain_kept_between_calls_gives_the_same_digests():
    """Round 5: _prefix_hashes_of keeps the last call's token bytes and digests and resumes the SHA-256 chain at the first
    chunk that differs (lookup / retrieve / store of one prompt, the next turn of a conversation).  Whatever the overlap
    with the previous call -- none, a prefix of whole chunks, a change inside a chunk, a shorter or longer prompt, another
    chunk size in between -- the digests are those of the reference's chain (cache_engine.py:58-96)."""
    import hashlib
    import random
    from lmcache_amd.cache_engine import LMCacheEngine
    e = LMCacheEngine.__new__(LMCacheEngine)

    def plain(toks, cs):
        r, out = "", []
        for s0 in range(0, len(toks), cs):
            r = hashlib.sha256(r.encode("ascii") + toks[s0:s0 + cs].numpy().tobytes()).hexdigest()
            out.append(r)
        return out
    rnd = random.Random(3)
    base = torch.randint(0, 32000, (3000,))
    for trial in range(150):
        e.chunk_size = rnd.choice([256, 256, 256, 100])
        n = rnd.randint(1, 4000)
        t = torch.randint(0, 32000, (n,))
        k = rnd.randint(0, min(n, len(base)))
        t[:k] = base[:k]
        if rnd.random() < 0.3:
            t[rnd.randrange(n)] += 1
        assert e._prefix_hashes_of(t) == plain(t, e.chunk_size), trial
        skip = rnd.randint(0, 3)
        assert e._prefix_hashes_of(t, skip) == plain(t, e.chunk_size)[skip:], trial
        if rnd.random() < 0.5:
            base = t.clone()


@pytest.mark.parametrize("T", [2, 3, 5, 17, 32, 100, 128, 200, 255, 256])
def test_pack_cap_covers_the_allocations_of_high_entropy_channels(oracle, T):
    """pack_cap (the pinned region a pack is written into) is a Python formula, the streams' allocations are
    the format's (lmc_counts_alloc_bytes over lmc_counts_lane_words of the lane's S).  Held against each other where the
    allocation is largest -- channels that use every symbol the plane has, as evenly as the chunk length allows (the
    bound is a sum of concave per-symbol terms) -- for every bins value a CacheGen table or a fuzz case can name."""
    from lmcache_amd.storage_backend.serde.cachegen_device import pack_cap
    H, D, L = 1, 128, 1
    C = H * D
    rng = np.random.default_rng(T)
    for bins in (4, 5, 6, 8, 9, 16, 17, 22, 32, 33):
        R = bins - 1
        m = bins // 2 - 1
        # symbols of token t in channel c: a permutation-shifted round robin over all R symbols (every count within 1 of
        # T / R), realised as inputs x = (sym - m) / m * max with a row maximum of exactly 1 on a channel of its own
        sym = (np.arange(T)[:, None] + rng.integers(0, R, size=C)[None, :]) % R
        x = (sym.astype(np.float32) - m) / m
        x[:, 0] = 1.0   # the row maximum (symbol 2m)
        kv = np.stack([x, x])[None].astype(np.float32)        # [L, 2, T, C]
        import torch
        bits, code = oracle.torch_to_bits(torch.from_numpy(kv).to(torch.bfloat16))
        blob = oracle.encode_blob(bits, code, H, D, np.array([bins, bins], np.int32))
        pack = oracle.pack_from_blobs([blob], T)
        assert len(pack) <= pack_cap(1, L, T, H, D, [bins, bins]), (T, bins, len(pack), pack_cap(1, L, T, H, D, [bins, bins]))
