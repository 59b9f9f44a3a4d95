"""INTEGRATION.md section 4 executed: a deployment swaps the package, not its call sites.  Here every module of
lmcache_amd is registered in sys.modules under the reference's name (`lmcache.*`), and the flows of the reference's own
serde and engine tests are run through THOSE imports -- the statements a user of the reference has in their code:
`from lmcache.cache_engine import LMCacheEngine`, `LMCacheEngineConfig.from_legacy(chunk_size=..., backend=...)`,
`CacheGenSerializer(config, metadata).to_bytes(kv)`, `engine.store(tokens, kv)` / `engine.retrieve(tokens)`.

What is restated, with the reference's parameters (the reference's files are not on the GPU box, and its sources are
not copied: the flows are re-written here, the assertions are theirs):
  tests/test_serde.py:30-107         encoder (vllm vs permuted huggingface sizes, CacheGenEncoderOutput.from_bytes),
                                     decoder (shape, non-zero mean), a chunk shorter than chunk_size
  tests/test_cache_engine.py:88-297  retrieve on the destination device, store -> retrieve equality, prefix retrieve,
                                     mixed retrieve, skip_existing, the builder
for the backends in scope ("cuda", "cpu", a disk path; redis / lm:// stay with the reference)."""
import importlib
import pkgutil
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lmcache():
    """Alias lmcache -> lmcache_amd for the duration of the module (and take the aliases away again)."""
    import lmcache_amd
    added = []
    mods = [lmcache_amd] + [importlib.import_module(m.name)
                            for m in pkgutil.walk_packages(lmcache_amd.__path__, "lmcache_amd.")]
    for m in mods:
        alias = "lmcache" + m.__name__[len("lmcache_amd"):]
        if alias not in sys.modules:
            sys.modules[alias] = m
            added.append(alias)
    yield sys.modules["lmcache"]
    for a in added:
        sys.modules.pop(a, None)


@pytest.fixture
def autorelease():
    objs = []
    yield lambda o: (objs.append(o), o)[1]
    for o in objs:
        o.close()


LAYERS, HEADS, HEAD_SIZE = 32, 8, 128


def make_kv(ntok, fmt, device):
    shape = (ntok, HEADS, HEAD_SIZE) if fmt == "vllm" else (HEADS, ntok, HEAD_SIZE)
    dtype = torch.bfloat16 if fmt == "vllm" else torch.float16
    return tuple((torch.rand(shape, dtype=dtype, device=device), torch.rand(shape, dtype=dtype, device=device))
                 for _ in range(LAYERS))


def as_blob(kv):
    return torch.stack([torch.stack(pair, dim=0) for pair in kv], dim=0)


def make_tokens(n, device):
    return torch.randint(0, 10000, size=[n]).to(device)


def cat_kv(parts, fmt):
    dim = 1 if fmt == "huggingface" else 0
    return tuple((torch.cat([p[l][0] for p in parts], dim=dim), torch.cat([p[l][1] for p in parts], dim=dim))
                 for l in range(LAYERS))


def assert_same_prefix(got, want, ntok, fmt):
    dim = 0 if fmt == "vllm" else 1
    assert len(got) == len(want)
    for (gk, gv), (wk, wv) in zip(got, want):
        for g, w in ((gk, wk), (gv, wv)):
            assert g.dim() == 3 and w.dim() == 3 and g.shape[dim] >= ntok and w.shape[dim] >= ntok
            assert torch.equal(g.narrow(dim, 0, ntok), w.to(g.device).narrow(dim, 0, ntok))


def serde_pair(chunk_size, fmt):
    from lmcache.config import LMCacheEngineConfig, LMCacheEngineMetadata
    from lmcache.storage_backend.serde.cachegen_decoder import CacheGenDeserializer
    from lmcache.storage_backend.serde.cachegen_encoder import CacheGenSerializer
    config = LMCacheEngineConfig.from_defaults(chunk_size=chunk_size)
    metadata = LMCacheEngineMetadata(model_name="mistralai/Mistral-7B-Instruct-v0.2", world_size=1, worker_id=0, fmt=fmt,
                                     dtype="bfloat16")
    return CacheGenSerializer(config, metadata), CacheGenDeserializer(config, metadata)


@pytest.mark.parametrize("chunk_size", [16, 128, 256])
def test_cachegen_encoder(lmcache, chunk_size):
    from lmcache.storage_backend.serde.cachegen_basics import CacheGenEncoderOutput
    ser_v, _ = serde_pair(chunk_size, "vllm")
    ser_h, _ = serde_pair(chunk_size, "huggingface")
    kv = as_blob(make_kv(chunk_size, "vllm", "cuda"))
    out_v = ser_v.to_bytes(kv)
    out_h = ser_h.to_bytes(kv.permute([0, 1, 3, 2, 4]))
    assert abs(len(out_v) - len(out_h)) < 10
    parsed = CacheGenEncoderOutput.from_bytes(out_v)
    assert parsed.num_heads == 8 and parsed.head_size == 128


@pytest.mark.parametrize("fmt", ["vllm", "huggingface"])
@pytest.mark.parametrize("chunk_size", [16, 128, 256])
def test_cachegen_decoder(lmcache, fmt, chunk_size):
    ser, des = serde_pair(chunk_size, fmt)
    kv = as_blob(make_kv(chunk_size, fmt, "cuda"))
    back = des.from_bytes(ser.to_bytes(kv))
    assert back.shape == kv.shape and back.mean() != 0


def test_cachegen_unmatched_size(lmcache):
    ser, des = serde_pair(256, "vllm")
    kv = as_blob(make_kv(256 - 20, "vllm", "cuda"))
    back = des.from_bytes(ser.to_bytes(kv))
    assert back.shape == kv.shape and back.mean() != 0


def engine_of(backend, fmt, autorelease, **kw):
    from lmcache.cache_engine import LMCacheEngine
    from lmcache.config import LMCacheEngineConfig, LMCacheEngineMetadata
    cfg = LMCacheEngineConfig.from_legacy(backend=backend, **kw)
    return autorelease(LMCacheEngine(cfg, LMCacheEngineMetadata("test_model", 3, 123, fmt, "half")))


@pytest.mark.parametrize("src_device", ["cuda:0", "cuda", "cpu"])
@pytest.mark.parametrize("backend", ["cuda", "cpu"])
def test_retrieve_device(lmcache, backend, src_device, autorelease):
    tokens, kv = make_tokens(500, src_device), make_kv(500, "vllm", src_device)
    engine = engine_of(backend, "vllm", autorelease, chunk_size=256)
    engine.store(tokens, kv)
    got, _ = engine.retrieve(tokens)
    assert all(k.device == torch.device("cuda:0") and v.device == torch.device("cuda:0") for k, v in got)


@pytest.mark.parametrize("fmt", ["vllm", "huggingface"])
@pytest.mark.parametrize("backend", ["cuda", "cpu"])
def test_same_retrieve_store(lmcache, fmt, backend, autorelease):
    device = "cpu" if backend == "cpu" else "cuda"
    tokens, kv = make_tokens(2000, device), make_kv(2000, fmt, device)
    engine = engine_of(backend, fmt, autorelease, chunk_size=256, remote_serde="torch")
    got, mask = engine.retrieve(tokens)
    assert len(got) == 0 and torch.sum(mask) == 0
    engine.store(tokens, kv)
    got, mask = engine.retrieve(tokens)
    assert torch.sum(mask) == 2000
    assert_same_prefix(got, kv, 2000, fmt)


@pytest.mark.parametrize("fmt", ["vllm", "huggingface"])
@pytest.mark.parametrize("chunk_size", [128, 256])
@pytest.mark.parametrize("backend", ["cuda", "cpu"])
def test_retrieve_prefix(lmcache, fmt, chunk_size, backend, autorelease):
    device = "cpu" if backend == "cpu" else "cuda"
    tokens, kv = make_tokens(2000, device), make_kv(2000, fmt, device)
    more = make_tokens(1000, device)
    engine = engine_of(backend, fmt, autorelease, chunk_size=chunk_size)
    engine.store(tokens, kv)
    got, mask = engine.retrieve(torch.cat([tokens, more]))
    want = 2000 // chunk_size * chunk_size
    assert torch.sum(mask) == want
    assert_same_prefix(got, kv, want, fmt)


@pytest.mark.parametrize("fmt", ["vllm", "huggingface"])
@pytest.mark.parametrize("chunk_size", [128, 256])
def test_mixed_retrieve(lmcache, fmt, chunk_size, autorelease):
    device = "cuda"
    tokens, kv = make_tokens(2000, device), make_kv(2000, fmt, device)
    new_tokens, new_kv = make_tokens(1000, device), make_kv(1000, fmt, device)
    engine = engine_of("cuda", fmt, autorelease, chunk_size=chunk_size)
    engine.store(tokens, kv)
    engine.store(new_tokens, new_kv)
    got, mask = engine.retrieve(torch.cat([tokens, new_tokens]))
    want = 2000 // chunk_size * chunk_size
    assert torch.sum(mask) == want
    assert_same_prefix(got, kv, want, fmt)
    got, mask = engine.retrieve(new_tokens)
    assert torch.sum(mask) == 1000
    assert_same_prefix(got, new_kv, 1000, fmt)
    final_tokens = torch.cat([tokens, new_tokens])
    final_kv = cat_kv([kv, make_kv(1000, fmt, device)], fmt)
    engine.store(final_tokens, final_kv)
    got, mask = engine.retrieve(final_tokens)
    assert torch.sum(mask) == 3000
    assert_same_prefix(got, final_kv, 3000, fmt)


@pytest.mark.parametrize("fmt", ["vllm", "huggingface"])
def test_skipping(lmcache, fmt, autorelease):
    device = "cuda"
    tokens, kv = make_tokens(12000, device), make_kv(12000, fmt, device)
    new_tokens, new_kv = make_tokens(200, device), make_kv(200, fmt, device)
    final_tokens, final_kv = torch.cat([tokens, new_tokens]), cat_kv([kv, new_kv], fmt)
    e1 = engine_of("cuda", fmt, autorelease, chunk_size=256)
    e2 = engine_of("cuda", fmt, autorelease, chunk_size=256)
    e1.store(tokens, kv)
    e2.store(tokens, kv)
    e1.store(final_tokens, final_kv, skip_existing=True)
    e2.store(final_tokens, final_kv, skip_existing=False)
    for e in (e1, e2):
        got, mask = e.retrieve(final_tokens)
        assert torch.sum(mask) == 12200
        assert_same_prefix(got, final_kv, 12200, fmt)


def test_builder(lmcache, autorelease):
    from lmcache.cache_engine import LMCacheEngineBuilder
    from lmcache.config import LMCacheEngineConfig, LMCacheEngineMetadata
    md = LMCacheEngineMetadata("test_model", 3, 123, "vllm", "half")
    cfg = LMCacheEngineConfig.from_legacy(chunk_size=256)
    cfg2 = LMCacheEngineConfig.from_legacy(chunk_size=512)
    assert LMCacheEngineBuilder.get("dropin-test") is None
    try:
        LMCacheEngineBuilder.get_or_create("dropin-test", cfg, md)
        assert LMCacheEngineBuilder.get("dropin-test") is not None
        with pytest.raises(ValueError):
            LMCacheEngineBuilder.get_or_create("dropin-test", cfg2, md)
    finally:
        LMCacheEngineBuilder.destroy("dropin-test")
