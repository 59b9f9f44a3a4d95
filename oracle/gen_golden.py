"""Generate tests/golden/*.npz|json by running the REFERENCE ITSELF on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/lmc_oracle.c header).

Runs only in the build container, where /root/reference exists; the fixtures
it writes are committed so that the GPU box (no /root/reference) can check
against them.  Usage:  python oracle/gen_golden.py

The reference imports three packages that are absent here (nvtx, redis,
torchac_cuda -- SURVEY.md section 8c); they are stubbed with empty modules.
``Tensor.cuda`` is neutralised so that the in-tree CDF spec
(CacheGenEncoderImpl.compute_cdf, cachegen_encoder.py:175-222) runs on CPU.
Nothing from the reference is copied: it is imported and executed.
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_stubs():
    nvtx = types.ModuleType("nvtx")

    def annotate(*a, **k):
        return lambda f: f

    nvtx.annotate = annotate
    sys.modules["nvtx"] = nvtx
    sys.modules["redis"] = types.ModuleType("redis")
    sys.modules["torchac_cuda"] = types.ModuleType("torchac_cuda")
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def bits(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def make_kv(kind, shape, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "rand":  # the reference's own test distribution (tests/test_serde.py:20-21)
        x = torch.rand(shape, generator=g)
    elif kind == "randn":
        x = torch.randn(shape, generator=g)
    elif kind == "outlier":  # randn * per-channel log-normal scale
        x = torch.randn(shape, generator=g) * torch.exp(1.5 * torch.randn(shape[-1], generator=g))
    else:
        raise ValueError(kind)
    return x.to(dtype)


def gen_hash():
    from lmcache.cache_engine import LMCacheEngine
    from lmcache.utils import CacheEngineKey

    class _E(LMCacheEngine):  # bypass __init__ (no backend needed for hashing)
        def __init__(self, chunk_size):
            self.chunk_size = chunk_size

    cases = []
    g = torch.Generator().manual_seed(7)
    specs = [("arange512", torch.arange(512, dtype=torch.int64), 256),
             ("rand600", torch.randint(0, 32000, (600,), generator=g, dtype=torch.int64), 256),
             ("rand33_c16", torch.randint(0, 10000, (33,), generator=g, dtype=torch.int64), 16),
             ("rand1", torch.randint(0, 10000, (1,), generator=g, dtype=torch.int64), 256),
             ("rand256", torch.randint(0, 10000, (256,), generator=g, dtype=torch.int64), 256)]
    for name, toks, cs in specs:
        e = _E(cs)
        hashes = e._prefix_hash(e._chunk_tokens(toks))
        cases.append({"name": name, "tokens": toks.tolist(), "chunk_size": cs, "hashes": hashes})
    key = CacheEngineKey("vllm", "meta-llama/Llama-3.1-8B-Instruct", 8, 3, cases[0]["hashes"][0])
    out = {"cases": cases, "key_string": key.to_string(),
           "key_fields": ["vllm", "meta-llama/Llama-3.1-8B-Instruct", 8, 3, cases[0]["hashes"][0]]}
    with open(os.path.join(OUT, "hash_chain.json"), "w") as f:
        json.dump(out, f)
    print("hash_chain.json", [c["hashes"][0][:8] for c in cases])


def gen_quant():
    from lmcache.storage_backend.serde.cachegen_decoder import do_dequantize
    from lmcache.storage_backend.serde.cachegen_encoder import torch_quant_vectorized

    L, T, C = 4, 24, 256
    kbins = torch.tensor([32., 32., 16., 16.])
    vbins = torch.tensor([32., 16., 16., 16.])
    store = {}
    idx = 0
    for dtype, dname in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        for kind in ("rand", "randn", "outlier"):
            idx += 1
            k = make_kv(kind, (L, T, C), dtype, 100 + idx)
            v = make_kv(kind, (L, T, C), dtype, 200 + idx)
            qk, mk = torch_quant_vectorized(kbins, k)   # cachegen_encoder.py:282
            qv, mv = torch_quant_vectorized(vbins, v)   # cachegen_encoder.py:283
            assert qk.dtype == torch.int8 and mk.dtype == dtype
            # decoder: uint8 buffer -> .float() -> do_dequantize -> cast (cachegen_decoder.py:95-104,177-200)
            dk = do_dequantize(qk.to(torch.uint8).float(), kbins, mk)
            dv = do_dequantize(qv.to(torch.uint8).float(), vbins, mv)
            blob = torch.stack([dk, dv])  # [2, L, T, C]
            tag = f"{dname}_{kind}"
            store[f"{tag}_kv"] = bits(torch.stack([k, v], dim=1))  # [L,2,T,C] vllm chunk layout
            store[f"{tag}_sym"] = torch.cat([qk, qv]).numpy()          # [2L,T,C] (encode_input order)
            store[f"{tag}_scale"] = bits(torch.cat([mk, mv]).squeeze(-1))  # [2L,T]
            store[f"{tag}_deq_bf16"] = bits(blob.permute(1, 0, 2, 3).to(torch.bfloat16))  # [L,2,T,C]
            store[f"{tag}_deq_fp16"] = bits(blob.permute(1, 0, 2, 3).to(torch.float16))
    store["bins"] = torch.cat([kbins, vbins]).to(torch.int32).numpy()
    np.savez_compressed(os.path.join(OUT, "quant.npz"), **store)
    print("quant.npz", len(store), "arrays")

    # edge rows: all-zero, inf, nan, denormal max, single spike, negative-only (own seeded generator: reproducible)
    ge = torch.Generator().manual_seed(4242)
    x = torch.randn(2, 8, 64, generator=ge).to(torch.bfloat16)
    x[0, 1, :] = 0
    x[1, 2, 0] = float("inf")
    x[1, 3, 5] = float("nan")
    x[0, 4, :] = 1e-40
    x[0, 5, :] = 0
    x[0, 5, 7] = -3.0
    x[1, 6, :] = -torch.rand(64, generator=ge).to(torch.bfloat16)
    eb = torch.tensor([32., 16.])
    q, m = torch_quant_vectorized(eb, x)
    d = do_dequantize(q.to(torch.uint8).float(), eb, m)
    np.savez_compressed(os.path.join(OUT, "quant_edge.npz"),
                        x=bits(x), sym=q.numpy(), scale=bits(m.squeeze(-1)),
                        deq_bf16=bits(d.to(torch.bfloat16)), bins=eb.to(torch.int32).numpy())
    print("quant_edge.npz rows", q[0, 1, :4].tolist(), q[1, 2, :4].tolist(), q[1, 3, :4].tolist())


def gen_bins():
    """The per-layer bins tables of every model the reference knows: CacheGenConfig.from_model_name
    (cachegen_basics.py:32-78) expanded the way CacheGenSerializer.make_key_bins / make_value_bins do
    (cachegen_encoder.py:339-350; the tensors are built on the CPU here, their .cuda() is all that is skipped)."""
    from lmcache.storage_backend.serde.cachegen_basics import CacheGenConfig
    out = {}
    for name in ("mistralai/Mistral-7B-Instruct-v0.2", "lmsys/longchat-7b-16k", "Qwen/Qwen-7B",
                 "meta-llama/Llama-3.1-8B-Instruct", "THUDM/glm-4-9b-chat"):
        cfg = CacheGenConfig.from_model_name(name)
        kb = torch.zeros(cfg["key_third_layers"])
        kb[:cfg["key_first_layers"]] = cfg["key_first_bins"]
        kb[cfg["key_first_layers"]:cfg["key_second_layers"]] = cfg["key_second_bins"]
        kb[cfg["key_second_layers"]:cfg["key_third_layers"]] = cfg["key_third_bins"]
        vb = torch.zeros(cfg["key_third_layers"])
        vb[:cfg["value_first_layers"]] = cfg["value_first_bins"]
        vb[cfg["value_first_layers"]:] = cfg["value_second_bins"]
        out[name] = {"key_bins": [int(b) for b in kb], "value_bins": [int(b) for b in vb]}
    with open(os.path.join(OUT, "bins.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("bins.json", {k: len(v["key_bins"]) for k, v in out.items()})


def gen_cdf():
    from lmcache.storage_backend.serde.cachegen_encoder import CacheGenEncoderImpl

    store = {}
    for T in (1, 7, 16, 128, 236, 250, 256, 768):
        g = torch.Generator().manual_seed(1000 + T)
        C, P = 96, 2
        # plane 0: 31 symbols (32-bin layer), plane 1: 15 symbols (16-bin layer); skewed so bins differ
        s0 = torch.clamp((torch.randn(T, C, generator=g) * 5 + 15).round(), 0, 30).to(torch.int8)
        s1 = torch.clamp((torch.randn(T, C, generator=g) * 2 + 7).round(), 0, 14).to(torch.int8)
        sym = torch.stack([s0, s1])  # [P,T,C]
        enc = CacheGenEncoderImpl(fp_k=[torch.zeros(T, C)] * P, fp_v=[], config=None)
        enc.quantized_key = {i: sym[i] for i in range(P)}
        cdf_float = enc.compute_cdf(is_key=True)  # [P, C, 33] float in [0,1]  (:175-222)
        from lmcache.storage_backend.serde.cachegen_encoder import _convert_to_int_and_normalize
        cdf_int = _convert_to_int_and_normalize(cdf_float, True)  # (:95-126)
        store[f"T{T}_sym"] = sym.numpy()
        store[f"T{T}_cdf"] = cdf_int.numpy().view(np.uint16)
    np.savez_compressed(os.path.join(OUT, "cdf.npz"), **store)
    print("cdf.npz", len(store), "arrays")


def gen_layout():
    """Decoder tail layout/dtype rule (cachegen_decoder.py:182-200) and encoder HF permute (:377-378)."""
    L, T, H, D = 2, 5, 3, 8
    g = torch.Generator().manual_seed(5)
    key = torch.randn(L, T, H * D, generator=g)
    value = torch.randn(L, T, H * D, generator=g)
    blob = torch.stack([key, value]).reshape(2, L, T, H, D)
    vllm = blob.permute(1, 0, 2, 3, 4).to(torch.bfloat16)
    hf = blob.permute(1, 0, 3, 2, 4).to(torch.float16)
    np.savez_compressed(os.path.join(OUT, "layout.npz"), key=key.numpy(), value=value.numpy(),
                        vllm=bits(vllm), vllm_shape=np.array(vllm.shape),
                        hf=bits(hf), hf_shape=np.array(hf.shape))
    print("layout.npz", tuple(vllm.shape), tuple(hf.shape))


def gen_engine():
    """store/retrieve semantics with the lossless local cpu backend (tests/test_cache_engine.py)."""
    from lmcache.cache_engine import LMCacheEngine
    from lmcache.config import LMCacheEngineConfig, LMCacheEngineMetadata

    out = []
    for fmt in ("vllm", "huggingface"):
        cfg = LMCacheEngineConfig.from_legacy(chunk_size=16, backend="cpu")
        meta = LMCacheEngineMetadata("test_model", 3, 123, fmt, "half")
        eng = LMCacheEngine(cfg, meta)
        eng.engine_.dst_device = "cpu"  # hard-coded "cuda" in local_backend.py:53
        g = torch.Generator().manual_seed(11)
        ntok = 50
        toks = torch.randint(0, 10000, (ntok,), generator=g)
        shape = [ntok, 2, 4] if fmt == "vllm" else [2, ntok, 4]
        kv = tuple((torch.rand(shape, generator=g).to(torch.bfloat16),
                    torch.rand(shape, generator=g).to(torch.bfloat16)) for _ in range(3))
        eng.store(toks, kv)
        rec = {"fmt": fmt, "chunk_size": 16, "tokens": toks.tolist(), "queries": []}
        # full, prefix, extended, mismatched-from-20, with suffix masks
        other = torch.randint(0, 10000, (30,), generator=g)
        queries = {
            "full": (toks, None),
            "prefix40": (toks[:40], None),
            "extended": (torch.cat([toks, other]), None),
            "diverge20": (torch.cat([toks[:20], other]), None),
            "miss": (other, None),
            "mask_skip16": (toks, 16),
            "mask_skip20": (toks, 20),
            "mask_skip48": (toks, 48),
        }
        for name, (q, skip) in queries.items():
            mask = None
            if skip is not None:
                mask = torch.ones(len(q), dtype=torch.bool)
                mask[:skip] = False
            ret, ret_mask = eng.retrieve(q, mask)
            tdim = 0 if fmt == "vllm" else 1
            rec["queries"].append({
                "name": name, "tokens": q.tolist(), "skip": skip,
                "ret_mask": ret_mask.to(torch.int8).tolist(),
                "ret_tokens": 0 if len(ret) == 0 else int(ret[0][0].shape[tdim]),
                "k0_sum": 0.0 if len(ret) == 0 else float(ret[0][0].float().sum()),
            })
        rec["kv_seed"] = 11
        out.append(rec)
        eng.close()
    with open(os.path.join(OUT, "engine_semantics.json"), "w") as f:
        json.dump(out, f)
    print("engine_semantics.json", [(q["name"], q["ret_tokens"]) for q in out[0]["queries"]])


def _sig(fn):
    """[name, kind, default repr or None] of every parameter of a callable."""
    import inspect
    out = []
    for p in inspect.signature(fn).parameters.values():
        out.append([p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)])
    return out


def gen_disk():
    """A chunk file written by the reference's LMCLocalDiskBackend (local_backend.py:163-310): the file name it gives
    the key and the safetensors payload.  lmcache_amd's disk tier in its raw mode reads this file and writes the same."""
    import shutil
    import tempfile
    from lmcache.config import LMCacheEngineConfig
    from lmcache.storage_backend.local_backend import LMCLocalDiskBackend
    from lmcache.utils import CacheEngineKey

    d = tempfile.mkdtemp() + "/"
    torch.cuda.Stream = lambda *a, **k: None  # the put worker makes one at start (local_backend.py:234); no GPU here
    be = LMCLocalDiskBackend(LMCacheEngineConfig.from_legacy(chunk_size=16, backend="file://" + d))
    g = torch.Generator().manual_seed(5)
    t = torch.rand([3, 2, 16, 2, 8], generator=g).to(torch.bfloat16)
    key = CacheEngineKey("vllm", "test_model", 1, 0, "ab" * 32)
    be.put(key, t, blocking=True)
    be.close()
    names = os.listdir(d)
    assert len(names) == 1
    os.makedirs(os.path.join(OUT, "disk"), exist_ok=True)
    shutil.copy(os.path.join(d, names[0]), os.path.join(OUT, "disk", names[0]))
    os.chmod(os.path.join(OUT, "disk", names[0]), 0o644)
    with open(os.path.join(OUT, "disk", "disk.json"), "w") as f:
        json.dump({"file": names[0], "key": key.to_string(), "shape": list(t.shape), "seed": 5,
                   "bits": bits(t).reshape(-1).tolist()}, f)
    shutil.rmtree(d)
    print("disk", names[0], os.path.getsize(os.path.join(OUT, "disk", names[0])), "bytes")


def gen_api():
    """The reference's plugin surface for the hot path as data: the signatures a drop-in must
    offer name for name, default for default.  tests/test_dropin_api.py holds lmcache_amd against it (on the GPU box,
    where /root/reference does not exist, the fixture is all there is)."""
    import dataclasses
    import importlib
    import inspect
    api = {"functions": {}, "classes": {}, "dataclasses": {}}
    for mod, name in (("lmcache.storage_backend", "CreateStorageBackend"),
                      ("lmcache.storage_backend.serde", "CreateSerde"),
                      ("lmcache.storage_backend.connector", "CreateConnector")):
        api["functions"][f"{mod}.{name}"] = _sig(getattr(importlib.import_module(mod), name))
    for mod, name in (("lmcache.cache_engine", "LMCacheEngine"), ("lmcache.cache_engine", "LMCacheEngineBuilder"),
                      ("lmcache.storage_backend.abstract_backend", "LMCBackendInterface"),
                      ("lmcache.storage_backend.local_backend", "LMCLocalBackend"),
                      ("lmcache.storage_backend.local_backend", "LMCLocalDiskBackend"),
                      ("lmcache.storage_backend.remote_backend", "LMCRemoteBackend"),
                      ("lmcache.storage_backend.remote_backend", "LMCPipelinedRemoteBackend"),
                      ("lmcache.storage_backend.hybrid_backend", "LMCHybridBackend"),
                      ("lmcache.storage_backend.serde.serde", "Serializer"),
                      ("lmcache.storage_backend.serde.serde", "Deserializer"),
                      ("lmcache.storage_backend.serde.cachegen_encoder", "CacheGenSerializer"),
                      ("lmcache.storage_backend.serde.cachegen_decoder", "CacheGenDeserializer"),
                      ("lmcache.storage_backend.serde.torch_serde", "TorchSerializer"),
                      ("lmcache.storage_backend.serde.torch_serde", "TorchDeserializer"),
                      ("lmcache.storage_backend.connector.base_connector", "RemoteConnector"),
                      ("lmcache.utils", "CacheEngineKey")):
        cls = getattr(importlib.import_module(mod), name)
        methods = {}
        for mname, member in inspect.getmembers(cls):
            if mname.startswith("_") and mname != "__init__":
                continue
            if inspect.isfunction(member) or inspect.ismethod(member):
                if mname == "__init__" and member is object.__init__:
                    continue
                raw = inspect.getattr_static(cls, mname)
                kind = "static" if isinstance(raw, staticmethod) else "class" if isinstance(raw, classmethod) else "method"
                methods[mname] = {"kind": kind, "params": _sig(member)}
        api["classes"][f"{mod}.{name}"] = {"methods": methods, "bases": [b.__name__ for b in cls.__mro__[1:-1]]}
    for mod, name in (("lmcache.config", "LMCacheEngineConfig"), ("lmcache.config", "LMCacheEngineMetadata"),
                      ("lmcache.utils", "CacheEngineKey")):
        cls = getattr(importlib.import_module(mod), name)
        if not dataclasses.is_dataclass(cls):
            continue
        api["dataclasses"][f"{mod}.{name}"] = {
            "fields": [[f.name, None if f.default is dataclasses.MISSING else repr(f.default)] for f in dataclasses.fields(cls)],
            "constructors": {n: _sig(getattr(cls, n)) for n in ("from_defaults", "from_legacy", "from_file") if hasattr(cls, n)}}
    with open(os.path.join(OUT, "reference_api.json"), "w") as f:
        json.dump(api, f, indent=1, sort_keys=True)
    print("reference_api.json", len(api["functions"]), "functions,", len(api["classes"]), "classes,",
          len(api["dataclasses"]), "dataclasses")


if __name__ == "__main__":
    _install_stubs()
    os.makedirs(OUT, exist_ok=True)
    gen_hash()
    gen_quant()
    gen_bins()
    gen_cdf()
    gen_layout()
    gen_engine()
    gen_disk()
    gen_api()
