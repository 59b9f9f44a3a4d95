"""The drop-in claim of INTEGRATION.md section 4, made executable on a box without /root/reference: the reference's
plugin surface for the hot path was dumped as data by oracle/gen_golden.py (gen_api -> tests/golden/reference_api.json:
inspect.signature of LMCacheEngine, LMCacheEngineBuilder, LMCBackendInterface and its backends, Serializer /
Deserializer and the CacheGen pair, RemoteConnector, the three Create* factories, the config dataclasses and their
constructors), and every entry is held against the lmcache_amd mirror of the same dotted name: the reference's
parameters must be there under the same names, in the same order, of the same kind, with the same defaults.  What the
mirror ADDS must be optional (a default, or keyword-only) -- a caller written against the reference never sees it --
and is listed in ADDITIONS below, which INTEGRATION.md section 5 quotes."""
import dataclasses
import importlib
import inspect
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
API = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_api.json")))

# every parameter / method the mirrors offer beyond the reference's surface; anything else that turns up fails the test
ADDITIONS = {
    "lmcache.config.LMCacheEngineConfig": {"fields": ["local_serde"]},
    "lmcache.config.LMCacheEngineConfig.from_defaults": ["local_serde"],
    "lmcache.config.LMCacheEngineConfig.from_legacy": ["local_serde"],
    # the local tier can hold ENCODED chunks (local_serde: cachegen), for which it needs the model's bins and format
    "lmcache.storage_backend.local_backend.LMCLocalBackend.__init__": ["metadata"],
    "lmcache.storage_backend.local_backend.LMCLocalDiskBackend.__init__": ["metadata"],
}

# public-looking names of the reference that are private machinery of ITS implementation, with what stands in their
# place here; the contract a caller programs against is the interface's methods, which are all checked
NOT_MIRRORED = {
    "lmcache.storage_backend.local_backend.LMCLocalBackend.put_blocking": "put(blocking=True) is the entry point; no separate method",
    "lmcache.storage_backend.local_backend.LMCLocalBackend.put_nonblocking": "put(blocking=False) queues for put_worker",
    "lmcache.storage_backend.remote_backend.LMCPipelinedRemoteBackend.network_worker": "one fetch thread (_fetch_worker)",
    "lmcache.storage_backend.remote_backend.LMCPipelinedRemoteBackend.deserialize_worker": "stream-ordered decode, no thread",
    "lmcache.storage_backend.serde.cachegen_decoder.CacheGenDeserializer.get_output_buffer": "torchac's staging buffer: the "
        "HIP decoder writes straight into the destination",
}


def mirror(dotted):
    mod, name = dotted.rsplit(".", 1)
    return getattr(importlib.import_module(mod.replace("lmcache", "lmcache_amd", 1)), name)


def params_of(fn):
    return [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
            for p in inspect.signature(fn).parameters.values()]


def check_params(where, want, got, extra_ok):
    """`want` is a prefix-compatible subset of `got`: same names, order, kinds and defaults; every extra is optional."""
    names = [p[0] for p in got]
    for i, (name, kind, default) in enumerate(want):
        assert name in names, f"{where}: parameter {name!r} of the reference is missing"
        g = got[names.index(name)]
        assert g[1] == kind, f"{where}: {name} is {g[1]}, the reference's is {kind}"
        assert g[2] == default, f"{where}: default of {name} is {g[2]}, the reference's is {default}"
    ref_names = [p[0] for p in want]
    assert [n for n in names if n in ref_names] == ref_names, f"{where}: parameter order differs: {names} vs {ref_names}"
    extras = [p for p in got if p[0] not in ref_names]
    for name, kind, default in extras:
        assert default is not None or kind in ("KEYWORD_ONLY", "VAR_KEYWORD", "VAR_POSITIONAL"), \
            f"{where}: extra parameter {name} is not optional"
        assert name in extra_ok or kind in ("VAR_KEYWORD", "VAR_POSITIONAL"), f"{where}: unlisted extra parameter {name}"
    # an extra positional parameter must not sit in front of a reference parameter
    last_ref = max((names.index(n) for n in ref_names), default=-1)
    for name, kind, _ in extras:
        assert kind != "POSITIONAL_OR_KEYWORD" or names.index(name) > last_ref, f"{where}: {name} shifts the positional order"


@pytest.mark.parametrize("dotted", sorted(API["functions"]))
def test_factories_match_the_reference(dotted):
    check_params(dotted, API["functions"][dotted], params_of(mirror(dotted)), ADDITIONS.get(dotted, []))


@pytest.mark.parametrize("dotted", sorted(API["classes"]))
def test_classes_offer_the_reference_methods(dotted):
    cls = mirror(dotted)
    ref = API["classes"][dotted]
    for base in ref["bases"]:  # the inheritance a caller may rely on (isinstance against the interface)
        assert base in [b.__name__ for b in cls.__mro__], f"{dotted}: does not derive from {base}"
    for mname, m in ref["methods"].items():
        if f"{dotted}.{mname}" in NOT_MIRRORED:
            continue
        assert hasattr(cls, mname), f"{dotted}.{mname} is missing"
        raw = inspect.getattr_static(cls, mname)
        kind = "static" if isinstance(raw, staticmethod) else "class" if isinstance(raw, classmethod) else "method"
        assert kind == m["kind"], f"{dotted}.{mname} is a {kind} method, the reference's is {m['kind']}"
        check_params(f"{dotted}.{mname}", m["params"], params_of(getattr(cls, mname)), ADDITIONS.get(f"{dotted}.{mname}", []))


@pytest.mark.parametrize("dotted", sorted(API["dataclasses"]))
def test_dataclasses_carry_the_reference_fields(dotted):
    cls = mirror(dotted)
    ref = API["dataclasses"][dotted]
    assert dataclasses.is_dataclass(cls)
    fields = [[f.name, None if f.default is dataclasses.MISSING else repr(f.default)] for f in dataclasses.fields(cls)]
    names = [f[0] for f in fields]
    ref_names = [f[0] for f in ref["fields"]]
    assert names[:len(ref_names)] == ref_names, f"{dotted}: fields {names} do not begin with the reference's {ref_names}"
    for (n, d), (gn, gd) in zip(ref["fields"], fields):
        assert d == gd, f"{dotted}.{n}: default {gd}, the reference's is {d}"
    extra = fields[len(ref_names):]
    assert [f[0] for f in extra] == ADDITIONS.get(dotted, {}).get("fields", []), f"{dotted}: unlisted extra fields {extra}"
    assert all(d is not None for _, d in extra), f"{dotted}: an extra field without a default breaks positional construction"
    for cname, params in ref["constructors"].items():
        check_params(f"{dotted}.{cname}", params, params_of(getattr(cls, cname)), ADDITIONS.get(f"{dotted}.{cname}", []))


def test_positional_construction_as_the_reference_tests_do_it():
    """tests/test_cache_engine.py:12-13 builds the metadata positionally; the config through from_legacy / from_defaults."""
    from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
    md = LMCacheEngineMetadata("test_model", 3, 123, "vllm", "half")
    assert (md.model_name, md.world_size, md.worker_id, md.fmt, md.dtype) == ("test_model", 3, 123, "vllm", "half")
    cfg = LMCacheEngineConfig.from_legacy(chunk_size=256, backend="cuda")
    assert cfg.chunk_size == 256 and cfg.local_device == "cuda" and cfg.remote_url is None
    cfg = LMCacheEngineConfig.from_defaults(chunk_size=128)
    assert cfg.chunk_size == 128
