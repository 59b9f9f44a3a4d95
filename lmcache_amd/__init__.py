"""lmcache_amd -- MI355X-native KV hot path behind LMCache's store()/retrieve().

Scope: per-layer KV gather -> CacheGen quantise + entropy
encode -> pinned host-DRAM offload, and the inverse decode + scatter, as
hand-written HIP for gfx950 behind the reference's LMCacheEngine /
LMCBackendInterface / Serializer interfaces.  Token-chunk hashing and prefix
matching stay in Python.
"""
__version__ = "0.1.0"
