"""Probe: warm-prefix retrieve from the HBM-resident encoded tier, layer by layer, on a side stream that is
restricted to a share of the CUs (hipExtStreamCreateWithCUMask) while the decode-step proxy streams its 16 GB on
the compute stream.  The decoder is VALU-bound and the proxy HBM-bound: do they overlap better when the decoder
cannot take wave slots on every CU?
    python tools/probes/cu_mask_overlap.py
"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from lmcache_amd.cache_engine import LMCacheEngine  # noqa: E402
from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata  # noqa: E402
from lmcache_amd.storage_backend.serde.cachegen_device import layer_ranges  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(nwords, pattern):
    """A stream whose kernels may only run on the CUs whose bit is set; `pattern(i)` -> 32-bit word i."""
    words = (ctypes.c_uint32 * nwords)(*[pattern(i) for i in range(nwords)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(nwords), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    nwords = (ncu + 31) // 32
    kv = bench.make_kv(dev, 0, "rand")
    meta = LMCacheEngineMetadata(bench.MODEL, 1, 0, "vllm", "bfloat16")
    proxy = bench.DecodeStepProxy(dev)
    proxy.time_steps(3)
    alone = bench.median(proxy.time_steps(10))
    cfg = LMCacheEngineConfig.from_legacy(chunk_size=bench.CHUNK, backend="cuda", local_serde="cachegen")
    engine = LMCacheEngine(cfg, meta)
    toks = torch.randint(0, 32000, (bench.CTX,), generator=torch.Generator().manual_seed(7))
    engine.store(toks, kv)
    sides = {"all CUs": torch.cuda.Stream(device=dev),
             "1/2 (low half of each word)": masked_stream(nwords, lambda i: 0x0000ffff)}
    print(f"{ncu} CUs; proxy step alone {alone:.3f} ms")
    for name, side in sides.items():
        for step in (2, 4, (2, 2, 4, 8, 16), (1, 1, 2, 4, 8, 16), (2, 6, 24), (4, 28), 32):
            ts, hs, ds = [], [], []
            for r in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with torch.cuda.stream(side):
                    res = engine.retrieve_layerwise(toks, layers_per_launch=step)
                t1 = time.perf_counter()
                for l0, l1 in layer_ranges(32, step):
                    res.wait_layer(l0)
                    for l in range(l0, l1):
                        torch.mv(proxy.w[l], proxy.x, out=proxy.y)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
                hs.append((t1 - t0) * 1e3)
                res.finish()
                del res
                # the decode alone on this stream
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with torch.cuda.stream(side):
                    res = engine.retrieve_layerwise(toks, layers_per_launch=step)
                res.finish()
                torch.cuda.synchronize()
                ds.append((time.perf_counter() - t0) * 1e3)
                del res
            m = bench.median(ts[1:])
            print(f"  {name:28s} {str(step):20s} layers/launch: {m:.3f} ms = {m / alone:.3f} x  "
                  f"(host part of retrieve_layerwise {bench.median(hs[1:]):.3f} ms, decode alone {bench.median(ds[1:]):.3f} ms)")
    engine.close()


if __name__ == "__main__":
    main()
