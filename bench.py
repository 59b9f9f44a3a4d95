#!/usr/bin/env python
"""bench.py -- KV encode(+offload) throughput of the MI355X hot path.

Metric (BASELINE.json): "KV encode+offload GB/s per GPU", GB = raw 16-bit KV
bytes consumed.  Workload at N=1: BASELINE config 2 -- Llama-3-8B, bf16,
16 384-token context (32 layers x 8 KV heads x 128, 2 GiB), chunk_size 256
(64 chunks), synthetic KV already resident in HBM as the per-layer (K, V)
tensors LMCacheEngine.store() receives.

One "step" = one pass of the encode path over the whole context through the C
ABI (lmc_encode_chunks): gather from the per-layer tensors, quantise, CDF,
entropy-encode, compact into 64 blobs in an HBM arena.  `value` is that rate
(inputs and outputs in HBM; PCIe never inside `value`).  The pinned-host
offload leg (blobs D2H on a side stream, pipelined against the next step's
kernels) and the decode leg are measured separately and reported in the
`offload` / `decode` objects of the same JSON line.

N>1 (torch.distributed, one rank per GPU): the path shards by chunk with no
data-path collective, so every rank encodes its own
16k-token context (weak scaling); the barrier and the max-over-ranks timing use
RCCL, the data path does not.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODEL = "meta-llama/Llama-3.1-8B-Instruct"
L, H, D = 32, 8, 128
CTX, CHUNK = 16384, 256
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# HBM bytes per step from the PMC passes committed under profiles/ (see profiles/r01_*_pmc.md): updated by hand
# whenever the kernels' data flow changes; None until measured.
TRAFFIC_BYTES_PER_STEP = 4_561_000_000  # profiles/r01_f_pmc.md
VALU_BUSY_DOMINANT = 1.0                 # k_cdf_encode: SQ_ACTIVE_INST_VALU x 4 / (SIMDs x cycles), profiles/r01_f_pmc.md


def cachegen_bins_llama8b():
    """key_bins ++ value_bins for the 7B/8B family (cachegen_basics.py:49-60)."""
    kb = [32] * 10 + [16] * 22
    vb = [32] * 2 + [16] * 30
    return kb + vb


def make_kv(dev, seed):
    """Synthetic KV of the named shape: uniform [0,1) like the reference's own
    tests (tests/test_serde.py:20-21), generated on the GPU per layer."""
    g = torch.Generator(device=dev).manual_seed(seed)
    kv = []
    for _ in range(L):
        k = torch.rand((CTX, H, D), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
        v = torch.rand((CTX, H, D), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
        kv.append((k, v))
    return tuple(kv)


def cpu_baseline(nchunks_sample):
    """The CPU oracle (port of the reference's quantise + CDF + our entropy coder)
    timed on the host cores over a bounded sample of the same workload."""
    from oracle import lmc_oracle as orc
    orc.build()
    ncores = len(os.sched_getaffinity(0))
    # the oracle parallelises inside one chunk (64 planes, then 1024 group streams): on a many-core host several
    # chunks are encoded at once, each by its own OpenMP team, so that every core has work
    workers = 4 if ncores >= 32 else 1
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, ncores // workers)))
    g = torch.Generator().manual_seed(0)
    kv = torch.rand((L, 2, CHUNK, H * D), generator=g).to(torch.bfloat16)
    bits, code = orc.torch_to_bits(kv)
    bins = np.array(cachegen_bins_llama8b(), np.int32)
    orc.encode_blob(bits, code, H, D, bins)  # warm
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as pool:  # the ctypes call releases the GIL
        list(pool.map(lambda _: orc.encode_blob(bits, code, H, D, bins), range(nchunks_sample)))
    dt = time.perf_counter() - t0
    raw = kv.numel() * 2 * nchunks_sample
    return {"value": round(raw / dt / 1e9, 4), "unit": "GB/s", "cores": ncores, "kind": "port",
            "sample": f"{nchunks_sample} chunks of 256 tokens (Llama-3-8B shape, {raw / 1e6:.0f} MB raw KV), "
                      f"oracle/lmc_oracle.c lmco_encode_blob, {workers} chunks at a time x OpenMP over planes/groups"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-chunks", type=int, default=0, help="chunks in the cpu_baseline sample (0 = auto)")
    ap.add_argument("--exchange", action="store_true",
                    help="N>1 only, opt-in: time one RCCL/xGMI exchange step of the encoded chunks between ranks "
                         "(XgmiShardStore; outside the timed region, reported as `exchange`)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LMC_BENCH_FORCE_DIST=1 runs the process-group code path with a single rank too (a 1-GPU box can then
    # exercise init / barrier / all_reduce / teardown exactly as the multi-GPU launch does)
    use_dist = world > 1 or os.environ.get("LMC_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    from lmcache_amd import native
    ctx = native.get_context(local_rank)
    bins = cachegen_bins_llama8b()
    kv = make_kv(dev, seed=rank)
    layout = native.KVLayout.from_kv_tuple(kv, "vllm")
    nchunks = CTX // CHUNK
    stride = native.r16(native.blob_bound(L, CHUNK, H, D))
    blobs = torch.empty(nchunks * stride, dtype=torch.uint8, device=dev)
    sizes = torch.zeros(nchunks, dtype=torch.int32, device=dev)
    ctx.reserve(L, H, D, CHUNK, nchunks)
    raw_bytes = L * 2 * CTX * H * D * 2

    stream = torch.cuda.Stream(device=dev)
    sp = stream.cuda_stream

    def step():
        ctx.encode_chunks(layout, 0, CTX, CHUNK, bins, blobs.data_ptr(), stride, sizes.data_ptr(), stream=sp)

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.raise_on_status("bench warmup")

    # ---- timed region: exactly K steps, barrier + synchronize on both sides ----
    barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gpu_ms_per_step = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream
    barrier()
    ctx.raise_on_status("bench")
    if use_dist:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * raw_bytes * args.steps / elapsed / 1e9

    # ---- optional: one exchange step of the encoded chunks over RCCL/xGMI (row f1; outside the timed region) ----
    exchange = None
    if args.exchange and use_dist:
        import torch.distributed as dist
        from lmcache_amd.storage_backend.connector.xgmi_exchange import XgmiShardStore
        szs = sizes.cpu().tolist()
        store = XgmiShardStore()
        items = [(f"bench@{world}@{rank}@{i:04x}", blobs[i * stride:i * stride + szs[i]]) for i in range(nchunks)]
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        store.exchange_put(items)
        torch.cuda.synchronize(); dist.barrier()
        t_put = time.perf_counter() - t0
        peer = (rank + 1) % world
        want = [f"bench@{world}@{peer}@{i:04x}" for i in range(nchunks)]
        t0 = time.perf_counter()
        got = store.exchange_get(want)
        torch.cuda.synchronize(); dist.barrier()
        t_get = time.perf_counter() - t0
        nbytes = torch.tensor([float(sum(szs))], device=dev, dtype=torch.float64)
        dist.all_reduce(nbytes)
        ok = all(g is not None for g in got)
        exchange = {"put_GBps_blob_all_ranks": round(float(nbytes.item()) / t_put / 1e9, 1),
                    "get_GBps_blob_all_ranks": round(float(nbytes.item()) / t_get / 1e9, 1),
                    "put_ms": round(t_put * 1e3, 2), "get_ms": round(t_get * 1e3, 2), "all_hits": ok,
                    "note": "one batch_isend_irecv per call; includes the all_gather_object control round trips"}

    if rank != 0:
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()
            dist.destroy_process_group()
        return

    # ---- rank 0 extras (outside the timed region) -------------------------------
    sz = sizes.cpu().numpy().astype(np.int64)
    blob_bytes = int(sz.sum())
    P, C = 2 * L, H * D
    G = (C + 63) // 64
    # algorithmic bytes per step, SURVEY.md 8(d): B_enc = P*T*C*e + S + P*C*66 + P*C*4 + P*T*2 per chunk,
    # with OUR container: lengths are one u32 per 64-channel group (P*G*4) instead of per channel
    static = native.blob_static_bytes(L, CHUNK, H, D, bins)
    S = blob_bytes - nchunks * static
    algo_bytes = raw_bytes + blob_bytes

    # per-kernel HIP-event timing on the launch stream (lmc_ctx_profile)
    ctx.profile(True)
    knames = ["k_quantize", "k_cdf_encode"]
    ksum = np.zeros(len(knames))
    reps = max(3, min(10, args.steps))
    for _ in range(reps):
        step()
        torch.cuda.synchronize()
        ksum += np.array(ctx.profile_read()[:len(knames)])
    ctx.profile(False)
    kms = ksum / reps
    achieved = algo_bytes / (gpu_ms_per_step / 1e3) / 1e9
    serial = algo_bytes / (float(kms.sum()) / 1e3) / 1e9
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": TRAFFIC_BYTES_PER_STEP,
                "kernel": knames[int(np.argmax(kms))],
                "gpu_ms_per_step": round(gpu_ms_per_step, 4),
                "kernels_ms_serial": {n: round(float(v), 4) for n, v in zip(knames, kms)},
                "achieved_serial": round(serial, 1),
                "algorithmic_bytes_per_step": int(algo_bytes),
                # from the SQ PMC pass committed under profiles/ (not measured live): the dominant kernel is an
                # integer entropy coder and sits under the VALU-issue roof, not the HBM one
                "valu_busy_dominant_kernel": VALU_BUSY_DOMINANT,
                "note": "achieved = (raw KV read once + blob written once) per step / HIP-event time of one step "
                        "(the whole encode job: k_quantize + k_cdf_encode) on the launch stream over the timed region; "
                        "kernels_ms_serial = per-kernel HIP events of one job (lmc_ctx_profile); traffic = HBM bytes per "
                        "step from rocprofv3 FETCH_SIZE/WRITE_SIZE (profiles/), FETCH_SIZE doubled for the 16-B/lane "
                        "streams per MI355X_MICROARCH.md"}

    # store leg as the product runs it (LMCLocalBackend, local_serde="cachegen"): fused encode on the compute
    # stream, blob sizes read back through a pinned word, exact-size hipMemcpyAsync of every blob to pinned host
    # DRAM on the side stream.  PCIe-inclusive, so never `value`.
    from lmcache_amd.storage_backend.serde.cachegen_device import PinnedArena, get_codec
    offload = retrieve = None
    try:
        codec = get_codec(local_rank)
        with torch.cuda.stream(stream):
            arena = PinnedArena(slab_bytes=int(blob_bytes) + (4 << 20))  # the backend's pinned slab, allocated once

            def store_once():
                arena.reset()
                job = codec.encode(layout, 0, CTX, CHUNK, bins)
                hblobs, done = codec.offload(job, None, arena)
                done.synchronize()
                return hblobs

            hblobs = store_once()  # warm: first touch of the pinned slab
            t0 = time.perf_counter()
            for _ in range(3):
                hblobs = store_once()
            dt = (time.perf_counter() - t0) / 3
            offload = {"encode_plus_offload_GBps_raw_kv": round(raw_bytes / dt / 1e9, 1),
                       "pcie_GBps_blob": round(blob_bytes / dt / 1e9, 1), "ms_per_context": round(dt * 1e3, 3),
                       "blob_bytes": blob_bytes, "compression": round(raw_bytes / blob_bytes, 3),
                       "note": "PCIe-inclusive (pinned slab pre-allocated, as in a running backend); not `value`"}
            # warm-prefix retrieve: pinned host -> HBM on the side stream, decode straight into per-layer tensors,
            # H2D of batch b+1 overlapping the decode of batch b
            out_r = tuple((torch.empty_like(k), torch.empty_like(v)) for k, v in kv)
            out_rl = native.KVLayout.from_kv_tuple(out_r, "vllm")
            codec.decode(hblobs, out_rl, 0, CHUNK)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            codec.decode(hblobs, out_rl, 0, CHUNK)
            torch.cuda.current_stream().synchronize()
            dt = time.perf_counter() - t0
            ctx.raise_on_status("bench retrieve")
            retrieve = {"host_to_decoded_kv_GBps_raw": round(raw_bytes / dt / 1e9, 1), "ms_per_context": round(dt * 1e3, 3),
                        "pcie_GBps_blob": round(blob_bytes / dt / 1e9, 1),
                        "raw_h2d_would_take_ms": round(raw_bytes / (blob_bytes / dt) * 1e3, 1),
                        "note": "warm 16k prefix: encoded chunks in pinned host DRAM -> decoded KV in HBM (PCIe-inclusive)"}
            del out_r
            arena.close()
    except Exception as e:  # these legs are informational
        offload = offload or {"error": repr(e)}
        retrieve = retrieve or {"error": repr(e)}

    # decode leg (retrieve): blobs in HBM -> decoded KV written straight into per-layer tensors
    out = tuple((torch.empty_like(k), torch.empty_like(v)) for k, v in kv)
    out_layout = native.KVLayout.from_kv_tuple(out, "vllm")
    ctx.decode_chunks(blobs.data_ptr(), stride, nchunks, out_layout, 0, CHUNK, stream=sp)
    torch.cuda.synchronize()
    ctx.raise_on_status("bench decode")
    ctx.profile(True)
    dsum = 0.0
    for _ in range(5):
        ctx.decode_chunks(blobs.data_ptr(), stride, nchunks, out_layout, 0, CHUNK, stream=sp)
        torch.cuda.synchronize()
        dsum += ctx.profile_read()[0]
    ctx.profile(False)
    dms = dsum / 5
    decode = {"GBps_raw_kv": round(raw_bytes / (dms / 1e3) / 1e9, 1), "ms_per_context": round(dms, 3),
              "roofline_frac": round(algo_bytes / (dms / 1e3) / 1e9 / HBM_PEAK_GBS, 4)}
    # size-independent property at full size: decode(encode(x)) reproduces x within the quantisation bound
    k0, o0 = kv[0][0].float(), out[0][0].float()
    mx = k0.abs().amax(dim=(1, 2), keepdim=True)
    err_ok = bool(((o0 - k0).abs() <= mx / (2 * 15) + mx * 2.0 ** -7).all())

    res = {"metric": "KV encode+offload GB/s per GPU (raw 16-bit KV bytes consumed; encode into HBM blobs)",
           "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16->u8 symbols (fp32 quantise, u32 rANS)", "data": "synthetic",
           "config": {"workload": "Llama-3-8B bf16 KV, 16384-token context, CacheGen encode, chunk_size=256 "
                                  "(BASELINE configs[1])",
                      "layers": L, "kv_heads": H, "head_dim": D, "context_tokens": CTX, "chunk_tokens": CHUNK,
                      "chunks": nchunks, "raw_kv_bytes": raw_bytes, "sharding": f"{world} x independent contexts"},
           "roofline": roofline, "offload": offload, "retrieve": retrieve, "decode": decode, "roundtrip_within_bound": err_ok}
    if exchange is not None:
        res["exchange"] = exchange
    if not args.no_cpu_baseline:
        n = args.cpu_chunks or 128
        res["cpu_baseline"] = cpu_baseline(n)
    print(json.dumps(res))
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
