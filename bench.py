#!/usr/bin/env python
"""bench.py -- KV encode(+offload) throughput of the MI355X hot path.

Metric (BASELINE.json): "KV encode+offload GB/s per GPU; warm-prefix TTFT vs cold", GB = raw 16-bit KV
bytes consumed.  Workload at N=1: BASELINE configs[1] -- Llama-3-8B, bf16, 16 384-token context
(32 layers x 8 KV heads x 128, 2 GiB), chunk_size 256 (64 chunks), synthetic KV already resident in HBM
as the per-layer (K, V) tensors LMCacheEngine.store() receives.

One "step" = one pass of the encode path over the whole context through the C ABI (lmc_encode_chunks):
gather from the per-layer tensors, quantise, CDF, entropy-encode, compact into 64 blobs in an HBM arena.
`value` is that rate (inputs and outputs in HBM; PCIe never inside `value`).  Everything else is measured
outside the timed region, on rank 0, and reported as extra objects of the same JSON line:

  roofline       HBM roof of the step (algorithmic bytes / HIP-event time) + the VALU-issue roof the coder sits under
  offload        the product's store leg: encode + exact-size pinned hipMemcpyAsync D2H (PCIe-inclusive), >= 5 reps
  retrieve       pinned host -> HBM -> decode (PCIe-inclusive), >= 5 reps
  store_hidden   the metric's other half, part 1: a decode-step proxy (HBM read of a 16 GB weight buffer on the
                 compute stream) alone vs with a non-blocking engine.store() of the 16k context in flight
  ttft_proxy     part 2: (retrieve of the warm 16k prefix + one proxy step) / (one proxy step), target <= 1.05
  decode         blobs in HBM -> decoded KV
  seeds          the encode step on seeds 0..4 of the chosen --dist (min / median)
  other_configs  the other BASELINE geometries (configs[0], [3], [4]) HBM-resident, one job each
  cpu_baseline   the CPU oracle (C port) on a bounded sample + the reference's own torch formula on the host cores

N>1 (torch.distributed, one rank per GPU): the path shards by chunk with no data-path collective
(SURVEY.md section 8e), so every rank encodes its own 16k-token context (weak scaling); the barrier and the
max-over-ranks timing use RCCL, the data path does not.  Every rank also runs the offload leg at the same
time (its own PCIe link, NUMA-local pinned arena) and, for the sharing path (BASELINE configs[2]), one
exchange step of encoded chunks through the xgmi:// connector.

LMC_BENCH_STUB=1 rehearses the launch plumbing on CPU (gloo, the timed kernel replaced by a sleep): that is how
tests/test_distributed_cpu.py runs this file with two ranks.
"""
import argparse
import json
import os
import statistics
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
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# Issue-slot prices measured on MI355X at 8 waves per SIMD (tools/probes/valu_rates.py, tools/probes/issue_model.py;
# profiles/r05_issue_model.md), ns of SIMD time per wave-instruction:
ISSUE_NS = {"valu_fast": 1.05,    # v_add/sub/and/or/xor/lshrrev/mov/mul_f32/add_f32 without modifiers
            "valu_normal": 1.85,  # every other VALU instruction (VOP3, SDWA, DPP, cvt, cmp, mbcnt, lshlrev, packed, mul_hi ...)
            # what a scalar instruction ADDS to a VALU-bound step (alone: 0.94; a taken branch: + 1.5): two boxes measured
            # 0.43 and 1.3 ns (profiles/r05_issue_model.md section 2) -- the floor is carried as the range between them
            "salu_beside_valu": 0.43, "salu_beside_valu_slow_box": 1.3}
PROXY_BYTES = 16 * (1 << 30)  # one Llama-3-8B decode step streams ~16 GB of bf16 weights
STUB = os.environ.get("LMC_BENCH_STUB") == "1"


def cachegen_bins_llama8b():
    """key_bins ++ value_bins for the 7B/8B family (cachegen_basics.py:49-60)."""
    kb = [32] * 10 + [16] * 22
    vb = [32] * 2 + [16] * 30
    return kb + vb


def make_kv(dev, seed, dist="rand", nl=L, ntok=CTX, nh=H, hd=D, dtype=torch.bfloat16):
    """Synthetic KV of the named shape, generated on the GPU per layer:
    rand    uniform [0,1) like the reference's own tests (tests/test_serde.py:20-21)
    randn   standard normal
    outlier randn x a per-channel log-normal scale (a few loud channels, as real KV has)"""
    g = torch.Generator(device=dev).manual_seed(seed)
    kv = []
    for _ in range(nl):
        pair = []
        for _ in range(2):
            if dist == "rand":
                x = torch.rand((ntok, nh, hd), generator=g, device=dev, dtype=torch.float32)
            else:
                x = torch.randn((ntok, nh, hd), generator=g, device=dev, dtype=torch.float32)
                if dist == "outlier":
                    x = x * torch.exp(1.5 * torch.randn((1, nh, hd), generator=g, device=dev, dtype=torch.float32))
            pair.append(x.to(dtype))
        kv.append(tuple(pair))
    return tuple(kv)


def stage(name):
    """Progress marker on stderr (LMC_BENCH_VERBOSE=1): says which leg a fault belongs to."""
    if os.environ.get("LMC_BENCH_VERBOSE"):
        print(f"[bench] {name}", file=sys.stderr, flush=True)


def median(xs):
    return float(statistics.median(xs))


def load_latest_profile():
    """profiles/latest.json (written by tools/rocpd_stats.py --json from the committed rocprofv3 passes): HBM
    bytes and VALU instructions per encode step.  None when absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "latest.json")) as f:
            return json.load(f)
    except Exception:
        return None


def issue_roof(prof, gpu_ms_per_step):
    """The issue-slot floor of the dominant kernel: its VALU instructions priced by class (the static class mix of its
    hot loops x the dynamic count of the PMC pass) plus what its scalar instructions add beside them, per SIMD."""
    k = (prof.get("kernels") or {}).get(prof.get("dominant_kernel") or "", {})
    if not k or "valu_insts" not in k or "salu_insts" not in k:
        return None
    fast = k.get("valu_fast_share", 0.3)
    valu_ns = k["valu_insts"] * (fast * ISSUE_NS["valu_fast"] + (1 - fast) * ISSUE_NS["valu_normal"])
    salu_ns = k["salu_insts"] * ISSUE_NS["salu_beside_valu"]
    salu_ns_hi = k["salu_insts"] * ISSUE_NS["salu_beside_valu_slow_box"]
    floor_ms = (valu_ns + salu_ns) / 1024.0 / 1e6  # 1024 SIMDs
    floor_hi = (valu_ns + salu_ns_hi) / 1024.0 / 1e6
    return {"bound": "issue slots (VALU by class + SALU)", "floor_ms": round(floor_ms, 4),
            "floor_ms_range": [round(floor_ms, 4), round(floor_hi, 4)],
            "frac": round(floor_ms / gpu_ms_per_step, 4),
            "frac_range": [round(floor_ms / gpu_ms_per_step, 4), round(min(1.0, floor_hi / gpu_ms_per_step), 4)],
            "valu_ms": round(valu_ns / 1024.0 / 1e6, 4),
            "salu_ms": round(salu_ns / 1024.0 / 1e6, 4), "valu_insts": k["valu_insts"], "salu_insts": k["salu_insts"],
            "valu_fast_share": fast, "prices_ns": ISSUE_NS,
            "note": "a FIT, not an independent bound: floor = (VALU instructions x class price + SALU instructions x their "
                    "measured cost beside a VALU stream) / 1024 SIMDs, prices from replicas of the token loops "
                    "(tools/probes/issue_model.py); the range spans the two scalar prices two boxes measured (0.43 / 1.3 ns); "
                    "frac = floor / measured step: what is left is waiting (barriers, look-back, loads in phase A)"}


def fabric_roof(prof, gpu_ms_per_step, quantize_ms, quantize_bytes):
    """The second roof of the dominant kernel: bytes through the fabric (L2 <-> Infinity Cache / HBM, PMC-counted and
    calibrated: profiles/latest.json) per step / the measured step, against the rate the product's own pure-streaming
    kernel -- k_quantize, HBM-bound -- sustains on THIS box in THIS process (its bytes are known analytically: raw KV read
    once, symbols and scales written once)."""
    traffic = prof.get("traffic_bytes_per_step")
    if not traffic or not quantize_ms:
        return None
    rate = traffic / (gpu_ms_per_step / 1e3) / 1e9
    ceiling = quantize_bytes / (quantize_ms / 1e3) / 1e9
    return {"bound": "fabric (L2 fills + write-backs of the step)", "traffic": int(traffic), "rate": round(rate, 1),
            "ceiling": round(ceiling, 1), "unit": "GB/s", "frac": round(rate / ceiling, 4),
            "floor_ms": round(traffic / ceiling / 1e6, 4),
            "ceiling_source": f"k_quantize on this box: {quantize_bytes} B in {quantize_ms:.4f} ms (lmc_ctx_profile, median of 5 jobs "
                              "of the two-kernel path)",
            "note": "traffic / algorithmic bytes = what the symbols' trip through the workspace adds; floor_ms = the step "
                    "if it moved its present traffic at k_quantize's rate"}


# ---------------------------------------------------------------------------------------------- clock / power
class GpuSampler:
    """Shader clock and board power of the GPU while the bench runs, from the amdgpu hwmon files of ITS PCI device
    (freq1_input, power1_average): a thread that reads them every millisecond and never touches the GPU.  VERDICT r04 #7:
    fresh boxes differ by +- 7 % on the same build; the line says what clock and power the timed region ran at."""

    def __init__(self, dev_index):
        import glob
        import threading
        self.dir = None
        try:
            p = torch.cuda.get_device_properties(dev_index)
            bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
            hw = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            if hw:
                self.dir = hw[0]
        except Exception:
            self.dir = None
        self.samples = []  # (t, MHz, W)
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True) if self.dir else None

    def _read(self, name, scale):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return int(f.read()) / scale
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            w = self._read("power1_average", 1e6)
            if w is None:
                w = self._read("power1_input", 1e6)
            self.samples.append((time.perf_counter(), self._read("freq1_input", 1e6), w))
            time.sleep(0.001)

    def start(self):
        if self._thread:
            self._thread.start()

    def stop(self):
        if self._thread:
            self._stop.set()
            self._thread.join()

    def window(self, t0, t1):
        rows = [r for r in self.samples if t0 <= r[0] <= t1 and r[1] is not None]
        if not rows:
            return None
        mhz = sorted(r[1] for r in rows)
        w = sorted(r[2] for r in rows if r[2] is not None)
        return {"samples": len(rows), "sclk_MHz_median": mhz[len(mhz) // 2], "sclk_MHz_min": mhz[0],
                "power_W_median": w[len(w) // 2] if w else None, "source": self.dir}


# ---------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline(nchunks_sample):
    """The CPU oracle (port of the reference's quantise + CDF + our entropy coder) timed on the host cores over a
    bounded sample of the same workload, ONE CHUNK PER CORE (lmco_encode_blobs_parallel: a 16 k context is 64
    independent chunks, a store of several contexts any number of them), and beside it the reference's own formula
    (torch_quant_vectorized + do_dequantize, cachegen_encoder.py:40-61 / cachegen_decoder.py:24-35) as CPU torch ops."""
    from oracle import lmc_oracle as orc
    orc.build()
    ncores = len(os.sched_getaffinity(0))
    threads = orc.set_threads(ncores)
    g = torch.Generator().manual_seed(0)
    kv = torch.rand((L, 2, CHUNK, H * D), generator=g).to(torch.bfloat16)
    bits, code = orc.torch_to_bits(kv)
    bins = np.array(cachegen_bins_llama8b(), np.int32)
    t0 = time.perf_counter()
    orc.encode_blobs_parallel(bits, code, H, D, bins, 1)  # warm, and what ONE core takes for a chunk
    one = time.perf_counter() - t0
    # Thread scaling: t threads, one chunk each -- a warm round (the threads' buffers and the output
    # arena are allocated and first-touched there, by the thread that uses them), then timed rounds.  Round 4's
    # 9.6 MB/s per core at 256 threads against 200 MB/s alone was the allocator: every chunk malloc'ed, page-faulted
    # and freed ~66 MB of temporaries plus its 18 MB output slot, and 256 threads queued for the process's memory-map
    # lock (oracle/lmc_oracle.c: ws_get keeps the buffers per thread now).  `value` is the BEST point of the table.
    chunk_raw = kv.numel() * 2
    table, best = [], None
    budget_s, t_all = 24.0, time.perf_counter()
    for t in [x for x in (1, 8, 32, 64, 128, 256, 512) if x < ncores] + [ncores]:
        if time.perf_counter() - t_all > budget_s or t > max(1, nchunks_sample):
            break
        got = orc.set_threads(t)
        orc.encode_blobs_parallel(bits, code, H, D, bins, got)  # warm at this width
        rounds, t0 = 0, time.perf_counter()
        while rounds < 3 and (rounds == 0 or time.perf_counter() - t0 < 1.5):
            orc.encode_blobs_parallel(bits, code, H, D, bins, got)
            rounds += 1
        dt = time.perf_counter() - t0
        row = {"threads": got, "chunks": got * rounds, "GBps": round(chunk_raw * got * rounds / dt / 1e9, 3),
               "MBps_per_thread": round(chunk_raw * rounds / dt / 1e6, 1)}
        table.append(row)
        if best is None or row["GBps"] > best["GBps"]:
            best = row
    # the workers' buffers (66 MB per thread) and the output arena go back before the next legs run
    orc.release_workspaces(max(r["threads"] for r in table))
    orc.set_threads(ncores)
    done = sum(r["chunks"] for r in table)
    out = {"value": best["GBps"], "unit": "GB/s", "cores": best["threads"], "kind": "port",
           "one_chunk_one_core_s": round(one, 3), "scaling": table, "host_cores": ncores,
           "sample": f"{done} chunks of 256 tokens (Llama-3-8B shape, {chunk_raw / 1e6:.1f} MB raw KV each) over the thread counts of "
                     f"`scaling`, oracle/lmc_oracle.c lmco_encode_blobs_parallel: one chunk per OpenMP thread; value = the best "
                     f"row ({best['threads']} threads), OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND', 'unset')}"}
    out["reference_formula"] = cpu_reference_formula(kv)
    out["torch_serde"] = cpu_torch_serde()
    return out


def cpu_torch_serde():
    """The reference's own CPU serde of BASELINE configs[0]: TorchSerializer.to_bytes =
    torch.save(t.cpu().clone().detach()) into a BytesIO, TorchDeserializer.from_bytes = torch.load
    (lmcache/storage_backend/serde/torch_serde.py:16-31), on one config-1 chunk [32, 2, 256, 32, 128] fp16 = 128 MiB,
    restated line for line (the reference package needs torchac_cuda / nvtx stubs to import and is not on the GPU
    box).  One thread is what the reference's store() runs it on; N threads = N chunks serialised at once."""
    import io
    from concurrent.futures import ThreadPoolExecutor
    t = torch.rand((32, 2, 256, 32, 128), generator=torch.Generator().manual_seed(0)).to(torch.float16)
    raw = t.numel() * 2

    def to_bytes(x):
        with io.BytesIO() as f:
            torch.save(x.cpu().clone().detach(), f)
            return f.getvalue()

    def from_bytes(b):
        with io.BytesIO(b) as f:
            return torch.load(f, weights_only=True)

    blob = to_bytes(t)
    res = {"chunk_bytes": raw, "blob_bytes": len(blob)}
    reps, t0 = 0, time.perf_counter()
    while reps < 5 and time.perf_counter() - t0 < 4.0:
        to_bytes(t)
        reps += 1
    res["to_bytes_GBps_1_thread"] = round(raw * reps / (time.perf_counter() - t0) / 1e9, 3)
    reps, t0 = 0, time.perf_counter()
    while reps < 5 and time.perf_counter() - t0 < 4.0:
        from_bytes(blob)
        reps += 1
    res["from_bytes_GBps_1_thread"] = round(raw * reps / (time.perf_counter() - t0) / 1e9, 3)
    nthreads = min(len(os.sched_getaffinity(0)), 16)  # 16 x (128 MiB tensor + 128 MiB of bytes) in flight is plenty
    with ThreadPoolExecutor(max_workers=nthreads) as pool:
        t0 = time.perf_counter()
        list(pool.map(lambda _: to_bytes(t), range(nthreads)))
        dt = time.perf_counter() - t0
    res["to_bytes_GBps_n_threads"] = round(raw * nthreads / dt / 1e9, 3)
    res["n_threads"] = nthreads
    res["sample"] = ("one BASELINE configs[0] chunk (fp16 [32, 2, 256, 32, 128], 134 MB): torch.save / torch.load through "
                     "BytesIO as TorchSerializer / TorchDeserializer do; lossless, no compression (blob = raw + header)")
    return res


def cpu_quota():
    """CPUs this container may use at once (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        return None if q == "max" else max(1, int(int(q) / int(per)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        return None if q <= 0 else max(1, q // per)
    except Exception:
        return None


def cpu_reference_formula(kv_chunk):
    """torch_quant_vectorized (per token absmax, x * (MAX / max) + MAX, round, int8) followed by do_dequantize
    (((q - MAX) / MAX) * max) and the bf16 cast, restated line for line as CPU torch ops on one chunk
    [L,2,T,C], with 1 thread and with every host core."""
    bins = cachegen_bins_llama8b()
    x = kv_chunk  # bf16 [L,2,T,C]
    raw = x.numel() * 2

    def once():
        outs = []
        for kvi in range(2):
            for l in range(L):
                t = x[l, kvi]                                  # [T, C] bf16
                # the reference's bins are a float32 tensor (torch.zeros(n).fill_(..), cachegen_encoder.py:339-350), so
                # MAX is fp32 [.., 1, 1] and MAX / max1 and x * factor are fp32 (cachegen_encoder.py:54-59)
                MAX = torch.tensor(float(bins[kvi * L + l] // 2 - 1), dtype=torch.float32).reshape(1, 1)
                max1 = torch.amax(torch.abs(t), dim=-1, keepdim=True)
                q = torch.round(t * (MAX / max1) + MAX).to(torch.int8)
                d = ((q.float() - MAX) / MAX) * max1.float()
                outs.append(d.to(torch.bfloat16))
        return outs

    res = {}
    ncores = len(os.sched_getaffinity(0))
    # torch's intra-op pool on [256 x 1024] tensors stops scaling long before 256 threads (it gets slower), and a box
    # with 256 logical CPUs may run this container under a CPU quota of 16 (cpu.max): more runnable threads than the quota
    # only queue (round 5 read 0.34 GB/s at 64 threads, 6 x slower than one).  The "all" leg uses what the quota allows.
    quota = cpu_quota()
    nthreads_all = max(1, min(ncores, 64, quota or ncores))
    before = torch.get_num_threads()
    for label, nt in (("threads_1", 1), ("threads_all", nthreads_all)):
        torch.set_num_threads(nt)
        once()
        reps, t0 = 0, time.perf_counter()
        while reps < 5 and time.perf_counter() - t0 < 5.0:  # bounded: a few seconds per leg
            once()
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        res[label + "_GBps"] = round(raw / dt / 1e9, 3)
    torch.set_num_threads(before)
    res["threads_all"] = nthreads_all
    res["cpu_quota"] = quota
    res["sample"] = "one 256-token Llama-3-8B chunk (33.5 MB raw KV): quantise + dequantise + bf16 cast, no entropy coding"
    return res


# ---------------------------------------------------------------------------------------------- helpers (GPU)
class DecodeStepProxy:
    """What a decode step does to the memory system: one pass over ~16 GB of bf16 weights on the compute stream, as 32
    GEMVs of [16384 x 16384] (one per "layer"; batch-1 decode is exactly weight streaming) -- an HBM-bound read, no
    vLLM in this image.  step() enqueues one pass."""

    def __init__(self, dev):
        n = 16384
        self.w = torch.empty((32, n, n), dtype=torch.bfloat16, device=dev)
        self.w.zero_()
        self.x = torch.ones(n, dtype=torch.bfloat16, device=dev)
        self.y = torch.empty(n, dtype=torch.bfloat16, device=dev)

    def step(self):
        for l in range(32):
            torch.mv(self.w[l], self.x, out=self.y)

    def time_steps(self, n):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for i in range(n):
            self.step()
            evs[i + 1].record()
        torch.cuda.synchronize()
        return [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]


class ColdPrefillProxy:
    """What a COLD 16k-token prefill costs at least: the dense GEMMs of Llama-3-8B over the whole prompt -- per layer
    QKV [T,4096]x[4096,6144], O [T,4096]x[4096,4096], gate+up [T,4096]x[4096,28672], down [T,14336]x[14336,4096]
    (2.29e14 FLOP over 32 layers, bf16, hipBLASLt through torch.matmul).  Attention, norms and the sampler are left
    out (no vLLM in this image), so the cold TTFT of a real engine is larger and warm_vs_cold smaller than reported.
    One layer's weights stand for all 32 (436 MB instead of 14 GB; they stream from HBM either way)."""

    def __init__(self, dev, ntok=CTX):
        g = torch.Generator(device=dev).manual_seed(0)
        mk = lambda *shape: (torch.randn(shape, generator=g, device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)
        self.x = mk(ntok, 4096)
        self.wqkv, self.wo, self.wgu, self.wd = mk(4096, 6144), mk(4096, 4096), mk(4096, 28672), mk(14336, 4096)
        self.flop = 32 * 2 * ntok * 4096 * (6144 + 4096 + 28672 + 14336)

    def run(self):
        x = self.x
        for _ in range(32):
            q = x @ self.wqkv
            o = q[:, :4096] @ self.wo
            gu = o @ self.wgu
            x = (gu[:, :14336] * gu[:, 14336:]) @ self.wd
        return x

    def time_ms(self, reps=3):
        self.run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.run()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return median(ts)


def time_encode(ctx, layout, ntok, chunk, bins, blobs, stride, sizes, sp, stream, reps):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    for _ in range(reps):
        ctx.encode_chunks(layout, 0, ntok, chunk, bins, blobs.data_ptr(), stride, sizes.data_ptr(), stream=sp)
    ev1.record(stream)
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps


def shape_rate(native, ctx, dev, name, nl, nh, hd, dtype, ntok, model, paged=False, cs=256, dist="rand", roundtrip=False):
    """Encode / decode rate of another geometry or input distribution (HBM-resident, one job)."""
    stage(f"shape {name}")
    from lmcache_amd.storage_backend.serde.cachegen_basics import CacheGenConfig
    kv = make_kv(dev, 0, dist, nl, ntok, nh, hd, dtype)
    lay = native.KVLayout.from_kv_tuple(kv, "vllm")
    bins = CacheGenConfig.from_model_name(model).plane_bins(nl)
    n = (ntok + cs - 1) // cs
    stride = native.r16(native.blob_bound(nl, cs, nh, hd))
    blobs = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    sizes = torch.zeros(n, dtype=torch.int32, device=dev)
    raw = nl * 2 * ntok * nh * hd * 2
    st = torch.cuda.current_stream(dev)
    time_encode(ctx, lay, ntok, cs, bins, blobs, stride, sizes, st.cuda_stream, st, 1)
    ctx.raise_on_status(name)
    # the same untimed clock ramp as the timed region of the main workload gets (--ramp-ms): ~0.1 s of the job itself
    t_r = time.perf_counter()
    while time.perf_counter() - t_r < 0.1:
        time_encode(ctx, lay, ntok, cs, bins, blobs, stride, sizes, st.cuda_stream, st, 10)
    tenc = time_encode(ctx, lay, ntok, cs, bins, blobs, stride, sizes, st.cuda_stream, st, 20)
    ol_token = None
    if paged:  # decode + scatter into a paged cache, non-contiguous (north-star NHBD layout): the mapping a vLLM block
        # manager produces -- blocks anywhere, a block's tokens in order -- and, as the worst case, every token at a random
        # slot of its own (the first decodes in blocks of eight rows, the second token by token: k_decode.h)
        bs = 16
        nblocks = (ntok + bs - 1) // bs + 5
        caches = [torch.zeros((2, nblocks, nh, bs, hd), dtype=dtype, device=dev) for _ in range(nl)]
        pos = torch.arange(ntok, device=dev)
        slots = torch.randperm(nblocks, device=dev)[pos // bs] * bs + pos % bs
        ol = native.KVLayout.paged(caches, slots, bs, "NHBD")
        ol_token = native.KVLayout.paged(caches, torch.randperm(nblocks * bs, device=dev)[:ntok], bs, "NHBD")
    else:
        out = tuple((torch.empty_like(k), torch.empty_like(v)) for k, v in kv)
        ol = native.KVLayout.from_kv_tuple(out, "vllm")
    ctx.decode_chunks(blobs.data_ptr(), stride, n, ol, 0, cs)
    torch.cuda.synchronize()
    ctx.raise_on_status(name)
    for _ in range(60):  # clock ramp
        ctx.decode_chunks(blobs.data_ptr(), stride, n, ol, 0, cs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ctx.decode_chunks(blobs.data_ptr(), stride, n, ol, 0, cs)
    e1.record()
    torch.cuda.synchronize()
    tdec = e0.elapsed_time(e1) / 20
    tdec_token = None
    if ol_token is not None:
        for _ in range(20):
            ctx.decode_chunks(blobs.data_ptr(), stride, n, ol_token, 0, cs)
        e0.record()
        for _ in range(20):
            ctx.decode_chunks(blobs.data_ptr(), stride, n, ol_token, 0, cs)
        e1.record()
        torch.cuda.synchronize()
        ctx.raise_on_status(name)
        tdec_token = e0.elapsed_time(e1) / 20
    row = {"workload": name, "raw_kv_MB": round(raw / 1e6, 1), "chunks": n, "chunk_tokens": cs,
           "encode_ms": round(tenc, 3), "encode_GBps_raw": round(raw / tenc / 1e6, 1),
           "decode_ms": round(tdec, 3), "decode_GBps_raw": round(raw / tdec / 1e6, 1),
           "compression": round(raw / int(sizes.sum()), 3)}
    if tdec_token is not None:
        row["slot_mapping"] = "blocks at random, a block's 16 tokens in order (vLLM)"
        row["decode_ms_every_token_at_a_random_slot"] = round(tdec_token, 3)
        # ... and the store side of the connector: the fused encode reading the paged cache itself (slot_mapping gather in
        # phase A; the cache holds what the decode above scattered into it)
        time_encode(ctx, ol, ntok, cs, bins, blobs, stride, sizes, st.cuda_stream, st, 10)
        row["encode_ms_from_the_paged_cache"] = round(time_encode(ctx, ol, ntok, cs, bins, blobs, stride, sizes, st.cuda_stream, st, 20), 3)
        ctx.raise_on_status(name)
    if roundtrip and not paged:
        # size-independent property at full size: decode(encode(x)) reproduces x within the quantisation bound,
        # |x^ - x| <= max1 / (2 M) + 1 ulp16(max1), per token row and plane -- checked on every plane
        ok = True
        for l in range(nl):
            for kvi in range(2):
                x, y = kv[l][kvi].float(), out[l][kvi].float()
                mx = x.abs().amax(dim=(1, 2), keepdim=True)
                m = bins[kvi * nl + l] // 2 - 1
                ok = ok and bool(((y - x).abs() <= mx / (2 * m) + mx * 2.0 ** -7).all())
        row["roundtrip_within_bound"] = ok
    return row


# ---------------------------------------------------------------------------------------------- main
def arm_watchdog(seconds, res, rank):
    """If the multi-GPU legs are not done after `seconds`, rank 0 prints the result of the timed region (which is
    complete by then) and every rank exits: a hung leg must not cost the run its one JSON line."""
    import threading

    def fire():
        if rank == 0:
            out = dict(res)
            out["multi_gpu_legs"] = f"not finished after {seconds} s (watchdog); the timed region above is unaffected"
            print(json.dumps(out), flush=True)
        os._exit(0)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dist", choices=["rand", "randn", "outlier"], default="rand",
                    help="synthetic KV distribution of the timed workload (SURVEY.md section 8d)")
    ap.add_argument("--ramp-ms", type=float, default=250.0,
                    help="untimed clock ramp before the warm-up steps: full steps for this many ms (0 = none)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = every rank encodes a 16 k context of its own (the default, what the driver's per-N "
                         "values assume); strong = ONE 16 k context, rank r takes chunks r, r + N, ... (SURVEY.md 8e: the "
                         "replicated-instance split, lmcache_amd.distributed.shard_chunks), value = the context's bytes / "
                         "max-over-ranks time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-legs", action="store_true",
                    help="for the rocprofv3 passes: after the timed region also 5 jobs of the two-kernel path and 5 HBM-resident "
                         "decodes of the context (k_quantize, k_cdf_encode, k_decode rows in the same database), then the line")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region and the roofline")
    ap.add_argument("--cpu-chunks", type=int, default=0, help="chunks in the cpu_baseline sample (0 = auto)")
    ap.add_argument("--no-exchange", action="store_true", help="N>1: skip the exchange leg")
    ap.add_argument("--legs-timeout", type=int, default=180, help="N>1: seconds the offload / exchange legs may take")
    ap.add_argument("--exchange-transport", choices=("both", "ipc", "rccl"), default="both",
                    help="N>1 exchange leg: HIP-IPC connector (xgmi://), RCCL batch_isend_irecv (XgmiShardStore), or both")
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LMC_BENCH_FORCE_DIST=1 runs the process-group code path with a single rank too (a 1-GPU box can then
    # exercise init / barrier / all_reduce / teardown exactly as the multi-GPU launch does)
    use_dist = world > 1 or os.environ.get("LMC_BENCH_FORCE_DIST") == "1"
    import torch.distributed as dist
    from lmcache_amd.distributed import bind_to_gpu_numa, max_over_ranks, sum_over_ranks
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if STUB:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"
    numa = None
    cpus_before = os.sched_getaffinity(0)
    if not STUB:
        torch.cuda.set_device(local_rank)
        # pinned arenas are first-touched by this process: run it on the NUMA node of its GPU (N>1: every rank its own)
        numa = bind_to_gpu_numa(local_rank)
    dev = torch.device("cpu") if STUB else torch.device(f"cuda:{local_rank}")
    raw_bytes = L * 2 * CTX * H * D * 2
    nchunks = CTX // CHUNK
    strong = args.scaling == "strong" and world > 1
    from lmcache_amd.distributed import shard_chunks
    my_chunks = shard_chunks(nchunks, rank, world) if strong else list(range(nchunks))
    # how many ranks the process group's backend really spans (an all_reduce of ones: on the GPU box that is RCCL)
    ranks_seen = int(round(sum_over_ranks(1.0, dev))) if use_dist else 1

    def sync():
        if not STUB:
            torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    if STUB:
        def step():
            time.sleep(0.002)
        ctx = None
    else:
        from lmcache_amd import native
        ctx = native.get_context(local_rank)
        bins = cachegen_bins_llama8b()
        kv = make_kv(dev, 0 if strong else rank, args.dist)  # strong: every rank holds the SAME context
        layout = native.KVLayout.from_kv_tuple(kv, "vllm")
        stride = native.r16(native.blob_bound(L, CHUNK, H, D))
        blobs = torch.empty(nchunks * stride, dtype=torch.uint8, device=dev)
        sizes = torch.zeros(nchunks, dtype=torch.int32, device=dev)
        ctx.reserve(L, H, D, CHUNK, nchunks)
        stream = torch.cuda.Stream(device=dev)
        sp = stream.cuda_stream

        if strong:
            def step():  # this rank's chunks of the shared context: i mod world == rank, one job per chunk
                for k, i in enumerate(my_chunks):
                    ctx.encode_chunks(layout, i * CHUNK, (i + 1) * CHUNK, CHUNK, bins, blobs.data_ptr() + k * stride, stride,
                                      sizes.data_ptr() + 4 * k, stream=sp)
        else:
            def step():
                ctx.encode_chunks(layout, 0, CTX, CHUNK, bins, blobs.data_ptr(), stride, sizes.data_ptr(), stream=sp)

    # Clock ramp, before the W warm-up steps and outside every timed region: a GPU that has just left idle runs its
    # first ~50 ms of work below its sustained clock (tools/probes/encode_ab.hip, alternating rounds: the first 40
    # jobs average 1.23 ms, every later round 1.185), which is longer than W + K steps of this workload.  The
    # same full steps on the same buffers, ~0.25 s of them.
    sampler = GpuSampler(local_rank) if (not STUB and rank == 0) else None
    if sampler:
        sampler.start()
    t_ramp0 = time.perf_counter()
    ramp_steps = 0
    if not STUB and args.ramp_ms > 0:
        t_r = time.perf_counter()
        while (time.perf_counter() - t_r) * 1e3 < args.ramp_ms:
            for _ in range(10):
                step()
            sync()
            ramp_steps += 10
    for _ in range(args.warmup):
        step()
    sync()
    if ctx is not None:
        ctx.raise_on_status("bench warmup")

    # ---- timed region: exactly K steps, barrier + synchronize on both sides ----
    barrier()
    sync()
    if not STUB:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if not STUB:
        ev0.record(stream)
    for _ in range(args.steps):
        step()
    if not STUB:
        ev1.record(stream)
    sync()
    elapsed = time.perf_counter() - t0
    barrier()
    if sampler:
        sampler.stop()
    gpu_ms_per_step = elapsed * 1e3 / args.steps if STUB else ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream
    if ctx is not None:
        ctx.raise_on_status("bench")
    elapsed = max_over_ranks(elapsed, dev)
    ms_per_step = elapsed * 1e3 / args.steps
    # weak: every rank encoded a context of its own; strong: all ranks together encoded ONE context
    value = (1 if strong else world) * raw_bytes * args.steps / elapsed / 1e9

    res = {"metric": "KV encode+offload GB/s per GPU -- value = CacheGen encode, HBM -> HBM (raw 16-bit KV bytes consumed, "
                     "PCIe never inside value); the PCIe-inclusive encode+offload rate is offload.encode_plus_offload_GBps_raw_kv",
           "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
           "vs_baseline": None, "rccl_ranks_seen": ranks_seen,
           "dtype": "bf16->u8 symbols (fp32 quantise, u32 rANS)", "data": f"synthetic ({args.dist})",
           "config": {"workload": "Llama-3-8B bf16 KV, 16384-token context, CacheGen encode, chunk_size=256 "
                                  "(BASELINE configs[1])",
                      "layers": L, "kv_heads": H, "head_dim": D, "context_tokens": CTX, "chunk_tokens": CHUNK,
                      "chunks": nchunks, "raw_kv_bytes": raw_bytes,
                      "sharding": (f"one context, chunks i mod {world} == rank ({len(my_chunks)} chunks on rank 0)" if strong
                                   else f"{world} x independent contexts"),
                      "numa_node": numa},
           "gpu_state": None if not sampler else {
               "timed_region": sampler.window(t0, t0 + elapsed), "ramp_and_warmup": sampler.window(t_ramp0, t0),
               "note": "amdgpu hwmon of the bench's GPU sampled every ms by a host thread: the shader clock and board power "
                       "the timed K steps ran at (fresh boxes differ by +- 7 % on one build; DESIGN.md section 6)"},
           "clock_ramp": {"untimed_steps_before_warmup": ramp_steps, "ms": args.ramp_ms,
                          "why": "a GPU fresh out of idle runs its first ~50 ms below its sustained clock; W + K steps "
                                 "of this workload are shorter than that (--ramp-ms 0 turns it off)"}}
    # ---- N>1: every rank's own offload leg at the same time (its PCIe link, its NUMA-local arena), and one
    # exchange step of encoded chunks (BASELINE configs[2]) by HIP-IPC and by RCCL; outside the timed region.  A
    # watchdog guards the contract: should a multi-GPU leg hang (a peer that died, a collective out of step), rank 0
    # still prints the JSON line of the timed region and every rank leaves.
    offload_all = exchange = None
    if use_dist and not STUB and not args.no_extras:
        wd = arm_watchdog(args.legs_timeout, res, rank)
        offload_all = all_ranks_offload(ctx, layout, bins, dev, local_rank, world)
        if not args.no_exchange:
            exchange = exchange_leg(blobs, sizes, stride, nchunks, rank, world, dev,
                                    ("ipc", "rccl") if args.exchange_transport == "both" else (args.exchange_transport,))
        wd.cancel()

    if rank != 0:
        if use_dist:
            dist.barrier()
            sync()
            dist.destroy_process_group()
        return

    if STUB:
        res["data"] = "stub (CPU rehearsal of the launch plumbing, no kernel ran)"
        print(json.dumps(res))
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- rank 0 extras (outside the timed region) -------------------------------
    from lmcache_amd import native
    sz = sizes.cpu().numpy().astype(np.int64)
    blob_bytes = int(sz.sum())
    # algorithmic bytes per step (with OUR container): raw KV read once + blobs written once
    algo_bytes = raw_bytes * len(my_chunks) // nchunks + blob_bytes  # (strong scaling: this rank's share of the context)

    # per-kernel HIP-event timing on the launch stream (lmc_ctx_profile).  At this size lmc_encode_chunks launches
    # the fused kernel (one entry); k_quantize + k_cdf_encode is timed beside it as `encode_paths` below.
    ctx.profile(True)
    step()
    torch.cuda.synchronize()
    knames = ["k_encode_fused"] if len(ctx.profile_read()) == 1 else ["k_quantize", "k_cdf_encode"]
    ksum = np.zeros(len(knames))
    reps = max(3, min(10, args.steps))
    for _ in range(reps):
        step()
        torch.cuda.synchronize()
        ksum += np.array(ctx.profile_read()[:len(knames)])
    ctx.profile(False)
    kms = ksum / reps
    achieved = algo_bytes / (gpu_ms_per_step / 1e3) / 1e9
    prof = load_latest_profile() or {}
    valu_insts = prof.get("valu_insts_per_step")
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": prof.get("traffic_bytes_per_step"),
                "kernel": knames[int(np.argmax(kms))],
                "gpu_ms_per_step": round(gpu_ms_per_step, 4),
                "kernels_ms_serial": {n: round(float(v), 4) for n, v in zip(knames, kms)},
                "algorithmic_bytes_per_step": int(algo_bytes),
                # the dominant kernel is an integer entropy coder: the roof it sits under is instruction issue, not HBM --
                # VALU instructions priced by class, with the scalar side (VERDICT r04 #1a / weak #2: the flat 4-cycle
                # `valu` roof of rounds 3-4 is gone): profiles/r05_issue_model.md
                "valu_insts_per_step": valu_insts,
                "issue": issue_roof(prof, gpu_ms_per_step),
                "profile_source": prof.get("source"),
                "note": "achieved = (raw KV read once + blob written once) per step / HIP-event time of one step "
                        "(the whole encode job: one k_encode_fused launch) on the launch stream over the timed region; "
                        "kernels_ms_serial = per-kernel HIP events of one job (lmc_ctx_profile); traffic and the VALU "
                        "instruction count come from the rocprofv3 PMC passes summarised in profiles/latest.json "
                        "(FETCH_SIZE doubled for the 16-B/lane streams per MI355X_MICROARCH.md)"}
    res["roofline"] = roofline
    if args.profile_legs:
        ctx.set_encode_path("two_kernels")
        for _ in range(5):
            step()
        ctx.set_encode_path("auto")
        out_kv = tuple((torch.empty_like(k), torch.empty_like(v)) for k, v in kv)
        ol = native.KVLayout.from_kv_tuple(out_kv, "vllm")
        for _ in range(5):
            ctx.decode_chunks(blobs.data_ptr(), stride, nchunks, ol, 0, CHUNK)
        torch.cuda.synchronize()
        ctx.raise_on_status("profile legs")
    res["encode_paths"] = encode_paths_ab(ctx, step, stream, max(5, min(20, args.steps)))
    if not strong:
        # the fabric roof next to the issue roof: the ceiling is k_quantize's rate measured here
        nib = sum(1 for b in bins if b <= 17)
        elems = raw_bytes // 2
        qbytes = raw_bytes + elems * nib // len(bins) // 2 + elems * (len(bins) - nib) // len(bins) + 2 * len(bins) * CTX
        roofline["fabric"] = fabric_roof(prof, gpu_ms_per_step, res["encode_paths"].get("k_quantize_ms"), qbytes)
    # what ONE store() out of idle costs (the timed region above is steady state: the clock ramp is outside it)
    cold = []
    for _ in range(3):
        torch.cuda.synchronize()
        time.sleep(1.0)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        step()
        c1.record(stream)
        torch.cuda.synchronize()
        cold.append(round(c0.elapsed_time(c1), 4))
    res["cold_single_store_ms"] = {"median": median(cold), "runs": cold,
                                   "note": "one lmc_encode_chunks of the whole context after 1 s of idle GPU (HIP events on "
                                           "the launch stream): the clock ramp and the launch's first and last generation "
                                           "of workgroups are all inside; ms_per_step is the steady state"}

    if not args.no_extras and world == 1:  # the single-GPU legs (store / retrieve / TTFT proxies, other geometries)
        extras(res, args, ctx, native, dev, kv, layout, bins, blobs, sizes, stride, stream, sp, raw_bytes, blob_bytes,
               algo_bytes, gpu_ms_per_step)
    if isinstance(res.get("offload"), dict) and "encode_plus_offload_GBps_raw_kv" in res["offload"]:
        # the metric's PCIe-inclusive form next to `value` (encode, then every blob over PCIe into pinned host DRAM)
        res["value_with_offload"] = {"GBps_raw_kv": res["offload"]["encode_plus_offload_GBps_raw_kv"],
                                     "GBps_blob_over_pcie": res["offload"]["pcie_GBps_blob"],
                                     "ms_per_context": res["offload"]["ms_per_context"]}
    if offload_all is not None:
        res["offload_all_ranks"] = offload_all
    if exchange is not None:
        res["exchange"] = exchange
    if not args.no_cpu_baseline and world == 1:  # a reported baseline of the N = 1 line
        os.sched_setaffinity(0, cpus_before)  # the CPU baseline gets every host core, not only the GPU's NUMA node
        res["cpu_baseline"] = cpu_baseline(args.cpu_chunks or 4 * len(os.sched_getaffinity(0)))
    print(json.dumps(res))
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()


def encode_paths_ab(ctx, step, stream, reps):
    """The same job through both launch paths of lmc_encode_chunks (lmc_ctx_set_encode_path), back to back on the
    launch stream: ms per 16k context."""
    import torch
    out = {}
    try:
        for rnd in range(2):  # alternating rounds, the second one reported: neither path gets the warmer GPU
            for name in ("two_kernels", "fused"):
                ctx.set_encode_path(name)
                step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(reps):
                    step()
                e1.record(stream)
                torch.cuda.synchronize()
                out[name + "_ms"] = round(e0.elapsed_time(e1) / reps, 4)
        # k_quantize alone (per-kernel HIP events of the two-kernel path): the product's pure-streaming kernel, whose
        # rate is the fabric roof's ceiling
        ctx.set_encode_path("two_kernels")
        ctx.profile(True)
        q = []
        for _ in range(5):
            step()
            torch.cuda.synchronize()
            pr = ctx.profile_read()
            if len(pr) >= 2:
                q.append(pr[0])
        ctx.profile(False)
        if q:
            out["k_quantize_ms"] = round(float(statistics.median(q)), 4)
    finally:
        ctx.profile(False)
        ctx.set_encode_path("auto")
    out["reps"] = reps
    out["note"] = "default = auto: fused when chunks x planes > 4 x CUs (this workload: 4096 > 1024)"
    return out


def all_ranks_offload(ctx, layout, bins, dev, local_rank, world):
    """Every rank stores its context to ITS pinned arena at the same time: what the host side (PCIe links, DRAM
    channels) sustains when all GPUs of the node offload together."""
    import torch.distributed as dist
    from lmcache_amd.distributed import max_over_ranks, sum_over_ranks
    from lmcache_amd.storage_backend.serde.cachegen_device import PinnedArena, get_codec
    codec = get_codec(local_rank)
    arena = PinnedArena(slab_bytes=640 << 20)

    def once():
        arena.reset()
        job = codec.encode(layout, 0, CTX, CHUNK, bins)
        hblobs, done = codec.offload(job, None, arena)
        done.synchronize()
        return sum(b.nbytes for b in hblobs)

    once()
    dist.barrier()
    t0 = time.perf_counter()
    nbytes = 0
    for _ in range(3):
        nbytes = once()
    dt = (time.perf_counter() - t0) / 3
    dt = max_over_ranks(dt, dev)
    tot = sum_over_ranks(float(nbytes), dev)
    arena.close()
    return {"pcie_GBps_blob_all_ranks": round(tot / dt / 1e9, 1), "ms_per_context_max": round(dt * 1e3, 3),
            "raw_kv_GBps_all_ranks": round(world * L * 2 * CTX * H * D * 2 / dt / 1e9, 1),
            "note": "all ranks store their 16k context to their own NUMA-local pinned arena concurrently"}


def exchange_leg(blobs, sizes, stride, nchunks, rank, world, dev, transports=("ipc", "rccl")):
    """One exchange of encoded chunks between the ranks (BASELINE configs[2]'s hand-over of a context between two
    instances), by either transport:
      ipc   the xgmi:// connector: every rank publishes its chunks (a blob lands in the HBM arena of the key's owner
            rank, a peer write through the HIP-IPC mapping), then fetches the next rank's chunks (peer reads);
            one-sided, no collective;
      rccl  xgmi_exchange.XgmiShardStore: the same routing as ONE batch_isend_irecv group per direction on the
            process group's nccl (= RCCL) backend; SPMD.
    Rates are blob bytes, whole job and per GPU."""
    import torch.distributed as dist
    from lmcache_amd.distributed import max_over_ranks, sum_over_ranks
    out = {}
    szs = sizes.cpu().tolist()
    views = [blobs[i * stride:i * stride + szs[i]] for i in range(nchunks)]
    tot = sum_over_ranks(float(sum(szs)), dev)
    peer = (rank + 1) % world

    def timed(fn):
        """fn() between barriers; a rank whose fn raises still takes part in every collective, and all ranks learn
        that the leg failed (so nobody waits in a barrier for a peer that has moved on)."""
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        r, err = None, None
        try:
            r = fn()
            torch.cuda.synchronize()
        except Exception as e:
            err = repr(e)
        dist.barrier()
        dt = max_over_ranks(time.perf_counter() - t0, dev)
        if sum_over_ranks(1.0 if err else 0.0, dev) > 0:
            raise RuntimeError(err or "a peer rank failed in this leg")
        return r, dt

    def line(t_put, t_get, ok, note):
        return {"put_GBps_blob_all_ranks": round(tot / t_put / 1e9, 1), "get_GBps_blob_all_ranks": round(tot / t_get / 1e9, 1),
                "put_GBps_blob_per_gpu": round(tot / world / t_put / 1e9, 1),
                "get_GBps_blob_per_gpu": round(tot / world / t_get / 1e9, 1),
                "put_ms": round(t_put * 1e3, 2), "get_ms": round(t_get * 1e3, 2), "all_hits": ok, "note": note}

    if "ipc" in transports:
        try:
            from lmcache_amd.storage_backend.connector import CreateConnector
            conn = CreateConnector(f"xgmi://bench{os.environ.get('MASTER_PORT', '0')}:{world}")
            _, t_put = timed(lambda: [conn.set_device(f"bench@{world}@{rank}@{i:04x}", views[i]) for i in range(nchunks)])
            got, t_get = timed(lambda: [conn.get_device(f"bench@{world}@{peer}@{i:04x}") for i in range(nchunks)])
            ok = all(g is not None and g.numel() > 0 for g in got)
            dist.barrier()
            conn.close()
            out["ipc"] = line(t_put, t_get, ok, "xgmi:// connector: blobs resident in the owner rank's HBM arena, peer "
                              "copies through HIP-IPC mappings, no collective")
        except Exception as e:  # informational leg
            out["ipc"] = {"error": repr(e)}
    if "rccl" in transports:
        try:
            from lmcache_amd.storage_backend.connector.xgmi_exchange import XgmiShardStore
            store = XgmiShardStore(device=dev)
            _, t_put = timed(lambda: store.exchange_put([(f"bench@{world}@{rank}@{i:04x}", views[i]) for i in range(nchunks)]))
            got, t_get = timed(lambda: store.exchange_get([f"bench@{world}@{peer}@{i:04x}" for i in range(nchunks)]))
            ok = all(g is not None and g.numel() == n for g, n in zip(got, _peer_sizes(szs, dev, world, peer)))
            out["rccl"] = line(t_put, t_get, ok, "XgmiShardStore: one batch_isend_irecv group per direction on the nccl "
                               "(= RCCL) backend, fixed-size metadata records by all_gather")
        except Exception as e:
            out["rccl"] = {"error": repr(e)}
    return out


def _peer_sizes(szs, dev, world, peer):
    """The chunk sizes of rank `peer` (all_gather of the size lists), to check what an exchange returned."""
    import torch.distributed as dist
    mine = torch.tensor(szs, dtype=torch.int64, device=dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return parts[peer].cpu().tolist()


def extras(res, args, ctx, native, dev, kv, layout, bins, blobs, sizes, stride, stream, sp, raw_bytes, blob_bytes,
           algo_bytes, gpu_ms_per_step):
    from lmcache_amd.cache_engine import LMCacheEngine
    from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
    from lmcache_amd.storage_backend.serde.cachegen_device import PinnedArena, get_codec
    nchunks = CTX // CHUNK
    local_rank = dev.index

    stage("store/retrieve legs")
    # ---- store / retrieve legs as the product runs them (PCIe-inclusive, never `value`) --------------------
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
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                hblobs = store_once()
                ts.append(time.perf_counter() - t0)
            dt = median(ts)
            # the D2H of the same blobs alone (already encoded): what PCIe takes by itself
            szl = sizes.cpu().tolist()
            td = []
            for _ in range(5):
                t0 = time.perf_counter()
                for i in range(nchunks):
                    native.memcpy_async(hblobs[i].ptr, blobs.data_ptr() + i * stride, szl[i], "d2h",
                                        (codec.copy_stream if i % 2 == 0 else codec.copy_stream2).cuda_stream)
                codec.copy_stream.synchronize()
                codec.copy_stream2.synchronize()
                td.append(time.perf_counter() - t0)
            d2h = median(td)
            enc = gpu_ms_per_step / 1e3
            overlap = (enc + d2h - dt) / min(enc, d2h) * 100.0
            offload = {"encode_plus_offload_GBps_raw_kv": round(raw_bytes / dt / 1e9, 1),
                       "pcie_GBps_blob": round(blob_bytes / dt / 1e9, 1), "ms_per_context": round(dt * 1e3, 3),
                       "d2h_alone_ms": round(d2h * 1e3, 3), "encode_alone_ms": round(enc * 1e3, 3),
                       "overlap_pct": round(max(0.0, min(100.0, overlap)), 1), "reps": 5,
                       "blob_bytes": blob_bytes, "compression": round(raw_bytes / blob_bytes, 3),
                       "note": "median of 5; PCIe-inclusive (pinned slab pre-allocated, as in a running backend); not "
                               "`value`.  overlap_pct = share of the shorter leg (encode) hidden behind the longer (D2H)"}
            # warm-prefix retrieve: pinned host -> HBM on the side stream, decode straight into per-layer tensors,
            # H2D of batch b+1 overlapping the decode of batch b
            out_r = tuple((torch.empty_like(k), torch.empty_like(v)) for k, v in kv)
            out_rl = native.KVLayout.from_kv_tuple(out_r, "vllm")
            codec.finish_decode(codec.decode(hblobs, out_rl, 0, CHUNK))
            tr = []
            for _ in range(5):
                t0 = time.perf_counter()
                codec.finish_decode(codec.decode(hblobs, out_rl, 0, CHUNK))
                tr.append(time.perf_counter() - t0)
            dt = median(tr)
            retrieve = {"host_to_decoded_kv_GBps_raw": round(raw_bytes / dt / 1e9, 1), "ms_per_context": round(dt * 1e3, 3),
                        "pcie_GBps_blob": round(blob_bytes / dt / 1e9, 1), "reps": 5,
                        "raw_h2d_would_take_ms": round(raw_bytes / (blob_bytes / dt) * 1e3, 1),
                        "note": "median of 5; warm 16k prefix: encoded chunks in pinned host DRAM -> decoded KV in HBM "
                                "(PCIe-inclusive; the decode kernel is hidden behind the H2D)"}
            del out_r
            arena.close()
    except Exception as e:  # these legs are informational
        offload = offload or {"error": repr(e)}
        retrieve = retrieve or {"error": repr(e)}
    res["offload"], res["retrieve"] = offload, retrieve

    stage("store leg through the C ABI")
    # ---- the same store leg as ONE C-ABI call (lmc_store_chunks): no Python sequencing, no host wait ---------------
    try:
        cap = int(blob_bytes) + (64 << 20)
        harena = native.PinnedBuffer(cap)
        hmeta = native.PinnedBuffer(8 * (nchunks + 1) + 4 * nchunks + 64)
        p_offs, p_sizes, p_status = hmeta.ptr, hmeta.ptr + 8 * (nchunks + 1), hmeta.ptr + 8 * (nchunks + 1) + 4 * nchunks
        hstatus = hmeta.tensor[8 * (nchunks + 1) + 4 * nchunks:8 * (nchunks + 1) + 4 * nchunks + 4].view(torch.int32)
        hstatus[0] = 0
        calls, totals = [], []
        for r in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.store_chunks(layout, 0, CTX, CHUNK, bins, harena.ptr, cap, p_offs, p_sizes, stream=sp, status_ptr=p_status)
            calls.append((time.perf_counter() - t0) * 1e3)
            stream.synchronize()
            totals.append((time.perf_counter() - t0) * 1e3)
        used = int(hmeta.tensor[8 * nchunks:8 * (nchunks + 1)].view(torch.int64)[0])
        res["offload_c_abi"] = {"ms_per_context": round(median(totals[1:]), 3),
                                "store_call_returns_after_ms": round(median(calls[1:]), 3),
                                "encode_plus_offload_GBps_raw_kv": round(raw_bytes / median(totals[1:]) / 1e6, 1),
                                "pcie_GBps_blob": round(used / median(totals[1:]) / 1e6, 1), "host_bytes": used,
                                "status": int(hstatus[0]), "reps": 5,
                                "note": "lmc_store_chunks: encode in 4 parts + a device-side copy kernel per part that "
                                        "reads the sizes on the GPU and writes the blobs, exact size, into the mapped "
                                        "pinned arena (k_offload.h); the call returns without any host wait"}
        # ... and its layer-major form (lmc_store_pack / lmc_load_pack): what the pinned tier of the engine stores
        hstatus[0] = 0
        calls, totals = [], []
        for r in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.store_pack(layout, 0, CTX, CHUNK, bins, harena.ptr, cap, p_sizes, stream=sp, status_ptr=p_status)
            calls.append((time.perf_counter() - t0) * 1e3)
            stream.synchronize()
            totals.append((time.perf_counter() - t0) * 1e3)
        ph = native.pack_info(harena.ptr, cap)
        out_p = torch.empty((L, 2, CTX, H, D), dtype=torch.bfloat16, device=dev)
        lay_p = native.KVLayout.from_chunk(out_p, "vllm")
        loads = {}
        for lpr in (0, 8):
            tl = []
            for r in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.load_pack(harena.ptr, ph.total_bytes, 0, 0, lay_p, 0, lpr, None, stream=sp, status_ptr=p_status)
                stream.synchronize()
                tl.append((time.perf_counter() - t0) * 1e3)
            loads[lpr] = median(tl[1:])
        res["offload_pack"] = {"store_ms_per_context": round(median(totals[1:]), 3),
                               "store_call_returns_after_ms": round(median(calls[1:]), 3),
                               "pack_bytes": int(ph.total_bytes), "pcie_GBps_store": round(ph.total_bytes / median(totals[1:]) / 1e6, 1),
                               "load_ms_per_context": round(loads[0], 3), "load_ms_per_context_8_layer_ranges": round(loads[8], 3),
                               "pcie_GBps_load": round(ph.total_bytes / loads[0] / 1e6, 1), "status": int(hstatus[0]),
                               "note": "lmc_store_pack: whole encode, then one device-side copy kernel writes the blobs "
                                       "transposed (static sections, then streams ordered plane / chunk: pack v3) into the "
                                       "mapped pinned region -- the idle-GPU form; the engine's store (store_hidden) ships the "
                                       "pack in parts by DMA; lmc_load_pack: table + static sections, then two "
                                       "hipMemcpyAsync (K run, V run) and one decode launch per range of layers"}
        del out_p
        harena.free()
        hmeta.free()
    except Exception as e:
        res.setdefault("offload_c_abi", {"error": repr(e)})
        res.setdefault("offload_pack", {"error": repr(e)})

    stage("decode leg")
    # ---- decode leg: blobs in HBM -> decoded KV written straight into per-layer tensors ----------------------
    out = tuple((torch.empty_like(k), torch.empty_like(v)) for k, v in kv)
    out_layout = native.KVLayout.from_kv_tuple(out, "vllm")
    ctx.decode_chunks(blobs.data_ptr(), stride, nchunks, out_layout, 0, CHUNK, stream=sp)
    torch.cuda.synchronize()
    ctx.raise_on_status("bench decode")
    ctx.profile(True)
    ds = []
    for _ in range(5):
        ctx.decode_chunks(blobs.data_ptr(), stride, nchunks, out_layout, 0, CHUNK, stream=sp)
        torch.cuda.synchronize()
        ds.append(ctx.profile_read()[0])
    ctx.profile(False)
    dms = median(ds)
    # ... and back to back, as the encode step is timed: the same untimed clock ramp first (--ramp-ms of the job itself;
    # the legs before this one are PCIe-bound and leave the shader clock low: without it ten launches, 9 ms, read 10 %
    # higher than the twenty launches of the other_dists / other_configs rows, which ramp), then HIP events on the launch
    # stream around 20 launches
    t_r = time.perf_counter()
    while args.ramp_ms > 0 and (time.perf_counter() - t_r) * 1e3 < args.ramp_ms:
        for _ in range(10):
            ctx.decode_chunks(blobs.data_ptr(), stride, nchunks, out_layout, 0, CHUNK, stream=sp)
        torch.cuda.synchronize()
    de0, de1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    de0.record(stream)
    for _ in range(20):
        ctx.decode_chunks(blobs.data_ptr(), stride, nchunks, out_layout, 0, CHUNK, stream=sp)
    de1.record(stream)
    torch.cuda.synchronize()
    ctx.raise_on_status("bench decode")
    dbb = de0.elapsed_time(de1) / 20
    res["decode"] = {"GBps_raw_kv": round(raw_bytes / (dbb / 1e3) / 1e9, 1), "ms_per_context": round(dbb, 3),
                     "ms_single_launch": round(dms, 3),
                     "roofline_frac": round(algo_bytes / (dbb / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                     "note": "ms_per_context: 20 launches back to back behind the same untimed clock ramp as the encode "
                             "step gets (as the encode step is timed); ms_single_launch: median of 5 isolated launches "
                             "out of a cold clock (ramp and tail included)"}
    # size-independent property at full size: decode(encode(x)) reproduces x within the quantisation bound
    k0, o0 = kv[0][0].float(), out[0][0].float()
    mx = k0.abs().amax(dim=(1, 2), keepdim=True)
    res["roundtrip_within_bound"] = bool(((o0 - k0).abs() <= mx / (2 * 15) + mx * 2.0 ** -7).all())
    del out, k0, o0

    stage("overlap legs")
    # ---- the metric's other half: store hidden behind decode, warm-prefix TTFT (BASELINE.json north_star) -------
    try:
        res.update(overlap_legs(dev, kv, raw_bytes))
    except Exception as e:
        res["store_hidden"] = {"error": repr(e)}

    stage("seeds")
    # ---- the encode step on seeds 0..4 of the chosen distribution -------------------------------------------
    per_seed = []
    for s in range(5):
        kvs = make_kv(dev, s, args.dist)
        ls = native.KVLayout.from_kv_tuple(kvs, "vllm")
        time_encode(ctx, ls, CTX, CHUNK, bins, blobs, stride, sizes, sp, stream, 1)
        per_seed.append(round(time_encode(ctx, ls, CTX, CHUNK, bins, blobs, stride, sizes, sp, stream, 5), 4))
        del kvs, ls
    ctx.raise_on_status("bench seeds")
    res["seeds"] = {"dist": args.dist, "ms_per_step": per_seed, "min": min(per_seed), "median": median(per_seed),
                    "GBps_raw_median": round(raw_bytes / median(per_seed) / 1e6, 1)}

    stage("two stores at a time")
    # ---- two stores of the context in flight at a time (two requests on two streams, each with its own blob arena; the
    # context owns two workspaces for exactly this).  NOT `value`: the timed region issues its steps on one stream, where
    # a launch's last generation of workgroups (coding only, fabric idle) and the next launch's first one (fetching
    # only, coders idle) cannot overlap -- ~0.1 ms per launch (profiles/r06_decoder_and_timelines.md section 3).
    try:
        s2 = torch.cuda.Stream(device=dev)
        blobs2 = torch.empty_like(blobs)
        sizes2 = torch.zeros_like(sizes)
        pairs = 10

        def two_at_a_time():
            for _ in range(pairs):
                ctx.encode_chunks(layout, 0, CTX, CHUNK, bins, blobs.data_ptr(), stride, sizes.data_ptr(), stream=sp)
                ctx.encode_chunks(layout, 0, CTX, CHUNK, bins, blobs2.data_ptr(), stride, sizes2.data_ptr(), stream=s2.cuda_stream)
        two_at_a_time()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        two_at_a_time()
        torch.cuda.synchronize()
        tt = (time.perf_counter() - t0) * 1e3 / (2 * pairs)
        ctx.raise_on_status("bench two stores")
        sz, sz2 = sizes.cpu().tolist(), sizes2.cpu().tolist()  # (the arenas are torch.empty: compare the blobs, not the slack)
        same = sz == sz2 and all(bool(torch.equal(blobs[i * stride:i * stride + sz[i]], blobs2[i * stride:i * stride + sz[i]]))
                                 for i in range(len(sz)))
        res["two_stores_at_a_time"] = {"ms_per_context": round(tt, 4), "GBps_raw_kv": round(raw_bytes / tt / 1e6, 1),
                                       "roofline_frac": round(algo_bytes / (tt / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "blobs_equal": same,
                                       "note": "two lmc_encode_chunks of the 16k context in flight on two streams, wall clock over "
                                               "20 contexts / 20: the fused kernel's steady state (not `value`: one request at a "
                                               "time is what the timed region measures)"}
        del blobs2, sizes2
    except Exception as e:
        res["two_stores_at_a_time"] = {"error": repr(e)}

    stage("other distributions")
    # ---- SURVEY.md 8d names three input distributions: the timed one (--dist, default rand) and the two others here ----
    try:
        res["other_dists"] = [dict(shape_rate(native, ctx, dev, f"Llama-3-8B bf16, 16k tokens, chunk_size 256, {d} KV", L, H, D,
                                              torch.bfloat16, CTX, MODEL, dist=d, roundtrip=True), dist=d)
                              for d in ("rand", "randn", "outlier") if d != args.dist]
    except Exception as e:
        res["other_dists"] = [{"error": repr(e)}]

    stage("other configs")
    # ---- the other BASELINE geometries, HBM-resident --------------------------------------------------------
    try:
        others = [shape_rate(native, ctx, dev, "configs[0] shape: fp16 [32 L, 32 H, 4096 tok, 128 hd] (C = 4096)",
                             32, 32, 128, torch.float16, 4096, "mistralai/Mistral-7B-Instruct-v0.2"),
                  shape_rate(native, ctx, dev, "configs[3] rank shape: Llama-3-70B TP=8, 32k context (80 L, C = 128)",
                             80, 1, 128, torch.bfloat16, 32768, "Llama-3-70B"),
                  shape_rate(native, ctx, dev, "configs[4]: Mistral-7B, 8 x 2048 tokens, decode + scatter into paged "
                                               "NHBD blocks", 32, 8, 128, torch.bfloat16, 16384,
                             "mistralai/Mistral-7B-Instruct-v0.2", paged=True),
                  # chunk lengths other than 256 (the reference's encode_function takes any, its
                  # tests/test_serde.py:87-107 uses 236): the counts model scaled to a sum of 256 + the fused kernel
                  shape_rate(native, ctx, dev, "Llama-3-8B bf16, 16 284 tokens in chunks of 236 (tests/test_serde.py:87-107's length)",
                             32, 8, 128, torch.bfloat16, 69 * 236, MODEL, cs=236),
                  shape_rate(native, ctx, dev, "Llama-3-8B bf16, 16k tokens, chunk_size = 128",
                             32, 8, 128, torch.bfloat16, 16384, MODEL, cs=128),
                  shape_rate(native, ctx, dev, "Llama-3-8B bf16, 16 484 tokens, chunk_size 256 + a ragged last chunk of 100",
                             32, 8, 128, torch.bfloat16, 16384 + 100, MODEL)]
    except Exception as e:
        others = [{"error": repr(e)}]
    res["other_configs"] = others


def overlap_legs(dev, kv, raw_bytes):
    """store_hidden and ttft_proxy through LMCacheEngine (local_device="cpu", local_serde="cachegen": encoded chunks
    in pinned host DRAM).  The model's decode step is a proxy: an HBM-bound pass over 16 GB on the compute stream;
    store() is issued on a side stream as the vLLM connector does (LLM_Engine.rst:91)."""
    from lmcache_amd.cache_engine import LMCacheEngine
    from lmcache_amd.config import LMCacheEngineConfig, LMCacheEngineMetadata
    cfg = LMCacheEngineConfig.from_legacy(chunk_size=CHUNK, backend="cpu", local_serde="cachegen")
    meta = LMCacheEngineMetadata(MODEL, 1, 0, "vllm", "bfloat16")
    proxy = DecodeStepProxy(dev)
    proxy.time_steps(3)
    alone = median(proxy.time_steps(10))
    side = torch.cuda.Stream(device=dev)
    hidden, store_ms = [], []
    reps = 5
    g = torch.Generator().manual_seed(1)
    engine = LMCacheEngine(cfg, meta)
    engine.engine_.host_arena.reserve((reps + 1) * (700 << 20), slab_bytes=2800 << 20)   # a backend sized for its working set
    for r in range(reps):
        toks = torch.randint(0, 32000, (CTX,), generator=g)
        last_key = engine._make_key(engine._prefix_hash(engine._chunk_tokens(toks))[-1], "vllm")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            engine.store(toks, kv, skip_existing=False, blocking=False)
        t_call = time.perf_counter() - t0
        steps = []
        done_at = None
        while done_at is None and len(steps) < 32:   # decode steps run while the store is in flight
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            proxy.step()
            e1.record()
            e1.synchronize()
            steps.append(e0.elapsed_time(e1))
            if engine.engine_.contains(last_key):
                done_at = time.perf_counter() - t0
        while done_at is None:
            time.sleep(0.0005)
            if engine.engine_.contains(last_key):
                done_at = time.perf_counter() - t0
        hidden.append(median(steps))
        # (the loop above looks once per proxy step, i.e. every 3 ms: the backend's own time stamp of the publication of the
        # store's last key is the completion time, the loop only decides how many steps to run)
        stamp = getattr(engine.engine_, "last_publish_time", 0.0)
        store_ms.append((stamp - t0 if stamp > t0 else done_at) * 1e3)
        last_toks = toks
    store_hidden = {"proxy_step_ms_alone": round(alone, 3), "proxy_step_ms_during_store": round(median(hidden), 3),
                    "ratio": round(median(hidden) / alone, 4), "target": "<= 1.05",
                    "store_completion_ms": round(median(store_ms), 3), "store_call_returns_after_ms": round(t_call * 1e3, 3),
                    "reps": reps,
                    "note": "median over 5 non-blocking engine.store() of the 16k context (side stream): proxy decode steps "
                            "(16 GB HBM read each) that ran while encode + pinned offload were in flight vs alone"}
    # warm-prefix TTFT proxy: retrieve the 16k prefix (pinned host -> HBM -> decode), then one decode step
    ttft = []
    for r in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ret, mask = engine.retrieve(last_toks)
        proxy.step()
        torch.cuda.synchronize()
        ttft.append((time.perf_counter() - t0) * 1e3)
        assert int(mask.sum()) == CTX
        del ret
    ttft = ttft[1:]
    # the same warm prefix cut by layers (engine.retrieve_layerwise -> lmc_load_pack): the store above left the 64 blobs
    # in pinned memory as ONE layer-major pack (lmc_format.h), so the streams of a range of layers are one contiguous
    # region: one hipMemcpyAsync + one decode launch + one event per range, and the model's layers of the ranges that
    # are complete run while the later ranges are still crossing PCIe.  (With one blob per chunk the same cut costs
    # 64 x 2 short copies per range -- 14.0 ms for four ranges against 9.8 ms for the 64 whole blobs, which is why
    # lmc_load_chunks moves whole blobs and only cuts the decode; a gather KERNEL reading the pinned blobs moves
    # 41 GB/s against the DMA's 52: DESIGN.md section 5)
    from lmcache_amd.storage_backend.serde.cachegen_device import layer_ranges
    piped = {}
    for lpr in (8, 16, 32):
        ts = []
        for r in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(side):
                lw = engine.retrieve_layerwise(last_toks, layers_per_launch=lpr)
            for l0, l1 in layer_ranges(L, lpr):
                lw.wait_layer(l0)
                for l in range(l0, l1):
                    torch.mv(proxy.w[l], proxy.x, out=proxy.y)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            lw.finish()
            assert int(lw.ret_mask.sum()) == CTX
            del lw
        piped[lpr] = median(ts[1:])
    best_lpr = min(piped, key=piped.get)
    engine.close()
    # the metric's "vs cold": a cold 16k prefill proxy (dense GEMMs only) + the first decode step
    try:
        cold = ColdPrefillProxy(dev)
        cold_ms = cold.time_ms()
        cold_tflops = cold.flop / (cold_ms / 1e3) / 1e12
        del cold
    except Exception as e:
        cold_ms, cold_tflops = None, repr(e)
    pcie_ms = raw_bytes / 4.2 / 52e9 * 1e3
    ttft_proxy = {"retrieve_plus_one_step_ms": round(median(ttft), 3), "one_step_ms": round(alone, 3),
                  "ratio": round(median(ttft) / alone, 3), "target": "<= 1.05", "reps": reps,
                  "layerwise_ms": round(piped[best_lpr], 3), "layerwise_ratio": round(piped[best_lpr] / alone, 3),
                  "layerwise_layers_per_range": best_lpr,
                  "layerwise_ms_by_layers_per_range": {str(k): round(v, 3) for k, v in piped.items()},
                  "layerwise_goal_ms": round(max(pcie_ms, alone) + 1.0, 3),
                  "cold_prefill_proxy_ms": None if cold_ms is None else round(cold_ms, 2),
                  "cold_prefill_proxy_TFLOPs": cold_tflops if cold_ms is None else round(cold_tflops, 1),
                  "warm_vs_cold": None if cold_ms is None else round(piped[best_lpr] / (cold_ms + alone), 4),
                  "warm_vs_cold_note": "(warm: layer-wise retrieve of the 16k prefix from pinned host DRAM + one decode "
                                       "step) / (cold: dense-GEMM prefill proxy of the 16k prompt + one decode step); the "
                                       "reference's published figure is this ratio on its own hardware, 0.4-0.7 s warm vs "
                                       "2.4-2.9 s cold (docs/source/examples/measuring_improvements.rst:42-48,70-78)",
                  "pcie_floor_ms": round(pcie_ms, 2),
                  "note": "engine.retrieve() of the warm 16k prefix from pinned host DRAM (510 MB of blobs over one PCIe "
                          "Gen5 x16 link, ~52 GB/s measured: that transfer alone is the floor shown) + one proxy step, "
                          "over one proxy step.  retrieve_plus_one_step_ms: the whole prefix, then the step; layerwise_ms: the "
                          "layer-major pack of the pinned tier, one transfer + decode launch + event per layer range with "
                          "the proxy's layers behind each event (DESIGN.md section 5)"}
    # the same question for the tier that can meet the target: encoded chunks resident in HBM (local_device="cuda",
    # local_serde="cachegen"), retrieved layer by layer on a side stream while the model's layers run
    ttft_proxy["hbm_tier"] = ttft_hbm_tier(dev, kv, proxy, alone, meta)
    del proxy
    return {"store_hidden": store_hidden, "ttft_proxy": ttft_proxy}


def ttft_hbm_tier(dev, kv, proxy, alone, meta):
    from lmcache_amd.cache_engine import LMCacheEngine
    from lmcache_amd.config import LMCacheEngineConfig
    cfg = LMCacheEngineConfig.from_legacy(chunk_size=CHUNK, backend="cuda", local_serde="cachegen")
    engine = LMCacheEngine(cfg, meta)
    try:
        toks = torch.randint(0, 32000, (CTX,), generator=torch.Generator().manual_seed(7))
        engine.store(toks, kv)
        # the decode runs on a HIGH-priority side stream: its waves go first wherever the model's layers leave issue
        # slots, so a range is complete as early as the hardware allows (schedule the slack)
        side = torch.cuda.Stream(device=dev, priority=-1)
        whole, host_ms = [], {}
        from lmcache_amd.storage_backend.serde.cachegen_device import layer_ranges
        # range size, or a schedule of range sizes (small ranges first, the last entry repeats)
        piped = {2: [], 4: [], 8: [], (2, 6, 24): [], (1, 3, 12, 16): [], (4, 28): [], 32: []}
        for r in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ret, mask = engine.retrieve(toks)        # all layers, then the step
            proxy.step()
            torch.cuda.synchronize()
            whole.append((time.perf_counter() - t0) * 1e3)
            del ret
            for step in piped:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with torch.cuda.stream(side):        # one decode launch per range of layers on a side stream ...
                    res = engine.retrieve_layerwise(toks, layers_per_launch=step)
                host_ms.setdefault(step, []).append((time.perf_counter() - t0) * 1e3)  # hashing, look-ups, launches: the GPU idles until the first launch
                for l0, l1 in layer_ranges(L, step):  # ... the model's layers of a range wait for THAT range's KV only
                    res.wait_layer(l0)
                    for l in range(l0, l1):
                        torch.mv(proxy.w[l], proxy.x, out=proxy.y)
                torch.cuda.synchronize()
                piped[step].append((time.perf_counter() - t0) * 1e3)
                res.finish()
                assert int(res.ret_mask.sum()) == CTX
                del res
        whole = whole[1:]
        med = {k: median(v[1:]) for k, v in piped.items()}
        best = min(med, key=med.get)
        return {"retrieve_then_step_ms": round(median(whole), 3), "ratio": round(median(whole) / alone, 3),
                "layerwise_ms": round(med[best], 3), "layerwise_ratio": round(med[best] / alone, 3),
                "layers_per_launch": best if isinstance(best, int) else list(best), "layerwise_ms_by_layers_per_launch": {str(k): round(v, 3) for k, v in med.items()},
                "target": "<= 1.05", "reps": 5,
                # the host part of the BEST schedule (what a serving engine would configure), and of every schedule:
                # one lmc_decode_chunks_schedule call issues all ranges, so it grows by ~7 us per range
                "host_ms_before_the_model_can_start": round(median(host_ms[best][1:]), 3),
                "host_ms_by_layers_per_launch": {str(k): round(median(v[1:]), 3) for k, v in host_ms.items()},
                "hbm_floor_ratio": round((PROXY_BYTES + 2.66e9) / PROXY_BYTES, 3),
                "note": "encoded chunks resident in HBM (4.2x more warm context than raw KV): retrieve = decode only; "
                        "layerwise = retrieve_layerwise on a side stream, one k_decode launch per range of layers, the "
                        "model's layers of a range wait for that range's event (lmc_decode_chunks_layers).  "
                        "hbm_floor_ratio: the decode has to write 2.1 GB of KV and read 0.5 GB of blobs through the "
                        "same HBM the 16 GB weight pass saturates, so no schedule gets below it in this proxy.  The kernel "
                        "timeline (profiles/r05_ttft_timeline.md) shows the floor that binds first: decoder and GEMVs do "
                        "run side by side, but both are bound by the wave slots they hold, so what one gains the other "
                        "loses -- (decode alone + step) / step = 1.31 plus the host's 0.2-0.3 ms in front of the first launch"}
    finally:
        engine.close()


if __name__ == "__main__":
    main()
