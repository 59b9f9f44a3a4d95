"""Does the encode / decode time depend on WHERE its buffers lie?  (the unexplained spread.)

Alternating builds showed an anti-correlation inside one box: processes whose fused encode ran 0.88 ms decoded in 0.95 ms,
processes that encoded in 0.92 ms decoded in 0.89 -- per-process state, with the clocks the same.  This probe holds
several copies of every buffer in ONE process and times every combination in rotation:

    python tools/probes/placement.py        -> one JSON line: encode ms by (kv copy, blob copy), decode ms by (blob copy, out copy)

If a copy is consistently slower than its siblings (same process, same clock, same kernel, same bytes), the difference is
the placement of its pages (fragment size / channel interleave as the allocator happened to map them), not the kernel."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import bench


def main():
    from lmcache_amd import native
    from lmcache_amd.storage_backend.serde.cachegen_basics import CacheGenConfig
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ctx = native.get_context(0)
    L, H, D, CTX, CS = bench.L, bench.H, bench.D, bench.CTX, bench.CHUNK
    bins = CacheGenConfig.from_model_name(bench.MODEL).plane_bins(L)
    n = CTX // CS
    stride = native.r16(native.blob_bound(L, CS, H, D))
    NK = int(os.environ.get("LMC_PLACEMENT_COPIES", "4"))
    kvs, lays, blobs, outs, pads = [], [], [], [], []
    for k in range(NK):
        # (the same seed: identical bytes in every copy; allocations of other sizes in between, as a process has)
        kvs.append(bench.make_kv(dev, 0, "rand"))
        lays.append(native.KVLayout.from_kv_tuple(kvs[-1], "vllm"))
        pads.append(torch.empty((37 + 11 * k) << 20, dtype=torch.uint8, device=dev))
        blobs.append(torch.empty(n * stride, dtype=torch.uint8, device=dev))
        outs.append(torch.empty((L, 2, CTX, H, D), dtype=torch.bfloat16, device=dev))
    big = torch.empty((L, 2, CTX, H, D), dtype=torch.bfloat16, device=dev)   # one 2 GiB region instead of 64 tensors
    for l in range(L):
        big[l, 0].copy_(kvs[0][l][0])
        big[l, 1].copy_(kvs[0][l][1])
    lays.append(native.KVLayout.from_chunk(big, "vllm"))
    sizes = torch.zeros(n, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev)

    def enc(i, j, reps):
        return bench.time_encode(ctx, lays[i], CTX, CS, bins, blobs[j], stride, sizes, st.cuda_stream, st, reps)

    def dec(j, o, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lay = native.KVLayout.from_chunk(outs[o], "vllm")
        e0.record()
        for _ in range(reps):
            ctx.decode_chunks(blobs[j].data_ptr(), stride, n, lay, 0, CS)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for _ in range(30):
        enc(0, 0, 10)   # clock ramp
    E = {}
    for rnd in range(3):
        for i in range(NK + 1):
            for j in range(NK):
                E.setdefault(f"kv{i if i < NK else 'BIG'}->blob{j}", []).append(round(enc(i, j, 20), 4))
    for j in range(NK):
        enc(0, j, 1)    # every blob copy holds the encoded context
    Dm = {}
    for rnd in range(3):
        for j in range(NK):
            for o in range(NK):
                Dm.setdefault(f"blob{j}->out{o}", []).append(round(dec(j, o, 10), 4))
    ctx.raise_on_status("placement")
    addr = {"kv": [hex(kvs[k][0][0].data_ptr()) for k in range(NK)], "big": hex(big.data_ptr()),
            "blobs": [hex(b.data_ptr()) for b in blobs], "outs": [hex(o.data_ptr()) for o in outs]}
    print(json.dumps({"encode_ms": E, "decode_ms": Dm, "addresses": addr}))


if __name__ == "__main__":
    main()
