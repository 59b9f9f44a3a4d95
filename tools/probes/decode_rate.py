"""Decode-only timing of the bench workload (BASELINE configs[1]): encode once, time k_decode alone with HIP
events, check the round trip against the encoder's own symbols by re-encoding the decoded KV (idempotence).
    python tools/probes/decode_rate.py [--dist rand|randn|outlier] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from lmcache_amd import native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dist", default="rand")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    kv = bench.make_kv(dev, 0, args.dist)
    layout = native.KVLayout.from_kv_tuple(kv, "vllm")
    bins = bench.cachegen_bins_llama8b()
    ctx = native.get_context(0)
    nchunks = bench.CTX // bench.CHUNK
    stride = native.r16(native.blob_bound(bench.L, bench.CHUNK, bench.H, bench.D))
    blobs = torch.empty(nchunks * stride, dtype=torch.uint8, device=dev)
    sizes = torch.zeros(nchunks, dtype=torch.int32, device=dev)
    ctx.encode_chunks(layout, 0, bench.CTX, bench.CHUNK, bins, blobs.data_ptr(), stride, sizes.data_ptr())
    torch.cuda.synchronize()
    out = tuple((torch.empty_like(k), torch.empty_like(v)) for k, v in kv)
    lo = native.KVLayout.from_kv_tuple(out, "vllm")
    for _ in range(3):
        ctx.decode_chunks(blobs.data_ptr(), stride, nchunks, lo, 0, bench.CHUNK)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        ctx.decode_chunks(blobs.data_ptr(), stride, nchunks, lo, 0, bench.CHUNK)
    e1.record()
    torch.cuda.synchronize()
    ctx.raise_on_status("decode probe")
    ms = e0.elapsed_time(e1) / args.reps
    raw = sum(k.numel() + v.numel() for k, v in kv) * 2
    # decode(encode(x)) is a fixed point of encode -> decode
    blobs2 = torch.empty_like(blobs)
    sizes2 = torch.zeros_like(sizes)
    ctx.encode_chunks(lo, 0, bench.CTX, bench.CHUNK, bins, blobs2.data_ptr(), stride, sizes2.data_ptr())
    out2 = tuple((torch.empty_like(k), torch.empty_like(v)) for k, v in kv)
    ctx.decode_chunks(blobs2.data_ptr(), stride, nchunks, native.KVLayout.from_kv_tuple(out2, "vllm"), 0, bench.CHUNK)
    torch.cuda.synchronize()
    same = all(torch.equal(a, c) and torch.equal(b, d) for (a, b), (c, d) in zip(out, out2))
    # layer ranges: all planes of layers 10.. have 16 bins (<= 16 symbols), both planes of layers 0-1 have 32
    table = native.pointer_table([blobs.data_ptr() + i * stride for i in range(nchunks)], dev)
    for lb, lc in ((10, bench.L - 10), (0, 2), (2, 8)):
        for _ in range(2):
            ctx.decode_chunks_layers(table.data_ptr(), stride, nchunks, lo, 0, bench.CHUNK, lb, lc)
        e0.record()
        for _ in range(args.reps):
            ctx.decode_chunks_layers(table.data_ptr(), stride, nchunks, lo, 0, bench.CHUNK, lb, lc)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / args.reps
        print(f"  layers [{lb}, {lb + lc}): {t:.4f} ms = {t / (2 * lc) * 1e3:.2f} us per plane")
    print(f"decode {ms:.4f} ms per 16k context ({raw / ms / 1e6:.1f} GB/s of KV), dist={args.dist}, idempotent={same}")


if __name__ == "__main__":
    main()
