"""HIP hot path vs the CPU oracle and the reference-generated golden vectors.
Every call goes through the C ABI (lmcache_amd.native -> liblmc_hip.so).
Bit-exact: symbols, scales, CDF, blob bytes, decoded 16-bit tensors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def nat():
    from lmcache_amd import native
    native.lib()
    return native


@pytest.fixture(scope="module")
def ctx(nat):
    return nat.get_context(0)


def bits_np(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().cpu().view(torch.int16).numpy().view(np.uint16)


def make_kv(L, T, H, D, dtype, kind="randn", seed=0):
    g = torch.Generator().manual_seed(seed)
    if kind == "rand":
        x = torch.rand(L, 2, T, H, D, generator=g)
    elif kind == "randn":
        x = torch.randn(L, 2, T, H, D, generator=g)
    else:
        x = torch.randn(L, 2, T, H, D, generator=g) * torch.exp(1.5 * torch.randn(H * D, generator=g)).reshape(H, D)
    return x.to(dtype)


def default_bins(L):
    kb = [32 if l < max(1, L // 3) else 16 for l in range(L)]
    vb = [32 if l < 1 else 16 for l in range(L)]
    return kb + vb


def encode(nat, ctx, layout, tok_begin, tok_end, chunk_tokens, bins):
    L, H, D = layout.L, layout.H, layout.D
    stride = nat.r16(nat.blob_bound(L, chunk_tokens, H, D))
    n = (tok_end - tok_begin + chunk_tokens - 1) // chunk_tokens
    blobs = torch.zeros(n * stride, dtype=torch.uint8, device=DEV)
    sizes = torch.zeros(n, dtype=torch.int32, device=DEV)
    ctx.encode_chunks(layout, tok_begin, tok_end, chunk_tokens, bins, blobs.data_ptr(), stride, sizes.data_ptr())
    torch.cuda.synchronize()
    ctx.raise_on_status("encode")
    sz = sizes.cpu().tolist()
    host = blobs.cpu().numpy()
    return [host[i * stride:i * stride + sz[i]].tobytes() for i in range(n)], blobs, stride


# --------------------------------------------------------------------------
def test_quantize_golden(nat, ctx, oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "quant.npz"))
    bins = z["bins"].tolist()
    for dname, tdt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        for kind in ("rand", "randn", "outlier"):
            tag = f"{dname}_{kind}"
            kvb = z[f"{tag}_kv"]  # [L,2,T,C] bits; C = 256 -> H=2, D=128
            L, _, T, C = kvb.shape
            kv = torch.from_numpy(kvb.view(np.int16)).view(tdt).reshape(L, 2, T, 2, C // 2).to(DEV)
            sym, scale = ctx.quantize(nat.KVLayout.from_chunk(kv, "vllm"), 0, T, bins)
            torch.cuda.synchronize()
            assert np.array_equal(sym.cpu().numpy(), z[f"{tag}_sym"]), tag
            assert np.array_equal(scale.cpu().numpy().view(np.uint16), z[f"{tag}_scale"]), tag


def test_quantize_edge_rows_golden(nat, ctx, golden_dir):
    z = np.load(os.path.join(golden_dir, "quant_edge.npz"))
    x = torch.from_numpy(z["x"].view(np.int16)).view(torch.bfloat16)  # [2, T, 64]
    Lk, T, C = x.shape
    kv = torch.stack([x, x], dim=1).reshape(Lk, 2, T, 1, C).to(DEV)
    bins = z["bins"].tolist() * 2
    sym, scale = ctx.quantize(nat.KVLayout.from_chunk(kv, "vllm"), 0, T, bins)
    torch.cuda.synchronize()
    assert np.array_equal(sym.cpu().numpy()[:Lk], z["sym"])
    gs, s = z["scale"], scale.cpu().numpy().view(np.uint16)[:Lk]
    nan = (gs & 0x7fff) > 0x7f80
    assert np.array_equal(s[~nan], gs[~nan]) and (((s & 0x7fff) > 0x7f80) == nan).all()


@pytest.mark.parametrize("shape", [(2, 24, 1, 128), (2, 24, 2, 128), (1, 9, 4, 128), (2, 13, 8, 128), (1, 6, 5, 128),
                                   (1, 5, 16, 128), (1, 5, 32, 128), (2, 7, 3, 40)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_quantize_vs_oracle_shapes(nat, ctx, oracle, shape, dtype):
    L, T, H, D = shape
    kv = make_kv(L, T, H, D, dtype, "outlier", seed=T)
    bins = default_bins(L)
    sym, scale = ctx.quantize(nat.KVLayout.from_chunk(kv.to(DEV), "vllm"), 0, T, bins)
    torch.cuda.synchronize()
    b, code = oracle.torch_to_bits(kv.reshape(L, 2, T, H * D))
    rs, rsc = oracle.quantize(b, code, np.array(bins, np.int32))
    assert np.array_equal(sym.cpu().numpy(), rs)
    assert np.array_equal(scale.cpu().numpy().view(np.uint16), rsc)


@pytest.mark.parametrize("T", [1, 7, 16, 128, 236, 250, 256, 768])
def test_cdf_golden(nat, ctx, golden_dir, T):
    z = np.load(os.path.join(golden_dir, "cdf.npz"))
    sym = torch.from_numpy(z[f"T{T}_sym"]).to(DEV)
    cdf = ctx.calculate_cdf(sym)
    torch.cuda.synchronize()
    assert np.array_equal(cdf.cpu().numpy().view(np.uint16), z[f"T{T}_cdf"])


# --------------------------------------------------------------------------
CHUNK_SHAPES = [
    # L, T, H, D, dtype, kind
    (4, 64, 8, 128, torch.bfloat16, "randn"),
    (2, 236, 8, 128, torch.bfloat16, "rand"),   # tests/test_serde.py:87-107 ragged chunk
    (2, 1, 8, 128, torch.bfloat16, "randn"),
    (2, 3, 1, 128, torch.bfloat16, "outlier"),  # C=128: one KV head per rank (70B TP=8)
    (2, 50, 2, 128, torch.float16, "randn"),
    (1, 33, 5, 128, torch.bfloat16, "outlier"),  # C=640: partial last group
    (1, 40, 32, 128, torch.float16, "rand"),     # C=4096: BASELINE config 1 head count (four waves share a row oct)
    (1, 256, 32, 128, torch.bfloat16, "outlier"),  # ... a full chunk of it, counts model
    (1, 70, 16, 128, torch.bfloat16, "randn"),   # C=2048: two waves share a row oct
    (1, 45, 12, 128, torch.float16, "outlier"),  # C=1536: the second wave's slice is half empty
    (1, 29, 20, 128, torch.bfloat16, "rand"),    # C=2560: the fourth wave's slice is empty
    (3, 16, 3, 40, torch.float16, "randn"),      # C=120 (not a multiple of 64)
    (1, 300, 2, 64, torch.bfloat16, "randn"),    # T > 256: no 256-token sub-chunking needed
]


@pytest.mark.parametrize("shape", CHUNK_SHAPES, ids=lambda s: f"L{s[0]}T{s[1]}H{s[2]}D{s[3]}{'bf' if s[4] == torch.bfloat16 else 'fp'}")
def test_blob_and_decode_bit_exact(nat, ctx, oracle, shape):
    L, T, H, D, dtype, kind = shape
    kv = make_kv(L, T, H, D, dtype, kind, seed=L * 1000 + T)
    bins = default_bins(L)
    blobs, blob_dev, stride = encode(nat, ctx, nat.KVLayout.from_chunk(kv.to(DEV), "vllm"), 0, T, T, bins)
    b, code = oracle.torch_to_bits(kv.reshape(L, 2, T, H * D))
    ref = oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))
    assert len(blobs[0]) == len(ref)
    assert blobs[0] == ref
    hdr = nat.blob_info(blobs[0])
    assert (hdr.num_heads, hdr.head_size, hdr.ntokens, hdr.total_bytes) == (H, D, T, len(ref))
    # entropy decode only
    sym = ctx.decode_symbols(blob_dev, L, H, D, T)
    torch.cuda.synchronize()
    assert np.array_equal(sym.cpu().numpy(), oracle.decode_blob_symbols(ref))
    # fused decode, both output dtypes (vllm -> bf16, huggingface -> fp16: cachegen_decoder.py:190-200)
    for odt, ocode in ((torch.bfloat16, oracle.BF16), (torch.float16, oracle.FP16)):
        out = torch.zeros(L, 2, T, H, D, dtype=odt, device=DEV)
        ctx.decode_chunks(blob_dev.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
        torch.cuda.synchronize()
        ctx.raise_on_status("decode")
        want = oracle.decode_blob(ref, ocode)
        assert np.array_equal(bits_np(out).reshape(L, 2, T, H * D), want)


def test_multi_chunk_tail_and_kv_tuple(nat, ctx, oracle):
    """engine.store shape: per-layer (K,V) tensors, chunked with a short tail."""
    L, Ttot, H, D, cs = 3, 150, 4, 128, 64
    g = torch.Generator().manual_seed(4)
    kvt = tuple((torch.randn(Ttot, H, D, generator=g).to(torch.bfloat16).to(DEV),
                 torch.randn(Ttot, H, D, generator=g).to(torch.bfloat16).to(DEV)) for _ in range(L))
    bins = default_bins(L)
    blobs, blob_dev, stride = encode(nat, ctx, nat.KVLayout.from_kv_tuple(kvt, "vllm"), 0, Ttot, cs, bins)
    assert len(blobs) == 3
    full = torch.stack([torch.stack(p, 0) for p in kvt], 0).cpu()  # [L,2,T,H,D]
    for i, blob in enumerate(blobs):
        t0, t1 = i * cs, min(Ttot, (i + 1) * cs)
        b, code = oracle.torch_to_bits(full[:, :, t0:t1].reshape(L, 2, t1 - t0, H * D))
        assert blob == oracle.encode_blob(b, code, H, D, np.array(bins, np.int32)), f"chunk {i}"
    # decode all chunks straight into per-layer destination tensors (no torch.cat)
    outt = tuple((torch.zeros(Ttot, H, D, dtype=torch.bfloat16, device=DEV),
                  torch.zeros(Ttot, H, D, dtype=torch.bfloat16, device=DEV)) for _ in range(L))
    ctx.decode_chunks(blob_dev.data_ptr(), stride, 3, nat.KVLayout.from_kv_tuple(outt, "vllm"), 0, cs)
    torch.cuda.synchronize()
    ctx.raise_on_status("decode")
    for i, blob in enumerate(blobs):
        t0, t1 = i * cs, min(Ttot, (i + 1) * cs)
        want = oracle.decode_blob(blob, oracle.BF16)  # [L,2,Tc,C]
        for l in range(L):
            for kvi in range(2):
                got = bits_np(outt[l][kvi][t0:t1]).reshape(t1 - t0, H * D)
                assert np.array_equal(got, want[l, kvi])


def test_huggingface_layout_and_skip(nat, ctx, oracle):
    """[L,2,H,T,D] fp16 chunk (cache_engine.py:139-140) + retrieve()'s drop-first-tokens rule (:360-365)."""
    L, T, H, D = 2, 40, 4, 64
    kv = make_kv(L, T, H, D, torch.float16, "randn", seed=9)  # vllm order
    hf = kv.permute(0, 1, 3, 2, 4).contiguous().to(DEV)       # [L,2,H,T,D]
    bins = default_bins(L)
    blobs, blob_dev, stride = encode(nat, ctx, nat.KVLayout.from_chunk(hf, "huggingface"), 0, T, T, bins)
    b, code = oracle.torch_to_bits(kv.reshape(L, 2, T, H * D))
    ref = oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))
    assert blobs[0] == ref
    # a permuted (non-contiguous) vllm view of the same memory encodes identically, with no copy
    view = hf.permute(0, 1, 3, 2, 4)
    blobs2, _, _ = encode(nat, ctx, nat.KVLayout.from_chunk(view, "vllm"), 0, T, T, bins)
    assert blobs2[0] == ref
    skip = 7
    out = torch.zeros(L, 2, H, T - skip, D, dtype=torch.float16, device=DEV)
    ctx.decode_chunks(blob_dev.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "huggingface"), -skip, T)
    torch.cuda.synchronize()
    want = oracle.decode_blob(ref, oracle.FP16).reshape(L, 2, T, H, D)[:, :, skip:]
    got = bits_np(out).reshape(L, 2, H, T - skip, D).transpose(0, 1, 3, 2, 4)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("H,D,T", [(2, 100, 70), (4, 30, 256), (2, 36, 33), (8, 12, 129)])
def test_head_sizes_that_are_no_multiple_of_eight(nat, ctx, oracle, H, D, T):
    """The reference's serde takes any [.., H, D] (torch_quant_vectorized works on the merged channels,
    cachegen_encoder.py:40-61, 76-91).  Here a plane needs a multiple of 8 channels; the encoders read any head_size where
    the heads of a token row lie back to back (vllm chunk), the decoder and lmc_copy_kv write any layout, and the codec
    brings a huggingface chunk of such a head_size into a vllm chunk first (element-wise lmc_copy_kv)."""
    from lmcache_amd.storage_backend.serde.cachegen_device import get_codec
    L = 2
    kv = make_kv(L, T, H, D, torch.bfloat16, "randn", seed=40 + D)  # vllm order
    bins = default_bins(L)
    b, code = oracle.torch_to_bits(kv.reshape(L, 2, T, H * D))
    ref = oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))
    vl = nat.KVLayout.from_chunk(kv.to(DEV), "vllm")
    assert vl.vector_readable()
    blobs, blob_dev, stride = encode(nat, ctx, vl, 0, T, T, bins)
    assert blobs[0] == ref
    want = oracle.decode_blob(ref, oracle.BF16).reshape(L, 2, T, H, D)
    # decode into a huggingface chunk (heads T * D apart: no 16-byte rows) and into a vllm chunk
    out_hf = torch.zeros(L, 2, H, T, D, dtype=torch.bfloat16, device=DEV)
    ctx.decode_chunks(blob_dev.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out_hf, "huggingface"), 0, T)
    out_v = torch.zeros(L, 2, T, H, D, dtype=torch.bfloat16, device=DEV)
    ctx.decode_chunks(blob_dev.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out_v, "vllm"), 0, T)
    torch.cuda.synchronize()
    ctx.raise_on_status("decode")
    assert np.array_equal(bits_np(out_hf.permute(0, 1, 3, 2, 4)), want)
    assert np.array_equal(bits_np(out_v), want)
    # a huggingface chunk as the encoder's input: not vector-readable -> the C ABI refuses it, lmc_copy_kv moves it
    # element-wise, and the codec does that by itself
    hf = kv.permute(0, 1, 3, 2, 4).contiguous().to(DEV)
    hl = nat.KVLayout.from_chunk(hf, "huggingface")
    assert not hl.vector_readable()
    with pytest.raises(nat.NativeError):
        encode(nat, ctx, hl, 0, T, T, bins)
    back = torch.zeros(L, 2, T, H, D, dtype=torch.bfloat16, device=DEV)
    ctx.copy_kv(hl, 0, T, nat.KVLayout.from_chunk(back, "vllm"), 0)
    torch.cuda.synchronize()
    assert torch.equal(back.cpu(), kv)
    codec = get_codec(torch.cuda.current_device())
    job = codec.encode(hl, 0, T, T, bins)
    sizes = codec.sizes_of(job)
    got = job.arena[:sizes[0]].cpu().numpy().tobytes()
    assert got == ref


@pytest.mark.parametrize("layout", ["NBHD", "NHBD"])
def test_paged_gather_encode_and_scatter_decode(nat, ctx, oracle, layout):
    """vLLM paged blocks through slot_mapping (LLM_Engine.rst:91-122), incl. BASELINE's
    [num_blocks, num_heads, block_size, head_dim] layout; non-contiguous random slots (config 5)."""
    L, T, H, D, bs, nb = 2, 48, 4, 128, 16, 9
    g = torch.Generator().manual_seed(21)
    shape = (2, nb, bs, H, D) if layout == "NBHD" else (2, nb, H, bs, D)
    caches = [torch.randn(shape, generator=g).to(torch.bfloat16).to(DEV) for _ in range(L)]
    slots = torch.randperm(nb * bs, generator=g)[:T]
    lay = nat.KVLayout.paged(caches, slots, bs, layout)
    bins = default_bins(L)
    blobs, blob_dev, stride = encode(nat, ctx, lay, 0, T, T, bins)
    # dense gather on the host as the oracle's input
    dense = torch.zeros(L, 2, T, H, D, dtype=torch.bfloat16)
    for l in range(L):
        c = caches[l].cpu()
        for t, s in enumerate(slots.tolist()):
            blk, w = divmod(s, bs)
            dense[l, :, t] = c[:, blk, w] if layout == "NBHD" else c[:, blk, :, w]
    b, code = oracle.torch_to_bits(dense.reshape(L, 2, T, H * D))
    ref = oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))
    assert blobs[0] == ref
    # lossless gather into a contiguous chunk == dense
    chunk = torch.zeros(L, 2, T, H, D, dtype=torch.bfloat16, device=DEV)
    ctx.copy_kv(lay, 0, T, nat.KVLayout.from_chunk(chunk, "vllm"), 0)
    torch.cuda.synchronize()
    assert torch.equal(chunk.cpu(), dense)
    # scatter-decode into fresh paged caches at other random slots
    caches2 = [torch.zeros(shape, dtype=torch.bfloat16, device=DEV) for _ in range(L)]
    slots2 = torch.randperm(nb * bs, generator=g)[:T]
    ctx.decode_chunks(blob_dev.data_ptr(), stride, 1, nat.KVLayout.paged(caches2, slots2, bs, layout), 0, T)
    torch.cuda.synchronize()
    ctx.raise_on_status("decode")
    want = oracle.decode_blob(ref, oracle.BF16).reshape(L, 2, T, H, D)
    touched = torch.zeros(nb * bs, dtype=torch.bool)
    touched[slots2] = True
    for l in range(L):
        c = caches2[l].cpu()
        for t, s in enumerate(slots2.tolist()):
            blk, w = divmod(s, bs)
            got = c[:, blk, w] if layout == "NBHD" else c[:, blk, :, w]
            assert np.array_equal(bits_np(got), want[l, :, t])
        flat = c.reshape(2, nb * bs, H, D) if layout == "NBHD" else c.permute(0, 1, 3, 2, 4).reshape(2, nb * bs, H, D)
        assert (flat[:, ~touched] == 0).all()  # nothing written outside slot_mapping


@pytest.mark.parametrize("layout", ["NBHD", "NHBD"])
@pytest.mark.parametrize("bs,start,skip,swaps", [(16, 0, 0, 0), (16, 5, 0, 0), (8, 3, 6, 0), (12, 7, 0, 0), (32, 9, 3, 0),
                                                 (16, 2, 8, 5), (4, 1, 0, 0)])
def test_scatter_decode_into_block_ordered_slots(nat, ctx, oracle, layout, bs, start, skip, swaps):
    """The slot mapping a vLLM block manager produces: blocks anywhere, a block's tokens in order, the first block entered
    at `start`.  Runs of eight tokens on consecutive rows of one block take the decoder's block path (k_decode.h), a run
    that crosses a block boundary, swapped tokens, the trimmed first tokens (`skip`) and the last T % 8 tokens the
    one-token path -- whichever path a token takes, what lands in its slot is the oracle's value and nothing else is
    written.  Two chunks of 256 / 203 tokens (counts model)."""
    L, H, D, cs, T = 2, 2, 64, 256, 459
    g = torch.Generator().manual_seed(1000 + 17 * bs + start + skip)
    kv = torch.randn((L, 2, T, H, D), generator=g).to(torch.bfloat16)
    bins = default_bins(L)
    blobs, blob_dev, stride = encode(nat, ctx, nat.KVLayout.from_chunk(kv.to(DEV), "vllm"), 0, T, cs, bins)
    n = T - skip
    nb = (start + n + bs - 1) // bs + 4
    blocks = torch.randperm(nb, generator=g)
    pos = torch.arange(start, start + n)
    slots = blocks[pos // bs] * bs + pos % bs
    for _ in range(swaps):
        i, j = (int(x) for x in torch.randint(0, n, (2,), generator=g))
        slots[[i, j]] = slots[[j, i]]
    shape = (2, nb, bs, H, D) if layout == "NBHD" else (2, nb, H, bs, D)
    caches = [torch.zeros(shape, dtype=torch.bfloat16, device=DEV) for _ in range(L)]
    ctx.decode_chunks(blob_dev.data_ptr(), stride, len(blobs), nat.KVLayout.paged(caches, slots, bs, layout), -skip, cs)
    torch.cuda.synchronize()
    ctx.raise_on_status("decode")
    want = np.concatenate([oracle.decode_blob(b, oracle.BF16).reshape(L, 2, -1, H, D) for b in blobs], axis=2)[:, :, skip:]
    touched = torch.zeros(nb * bs, dtype=torch.bool)
    touched[slots] = True
    for l in range(L):
        c = caches[l].cpu()
        flat = c.reshape(2, nb * bs, H, D) if layout == "NBHD" else c.permute(0, 1, 3, 2, 4).reshape(2, nb * bs, H, D)
        assert np.array_equal(bits_np(flat[:, slots]), want[l])
        assert (flat[:, ~touched] == 0).all()


@pytest.mark.parametrize("case", ["T2048", "T768_fp16", "tiny_values"])
def test_long_chunks_and_denormal_scales(nat, ctx, oracle, case):
    """Chunks longer than the decoder's 512-token scale window and 512-word ring (many refills);
    chunk_size=768 as in the reference's benchmark (tests/benchmarks/test_benchmark.py:46);
    KV so small that scales and outputs are bf16 denormals (hardware bf16 conversion path)."""
    if case == "T2048":
        L, T, H, D, dt, scale = 1, 2048, 2, 64, torch.bfloat16, 1.0
    elif case == "T768_fp16":
        L, T, H, D, dt, scale = 2, 768, 8, 128, torch.float16, 1.0
    else:
        L, T, H, D, dt, scale = 2, 40, 2, 128, torch.bfloat16, 1e-39
    g = torch.Generator().manual_seed(77)
    kv = (torch.randn(L, 2, T, H, D, generator=g) * scale).to(dt)
    bins = default_bins(L)
    blobs, blob_dev, stride = encode(nat, ctx, nat.KVLayout.from_chunk(kv.to(DEV), "vllm"), 0, T, T, bins)
    b, code = oracle.torch_to_bits(kv.reshape(L, 2, T, H * D))
    ref = oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))
    assert blobs[0] == ref
    for odt, ocode in ((torch.bfloat16, oracle.BF16), (torch.float16, oracle.FP16)):
        out = torch.zeros(L, 2, T, H, D, dtype=odt, device=DEV)
        ctx.decode_chunks(blob_dev.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
        torch.cuda.synchronize()
        ctx.raise_on_status("decode")
        assert np.array_equal(bits_np(out).reshape(L, 2, T, H * D), oracle.decode_blob(ref, ocode))


@pytest.mark.parametrize("shape", [(2, 256, 8, 128, torch.bfloat16), (3, 236, 8, 128, torch.float16),
                                   (2, 100, 4, 128, torch.bfloat16), (1, 256, 4, 128, torch.float16),
                                   (2, 5, 8, 128, torch.bfloat16), (1, 129, 16, 64, torch.bfloat16),
                                   (4, 64, 8, 128, torch.bfloat16), (3, 40, 8, 64, torch.float16)])
def test_multichunk_encode_shapes_agree_with_oracle(nat, ctx, oracle, shape):
    """Multi-chunk jobs with a ragged tail and edge rows (all-zero row, inf element) over the channel counts
    that select different k_quantize instantiations (C = 512 / 1024, bf16 / fp16) produce the oracle's bytes."""
    L, T, H, D, dt = shape
    g = torch.Generator().manual_seed(T * 7 + H)
    Ttot = 6 * T + 37 if T >= 40 else T  # several chunks + short tail for the larger cases
    kv = torch.randn(L, 2, Ttot, H, D, generator=g).to(dt)
    kv[0, 0, 0] = 0                      # all-zero row -> "special" path
    if Ttot > 3:
        kv[0, 1, 2, 0, 0] = float("inf")
    bins = default_bins(L)
    lay = nat.KVLayout.from_chunk(kv.to(DEV), "vllm")
    blobs, _, _ = encode(nat, ctx, lay, 0, Ttot, T, bins)
    for i, b in enumerate(blobs):
        t0, t1 = i * T, min(Ttot, (i + 1) * T)
        bits, code = oracle.torch_to_bits(kv[:, :, t0:t1].reshape(L, 2, t1 - t0, H * D))
        assert b == oracle.encode_blob(bits, code, H, D, np.array(bins, np.int32)), f"chunk {i}"


def test_encode_repeated_on_changing_paged_data(nat, ctx, oracle):
    """The encode workspace (symbols, scratch streams, look-back granules) is re-used by every call: encode
    repeatedly from a paged source whose data and block table change (a stale read would show as a wrong blob)."""
    L, T, H, D, nchunks, bs = 4, 64, 8, 128, 7, 16
    ntok = nchunks * T - 9
    bins = default_bins(L)
    nblocks = (ntok + bs - 1) // bs + 3
    g = torch.Generator().manual_seed(77)
    if True:
        for it in range(4):
            kv = torch.randn(L, 2, ntok, H, D, generator=g).to(torch.bfloat16)
            perm = torch.randperm(nblocks, generator=g)
            slot = torch.empty(ntok, dtype=torch.int64)
            for t in range(ntok):
                slot[t] = perm[t // bs] * bs + t % bs
            paged = torch.zeros(L, 2, nblocks, bs, H, D, dtype=torch.bfloat16)
            for t in range(ntok):
                paged[:, :, slot[t] // bs, slot[t] % bs] = kv[:, :, t]
            caches = [paged[l].contiguous().to(DEV) for l in range(L)]
            lay = nat.KVLayout.paged(caches, slot, bs, "NBHD")
            blobs, _, _ = encode(nat, ctx, lay, 0, ntok, T, bins)
            for i, b in enumerate(blobs):
                t0, t1 = i * T, min(ntok, (i + 1) * T)
                bits, code = oracle.torch_to_bits(kv[:, :, t0:t1].reshape(L, 2, t1 - t0, H * D))
                assert b == oracle.encode_blob(bits, code, H, D, np.array(bins, np.int32)), f"iter {it} chunk {i}"


def test_llama70b_tp8_rank_shape(nat, ctx, oracle):
    """BASELINE config 4, one rank of Llama-3-70B TP=8: 80 layers (160 planes), one KV head per rank
    (C = 128), per-layer tensors, two chunks with a short tail."""
    from lmcache_amd.storage_backend.serde.cachegen_basics import CacheGenConfig
    L, H, D, Ttot, cs = 80, 1, 128, 300, 256
    bins = CacheGenConfig.from_model_name("Llama-3-70B").plane_bins(L)
    g = torch.Generator().manual_seed(70)
    kvt = tuple((torch.randn(Ttot, H, D, generator=g).to(torch.bfloat16).to(DEV),
                 torch.randn(Ttot, H, D, generator=g).to(torch.bfloat16).to(DEV)) for _ in range(L))
    blobs, blob_dev, stride = encode(nat, ctx, nat.KVLayout.from_kv_tuple(kvt, "vllm"), 0, Ttot, cs, bins)
    full = torch.stack([torch.stack(p, 0) for p in kvt], 0).cpu()
    for i, blob in enumerate(blobs):
        t0, t1 = i * cs, min(Ttot, (i + 1) * cs)
        b, code = oracle.torch_to_bits(full[:, :, t0:t1].reshape(L, 2, t1 - t0, H * D))
        assert blob == oracle.encode_blob(b, code, H, D, np.array(bins, np.int32)), f"chunk {i}"
    outt = tuple((torch.zeros_like(k), torch.zeros_like(v)) for k, v in kvt)
    ctx.decode_chunks(blob_dev.data_ptr(), stride, len(blobs), nat.KVLayout.from_kv_tuple(outt, "vllm"), 0, cs)
    torch.cuda.synchronize()
    ctx.raise_on_status("decode")
    want0 = oracle.decode_blob(blobs[0], oracle.BF16)
    assert np.array_equal(bits_np(outt[79][1][:cs]).reshape(cs, H * D), want0[79, 1])


def test_encodes_on_two_streams_do_not_wait_for_each_other(nat, ctx, oracle):
    """lmc_ctx holds two encode workspaces: a job on stream B must not queue behind a job on stream A (the reference's
    model is a worker thread putting while the engine thread puts, local_backend.py:41-45, 72-80).  Stream A is kept busy
    by a spin kernel with an encode queued BEHIND it; the encode on stream B has to finish long before A's spin ends --
    with one workspace it would wait for A's job, i.e. for the spin.  Bytes of both jobs against the oracle."""
    L, T, H, D = 2, 256, 8, 128
    bins = default_bins(L)
    kva, kvb = make_kv(L, T, H, D, torch.bfloat16, "randn", 5), make_kv(L, T, H, D, torch.bfloat16, "rand", 6)
    da, db = kva.to(DEV), kvb.to(DEV)
    stride = nat.r16(nat.blob_bound(L, T, H, D))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ba, bb = torch.zeros(stride, dtype=torch.uint8, device=DEV), torch.zeros(stride, dtype=torch.uint8, device=DEV)
    za, zb = torch.zeros(1, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    # warm both workspaces (allocation of the second one synchronises nothing, but keep it out of the timing)
    ctx.encode_chunks(nat.KVLayout.from_chunk(da, "vllm"), 0, T, T, bins, ba.data_ptr(), stride, za.data_ptr(), stream=sa.cuda_stream)
    ctx.encode_chunks(nat.KVLayout.from_chunk(db, "vllm"), 0, T, T, bins, bb.data_ptr(), stride, zb.data_ptr(), stream=sb.cuda_stream)
    torch.cuda.synchronize()
    spin_cycles = 400_000_000  # torch.cuda._sleep counts shader clocks: ~0.2 s
    a_done, b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = torch.cuda.Event(enable_timing=True)
    t0.record(sa)
    with torch.cuda.stream(sa):
        torch.cuda._sleep(spin_cycles)
    ctx.encode_chunks(nat.KVLayout.from_chunk(da, "vllm"), 0, T, T, bins, ba.data_ptr(), stride, za.data_ptr(), stream=sa.cuda_stream)
    a_done.record(sa)
    b0.record(sb)
    ctx.encode_chunks(nat.KVLayout.from_chunk(db, "vllm"), 0, T, T, bins, bb.data_ptr(), stride, zb.data_ptr(), stream=sb.cuda_stream)
    b1.record(sb)
    torch.cuda.synchronize()
    ctx.raise_on_status("two streams")
    a_ms, b_ms = t0.elapsed_time(a_done), b0.elapsed_time(b1)
    assert a_ms > 20.0, f"the spin did not hold stream A ({a_ms:.1f} ms)"
    assert b_ms < a_ms / 4, f"the encode on stream B took {b_ms:.1f} ms while stream A was held for {a_ms:.1f} ms"
    for kv, blob, sz in ((kva, ba, za), (kvb, bb, zb)):
        b, code = oracle.torch_to_bits(kv.reshape(L, 2, T, H * D))
        assert blob[:int(sz.item())].cpu().numpy().tobytes() == oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))


def test_concurrent_calls_share_one_context(nat, ctx, oracle):
    """Re-entrancy (SURVEY.md 8b threading): two host threads, each on its own stream, encode and decode
    different KV through the SAME context; the workspace is ordered by events, results must not mix."""
    import threading
    L, T, H, D = 2, 200, 8, 128
    bins = default_bins(L)
    results, errors = {}, []

    def worker(seed):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            kv = make_kv(L, T, H, D, torch.bfloat16, "randn", seed)
            kvd = kv.to(DEV)
            torch.cuda.current_stream().synchronize()  # the upload ran on this thread's current stream
            stride = nat.r16(nat.blob_bound(L, T, H, D))
            for it in range(6):
                with torch.cuda.stream(st):  # the fills run on the stream that uses the buffers
                    blobs = torch.zeros(stride, dtype=torch.uint8, device=DEV)
                    sizes = torch.zeros(1, dtype=torch.int32, device=DEV)
                    out = torch.zeros_like(kvd)
                    ctx.encode_chunks(nat.KVLayout.from_chunk(kvd, "vllm"), 0, T, T, bins, blobs.data_ptr(), stride,
                                      sizes.data_ptr(), stream=st.cuda_stream)
                    ctx.decode_chunks(blobs.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T,
                                      stream=st.cuda_stream)
                st.synchronize()
                results[(seed, it)] = (blobs[:int(sizes.item())].cpu().numpy().tobytes(), bits_np(out))
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(s,)) for s in (11, 22, 33)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert ctx.status(clear=True) == 0
    for seed in (11, 22, 33):
        kv = make_kv(L, T, H, D, torch.bfloat16, "randn", seed)
        b, code = oracle.torch_to_bits(kv.reshape(L, 2, T, H * D))
        ref = oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))
        want = oracle.decode_blob(ref, oracle.BF16)
        for it in range(6):
            blob, dec = results[(seed, it)]
            assert blob == ref, (seed, it)
            assert np.array_equal(dec.reshape(L, 2, T, H * D), want), (seed, it)


def test_full_16k_context_roundtrip_equals_reference_formula(nat, ctx):
    """BASELINE config 2 at FULL size (32 layers x 16384 tokens x 8 x 128 bf16 = 2 GiB, 64 chunks): the
    size-independent property decode(encode(x)) == do_dequantize(torch_quant_vectorized(x)).to(bf16), with the
    right-hand side evaluated plane by plane by torch ON THE CPU exactly as the reference writes it
    (cachegen_encoder.py:54-59, cachegen_decoder.py:31-35,190-193; eager ops, so mul and add round separately).
    The same formula run by torch's own GPU kernels is checked against it on two layers (informational)."""
    from lmcache_amd.storage_backend.serde.cachegen_basics import CacheGenConfig
    L, Ttot, H, D, cs = 32, 16384, 8, 128, 256
    cfg = CacheGenConfig.from_model_name("meta-llama/Llama-3.1-8B-Instruct")
    bins = cfg.plane_bins(L)
    g = torch.Generator(device=DEV).manual_seed(123)
    kv = tuple((torch.randn((Ttot, H, D), generator=g, device=DEV).to(torch.bfloat16),
                (torch.rand((Ttot, H, D), generator=g, device=DEV) * 3 - 1).to(torch.bfloat16)) for _ in range(L))
    n = Ttot // cs
    stride = nat.r16(nat.blob_bound(L, cs, H, D))
    blobs = torch.empty(n * stride, dtype=torch.uint8, device=DEV)
    sizes = torch.zeros(n, dtype=torch.int32, device=DEV)
    ctx.encode_chunks(nat.KVLayout.from_kv_tuple(kv, "vllm"), 0, Ttot, cs, bins, blobs.data_ptr(), stride, sizes.data_ptr())
    out = tuple((torch.empty_like(k), torch.empty_like(v)) for k, v in kv)
    ctx.decode_chunks(blobs.data_ptr(), stride, n, nat.KVLayout.from_kv_tuple(out, "vllm"), 0, cs)
    torch.cuda.synchronize()
    ctx.raise_on_status("full context")
    sz = sizes.cpu()
    assert int(sz.min()) > 0 and 3.0 < (L * 2 * Ttot * H * D * 2) / int(sz.sum()) < 4.5

    def reference(x, nbins):  # x [T, H, D] bf16
        xc = x.reshape(x.shape[0], H * D)
        # [1, 1] like the reference's (bins // 2 - 1)[:, None, None]: a dim>0 fp32 tensor, so MAX / max1 is fp32
        # (a 0-dim tensor would not take part in type promotion and the division would happen in bf16)
        MAX = torch.full((1, 1), float(nbins // 2 - 1), device=x.device)
        max1 = torch.amax(torch.abs(xc), dim=-1, keepdim=True)
        factor = MAX / max1
        xq = torch.round(xc * factor + MAX).to(torch.int8)
        t = xq.to(torch.uint8).float()
        t = t - MAX
        t = t / MAX
        t = t * max1
        return t.to(torch.bfloat16).reshape(x.shape)

    gpu_torch_mismatch = 0
    for l in range(L):
        for kvi in range(2):
            want = reference(kv[l][kvi].cpu(), bins[kvi * L + l])
            got = out[l][kvi].cpu()
            assert torch.equal(got, want), (l, kvi)
            if l < 2:  # informational: the same formula run by torch's GPU kernels
                gpu_torch_mismatch += int((reference(kv[l][kvi], bins[kvi * L + l]).cpu() != want).sum())
    print(f"torch-GPU vs torch-CPU evaluation of the reference formula: {gpu_torch_mismatch} mismatches in "
          f"{4 * Ttot * H * D} elements")


def test_corrupt_blob_is_flagged(nat, ctx):
    L, T, H, D = 1, 32, 1, 128
    kv = make_kv(L, T, H, D, torch.bfloat16, "randn", 1).to(DEV)
    blobs, blob_dev, stride = encode(nat, ctx, nat.KVLayout.from_chunk(kv, "vllm"), 0, T, T, [32, 16])
    out = torch.zeros_like(kv)
    bad = blob_dev.clone()
    bad[0] ^= 0xff  # magic
    ctx.decode_chunks(bad.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
    torch.cuda.synchronize()
    assert ctx.status(clear=True) & 2
    hdr = nat.blob_info(blobs[0])
    bad = blob_dev.clone()
    bad[hdr.off_streams + 40] ^= 0x5a  # a stream word
    ctx.decode_chunks(bad.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
    torch.cuda.synchronize()
    assert ctx.status(clear=True) & 4
    # a header that claims more bytes than its slot holds, or more tokens than the chunk spacing: refused
    # before any section is read (every section offset is a function of validated header fields)
    import struct
    bad = blob_dev.clone()
    bad[68:72] = torch.tensor(list(struct.pack("<I", stride + 16)), dtype=torch.uint8, device=DEV)  # total_bytes
    ctx.decode_chunks(bad.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
    torch.cuda.synchronize()
    assert ctx.status(clear=True) & 2
    ctx.decode_chunks(blob_dev.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T - 1)
    torch.cuda.synchronize()
    assert ctx.status(clear=True) & 2
    # garbage in a stream's head (widths, count planes) or in the stream directory: flagged, nothing hangs or faults
    g = torch.Generator().manual_seed(3)
    for n in (8, 32, 600):  # the widths alone; widths and the first planes; the whole head and the first words
        bad = blob_dev.clone()
        bad[hdr.off_streams:hdr.off_streams + n] = torch.randint(0, 256, (n,), generator=g, dtype=torch.uint8).to(DEV)
        ctx.decode_chunks(bad.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
        torch.cuda.synchronize()
        assert ctx.status(clear=True) & 4
    bad = blob_dev.clone()
    bad[hdr.off_streams:hdr.off_streams + 8] = torch.tensor([3] * 8, dtype=torch.uint8, device=DEV)  # legal widths, wrong ones
    ctx.decode_chunks(bad.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
    torch.cuda.synchronize()
    assert ctx.status(clear=True) & 4
    bad = blob_dev.clone()
    bad[hdr.off_gdir:hdr.off_gdir + 8] = torch.tensor([0xff] * 8, dtype=torch.uint8, device=DEV)
    ctx.decode_chunks(bad.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
    torch.cuda.synchronize()
    assert ctx.status(clear=True) & 4
    # a flipped scale would rescale a whole token row without the coder noticing: the per-plane checksums do
    for where in (hdr.off_scales + 2, hdr.off_scales + 2 * T + 31, hdr.off_scsum + 5):
        bad = blob_dev.clone()
        bad[where] ^= 0x01
        ctx.decode_chunks(bad.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
        torch.cuda.synchronize()
        assert ctx.status(clear=True) & 16, where
    # and the intact blob still decodes
    ctx.decode_chunks(blob_dev.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
    torch.cuda.synchronize()
    assert ctx.status(clear=True) == 0


@pytest.mark.parametrize("T", [256, 100])
def test_fuzzed_blobs_never_fault_and_are_flagged(nat, ctx, T):
    """Random damage anywhere behind the header of a multi-block blob (256 tokens: every stream refills its LDS
    ring several times; 32- and 16-bin planes, a ragged last channel group): the decoder stays inside its
    buffers, finishes, and either flags the blob or -- when only padding was hit -- decodes it unchanged."""
    # (T = 100: the decoder's prologue scales the head's counts to a sum of 256 -- round 5 -- and a damaged head must not
    # push a model count past 256)
    L, H, D = 2, 3, 40  # C = 120: one full group stream would be 64 channels, the second has 56
    kv = make_kv(L, T, H, D, torch.bfloat16, "randn", 5).to(DEV)
    blobs, blob_dev, stride = encode(nat, ctx, nat.KVLayout.from_chunk(kv, "vllm"), 0, T, T, [32, 16, 17, 20])
    hdr = nat.blob_info(blobs[0])
    clean = torch.zeros_like(kv)
    ctx.decode_chunks(blob_dev.data_ptr(), stride, 1, nat.KVLayout.from_chunk(clean, "vllm"), 0, T)
    torch.cuda.synchronize()
    assert ctx.status(clear=True) == 0
    g = torch.Generator().manual_seed(11)
    total = len(blobs[0])
    flagged = 0
    for it in range(40):
        bad = blob_dev.clone()
        lo = (hdr.off_scales, hdr.off_gdir, hdr.off_streams)[it % 3]
        npos = int(torch.randint(1, 17, (1,), generator=g))
        pos = torch.randint(lo, total, (npos,), generator=g)
        if it % 5 == 4:  # a run of garbage instead of single bytes
            a = int(pos[0])
            n = min(total - a, 4096)
            bad[a:a + n] = torch.randint(0, 256, (n,), generator=g, dtype=torch.uint8).to(DEV)
        else:
            bad[pos.to(DEV)] ^= torch.randint(1, 256, (npos,), generator=g, dtype=torch.uint8).to(DEV)
        out = torch.zeros_like(kv)
        ctx.decode_chunks(bad.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
        torch.cuda.synchronize()
        st = ctx.status(clear=True)
        if st:
            flagged += 1
        else:
            assert torch.equal(out, clean), it
    assert flagged >= 30


def test_full_size_llama8b_chunk_vs_oracle(nat, ctx, oracle):
    """BASELINE config 2, one full chunk: L=32, 8 KV heads x 128, T=256 bf16 (32 MiB)."""
    L, T, H, D = 32, 256, 8, 128
    bins, nl = oracle.cachegen_bins("meta-llama/Llama-3.1-8B-Instruct")
    assert nl == L
    kv = make_kv(L, T, H, D, torch.bfloat16, "rand", seed=0)  # the reference's test distribution
    blobs, blob_dev, stride = encode(nat, ctx, nat.KVLayout.from_chunk(kv.to(DEV), "vllm"), 0, T, T, bins.tolist())
    b, code = oracle.torch_to_bits(kv.reshape(L, 2, T, H * D))
    ref = oracle.encode_blob(b, code, H, D, bins)
    assert blobs[0] == ref
    ratio = kv.numel() * 2 / len(ref)
    # SURVEY.md 8a/a11 measured ~3.0x for the reference's container on uniform data; ours stores one byte per
    # symbol count (31 / 15 per channel) instead of 33 u16 CDF entries (lmc_format.h): the same streams at ~4.2x
    assert 3.8 < ratio < 4.7, ratio
    out = torch.zeros(L, 2, T, H, D, dtype=torch.bfloat16, device=DEV)
    ctx.decode_chunks(blob_dev.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
    torch.cuda.synchronize()
    assert np.array_equal(bits_np(out).reshape(L, 2, T, H * D), oracle.decode_blob(ref, oracle.BF16))


@pytest.mark.parametrize("T", [256, 255, 300])
def test_saturated_and_wide_symbol_counts(nat, ctx, oracle, T):
    """Format v3's count section on the GPU: channels whose every token carries the same symbol (count == T: at
    T = 256 the byte saturates and the decoder restores it), and u16 counts for chunks longer than 256 tokens."""
    L, H, D = 2, 8, 128
    g = torch.Generator().manual_seed(1000 + T)
    kv = torch.randn(L, 2, T, H, D, generator=g).to(torch.bfloat16)
    kv[:, :, :, 0, 5] = 0                                        # constant channels in every plane
    kv[0, 0, :, 3, 7] = kv[0, 0].abs().amax(dim=(-1, -2))        # always the row max
    kv[1, 1, :, 7, 127] = -kv[1, 1].abs().amax(dim=(-1, -2))     # always the row min (symbol 0)
    bins = default_bins(L)
    blobs, blob_dev, stride = encode(nat, ctx, nat.KVLayout.from_chunk(kv.to(DEV), "vllm"), 0, T, T, bins)
    b, code = oracle.torch_to_bits(kv.reshape(L, 2, T, H * D))
    ref = oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))
    assert blobs[0] == ref
    out = torch.zeros(L, 2, T, H, D, dtype=torch.bfloat16, device=DEV)
    ctx.decode_chunks(blob_dev.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
    torch.cuda.synchronize()
    ctx.raise_on_status("decode")
    assert np.array_equal(bits_np(out).reshape(L, 2, T, H * D), oracle.decode_blob(ref, oracle.BF16))


def test_unusual_bins_and_ragged_channel_counts(nat, ctx, oracle):
    """Bins other than the CacheGen tables' 16 / 32 (4 .. 32, odd ones too: both workspace formats, both decoder
    search depths, count rows of every length) on channel counts that leave idle lanes in the last group, chunk
    lengths on both sides of the one-byte / two-byte count boundary; blob and decode bit-exact against the oracle."""
    rng = np.random.default_rng(11)
    cases = [(1, 64, 3, 40, [4, 32]), (2, 255, 1, 72, [6, 17, 18, 30]), (1, 257, 2, 64, [20, 8]),
             (2, 300, 5, 24, [31, 5, 16, 28]), (1, 256, 1, 8, [32, 4])]
    for L, T, H, D, bins in cases:
        x = rng.standard_normal((L, 2, T, H, D)).astype(np.float32)
        x[:, :, :, :, ::5] = np.round(x[:, :, :, :, ::5])   # peaky channels
        kv = torch.from_numpy(x).to(torch.bfloat16)
        blobs, blob_dev, stride = encode(nat, ctx, nat.KVLayout.from_chunk(kv.to(DEV), "vllm"), 0, T, T, bins)
        b, code = oracle.torch_to_bits(kv.reshape(L, 2, T, H * D))
        ref = oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))
        assert blobs[0] == ref, (L, T, H, D, bins)
        out = torch.zeros(L, 2, T, H, D, dtype=torch.bfloat16, device=DEV)
        ctx.decode_chunks(blob_dev.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
        torch.cuda.synchronize()
        ctx.raise_on_status("decode")
        assert np.array_equal(bits_np(out).reshape(L, 2, T, H * D), oracle.decode_blob(ref, oracle.BF16)), (L, T, H, D, bins)


# --------------------------------------------------------------------------
# The two launch paths of lmc_encode_chunks (include/lmc_hip.h: lmc_ctx_set_encode_path).  Small jobs take
# k_quantize + k_cdf_encode under the default setting, so the fused kernel is forced here; what AUTO picks at
# full size is covered by test_full_16k_context_roundtrip_equals_reference_formula and the bench.
def encode_with_path(nat, ctx, path, *args):
    ctx.set_encode_path(path)
    try:
        return encode(nat, ctx, *args)
    finally:
        ctx.set_encode_path("auto")


FUSED_SHAPES = [
    # L, T total, chunk, H, D, dtype, kind.  The fused kernel takes the 256-token chunks of a job (the counts model);
    # a ragged last chunk and every other chunk length go through k_quantize + k_cdf_encode whatever the setting.
    (4, 256, 256, 8, 128, torch.bfloat16, "randn"),   # C = 1024, one chunk
    (2, 600, 256, 8, 128, torch.bfloat16, "rand"),    # two fused chunks + a ragged tail of 88 tokens (two-kernel path)
    (2, 512, 256, 8, 128, torch.float16, "outlier"),  # fp16
    (1, 256, 256, 5, 128, torch.bfloat16, "outlier"), # C = 640: partial second channel run
    (1, 256, 256, 5, 72, torch.bfloat16, "randn"),    # C = 360: partial last group (idle lanes in the coder)
    (3, 768, 256, 4, 128, torch.float16, "randn"),    # C = 512: one channel run per lane
    (2, 256, 256, 3, 128, torch.bfloat16, "randn"),   # C = 384
    # narrow planes: several planes per work item, 16 / 32 lanes per quantise task
    (10, 512, 256, 1, 128, torch.bfloat16, "randn"),  # C = 128 (Llama-3-70B TP=8 rank): 20 planes = items of 8, 8 and 4 planes
    (3, 256, 256, 1, 64, torch.float16, "outlier"),   # C = 64: one group stream per plane, 6 planes in one item
    (5, 512, 256, 2, 128, torch.bfloat16, "rand"),    # C = 256: items of 4, 4 and 2 planes
    (2, 256, 256, 1, 200, torch.bfloat16, "randn"),   # C = 200: the 32-lane task with a partial last group
    # planes of more than LMC_FUSED_MAX_CHANNELS = 1024 channels: two kernels whatever the setting (the fused form built
    # for them in round 4 was removed in round 5) -- the wide k_quantize variants (2 / 4 waves per row oct) run here
    (1, 512, 256, 32, 128, torch.float16, "rand"),    # C = 4096 (BASELINE configs[0])
    (2, 256, 256, 16, 128, torch.bfloat16, "randn"),  # C = 2048
    (1, 256, 256, 25, 128, torch.bfloat16, "outlier"),# C = 3200: partial channel runs in the last wave of a row
    (1, 256, 256, 9, 136, torch.bfloat16, "randn"),   # C = 1224: an almost empty second slice
    # chunk lengths below 256 (round 5: the counts model scaled to a sum of 256; fused from 32 tokens on)
    (2, 236, 236, 8, 128, torch.float16, "outlier"),  # tests/test_serde.py:87-107 chunk length: 7 whole 32-token blocks + 12 tokens
    (2, 472, 236, 8, 128, torch.bfloat16, "rand"),    # two of them
    (2, 300, 128, 8, 128, torch.float16, "randn"),    # chunk_size 128: two fused chunks + a tail of 44 tokens (counts-only launch)
    (1, 165, 32, 4, 128, torch.bfloat16, "randn"),    # one block per chunk, tail of 5 tokens
    (2, 200, 40, 3, 128, torch.bfloat16, "outlier"),  # C = 384, chunks of 40 = one block + 8 tokens
    (3, 250, 100, 1, 128, torch.bfloat16, "randn"),   # narrow planes: 13 row octs = 3 full oct groups + 1 partial, tail of 50
    (2, 255, 255, 2, 128, torch.bfloat16, "rand"),    # the longest scaled chunk
    (1, 300, 300, 8, 128, torch.bfloat16, "randn"),   # T > 256 in one chunk (two-byte counts): CDF16, not a fused geometry
    (2, 9, 4, 3, 128, torch.bfloat16, "randn"),       # chunks shorter than a row oct (counts-only launch); the 1-token tail is CDF16
    (1, 67, 33, 1, 72, torch.bfloat16, "randn"),      # P G = 4: streams do not fill 8-wave workgroups -> general launch, counts model per stream
    # round 6: the histogram taken in phase A -- items of TWO narrow planes (C = 256 / 192: GL = 32) with different counter
    # formats side by side (default_bins: a 32-bin and a 16-bin plane per item), and a partial last oct
    (3, 768, 256, 2, 128, torch.bfloat16, "rand"),
    (2, 512, 256, 3, 64, torch.float16, "outlier"),
    (2, 300, 100, 2, 128, torch.bfloat16, "randn"),
]


@pytest.mark.parametrize("shape", FUSED_SHAPES, ids=lambda s: f"L{s[0]}T{s[1]}c{s[2]}H{s[3]}D{s[4]}{'bf' if s[5] == torch.bfloat16 else 'fp'}")
def test_fused_encode_equals_two_kernel_encode_and_oracle(nat, ctx, oracle, shape):
    L, Ttot, cs, H, D, dtype, kind = shape
    kv = make_kv(L, Ttot, H, D, dtype, kind, seed=7 * L + Ttot)
    bins = default_bins(L)
    lay = nat.KVLayout.from_chunk(kv.to(DEV), "vllm")
    fused, _, _ = encode_with_path(nat, ctx, "fused", lay, 0, Ttot, cs, bins)
    two, _, _ = encode_with_path(nat, ctx, "two_kernels", lay, 0, Ttot, cs, bins)
    assert len(fused) == len(two) == (Ttot + cs - 1) // cs
    for i, (bf, bt) in enumerate(zip(fused, two)):
        t0, t1 = i * cs, min(Ttot, (i + 1) * cs)
        b, code = oracle.torch_to_bits(kv[:, :, t0:t1].reshape(L, 2, t1 - t0, H * D))
        ref = oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))
        assert bt == ref, f"two-kernel path, chunk {i}"
        assert bf == ref, f"fused path, chunk {i}"


def test_random_geometries_and_chunk_lengths_both_paths_equal_the_oracle(nat, ctx, oracle):
    """A seeded sweep over what round 5 opened up: chunk lengths 2 .. 256 (and a few above), ragged tails of any length,
    plane widths 8 .. 4096 with partial channel groups, bins 4 .. 32 per plane, both dtypes, both layouts, three data kinds,
    the job a token range inside a larger cache -- every blob of both launch paths byte-equal to the oracle's, and the decode of the job equal to the oracle's decode.
    LMC_FUZZ_CASES (default 24) sets how many geometries are drawn, LMC_FUZZ_SEED the generator's seed.  (Round 5: the
    first run of this sweep found the fp16 output rounded once instead of twice -- a fused v_fma_mixlo_f16 -- at bin
    counts the fixed parity cases do not use.)"""
    rnd = np.random.default_rng(int(os.environ.get("LMC_FUZZ_SEED", "2026")))
    ncase = int(os.environ.get("LMC_FUZZ_CASES", "24"))
    for case in range(ncase):
        L = int(rnd.integers(1, 4))
        D = int(rnd.choice([8, 40, 64, 72, 128]))
        cmax = 4096 if rnd.integers(0, 5) == 0 else 1024  # (one geometry in five: planes wider than the fused kernel takes)
        H = int(rnd.integers(1, max(2, cmax // D) + 1))
        cs = int(rnd.choice([2, 3, 7, 8, 31, 32, 33, 40, 64, 100, 128, 200, 236, 255, 256, 256, 300]))
        nchunk = int(rnd.integers(1, 4))
        tail = int(rnd.integers(0, cs))
        Ttot = max(1, (nchunk - 1) * cs + (tail if tail else cs))
        dtype = torch.bfloat16 if rnd.integers(0, 2) else torch.float16
        kind = ["rand", "randn", "outlier"][int(rnd.integers(0, 3))]
        bins = [int(b) for b in rnd.integers(4, 33, 2 * L)]
        kv = make_kv(L, Ttot, H, D, dtype, kind, seed=1000 + case)
        if rnd.integers(0, 3) == 0:
            kv[:, :, :, 0, 0] = 1.0  # a constant channel in every plane: one symbol holds every token
        special = int(rnd.integers(0, 6))  # the quantiser's special rows (zero / inf / NaN max) at a random place
        if special < 3:
            l, kvi, t = int(rnd.integers(0, L)), int(rnd.integers(0, 2)), int(rnd.integers(0, Ttot))
            if special == 0:
                kv[l, kvi, t] = 0.0
            else:
                kv[l, kvi, t, int(rnd.integers(0, H)), int(rnd.integers(0, D))] = float("inf") if special == 1 else float("nan")
        # the job is a token range INSIDE a larger cache (tok_begin / dst_tok0 > 0, tokens behind it), in either layout
        front, back = (int(v) for v in rnd.integers(0, 6, 2))
        fmt = "vllm" if rnd.integers(0, 2) else "huggingface"
        big = torch.full((L, 2, front + Ttot + back, H, D), 3.0, dtype=dtype)
        big[:, :, front:front + Ttot] = kv
        src = big.to(DEV) if fmt == "vllm" else big.permute(0, 1, 3, 2, 4).contiguous().to(DEV)
        lay = nat.KVLayout.from_chunk(src, fmt)
        tag = f"case {case} (special {special}): L{L} T{Ttot} cs{cs} H{H} D{D} {dtype} {kind} {fmt} +{front}/+{back} bins{bins}"
        fused, blob_dev, stride = encode_with_path(nat, ctx, "fused", lay, front, front + Ttot, cs, bins)
        two, _, _ = encode_with_path(nat, ctx, "two_kernels", lay, front, front + Ttot, cs, bins)
        n = (Ttot + cs - 1) // cs
        assert len(fused) == len(two) == n, tag
        out = torch.zeros_like(src)
        ctx.decode_chunks(blob_dev.data_ptr(), stride, n, nat.KVLayout.from_chunk(out, fmt), front, cs)
        torch.cuda.synchronize()
        assert ctx.status(clear=True) == 0, tag
        out = out if fmt == "vllm" else out.permute(0, 1, 3, 2, 4)
        assert not bits_np(out[:, :, :front].contiguous()).any() and not bits_np(out[:, :, front + Ttot:].contiguous()).any(), \
            f"{tag}: the decode wrote outside its token range"
        out = out[:, :, front:front + Ttot]
        code = oracle.BF16 if dtype == torch.bfloat16 else oracle.FP16
        for i in range(n):
            t0, t1 = i * cs, min(Ttot, (i + 1) * cs)
            b, c = oracle.torch_to_bits(kv[:, :, t0:t1].reshape(L, 2, t1 - t0, H * D))
            ref = oracle.encode_blob(b, c, H, D, np.array(bins, np.int32))
            assert two[i] == ref, f"{tag}: two-kernel path, chunk {i}"
            assert fused[i] == ref, f"{tag}: fused setting, chunk {i}"
            want = oracle.decode_blob(ref, code)
            got = bits_np(out[:, :, t0:t1].contiguous()).reshape(L, 2, t1 - t0, H * D)
            # (NaN rows -- a NaN or 0 x inf product -- are NaN on both sides, whatever the payload bits of each cast)
            inf_bits = 0x7f80 if dtype == torch.bfloat16 else 0x7c00
            nan_w, nan_g = (want & 0x7fff) > inf_bits, (got & 0x7fff) > inf_bits
            assert np.array_equal(nan_w, nan_g) and np.array_equal(got[~nan_w], want[~nan_w]), f"{tag}: decode, chunk {i}"


def test_fused_encode_special_rows_and_repeated_jobs(nat, ctx, oracle):
    """Zero / inf / NaN / denormal-scale rows through the fused kernel's own quantise stage, and back-to-back jobs
    into the same workspace: the look-back granules of job n must not be taken for job n + 1's (epoch tags)."""
    L, T, H, D, cs = 2, 768, 8, 128, 256
    bins = default_bins(L)
    g = torch.Generator().manual_seed(77)
    for rep in range(3):
        kv = torch.randn(L, 2, T, H, D, generator=g)
        kv[:, :, 5 + rep] = 0.0
        kv[0, 0, 17, 3, 11] = float("inf")
        kv[1, 1, 40, 0, 0] = float("nan")
        kv[:, :, 60:64] *= 1e-30
        kv[:, :, 300:310] = 0.0
        kv[1, 0, 600, 2, 5] = float("-inf")
        kv = kv.to(torch.bfloat16)
        lay = nat.KVLayout.from_chunk(kv.to(DEV), "vllm")
        fused, _, _ = encode_with_path(nat, ctx, "fused", lay, 0, T, cs, bins)
        for i, blob in enumerate(fused):
            b, code = oracle.torch_to_bits(kv[:, :, cs * i:cs * (i + 1)].reshape(L, 2, cs, H * D))
            assert blob == oracle.encode_blob(b, code, H, D, np.array(bins, np.int32)), f"job {rep}, chunk {i}"


@pytest.mark.parametrize("H", [3, 2], ids=["C384", "C256_two_planes_per_item"])
@pytest.mark.parametrize("path", ["fused", "two_kernels"])
def test_counts_model_constant_and_sparse_channels(nat, ctx, oracle, path, H):
    """LMC_MODEL_COUNTS (256-token chunks) on the channels that stress its table: channels whose 256 symbols are all
    equal (count 256 -> 255 + 1 on a neighbour, lmc_counts_model: symbol 0 and another one), channels with a symbol
    that occurs once (frequency 2, the smallest the reciprocal table serves), two-symbol channels; on 32-bin and
    16-bin planes, both launch paths; blob and decode bit-exact against the oracle."""
    L, T, D = 2, 512, 128
    g = torch.Generator().manual_seed(5)
    x = torch.randn(L, 2, T, H, D, generator=g)
    big = x.abs().amax(dim=(-1, -2), keepdim=True)
    x[:, :, :, 0, 0:8] = 0.0                                  # the middle symbol, every token
    x[:, :, :, 0, 8:16] = big[..., 0]                         # the top symbol, every token
    x[:, :, :, 0, 16:24] = -big[..., 0]                       # symbol 0, every token
    x[:, :, :, 1, 0:8] = 0.0
    x[:, :, 100, 1, 0:8] = big[:, :, 100, 0]                  # one outlier token in a constant channel
    x[:, :, ::2, H - 1, 5] = 0.0                              # two-symbol-ish channel
    kv = x.to(torch.bfloat16)
    bins = [32, 16, 16, 32]
    lay = nat.KVLayout.from_chunk(kv.to(DEV), "vllm")
    blobs, blob_dev, stride = encode_with_path(nat, ctx, path, lay, 0, T, 256, bins)
    out = torch.zeros(L, 2, T, H, D, dtype=torch.bfloat16, device=DEV)
    ctx.decode_chunks(blob_dev.data_ptr(), stride, 2, nat.KVLayout.from_chunk(out, "vllm"), 0, 256)
    torch.cuda.synchronize()
    ctx.raise_on_status("decode")
    for i in range(2):
        b, code = oracle.torch_to_bits(kv[:, :, 256 * i:256 * (i + 1)].reshape(L, 2, 256, H * D))
        ref = oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))
        assert nat.blob_info(ref).model == 1
        assert blobs[i] == ref, f"{path}, chunk {i}"
        assert np.array_equal(bits_np(out[:, :, 256 * i:256 * (i + 1)]).reshape(L, 2, 256, H * D),
                              oracle.decode_blob(ref, oracle.BF16)), f"decode, chunk {i}"


@pytest.mark.parametrize("layout", ["NBHD", "NHBD"])
def test_fused_encode_gathers_paged_blocks(nat, ctx, oracle, layout):
    """slot_mapping gather inside the fused kernel (LLM_Engine.rst:91-122), kv-tuple plane pointers."""
    L, T, H, D, bs, nb, cs = 2, 600, 8, 128, 16, 41, 256
    g = torch.Generator().manual_seed(31)
    shape = (2, nb, bs, H, D) if layout == "NBHD" else (2, nb, H, bs, D)
    caches = [torch.randn(shape, generator=g).to(torch.bfloat16).to(DEV) for _ in range(L)]
    slots = torch.randperm(nb * bs, generator=g)[:T]
    lay = nat.KVLayout.paged(caches, slots, bs, layout)
    bins = default_bins(L)
    fused, _, _ = encode_with_path(nat, ctx, "fused", lay, 0, T, cs, bins)
    dense = torch.zeros(L, 2, T, H, D, dtype=torch.bfloat16)
    for l in range(L):
        c = caches[l].cpu()
        for t, s in enumerate(slots.tolist()):
            blk, w = divmod(s, bs)
            dense[l, :, t] = c[:, blk, w] if layout == "NBHD" else c[:, blk, :, w]
    for i, blob in enumerate(fused):
        t0, t1 = cs * i, min(T, cs * (i + 1))
        b, code = oracle.torch_to_bits(dense[:, :, t0:t1].reshape(L, 2, t1 - t0, H * D))
        assert blob == oracle.encode_blob(b, code, H, D, np.array(bins, np.int32)), f"chunk {i}"


def test_fused_setting_is_a_preference_for_other_chunk_lengths(nat, ctx, oracle):
    """The fused kernel codes 256-token chunks; with any other chunk length the setting is a preference and the job takes
    k_quantize + k_cdf_encode (C = 128 and C = 4096 here: geometries the fused kernel does take at 256 tokens)."""
    for (L, T, H, D) in ((2, 20, 1, 128), (1, 12, 32, 128)):
        kv = make_kv(L, T, H, D, torch.bfloat16, "randn", seed=H)
        bins = default_bins(L)
        blobs, _, _ = encode_with_path(nat, ctx, "fused", nat.KVLayout.from_chunk(kv.to(DEV), "vllm"), 0, T, T, bins)
        b, code = oracle.torch_to_bits(kv.reshape(L, 2, T, H * D))
        assert blobs[0] == oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))


def test_fused_encode_of_the_70b_tp8_rank_shape_at_32k(nat, ctx, oracle):
    """BASELINE configs[3]: Llama-3-70B, TP = 8 -- a rank holds 80 layers x 1 KV head x 128 = planes of 128 channels --
    32 768 tokens = 128 chunks.  The whole job through lmc_encode_chunks' default path (AUTO picks the fused kernel:
    20 work items of 8 planes per chunk x 128 chunks), decoded again and held against the quantisation bound, and three
    chunks byte for byte against the oracle."""
    L, T, H, D, cs = 80, 32768, 1, 128, 256
    bins = [32] * 10 + [16] * 70 + [32] * 2 + [16] * 78
    kv = make_kv(L, T, H, D, torch.bfloat16, "randn", seed=3).to(DEV)
    lay = nat.KVLayout.from_chunk(kv, "vllm")
    n = T // cs
    stride = nat.r16(nat.blob_bound(L, cs, H, D))
    blob_dev = torch.empty(n * stride, dtype=torch.uint8, device=DEV)
    sizes = torch.zeros(n, dtype=torch.int32, device=DEV)
    ctx.encode_chunks(lay, 0, T, cs, bins, blob_dev.data_ptr(), stride, sizes.data_ptr())
    out = torch.zeros_like(kv)
    ctx.decode_chunks(blob_dev.data_ptr(), stride, n, nat.KVLayout.from_chunk(out, "vllm"), 0, cs)
    torch.cuda.synchronize()
    ctx.raise_on_status("70B rank shape")
    szs = sizes.cpu().tolist()
    assert all(0 < s <= stride for s in szs)
    for i in (0, 57, n - 1):
        chunk = kv[:, :, i * cs:(i + 1) * cs].cpu()
        b, code = oracle.torch_to_bits(chunk.reshape(L, 2, cs, H * D))
        ref = oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))
        assert blob_dev[i * stride:i * stride + szs[i]].cpu().numpy().tobytes() == ref, i
        assert np.array_equal(bits_np(out[:, :, i * cs:(i + 1) * cs]).reshape(L, 2, cs, H * D), oracle.decode_blob(ref, oracle.BF16))
    # size-independent property over the whole job: |decode(encode(x)) - x| within the quantisation bound
    mx = kv.float().abs().amax(dim=(3, 4), keepdim=True)
    M = torch.tensor([b // 2 - 1 for b in bins], dtype=torch.float32, device=DEV).reshape(2, L).T.reshape(L, 2, 1, 1, 1)
    assert ((out.float() - kv.float()).abs() <= mx / (2 * M) + mx * 2.0 ** -7).all()


@pytest.mark.parametrize("T", [44, 100, 255, 256])
def test_round4_cdf16_blobs_of_short_chunks_still_decode(nat, ctx, oracle, T):
    """A format-v6 blob whose header says CDF16 at a chunk length the present encoder codes on the counts model (what
    rounds 3-4 wrote for ragged and < 256-token chunks) is decoded, not rejected as BAD_HEADER: k_decode and
    lmc_blob_info go by the header's model word (such blobs live on in remote stores)."""
    L, H, D = 2, 2, 128
    kv = make_kv(L, T, H, D, torch.bfloat16, "randn", seed=T)
    bins = [32, 16, 16, 22]
    b, code = oracle.torch_to_bits(kv.reshape(L, 2, T, H * D))
    old = oracle.encode_blob(b, code, H, D, np.array(bins, np.int32), model=oracle.MODEL_CDF16)
    hdr = nat.blob_info(old)
    assert (hdr.ntokens, hdr.total_bytes) == (T, len(old))
    stride = nat.r16(nat.blob_bound(L, T, H, D))
    dev = torch.zeros(stride, dtype=torch.uint8, device=DEV)
    dev[:len(old)] = torch.frombuffer(bytearray(old), dtype=torch.uint8).to(DEV)
    for odt, ocode in ((torch.bfloat16, oracle.BF16), (torch.float16, oracle.FP16)):
        out = torch.zeros(L, 2, T, H, D, dtype=odt, device=DEV)
        ctx.decode_chunks(dev.data_ptr(), stride, 1, nat.KVLayout.from_chunk(out, "vllm"), 0, T)
        torch.cuda.synchronize()
        ctx.raise_on_status("decode of a CDF16 blob")
        assert np.array_equal(bits_np(out).reshape(L, 2, T, H * D), oracle.decode_blob(old, ocode))


def test_decode_schedule_in_one_call_equals_the_oracle(nat, ctx, oracle):
    """lmc_decode_chunks_schedule (round 6): every range of layers of a retrieve launched, and its event recorded, by ONE
    C-ABI call -- the result is the oracle's decode, the events fire in order, a schedule that does not cover the layers
    exactly is refused before anything is launched."""
    L, T, H, D, n = 5, 256, 2, 128, 3
    kv = make_kv(L, n * T, H, D, torch.bfloat16, "randn", seed=11)
    bins = default_bins(L)
    blobs, blob_dev, stride = encode(nat, ctx, nat.KVLayout.from_chunk(kv.to(DEV), "vllm"), 0, n * T, T, bins)
    table = torch.tensor([blob_dev.data_ptr() + i * stride for i in range(n)], dtype=torch.int64, device=DEV)
    out = torch.zeros(L, 2, n * T, H, D, dtype=torch.bfloat16, device=DEV)
    lay = nat.KVLayout.from_chunk(out, "vllm")
    evs = [nat.NativeEvent() for _ in range(3)]
    ctx.decode_chunks_schedule(table.data_ptr(), stride, n, lay, 0, T, [1, 3, L], evs)
    evs[-1].synchronize()
    assert all(e.query() for e in evs)
    ctx.raise_on_status("decode schedule")
    for i in range(n):
        b, code = oracle.torch_to_bits(kv[:, :, i * T:(i + 1) * T].reshape(L, 2, T, H * D))
        want = oracle.decode_blob(oracle.encode_blob(b, code, H, D, np.array(bins, np.int32)), oracle.BF16)
        assert blobs[i] == oracle.encode_blob(b, code, H, D, np.array(bins, np.int32))
        assert np.array_equal(bits_np(out[:, :, i * T:(i + 1) * T]).reshape(L, 2, T, H * D), want)
    for bad in ([1, 3], [3, 1, L], [0, L], [2, 2, L], [1, L + 1]):
        with pytest.raises(nat.NativeError):
            ctx.decode_chunks_schedule(table.data_ptr(), stride, n, lay, 0, T, bad, None)
