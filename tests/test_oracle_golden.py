"""Pin the CPU oracle against fixtures produced by the reference itself
(oracle/gen_golden.py -> tests/golden/).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest


def test_sha256_matches_hashlib(oracle):
    rng = np.random.default_rng(0)
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 2048, 2112):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.sha256_hex(data) == hashlib.sha256(data).hexdigest()


def test_prefix_hash_chain_golden(oracle, golden_dir):
    with open(os.path.join(golden_dir, "hash_chain.json")) as f:
        gold = json.load(f)
    # KAT quoted in SURVEY.md 8(c)
    c0 = gold["cases"][0]
    assert c0["hashes"][0] == "bbd330b12e8159e117376ef24fa106413bc9fc18032a0d43e95c5dae5e47953f"
    assert c0["hashes"][1] == "da67b0aaefba655d2edadd2cc5d11cd4564db9059ccf6264056d62d170b11ff5"
    for case in gold["cases"]:
        got = oracle.prefix_hash(np.array(case["tokens"], dtype=np.int64), case["chunk_size"])
        assert got == case["hashes"], case["name"]


@pytest.mark.parametrize("dname", ["bf16", "fp16"])
@pytest.mark.parametrize("kind", ["rand", "randn", "outlier"])
def test_quantize_dequantize_golden(oracle, golden_dir, dname, kind):
    z = np.load(os.path.join(golden_dir, "quant.npz"))
    code = oracle.BF16 if dname == "bf16" else oracle.FP16
    tag = f"{dname}_{kind}"
    sym, scale = oracle.quantize(z[f"{tag}_kv"], code, z["bins"])
    assert np.array_equal(sym, z[f"{tag}_sym"])
    assert np.array_equal(scale, z[f"{tag}_scale"])
    for out_name, out_code in (("bf16", oracle.BF16), ("fp16", oracle.FP16)):
        deq = oracle.dequantize(sym, scale, code, z["bins"], out_code)
        assert np.array_equal(deq, z[f"{tag}_deq_{out_name}"]), (tag, out_name)


def test_quantize_edge_rows_golden(oracle, golden_dir):
    z = np.load(os.path.join(golden_dir, "quant_edge.npz"))
    x = z["x"]  # [L=2, T, C] keys only; build a [L,2,T,C] chunk with V = K
    kv = np.stack([x, x], axis=1)
    bins = np.concatenate([z["bins"], z["bins"]])
    sym, scale = oracle.quantize(kv, oracle.BF16, bins)
    assert np.array_equal(sym[:2], z["sym"])
    gs = z["scale"]
    nan = (gs & 0x7fff) > 0x7f80
    assert np.array_equal(scale[:2][~nan], gs[~nan])
    assert (((scale[:2] & 0x7fff) > 0x7f80) == nan).all()
    deq = oracle.dequantize(sym, scale, oracle.BF16, bins, oracle.BF16)[:, 0]
    g = z["deq_bf16"]
    gnan = (g & 0x7fff) > 0x7f80
    assert np.array_equal(deq[~gnan], g[~gnan])
    assert (((deq & 0x7fff) > 0x7f80) == gnan).all()


@pytest.mark.parametrize("T", [1, 7, 16, 128, 236, 250, 256, 768])
def test_cdf_golden(oracle, golden_dir, T):
    z = np.load(os.path.join(golden_dir, "cdf.npz"))
    got = oracle.cdf(z[f"T{T}_sym"], 32)
    assert np.array_equal(got, z[f"T{T}_cdf"])
    # strictly increasing as uint16 except the final wrap to 65536 == 0
    g = got.astype(np.int64)
    g[..., -1] += 65536
    assert (np.diff(g, axis=-1) >= 1).all() and (g[..., 0] == 0).all() and (g[..., -1] == 65536).all()


def test_group_coder_roundtrip_and_bound(oracle):
    rng = np.random.default_rng(3)
    for T, C in ((1, 64), (5, 70), (256, 128), (300, 64)):
        for spread in (0.3, 2.0, 9.0):
            sym = np.clip(np.rint(rng.normal(15, spread, (1, T, C))), 0, 30).astype(np.int8)
            cdf = oracle.cdf(sym, 32)
            G = (C + 63) // 64
            for g in range(G):
                stream = oracle.encode_group(sym[0], g, cdf[0])
                assert len(stream) % 2 == 0 and len(stream) <= oracle.group_cap_bytes(T)
                out = np.full((T, C), -1, np.int8)
                assert oracle.decode_group(stream, T, C, g, cdf[0], out) == 0
                lo, hi = g * 64, min(C, g * 64 + 64)
                assert np.array_equal(out[:, lo:hi], sym[0][:, lo:hi])
                # corruption is detected by the final-state check
                bad = bytearray(stream)
                bad[len(bad) // 3] ^= 0x10
                if len(stream) > 256 + 8:
                    rc = oracle.decode_group(bytes(bad), T, C, g, cdf[0], out.copy())
                    # rANS is a bijection: same symbols + different words => the final state is not L
                    assert rc != 0


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_blob_roundtrip_equals_quant_dequant(oracle, dtype):
    """decode(encode(x)) == do_dequantize(torch_quant_vectorized(x)) -- SURVEY.md 8(c) oracle."""
    import torch
    torch.manual_seed(0)
    L, T, H, D = 3, 37, 2, 40
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    kv = torch.randn(L, 2, T, H * D).to(tdt)
    bits, code = oracle.torch_to_bits(kv)
    bins = np.array([32, 16, 16, 32, 16, 16], np.int32)
    blob = oracle.encode_blob(bits, code, H, D, bins)
    hdr = oracle.parse_header(blob)
    assert hdr["total_bytes"] == len(blob) and hdr["num_heads"] == H and hdr["head_size"] == D
    sym, scale = oracle.quantize(bits, code, bins)
    assert np.array_equal(oracle.decode_blob_symbols(blob), sym)
    for oc in (oracle.BF16, oracle.FP16):
        assert np.array_equal(oracle.decode_blob(blob, oc), oracle.dequantize(sym, scale, code, bins, oc))
    # error bound vs the original: |x^ - x| <= max1/(2M) + 1ulp16(max1)   
    dec = oracle.bits_to_torch(oracle.decode_blob(blob, code), code).float()
    mx = kv.float().abs().amax(-1, keepdim=True)
    M = torch.tensor(bins.reshape(2, L).T // 2 - 1).float()[:, :, None, None]
    tol = mx / (2 * M) + mx * (2.0 ** -7 if dtype == "bf16" else 2.0 ** -10)
    assert ((dec - kv.float()).abs() <= tol).all()


def test_blob_roundtrip_random_geometries(oracle):
    """The blob format end to end in the oracle over random geometries: ragged channel counts (idle lanes in the last
    group), bins anywhere in 4..32, chunk lengths on both sides of the one-byte / two-byte count boundary, and
    peaky data (few symbols per channel, counts that saturate)."""
    rng = np.random.default_rng(7)
    for it in range(14):
        L = int(rng.integers(1, 3))
        T = int(rng.choice([1, 2, 7, 31, 64, 255, 256, 257, 300]))
        H = int(rng.integers(1, 4))
        D = int(rng.choice([8, 16, 40, 64, 72]))
        bins = rng.choice(np.arange(4, 33, 2), size=2 * L).astype(np.int32)
        x = rng.standard_normal((L, 2, T, H * D)).astype(np.float32)
        if it % 3 == 0:
            x = np.round(x)            # few distinct values -> few symbols, saturating counts
        if it % 4 == 1:
            x[:, :, :, ::3] = 0.0      # constant channels
        import torch
        kv = torch.from_numpy(x).to(torch.bfloat16)
        bits, code = oracle.torch_to_bits(kv)
        blob = oracle.encode_blob(bits, code, H, D, bins)
        sym, scale = oracle.quantize(bits, code, bins)
        assert np.array_equal(oracle.decode_blob_symbols(blob), sym), (it, L, T, H, D)
        assert np.array_equal(oracle.blob_cdf(blob), oracle.cdf(sym)), (it, L, T, H, D)
        assert np.array_equal(oracle.decode_blob(blob, oracle.BF16),
                              oracle.dequantize(sym, scale, code, bins, oracle.BF16)), (it, L, T, H, D)
        assert len(blob) <= oracle.blob_bound(L, T, H, D)
        # the scales carry a per-plane checksum (format v4): a flipped scale bit, or a flipped checksum bit, fails
        h = oracle.parse_header(blob)
        assert h["version"] == 6 and h["off_scsum"] == h["off_scales"] + ((2 * 2 * L * T + 15) & ~15)
        for where in (h["off_scales"] + int(rng.integers(0, 2 * 2 * L * T)), h["off_scsum"] + int(rng.integers(0, 8 * L))):
            bad = bytearray(blob)
            bad[where] ^= 1 << int(rng.integers(0, 8))
            with pytest.raises(AssertionError, match="rc=-4"):
                oracle.decode_blob(bytes(bad), oracle.BF16)


def test_rans_magic_gives_the_exact_quotient_for_every_count(oracle):
    """lmc_rans_magic (lmc_format.h): x // freq == mulhi32(x, magic) >> shift for every freq = 2 * count of the counts
    model and every state x < 2^31.  The premise of the proof in the header (magic * freq - 2^(31 + l) < freq) is
    checked for every count; on top of it the identity itself on the multiples of freq and their neighbours over
    the whole state range (sampled: every multiple for small quotients, a stride above), the top 2^16 states, and
    a million random states per count."""
    rng = np.random.default_rng(5)
    for count in range(1, 256):
        f = 2 * count
        magic, shift = oracle.rans_magic(count)
        l = shift + 1
        assert (1 << (l - 1)) < f <= (1 << l) or f == 2 and l == 1
        assert 2**31 <= magic < 2**32
        assert magic * f - (1 << (31 + l)) < f and magic * f >= (1 << (31 + l))
        kmax = (2**31 - 1) // f
        ks = np.unique(np.concatenate([np.arange(0, min(kmax, 70000) + 1, dtype=np.uint64),
                                       np.arange(0, kmax + 1, max(1, kmax // 60000), dtype=np.uint64),
                                       np.array([kmax, max(kmax - 1, 0)], dtype=np.uint64)]))
        xs = np.concatenate([ks * np.uint64(f), ks * np.uint64(f) + np.uint64(f - 1), ks * np.uint64(f) + np.uint64(1),
                             np.arange(2**31 - 2**16, 2**31, dtype=np.uint64),
                             rng.integers(0, 2**31, 1_000_000, dtype=np.uint64)])
        xs = xs[xs < np.uint64(2**31)]
        q = ((xs * np.uint64(magic)) >> np.uint64(32)) >> np.uint64(shift)
        assert np.array_equal(q, xs // np.uint64(f)), count


def test_counts_bound_table_and_adversarial_streams(oracle):
    """Format v6 places a counts-model stream BEFORE it is coded, in an allocation computed from the channels' counts
    (lmc_format.h: lmc_counts_bits, lmc_counts_lane_words, lmc_counts_alloc_bytes).  (1) the table is what its comment says; (2) the bound
    holds -- lmco_encode_blob fails if a stream outgrows its allocation -- on channels built to stress it: one
    dominant symbol (the state idles near 2^15, where the per-step excess is largest), symbols in sorted order (every
    renormalisation pattern a lane can have), two-symbol and constant channels; (3) the slack it costs on iid data
    stays below 1.5 % of the streams."""
    import math, re
    src = open(os.path.join(os.path.dirname(__file__), "..", "include", "lmc_format.h")).read()
    body = re.search(r"#define LMC_COUNTS_BITS_LIST(.*?)\nstatic const", src, re.S).group(1)
    tab = [int(v) for v in body.replace("\\", " ").replace("\n", " ").split(",")]
    assert len(tab) == 257 and tab[0] == 0 and tab[256] == 0
    for c in range(1, 256):
        bits = c * math.log2(256 / c) + c * math.log2(1 + 2 * c / 32768)
        assert tab[c] == math.ceil(256 * bits) + 1, c
    rng = np.random.default_rng(11)
    L, T, H, D = 1, 256, 1, 64
    bins = np.array([32, 16], np.int32)

    def encode_symbols(sym_k, sym_v):
        # KV whose quantised symbols are exactly the wanted ones: x = (s - M) with one full-scale element per row
        out = np.zeros((L, 2, T, H * D), np.float32)
        for kv, (sy, b) in enumerate(((sym_k, 32), (sym_v, 16))):
            M = b // 2 - 1
            out[0, kv] = (sy.astype(np.float32) - M)
            out[0, kv, :, 0] = M  # row max = M: factor 1, symbol of lane 0 is 2 M
        import torch
        kvt = torch.from_numpy(out).to(torch.bfloat16)
        bits, code = oracle.torch_to_bits(kvt)
        blob = oracle.encode_blob(bits, code, H, D, bins)   # asserts rc == 0: no stream outgrew its allocation
        sym, _ = oracle.quantize(bits, code, bins)
        assert np.array_equal(oracle.decode_blob_symbols(blob), sym)
        return blob

    cases = []
    for dom in (1, 2, 3, 5, 9, 17, 40, 100, 200, 254):          # `dom` rare tokens, the rest one symbol
        for order in ("front", "back", "spread"):
            s = np.full((T, 64), 7, np.int64)
            idx = {"front": np.arange(dom), "back": np.arange(T - dom, T),
                   "spread": np.linspace(0, T - 1, dom).astype(int)}[order]
            s[idx] = rng.integers(0, 14, (len(idx), 64))
            cases.append(s)
    for k in (2, 3, 5, 8, 14):                                   # k symbols, sorted either way / random
        base = rng.integers(0, k, (T, 64))
        cases += [np.sort(base, axis=0), np.sort(base, axis=0)[::-1].copy(), base]
    cases.append(np.zeros((T, 64), np.int64))
    for s in cases:
        s = np.clip(s, 0, 14)
        encode_symbols(np.clip(s * 2, 0, 30), s)
    # slack of the bound on iid symbols
    s = rng.integers(0, 15, (T, 64))
    blob = encode_symbols(s * 2, s)
    h = oracle.parse_header(blob)
    gdir = oracle.stream_dir(blob)
    used = int((gdir[:, 1] - gdir[:, 0]).sum())
    assert used <= h["stream_bytes"] <= used * 1.015 + 32
    # every stream begins where the allocations in front of it end, and what lies between a stream and the next is zero
    raw = np.frombuffer(blob, np.uint8, h["stream_bytes"], h["off_streams"])
    ends = np.concatenate([gdir[1:, 0], [h["stream_bytes"]]])
    assert gdir[0, 0] == 0 and (gdir[:, 0] % 16 == 0).all() and (gdir[:, 1] <= ends).all()
    assert all(not raw[e:n].any() for e, n in zip(gdir[:, 1], ends))


def test_counts_model_below_256_tokens_bound_and_roundtrip(oracle):
    """Round 5: LMC_MODEL_COUNTS codes every chunk of 2 .. 256 tokens (lmc_format.h: lmc_counts_model scales the counts to
    a sum of 256 along the cumulative sum; the stream bound of T < 256 is sum cnt * lmc_counts_bpo[model count]).
    (1) the per-occurrence table is what its comment says; (2) the model: sums to 256, keeps every occurring symbol at
    >= its count, absent ones at 0, is the identity at T = 256, and its multiply-high scaling is the exact quotient;
    (3) the bound holds (lmco_encode_blob fails otherwise) and the round trip is exact on stress channels at
    T = 2, 3, 31, 33, 128, 236, 255; (4) the allocation slack on iid symbols stays below 2 %; T = 1 stays CDF16."""
    import math, re
    src = open(os.path.join(os.path.dirname(__file__), "..", "include", "lmc_format.h")).read()
    body = re.search(r"#define LMC_COUNTS_BPO_LIST(.*?)\nstatic const", src, re.S).group(1)
    tab = [int(v) for v in body.replace("\\", " ").replace("\n", " ").split(",")]
    assert len(tab) == 257 and tab[0] == 0 and tab[256] == 0
    for n in range(1, 256):
        assert tab[n] == math.ceil(256 * (math.log2(256 / n) + math.log2(1 + 2 * n / 32768))) + 1, n
    # (2) the model, restated in Python
    rng = np.random.default_rng(5)

    def model(cnt, T):
        cum = np.cumsum(cnt)
        N = cum if T == 256 else (cum * 256) // T
        n = np.diff(np.concatenate([[0], N]))
        if n.max() >= 256:
            s = int(n.argmax())
            n[s] = 255
            n[1 if s == 0 else 0] = 1
        return n
    for T in (2, 3, 7, 31, 33, 100, 128, 236, 255, 256):
        magic = (2**32 + T - 1) // T
        for x in range(T + 1):
            assert ((x << 8) * magic) >> 32 == (256 * x) // T, (T, x)
        for _ in range(50):
            k = int(rng.integers(1, 16))
            cnt = np.bincount(rng.integers(0, k, T), minlength=15)
            n = model(cnt.copy(), T)
            assert n.sum() == 256 and ((n > 0) == (cnt > 0)).all() or cnt.max() == T
            if cnt.max() < T:
                assert (n >= cnt).all()
            if T == 256 and cnt.max() < 256:
                assert (n == cnt).all()
    # (3) + (4) through the oracle's blob encoder (it checks every stream against its allocation)
    L, H, D = 1, 1, 64
    bins = np.array([32, 16], np.int32)

    def encode_symbols(sym_k, sym_v, T):
        out = np.zeros((L, 2, T, H * D), np.float32)
        for kv, (sy, b) in enumerate(((sym_k, 32), (sym_v, 16))):
            M = b // 2 - 1
            out[0, kv] = (sy.astype(np.float32) - M)
            out[0, kv, :, 0] = M
        import torch
        bits, code = oracle.torch_to_bits(torch.from_numpy(out).to(torch.bfloat16))
        blob = oracle.encode_blob(bits, code, H, D, bins)
        sym, _ = oracle.quantize(bits, code, bins)
        assert np.array_equal(oracle.decode_blob_symbols(blob), sym)
        return blob
    for T in (2, 3, 31, 33, 128, 236, 255):
        assert oracle.parse_header(encode_symbols(np.zeros((T, 64), np.int64), np.zeros((T, 64), np.int64), T))["model"] == 1
        cases = [np.zeros((T, 64), np.int64), rng.integers(0, 15, (T, 64)), np.sort(rng.integers(0, 15, (T, 64)), axis=0)]
        for dom in (1, 2, 5, T // 2, T - 1):
            if 0 < dom < T:
                s = np.full((T, 64), 7, np.int64)
                s[np.linspace(0, T - 1, dom).astype(int)] = rng.integers(0, 14, (dom, 64))
                cases.append(s)
        for s in cases:
            s = np.clip(s, 0, 14)
            encode_symbols(np.clip(s * 2, 0, 30), s, T)
    for T in (128, 236):
        s = rng.integers(0, 15, (T, 64))
        blob = encode_symbols(s * 2, s, T)
        h = oracle.parse_header(blob)
        gdir = oracle.stream_dir(blob)
        used = int((gdir[:, 1] - gdir[:, 0]).sum())
        assert used <= h["stream_bytes"] <= used * 1.02 + 32, (T, used, h["stream_bytes"])
    one = encode_symbols(np.zeros((1, 64), np.int64), np.zeros((1, 64), np.int64), 1)
    assert oracle.parse_header(one)["model"] == 0


def test_stream_head_is_the_bit_sliced_counts(oracle):
    """lmc_format.h "head": widths = significant bits of the largest stored count of a symbol over the 64 lanes, the
    counts as 8-byte bit planes, most significant first; a count of 256 is stored as 255 (T <= 256); idle lanes store 0;
    T > 256 takes wider planes.  Parsed here in numpy, independently of the C parser."""
    import torch
    rng = np.random.default_rng(3)
    for T, H, D, bins in ((256, 1, 72, [32, 16]), (256, 2, 64, [16, 8]), (300, 1, 64, [32, 4]), (7, 1, 8, [6, 32])):
        L = 1
        x = rng.standard_normal((L, 2, T, H * D)).astype(np.float32)
        x[:, :, :, 1] = 1.0   # a constant channel: count T on one symbol
        kv = torch.from_numpy(x).to(torch.bfloat16)
        bits, code = oracle.torch_to_bits(kv)
        blob = oracle.encode_blob(bits, code, H, D, np.array(bins, np.int32))
        sym, _ = oracle.quantize(bits, code, np.array(bins, np.int32))
        h = oracle.parse_header(blob)
        G, C = h["ngroups"], H * D
        gdir = oracle.stream_dir(blob)
        for pg in range(2 * G):
            p, g = divmod(pg, G)
            R = bins[p] - 1
            widths, cnt, hb = oracle.stream_head(blob, pg)
            want = np.zeros((R, 64), np.int64)
            for lane in range(64):
                c = g * 64 + lane
                if c < C:
                    want[:, lane] = np.bincount(sym[p, :, c].astype(np.int64), minlength=R)[:R]
            stored = np.where((want > 255) & (T <= 256), 255, want)
            assert np.array_equal(cnt, stored), (T, pg)
            assert np.array_equal(widths, [int(v).bit_length() for v in stored.max(axis=1)]), (T, pg)
            assert hb % 16 == 0 and gdir[pg, 1] - gdir[pg, 0] >= hb + 256


@pytest.mark.parametrize("T", [2, 44, 100, 255, 256])
def test_cdf16_blobs_of_short_chunks_still_decode(oracle, T):
    """Rounds 3-4 coded every chunk other than 256 tokens on the 16-bit CDF (header model 0, format v6); round 5 moved
    2 .. 255 tokens to the counts model WITHOUT a version bump.  A store that outlives a build still holds the old
    blobs: the header's model word is what a decoder goes by (lmc_model_valid), not lmc_model_for(T)."""
    g = np.random.default_rng(T)
    import torch
    L, H, D = 2, 2, 64
    kv = torch.from_numpy(g.standard_normal((L, 2, T, H * D)).astype(np.float32)).to(torch.bfloat16)
    bits, code = oracle.torch_to_bits(kv)
    bins = np.array([32, 16, 16, 22], np.int32)
    old = oracle.encode_blob(bits, code, H, D, bins, model=oracle.MODEL_CDF16)
    new = oracle.encode_blob(bits, code, H, D, bins)
    assert oracle.parse_header(old)["model"] == oracle.MODEL_CDF16
    assert oracle.parse_header(new)["model"] == oracle.MODEL_COUNTS
    assert old != new
    for dt in (oracle.BF16, oracle.FP16):
        assert np.array_equal(oracle.decode_blob(old, dt), oracle.decode_blob(new, dt))
    assert np.array_equal(oracle.decode_blob_symbols(old), oracle.decode_blob_symbols(new))
    assert np.array_equal(oracle.blob_cdf(old), oracle.blob_cdf(new))
