"""ctypes/numpy front-end of the CPU oracle (oracle/lmc_oracle.c).

TEST INFRASTRUCTURE ONLY.  Nothing under ``lmcache_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker / timed CPU baseline.

Each wrapper keeps the reference citation of the C function it calls; see the C
file for the restated algorithm.  Arrays are numpy; 16-bit floats travel as raw
``uint16`` bit patterns (dtype 0 = bf16, 1 = fp16).
"""
import ctypes
import os
import subprocess
from typing import List, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblmc_oracle.so")

BF16, FP16 = 0, 1
LANES, MAX_BINS, LP = 64, 32, 33


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "lmc_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "lmc_format.h")
    stale = (not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src)
             or os.path.getmtime(_SO) < os.path.getmtime(hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblmc_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        vp, i32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        L.lmco_sha256.argtypes = [vp, sz, vp]
        L.lmco_prefix_hash.argtypes = [vp, sz, sz, vp]
        L.lmco_prefix_hash.restype = sz
        L.lmco_quantize.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp]
        L.lmco_dequantize.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32]
        L.lmco_cdf.argtypes = [vp, i32, i32, i32, i32, vp]
        L.lmco_encode_group.argtypes = [vp, i32, i32, i32, vp, vp, sz]
        L.lmco_encode_group.restype = sz
        L.lmco_decode_group.argtypes = [vp, sz, i32, i32, i32, vp, vp]
        L.lmco_decode_group.restype = i32
        L.lmco_encode_blob.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, sz, vp]
        L.lmco_encode_blob.restype = i32
        L.lmco_encode_blob_model.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, sz, vp, i32]
        L.lmco_encode_blob_model.restype = i32
        L.lmco_ws_release_all.argtypes = [i32]
        L.lmco_ws_release_all.restype = None
        L.lmco_set_threads.argtypes = [i32]
        L.lmco_set_threads.restype = i32
        L.lmco_encode_blobs_parallel.argtypes = [vp, sz, i32, i32, i32, i32, i32, vp, i32, vp, sz, vp]
        L.lmco_encode_blobs_parallel.restype = i32
        L.lmco_blob_cdf.argtypes = [vp, sz, vp]
        L.lmco_blob_cdf.restype = i32
        L.lmco_decode_blob_symbols.argtypes = [vp, sz, vp]
        L.lmco_decode_blob_symbols.restype = i32
        L.lmco_decode_blob.argtypes = [vp, sz, vp, i32]
        L.lmco_decode_blob.restype = i32
        L.lmco_rans_magic.argtypes = [ctypes.c_uint32, vp, vp]
        L.lmco_blob_bound.argtypes = [i32, i32, i32, i32]
        L.lmco_blob_bound.restype = ctypes.c_uint64
        L.lmco_h2f.argtypes = [ctypes.c_uint16, i32]
        L.lmco_h2f.restype = ctypes.c_float
        L.lmco_f2h.argtypes = [ctypes.c_float, i32]
        L.lmco_f2h.restype = ctypes.c_uint16
        _lib = L
    return _lib


def _p(a: np.ndarray) -> ctypes.c_void_p:
    assert a.flags["C_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


def rans_magic(count: int):
    """(magic, shift) of lmc_rans_magic for freq = 2 * count."""
    m, sh = ctypes.c_uint32(0), ctypes.c_uint32(0)
    lib().lmco_rans_magic(count, ctypes.byref(m), ctypes.byref(sh))
    return m.value, sh.value


def sha256_hex(data: bytes) -> str:
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(0, np.uint8)
    out = np.zeros(32, np.uint8)
    lib().lmco_sha256(_p(np.ascontiguousarray(buf)), len(data), _p(out))
    return out.tobytes().hex()


def prefix_hash(tokens: np.ndarray, chunk_size: int) -> List[str]:
    """cache_engine.py:58-96 -- tokens int64 1-D."""
    tokens = np.ascontiguousarray(tokens, dtype=np.int64)
    n = (len(tokens) + chunk_size - 1) // chunk_size
    out = np.zeros(max(n, 1) * 65, np.uint8)
    got = lib().lmco_prefix_hash(_p(tokens), len(tokens), chunk_size, _p(out))
    assert got == n
    return [out[i * 65:i * 65 + 64].tobytes().decode("ascii") for i in range(n)]


def quantize(kv_bits: np.ndarray, dtype: int, bins: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """cachegen_encoder.py:40-61,278-285.  kv_bits uint16 [L,2,T,C] -> (sym int8 [2L,T,C], scale uint16 [2L,T])."""
    L, two, T, C = kv_bits.shape
    assert two == 2 and kv_bits.dtype == np.uint16
    bins = np.ascontiguousarray(bins, dtype=np.int32)
    assert bins.shape == (2 * L,)
    sym = np.empty((2 * L, T, C), np.int8)
    scale = np.empty((2 * L, T), np.uint16)
    lib().lmco_quantize(_p(np.ascontiguousarray(kv_bits)), dtype, L, T, C, _p(bins), _p(sym), _p(scale))
    return sym, scale


def dequantize(sym: np.ndarray, scale: np.ndarray, scale_dtype: int, bins: np.ndarray,
               out_dtype: int) -> np.ndarray:
    """cachegen_decoder.py:24-35,177-200.  -> uint16 [L,2,T,C] in out_dtype."""
    P, T, C = sym.shape
    L = P // 2
    bins = np.ascontiguousarray(bins, dtype=np.int32)
    out = np.empty((L, 2, T, C), np.uint16)
    lib().lmco_dequantize(_p(np.ascontiguousarray(sym)), _p(np.ascontiguousarray(scale)), scale_dtype,
                          L, T, C, _p(bins), _p(out), out_dtype)
    return out


def cdf(sym: np.ndarray, max_bins: int = MAX_BINS) -> np.ndarray:
    """cachegen_encoder.py:95-126,175-222 (in-tree spec of torchac_cuda.calculate_cdf).  -> uint16 [P,C,max_bins+1]."""
    P, T, C = sym.shape
    out = np.empty((P, C, max_bins + 1), np.uint16)
    lib().lmco_cdf(_p(np.ascontiguousarray(sym)), P, T, C, max_bins, _p(out))
    return out


def group_cap_bytes(T: int) -> int:
    """lmc_group_cap_bytes: the largest head + words and states."""
    stored = 255 if T == 256 else T
    head = (32 + 8 * 31 * stored.bit_length() + 15) & ~15
    return head + ((LANES * (T + 8) + 15) & ~15)


def encode_group(sym_plane: np.ndarray, g: int, cdf_plane: np.ndarray) -> bytes:
    """One 64-channel group stream (exact bytes, no padding)."""
    T, C = sym_plane.shape
    cap = group_cap_bytes(T)
    out = np.zeros(cap // 2, np.uint16)
    n = lib().lmco_encode_group(_p(np.ascontiguousarray(sym_plane)), T, C, g,
                                _p(np.ascontiguousarray(cdf_plane)), _p(out), cap // 2)
    assert n > 0, "group stream overflow"
    return out.tobytes()[:n]


def decode_group(stream: bytes, T: int, C: int, g: int, cdf_plane: np.ndarray,
                 sym_plane: np.ndarray) -> int:
    buf = np.frombuffer(stream, dtype=np.uint16).copy()
    return lib().lmco_decode_group(_p(buf), len(stream), T, C, g, _p(np.ascontiguousarray(cdf_plane)),
                                   _p(sym_plane))


def blob_bound(L: int, T: int, H: int, D: int) -> int:
    return int(lib().lmco_blob_bound(L, T, H, D))


MODEL_CDF16, MODEL_COUNTS = 0, 1  # include/lmc_format.h: LMC_MODEL_*


def encode_blob(kv_bits: np.ndarray, dtype: int, H: int, D: int, bins: np.ndarray, model: int = -1) -> bytes:
    """cachegen_encoder.py:266-325,352-389 with our container.  kv_bits uint16 [L,2,T,H*D].
    model: -1 = the encoder's choice for the chunk length (lmc_model_for); MODEL_CDF16 = the 16-bit CDF whatever the
    length -- the form rounds 3-4 wrote chunks other than 256 tokens in, which a decoder must keep reading."""
    L, two, T, C = kv_bits.shape
    assert two == 2 and C == H * D
    bins = np.ascontiguousarray(bins, dtype=np.int32)
    cap = blob_bound(L, T, H, D)
    blob = np.zeros(cap, np.uint8)
    nbytes = ctypes.c_size_t(0)
    rc = lib().lmco_encode_blob_model(_p(np.ascontiguousarray(kv_bits)), dtype, L, T, H, D, _p(bins), _p(blob),
                                      cap, ctypes.byref(nbytes), model)
    assert rc == 0, f"lmco_encode_blob rc={rc}"
    return blob[:nbytes.value].tobytes()


def release_workspaces(nthreads: int = 0) -> None:
    """Free the per-thread buffers of lmco_encode_blob on every OpenMP worker (a region of `nthreads` threads; 0: as
    many as the runtime gives) and the output arena of encode_blobs_parallel."""
    global _par_out
    lib().lmco_ws_release_all(nthreads)
    _par_out = None


def set_threads(n: int) -> int:
    """OpenMP threads for the calls that follow (0: leave as is); returns what a parallel region will get."""
    return int(lib().lmco_set_threads(n))


_par_out = None  # the output arena of encode_blobs_parallel, kept between calls (a fresh np.empty of n x 18 MB is page-faulted by the workers every time)


def encode_blobs_parallel(kv_bits: np.ndarray, dtype: int, H: int, D: int, bins: np.ndarray, n: int) -> List[int]:
    """The same chunk n times, one chunk per OpenMP thread (OMP_NUM_THREADS): the CPU baseline's timed call.
    Returns the blob sizes."""
    global _par_out
    L, two, T, C = kv_bits.shape
    assert two == 2 and C == H * D
    bins = np.ascontiguousarray(bins, dtype=np.int32)
    stride = (blob_bound(L, T, H, D) + 63) & ~63
    if _par_out is None or _par_out.size < n * stride:
        _par_out = np.empty(n * stride, np.uint8)
    blobs = _par_out
    sizes = np.zeros(n, np.uint64)
    rc = lib().lmco_encode_blobs_parallel(_p(np.ascontiguousarray(kv_bits)), 0, dtype, L, T, H, D, _p(bins), n, _p(blobs),
                                          stride, _p(sizes))
    assert rc == 0, f"{rc} chunks failed"
    return sizes.astype(np.int64).tolist()


def parse_header(blob: bytes) -> dict:
    names = ["dtype", "num_layers", "ntokens", "num_heads", "head_size", "nchannels", "nplanes", "ngroups",
             "lp", "off_bins", "off_scales", "zero13", "off_gdir", "off_streams", "stream_bytes",
             "total_bytes", "zero18", "zero19", "zero20", "off_scsum", "model"]
    head = np.frombuffer(blob[:128], dtype=np.uint32)
    assert head[0] == 0x31434D4C, "bad magic"
    d = {n: int(v) for n, v in zip(names, head[2:2 + len(names)])}
    d["version"] = int(head[1] & 0xffff)
    return d


def stream_dir(blob: bytes) -> np.ndarray:
    """The stream directory [P * G, 2] = {beg, end} relative to the streams section."""
    h = parse_header(blob)
    return np.frombuffer(blob, np.uint32, 2 * h["nplanes"] * h["ngroups"], h["off_gdir"]).reshape(-1, 2).astype(np.int64)


def stream_head(blob: bytes, pg: int):
    """(widths uint8 [R], stored counts int64 [R, 64], head bytes) of stream pg, parsed in numpy (lmc_format.h: head)."""
    h = parse_header(blob)
    beg, end = stream_dir(blob)[pg]
    R = blob[h["off_bins"] + pg // h["ngroups"]] - 1
    R8 = (R + 7) & ~7
    base = h["off_streams"] + int(beg)
    widths = np.frombuffer(blob, np.uint8, R8, base)
    assert not widths[R:].any() and widths.max(initial=0) <= 16
    W = int(widths.sum())
    planes = np.frombuffer(blob, np.uint64, W, base + R8)
    cnt = np.zeros((R, 64), np.int64)
    j = 0
    lanes = np.arange(64, dtype=np.uint64)
    for i in range(R):
        for b in range(int(widths[i]) - 1, -1, -1):
            cnt[i] |= ((planes[j] >> lanes) & np.uint64(1)).astype(np.int64) << b
            j += 1
    return widths[:R].copy(), cnt, (R8 + 8 * W + 15) & ~15


def blob_cdf(blob: bytes) -> np.ndarray:
    """The reference's `cdf` tensor [2L, C, 33] rebuilt from the symbol counts the blob stores."""
    h = parse_header(blob)
    out = np.zeros((h["nplanes"], h["nchannels"], LP), np.uint16)
    buf = np.frombuffer(blob, dtype=np.uint8).copy()
    rc = lib().lmco_blob_cdf(_p(buf), len(blob), _p(out))
    assert rc == 0, f"lmco_blob_cdf rc={rc}"
    return out


def decode_blob_symbols(blob: bytes) -> np.ndarray:
    h = parse_header(blob)
    sym = np.zeros((h["nplanes"], h["ntokens"], h["nchannels"]), np.int8)
    buf = np.frombuffer(blob, dtype=np.uint8).copy()
    rc = lib().lmco_decode_blob_symbols(_p(buf), len(blob), _p(sym))
    assert rc == 0, f"lmco_decode_blob_symbols rc={rc}"
    return sym


def decode_blob(blob: bytes, out_dtype: int) -> np.ndarray:
    """cachegen_decoder.py:142-202 (vllm layout).  -> uint16 [L,2,T,C] in out_dtype."""
    h = parse_header(blob)
    out = np.zeros((h["num_layers"], 2, h["ntokens"], h["nchannels"]), np.uint16)
    buf = np.frombuffer(blob, dtype=np.uint8).copy()
    rc = lib().lmco_decode_blob(_p(buf), len(blob), _p(out), out_dtype)
    assert rc == 0, f"lmco_decode_blob rc={rc}"
    return out


# --- helpers shared by tests / bench -----------------------------------------
def cachegen_bins(model_name: str) -> Tuple[np.ndarray, int]:
    """Per-plane bins (key_bins ++ value_bins) restating CacheGenConfig.from_model_name and
    make_key_bins/make_value_bins (cachegen_basics.py:32-78, cachegen_encoder.py:339-350)."""
    if model_name in ("mistralai/Mistral-7B-Instruct-v0.2", "lmsys/longchat-7b-16k", "Qwen/Qwen-7B",
                      "meta-llama/Llama-3.1-8B-Instruct"):
        nl = 32
    elif model_name == "THUDM/glm-4-9b-chat":
        nl = 40
    else:
        raise ValueError(f"Model {model_name} is not supported")
    kb = np.full(nl, 16, np.int32)
    kb[:20] = 16
    kb[:10] = 32
    vb = np.full(nl, 16, np.int32)
    vb[:2] = 32
    return np.concatenate([kb, vb]), nl


def torch_to_bits(t) -> Tuple[np.ndarray, int]:
    """torch bf16/fp16 tensor -> (uint16 ndarray, dtype code)."""
    import torch
    code = BF16 if t.dtype == torch.bfloat16 else FP16
    assert t.dtype in (torch.bfloat16, torch.float16)
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16), code


def bits_to_torch(a: np.ndarray, code: int):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a).view(np.int16))
    return t.view(torch.bfloat16 if code == BF16 else torch.float16)


def pack_from_blobs(blobs, chunk_tokens: int) -> bytes:
    """CPU restatement of the pack layout (include/lmc_format.h, "pack"): the blobs of one store call -- all of the same
    geometry, every chunk but possibly the last of `chunk_tokens` tokens -- transposed plane-major.  Test infrastructure:
    what lmc_store_pack must produce byte for byte."""
    import struct
    hs = [parse_header(b) for b in blobs]
    h0 = hs[0]
    n, L, H, D, G = len(blobs), h0["num_layers"], h0["num_heads"], h0["head_size"], h0["ngroups"]
    assert all((h["num_layers"], h["num_heads"], h["head_size"]) == (L, H, D) for h in hs)
    r16 = lambda x: (x + 15) & ~15
    # lmc_blob_layout for a chunk_tokens-token blob of this geometry: where its streams would start
    P, C = 2 * L, H * D
    off = 128 + r16(P)                       # header | bins
    off += r16(2 * P * chunk_tokens)         # scales
    off += r16(4 * P)                        # scale checksums
    off += r16(8 * P * G)                    # stream directory {beg, end}
    static_stride = r16(off)
    assert all(h["ntokens"] == chunk_tokens for h in hs[:-1]) and all(h["off_streams"] <= static_stride for h in hs)
    assert any(h["ntokens"] != chunk_tokens for h in hs) or h0["off_streams"] == off
    N = 2 * L * n
    off_table = 256
    off_static = r16(off_table + 8 * (N + 1))
    off_streams = off_static + n * static_stride
    segs, table, at = [], [], 0
    for p in range(P):  # pack v3: plane order -- K planes of every layer, then V planes (p = kv * L + layer)
        if True:
            for c, (b, h) in enumerate(zip(blobs, hs)):
                gdir = np.frombuffer(b, dtype=np.uint32, count=2 * P * G, offset=h["off_gdir"])
                s0 = int(gdir[2 * p * G])  # beg of stream (p, 0) ... of stream (p + 1, 0), or the end of the section
                s1 = int(gdir[2 * (p + 1) * G]) if p + 1 < P else h["stream_bytes"]
                table.append(at)
                segs.append(b[h["off_streams"] + s0:h["off_streams"] + s1])
                at += s1 - s0
    table.append(at)
    ntok = sum(h["ntokens"] for h in hs)
    head = struct.pack("<12I4Q", 0x4b504d4c, 3, 256, n, L, H, D, chunk_tokens, G, static_stride, ntok, 0,
                       off_table, off_static, off_streams, off_streams + at)
    out = bytearray(off_streams + at)
    out[:len(head)] = head
    out[off_table:off_table + 8 * (N + 1)] = struct.pack(f"<{N + 1}Q", *table)
    for c, (b, h) in enumerate(zip(blobs, hs)):
        out[off_static + c * static_stride:off_static + c * static_stride + h["off_streams"]] = b[:h["off_streams"]]
    pos = off_streams
    for sgm in segs:
        out[pos:pos + len(sgm)] = sgm
        pos += len(sgm)
    return bytes(out)
