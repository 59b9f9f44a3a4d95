/*
 * lmc_format.h -- on-the-wire layout of one encoded KV chunk ("blob").
 *
 * Shared by the HIP product (lmcache_amd/csrc), the CPU oracle (oracle/) and
 * host-side parsers.  Plain C, no dependencies.
 *
 * What the blob replaces in the reference: the pickled
 * CacheGenGPUEncoderOutput {data_chunks[{bytestream, bytestream_lengths,
 * ntokens}], cdf, max_tensors_key, max_tensors_value, num_heads, head_size}
 * (lmcache/storage_backend/serde/cachegen_basics.py:109-142).  The byte format
 * is private to the serializer/deserializer pair: the backends treat it as
 * opaque bytes (lmcache/storage_backend/remote_backend.py:124-125,164-168).
 *
 * Plane order follows the reference's encode_input / cdf tensors
 * (cachegen_encoder.py:284,290): plane p = kv * L + layer, i.e. all K planes
 * first, then all V planes.  A plane is a [T tokens, C = H*D channels] matrix.
 *
 * All integers little-endian.  Every section starts on a 16-byte boundary.
 *
 *   header   128 B          struct lmc_blob_header
 *   bins     u8  [P]        quantisation bins of each plane (32 or 16 ...)
 *   rowpre   u16 [P+1]      rowpre[p] = sum over planes before p of R, R = bins - 1
 *   scales   u16 [P][T]     per-(plane,token) absmax, raw bits of the KV dtype
 *                           (= max_tensors_key ++ max_tensors_value)
 *   scsum    u32 [P]        checksum of each plane's scales (v4): sum over t of (t + 1) * (bits_t + 1)
 *                           mod 2^32 -- lmc_scale_checksum_term.  The entropy coder detects damage to the
 *                           counts and the streams by itself (the final rANS states must come out at
 *                           their start value); the scales are the one section it cannot see, and a
 *                           flipped scale would silently rescale a whole token row.
 *   cdf      the per-channel symbol statistics the 16-bit CDF is a function of.  Per plane p:
 *                           [R_p][C] counts of the symbols 0 .. R_p - 1 (symbol-major since v5: a
 *                           coder lane owns a channel and reads / writes its counts with one
 *                           coalesced access per symbol), R_p = bins - 1 being the number of
 *                           symbols the quantiser can emit (plane p starts C * rowpre[p] entries in).  header.count_bytes = 1 when T <= 256:
 *                           one byte per count, a count of 256 stored as 255 (the counts of
 *                           a channel sum to T, so a reader adds T - sum to the entry that
 *                           reads 255); otherwise 2 (u16).  The CDF of a channel is
 *                             cdf[i] = RNE(N_i * 65504 / T) + i  (mod 2^16),
 *                             N_i = number of its symbols < i,  i = 0 .. 32
 *                           -- exactly the values of the reference's `cdf` tensor
 *                           [2L, C, 33] (cachegen_encoder.py:95-126, 175-222; entries above
 *                           R_p come out as 65504 + i because N_i = T there).  Counts
 *                           instead of the 33 (v1) or bins - 2 (v2) u16 entries take a
 *                           Llama-3-8B chunk blob from 11.1 MB (v1) over 9.0 MB to 7.9 MB.
 *   gend     u32 [P][G]     EXACT end offset (bytes, relative to the streams
 *                           section) of group stream (p,g); G = ceil(C/64).
 *                           Stream (p,g) starts at roundup16(gend[prev]) (0 for
 *                           the first) -- so every stream starts 16-B aligned.
 *   streams  bytes          group streams in (p,g) order, each padded with
 *                           zero bytes to a multiple of 16.
 *
 * Group stream = interleaved rANS over 64 adjacent channels ("lanes"),
 * 32-bit state words, 16-bit renormalisation words:
 *
 *   [ u16 words, in the order the ENCODER emitted them ][ u32 state[64] ]
 *
 * The encoder walks tokens T-1 .. 0; at each token the lanes that must
 * renormalise append their low 16 state bits in ascending lane order.  The
 * decoder starts from the tail (states), walks tokens 0 .. T-1 and pops words
 * from the end, again in ascending lane order inside one token step.
 *
 * header.model says which probabilities the coder runs on (v5); both are
 * functions of the counts section alone:
 *
 *   LMC_MODEL_CDF16 (0)  any T.  The reference's 16-bit CDF itself:
 *       start = cdf[s], freq = cdf[s+1] - cdf[s], total 2^16, state in
 *       [2^16, 2^32):   emit while x >= freq << 16;
 *       x = (x / freq << 16) + x % freq + start.
 *   LMC_MODEL_COUNTS (1) T == 256 (the reference's chunk size).  The counts
 *       themselves: a channel's 256 symbols give counts that sum to 2^8, so
 *       freq = 2 * count, start = 2 * (number of smaller symbols), total 2^9
 *       (the factor 2 keeps every frequency >= 2, which lets the encoder divide
 *       with one multiply-high by a 32-bit reciprocal: lmc_rans_magic), state in
 *       [2^15, 2^31):  emit while x >= freq << 22;
 *       x = (x / freq << 9) + x % freq + start.
 *       A channel whose 256 symbols are all equal would need count 256: it is
 *       coded with count 255 and a count of 1 for symbol 0 (for symbol 1 if the
 *       channel's own symbol is 0) -- lmc_counts_model.  The code length is the
 *       channel's empirical entropy to within the 31 bits of final state.
 * See DESIGN.md "Entropy coder" for the recurrences and their bounds.
 */
#ifndef LMC_FORMAT_H
#define LMC_FORMAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMC_BLOB_MAGIC 0x31434D4Cu /* "LMC1" */
#define LMC_BLOB_VERSION 5u
#define LMC_HEADER_BYTES 128u

#define LMC_DTYPE_BF16 0
#define LMC_DTYPE_FP16 1

#define LMC_LANES 64        /* channels per group stream = wavefront width  */
#define LMC_MAX_BINS 32     /* torchac_cuda.calculate_cdf(sym, 32)          */
#define LMC_LP 33           /* CDF entries per channel = max_bins + 1       */
#define LMC_PROB_BITS 16
#define LMC_RANS_L (1u << 16) /* CDF16 model: lower bound of the normalised state interval */

#define LMC_MODEL_CDF16 0u
#define LMC_MODEL_COUNTS 1u
#define LMC_COUNTS_T 256u          /* chunk length the counts model codes */
#define LMC_COUNTS_BITS 9u         /* its probabilities are 2 * count out of 2^9 */
#define LMC_COUNTS_L (1u << 15)    /* ... and its state lives in [2^15, 2^31) */
#define LMC_CDF_SCALE (65536u - (LMC_LP - 1u)) /* 2^16 - (Lp-1), cachegen_encoder.py:117-119 */

typedef struct lmc_blob_header {
  uint32_t magic;
  uint16_t version;
  uint16_t header_bytes;
  uint32_t dtype;       /* LMC_DTYPE_* of the encoded KV (and of `scales`) */
  uint32_t num_layers;  /* L */
  uint32_t ntokens;     /* T */
  uint32_t num_heads;   /* H */
  uint32_t head_size;   /* D */
  uint32_t nchannels;   /* C = H*D */
  uint32_t nplanes;     /* P = 2L */
  uint32_t ngroups;     /* G = ceil(C/64) */
  uint32_t lp;          /* LMC_LP */
  uint32_t off_bins;
  uint32_t off_scales;
  uint32_t off_cdf;
  uint32_t off_gend;
  uint32_t off_streams;
  uint32_t stream_bytes; /* padded size of the streams section */
  uint32_t total_bytes;  /* off_streams + stream_bytes */
  uint32_t off_rowpre;
  uint32_t cdf_rows;     /* rowpre[P] = sum of R over all planes */
  uint32_t count_bytes;  /* bytes per stored count: 1 (T <= 256) or 2 */
  uint32_t off_scsum;    /* per-plane scale checksums */
  uint32_t model;        /* LMC_MODEL_*: = lmc_model_for(ntokens) */
  uint32_t reserved[9];
} lmc_blob_header;

static inline uint32_t lmc_r16(uint32_t x) { return (x + 15u) & ~15u; }

/* One term of a plane's scale checksum: token t (0-based) whose scale has the raw bits `bits`. */
static inline uint32_t lmc_scale_checksum_term(uint32_t t, uint32_t bits) { return (t + 1u) * (bits + 1u); }

/* Counts stored per channel of a plane quantised with `bins` bins: one per symbol 0 .. bins-2. */
static inline uint32_t lmc_cdf_row(uint32_t bins) { return bins - 1u; }
/* Bytes per stored count for a chunk of T tokens. */
static inline uint32_t lmc_count_bytes(uint32_t T) { return T <= 256u ? 1u : 2u; }

/* The coder model of a chunk of T tokens. */
static inline uint32_t lmc_model_for(uint32_t T) { return T == LMC_COUNTS_T ? LMC_MODEL_COUNTS : LMC_MODEL_CDF16; }

/* LMC_MODEL_COUNTS: the counts the coder runs on, from a channel's exact counts cnt[0 .. R) (sum 256, R >= 2):
 * themselves, except that a count of 256 becomes 255 and symbol 0 (symbol 1 if the channel's only symbol is 0)
 * gets 1.  freq[s] = 2 * c[s], start[s] = 2 * sum of c below s. */
static inline void lmc_counts_model(const uint32_t* cnt, uint32_t R, uint32_t* c) {
  for (uint32_t s = 0; s < R; s++) c[s] = cnt[s];
  for (uint32_t s = 0; s < R; s++)
    if (cnt[s] >= LMC_COUNTS_T) { c[s] = LMC_COUNTS_T - 1u; c[s == 0u ? 1u : 0u] = 1u; break; }
}

/* Reciprocal of the frequency f = 2 * count (count 1 .. 255) for the counts model's encoder:
 *   x / f == mulhi32(x, magic) >> shift   for every x < 2^31,
 * magic = ceil(2^(31 + l) / f), shift = l - 1, l = ceil(log2 f) >= 1 (f is even, so l >= 1).  With
 * e = magic * f - 2^(31 + l) < f <= 2^l the estimate exceeds x / f by x * e / (f * 2^(31 + l)) < 1 / f: its floor
 * is the quotient.  magic lies in [2^31, 2^32). */
static inline void lmc_rans_magic(uint32_t count, uint32_t* magic, uint32_t* shift) {
  uint32_t f = 2u * count, l = 1u;
  while ((1u << l) < f) l++;
  uint64_t num = 1ull << (31u + l);
  *magic = (uint32_t)((num + f - 1u) / f);
  *shift = l - 1u;
}

/* Section offsets for a chunk geometry.  cdf_rows = sum over planes of lmc_cdf_row(bins[p]);
 * pass 31 * P (all planes at 32 bins) for an upper bound. */
static inline void lmc_blob_layout(uint32_t L, uint32_t T, uint32_t H, uint32_t D, uint32_t cdf_rows,
                                   lmc_blob_header* h) {
  uint32_t C = H * D, P = 2u * L, G = (C + LMC_LANES - 1u) / LMC_LANES;
  h->magic = LMC_BLOB_MAGIC;
  h->version = LMC_BLOB_VERSION;
  h->header_bytes = LMC_HEADER_BYTES;
  h->num_layers = L; h->ntokens = T; h->num_heads = H; h->head_size = D;
  h->nchannels = C; h->nplanes = P; h->ngroups = G; h->lp = LMC_LP;
  h->cdf_rows = cdf_rows;
  h->count_bytes = lmc_count_bytes(T);
  h->model = lmc_model_for(T);
  h->off_bins = LMC_HEADER_BYTES;
  h->off_rowpre = h->off_bins + lmc_r16(P);
  h->off_scales = h->off_rowpre + lmc_r16(2u * (P + 1u));
  h->off_scsum = h->off_scales + lmc_r16(2u * P * T);
  h->off_cdf = h->off_scsum + lmc_r16(4u * P);
  h->off_gend = h->off_cdf + lmc_r16(h->count_bytes * C * cdf_rows);
  h->off_streams = h->off_gend + lmc_r16(4u * P * G);
}

/* Capacity (bytes) reserved for one group stream while encoding.  Proof that it
 * cannot overflow is in DESIGN.md ("stream bound"): every occurring symbol has
 * freq >= count*65504/T (CDF16; counts model: exactly count/256), so a lane emits
 * <= T*log2(31)+48 bits < 8*(T+8). */
static inline uint32_t lmc_group_cap_bytes(uint32_t T) {
  return lmc_r16(LMC_LANES * (T + 8u));
}

/* Worst-case blob size for a chunk geometry (every plane at 32 bins). */
static inline uint64_t lmc_blob_bound(uint32_t L, uint32_t T, uint32_t H, uint32_t D) {
  lmc_blob_header h;
  lmc_blob_layout(L, T, H, D, 31u * 2u * L, &h);
  return (uint64_t)h.off_streams + (uint64_t)h.nplanes * h.ngroups * lmc_group_cap_bytes(T);
}


/* ---- pack: the blobs of ONE store call, laid out layer-major for the pinned host tier -------------------------
 *
 * A chunk's blob is chunk-major: header, static sections, then the streams of plane 0, 1, ...  Retrieving a context
 * layer range by layer range from such blobs means one short copy per (chunk, K run, V run, range).  A pack holds the
 * same bytes transposed, so that the streams of a range of LAYERS of all chunks are one contiguous region:
 *
 *   [0, 256)        lmc_pack_header
 *   off_table       uint64 seg_off[2 L n + 1]: segment (layer, kv, chunk) -- index (2 layer + kv) n + chunk -- starts at
 *                   off_streams + seg_off[index]; the last entry is the size of the streams region
 *   off_static      n slots of static_stride bytes: bytes [0, off_streams) of chunk i's blob (header, bins, row prefix,
 *                   scales, checksums, counts, stream directory), unchanged
 *   off_streams     the segments, in table order; segment (layer, kv, chunk) = the streams of plane kv L + layer of
 *                   chunk `chunk`: bytes [S, E) of the blob's streams section, S = r16(gend[p G - 1]) (0 for p = 0),
 *                   E = r16(gend[(p + 1) G - 1])
 * Every offset is a multiple of 16.  The blob of chunk i is recovered byte for byte from its static slot and its 2 L
 * segments (lmc_pack_extract, lmc_hip.h); the pack is written by the GPU (lmc_store_pack) and read by lmc_load_pack. */
#define LMC_PACK_MAGIC 0x4b504d4cu /* "LMPK" */
#define LMC_PACK_VERSION 1u
#define LMC_PACK_HEADER_BYTES 256u
typedef struct lmc_pack_header {
  uint32_t magic;
  uint32_t version;
  uint32_t header_bytes;
  uint32_t nchunks;       /* n */
  uint32_t num_layers;    /* L */
  uint32_t num_heads;     /* H */
  uint32_t head_size;     /* D */
  uint32_t chunk_tokens;  /* tokens of every chunk but (possibly) the last */
  uint32_t ngroups;       /* G */
  uint32_t static_stride; /* bytes per static slot = r16(off_streams of a chunk_tokens-token blob) */
  uint32_t ntokens;       /* tokens of all chunks together */
  uint32_t reserved0;
  uint64_t off_table;
  uint64_t off_static;
  uint64_t off_streams;
  uint64_t total_bytes;   /* off_streams + seg_off[2 L n] */
  uint32_t reserved[44];
} lmc_pack_header;

static inline uint64_t lmc_r16_64(uint64_t x) { return (x + 15u) & ~(uint64_t)15u; }
/* Section offsets of a pack of n chunks (everything but total_bytes, which only the writer knows). */
static inline void lmc_pack_layout(uint32_t n, uint32_t L, uint32_t chunk_tokens, uint32_t H, uint32_t D, uint32_t cdf_rows,
                                   lmc_pack_header* h) {
  lmc_blob_header b;
  lmc_blob_layout(L, chunk_tokens, H, D, cdf_rows, &b);
  h->magic = LMC_PACK_MAGIC; h->version = LMC_PACK_VERSION; h->header_bytes = LMC_PACK_HEADER_BYTES;
  h->nchunks = n; h->num_layers = L; h->num_heads = H; h->head_size = D; h->chunk_tokens = chunk_tokens;
  h->ngroups = b.ngroups; h->static_stride = lmc_r16(b.off_streams);
  h->off_table = LMC_PACK_HEADER_BYTES;
  h->off_static = lmc_r16_64(h->off_table + 8ull * (2ull * L * n + 1ull));
  h->off_streams = h->off_static + (uint64_t)n * h->static_stride;
}
/* Worst-case size of a pack (every plane at 32 bins, every stream at its capacity). */
static inline uint64_t lmc_pack_bound(uint32_t n, uint32_t L, uint32_t chunk_tokens, uint32_t H, uint32_t D) {
  lmc_pack_header h;
  lmc_pack_layout(n, L, chunk_tokens, H, D, 31u * 2u * L, &h);
  return h.off_streams + (uint64_t)n * 2u * L * h.ngroups * lmc_group_cap_bytes(chunk_tokens);
}

#ifdef __cplusplus
}
#endif
#endif /* LMC_FORMAT_H */
