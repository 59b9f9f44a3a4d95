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
 *   scales   u16 [P][T]     per-(plane,token) absmax, raw bits of the KV dtype
 *                           (= max_tensors_key ++ max_tensors_value)
 *   scsum    u32 [P]        checksum of each plane's scales (v4): sum over t of (t + 1) * (bits_t + 1)
 *                           mod 2^32 -- lmc_scale_checksum_term.  The entropy coder detects damage to the
 *                           counts and the streams by itself (the final rANS states must come out at
 *                           their start value); the scales are the one section it cannot see, and a
 *                           flipped scale would silently rescale a whole token row.
 *   gdir     u32 [P][G][2]  stream directory (v6): {beg, end} of group stream (p,g), byte offsets
 *                           relative to the streams section; G = ceil(C/64).  beg is a multiple
 *                           of 16, end is EXACT.
 *   streams  bytes          group streams in (p,g) order.  Every stream owns an ALLOCATION
 *                           alloc(p,g), a multiple of 16, and begins at the sum of the allocations
 *                           of the streams before it; the bytes between its end and the end of
 *                           its allocation are zero.
 *                             LMC_MODEL_COUNTS  alloc = lmc_counts_alloc_bytes(head, words): an UPPER
 *                                               BOUND of the stream's length that follows from the
 *                                               channels' symbol counts alone (lmc_counts_bits);
 *                             LMC_MODEL_CDF16   alloc = r16(its exact length).
 *                           Why a bound: with the allocation known BEFORE the coder runs, the
 *                           encoder's single-pass prefix over the chunk's streams runs in front of
 *                           the coding pass and every 256-byte piece of a stream goes straight to
 *                           its final place -- no padded scratch, no second pass that moves the
 *                           streams (v5 wrote every stream twice and read it once in between).  The
 *                           price is the slack of the bound: < 1 % of the blob.
 *
 * Group stream (p,g) = the symbol statistics of its 64 channels ("lanes"), then their symbols as
 * one interleaved rANS stream:
 *
 *   [ head: widths u8[R8] | planes u64[W] | zeros to 16 B ][ u16 words ][ u32 state[64] ]
 *
 * head (v6; v1-v5 kept a [P][R][C] counts / CDF section of its own: 15 % of a blob, and every byte of it had
 * to be there before the first stream could be decoded).  R = bins - 1 symbols the plane's quantiser can
 * emit, R8 = R rounded up to 8.  The STORED count of (symbol i, lane) is the number of the channel's
 * tokens with that symbol -- for T <= 256 a count of 256 is stored as 255 (the counts of a channel sum
 * to T: a reader adds T - sum to the entry that reads 255); lanes whose channel is >= C store 0.
 * widths[i] = number of significant bits of the largest stored count of symbol i over the 64 lanes
 * (0: the symbol does not occur in the group; <= 16), widths[R .. R8) = 0.  The counts are stored
 * bit-sliced: W = sum of the widths planes of 8 bytes, symbol 0's first, a symbol's most
 * significant bit first; bit l of a plane = that bit of lane l's stored count.  (A wave writes a
 * plane with one ballot and reads its lane's bit with one add-with-carry; a symbol that occurs
 * 40 times at most costs 6 bits per channel instead of 8, one that does not occur costs nothing:
 * the counts of a Llama-3-8B chunk take 0.44 MB instead of 1.18.)  The CDF of a channel is
 *   cdf[i] = RNE(N_i * 65504 / T) + i  (mod 2^16),  N_i = number of its symbols < i,  i = 0 .. 32
 * -- exactly the values of the reference's `cdf` tensor [2L, C, 33] (cachegen_encoder.py:95-126,
 * 175-222; entries above R come out as 65504 + i because N_i = T there).
 *
 * words / states: interleaved rANS over the 64 lanes, 32-bit states, 16-bit renormalisation words in
 * the order the ENCODER emitted them.  The encoder walks tokens T-1 .. 0; at each token the lanes that must
 * renormalise append their low 16 state bits in ascending lane order.  The
 * decoder starts from the tail (states), walks tokens 0 .. T-1 and pops words
 * from the end, again in ascending lane order inside one token step.
 *
 * header.model says which probabilities the coder runs on (v5); both are
 * functions of the counts in the stream's head alone:
 *
 *   LMC_MODEL_CDF16 (0)  any T.  The reference's 16-bit CDF itself:
 *       start = cdf[s], freq = cdf[s+1] - cdf[s], total 2^16, state in
 *       [2^16, 2^32):   emit while x >= freq << 16;
 *       x = (x / freq << 16) + x % freq + start.
 *   LMC_MODEL_COUNTS (1) 2 <= T <= 256 (every chunk of the reference's chunk size 256 and below, a ragged last
 *       chunk included; round 5 -- rounds 3-4 coded T == 256 only this way and every other length on CDF16).
 *       MODEL counts n[s] that sum to 2^8 (lmc_counts_model): for T == 256 the channel's symbol counts themselves;
 *       for T < 256 the counts scaled by 256 / T with the rounding carried along the cumulative sum,
 *       n[s] = floor(256 C_s / T) - floor(256 C_(s-1) / T), C_s = number of tokens with a symbol <= s -- every
 *       occurring symbol keeps n >= its count >= 1, absent symbols keep 0.  Then
 *       freq = 2 * n, start = 2 * (sum of n below), total 2^9
 *       (the factor 2 keeps every frequency >= 2, which lets the encoder divide
 *       with one multiply-high by a 32-bit reciprocal: lmc_rans_magic), state in
 *       [2^15, 2^31):  emit while x >= freq << 22;
 *       x = (x / freq << 9) + x % freq + start.
 *       A channel whose symbols are all equal would need n = 256: it is
 *       coded with 255 and a count of 1 for symbol 0 (for symbol 1 if the
 *       channel's own symbol is 0) -- lmc_counts_model.  The code length is the
 *       channel's empirical entropy to within the 31 bits of final state (T < 256: plus the
 *       rounding of the scaled counts, < 0.5 % on 236-token chunks).
 * See DESIGN.md "Entropy coder" for the recurrences and their bounds.
 */
#ifndef LMC_FORMAT_H
#define LMC_FORMAT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMC_BLOB_MAGIC 0x31434D4Cu /* "LMC1" */
#define LMC_BLOB_VERSION 6u
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
#define LMC_COUNTS_T 256u          /* longest chunk the counts model codes (its probabilities are out of 2^8) ... */
#define LMC_COUNTS_T_MIN 2u        /* ... and the shortest (T = 1 has no 32-bit reciprocal; it stays on CDF16) */
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
  uint32_t zero13;       /* (v1-v5: off_cdf, the counts / CDF section; v6 keeps the counts in the streams' heads) */
  uint32_t off_gdir;     /* stream directory {beg, end} */
  uint32_t off_streams;
  uint32_t stream_bytes; /* size of the streams section = sum of the streams' allocations */
  uint32_t total_bytes;  /* off_streams + stream_bytes */
  uint32_t zero18[3];    /* (v2-v5: off_rowpre, cdf_rows, count_bytes) */
  uint32_t off_scsum;    /* per-plane scale checksums */
  uint32_t model;        /* LMC_MODEL_*: = lmc_model_for(ntokens) */
  uint32_t reserved[9];
} lmc_blob_header;

static inline uint32_t lmc_r16(uint32_t x) { return (x + 15u) & ~15u; }

/* One term of a plane's scale checksum: token t (0-based) whose scale has the raw bits `bits`. */
static inline uint32_t lmc_scale_checksum_term(uint32_t t, uint32_t bits) { return (t + 1u) * (bits + 1u); }

/* Symbols the quantiser of a plane with `bins` bins can emit: 0 .. bins - 2. */
static inline uint32_t lmc_cdf_row(uint32_t bins) { return bins - 1u; }

/* ---- stream head (the symbol counts of a group's 64 channels, bit-sliced) ------------------------------------- */
/* A count as it is stored in a chunk of T tokens. */
static inline uint32_t lmc_stored_count(uint32_t count, uint32_t T) { return (T <= 256u && count > 255u) ? 255u : count; }
/* Significant bits of v (0 for 0). */
static inline uint32_t lmc_bit_width(uint32_t v) { uint32_t w = 0; while (v) { w++; v >>= 1; } return w; }
/* Bytes of the head of a stream of a plane with R symbols whose widths sum to W. */
static inline uint32_t lmc_head_bytes(uint32_t R, uint32_t W) { return lmc_r16(((R + 7u) & ~7u) + 8u * W); }
/* ... and its largest possible value in a chunk of T tokens (31 symbols, every width at its maximum). */
static inline uint32_t lmc_head_cap_bytes(uint32_t T) { return lmc_head_bytes(31u, 31u * lmc_bit_width(lmc_stored_count(T, T))); }

/* The coder model of a chunk of T tokens. */
static inline uint32_t lmc_model_for(uint32_t T) { return (T >= LMC_COUNTS_T_MIN && T <= LMC_COUNTS_T) ? LMC_MODEL_COUNTS : LMC_MODEL_CDF16; }
/* ... and what a DECODER accepts in a header's `model` word: lmc_model_for(T) is the encoder's choice, not a law of the
 * format.  CDF16 decodes at any T (rounds 3-4 wrote their ragged and < 256-token chunks that way, and such blobs outlive a
 * build in a remote store); COUNTS needs 2 <= T <= 256. */
static inline int lmc_model_valid(uint32_t model, uint32_t T) {
  return model == LMC_MODEL_CDF16 || (model == LMC_MODEL_COUNTS && T >= LMC_COUNTS_T_MIN && T <= LMC_COUNTS_T);
}

/* floor(256 * x / T) for x <= T <= 256 the way the kernels compute it: one multiply-high by ceil(2^32 / T).  Exact: the
 * estimate exceeds 256 x / T by 256 x e / (T 2^32) with e = ceil(2^32 / T) T - 2^32 < T, i.e. by less than 2^-16 < 1 / T. */
static inline uint32_t lmc_counts_scale_magic(uint32_t T) { return (uint32_t)((0x100000000ull + T - 1u) / T); }  /* T >= 2 */
static inline uint32_t lmc_counts_scaled(uint32_t x, uint32_t magic) { return (uint32_t)(((uint64_t)(x << 8) * magic) >> 32); }

/* LMC_MODEL_COUNTS: the counts the coder runs on, from a channel's exact counts cnt[0 .. R) (sum T, 2 <= T <= 256,
 * R >= 2): n[s] = floor(256 C_s / T) - floor(256 C_(s-1) / T) over the cumulative counts C (T == 256: cnt itself);
 * then a count of 256 becomes 255 and symbol 0 (symbol 1 if the channel's only symbol is 0) gets 1.
 * freq[s] = 2 * c[s], start[s] = 2 * sum of c below s. */
static inline void lmc_counts_model(const uint32_t* cnt, uint32_t R, uint32_t T, uint32_t* c) {
  const uint32_t magic = lmc_counts_scale_magic(T);
  uint32_t cum = 0, prev = 0;
  for (uint32_t s = 0; s < R; s++) {
    cum += cnt[s];
    const uint32_t now = T == LMC_COUNTS_T ? cum : lmc_counts_scaled(cum, magic);
    c[s] = now - prev;
    prev = now;
  }
  for (uint32_t s = 0; s < R; s++)
    if (c[s] >= LMC_COUNTS_T) { c[s] = LMC_COUNTS_T - 1u; c[s == 0u ? 1u : 0u] = 1u; break; }
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

/* ---- LMC_MODEL_COUNTS: how long can a stream get?  (v6: streams are placed BEFORE they are coded) ----------------
 * A lane starts at x = 2^15 and ends at x >= 2^15; a renormalisation takes 16 bits out, and coding a symbol of model
 * count c (frequency f = 2 c of 2^9) takes the state from x_r to x' < 2^9 (x_r / f + 1) = (2^9 / f) x_r (1 + f / x_r):
 *   16 * words <= sum over tokens of [ log2(256 / c) + log2(1 + f / x_r) ].
 * x_r >= 2^15 on a step without renormalisation, x_r >= f * 2^6 on a step with one (the emit test is x >= f << 22), so
 *   words <= floor( ( sum_s c_s (log2(256 / c_s) + log2(1 + 2 c_s / 2^15)) + words * log2(1 + 2^-6) ) / 16 ).
 * lmc_counts_bits[c] = the first bracket for a symbol that occurs c times, in units of 2^-8 bit, rounded up (+1);
 * log2(1 + 2^-6) < 6 * 2^-8.  The table is a constant of the FORMAT (written out so that every implementation
 * uses the same integers; tests/test_oracle_golden.py recomputes it and attacks the bound with adversarial
 * channels).  Index 0 and 256 are never used by a model (lmc_counts_model). */
#define LMC_COUNTS_BITS_LIST \
  0, 2050, 3586, 4928, 6146, 7270, 8320, 9308, 10243, 11132, 11980, 12790, 13568, 14314, 15032, 15724, 16391, \
  17035, 17658, 18260, 18842, 19406, 19953, 20483, 20996, 21495, 21979, 22448, 22904, 23347, 23777, 24195, 24601, \
  24995, 25378, 25751, 26113, 26464, 26806, 27138, 27461, 27774, 28079, 28375, 28662, 28941, 29212, 29474, 29729, \
  29977, 30216, 30449, 30674, 30892, 31104, 31308, 31506, 31697, 31882, 32060, 32233, 32399, 32559, 32713, 32862, \
  33004, 33141, 33273, 33399, 33520, 33635, 33745, 33850, 33950, 34045, 34135, 34220, 34300, 34375, 34446, 34512, \
  34574, 34631, 34684, 34732, 34776, 34815, 34850, 34882, 34908, 34931, 34950, 34965, 34976, 34982, 34985, 34985, \
  34980, 34972, 34959, 34944, 34924, 34901, 34874, 34844, 34810, 34773, 34732, 34688, 34641, 34590, 34536, 34479, \
  34418, 34354, 34287, 34217, 34144, 34067, 33988, 33905, 33820, 33731, 33639, 33545, 33447, 33347, 33244, 33137, \
  33028, 32917, 32802, 32685, 32564, 32441, 32316, 32188, 32057, 31923, 31787, 31648, 31506, 31362, 31216, 31067, \
  30915, 30761, 30604, 30445, 30284, 30120, 29953, 29785, 29613, 29440, 29264, 29086, 28905, 28722, 28537, 28350, \
  28160, 27968, 27774, 27577, 27379, 27178, 26975, 26770, 26562, 26353, 26141, 25928, 25712, 25494, 25274, 25052, \
  24828, 24602, 24373, 24143, 23911, 23677, 23441, 23203, 22962, 22720, 22476, 22230, 21983, 21733, 21481, 21228, \
  20972, 20715, 20456, 20195, 19932, 19667, 19401, 19132, 18862, 18590, 18317, 18041, 17764, 17485, 17204, 16922, \
  16637, 16351, 16064, 15774, 15483, 15191, 14896, 14600, 14302, 14003, 13702, 13399, 13095, 12789, 12481, 12172, \
  11861, 11549, 11235, 10919, 10602, 10283, 9963, 9641, 9318, 8993, 8666, 8338, 8009, 7678, 7345, 7011, 6676, \
  6339, 6000, 5660, 5319, 4976, 4631, 4286, 3938, 3590, 3239, 2888, 2535, 2180, 1825, 0
static const uint16_t lmc_counts_bits[257] = {LMC_COUNTS_BITS_LIST};

/* T < 256: a symbol with model count n occurs cnt <= n times, and each occurrence costs the bracket above with c = n:
 * lmc_counts_bpo[n] = ceil(256 (log2(256 / n) + log2(1 + 2 n / 2^15))) + 1 (units of 2^-8 bit per OCCURRENCE), and
 * S = sum over the symbols of cnt[s] * lmc_counts_bpo[n[s]].  (T == 256 keeps lmc_counts_bits -- per symbol, rounded once
 * -- so that its blobs are byte for byte those of rounds 3-4.) */
#define LMC_COUNTS_BPO_LIST \
  0, 2050, 1794, 1644, 1538, 1455, 1388, 1331, 1282, 1238, 1199, 1164, 1132, 1102, 1075, 1050, 1026, 1003, 982, \
  962, 944, 926, 908, 892, 876, 861, 847, 833, 819, 807, 794, 782, 770, 759, 748, 737, 727, 717, 707, 697, 688, \
  679, 670, 661, 653, 645, 637, 629, 621, 613, 606, 599, 591, 584, 577, 571, 564, 558, 551, 545, 539, 533, 527, \
  521, 515, 509, 504, 498, 493, 487, 482, 477, 472, 467, 462, 457, 452, 447, 442, 438, 433, 428, 424, 419, 415, \
  411, 406, 402, 398, 394, 390, 386, 382, 378, 374, 370, 366, 362, 358, 355, 351, 347, 344, 340, 337, 333, 330, \
  326, 323, 319, 316, 313, 309, 306, 303, 300, 296, 293, 290, 287, 284, 281, 278, 275, 272, 269, 266, 263, 260, \
  258, 255, 252, 249, 246, 244, 241, 238, 235, 233, 230, 228, 225, 222, 220, 217, 215, 212, 210, 207, 205, 202, \
  200, 197, 195, 193, 190, 188, 186, 183, 181, 179, 176, 174, 172, 170, 167, 165, 163, 161, 159, 157, 154, 152, \
  150, 148, 146, 144, 142, 140, 138, 136, 134, 132, 130, 128, 126, 124, 122, 120, 118, 116, 114, 112, 110, 108, \
  106, 105, 103, 101, 99, 97, 95, 94, 92, 90, 88, 86, 85, 83, 81, 79, 78, 76, 74, 72, 71, 69, 67, 66, 64, 62, 61, \
  59, 57, 56, 54, 53, 51, 49, 48, 46, 45, 43, 41, 40, 38, 37, 35, 34, 32, 31, 29, 28, 26, 25, 23, 22, 20, 19, 17, \
  16, 14, 13, 12, 10, 9, 0
static const uint16_t lmc_counts_bpo[257] = {LMC_COUNTS_BPO_LIST};

/* S of one channel: exact counts cnt, model counts n (lmc_counts_model), chunk length T. */
static inline uint32_t lmc_counts_lane_S(const uint32_t* cnt, const uint32_t* n, uint32_t R, uint32_t T) {
  uint32_t S = 0;
  for (uint32_t s = 0; s < R; s++) S += T == LMC_COUNTS_T ? lmc_counts_bits[n[s]] : cnt[s] * lmc_counts_bpo[n[s]];
  return S;
}

/* Upper bound of the 16-bit words one lane emits, from S = sum over the symbols of lmc_counts_bits[model count]. */
static inline uint32_t lmc_counts_lane_words(uint32_t S) { return (S + 6u * ((S >> 12) + 2u)) >> 12; }

/* Allocation of a counts-model stream with a head of `head` bytes whose active lanes may emit `words` words in all
 * (+ the 64 final states). */
static inline uint32_t lmc_counts_alloc_bytes(uint32_t head, uint32_t words) { return lmc_r16(head + 2u * (words + 2u * LMC_LANES)); }

/* Section offsets for a chunk geometry (they do not depend on the data). */
static inline void lmc_blob_layout(uint32_t L, uint32_t T, uint32_t H, uint32_t D, lmc_blob_header* h) {
  uint32_t C = H * D, P = 2u * L, G = (C + LMC_LANES - 1u) / LMC_LANES;
  h->magic = LMC_BLOB_MAGIC;
  h->version = LMC_BLOB_VERSION;
  h->header_bytes = LMC_HEADER_BYTES;
  h->num_layers = L; h->ntokens = T; h->num_heads = H; h->head_size = D;
  h->nchannels = C; h->nplanes = P; h->ngroups = G; h->lp = LMC_LP;
  h->model = lmc_model_for(T);
  h->zero13 = 0; h->zero18[0] = h->zero18[1] = h->zero18[2] = 0;
  h->off_bins = LMC_HEADER_BYTES;
  h->off_scales = h->off_bins + lmc_r16(P);
  h->off_scsum = h->off_scales + lmc_r16(2u * P * T);
  h->off_gdir = h->off_scsum + lmc_r16(4u * P);
  h->off_streams = h->off_gdir + lmc_r16(8u * P * G);
}

/* Capacity (bytes) reserved for one group stream: the largest head, and for words and states -- proof that
 * they cannot overflow is in DESIGN.md ("stream bound"): every occurring symbol has
 * freq >= count*65504/T (CDF16; counts model: exactly count/256), so a lane emits
 * <= T*log2(31)+48 bits < 8*(T+8). */
static inline uint32_t lmc_group_cap_bytes(uint32_t T) {
  return lmc_head_cap_bytes(T) + lmc_r16(LMC_LANES * (T + 8u));
}

/* Worst-case blob size for a chunk geometry. */
static inline uint64_t lmc_blob_bound(uint32_t L, uint32_t T, uint32_t H, uint32_t D) {
  lmc_blob_header h;
  lmc_blob_layout(L, T, H, D, &h);
  return (uint64_t)h.off_streams + (uint64_t)h.nplanes * h.ngroups * lmc_group_cap_bytes(T);
}


/* ---- pack: the blobs of ONE store call, laid out plane-major for the pinned host tier -------------------------
 *
 * A chunk's blob is chunk-major: header, static sections, then the streams of plane 0, 1, ...  Retrieving a context
 * layer range by layer range from such blobs means one short copy per (chunk, K run, V run, range).  A pack holds the
 * same bytes transposed, so that the streams of a range of LAYERS of all chunks are two contiguous regions (their K
 * planes, their V planes):
 *
 *   [0, 256)        lmc_pack_header
 *   off_table       uint64 seg_off[2 L n + 1]: segment (plane, chunk) -- index p n + chunk, p = kv L + layer: the blob's
 *                   plane order, K planes of every layer, then V planes -- starts at off_streams + seg_off[index]; the
 *                   last entry is the size of the streams region
 *   off_static      n slots of static_stride bytes: bytes [0, off_streams) of chunk i's blob (header, bins, scales,
 *                   checksums, stream directory), unchanged
 *   off_streams     the segments, in table order; segment (p, chunk) = the streams of plane p of chunk `chunk`: bytes
 *                   [S, E) of the blob's streams section, S = beg of stream (p, 0), E = beg of stream (p + 1, 0) (the end
 *                   of the section for the last plane)
 * Every offset is a multiple of 16.  The blob of chunk i is recovered byte for byte from its static slot and its 2 L
 * segments (lmc_pack_extract, lmc_hip.h); the pack is written by the GPU (lmc_store_pack) and read by lmc_load_pack.
 * Version 3 (round 6): plane order.  Version 2 ordered the segments (layer, K/V, chunk) -- one region per range of
 * layers, but a region that is complete only when the encoder has reached the layers' V planes, i.e. at the very end;
 * in plane order the K half of a range is final after a fraction of the encode, and a store that launches its encode
 * in plane ranges (lmc_store_pack_parts) sends each range over PCIe while the later planes are still being coded. */
#define LMC_PACK_MAGIC 0x4b504d4cu /* "LMPK" */
#define LMC_PACK_VERSION 3u
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
static inline void lmc_pack_layout(uint32_t n, uint32_t L, uint32_t chunk_tokens, uint32_t H, uint32_t D, lmc_pack_header* h) {
  lmc_blob_header b;
  lmc_blob_layout(L, chunk_tokens, H, D, &b);
  h->magic = LMC_PACK_MAGIC; h->version = LMC_PACK_VERSION; h->header_bytes = LMC_PACK_HEADER_BYTES;
  h->nchunks = n; h->num_layers = L; h->num_heads = H; h->head_size = D; h->chunk_tokens = chunk_tokens;
  h->ngroups = b.ngroups; h->static_stride = lmc_r16(b.off_streams);
  h->off_table = LMC_PACK_HEADER_BYTES;
  h->off_static = lmc_r16_64(h->off_table + 8ull * (2ull * L * n + 1ull));
  h->off_streams = h->off_static + (uint64_t)n * h->static_stride;
}
/* Worst-case size of a pack (every stream at its capacity). */
static inline uint64_t lmc_pack_bound(uint32_t n, uint32_t L, uint32_t chunk_tokens, uint32_t H, uint32_t D) {
  lmc_pack_header h;
  lmc_pack_layout(n, L, chunk_tokens, H, D, &h);
  return h.off_streams + (uint64_t)n * 2u * L * h.ngroups * lmc_group_cap_bytes(chunk_tokens);
}

#ifdef __cplusplus
}
#endif
#endif /* LMC_FORMAT_H */
