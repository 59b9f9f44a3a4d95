// k_copy.h -- strided / paged 16-byte-vector copy between two KV layouts
// (SURVEY.md section 8a rows a3, a4, a18, a19).
//
// Replaces _tuple_kv_to_blob + _slice_kv_at (cache_engine.py:98-161),
// torch.cat + _blob_to_tuple_kv (:120-129, 362-368) and the external
// connector's slot_mapping gather / reshape_and_cache_flash scatter
// (docs/source/developer_tutorial/LLM_Engine.rst:91-122) with one pass.
// One thread moves one run of 8 channels (16 B) of one (plane, token).
#pragma once
#include "lmc_device.h"

struct CopyArgs {
  KvAddr src, dst;
  int tok_begin, ntok, dst_tok0;
  int P, C;
  long long nvec;  // P * ntok * C/8 (k_copy_kv) or P * ntok * C (k_copy_kv_elem)
};

__global__ __launch_bounds__(256) void k_copy_kv(CopyArgs a) {
  const int C8 = a.C >> 3;
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < a.nvec; id += (long long)gridDim.x * 256) {
    const int j = (int)(id % C8);
    const long long r = id / C8;
    const int t = (int)(r % a.ntok);
    const int p = (int)(r / a.ntok);
    const int c0 = j * 8;
    const int hs = c0 / a.src.D, ds = c0 - hs * a.src.D;
    const u16* sp = lmc_plane_base(a.src, p) + lmc_tok_off(a.src, a.tok_begin + t) + (long long)hs * a.src.stride_head + ds;
    u16* dp = const_cast<u16*>(lmc_plane_base(a.dst, p)) + lmc_tok_off(a.dst, a.dst_tok0 + t) +
              (long long)hs * a.dst.stride_head + ds;
    st_global_u4(dp, ld_global_u4(sp));
  }
}

// The same copy one ELEMENT per thread: layouts whose rows are not on 16-byte boundaries, or whose head_size is no
// multiple of 8 while the heads of a token row are apart (the reference's serde takes any shape:
// cachegen_encoder.py:40-61, 76-91).  Not a fast path: the Python side uses it to bring such a range into a contiguous
// chunk the encoders can read.
__global__ __launch_bounds__(256) void k_copy_kv_elem(CopyArgs a) {
  for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < a.nvec; id += (long long)gridDim.x * 256) {
    const int ch = (int)(id % a.C);
    const long long r = id / a.C;
    const int t = (int)(r % a.ntok);
    const int p = (int)(r / a.ntok);
    const int hs = ch / a.src.D, ds = ch - hs * a.src.D;
    const u16* sp = lmc_plane_base(a.src, p) + lmc_tok_off(a.src, a.tok_begin + t) + (long long)hs * a.src.stride_head + ds;
    u16* dp = const_cast<u16*>(lmc_plane_base(a.dst, p)) + lmc_tok_off(a.dst, a.dst_tok0 + t) +
              (long long)hs * a.dst.stride_head + ds;
    *dp = *sp;
  }
}
