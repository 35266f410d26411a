// Paged (token-granular) decode attention and its metadata kernels.
//
// Data layout in HBM (the reference's pool, python/sglang/srt/mem_cache/memory_pool.py:277-296):
//   k_buffer [N, Hkv, D], v_buffer [N, Hkv, Dv] -- one row per token slot, slot 0 a dummy row;
//   kv_indices[kv_indptr[b] .. kv_indptr[b+1]) lists request b's slots in sequence order.
//
// Fast path (16-bit types, D = Dv = 128, any GQA group): one wave per (request, kv head, split).
// The wave walks its slots in tiles of 32 tokens: each K/V row (256 B) is fetched by 16 lanes as
// one contiguous 16-byte piece per lane (4 whole rows per wave instruction), parked in registers
// while the previous tile is being multiplied, then written to a wave-private, XOR-swizzled LDS
// tile.  S^T = K.Q^T and O^T += V^T.P^T run on the matrix cores (v_mfma_f32_16x16x32): the K tile
// is read by rows (ds_read_b128), the V tile by columns (ds_read_b64_tr_b16), and the S^T
// accumulator registers are, as they stand, the P^T operand of the second product.  Softmax
// statistics live one head per lane (online max / sum, exp2 domain), combined across the four
// 16-lane groups with wave shuffles.
//
// Everything else (fp32, odd head sizes, MLA 576/512) takes the generic wave-per-(request, head,
// split) kernel.  Partials are merged by log-sum-exp in a second kernel, as in the reference
// (triton_ops/decode_attention.py:497-557).
#include <type_traits>

#include "attention_tile.h"
#include "common.h"

namespace sglk {

static constexpr int kMinBlockKV = 32;  // decode_attention.py:40 (_MIN_BLOCK_KV)

__host__ __device__ __forceinline__ int cdiv_i(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// metadata
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) create_kv_indices_kernel(const int32_t* __restrict__ req_to_token,
                                                                int64_t row_stride, const void* req_pool_indices,
                                                                int idx_is64, const void* lens, int lens_is64,
                                                                const int32_t* __restrict__ kv_indptr,
                                                                const void* kv_start, int start_is64,
                                                                int32_t* __restrict__ kv_indices) {
  const int b = blockIdx.x;
  const int64_t row = ld_idx(req_pool_indices, idx_is64, b);
  const int64_t n = ld_idx(lens, lens_is64, b);
  const int64_t st = kv_start ? ld_idx(kv_start, start_is64, b) : 0;
  const int32_t* src = req_to_token + row * row_stride + st;
  int32_t* dst = kv_indices + kv_indptr[b];
  for (int64_t j = (int64_t)blockIdx.y * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.y * 256) dst[j] = src[j];
}

// inclusive scan of up to any bs by one block (bs is a batch size: at most a few thousand)
__global__ void __launch_bounds__(256) cumsum_kernel(const void* seq_lens, int lens_is64, int32_t* kv_indptr, int bs) {
  __shared__ int32_t part[256];
  const int per = (bs + 255) / 256;
  const int lo = threadIdx.x * per, hi = min(lo + per, bs);
  int32_t s = 0;
  for (int i = lo; i < hi; ++i) s += (int32_t)ld_idx(seq_lens, lens_is64, i);
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t run = 0;
    for (int i = 0; i < 256; ++i) {
      const int32_t v = part[i];
      part[i] = run;
      run += v;
    }
    kv_indptr[0] = 0;
  }
  __syncthreads();
  int32_t run = part[threadIdx.x];
  for (int i = lo; i < hi; ++i) {
    run += (int32_t)ld_idx(seq_lens, lens_is64, i);
    kv_indptr[i + 1] = run;
  }
}

// get_num_kv_splits_triton (triton_backend.py:23-72): one block, integer arithmetic except the
// log2 growth heuristic which the reference evaluates in fp32.
__device__ __forceinline__ void num_kv_splits_body(int32_t* num_kv_splits, const void* seq_lens, int lens_is64,
                                                   int num_seq, int num_group, int num_head, int num_kv_head,
                                                   int max_kv_splits, int device_core_count, int* red) {
  int mx = 0;
  for (int i = threadIdx.x; i < num_seq; i += blockDim.x) mx = max(mx, (int)ld_idx(seq_lens, lens_is64, i));
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  const int max_len = red[0];
  __syncthreads();
  int mn = max_len;
  for (int i = threadIdx.x; i < num_seq; i += blockDim.x) mn = min(mn, (int)ld_idx(seq_lens, lens_is64, i));
  red[threadIdx.x] = mn;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = min(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  int min_len = red[0];
  __syncthreads();
  if ((int64_t)max_len * 8 < (int64_t)min_len * 10) min_len = max_len;
  if (min_len < 1) min_len = 1;
  const int splits1 = min(cdiv_i(max_len, min_len), max_kv_splits);
  const int chunk1 = max(cdiv_i(max_len, max(splits1, 1)), 1);
  const float ext_seq_len = (float)max_len / 64.0f;
  const float lg = fmaxf(log2f(ext_seq_len), 1.0f);
  const int ext_cores = (int)((float)device_core_count * lg);
  int block_h = 16;
  const int group = num_head / num_kv_head;
  int64_t token_grid;
  if (group == 1) {
    token_grid = (int64_t)num_seq * num_group * num_head;
  } else {
    block_h = min(block_h, group);
    token_grid = (int64_t)num_seq * num_group * cdiv_i(num_head, block_h);
  }
  const int splits2 = (int)min((int64_t)((ext_cores + token_grid - 1) / token_grid), (int64_t)max_kv_splits);
  const int chunk2 = max(cdiv_i(max_len, max(splits2, 1)), 1);
  for (int i = threadIdx.x; i < num_seq; i += blockDim.x) {
    const int L = (int)ld_idx(seq_lens, lens_is64, i);
    const int v = max(cdiv_i(L, chunk1), cdiv_i(L, chunk2));
    for (int g = 0; g < num_group; ++g) num_kv_splits[i * num_group + g] = v;
  }
}

__global__ void __launch_bounds__(256) num_kv_splits_kernel(int32_t* num_kv_splits, const void* seq_lens,
                                                            int lens_is64, int num_seq, int num_group, int num_head,
                                                            int num_kv_head, int max_kv_splits, int device_core_count) {
  __shared__ int red[256];
  num_kv_splits_body(num_kv_splits, seq_lens, lens_is64, num_seq, num_group, num_head, num_kv_head, max_kv_splits,
                     device_core_count, red);
}

// Longest common slot prefix of the batch (one group = whole batch). out[0] is pre-set by the launcher's
// memset-free protocol: block 0 thread 0 seeds nothing; every block atomically lowers out[0].
__global__ void __launch_bounds__(256) common_prefix_kernel(const int32_t* __restrict__ req_to_token,
                                                            int64_t row_stride, const void* req_pool_indices,
                                                            int idx_is64, const void* seq_lens, int lens_is64, int bs,
                                                            int32_t* out) {
  // one block per request: first mismatch against request 0
  __shared__ int red[256];
  const int b = blockIdx.x;
  const int64_t r0 = ld_idx(req_pool_indices, idx_is64, 0), rb = ld_idx(req_pool_indices, idx_is64, b);
  const int len0 = (int)ld_idx(seq_lens, lens_is64, 0), lenb = (int)ld_idx(seq_lens, lens_is64, b);
  const int n = min(len0, lenb) - 1;  // the newest token is never shared
  const int32_t* a = req_to_token + r0 * row_stride;
  const int32_t* c = req_to_token + rb * row_stride;
  int first = n < 0 ? 0 : n;
  for (int j = threadIdx.x; j < n; j += 256) {
    if (a[j] != c[j]) {
      first = j;
      break;
    }
  }
  red[threadIdx.x] = first;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = min(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMin(out, red[0]);
}

__global__ void set_i32_kernel(int32_t* p, int32_t v) { *p = v; }
__global__ void __launch_bounds__(256) fill_i32_kernel(int32_t* p, int32_t v, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

// kv_indptr, num_kv_splits and kv_indices in one launch: block b finds its own offset (the sum of the lengths
// before it) and gathers its row; block 0 also writes the offsets and the split counts.
__global__ void __launch_bounds__(256) decode_metadata_kernel(const int32_t* __restrict__ req_to_token, int64_t row_stride,
                                                              const void* req_pool_indices, int idx_is64,
                                                              const void* seq_lens, int lens_is64,
                                                              int32_t* __restrict__ kv_indptr,
                                                              int32_t* __restrict__ kv_indices,
                                                              int32_t* __restrict__ num_kv_splits, int bs, int num_head,
                                                              int num_kv_head, int max_kv_splits, int device_core_count) {
  __shared__ int red[256];
  const int b = blockIdx.x;
  int s = 0;
  for (int i = threadIdx.x; i < b; i += 256) s += (int)ld_idx(seq_lens, lens_is64, i);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const int start = red[0];
  __syncthreads();
  const int64_t row = ld_idx(req_pool_indices, idx_is64, b);
  const int64_t n = ld_idx(seq_lens, lens_is64, b);
  const int32_t* src = req_to_token + row * row_stride;
  int32_t* dst = kv_indices + start;
  for (int64_t j = (int64_t)blockIdx.y * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.y * 256) dst[j] = src[j];
  if (blockIdx.y == 0 && threadIdx.x == 0) {
    kv_indptr[b + 1] = start + (int)n;
    if (b == 0) kv_indptr[0] = 0;
  }
  if (b == 0 && blockIdx.y == 0)
    num_kv_splits_body(num_kv_splits, seq_lens, lens_is64, bs, 1, num_head, num_kv_head, max_kv_splits, device_core_count,
                       red);
}

// schedule_batch.py:1464-1527 (prepare_for_decode) on device state: the new token's position is the number of
// tokens cached so far, seq_lens grows by one, out_cache_loc is the slot the request table holds for that position
__global__ void __launch_bounds__(256) prepare_for_decode_kernel(const int32_t* __restrict__ req_to_token, int64_t row_stride,
                                                                 const void* req_pool_indices, int idx_is64,
                                                                 void* seq_lens, int lens_is64, int64_t* positions,
                                                                 void* out_cache_loc, int loc_is64, int bs) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= bs) return;
  const int64_t pos = ld_idx(seq_lens, lens_is64, b);
  const int64_t row = ld_idx(req_pool_indices, idx_is64, b);
  const int32_t slot = req_to_token[row * row_stride + pos];
  positions[b] = pos;
  if (lens_is64)
    reinterpret_cast<int64_t*>(seq_lens)[b] = pos + 1;
  else
    reinterpret_cast<int32_t*>(seq_lens)[b] = (int32_t)(pos + 1);
  if (loc_is64)
    reinterpret_cast<int64_t*>(out_cache_loc)[b] = slot;
  else
    reinterpret_cast<int32_t*>(out_cache_loc)[b] = slot;
}

// ---------------------------------------------------------------------------------------------
// generic stage 1: one wave per (request, query head, split); any head size, any float type
// ---------------------------------------------------------------------------------------------
template <typename T, typename KV, int DV_PER_LANE>
__global__ void __launch_bounds__(64) decode_stage1_generic(
    const void* __restrict__ q, const void* __restrict__ k_buffer, const void* __restrict__ v_buffer,
    const int32_t* __restrict__ kv_indptr, const int32_t* __restrict__ kv_indices, float* __restrict__ attn_logits,
    float* __restrict__ attn_lse, const int32_t* __restrict__ num_kv_splits, int max_kv_splits, int Hq, int Hkv, int D,
    int Dv, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n,
    int64_t v_stride_h, float sm_scale, float logit_cap) {
  extern __shared__ float qs[];  // D floats
  const int b = blockIdx.x, h = blockIdx.y, split = blockIdx.z;
  const int lane = threadIdx.x;
  const int kvh = h / (Hq / Hkv);
  const int start = kv_indptr[b];
  const int len = kv_indptr[b + 1] - start;
  const int splits = max(num_kv_splits[b], 1);
  const int per_split = cdiv_i(cdiv_i(len, splits), kMinBlockKV) * kMinBlockKV;
  const int s0 = per_split * split, s1 = min(s0 + per_split, len);
  if (s1 <= s0) return;
  for (int d = lane; d < D; d += 64) qs[d] = ld<T>(q, (int64_t)b * q_stride_b + (int64_t)h * q_stride_h + d);
  __syncthreads();
  float m = -INFINITY, l = 0.f;
  float acc[DV_PER_LANE];
#pragma unroll
  for (int i = 0; i < DV_PER_LANE; ++i) acc[i] = 0.f;
  for (int t0 = s0; t0 < s1; t0 += 64) {
    const int tok = t0 + lane;
    const bool valid = tok < s1;
    const int64_t loc = valid ? (int64_t)kv_indices[start + tok] : 0;
    float s = 0.f;
    if (valid) {
      const int64_t kb = loc * k_stride_n + (int64_t)kvh * k_stride_h;
      for (int d = 0; d < D; ++d) s += qs[d] * KvElem<KV>::ld(k_buffer, kb + d);
      s *= sm_scale;
      if (logit_cap > 0.f) s = logit_cap * tanhf(s / logit_cap);
    } else {
      s = -INFINITY;
    }
    const float m_new = fmaxf(m, wave_max(s));
    const float alpha = expf(m - m_new);
    const float p = valid ? expf(s - m_new) : 0.f;
    l = l * alpha + wave_sum(p);
    m = m_new;
#pragma unroll
    for (int i = 0; i < DV_PER_LANE; ++i) acc[i] *= alpha;
    const int cnt = min(64, s1 - t0);
    for (int j = 0; j < cnt; ++j) {
      const float pj = __shfl(p, j, 64);
      const int64_t lj = __shfl((long long)loc, j, 64);
      const int64_t vb = lj * v_stride_n + (int64_t)kvh * v_stride_h;
#pragma unroll
      for (int i = 0; i < DV_PER_LANE; ++i) {
        const int d = lane + 64 * i;
        if (d < Dv) acc[i] += pj * KvElem<KV>::ld(v_buffer, vb + d);
      }
    }
  }
  const int64_t ob = (((int64_t)b * Hq + h) * max_kv_splits + split);
#pragma unroll
  for (int i = 0; i < DV_PER_LANE; ++i) {
    const int d = lane + 64 * i;
    if (d < Dv) attn_logits[ob * Dv + d] = acc[i] / l;
  }
  if (lane == 0) attn_lse[ob] = m + logf(l);
}

// ---------------------------------------------------------------------------------------------
// stage 2: LSE merge of the split partials (decode_attention.py:497-557)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(128) decode_stage2(const float* __restrict__ attn_logits,
                                                     const float* __restrict__ attn_lse, void* __restrict__ o,
                                                     const int32_t* __restrict__ kv_indptr,
                                                     const int32_t* __restrict__ num_kv_splits, int max_kv_splits,
                                                     int Hq, int Dv, int64_t o_stride_b, int64_t o_stride_h,
                                                     float* __restrict__ xsum_out, int64_t xs_stride,
                                                     const int32_t* __restrict__ gate = nullptr) {
  __shared__ float xs_red[2];
  if (gate && *gate != 0) return;  // see decode_stage1_mfma128
  const int b = blockIdx.x, h = blockIdx.y;
  const int len = kv_indptr[b + 1] - kv_indptr[b];
  const int splits = max(num_kv_splits[b], 1);
  const int per_split = cdiv_i(cdiv_i(len, splits), kMinBlockKV) * kMinBlockKV;
  const int64_t base = ((int64_t)b * Hq + h) * max_kv_splits;
  if (Dv % 512 == 0 && !xsum_out && (reinterpret_cast<uintptr_t>(attn_logits) & 15) == 0) {
    // wide values (absorbed MLA): four consecutive elements a thread, a split's weights formed once per thread; the
    // same operations per element, in the same order, as the loop below
    for (int d = 4 * threadIdx.x; d < Dv; d += 512) {
      float e_sum = 0.f, e_max = -INFINITY;
      float4v acc = float4v{0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < max_kv_splits; ++s) {
        const int s0 = per_split * s, s1 = min(s0 + per_split, len);
        if (s1 > s0) {
          const float4v tv = *reinterpret_cast<const float4v*>(attn_logits + (base + s) * Dv + d);
          const float tl = attn_lse[base + s];
          const float n_max = fmaxf(tl, e_max);
          const float old = expf(e_max - n_max), w = expf(tl - n_max);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = acc[r] * old + w * tv[r];
          e_sum = e_sum * old + w;
          e_max = n_max;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) st<T>(o, (int64_t)b * o_stride_b + (int64_t)h * o_stride_h + d + r, acc[r] / e_sum);
    }
    return;
  }
  for (int d = threadIdx.x; d < Dv; d += 128) {
    float e_sum = 0.f, e_max = -INFINITY, acc = 0.f;
    for (int s = 0; s < max_kv_splits; ++s) {
      const int s0 = per_split * s, s1 = min(s0 + per_split, len);
      if (s1 > s0) {
        const float tv = attn_logits[(base + s) * Dv + d];
        const float tl = attn_lse[base + s];
        const float n_max = fmaxf(tl, e_max);
        const float old = expf(e_max - n_max), w = expf(tl - n_max);
        acc = acc * old + w * tv;
        e_sum = e_sum * old + w;
        e_max = n_max;
      }
    }
    st<T>(o, (int64_t)b * o_stride_b + (int64_t)h * o_stride_h + d, acc / e_sum);
    if (xsum_out) {  // Dv == 128: one head is one 128-element block of the following projection's input row
      const float r = wave_sum(Conv<T>::to_f32(Conv<T>::from_f32(acc / e_sum)));
      if ((threadIdx.x & 63) == 0) xs_red[threadIdx.x >> 6] = r;
      __syncthreads();
      if (threadIdx.x == 0) xsum_out[(int64_t)h * xs_stride + b] = xs_red[0] + xs_red[1];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fast path: D = Dv = 128, 16-bit types, matrix cores, wave-private LDS tiles
// ---------------------------------------------------------------------------------------------
static constexpr int kWavesPerBlock = 4;
static constexpr int kLdsPerWave = 2 * kTileBytes;  // K tile + V tile

// group g with group_indptr[g] <= b < group_indptr[g + 1]
__device__ __forceinline__ int find_group(const int32_t* __restrict__ group_indptr, int num_groups, int b) {
  int lo = 0, hi = num_groups - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (group_indptr[mid] <= b)
      lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}
// split count of the per-request remainder in the shared-prefix path: one 32-token tile per wave while they last
__device__ __forceinline__ int tail_splits(int len, int max_splits) {
  return min(max(cdiv_i(len, 32), 1), max_splits);
}

// Tail blocks of a request's private remainder in the planned shared-prefix kernel: one block of four waves up to 512
// tokens (W's 128-token remainders: one tile per wave), more where the remainder is long -- a request that shares
// nothing walks its whole context here and must not do so on four waves.
// `first_blocks` = the launch's tail blocks at one per remainder: once those alone fill the chip (three blocks of four
// waves per CU), further blocks only add the hand-over between them (measured, 64 requests of 1152 tokens that
// share nothing: 65 us with one block per remainder, 75 us with two).
static constexpr int kTailTokensPerBlock = 512;
static constexpr int kTailBlocksToFill = 768;
__host__ __device__ inline int tail_blocks(int len, int tmax, int first_blocks) {
  int t = (len + kTailTokensPerBlock - 1) / kTailTokensPerBlock;
  const int room = kTailBlocksToFill / (first_blocks > 0 ? first_blocks : 1);
  t = t < room ? t : room;
  return t < 1 ? 1 : (t < tmax ? t : tmax);
}

// pair64 > 0: head size 64.  Two neighbouring kv heads of 64 elements are one 128-element pool row piece (the launcher
// checks that they are contiguous) and are walked as ONE kv head of 128: Hkv counts such pairs, a pair's query heads are
// the pair64 heads of its first kv head followed by the pair64 heads of its second.  A query row is placed in the half
// of the 128-wide operand that faces its own kv head, the other half is zero, so S is exactly q . k of its own head;
// O = P . [v_first, v_second] holds the head's output in the same half, which alone is stored (rows of 64).
template <typename T, typename KV = T>
__global__ void __launch_bounds__(256, 2) decode_stage1_mfma128(
    const void* __restrict__ q, const void* __restrict__ k_buffer, const void* __restrict__ v_buffer,
    const int32_t* __restrict__ kv_indptr, const int32_t* __restrict__ kv_indices, float* __restrict__ attn_logits,
    float* __restrict__ attn_lse, const int32_t* __restrict__ num_kv_splits, int max_kv_splits, int Hq, int Hkv,
    int head_tiles, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n,
    int64_t v_stride_h, float sm_scale, float logit_cap, int pair64, const int32_t* __restrict__ gate = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char lds_all[];
  // gate: a device word that says, when non-zero, that another kernel of the same launch sequence does this work
  // (the shared-prefix kernel found groups to share); the whole grid then leaves at once
  if (gate && *gate != 0) return;
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  char* kl = lds_all + wid * kLdsPerWave;
  char* vl = kl + kTileBytes;
  // fp8 pool: code -> value table behind the waves' tiles, built before any wave may leave
  [[maybe_unused]] uint16_t* lut = reinterpret_cast<uint16_t*>(lds_all + kWavesPerBlock * kLdsPerWave);
  if constexpr (!std::is_same<KV, T>::value) {
    lut[threadIdx.x] = Conv<T>::from_f32(KvElem<KV>::value(threadIdx.x));  // 256 threads, 256 codes
    __syncthreads();
  }

  const int b = blockIdx.x;
  const int kvh = blockIdx.y / head_tiles, ht = blockIdx.y - kvh * head_tiles;
  const int split = blockIdx.z * kWavesPerBlock + wid;
  const int group = Hq / Hkv;
  const int start = kv_indptr[b];
  const int len = kv_indptr[b + 1] - start;
  const int splits = max(num_kv_splits[b], 1);
  const int per_split = cdiv_i(cdiv_i(len, splits), kMinBlockKV) * kMinBlockKV;
  const int s0 = per_split * split;
  const int s1 = min(s0 + per_split, len);
  if (split >= max_kv_splits || s1 <= s0) return;  // wave-uniform; no block-level synchronisation is used

  const int hl = lane & 15;  // operand column: query head within this head tile
  const int g = lane >> 4;   // 16-lane group
  const int head_in_group = ht * 16 + hl;
  const bool head_ok = head_in_group < group;
  const int head = kvh * group + head_in_group;

  WaveAttn<T> wa;
  wa.init();
  const int half = pair64 > 0 ? min(head_in_group / pair64, 1) : 0;
  if (pair64 > 0)
    wa.load_q_half(head_ok ? reinterpret_cast<const uint16_t*>(q) + (int64_t)b * q_stride_b + (int64_t)head * q_stride_h
                           : nullptr, half);
  else
    wa.load_q(head_ok ? reinterpret_cast<const uint16_t*>(q) + (int64_t)b * q_stride_b + (int64_t)head * q_stride_h
                      : nullptr);
  if constexpr (std::is_same<KV, T>::value)
    wa.run(kl, vl, reinterpret_cast<const uint16_t*>(k_buffer) + (int64_t)kvh * k_stride_h,
           reinterpret_cast<const uint16_t*>(v_buffer) + (int64_t)kvh * v_stride_h, k_stride_n, v_stride_n,
           kv_indices + start, s0, s1, sm_scale, logit_cap);
  else
    wa.run8(kl, vl, lut, reinterpret_cast<const uint8_t*>(k_buffer) + (int64_t)kvh * k_stride_h,
            reinterpret_cast<const uint8_t*>(v_buffer) + (int64_t)kvh * v_stride_h, k_stride_n, v_stride_n,
            kv_indices + start, s0, s1, sm_scale, logit_cap);
  const float m = wa.m;
  const float4v(&oacc)[8] = wa.oacc;

  // ---- epilogue: total sum over the four lane groups, normalise, store partial
  const float lsum = wa.total();
  if (head_ok) {
    const float inv = 1.0f / lsum;
    const int64_t ob = ((int64_t)b * Hq + head) * max_kv_splits + split;
    if (pair64 > 0) {
      float* dst = attn_logits + ob * 64;
#pragma unroll
      for (int n = 0; n < 8; ++n)
        if ((n >> 2) == half) *reinterpret_cast<float4v*>(dst + 16 * (n & 3) + 4 * g) = oacc[n] * inv;
    } else {
      float* dst = attn_logits + ob * 128;
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        // lane (hl, g) holds O[head][16n + 4g + r], r = 0..3
        *reinterpret_cast<float4v*>(dst + 16 * n + 4 * g) = oacc[n] * inv;
      }
    }
    if (g == 0) attn_lse[ob] = (m + log2f(lsum)) * 0.6931471805599453f;
  }
}


// ---------------------------------------------------------------------------------------------
// absorbed-MLA decode (D = 576, Dv = 512, V = the first 512 elements of the K row): a token's latent row
// is fetched once and serves as K (all 576 elements) and as V.  A block of four waves owns one tile of
// 16 query heads and walks its token range in tiles of 32 rows through one LDS stage:
//   S^T = K . Q^T   the 18 k-steps of the 576-deep product are dealt round-robin to the four waves,
//                   whose partial score tiles meet in LDS (every wave then holds the full scores and
//                   runs the same online softmax);
//   O^T += V^T . P^T  wave w owns output elements [128w, 128w + 128).
// Rows sit in LDS at a padded pitch of 1168 bytes: the 16 rows of a K operand read fall on 16
// different 16-byte slots of the bank row without a swizzle, and the transposed V reads address the
// same bytes.
// ---------------------------------------------------------------------------------------------
static constexpr int kMlaD = 576, kMlaDv = 512;
static constexpr int kMlaPitch = kMlaD * 2 + 16;
static constexpr int kMlaTileBytes = kTile * kMlaPitch;  // 37376
static constexpr int kMlaScoreBytes = 4 * 2 * 64 * 16;   // four waves' partial score tiles, per tile of 16 heads

// HT: tiles of 16 query heads per block (2 where the group has 32 heads or more: a latent tile in LDS and every
// operand read out of it then serve twice the heads).
// CAP: the logit cap (tanh) compiled in; a run-time test per score splits the tile loop into blocks the scheduler
// cannot interleave.
// The tile loop of the absorbed-MLA kernels: a block of four waves multiplies HT tiles of 16 query columns (lane hl of
// tile t holds the column whose query row is qrow[t]; nullptr: a column that does not exist) against tokens
// [s0, s1) (non-empty) of the slot list idx, whose latent rows are at kbase + slot * k_stride_n.  On return m / lsum /
// oacc hold the running maximum (log2 domain), this lane's share of the sum and this wave's 128 output elements
// (wave w: elements [128 w, 128 w + 128)) per tile.
template <typename T, int HT, bool CAP>
struct MlaRun {
  float m[HT], lsum[HT];
  float4v oacc[HT][8];

  __device__ __forceinline__ void run(char* lds, const uint16_t* const (&qrow)[HT], const uint16_t* kbase,
                                      int64_t k_stride_n, const int32_t* idx, int s0, int s1, float sm_scale,
                                      float logit_cap) {
    char* tile = lds;
    float4v* sbuf = reinterpret_cast<float4v*>(lds + kMlaTileBytes);  // [HT][4 waves][2 halves][64 lanes]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int hl = lane & 15, g = lane >> 4;
    // Q^T operand of this wave's k-steps wid, wid + 4, ...: lane (hl, g) holds q[column][32*ks + 8*g .. +8)
    constexpr int kSteps = kMlaD / 32;  // 18
    short8 qf[HT][5];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int ks = wid + 4 * i;
        if (qrow[t] && ks < kSteps)
          qf[t][i] = *reinterpret_cast<const short8*>(qrow[t] + 32 * ks + 8 * g);
        else
          qf[t][i] = short8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    const float LOG2E = 1.4426950408889634f;
    const float qk_scale = CAP ? sm_scale : sm_scale * LOG2E;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      m[t] = -INFINITY, lsum[t] = 0.f;
#pragma unroll
      for (int n = 0; n < 8; ++n) oacc[t][n] = float4v{0.f, 0.f, 0.f, 0.f};
    }

    // tile fetch: eight threads per row, thread (row, sub) takes 16-byte pieces sub, sub + 8, ..., sub + 64
    const int frow = threadIdx.x >> 3, fsub = threadIdx.x & 7;
    // FD tiles are in flight (registers) while one is multiplied (two measured no faster than one, with or without
    // the slots read ahead: the step is not bound by the way to the rows and back)
    constexpr int FD = 1;
    int4v kreg[FD][9];
    // a row's slot is read one tile before its bytes are requested (slot and bytes are two dependent trips to memory)
    auto slot_of = [&](int t0) {
      int tok = t0 + frow;
      tok = tok < s1 ? tok : s0;  // rows past the end re-read a valid slot; their scores are masked
      return idx[tok];
    };
    auto fetch = [&](int4v(&kr)[9], int slot) {
      const uint16_t* row = kbase + (int64_t)slot * k_stride_n + fsub * 8;
#pragma unroll
      for (int j = 0; j < 9; ++j) kr[j] = *reinterpret_cast<const int4v*>(row + 64 * j);
    };
    int slot_next = 0;
    auto park = [&](const int4v(&kr)[9]) {
#pragma unroll
      for (int j = 0; j < 9; ++j) *reinterpret_cast<int4v*>(tile + frow * kMlaPitch + (fsub + 8 * j) * 16) = kr[j];
    };

    auto step = [&](int4v(&kr)[9], int t0) {
      __syncthreads();  // every wave is done with the previous tile
      park(kr);
      __syncthreads();
      if (t0 + FD * kTile < s1) {  // later tiles' rows fly while this one is multiplied
        fetch(kr, slot_next);
        slot_next = slot_of(t0 + (FD + 1) * kTile);
      }
      // ---- this wave's share of S^T = K . Q^T: a K operand read serves every column tile
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float4v sacc[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) sacc[t] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          // k-steps past the row (waves 2 and 3 in their fifth round) multiply the row's last step by the zeros their
          // q registers hold there: no branch in the tile loop
          const int ks = min(wid + 4 * i, kSteps - 1);
          const short8 kf = *reinterpret_cast<const short8*>(tile + (16 * hh + hl) * kMlaPitch + (4 * ks + g) * 16);
#pragma unroll
          for (int t = 0; t < HT; ++t) sacc[t] = Mfma16<T>::mma(kf, qf[t][i], sacc[t]);
        }
#pragma unroll
        for (int t = 0; t < HT; ++t) sbuf[((t * 4 + wid) * 2 + hh) * 64 + lane] = sacc[t];
      }
      __syncthreads();
      // ---- full scores (same sum order in every wave), scale / cap / mask, online softmax (exp2 domain)
      short8 pf[HT];  // P^T operand: k-slot j <-> token (j < 4 ? 4g + j : 16 + 4g + j - 4), as the S^T registers stand
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        float sv[8];
        float tmax = -INFINITY;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          float4v tot = sbuf[((t * 4) * 2 + hh) * 64 + lane];
#pragma unroll
          for (int w = 1; w < 4; ++w) tot += sbuf[((t * 4 + w) * 2 + hh) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float sc = tot[r] * qk_scale;
            if constexpr (CAP) sc = logit_cap * tanhf(sc / logit_cap) * LOG2E;
            const int tok = t0 + 16 * hh + 4 * g + r;
            sc = tok < s1 ? sc : -INFINITY;
            sv[hh * 4 + r] = sc;
            tmax = fmaxf(tmax, sc);
          }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m[t], tmax);  // finite: token t0 of this tile is always valid
        const float alpha = exp2f(m[t] - m_new);
        m[t] = m_new;
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = exp2f(sv[j] - m_new);
          psum += p;
          pf[t][j] = (short)Conv<T>::from_f32(p);
        }
        lsum[t] = lsum[t] * alpha + psum;
#pragma unroll
        for (int n = 0; n < 8; ++n) oacc[t][n] *= alpha;
      }
      // ---- O^T += V^T . P^T over this wave's 128 output elements, V read transposed out of the K rows
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const int byte_in_row = 256 * wid + 32 * n + 8 * (hl & 3);
        const short4v lo = lds_read_tr16(tile + (4 * g + (hl >> 2)) * kMlaPitch + byte_in_row);
        const short4v hi = lds_read_tr16(tile + (16 + 4 * g + (hl >> 2)) * kMlaPitch + byte_in_row);
        const short8 vf = short8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int t = 0; t < HT; ++t) oacc[t][n] = Mfma16<T>::mma(vf, pf[t], oacc[t][n]);
      }
    };

#pragma unroll
    for (int f = 0; f < FD; ++f)
      if (s0 + f * kTile < s1) fetch(kreg[f], slot_of(s0 + f * kTile));
    slot_next = slot_of(s0 + FD * kTile);
    for (int t0 = s0; t0 < s1; t0 += FD * kTile) {
#pragma unroll
      for (int f = 0; f < FD; ++f)
        if (t0 + f * kTile < s1) step(kreg[f], t0 + f * kTile);  // block-uniform
    }
  }

  // tile t's partial (normalised O[512] + log-sum-exp) into row `ob` of [rows][512] / [rows]
  __device__ __forceinline__ void store(int t, bool ok, int64_t ob, float* __restrict__ logits, float* __restrict__ lse) const {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int g = lane >> 4;
    float ls = lsum[t];
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    if (ok) {
      const float inv = 1.0f / ls;
      float* dst = logits + ob * kMlaDv + 128 * wid;
#pragma unroll
      for (int n = 0; n < 8; ++n) *reinterpret_cast<float4v*>(dst + 16 * n + 4 * g) = oacc[t][n] * inv;
      if (wid == 0 && g == 0) lse[ob] = (m[t] + log2f(ls)) * 0.6931471805599453f;
    }
  }
};

template <typename T, int HT, bool CAP>
__global__ void __launch_bounds__(256, HT == 1 ? 3 : 2) decode_stage1_mla(
    const void* __restrict__ q, const void* __restrict__ kv_buffer, const int32_t* __restrict__ kv_indptr,
    const int32_t* __restrict__ kv_indices, float* __restrict__ attn_logits, float* __restrict__ attn_lse,
    const int32_t* __restrict__ num_kv_splits, int max_kv_splits, int Hq, int Hkv, int head_tiles, int64_t q_stride_b,
    int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, float sm_scale, float logit_cap,
    const int32_t* __restrict__ gate = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  if (gate && *gate != 0) return;  // see decode_stage1_mfma128
  const int lane = threadIdx.x & 63;
  const int hl = lane & 15;
  const int b = blockIdx.x;
  const int kvh = blockIdx.y / head_tiles, ht = blockIdx.y - kvh * head_tiles;
  const int split = blockIdx.z;
  const int group = Hq / Hkv;
  const int start = kv_indptr[b];
  const int len = kv_indptr[b + 1] - start;
  const int splits = max(num_kv_splits[b], 1);
  const int per_split = cdiv_i(cdiv_i(len, splits), kMinBlockKV) * kMinBlockKV;
  const int s0 = per_split * split, s1 = min(s0 + per_split, len);
  if (s1 <= s0) return;  // block-uniform
  bool head_ok[HT];
  int head[HT];
  const uint16_t* qrow[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    const int head_in_group = (ht * HT + t) * 16 + hl;
    head_ok[t] = head_in_group < group;
    head[t] = kvh * group + head_in_group;
    qrow[t] = head_ok[t] ? reinterpret_cast<const uint16_t*>(q) + (int64_t)b * q_stride_b + (int64_t)head[t] * q_stride_h
                         : nullptr;
  }
  MlaRun<T, HT, CAP> r;
  r.run(lds, qrow, reinterpret_cast<const uint16_t*>(kv_buffer) + (int64_t)kvh * k_stride_h, k_stride_n,
        kv_indices + start, s0, s1, sm_scale, logit_cap);
#pragma unroll
  for (int t = 0; t < HT; ++t)
    r.store(t, head_ok[t], ((int64_t)b * Hq + head[t]) * max_kv_splits + split, attn_logits, attn_lse);
}

// ---------------------------------------------------------------------------------------------
// shared-prefix decode in ONE launch.  Two kinds of blocks run side by side:
//
//   prefix blocks  (split, 64-column block, kv head): stream one slice of a group's shared prefix once and
//                  multiply it against 64 (request, head) columns; one partial (normalised O[128] + log-sum-exp,
//                  fp32) per column and split goes to the workspace with agent-scope stores, then the block adds
//                  one to the counter of every (request, kv head) pair it contributed to.
//   tail blocks    (request, kv head, head tile): the four waves walk the request's private remainder in four
//                  splits and keep their partials in LDS; the block then waits until the pair's counter shows
//                  that every prefix partial has landed, merges prefix and remainder in slot order and writes
//                  the output rows.
//
// Prefix blocks hold the lowest block indices and never wait for anything; a tail block only waits for prefix
// blocks, all of which were handed to the hardware before it.  The wait is bounded all the same: a tail block that
// sees no progress within kSharedWaitTicks fills its output rows with NaN and raises the workspace's error word
// instead of hanging the GPU.  The last tail block of a pair to finish clears the pair's counter.
// ---------------------------------------------------------------------------------------------
#ifdef SGLK_ATTN_STAMPS
// development aid: wall-clock stamps per block and phase (tools/bench_attn.py --stamps)
__device__ long long g_attn_stamps[1024 * 8];
#define ATTN_STAMP(k)                                                                                           \
  do {                                                                                                          \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_attn_stamps[blockIdx.x * 8 + (k)] = (long long)wall_clock64(); \
  } while (0)
#else
#define ATTN_STAMP(k) \
  do {                \
  } while (0)
#endif

static constexpr int kSharedMaxSplits = 16;
static constexpr int kSc1 = 16;                             // cache policy of the buffer instructions below: agent scope
static constexpr uint64_t kSharedWaitTicks = 200000000ull;  // 2 s of the 100 MHz wall clock
static constexpr int kTailDone = 1 << 16;                   // a finished tail block adds this to the pair's counter
static constexpr int kPrefixPF = 2;                         // prefix tiles in flight per block

// ---------------------------------------------------------------------------------------------
// The per-step plan of the shared-prefix kernel.  One decode step runs the kernel once per layer over the same
// request table, so everything that does not depend on the layer is resolved once, on the device, at fixed
// addresses (a graph replay only rewrites the plan's contents):
//   * which requests share their first slots (groups), found from the request table itself;
//   * per request: where its private remainder starts in kv_indices, how long it is, how many prefix partials
//     its merge has to wait for;
//   * per prefix block ("item"): the slot range it streams, the request that owns each of its 64 columns, and
//     the pool slots of its first two tiles -- one trip to memory fetches all of it, the second fetches rows.
// Layout (int32 words): header[16] | order[Bcap] | tail[Bcap][4] | group_indptr[Bcap + 1] | prefix_lens[Bcap] |
// leader[Bcap] | lcp[Bcap] | items[max_items][kPlanItemInts] (16-byte aligned).
// ---------------------------------------------------------------------------------------------
static constexpr int kPlanHdr = 16;
static constexpr int kPlanItemInts = 144;   // header[16], column owner[64], first slots[64]
static constexpr int kPlanMaxB = 1024;      // one block assembles the plan
enum { kPlanNItems = 0, kPlanNGroups = 1, kPlanB = 2, kPlanHeads = 3, kPlanSplits = 4, kPlanMaxItems = 5 };

// Prefix blocks a launch aims for: two per CU.  The prefix blocks are the launch's critical path (they stream, store and
// count before anybody can merge), so a batch of many groups gets as many splits as that allows: ten groups of 32
// requests run 71 instead of 95 us per layer with a target of 512 instead of 256 (768: 70.7), a single group of 64
// is at max_kv_splits either way.  SGLK_PREFIX_TARGET overrides (the plan and the launch must see the same value).
static int prefix_block_target() {
  static const int t = getenv("SGLK_PREFIX_TARGET") ? atoi(getenv("SGLK_PREFIX_TARGET")) : 512;
  return t > 0 ? t : 512;
}
inline int shared_prefix_splits(int B, int heads, int Hkv, int max_kv_splits) {
  const int col_blocks = (B * heads + 63) / 64;
  const int per = col_blocks * Hkv > 0 ? col_blocks * Hkv : 1;
  int s = prefix_block_target() / per;
  s = s < max_kv_splits ? s : max_kv_splits;
  return s < 1 ? 1 : s;
}
__host__ __device__ inline int plan_max_items(int B, int heads, int splits) {
  // sum over groups of ceil(columns / 64) <= ceil(B * heads / 64) + (groups that share: at most B / 2)
  return splits * ((B * heads + 63) / 64 + B / 2 + 1);
}
struct PlanView {
  int32_t *hdr, *order, *tail, *group_indptr, *prefix_lens, *leader, *lcp, *items;
  int64_t words;
};
__host__ __device__ inline PlanView plan_view(int32_t* base, int Bcap, int max_items) {
  PlanView v;
  v.hdr = base;
  v.order = base + kPlanHdr;
  v.tail = v.order + Bcap;
  v.group_indptr = v.tail + 4 * (int64_t)Bcap;
  v.prefix_lens = v.group_indptr + Bcap + 1;
  v.leader = v.prefix_lens + Bcap;
  v.lcp = v.leader + Bcap;
  const int64_t off = ((v.lcp + Bcap - base) + 3) / 4 * 4;
  v.items = base + off;
  v.words = off + (int64_t)max_items * kPlanItemInts;
  return v;
}

__device__ __forceinline__ int block_min_256(int v, int* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = min(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  const int r = red[0];
  __syncthreads();
  return r;
}

// One block per request: its leader (the first request of the batch whose list starts with the same pool slot --
// two requests that matched the same radix-tree node hold the same slots) and the length of the slot prefix it
// shares with that leader, the newest token excluded.
__global__ void __launch_bounds__(256) shared_plan_lcp_kernel(const int32_t* __restrict__ req_to_token, int64_t row_stride,
                                                              const void* req_pool_indices, int idx_is64,
                                                              const void* seq_lens, int lens_is64, int B,
                                                              int32_t* __restrict__ leader_out,
                                                              int32_t* __restrict__ lcp_out) {
  __shared__ int red[256];
  const int b = blockIdx.x;
  const int64_t rb = ld_idx(req_pool_indices, idx_is64, b);
  const int lenb = (int)ld_idx(seq_lens, lens_is64, b);
  const int32_t* c = req_to_token + rb * row_stride;
  int lead = b;
  if (lenb >= 2) {
    const int fb = c[0];
    for (int i = threadIdx.x; i < b; i += 256) {
      const int64_t ri = ld_idx(req_pool_indices, idx_is64, i);
      if ((int)ld_idx(seq_lens, lens_is64, i) >= 2 && req_to_token[ri * row_stride] == fb) {
        lead = i;
        break;
      }
    }
  }
  lead = block_min_256(lead, red);
  if (lead == b) {
    if (threadIdx.x == 0) {
      leader_out[b] = b;
      lcp_out[b] = max(lenb - 1, 0);
    }
    return;
  }
  const int64_t rl = ld_idx(req_pool_indices, idx_is64, lead);
  const int n = min((int)ld_idx(seq_lens, lens_is64, lead), lenb) - 1;
  const int32_t* a = req_to_token + rl * row_stride;
  int first = n;
  for (int base = 0; base < n; base += 1024) {
    int mine = n;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = base + (int)threadIdx.x + 256 * u;
      if (j < n && a[j] != c[j]) mine = min(mine, j);
    }
    const int m = block_min_256(mine, red);
    if (m < n) {
      first = m;
      break;
    }
  }
  if (threadIdx.x == 0) {
    leader_out[b] = lead;
    lcp_out[b] = first;
  }
}

// caller-given groups of adjacent requests -> the (leader, shared length) form the assembly takes
__global__ void __launch_bounds__(256) shared_plan_from_groups_kernel(const int32_t* __restrict__ group_indptr,
                                                                      const int32_t* __restrict__ prefix_lens, int G, int B,
                                                                      int32_t* __restrict__ leader_out,
                                                                      int32_t* __restrict__ lcp_out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  int lo = 0, hi = G - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (group_indptr[mid] <= b)
      lo = mid;
    else
      hi = mid - 1;
  }
  leader_out[b] = group_indptr[lo];
  lcp_out[b] = prefix_lens[lo];
}

__device__ __forceinline__ int plan_active_splits(int plen, int prefix_splits) {
  if (plen <= 0) return 0;
  const int per_p = cdiv_i(cdiv_i(plen, prefix_splits), kMinBlockKV) * kMinBlockKV;
  return cdiv_i(plen, per_p);
}

// exclusive scan of v[0 .. n) (n <= 1024) by a block of 1024 threads; returns the total
__device__ __forceinline__ int block_excl_scan_1024(int* v, int* tmp, int n) {
  const int t = threadIdx.x;
  const int x = t < n ? v[t] : 0;
  tmp[t] = x;
  __syncthreads();
  for (int o = 1; o < n; o <<= 1) {  // entries at and beyond n are zero: strides of n and more add nothing below n
    const int add = (t >= o && t < n) ? tmp[t - o] : 0;
    __syncthreads();
    tmp[t] += add;
    __syncthreads();
  }
  const int total = n > 0 ? tmp[n - 1] : 0;
  if (t < n) v[t] = tmp[t] - x;
  __syncthreads();
  return total;
}

// One block: groups, member order, per-request tail records and the prefix items, from (leader, lcp).
// single_ok: groups of one request keep their prefix (the caller asked for it); otherwise a request alone in its
// group, or a group that shares fewer than min_share slots, walks its whole list as a private remainder.
__global__ void __launch_bounds__(1024) shared_plan_assemble_kernel(const int32_t* __restrict__ kv_indptr,
                                                                    const int32_t* __restrict__ kv_indices, int B,
                                                                    int heads, int prefix_splits, int max_items,
                                                                    int min_share, int single_ok, int32_t* plan_base) {
  __shared__ int lead[kPlanMaxB], gid[kPlanMaxB], rank[kPlanMaxB], gsize[kPlanMaxB], gplen[kPlanMaxB], ioff[kPlanMaxB],
      tmp[1024];
  const PlanView pv = plan_view(plan_base, B, max_items);
  const int t = threadIdx.x;
  if (t < B) {
    lead[t] = pv.leader[t];
    gsize[t] = lead[t] == t ? 1 : 0;  // for the scan: leaders in batch order number the groups
  }
  __syncthreads();
  const int G = block_excl_scan_1024(gsize, tmp, B);  // gsize[leader] = group index of that leader
  int my_gid = 0;
  if (t < B) my_gid = gsize[lead[t]];
  __syncthreads();
  if (t < B) {
    gid[t] = my_gid;
    gsize[t] = 0;
    gplen[t] = 0x7fffffff;
  }
  __syncthreads();
  if (t < B) {
    atomicAdd(&gsize[my_gid], 1);
    atomicMin(&gplen[my_gid], pv.lcp[t]);
    int r = 0;
    for (int i = 0; i < t; ++i) r += lead[i] == lead[t] ? 1 : 0;
    rank[t] = r;
  }
  __syncthreads();
  int my_plen = 0, my_cbs = 0, my_aps = 0;
  if (t < G) {
    my_plen = gplen[t];
    if ((!single_ok && gsize[t] < 2) || my_plen < min_share) my_plen = 0;
    my_aps = plan_active_splits(my_plen, prefix_splits);
    my_cbs = my_plen > 0 ? cdiv_i(gsize[t] * heads, 64) : 0;
    gplen[t] = my_plen;
    ioff[t] = my_aps * my_cbs;
  }
  __syncthreads();
  // group starts (member offsets) and item offsets
  const int my_size = t < G ? gsize[t] : 0;
  __syncthreads();
  block_excl_scan_1024(gsize, tmp, G);   // gsize[g] = first member's position in `order`
  const int n_items = min(block_excl_scan_1024(ioff, tmp, G), max_items);
  if (t < G) {
    pv.group_indptr[t] = gsize[t];
    pv.prefix_lens[t] = my_plen;
    if (t == G - 1) pv.group_indptr[G] = gsize[t] + my_size;
  }
  if (t < B) {
    const int g = gid[t];
    pv.order[gsize[g] + rank[t]] = t;
    const int plen = gplen[g];
    const int aps = plan_active_splits(plen, prefix_splits);
    const int start = kv_indptr[t] + plen;
    // prefix blocks whose 64 columns touch this request (its rank in the group decides the columns)
    const int pb = ((rank[t] + 1) * heads - 1) / 64 - (rank[t] * heads) / 64 + 1;
    pv.tail[4 * t + 0] = start;
    pv.tail[4 * t + 1] = kv_indptr[t + 1] - start;
    pv.tail[4 * t + 2] = pb * aps;
    pv.tail[4 * t + 3] = aps;
  }
  if (t == 0) {
    pv.hdr[kPlanNItems] = n_items;
    pv.hdr[kPlanNGroups] = G;
    pv.hdr[kPlanB] = B;
    pv.hdr[kPlanHeads] = heads;
    pv.hdr[kPlanSplits] = prefix_splits;
    pv.hdr[kPlanMaxItems] = max_items;
  }
  __syncthreads();
  __threadfence_block();  // `order` is read back below
  // items of group g: (split, column block) in that order, so that the blocks of one split start together; one
  // thread per item (its group found by a walk over the item offsets: a handful of groups)
  for (int j = t; j < n_items; j += 1024) {
    int gi = 0;
    while (gi + 1 < G && ioff[gi + 1] <= j) ++gi;
    const int plen = gplen[gi];
    const int size = pv.group_indptr[gi + 1] - gsize[gi];
    const int cbs = cdiv_i(size * heads, 64);
    const int k = j - ioff[gi];
    const int split = k / cbs, cb = k % cbs;
    const int per_split = cdiv_i(cdiv_i(plen, prefix_splits), kMinBlockKV) * kMinBlockKV;
    int32_t* it = pv.items + (int64_t)j * kPlanItemInts;
    it[0] = kv_indptr[pv.order[gsize[gi]]];  // every member lists the same slots for the prefix: walk the first one's
    it[1] = per_split * split;
    it[2] = min(per_split * (split + 1), plen);
    it[3] = cb;
    it[4] = split;
    it[5] = gi;
    it[6] = gsize[gi];
    it[7] = size * heads;
  }
  __syncthreads();
  __threadfence_block();
  if (t < B) pv.lcp[t] = 0;  // from here on: ready[token] of the attention launches' rope role (read above, long ago)
  for (int w = t; w < n_items * 128; w += 1024) {
    int32_t* it = pv.items + (int64_t)(w >> 7) * kPlanItemInts;
    const int e = w & 127;
    if (e < 64) {
      const int col = it[3] * 64 + e;
      it[16 + e] = col < it[7] ? pv.order[it[6] + col / heads] : -1;
    } else {
      int tok = it[1] + (e - 64);
      tok = tok < it[2] ? tok : it[1];
      it[80 + (e - 64)] = kv_indices[it[0] + tok];
    }
  }
}

struct SharedWs {
  float* part_o;    // [B * Hq][slots][128]: prefix splits first, then the wave splits of remainders that span blocks
  float* part_lse;  // [B * Hq][slots]
  int* counters;    // [B * Hkv] (low half: prefix partials landed, high half: merging tail blocks done)
  int* error;       // raised by a tail block whose wait ran out
  int* tickets;     // [B * Hkv * head_tiles]: tail blocks of a (request, kv head, head tile) that have published
  int slots;        // prefix_splits + kWavesPerBlock * tail_tmax
  int tail_tmax;    // most tail blocks per (request, kv head, head tile)
  int rows;         // B * Hq * slots (the host keeps rows * 512 below 2^31: byte offsets are 32-bit)
  __device__ __forceinline__ auto rsrc_o() const {
    return __builtin_amdgcn_make_buffer_rsrc(part_o, 0, rows * 512, 0x00020000);
  }
  __device__ __forceinline__ auto rsrc_lse() const {
    return __builtin_amdgcn_make_buffer_rsrc(part_lse, 0, rows * 4, 0x00020000);
  }
};

// number of 64-column prefix blocks whose columns touch request r (r relative to its group's first request)
__device__ __forceinline__ int prefix_blocks_of(int r, int heads) {
  return ((r + 1) * heads - 1) / 64 - (r * heads) / 64 + 1;
}
__device__ __forceinline__ int active_prefix_splits(int plen, int prefix_splits) {
  if (plen <= 0) return 0;
  const int per_p = cdiv_i(cdiv_i(plen, prefix_splits), kMinBlockKV) * kMinBlockKV;
  return cdiv_i(plen, per_p);
}

// ROPE: q is not read; the query rows, and every request's new K / V row, come from the qkv projection's split-K
// partial sums (qp).  A (request, kv head) pair's tail block (one head tile only) writes the pair's new pool rows
// before it walks the request's list, whose last entry is that row's slot.
// ---------------------------------------------------------------------------------------------
// The rotation + pool write of a decode layer as blocks of the attention launch ("rope role").  The separate kernel
// (rope_store_partial_kernel: sum the qkv projection's split-K partials, add the bias, round, rotate q and k, write q
// and the token's new K / V rows) is a launch of a few microseconds between two kernels that wait for each other.
// Here its blocks hold the lowest indices of the attention launch, never wait for anything, and count every finished
// (token, part) in ready[token]; the attention blocks fetch their slot lists and K / V rows as before and wait for the
// count only where they first need a query row (or, the block that walks a request's remainder, the new row's slot).
// ready[] lives in the step's plan, which clears it once per step; the n-th launch of a step waits for parts * n.
// Same bits as the separate kernel.  Full rotation of 128-element heads, pairs (d, d + 64).
// ---------------------------------------------------------------------------------------------
struct RopeRole {
  int blocks;        // rope blocks at the head of the grid (a multiple of 8; 0: no rope role)
  int parts;         // blocks per token
  int tokens;
  int target;        // value of ready[t] that says token t is done in this launch
  int* ready;
  uint16_t* q_out;   // rotated query rows go here (row stride q_stride_b)
};

template <typename T>
__device__ __forceinline__ void store_wt32(uint16_t* p, uint16_t lo, uint16_t hi) {
  // write-through (agent scope): the readers are other blocks of this launch, possibly on another XCD
  __hip_atomic_store(reinterpret_cast<uint32_t*>(p), (uint32_t)lo | ((uint32_t)hi << 16), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T>
__device__ __forceinline__ void rope_role_block(int r, const RopeRole& rr, const QkvPartials& qp, int64_t q_stride_b,
                                                void* k_buffer, void* v_buffer, int64_t k_stride_n, int64_t k_stride_h,
                                                int64_t v_stride_n, int64_t v_stride_h, int Hq, int Hkv, char* lds) {
  uint16_t* rowS = reinterpret_cast<uint16_t*>(lds);
  const int t = r / rr.parts, part = r - t * rr.parts;
  if (t >= rr.tokens) return;  // padding blocks (the role's block count is rounded up to a multiple of 8)
  const int total_heads = Hq + 2 * Hkv;
  const int hpb = cdiv_i(total_heads, rr.parts);
  const int h0 = part * hpb, h1 = min(h0 + hpb, total_heads);
  if (h0 < h1) {
    const int c0 = h0 * 128, ncol = (h1 - h0) * 128;
    const int64_t pos = qp.positions[t];
    const int64_t slot = ld_idx(qp.loc, qp.loc_is64, t);
    for (int v = threadIdx.x; v < (ncol >> 3); v += blockDim.x) {
      float acc[8];
      sum_partials(qp.partials, qp.splits, qp.part_stride, (int64_t)t * qp.row_len + c0 + 8 * v, acc);
      short8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float bj = qp.bias ? Conv<T>::to_f32(qp.bias[c0 + 8 * v + j]) : 0.f;
        o[j] = (short)Conv<T>::from_f32(acc[j] + bj);
      }
      reinterpret_cast<short8*>(rowS)[v] = o;
    }
    __syncthreads();
    const float* cs = qp.cos_sin + pos * 128;
    uint16_t* qdst = rr.q_out + (int64_t)t * q_stride_b;
    uint16_t* kdst = reinterpret_cast<uint16_t*>(k_buffer) + slot * k_stride_n;
    uint16_t* vdst = reinterpret_cast<uint16_t*>(v_buffer) + slot * v_stride_n;
    const int rot_heads = max(0, min(h1, Hq + Hkv) - h0);  // q and k heads of this block
    // two neighbouring pairs per thread: elements (p, p + 1) and (p + 64, p + 65) leave as 4-byte stores
    for (int i = threadIdx.x; i < rot_heads * 32; i += blockDim.x) {
      const int hl = i >> 5, p = 2 * (i & 31), h = h0 + hl;
      const uint16_t* v = rowS + hl * 128;
      uint16_t ra[2], rb[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float fa, fb;
        rope_pair(Conv<T>::to_f32(v[p + e]), Conv<T>::to_f32(v[p + e + 64]), cs[p + e], cs[64 + p + e], fa, fb);
        ra[e] = Conv<T>::from_f32(fa);
        rb[e] = Conv<T>::from_f32(fb);
      }
      uint16_t* d = h < Hq ? qdst + h * 128 : kdst + (int64_t)(h - Hq) * k_stride_h;
      store_wt32<T>(d + p, ra[0], ra[1]);
      store_wt32<T>(d + p + 64, rb[0], rb[1]);
    }
    // v heads of this block: as they are
    const int v0 = max(h0, Hq + Hkv);
    for (int i = threadIdx.x; i < max(0, h1 - v0) * 64; i += blockDim.x) {
      const int hl = i >> 6, e = 2 * (i & 63), h = v0 + hl;
      const uint16_t* v = rowS + (h - h0) * 128;
      store_wt32<T>(vdst + (int64_t)(h - Hq - Hkv) * v_stride_h + e, v[e], v[e + 1]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every store of this wave has been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(rr.ready + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave-level: returns once ready[tok] has reached target for every lane's token (tok < 0: nothing to wait for)
__device__ __forceinline__ void rope_wait(const RopeRole& rr, int tok, int* error_word) {
  const uint64_t t_start = wall_clock64();
  uint32_t polls = 0;
  while (true) {
    const bool ok = tok < 0 || __hip_atomic_load(rr.ready + tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= rr.target;
    if (__all(ok)) break;
    if ((++polls & 255u) == 0 && wall_clock64() - t_start > 200000000ull) {  // two seconds: raise the error word, go on
      __hip_atomic_store(error_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// PLAN: group_indptr / prefix_lens are not read; plan_hdr, plan_tail and plan_items (the per-step plan above) say what
// every block does, and the prefix blocks loop over the items (their number varies from step to step, the grid of a
// captured launch does not).
// RR: rope role (see above; PLAN only): blocks [0, rr.blocks) rotate and write, the attention blocks follow.
// KV: the pool's element type (T, or an fp8 format: rows are then widened through a table on their way into LDS,
// the tail role parks K rows too, and the launch carries 64 KiB + the table instead of 32 KiB).
// PAIR: head size 64; Hkv counts PAIRS of neighbouring kv heads (one 128-element row piece), a pair's query heads
// are those of its first kv head followed by those of its second, a query row sits in the half of the 128-wide
// operand that faces its own kv head (decode_stage1_mfma128 has the details) and only that half of O is stored.
// MT: remainders may span several tail blocks (a launch whose first tail blocks already fill the chip is compiled
// without that path: it needs registers the single-block form spills for).
template <typename T, bool ROPE, bool PLAN, bool RR = false, typename KV = T, bool PAIR = false, bool MT = false>
__global__ void __launch_bounds__(256, 3) decode_shared_fused_kernel(
    const void* q, QkvPartials qp, void* k_buffer, void* v_buffer,
    const int32_t* __restrict__ kv_indptr, const int32_t* __restrict__ kv_indices,
    const int32_t* __restrict__ group_indptr, const int32_t* __restrict__ prefix_lens, int num_groups,
    const int32_t* __restrict__ plan_hdr, const int32_t* __restrict__ plan_tail, const int32_t* __restrict__ plan_items,
    SharedWs ws,
    int prefix_splits, int n_prefix_blocks, int col_blocks, int B, int Hq, int Hkv, int head_tiles, int64_t q_stride_b,
    int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, float sm_scale,
    float logit_cap, void* __restrict__ o, int64_t o_stride_b, int64_t o_stride_h, float* __restrict__ xsum_out,
    int64_t xs_stride, int gated, RopeRole rr) {
  extern __shared__ __attribute__((aligned(16))) char lds_all[];  // 32 KiB: the prefix role's two stages of K + V tiles,
                                                                  // or the tail role's four V tiles
  __shared__ int wait_failed;
  __shared__ float tail_lse[kWavesPerBlock][16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int hl = lane & 15, g = lane >> 4;
  const int heads = Hq / Hkv;
  constexpr bool KV8 = !std::is_same<KV, T>::value;
  static_assert(!(KV8 || PAIR || MT) || (PLAN && !ROPE && !RR), "fp8 pools, head-size-64 pairs, multi-block remainders: planned form only");
  [[maybe_unused]] uint16_t* lut = reinterpret_cast<uint16_t*>(lds_all + 2 * kBlockAttnLds);
  if constexpr (KV8) {
    lut[threadIdx.x] = Conv<T>::from_f32(KvElem<KV>::value(threadIdx.x));  // 256 threads, 256 codes
    __syncthreads();
  }
  ATTN_STAMP(0);

  int bid = blockIdx.x;  // index among the attention blocks
  // a plan laid out for another geometry (heads per kv head, splits, batch) must not be walked
  bool geom_bad = false;
  if constexpr (PLAN)
    geom_bad = plan_hdr[kPlanHeads] != heads || plan_hdr[kPlanSplits] != prefix_splits || plan_hdr[kPlanB] != B;
  if constexpr (RR) {
    if (bid < rr.blocks) {
      rope_role_block<T>(bid, rr, qp, q_stride_b, k_buffer, v_buffer, k_stride_n, k_stride_h, v_stride_n, v_stride_h, Hq,
                         Hkv, lds_all);
      return;
    }
    bid -= rr.blocks;
  }
  if constexpr (PLAN) {
    if (bid < n_prefix_blocks) {
      // ---- prefix block (planned): block index = item * Hkv + kvh; items past the grid are taken in further rounds
      if (geom_bad) return;  // the tail blocks see the same words: they do not wait and report the error
      const int kvh = bid % Hkv;
      const int items_per_round = n_prefix_blocks / Hkv;
      int j = bid / Hkv;
      const int n_items = plan_hdr[kPlanNItems];
      const int c = wid * 16 + hl;
      const int frow = threadIdx.x >> 4;
      while (true) {
        // One trip: the item's header (uniform address: scalar loads), this lane's column owner and this thread's four
        // first rows.  The first item is read before n_items is known (its address is inside the plan whatever j is).
        const int32_t* it = plan_items + (int64_t)j * kPlanItemInts;
        const int list_off = it[0], t0 = it[1], t1 = it[2], cb = it[3], split = it[4];
        const int req = it[16 + c];
        const int pre[4] = {it[80 + frow], it[80 + frow + 16], it[80 + 32 + frow], it[80 + 32 + frow + 16]};
        if (j >= n_items) break;  // block-uniform
        const bool ok = req >= 0;
        const int head = kvh * heads + (cb * 64 + c) % heads;
        BlockAttn<T, 1> st;
        st.init();
        int qpos[1] = {0};
        const int32_t* const plist = kv_indices + list_off;
        if constexpr (ROPE) {
          const int64_t pos = ok ? qp.positions[req] : 0;
          st.template run<false, true, kPrefixPF>(
              lds_all, reinterpret_cast<const uint16_t*>(k_buffer) + (int64_t)kvh * k_stride_h,
              reinterpret_cast<const uint16_t*>(v_buffer) + (int64_t)kvh * v_stride_h, k_stride_n, v_stride_n, plist, t0, t1,
              qpos, 0, sm_scale, logit_cap, [&]() { load_q_rotated<T>(st.qf[0], qp, req, pos, head * 128, ok); }, nullptr,
              pre);
        } else if constexpr (RR) {
          // the rows are on their way before the query rows are asked for: the rotation runs in their shadow
          st.template run<false, true, kPrefixPF>(
              lds_all, reinterpret_cast<const uint16_t*>(k_buffer) + (int64_t)kvh * k_stride_h,
              reinterpret_cast<const uint16_t*>(v_buffer) + (int64_t)kvh * v_stride_h, k_stride_n, v_stride_n, plist, t0, t1,
              qpos, 0, sm_scale, logit_cap,
              [&]() {
                rope_wait(rr, req, ws.error);
                st.load_q(0, ok ? reinterpret_cast<const uint16_t*>(q) + (int64_t)req * q_stride_b + (int64_t)head * q_stride_h
                                : nullptr);
              },
              nullptr, pre);
        } else {
          const uint16_t* qrow = ok ? reinterpret_cast<const uint16_t*>(q) + (int64_t)req * q_stride_b + (int64_t)head * q_stride_h
                                    : nullptr;
          if constexpr (PAIR)
            st.load_q_half(0, qrow, ((cb * 64 + c) % heads) >= (heads >> 1) ? 1 : 0);
          else
            st.load_q(0, qrow);
          if constexpr (KV8)  // byte pointers, strides in bytes
            st.template run<false, true, kPrefixPF, typename BlockAttn<T, 1>::NoHook, true>(
                lds_all,
                reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(k_buffer) + (int64_t)kvh * k_stride_h),
                reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(v_buffer) + (int64_t)kvh * v_stride_h),
                k_stride_n, v_stride_n, plist, t0, t1, qpos, 0, sm_scale, logit_cap, typename BlockAttn<T, 1>::NoHook(), lut,
                pre);
          else
            st.template run<false, true, kPrefixPF>(
                lds_all, reinterpret_cast<const uint16_t*>(k_buffer) + (int64_t)kvh * k_stride_h,
                reinterpret_cast<const uint16_t*>(v_buffer) + (int64_t)kvh * v_stride_h, k_stride_n, v_stride_n, plist, t0,
                t1, qpos, 0, sm_scale, logit_cap, typename BlockAttn<T, 1>::NoHook(), nullptr, pre);
        }
        const float tot = st.total(0);
        ATTN_STAMP(1);
        if (ok) {
          const float inv = 1.0f / tot;
          const int ob = (req * Hq + head) * ws.slots + split;
          const auto ro = ws.rsrc_o();
#pragma unroll
          for (int n = 0; n < 8; ++n)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, st.oacc[0][n] * inv), ro,
                                                   ob * 512 + (16 * n + 4 * g) * 4, 0, kSc1);
          if (g == 0)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (st.m[0] + log2f(tot)) * 0.6931471805599453f),
                                                  ws.rsrc_lse(), ob * 4, 0, kSc1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every store of this wave has been acknowledged
        __syncthreads();
        ATTN_STAMP(2);
        // one count per request this block's columns touch: the first column of every owner
        if (threadIdx.x < 64) {
          const int owner = it[16 + threadIdx.x];
          const int before = threadIdx.x ? it[16 + threadIdx.x - 1] : -1;
          if (owner >= 0 && owner != before)
            __hip_atomic_fetch_add(ws.counters + (int64_t)owner * Hkv + kvh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        j += items_per_round;
        if (j >= n_items) break;
      }
      return;
    }
  } else if ((int)blockIdx.x < n_prefix_blocks) {
    // ---- prefix block: index = ((grp * prefix_splits + split) * col_blocks + cb) * Hkv + kvh, so that the blocks of
    // one kv head are congruent modulo Hkv (with 8 kv heads: one XCD and its L2 per kv head)
#ifdef SGLK_ATTN_PRIO
    __builtin_amdgcn_s_setprio(3);  // every tail block waits for these: issue them ahead of the co-resident tail waves
#endif
    int i = blockIdx.x;
    const int kvh = i % Hkv;
    i /= Hkv;
    const int cb = i % col_blocks;
    i /= col_blocks;
    const int split = i % prefix_splits, grp = i / prefix_splits;
    // the first request's offset into kv_indices depends on gstart; fetching the head of kv_indptr alongside the
    // group's metadata saves one of the four dependent trips to memory this block starts with
    const int spec_indptr = kv_indptr[min(lane, B)];
    const int gstart = group_indptr[grp];
    const int gsize = group_indptr[grp + 1] - gstart;
    const int col0 = cb * 64;
    const int plen = prefix_lens[grp];
    const int per_split = cdiv_i(cdiv_i(plen, prefix_splits), kMinBlockKV) * kMinBlockKV;
    const int t0 = per_split * split, t1 = min(t0 + per_split, plen);
    if (col0 >= gsize * heads || t1 <= t0) return;  // block-uniform: contributes nothing, counts nowhere

    BlockAttn<T, 1> st;
    st.init();
    int qpos[1] = {0};
    const int col = col0 + wid * 16 + hl;
    const bool ok = col < gsize * heads;
    const int req = ok ? gstart + col / heads : -1;
    const int head = kvh * heads + col % heads;
    const int32_t* const plist = kv_indices + (gstart < 64 ? __shfl(spec_indptr, gstart, 64) : kv_indptr[gstart]);
    // every request of the group lists the same slots for the prefix: walk the first request's list
    if constexpr (ROPE) {
      // the position travels with the first tiles' slots, the partial sums and the cos / sin row with their rows
      const int64_t pos = ok ? qp.positions[req] : 0;
      st.template run<false, true, kPrefixPF>(
          lds_all, reinterpret_cast<const uint16_t*>(k_buffer) + (int64_t)kvh * k_stride_h,
          reinterpret_cast<const uint16_t*>(v_buffer) + (int64_t)kvh * v_stride_h, k_stride_n, v_stride_n, plist, t0, t1,
          qpos, 0, sm_scale, logit_cap, [&]() { load_q_rotated<T>(st.qf[0], qp, req, pos, head * 128, ok); });
    } else {
      st.load_q(0, ok ? reinterpret_cast<const uint16_t*>(q) + (int64_t)req * q_stride_b + (int64_t)head * q_stride_h
                      : nullptr);
      st.template run<false, true, kPrefixPF>(
          lds_all, reinterpret_cast<const uint16_t*>(k_buffer) + (int64_t)kvh * k_stride_h,
          reinterpret_cast<const uint16_t*>(v_buffer) + (int64_t)kvh * v_stride_h, k_stride_n, v_stride_n, plist, t0, t1,
          qpos, 0, sm_scale, logit_cap);
    }
    const float tot = st.total(0);
    ATTN_STAMP(1);
    if (ok) {
      const float inv = 1.0f / tot;
      const int ob = (req * Hq + head) * ws.slots + split;
      const auto ro = ws.rsrc_o();
#pragma unroll
      for (int n = 0; n < 8; ++n)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, st.oacc[0][n] * inv), ro,
                                               ob * 512 + (16 * n + 4 * g) * 4, 0, kSc1);
      if (g == 0)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (st.m[0] + log2f(tot)) * 0.6931471805599453f),
                                              ws.rsrc_lse(), ob * 4, 0, kSc1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every store of this wave has been acknowledged
    __syncthreads();
    ATTN_STAMP(2);
    // one count per request this block's columns touch
    const int r_lo = col0 / heads, r_hi = (min(col0 + 64, gsize * heads) - 1) / heads;
    if ((int)threadIdx.x <= r_hi - r_lo)
      __hip_atomic_fetch_add(ws.counters + (int64_t)(gstart + r_lo + threadIdx.x) * Hkv + kvh, 1, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    return;
  }

  // ---- tail block: index = (b * Hkv + kvh) * head_tiles + ht (the kv heads of one request are neighbours in dispatch
  // order: together they fetch whole pool rows)
  // (the first block of every remainder before anybody's second: for short remainders the later blocks have nothing
  // to do and leave; at the end of the grid they are not in the way of blocks with work)
  int i = bid - n_prefix_blocks;
  const int per_round = B * Hkv * head_tiles;
  const int tb = i / per_round;  // which of the remainder's tail blocks
  i -= tb * per_round;
  const int ht = i % head_tiles;
  i /= head_tiles;
  const int kvh = i % Hkv, b = i / Hkv;
  char* vl = lds_all + wid * kTileBytes;  // K goes straight into registers: a wave owns one 8 KiB V tile
  int64_t pos = 0, new_slot = 0;
  if constexpr (ROPE) {
    pos = qp.positions[b];
    new_slot = ld_idx(qp.loc, qp.loc_is64, b);
  }
  int start, len, expect_partials, aps;
  if constexpr (PLAN) {
    // gated: a plan without prefix items hands the whole batch to the per-request kernels launched beside this one
    if ((gated & 1) && plan_hdr[kPlanNItems] == 0) return;
    // The prefix blocks are the launch's critical path (stream, store, count, and only then can anybody merge), while
    // the remainders are a bandwidth-bound walk that ends long before: the tail blocks hold their row requests back
    // for a microsecond so that the prefix blocks' first rows are not queued behind them (measured on 64 x (1024 +
    // 128) tokens: prefix slices streamed in 8.6 instead of 11.0 us, the launch 1.7 us shorter).
    if ((gated >> 8) && plan_hdr[kPlanNItems] > 0) {
      const uint64_t t_hold = wall_clock64();
      while (wall_clock64() - t_hold < (uint64_t)(gated >> 8)) __builtin_amdgcn_s_sleep(8);
    }
    const int4v rec = *reinterpret_cast<const int4v*>(plan_tail + 4 * (int64_t)b);  // uniform address: one scalar load
    start = rec[0];
    len = rec[1];
    expect_partials = geom_bad ? 0 : rec[2];
    aps = rec[3];
    if (geom_bad) len = 0;
  } else {
    const int grp = find_group(group_indptr, num_groups, b);
    const int plen = prefix_lens[grp];
    start = kv_indptr[b] + plen;
    len = kv_indptr[b + 1] - start;
    aps = active_prefix_splits(plen, prefix_splits);
    expect_partials = prefix_blocks_of(b - group_indptr[grp], heads) * aps;
  }
  const int n_tb = MT ? tail_blocks(len, ws.tail_tmax, per_round) : 1;  // block-uniform
  if (tb >= n_tb) return;
  const int splits = tail_splits(len, kWavesPerBlock * n_tb);
  const int per_split = cdiv_i(cdiv_i(len, splits), kMinBlockKV) * kMinBlockKV;
  const int wg = tb * kWavesPerBlock + wid;  // this wave's split of the remainder
  const int s0 = per_split * wg;
  const int s1 = max(min(s0 + per_split, len), s0);
  const bool active = s1 > s0;  // wave-uniform
  const int head_in_group = ht * 16 + hl;
  const bool head_ok = head_in_group < heads;
  const int head = kvh * heads + head_in_group;

  WaveAttn<T> wa;
  wa.init();
  if constexpr (ROPE) {
    // One wave writes the pair's new rows while the slot lists are fetched; every wave then waits for its own loads and
    // stores and for the block, so the wave whose range ends with the new row's slot reads it back from L2 (the
    // vector cache writes through, and nothing on this CU has read the row before).
    if (wid == kWavesPerBlock - 1)
      store_new_kv<T>(qp, b, pos, (Hq + kvh) * 128, (Hq + Hkv + kvh) * 128,
                      reinterpret_cast<uint16_t*>(k_buffer) + new_slot * k_stride_n + (int64_t)kvh * k_stride_h,
                      reinterpret_cast<uint16_t*>(v_buffer) + new_slot * v_stride_n + (int64_t)kvh * v_stride_h);
    load_q_rotated<T>(wa.qf, qp, b, pos, head * 128, head_ok);
    const auto ks = wa.begin_kd(kv_indices + start, s0, s1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    wa.run_kd(vl, reinterpret_cast<const uint16_t*>(k_buffer) + (int64_t)kvh * k_stride_h,
              reinterpret_cast<const uint16_t*>(v_buffer) + (int64_t)kvh * v_stride_h, k_stride_n, v_stride_n,
              kv_indices + start, s0, s1, sm_scale, logit_cap, ks);
  } else if constexpr (RR) {
    // slot lists first, then the wait: the request's query rows and its newest K / V row come from the rope blocks
    const auto ks = wa.begin_kd(kv_indices + start, s0, s1);
    rope_wait(rr, b, ws.error);
    wa.load_q(head_ok ? reinterpret_cast<const uint16_t*>(q) + (int64_t)b * q_stride_b + (int64_t)head * q_stride_h
                      : nullptr);
    wa.run_kd(vl, reinterpret_cast<const uint16_t*>(k_buffer) + (int64_t)kvh * k_stride_h,
              reinterpret_cast<const uint16_t*>(v_buffer) + (int64_t)kvh * v_stride_h, k_stride_n, v_stride_n,
              kv_indices + start, s0, s1, sm_scale, logit_cap, ks);
  } else {
    const uint16_t* qrow = head_ok ? reinterpret_cast<const uint16_t*>(q) + (int64_t)b * q_stride_b + (int64_t)head * q_stride_h
                                   : nullptr;
    if constexpr (PAIR)
      wa.load_q_half(qrow, head_in_group >= (heads >> 1) ? 1 : 0);
    else
      wa.load_q(qrow);
    if constexpr (KV8)  // K rows are parked too (behind the four V tiles), both widened through the table
      wa.run8(lds_all + kBlockAttnLds + wid * kTileBytes, vl, lut,
              reinterpret_cast<const uint8_t*>(k_buffer) + (int64_t)kvh * k_stride_h,
              reinterpret_cast<const uint8_t*>(v_buffer) + (int64_t)kvh * v_stride_h, k_stride_n, v_stride_n,
              kv_indices + start, s0, s1, sm_scale, logit_cap);
    else
      wa.run_kd(vl, reinterpret_cast<const uint16_t*>(k_buffer) + (int64_t)kvh * k_stride_h,
                reinterpret_cast<const uint16_t*>(v_buffer) + (int64_t)kvh * v_stride_h, k_stride_n, v_stride_n,
                kv_indices + start, s0, s1, sm_scale, logit_cap);
  }
  const float wave_lsum = wa.total();
  const float wave_inv = active ? 1.0f / wave_lsum : 0.f;
  const float wave_lse = active ? (wa.m + log2f(wave_lsum)) * 0.6931471805599453f : -INFINITY;
  if (!MT || n_tb == 1) {
    // this wave's partial: O[16 heads][128] fp32 over its V tile, log-sum-exp[16] beside it
    float* po = reinterpret_cast<float*>(vl);
#pragma unroll
    for (int n = 0; n < 8; ++n) *reinterpret_cast<float4v*>(po + hl * 128 + 16 * n + 4 * g) = wa.oacc[n] * wave_inv;
    if (g == 0) tail_lse[wid][hl] = wave_lse;
    ATTN_STAMP(1);
  } else {
    // The remainder spans blocks: every block publishes its waves' partials (write-through, as the prefix blocks do) and
    // draws a ticket; the block that draws the last one merges everything, the others are done.
    if (head_ok) {
      const int ob = (b * Hq + head) * ws.slots + prefix_splits + wg;
      const auto ro = ws.rsrc_o();
#pragma unroll
      for (int n = 0; n < 8; ++n)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, wa.oacc[n] * wave_inv), ro,
                                               ob * 512 + (16 * n + 4 * g) * 4, 0, kSc1);
      if (g == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, wave_lse), ws.rsrc_lse(), ob * 4, 0, kSc1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every store of this wave has been acknowledged
    __syncthreads();
    ATTN_STAMP(1);
    if (threadIdx.x == 0) {
      int* const tk = ws.tickets + ((int64_t)b * Hkv + kvh) * head_tiles + ht;
      const int drawn = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wait_failed = drawn == n_tb - 1 ? 0 : -1;
      if (drawn == n_tb - 1) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
    }
    __syncthreads();
    if (wait_failed < 0) return;  // block-uniform: another block merges
    __syncthreads();              // wait_failed is written again below
  }
  // ---- wait for the pair's prefix partials
  int* const counter = ws.counters + (int64_t)b * Hkv + kvh;
  if (threadIdx.x == 0) {
    const int expect = expect_partials;
    int failed = geom_bad ? 1 : 0;
    if (geom_bad) __hip_atomic_store(ws.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (expect > 0) {
      const uint64_t t_start = wall_clock64();
      uint32_t polls = 0;
      while ((__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (kTailDone - 1)) < expect) {
        if ((++polls & 255u) == 0 && wall_clock64() - t_start > kSharedWaitTicks) {
          failed = 1;
          __hip_atomic_store(ws.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    wait_failed = failed;
  }
  __syncthreads();  // also: every wave's partial is in LDS
  ATTN_STAMP(2);
  const bool failed = wait_failed != 0;
  const int heads_here = min(16, heads - ht * 16);
  const auto ro = ws.rsrc_o();
  const auto rl = ws.rsrc_lse();
  // Merge, in slot order (prefix splits, then the four waves' remainders), two neighbouring elements per thread: a
  // wave covers one head (64 lanes x 2 elements), so a head's block sum is one wave reduction and the partials come
  // in 8-byte pieces.  Four heads per trip; the loads of a trip are one journey to memory (agent-scope loads).
  for (int h0 = 0; h0 < heads_here; h0 += 4) {  // block-uniform trip count
    const int hh = h0 + wid;
    const bool ok = hh < heads_here;
    const int d = 2 * lane;
    const int hq = kvh * heads + ht * 16 + (ok ? hh : 0);
    const int row0 = (b * Hq + hq) * ws.slots;
    float2v tv[kSharedMaxSplits];
    float tl[kSharedMaxSplits];
#pragma unroll
    for (int sp = 0; sp < kSharedMaxSplits; ++sp) {
      const bool live = ok && !failed && sp < aps;
      if (live) {
        tv[sp] = __builtin_bit_cast(float2v, (uint2v)__builtin_amdgcn_raw_buffer_load_b64(ro, (row0 + sp) * 512 + d * 4, 0, kSc1));
        tl[sp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, (row0 + sp) * 4, 0, kSc1));
      } else {
        tv[sp] = float2v{0.f, 0.f};
        tl[sp] = -INFINITY;
      }
    }
    float e_max = -INFINITY, e_sum = 0.f, acc0 = 0.f, acc1 = 0.f;
    auto fold = [&](float2v v, float l) {
      if (l > -INFINITY) {
        const float n_max = fmaxf(l, e_max);
        const float old = expf(e_max - n_max), w = expf(l - n_max);
        // explicit fused operations: left to the compiler, the two instantiations of this kernel contracted
        // "acc * old + w * v" differently and parted by an ulp
        acc0 = fmaf(w, v[0], acc0 * old);
        acc1 = fmaf(w, v[1], acc1 * old);
        e_sum = fmaf(e_sum, old, w);
        e_max = n_max;
      }
    };
#pragma unroll
    for (int sp = 0; sp < kSharedMaxSplits; ++sp) fold(tv[sp], tl[sp]);
    if (!MT || n_tb == 1) {  // the remainder's four partials are this block's own, in LDS
#pragma unroll
      for (int w4 = 0; w4 < kWavesPerBlock; ++w4)
        fold(*reinterpret_cast<const float2v*>(lds_all + w4 * kTileBytes + ((ok ? hh : 0) * 128 + d) * 4),
             tail_lse[w4][ok ? hh : 0]);
    } else {          // published by n_tb blocks: one more trip, in split order
      for (int w0 = 0; w0 < kWavesPerBlock * n_tb; w0 += kWavesPerBlock) {  // block-uniform
        float2v uv[kWavesPerBlock];
        float ul[kWavesPerBlock];
#pragma unroll
        for (int w4 = 0; w4 < kWavesPerBlock; ++w4) {
          const int r = row0 + prefix_splits + w0 + w4;
          if (ok && !failed) {
            uv[w4] = __builtin_bit_cast(float2v, (uint2v)__builtin_amdgcn_raw_buffer_load_b64(ro, r * 512 + d * 4, 0, kSc1));
            ul[w4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, r * 4, 0, kSc1));
          } else {
            uv[w4] = float2v{0.f, 0.f};
            ul[w4] = -INFINITY;
          }
        }
#pragma unroll
        for (int w4 = 0; w4 < kWavesPerBlock; ++w4) fold(uv[w4], ul[w4]);
      }
    }
    const float r0 = failed ? __builtin_nanf("") : acc0 / e_sum;
    const float r1 = failed ? __builtin_nanf("") : acc1 / e_sum;
    if constexpr (PAIR) {
      // the head's output is the half of the 128-wide row that faces its own kv head: rows of 64
      const int half = (ht * 16 + (ok ? hh : 0)) >= (heads >> 1) ? 1 : 0;
      if (ok && (d >> 6) == half) {
        st<T>(o, (int64_t)b * o_stride_b + (int64_t)hq * o_stride_h + d - 64 * half, r0);
        st<T>(o, (int64_t)b * o_stride_b + (int64_t)hq * o_stride_h + d - 64 * half + 1, r1);
      }
    } else if (ok) {
      st<T>(o, (int64_t)b * o_stride_b + (int64_t)hq * o_stride_h + d, r0);
      st<T>(o, (int64_t)b * o_stride_b + (int64_t)hq * o_stride_h + d + 1, r1);
    }
    if (xsum_out) {
      const float r = wave_sum(ok ? Conv<T>::to_f32(Conv<T>::from_f32(r0)) + Conv<T>::to_f32(Conv<T>::from_f32(r1)) : 0.f);
      if (lane == 0 && ok) xsum_out[(int64_t)hq * xs_stride + b] = r;
    }
  }
  ATTN_STAMP(3);
  // the pair's last tail block leaves the counter at zero for the next launch
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(counter, kTailDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old >> 16) == head_tiles - 1) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------
// Shared-prefix decode for absorbed MLA (latent rows of 576, values = their first 512 elements), driven by the same
// per-step plan, in two launches:
//   mla_shared_stage1  prefix blocks: half an item's 64 (request, head) columns -- two tiles of 16 -- against the
//                      item's slice of the group's shared rows, which are fetched once per 32 columns whatever
//                      request they belong to (the per-request kernel fetches them once per request and head tile);
//                      tail blocks: (request, head tile, split w < 4) over the request's private remainder.
//                      Partials go to rows [(b * Hq + head) * slots + s], s = prefix split, or prefix_splits + w.
//   mla_shared_merge   one block per (request, head): the partials of its live slots, in slot order.
// No block waits for another: the second launch is the hand-over.
// ---------------------------------------------------------------------------------------------
static constexpr int kMlaTailSplits = 4;

template <typename T, int HT, bool CAP>
__global__ void __launch_bounds__(256, HT == 1 ? 3 : 2) mla_shared_stage1(
    const void* __restrict__ q, const void* __restrict__ kv_buffer, const int32_t* __restrict__ kv_indices,
    const int32_t* __restrict__ plan_hdr, const int32_t* __restrict__ plan_tail, const int32_t* __restrict__ plan_items,
    float* __restrict__ logits, float* __restrict__ lse, int* __restrict__ error, int prefix_splits, int n_prefix_blocks,
    int B, int Hq, int head_tiles, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, float sm_scale,
    float logit_cap, int gated) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int hl = lane & 15;
  const int heads = Hq;  // one latent "kv head"
  const int slots = prefix_splits + kMlaTailSplits;
  if (plan_hdr[kPlanHeads] != heads || plan_hdr[kPlanSplits] != prefix_splits || plan_hdr[kPlanB] != B) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;  // the merge sees the same words and writes NaN
  }
  const int n_items = plan_hdr[kPlanNItems];
  if (gated && n_items == 0) return;  // nobody shares anything: the per-request kernels launched beside this one run
  const uint16_t* kbase = reinterpret_cast<const uint16_t*>(kv_buffer);
  constexpr int kColsPerBlock = 16 * HT;          // 64 columns of an item over 64 / kColsPerBlock blocks
  constexpr int kParts = 64 / kColsPerBlock;
  MlaRun<T, HT, CAP> r;
  if ((int)blockIdx.x < n_prefix_blocks) {
    const int part = blockIdx.x % kParts;
    const int items_per_round = n_prefix_blocks / kParts;
    for (int j = blockIdx.x / kParts; j < n_items; j += items_per_round) {  // block-uniform
      const int32_t* it = plan_items + (int64_t)j * kPlanItemInts;
      const int list_off = it[0], t0 = it[1], t1 = it[2], cb = it[3], split = it[4];
      const uint16_t* qrow[HT];
      int req[HT], head[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const int c = part * kColsPerBlock + 16 * t + hl;
        req[t] = it[16 + c];
        head[t] = (cb * 64 + c) % heads;
        qrow[t] = req[t] >= 0 ? reinterpret_cast<const uint16_t*>(q) + (int64_t)req[t] * q_stride_b + (int64_t)head[t] * q_stride_h
                              : nullptr;
      }
      // a part of the item without a live column has nothing to do (the last item of a group)
      const int first = it[16 + part * kColsPerBlock];  // owners are sorted: dead columns come last
      if (first < 0 || t1 <= t0) continue;              // block-uniform
      r.run(lds, qrow, kbase, k_stride_n, kv_indices + list_off, t0, t1, sm_scale, logit_cap);
#pragma unroll
      for (int t = 0; t < HT; ++t)
        r.store(t, req[t] >= 0, ((int64_t)req[t] * Hq + head[t]) * slots + split, logits, lse);
      __syncthreads();  // the next item's first tile is parked over this one's
    }
    return;
  }
  int i = blockIdx.x - n_prefix_blocks;
  const int w = i % kMlaTailSplits;
  i /= kMlaTailSplits;
  const int ht = i % head_tiles, b = i / head_tiles;
  const int4v rec = *reinterpret_cast<const int4v*>(plan_tail + 4 * (int64_t)b);
  const int start = rec[0], len = rec[1];
  const int splits = tail_splits(len, kMlaTailSplits);
  const int per_split = cdiv_i(cdiv_i(len, splits), kMinBlockKV) * kMinBlockKV;
  const int s0 = per_split * w, s1 = min(s0 + per_split, len);
  if (w >= splits || s1 <= s0) return;  // block-uniform
  const uint16_t* qrow[HT];
  bool ok[HT];
  int head[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    head[t] = (ht * HT + t) * 16 + hl;
    ok[t] = head[t] < heads;
    qrow[t] = ok[t] ? reinterpret_cast<const uint16_t*>(q) + (int64_t)b * q_stride_b + (int64_t)head[t] * q_stride_h : nullptr;
  }
  r.run(lds, qrow, kbase, k_stride_n, kv_indices + start, s0, s1, sm_scale, logit_cap);
#pragma unroll
  for (int t = 0; t < HT; ++t) r.store(t, ok[t], ((int64_t)b * Hq + head[t]) * slots + prefix_splits + w, logits, lse);
}

template <typename T>
__global__ void __launch_bounds__(128) mla_shared_merge(const float* __restrict__ logits, const float* __restrict__ lse,
                                                        void* __restrict__ o, const int32_t* __restrict__ plan_hdr,
                                                        const int32_t* __restrict__ plan_tail, int prefix_splits, int B,
                                                        int Hq, int64_t o_stride_b, int64_t o_stride_h, int gated) {
  const int b = blockIdx.x, h = blockIdx.y;
  const bool bad = plan_hdr[kPlanHeads] != Hq || plan_hdr[kPlanSplits] != prefix_splits || plan_hdr[kPlanB] != B;
  if (!bad && gated && plan_hdr[kPlanNItems] == 0) return;
  const int4v rec = *reinterpret_cast<const int4v*>(plan_tail + 4 * (int64_t)b);
  const int len = rec[1], aps = rec[3];
  const int splits = tail_splits(len, kMlaTailSplits);
  const int per_split = cdiv_i(cdiv_i(len, splits), kMinBlockKV) * kMinBlockKV;
  const int slots = prefix_splits + kMlaTailSplits;
  const int64_t base = ((int64_t)b * Hq + h) * slots;
  const int d = 4 * threadIdx.x;  // 128 threads x 4 elements = 512
  float e_sum = 0.f, e_max = -INFINITY;
  float4v acc = float4v{0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < slots; ++s) {
    const bool live = s < prefix_splits ? s < aps : (s - prefix_splits < splits && per_split * (s - prefix_splits) < len);
    if (live) {
      const float4v tv = *reinterpret_cast<const float4v*>(logits + (base + s) * kMlaDv + d);
      const float tl = lse[base + s];
      const float n_max = fmaxf(tl, e_max);
      const float old = expf(e_max - n_max), wgt = expf(tl - n_max);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = fmaf(wgt, tv[r], acc[r] * old);
      e_sum = fmaf(e_sum, old, wgt);
      e_max = n_max;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    st<T>(o, (int64_t)b * o_stride_b + (int64_t)h * o_stride_h + d + r, bad ? __builtin_nanf("") : acc[r] / e_sum);
}

}  // namespace sglk

using namespace sglk;

extern "C" {

int sglk_create_kv_indices(const int32_t* req_to_token, int64_t row_stride, const void* req_pool_indices, int idx_is64,
                           const void* lens, int lens_is64, const int32_t* kv_indptr, const void* kv_start,
                           int start_is64, int32_t* kv_indices, int bs, void* stream) {
  SGLK_CHECK(req_to_token && req_pool_indices && lens && kv_indptr, "create_kv_indices: null pointer");
  SGLK_CHECK(bs >= 0, "create_kv_indices: negative batch");
  if (bs == 0) return SGLK_OK;
  SGLK_CHECK(kv_indices, "create_kv_indices: null output");
  hipLaunchKernelGGL(create_kv_indices_kernel, dim3(bs, 4), dim3(256), 0, (hipStream_t)stream, req_to_token,
                     row_stride, req_pool_indices, idx_is64, lens, lens_is64, kv_indptr, kv_start, start_is64,
                     kv_indices);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_kv_indptr_cumsum(const void* seq_lens, int lens_is64, int32_t* kv_indptr, int bs, void* stream) {
  SGLK_CHECK(seq_lens && kv_indptr && bs >= 0, "kv_indptr_cumsum: bad arguments");
  hipLaunchKernelGGL(cumsum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, seq_lens, lens_is64, kv_indptr, bs);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_get_num_kv_splits(int32_t* num_kv_splits, const void* seq_lens, int lens_is64, int num_seq, int num_group,
                           int num_head, int num_kv_head, int max_kv_splits, int device_core_count, void* stream) {
  SGLK_CHECK(num_kv_splits && seq_lens, "get_num_kv_splits: null pointer");
  SGLK_CHECK(num_seq > 0 && num_group > 0 && num_head > 0 && num_kv_head > 0 && max_kv_splits > 0,
             "get_num_kv_splits: bad sizes");
  hipLaunchKernelGGL(num_kv_splits_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, num_kv_splits, seq_lens,
                     lens_is64, num_seq, num_group, num_head, num_kv_head, max_kv_splits, device_core_count);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_decode_metadata(const int32_t* req_to_token, int64_t row_stride, const void* req_pool_indices, int idx_is64,
                         const void* seq_lens, int lens_is64, int32_t* kv_indptr, int32_t* kv_indices,
                         int32_t* num_kv_splits, int bs, int num_head, int num_kv_head, int max_kv_splits,
                         int device_core_count, void* stream) {
  SGLK_CHECK(bs > 0, "decode_metadata: empty batch");
  SGLK_CHECK(req_to_token && req_pool_indices && seq_lens && kv_indptr && kv_indices && num_kv_splits,
             "decode_metadata: null pointer");
  SGLK_CHECK(num_head > 0 && num_kv_head > 0 && max_kv_splits > 0 && device_core_count > 0, "decode_metadata: bad argument");
  const int ysl = bs >= 128 ? 1 : (bs >= 32 ? 4 : 8);  // slices per row: keep a few hundred blocks in flight
  hipLaunchKernelGGL(decode_metadata_kernel, dim3(bs, ysl), dim3(256), 0, (hipStream_t)stream, req_to_token, row_stride,
                     req_pool_indices, idx_is64, seq_lens, lens_is64, kv_indptr, kv_indices, num_kv_splits, bs, num_head,
                     num_kv_head, max_kv_splits, device_core_count);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_prepare_for_decode(const int32_t* req_to_token, int64_t row_stride, const void* req_pool_indices, int idx_is64,
                            void* seq_lens, int lens_is64, int64_t* positions, void* out_cache_loc, int loc_is64, int bs,
                            void* stream) {
  SGLK_CHECK(bs >= 0, "prepare_for_decode: negative batch");
  if (bs == 0) return SGLK_OK;
  SGLK_CHECK(req_to_token && req_pool_indices && seq_lens && positions && out_cache_loc, "prepare_for_decode: null pointer");
  hipLaunchKernelGGL(prepare_for_decode_kernel, dim3((bs + 255) / 256), dim3(256), 0, (hipStream_t)stream, req_to_token,
                     row_stride, req_pool_indices, idx_is64, seq_lens, lens_is64, positions, out_cache_loc, loc_is64, bs);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_common_prefix_len(const int32_t* req_to_token, int64_t row_stride, const void* req_pool_indices, int idx_is64,
                           const void* seq_lens, int lens_is64, int bs, int32_t* out, void* stream) {
  SGLK_CHECK(req_to_token && req_pool_indices && seq_lens && out && bs > 0, "common_prefix_len: bad arguments");
  hipLaunchKernelGGL(set_i32_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, out, 0x7fffffff);
  hipLaunchKernelGGL(common_prefix_kernel, dim3(bs), dim3(256), 0, (hipStream_t)stream, req_to_token, row_stride,
                     req_pool_indices, idx_is64, seq_lens, lens_is64, bs, out);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

static int decode_attention_impl(const void* q, const void* k_buffer, const void* v_buffer, void* o, const int32_t* kv_indptr,
                             const int32_t* kv_indices, float* attn_logits, float* attn_lse, const int32_t* num_kv_splits,
                             int max_kv_splits, int B, int Hq, int Hkv, int D, int Dv, int64_t q_stride_b,
                             int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n,
                             int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
                             int dtype, float* xsum_out, int64_t xs_stride, void* stream, const int32_t* gate) {
  const int kvf = dtype >> SGLK_KV_SHIFT;  // fp8 pool format, 0: pool elements of the query type
  dtype &= (1 << SGLK_KV_SHIFT) - 1;
  SGLK_CHECK(kvf == 0 || dtype == SGLK_F16 || dtype == SGLK_BF16, "decode_attention: an fp8 pool needs fp16 / bf16 queries");
  SGLK_CHECK(!xsum_out || (Dv == 128 && xs_stride >= B), "decode_attention: block sums need Dv == 128 and xs_stride >= B");
  SGLK_CHECK(q && k_buffer && v_buffer && o && kv_indptr && kv_indices && attn_logits && attn_lse && num_kv_splits,
             "decode_attention: null pointer");
  SGLK_CHECK(B >= 0 && Hq > 0 && Hkv > 0 && D > 0 && Dv > 0, "decode_attention: bad shape");
  SGLK_CHECK(Hq % Hkv == 0, "decode_attention: Hq %d not a multiple of Hkv %d", Hq, Hkv);
  SGLK_CHECK(max_kv_splits > 0, "decode_attention: max_kv_splits must be positive");
  SGLK_CHECK(Dv <= 1024 && D <= 4096, "decode_attention: head size too large");
  if (B == 0) return SGLK_OK;
  hipStream_t s = (hipStream_t)stream;
  const int group = Hq / Hkv;
  const bool fast = kvf == 0 && (dtype == SGLK_F16 || dtype == SGLK_BF16) && D == 128 && Dv == 128 && aligned16(q) &&
                    aligned16(k_buffer) && aligned16(v_buffer) && aligned16(attn_logits) && (q_stride_b % 8 == 0) &&
                    (q_stride_h % 8 == 0) && (k_stride_n % 8 == 0) && (k_stride_h % 8 == 0) && (v_stride_n % 8 == 0) &&
                    (v_stride_h % 8 == 0);
  // head size 64 with an even number of kv heads whose rows are contiguous: pairs of kv heads on the 128-wide kernel
  const bool pair64 = kvf == 0 && (dtype == SGLK_F16 || dtype == SGLK_BF16) && D == 64 && Dv == 64 && Hkv % 2 == 0 &&
                      k_stride_h == 64 && v_stride_h == 64 && aligned16(q) && aligned16(k_buffer) && aligned16(v_buffer) &&
                      aligned16(attn_logits) && (q_stride_b % 8 == 0) && (q_stride_h % 8 == 0) && (k_stride_n % 8 == 0) &&
                      (v_stride_n % 8 == 0);
  // absorbed MLA: V is a view of the first 512 elements of the 576-element K rows (memory_pool.py:492-521)
  const bool mla = kvf == 0 && (dtype == SGLK_F16 || dtype == SGLK_BF16) && D == kMlaD && Dv == kMlaDv && v_buffer == k_buffer &&
                   v_stride_n == k_stride_n && v_stride_h == k_stride_h && aligned16(q) && aligned16(k_buffer) &&
                   aligned16(attn_logits) && (q_stride_b % 8 == 0) && (q_stride_h % 8 == 0) && (k_stride_n % 8 == 0) &&
                   (k_stride_h % 8 == 0);
  // an fp8 pool of head size 128 on the same tile kernel: rows widened through a table on their way into LDS
  auto al8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
  const bool fast8 = kvf != 0 && D == 128 && Dv == 128 && aligned16(q) && al8(k_buffer) && al8(v_buffer) &&
                     aligned16(attn_logits) && (q_stride_b % 8 == 0) && (q_stride_h % 8 == 0) && (k_stride_n % 8 == 0) &&
                     (k_stride_h % 8 == 0) && (v_stride_n % 8 == 0) && (v_stride_h % 8 == 0);
  SGLK_CHECK(!gate || fast || fast8 || pair64 || mla, "decode_attention: a gated launch needs one of the tile kernels");
  if (fast8) {
    const int head_tiles = (group + 15) / 16;
    dim3 grid(B, Hkv * head_tiles, (max_kv_splits + kWavesPerBlock - 1) / kWavesPerBlock);
    const size_t lds = (size_t)kWavesPerBlock * kLdsPerWave + 512;
    SGLK_DISPATCH_HALF(dtype, T, {
      SGLK_DISPATCH_KV(kvf, T, KV, {
        hipLaunchKernelGGL((decode_stage1_mfma128<T, KV>), grid, dim3(64 * kWavesPerBlock), lds, s, q, k_buffer, v_buffer,
                           kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits, max_kv_splits, Hq, Hkv, head_tiles,
                           q_stride_b, q_stride_h, k_stride_n, k_stride_h, v_stride_n, v_stride_h, sm_scale, logit_cap, 0,
                           gate);
      });
    });
  } else if (mla) {
    if (group >= 32) {  // two tiles of 16 heads per block
      const int head_tiles = (group + 31) / 32;
      dim3 grid(B, Hkv * head_tiles, max_kv_splits);
#define SGLK_MLA_LAUNCH(HT, CAP)                                                                                         \
  hipLaunchKernelGGL((decode_stage1_mla<T, HT, CAP>), grid, dim3(256), kMlaTileBytes + HT * kMlaScoreBytes, s, q, k_buffer, \
                     kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits, max_kv_splits, Hq, Hkv, head_tiles,      \
                     q_stride_b, q_stride_h, k_stride_n, k_stride_h, sm_scale, logit_cap, gate)
      SGLK_DISPATCH_HALF(dtype, T, {
        if (logit_cap > 0.f)
          SGLK_MLA_LAUNCH(2, true);
        else
          SGLK_MLA_LAUNCH(2, false);
      });
    } else {
      const int head_tiles = (group + 15) / 16;
      dim3 grid(B, Hkv * head_tiles, max_kv_splits);
      SGLK_DISPATCH_HALF(dtype, T, {
        if (logit_cap > 0.f)
          SGLK_MLA_LAUNCH(1, true);
        else
          SGLK_MLA_LAUNCH(1, false);
      });
#undef SGLK_MLA_LAUNCH
    }
  } else if (fast) {
    const int head_tiles = (group + 15) / 16;
    dim3 grid(B, Hkv * head_tiles, (max_kv_splits + kWavesPerBlock - 1) / kWavesPerBlock);
    const size_t lds = (size_t)kWavesPerBlock * kLdsPerWave;
    SGLK_DISPATCH_HALF(dtype, T,
                       hipLaunchKernelGGL((decode_stage1_mfma128<T>), grid, dim3(64 * kWavesPerBlock), lds, s, q, k_buffer,
                                          v_buffer, kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits,
                                          max_kv_splits, Hq, Hkv, head_tiles, q_stride_b, q_stride_h, k_stride_n,
                                          k_stride_h, v_stride_n, v_stride_h, sm_scale, logit_cap, 0, gate));
  } else if (pair64) {
    // head size 64 on the 128-wide tile kernel: Hkv / 2 pairs of kv heads, 2 * group query heads per pair
    const int head_tiles = (2 * group + 15) / 16;
    dim3 grid(B, (Hkv / 2) * head_tiles, (max_kv_splits + kWavesPerBlock - 1) / kWavesPerBlock);
    const size_t lds = (size_t)kWavesPerBlock * kLdsPerWave;
    SGLK_DISPATCH_HALF(dtype, T,
                       hipLaunchKernelGGL((decode_stage1_mfma128<T>), grid, dim3(64 * kWavesPerBlock), lds, s, q, k_buffer,
                                          v_buffer, kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits,
                                          max_kv_splits, Hq, Hkv / 2, head_tiles, q_stride_b, q_stride_h, k_stride_n,
                                          (int64_t)128, v_stride_n, (int64_t)128, sm_scale, logit_cap, group, gate));
  } else {
    dim3 grid(B, Hq, max_kv_splits);
    const size_t lds = (size_t)D * sizeof(float);
    const int per_lane = (Dv + 63) / 64;
#define SGLK_GEN(N)                                                                                                \
  hipLaunchKernelGGL((decode_stage1_generic<T, KV, N>), grid, dim3(64), lds, s, q, k_buffer, v_buffer, kv_indptr,  \
                     kv_indices, attn_logits, attn_lse, num_kv_splits, max_kv_splits, Hq, Hkv, D, Dv, q_stride_b,  \
                     q_stride_h, k_stride_n, k_stride_h, v_stride_n, v_stride_h, sm_scale, logit_cap)
#define SGLK_GEN_ANY()   \
  if (per_lane <= 1) {   \
    SGLK_GEN(1);         \
  } else if (per_lane <= 2) { \
    SGLK_GEN(2);         \
  } else if (per_lane <= 4) { \
    SGLK_GEN(4);         \
  } else if (per_lane <= 8) { \
    SGLK_GEN(8);         \
  } else {               \
    SGLK_GEN(16);        \
  }
    if (kvf == 0) {
      SGLK_DISPATCH_FLOAT(dtype, T, {
        using KV = T;
        SGLK_GEN_ANY();
      });
    } else {
      SGLK_DISPATCH_HALF(dtype, T, { SGLK_DISPATCH_KV(kvf, T, KV, { SGLK_GEN_ANY(); }); });
    }
#undef SGLK_GEN_ANY
#undef SGLK_GEN
  }
  SGLK_LAUNCH_CHECK();
  SGLK_DISPATCH_FLOAT(dtype, T,
                      hipLaunchKernelGGL((decode_stage2<T>), dim3(B, Hq), dim3(128), 0, s, attn_logits, attn_lse, o,
                                         kv_indptr, num_kv_splits, max_kv_splits, Hq, Dv, o_stride_b, o_stride_h,
                                         xsum_out, xs_stride, gate));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_decode_attention_xs(const void* q, const void* k_buffer, const void* v_buffer, void* o, const int32_t* kv_indptr,
                             const int32_t* kv_indices, float* attn_logits, float* attn_lse, const int32_t* num_kv_splits,
                             int max_kv_splits, int B, int Hq, int Hkv, int D, int Dv, int64_t q_stride_b,
                             int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n,
                             int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
                             int dtype, float* xsum_out, int64_t xs_stride, void* stream) {
  return decode_attention_impl(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits,
                               max_kv_splits, B, Hq, Hkv, D, Dv, q_stride_b, q_stride_h, k_stride_n, k_stride_h, v_stride_n,
                               v_stride_h, o_stride_b, o_stride_h, sm_scale, logit_cap, dtype, xsum_out, xs_stride, stream,
                               nullptr);
}

int sglk_decode_attention(const void* q, const void* k_buffer, const void* v_buffer, void* o, const int32_t* kv_indptr,
                          const int32_t* kv_indices, float* attn_logits, float* attn_lse, const int32_t* num_kv_splits,
                          int max_kv_splits, int B, int Hq, int Hkv, int D, int Dv, int64_t q_stride_b,
                          int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n,
                          int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
                          int dtype, void* stream) {
  return sglk_decode_attention_xs(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits,
                                  max_kv_splits, B, Hq, Hkv, D, Dv, q_stride_b, q_stride_h, k_stride_n, k_stride_h,
                                  v_stride_n, v_stride_h, o_stride_b, o_stride_h, sm_scale, logit_cap, dtype, nullptr, 0,
                                  stream);
}

// Workspace of the shared-prefix path: [counters: B * Hq int32 + one error word, rounded up to 256 bytes]
// [O partials: B * Hq * max_kv_splits * Dv fp32][log-sum-exp: B * Hq * max_kv_splits fp32].
// counters [B * Hq] (B * Hkv used), the error word, then the tail tickets [B * Hq] (B * Hkv * head_tiles used)
static int64_t shared_counter_bytes(int B, int Hq) { return ((((int64_t)2 * B * Hq + 2) * 4 + 255) / 256) * 256; }
// most tail blocks per (request, kv head, head tile) of the planned kernel: as many wave splits as max_kv_splits allows
static int shared_tail_tmax(int max_kv_splits) { return std::max(1, max_kv_splits / kWavesPerBlock); }

int64_t sglk_decode_shared_workspace_bytes(int B, int Hq, int Dv, int max_kv_splits) {
  if (B <= 0 || Hq <= 0 || Dv <= 0 || max_kv_splits <= 0) return 0;
  // counters, then partial outputs + log-sum-exps, then one split count per request (used by the shapes that take
  // the per-request kernels).  The one-launch kernel keeps partial rows of 128 floats whatever the head size (head
  // size 64 runs on pairs of kv heads, 128 wide).
  Dv = std::max(Dv, 128);
  if (Dv == kMlaDv)
    max_kv_splits += kMlaTailSplits;  // the absorbed-MLA form keeps the remainders' partials there too
  else
    max_kv_splits += kWavesPerBlock * shared_tail_tmax(max_kv_splits);  // remainders that span blocks publish theirs
  return shared_counter_bytes(B, Hq) + (int64_t)B * Hq * max_kv_splits * (Dv + 1) * (int64_t)sizeof(float) +
         (((int64_t)B * 4 + 255) / 256) * 256;
}

int sglk_decode_shared_workspace_init(void* workspace, int B, int Hq, void* stream) {
  SGLK_CHECK(workspace && B > 0 && Hq > 0, "decode_shared_workspace_init: bad arguments");
  SGLK_HIP(hipMemsetAsync(workspace, 0, (size_t)shared_counter_bytes(B, Hq), (hipStream_t)stream));
  return SGLK_OK;
}

int sglk_decode_shared_workspace_status(const void* workspace, int B, int Hq) {
  if (!workspace || B <= 0 || Hq <= 0) return fail(SGLK_EINVAL, "decode_shared_workspace_status: bad arguments");
  int word = 0;
  SGLK_HIP(hipMemcpy(&word, reinterpret_cast<const int*>(workspace) + (int64_t)B * Hq, sizeof(int), hipMemcpyDeviceToHost));
  return word != 0 ? 1 : 0;
}

#ifdef SGLK_ATTN_STAMPS
SGLK_API int sglk_debug_attn_stamps(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_stamps), sizeof(long long) * 1024 * 8);
}
#endif

static int launch_shared_prefix(const void* q, const QkvPartials* qp, void* k_buffer, void* v_buffer, void* o,
                                const int32_t* kv_indptr, const int32_t* kv_indices, const int32_t* group_indptr,
                                const int32_t* prefix_lens, int G, const int32_t* plan, float* workspace,
                                int max_kv_splits, int B, int Hq,
                                int Hkv, int D, int Dv, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n,
                                int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b,
                                int64_t o_stride_h, float sm_scale, float logit_cap, int dtype, float* xsum_out,
                                int64_t xs_stride, void* stream, bool gated = false, const QkvPartials* rope_src = nullptr,
                                void* rope_q_out = nullptr, int rope_seq = 0) {
  SGLK_CHECK(!xsum_out || xs_stride >= B, "decode_attention_shared_prefix: xs_stride smaller than B");
  SGLK_CHECK((q || qp) && k_buffer && v_buffer && o && kv_indptr && kv_indices && workspace &&
                 (plan || (group_indptr && prefix_lens)),
             "decode_attention_shared_prefix: null pointer");
  SGLK_CHECK(B >= 0 && (plan || G > 0) && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "decode_attention_shared_prefix: bad shape");
  SGLK_CHECK(max_kv_splits > 0 && max_kv_splits <= kSharedMaxSplits,
             "decode_attention_shared_prefix: max_kv_splits must be in 1..16");
  if (B == 0) return SGLK_OK;
  auto mult8 = [](int64_t v) { return v % 8 == 0; };
  const int kvf = dtype >> SGLK_KV_SHIFT;  // fp8 pool format, 0: pool elements of the query type
  const int qdtype = dtype & ((1 << SGLK_KV_SHIFT) - 1);
  const bool half_q = qdtype == SGLK_F16 || qdtype == SGLK_BF16;
  const bool fast = kvf == 0 && half_q && D == 128 && Dv == 128 && (qp || aligned16(q)) &&
                    aligned16(k_buffer) && aligned16(v_buffer) && aligned16(workspace) &&
                    (qp || (mult8(q_stride_b) && mult8(q_stride_h))) && mult8(k_stride_n) && mult8(k_stride_h) &&
                    mult8(v_stride_n) && mult8(v_stride_h);
  // the planned launch also takes fp8 pools of head size 128 (rows widened through a table on their way into LDS) and
  // head size 64 with an even number of kv heads whose rows are contiguous (pairs of kv heads as one 128-wide head)
  auto al8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
  const bool plain_planned = plan && !qp && !rope_src && half_q && q && aligned16(q) && aligned16(workspace) &&
                             mult8(q_stride_b) && mult8(q_stride_h) && mult8(k_stride_n) && mult8(v_stride_n);
  const bool fast8 = plain_planned && kvf != 0 && D == 128 && Dv == 128 && al8(k_buffer) && al8(v_buffer) &&
                     mult8(k_stride_h) && mult8(v_stride_h);
  const bool pair64 = plain_planned && kvf == 0 && D == 64 && Dv == 64 && Hkv % 2 == 0 && k_stride_h == 64 &&
                      v_stride_h == 64 && aligned16(k_buffer) && aligned16(v_buffer) && !xsum_out;
  hipStream_t s = (hipStream_t)stream;
  // absorbed MLA (one latent kv head, values a view of the key rows): the plan's items through the MLA tile kernel
  const bool mla = plain_planned && kvf == 0 && D == kMlaD && Dv == kMlaDv && Hkv == 1 && v_buffer == k_buffer &&
                   v_stride_n == k_stride_n && aligned16(k_buffer) && !xsum_out;
  if (mla) {
    SGLK_CHECK(B <= kPlanMaxB, "decode_attention_shared_prefix: a plan covers at most %d requests", kPlanMaxB);
    SGLK_CHECK(aligned16(plan), "decode_attention_shared_prefix: plan must be 16-byte aligned");
    const int heads = Hq;
    const int splits = shared_prefix_splits(B, heads, 1, max_kv_splits);
    const int slots = splits + kMlaTailSplits;
    const int col_blocks = (B * heads + 63) / 64;
    constexpr int HT = 2;
    const int n_prefix = (64 / (16 * HT)) * splits * col_blocks;
    const int head_tiles = (heads + 16 * HT - 1) / (16 * HT);
    const int n_tail = B * head_tiles * kMlaTailSplits;
    const PlanView pv = plan_view(const_cast<int32_t*>(plan), B, plan_max_items(B, heads, splits));
    int* error = reinterpret_cast<int*>(workspace) + (int64_t)B * Hq;
    float* logits = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + shared_counter_bytes(B, Hq));
    float* lse = logits + (int64_t)B * Hq * slots * kMlaDv;
    const size_t lds = kMlaTileBytes + HT * kMlaScoreBytes;
#define SGLK_MLA_SHARED(CAP)                                                                                            \
  hipLaunchKernelGGL((mla_shared_stage1<T, HT, CAP>), dim3(n_prefix + n_tail), dim3(256), lds, s, q, k_buffer, kv_indices, \
                     pv.hdr, pv.tail, pv.items, logits, lse, error, splits, n_prefix, B, Hq, head_tiles, q_stride_b,       \
                     q_stride_h, k_stride_n, sm_scale, logit_cap, gated ? 1 : 0)
    SGLK_DISPATCH_HALF(qdtype, T, {
      if (logit_cap > 0.f)
        SGLK_MLA_SHARED(true);
      else
        SGLK_MLA_SHARED(false);
    });
#undef SGLK_MLA_SHARED
    SGLK_LAUNCH_CHECK();
    SGLK_DISPATCH_HALF(qdtype, T,
                       hipLaunchKernelGGL((mla_shared_merge<T>), dim3(B, Hq), dim3(128), 0, s, logits, lse, o, pv.hdr, pv.tail,
                                          splits, B, Hq, o_stride_b, o_stride_h, gated ? 1 : 0));
    SGLK_LAUNCH_CHECK();
    return SGLK_OK;
  }
  if (!fast && !fast8 && !pair64) {
    // Other shapes (fp32, absorbed MLA without a plan, layouts the tile kernels do not take, and everything but the plain
    // planned form for fp8 pools and head size 64): the same result from the per-request kernels over the same
    // lists, with the groups ignored: a group's shared rows are then fetched once per request, from L2 after the
    // first.  The workspace holds their partials ([B, Hq, S, Dv] + [B, Hq, S]) and one split count per request.
    if (qp)
      return fail(SGLK_EUNSUP, "decode_attention_shared_prefix_rope: only 16-bit types with head size 128 (got D=%d Dv=%d)",
                  D, Dv);
    SGLK_CHECK(aligned16(workspace), "decode_attention_shared_prefix: workspace must be 16-byte aligned");
    float* logits = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + shared_counter_bytes(B, Hq));
    float* lse = logits + (int64_t)B * Hq * max_kv_splits * Dv;
    int32_t* ns = reinterpret_cast<int32_t*>(lse + (int64_t)B * Hq * max_kv_splits);
    hipLaunchKernelGGL(fill_i32_kernel, dim3((B + 255) / 256), dim3(256), 0, s, ns, max_kv_splits, B);
    SGLK_LAUNCH_CHECK();
    return sglk_decode_attention_xs(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, logits, lse, ns, max_kv_splits, B, Hq,
                                    Hkv, D, Dv, q_stride_b, q_stride_h, k_stride_n, k_stride_h, v_stride_n, v_stride_h,
                                    o_stride_b, o_stride_h, sm_scale, logit_cap, dtype, xsum_out, xs_stride, stream);
  }
  if (pair64) {  // from here on a pair of kv heads is one kv head of 128 elements
    Hkv /= 2;
    k_stride_h = v_stride_h = 128;
  }
  const int heads = Hq / Hkv;
  static const int env_sp = getenv("SGLK_SHARED_SPLITS") ? atoi(getenv("SGLK_SHARED_SPLITS")) : 0;
  static const int env_delay = getenv("SGLK_TAIL_DELAY") ? atoi(getenv("SGLK_TAIL_DELAY")) : 100;  // 10 ns ticks
  const int head_tiles = (heads + 15) / 16;
  const int col_blocks = (B * heads + 63) / 64;  // upper bound per group; blocks past a group's columns leave at once
  int prefix_splits = shared_prefix_splits(B, heads, Hkv, max_kv_splits);
  if (env_sp > 0 && !plan) prefix_splits = std::min(env_sp, max_kv_splits);  // (a plan was built for its own count)
  SharedWs ws;
  // remainders may span blocks in the plain planned form only (the forms that write the new K / V rows in a tail block
  // keep one block per remainder)
  static const int env_tmax = getenv("SGLK_TAIL_TMAX") ? atoi(getenv("SGLK_TAIL_TMAX")) : 0;
  ws.tail_tmax = (plan && !qp && !rope_src) ? shared_tail_tmax(max_kv_splits) : 1;
  if (env_tmax > 0) ws.tail_tmax = std::min(ws.tail_tmax, env_tmax);
  // tail_blocks() gives no remainder more blocks than the chip has room for beside everybody's first: a launch whose
  // first blocks fill it takes the kernel compiled without the multi-block path
  ws.tail_tmax = std::max(1, std::min(ws.tail_tmax, kTailBlocksToFill / std::max(1, B * Hkv * ((Hq / Hkv + 15) / 16))));
  const bool multi_tail = ws.tail_tmax > 1;
  ws.slots = prefix_splits + kWavesPerBlock * ws.tail_tmax;
  const int64_t rows = (int64_t)B * Hq * ws.slots;
  SGLK_CHECK(rows * 512 < ((int64_t)1 << 31), "decode_attention_shared_prefix: B * Hq * max_kv_splits too large");
  SGLK_CHECK(heads <= kTailDone / 64, "decode_attention_shared_prefix: too many query heads per kv head");
  ws.rows = (int)rows;
  ws.counters = reinterpret_cast<int*>(workspace);
  ws.error = ws.counters + (int64_t)B * Hq;
  ws.tickets = ws.error + 1;
  ws.part_o = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + shared_counter_bytes(B, Hq));
  ws.part_lse = ws.part_o + rows * 128;
  // planned: one round of blocks for a batch that is one group; more items (many small groups) take further rounds
  const int n_prefix = (plan ? 1 : G) * prefix_splits * col_blocks * Hkv;
  const int n_tail = B * Hkv * head_tiles * ws.tail_tmax;
  PlanView pv{};
  if (plan) {
    SGLK_CHECK(B <= kPlanMaxB, "decode_attention_shared_prefix: a plan covers at most %d requests", kPlanMaxB);
    pv = plan_view(const_cast<int32_t*>(plan), B, plan_max_items(B, heads, prefix_splits));
    SGLK_CHECK(aligned16(plan), "decode_attention_shared_prefix: plan must be 16-byte aligned");
  }
  // the prefix role's two stages; the tail role's four V tiles fit in the same bytes (fp8 pools: four K tiles and the
  // code table behind them)
  const size_t lds = fast8 ? 2 * kBlockAttnLds + kKv8LutBytes : kBlockAttnLds;
  static_assert(kWavesPerBlock * kTileBytes <= kBlockAttnLds, "LDS of the prefix role must cover the tail role");
  RopeRole rr{};
  if (rope_src) {
    // rope role: four blocks per token (a block of 256 threads sums, rotates and writes a quarter of the row's heads)
    SGLK_CHECK(plan && !qp && rope_q_out && rope_seq > 0, "decode_attention_shared_planned_rope: bad arguments");
    rr.parts = std::min(4, Hq + 2 * Hkv);
    rr.tokens = B;
    rr.blocks = (B * rr.parts + 7) / 8 * 8;
    rr.target = rr.parts * rope_seq;
    rr.ready = pv.lcp;
    rr.q_out = reinterpret_cast<uint16_t*>(rope_q_out);
    SGLK_CHECK((int64_t)cdiv_i(Hq + 2 * Hkv, rr.parts) * 128 * 2 <= (int64_t)kBlockAttnLds,
               "decode_attention_shared_planned_rope: too many heads per rope block");
  }
#define SGLK_SHARED_LAUNCH(ROPE, PLAN, QP)                                                                             \
  SGLK_DISPATCH_HALF(dtype, T,                                                                                         \
                     hipLaunchKernelGGL((decode_shared_fused_kernel<T, ROPE, PLAN>), dim3(n_prefix + n_tail), dim3(256), \
                                        lds, s, q, QP, k_buffer, v_buffer, kv_indptr, kv_indices, group_indptr,          \
                                        prefix_lens, G, pv.hdr, pv.tail, pv.items, ws, prefix_splits, n_prefix,          \
                                        col_blocks, B, Hq, Hkv, head_tiles, q_stride_b, q_stride_h, k_stride_n,          \
                                        k_stride_h, v_stride_n, v_stride_h, sm_scale, logit_cap, o, o_stride_b,          \
                                        o_stride_h, xsum_out, xs_stride, (gated ? 1 : 0) | (env_delay << 8), rr))
  dtype = qdtype;
#define SGLK_SHARED_LAUNCH_MT(KVT, PAIRV, MTV)                                                                          \
  hipLaunchKernelGGL((decode_shared_fused_kernel<T, false, true, false, KVT, PAIRV, MTV>), dim3(n_prefix + n_tail),     \
                     dim3(256), lds, s, q, QkvPartials{}, k_buffer, v_buffer, kv_indptr, kv_indices, group_indptr,       \
                     prefix_lens, G, pv.hdr, pv.tail, pv.items, ws, prefix_splits, n_prefix, col_blocks, B, Hq, Hkv,     \
                     head_tiles, q_stride_b, q_stride_h, k_stride_n, k_stride_h, v_stride_n, v_stride_h, sm_scale,       \
                     logit_cap, o, o_stride_b, o_stride_h, xsum_out, xs_stride, (gated ? 1 : 0) | (env_delay << 8), rr)
#define SGLK_SHARED_LAUNCH_KV(KVT, PAIRV) \
  do {                                    \
    if (multi_tail)                       \
      SGLK_SHARED_LAUNCH_MT(KVT, PAIRV, true);  \
    else                                  \
      SGLK_SHARED_LAUNCH_MT(KVT, PAIRV, false); \
  } while (0)
  if (fast8) {
    static bool lds_set[64][4][2];  // per device: the attribute belongs to the device's copy of the code object
    int dev = 0;
    SGLK_HIP(hipGetDevice(&dev));
    SGLK_DISPATCH_HALF(dtype, T, {
      SGLK_DISPATCH_KV(kvf, T, KV, {
        bool& done = lds_set[dev & 63][kvf & 3][dtype == SGLK_F16 ? 1 : 0];
        if (!done) {  // 64.5 KiB of dynamic LDS: above the default limit
          SGLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_shared_fused_kernel<T, false, true, false, KV, false, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          SGLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_shared_fused_kernel<T, false, true, false, KV, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          done = true;
        }
        SGLK_SHARED_LAUNCH_KV(KV, false);
      });
    });
  } else if (pair64) {
    SGLK_DISPATCH_HALF(dtype, T, SGLK_SHARED_LAUNCH_KV(T, true));
  } else if (rope_src) {
    SGLK_DISPATCH_HALF(dtype, T,
                       hipLaunchKernelGGL((decode_shared_fused_kernel<T, false, true, true>),
                                          dim3(rr.blocks + n_prefix + n_tail), dim3(256), lds, s, q, *rope_src, k_buffer,
                                          v_buffer, kv_indptr, kv_indices, group_indptr, prefix_lens, G, pv.hdr, pv.tail,
                                          pv.items, ws, prefix_splits, n_prefix, col_blocks, B, Hq, Hkv, head_tiles,
                                          q_stride_b, q_stride_h, k_stride_n, k_stride_h, v_stride_n, v_stride_h, sm_scale,
                                          logit_cap, o, o_stride_b, o_stride_h, xsum_out, xs_stride,
                                          (gated ? 1 : 0) | (env_delay << 8), rr));
  } else if (qp) {
    // a pair's single tail block writes the pair's new rows: more than one head tile would need them across blocks
    if (head_tiles != 1)
      return fail(SGLK_EUNSUP, "decode_attention_shared_prefix_rope: at most 16 query heads per kv head (got %d)", heads);
    if (plan) {
      SGLK_SHARED_LAUNCH(true, true, *qp);
    } else {
      SGLK_SHARED_LAUNCH(true, false, *qp);
    }
  } else if (plan) {
    SGLK_DISPATCH_HALF(dtype, T, SGLK_SHARED_LAUNCH_KV(T, false));
  } else {
    SGLK_SHARED_LAUNCH(false, false, QkvPartials{});
  }
#undef SGLK_SHARED_LAUNCH
#undef SGLK_SHARED_LAUNCH_KV
#undef SGLK_SHARED_LAUNCH_MT
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_decode_attention_shared_prefix_xs(const void* q, const void* k_buffer, const void* v_buffer, void* o,
                                           const int32_t* kv_indptr, const int32_t* kv_indices,
                                           const int32_t* group_indptr, const int32_t* prefix_lens, int G,
                                           float* workspace, int max_kv_splits, int B, int Hq, int Hkv, int D, int Dv,
                                           int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h,
                                           int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h,
                                           float sm_scale, float logit_cap, int dtype, float* xsum_out, int64_t xs_stride,
                                           void* stream) {
  SGLK_CHECK(q, "decode_attention_shared_prefix: null pointer");
  return launch_shared_prefix(q, nullptr, const_cast<void*>(k_buffer), const_cast<void*>(v_buffer), o, kv_indptr,
                              kv_indices, group_indptr, prefix_lens, G, nullptr, workspace, max_kv_splits, B, Hq, Hkv, D, Dv,
                              q_stride_b, q_stride_h, k_stride_n, k_stride_h, v_stride_n, v_stride_h, o_stride_b, o_stride_h,
                              sm_scale, logit_cap, dtype, xsum_out, xs_stride, stream);
}

int sglk_decode_attention_shared_prefix_rope(const float* qkv_partials, int splits, int64_t part_stride,
                                             const void* qkv_bias, const float* cos_sin_cache, const int64_t* positions,
                                             int rot_dim, int interleave, const void* loc, int loc_is64, void* k_buffer,
                                             void* v_buffer, void* o, const int32_t* kv_indptr, const int32_t* kv_indices,
                                             const int32_t* group_indptr, const int32_t* prefix_lens, int G,
                                             float* workspace, int max_kv_splits, int B, int Hq, int Hkv, int D, int Dv,
                                             int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
                                             int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap,
                                             int dtype, float* xsum_out, int64_t xs_stride, void* stream) {
  SGLK_CHECK(qkv_partials && cos_sin_cache && positions && loc, "decode_attention_shared_prefix_rope: null pointer");
  if (rot_dim != 128 || interleave != 0 || D != 128)
    return fail(SGLK_EUNSUP, "decode_attention_shared_prefix_rope: full rotation of 128-element heads in the (d, d + 64) "
                "pairing only (got rot_dim=%d interleave=%d D=%d)", rot_dim, interleave, D);
  const int64_t row_len = (int64_t)(Hq + 2 * Hkv) * 128;
  SGLK_CHECK(splits >= 1 && part_stride >= (int64_t)B * row_len && part_stride % 4 == 0 && aligned16(qkv_partials) &&
                 aligned16(cos_sin_cache) && (!qkv_bias || aligned16(qkv_bias)),
             "decode_attention_shared_prefix_rope: bad partial layout");
  QkvPartials qp;
  qp.partials = qkv_partials;
  qp.splits = splits;
  qp.part_stride = part_stride;
  qp.row_len = (int)row_len;
  qp.bias = reinterpret_cast<const uint16_t*>(qkv_bias);
  qp.cos_sin = cos_sin_cache;
  qp.positions = positions;
  qp.loc = loc;
  qp.loc_is64 = loc_is64;
  return launch_shared_prefix(nullptr, &qp, k_buffer, v_buffer, o, kv_indptr, kv_indices, group_indptr, prefix_lens, G,
                              nullptr, workspace, max_kv_splits, B, Hq, Hkv, D, Dv, 0, 0, k_stride_n, k_stride_h, v_stride_n,
                              v_stride_h, o_stride_b, o_stride_h, sm_scale, logit_cap, dtype, xsum_out, xs_stride, stream);
}

int sglk_decode_attention_shared_prefix(const void* q, const void* k_buffer, const void* v_buffer, void* o,
                                        const int32_t* kv_indptr, const int32_t* kv_indices,
                                        const int32_t* group_indptr, const int32_t* prefix_lens, int G, float* workspace,
                                        int max_kv_splits, int B, int Hq, int Hkv, int D, int Dv, int64_t q_stride_b,
                                        int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n,
                                        int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale,
                                        float logit_cap, int dtype, void* stream) {
  return sglk_decode_attention_shared_prefix_xs(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, group_indptr, prefix_lens,
                                                G, workspace, max_kv_splits, B, Hq, Hkv, D, Dv, q_stride_b, q_stride_h,
                                                k_stride_n, k_stride_h, v_stride_n, v_stride_h, o_stride_b, o_stride_h,
                                                sm_scale, logit_cap, dtype, nullptr, 0, stream);
}

// ---- the per-step plan of the shared-prefix kernel -------------------------------------------------------------
int64_t sglk_decode_shared_plan_bytes(int B, int Hq, int Hkv, int max_kv_splits) {
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0 || max_kv_splits <= 0) return 0;
  const int heads = Hq / Hkv;
  const PlanView pv = plan_view(nullptr, B, plan_max_items(B, heads, shared_prefix_splits(B, heads, Hkv, max_kv_splits)));
  return ((pv.words * 4 + 255) / 256) * 256;
}

static int plan_assemble(const int32_t* kv_indptr, const int32_t* kv_indices, int B, int Hq, int Hkv, int max_kv_splits,
                         int min_share, int single_ok, int32_t* plan, hipStream_t s) {
  const int heads = Hq / Hkv;
  const int splits = shared_prefix_splits(B, heads, Hkv, max_kv_splits);
  hipLaunchKernelGGL(shared_plan_assemble_kernel, dim3(1), dim3(1024), 0, s, kv_indptr, kv_indices, B, heads, splits,
                     plan_max_items(B, heads, splits), min_share, single_ok, plan);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_decode_shared_plan(const int32_t* req_to_token, int64_t row_stride, const void* req_pool_indices, int idx_is64,
                            const void* seq_lens, int lens_is64, const int32_t* kv_indptr, const int32_t* kv_indices, int B,
                            int Hq, int Hkv, int max_kv_splits, int min_share, int32_t* plan, void* stream) {
  SGLK_CHECK(req_to_token && req_pool_indices && seq_lens && kv_indptr && kv_indices && plan,
             "decode_shared_plan: null pointer");
  SGLK_CHECK(B > 0 && B <= kPlanMaxB && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && max_kv_splits > 0 &&
                 max_kv_splits <= kSharedMaxSplits,
             "decode_shared_plan: bad shape (1 <= B <= %d, max_kv_splits in 1..16)", kPlanMaxB);
  SGLK_CHECK(aligned16(plan), "decode_shared_plan: plan must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int heads = Hq / Hkv;
  const PlanView pv = plan_view(plan, B, plan_max_items(B, heads, shared_prefix_splits(B, heads, Hkv, max_kv_splits)));
  hipLaunchKernelGGL(shared_plan_lcp_kernel, dim3(B), dim3(256), 0, s, req_to_token, row_stride, req_pool_indices, idx_is64,
                     seq_lens, lens_is64, B, pv.leader, pv.lcp);
  SGLK_LAUNCH_CHECK();
  return plan_assemble(kv_indptr, kv_indices, B, Hq, Hkv, max_kv_splits, min_share > 0 ? min_share : kMinBlockKV, 0, plan, s);
}

int sglk_decode_shared_plan_from_groups(const int32_t* kv_indptr, const int32_t* kv_indices, const int32_t* group_indptr,
                                        const int32_t* prefix_lens, int G, int B, int Hq, int Hkv, int max_kv_splits,
                                        int32_t* plan, void* stream) {
  SGLK_CHECK(kv_indptr && kv_indices && group_indptr && prefix_lens && plan, "decode_shared_plan_from_groups: null pointer");
  SGLK_CHECK(B > 0 && B <= kPlanMaxB && G > 0 && G <= B && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && max_kv_splits > 0 &&
                 max_kv_splits <= kSharedMaxSplits,
             "decode_shared_plan_from_groups: bad shape (1 <= G <= B <= %d, max_kv_splits in 1..16)", kPlanMaxB);
  SGLK_CHECK(aligned16(plan), "decode_shared_plan_from_groups: plan must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int heads = Hq / Hkv;
  const PlanView pv = plan_view(plan, B, plan_max_items(B, heads, shared_prefix_splits(B, heads, Hkv, max_kv_splits)));
  hipLaunchKernelGGL(shared_plan_from_groups_kernel, dim3((B + 255) / 256), dim3(256), 0, s, group_indptr, prefix_lens, G, B,
                     pv.leader, pv.lcp);
  SGLK_LAUNCH_CHECK();
  return plan_assemble(kv_indptr, kv_indices, B, Hq, Hkv, max_kv_splits, 1, 1, plan, s);
}

int sglk_decode_metadata_shared(const int32_t* req_to_token, int64_t row_stride, const void* req_pool_indices, int idx_is64,
                                const void* seq_lens, int lens_is64, int32_t* kv_indptr, int32_t* kv_indices,
                                int32_t* num_kv_splits, int bs, int num_head, int num_kv_head, int max_kv_splits,
                                int device_core_count, int min_share, int32_t* plan, void* stream) {
  if (int rc = sglk_decode_metadata(req_to_token, row_stride, req_pool_indices, idx_is64, seq_lens, lens_is64, kv_indptr,
                                    kv_indices, num_kv_splits, bs, num_head, num_kv_head, max_kv_splits, device_core_count,
                                    stream))
    return rc;
  if (bs == 0) return SGLK_OK;
  return sglk_decode_shared_plan(req_to_token, row_stride, req_pool_indices, idx_is64, seq_lens, lens_is64, kv_indptr,
                                 kv_indices, bs, num_head, num_kv_head, max_kv_splits, min_share, plan, stream);
}

int sglk_decode_attention_shared_planned(const void* q, const void* k_buffer, const void* v_buffer, void* o,
                                         const int32_t* kv_indptr, const int32_t* kv_indices, const int32_t* plan,
                                         float* workspace, int max_kv_splits, int B, int Hq, int Hkv, int D, int Dv,
                                         int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h,
                                         int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h,
                                         float sm_scale, float logit_cap, int dtype, float* xsum_out, int64_t xs_stride,
                                         const int32_t* num_kv_splits, void* stream) {
  SGLK_CHECK(q && plan, "decode_attention_shared_planned: null pointer");
  // The one-launch kernel serves every batch by itself: requests that share nothing walk their whole context as a
  // "remainder", over as many tail blocks as the chip has room for (measured against the per-request kernels on
  // batches in which nobody shares: between 6 % faster and 9 % slower; two more launches per layer that leave at once
  // for every batch that does share would cost 4 us each time).  Only the absorbed-MLA form, whose plan-driven kernels
  // are slower than the per-request ones, keeps the pair: with num_kv_splits both are launched and the plan's item
  // count -- a device word both read -- decides which of the two leaves at once, so that a captured launch sequence
  // serves either kind of batch.
  bool gated = false;
  if (num_kv_splits) {
    const int kvf = dtype >> SGLK_KV_SHIFT;
    const int qd = dtype & ((1 << SGLK_KV_SHIFT) - 1);
    const bool mla_shape = (qd == SGLK_F16 || qd == SGLK_BF16) && kvf == 0 && D == kMlaD && Dv == kMlaDv && Hkv == 1 &&
                           v_buffer == k_buffer && v_stride_n == k_stride_n && !xsum_out;
    if (mla_shape) {
      SGLK_CHECK(workspace, "decode_attention_shared_planned: null pointer");
      float* logits = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + shared_counter_bytes(B, Hq));
      float* lse = logits + (int64_t)B * Hq * max_kv_splits * Dv;
      if (int rc = decode_attention_impl(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, logits, lse, num_kv_splits,
                                         max_kv_splits, B, Hq, Hkv, D, Dv, q_stride_b, q_stride_h, k_stride_n, k_stride_h,
                                         v_stride_n, v_stride_h, o_stride_b, o_stride_h, sm_scale, logit_cap, dtype, xsum_out,
                                         xs_stride, stream, plan + kPlanNItems))
        return rc;
      gated = true;
    }
  }
  return launch_shared_prefix(q, nullptr, const_cast<void*>(k_buffer), const_cast<void*>(v_buffer), o, kv_indptr,
                              kv_indices, nullptr, nullptr, 0, plan, workspace, max_kv_splits, B, Hq, Hkv, D, Dv, q_stride_b,
                              q_stride_h, k_stride_n, k_stride_h, v_stride_n, v_stride_h, o_stride_b, o_stride_h, sm_scale,
                              logit_cap, dtype, xsum_out, xs_stride, stream, gated);
}

int sglk_decode_attention_shared_planned_rope(const float* qkv_partials, int splits, int64_t part_stride,
                                              const void* qkv_bias, const float* cos_sin_cache, const int64_t* positions,
                                              int rot_dim, int interleave, const void* loc, int loc_is64, void* q_out,
                                              int64_t q_stride_b, void* k_buffer, void* v_buffer, void* o,
                                              const int32_t* kv_indptr, const int32_t* kv_indices, int32_t* plan,
                                              int launch_in_step, float* workspace, int max_kv_splits, int B, int Hq, int Hkv,
                                              int D, int Dv, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n,
                                              int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale,
                                              float logit_cap, int dtype, float* xsum_out, int64_t xs_stride, void* stream) {
  SGLK_CHECK(qkv_partials && cos_sin_cache && positions && loc && q_out && plan,
             "decode_attention_shared_planned_rope: null pointer");
  if (rot_dim != 128 || interleave != 0 || D != 128 || Dv != 128 || (dtype != SGLK_F16 && dtype != SGLK_BF16))
    return fail(SGLK_EUNSUP, "decode_attention_shared_planned_rope: 16-bit types, full rotation of 128-element heads in the "
                "(d, d + 64) pairing only (got rot_dim=%d interleave=%d D=%d Dv=%d)", rot_dim, interleave, D, Dv);
  const int64_t row_len = (int64_t)(Hq + 2 * Hkv) * 128;
  SGLK_CHECK(splits >= 1 && part_stride >= (int64_t)B * row_len && part_stride % 4 == 0 && aligned16(qkv_partials) &&
                 aligned16(cos_sin_cache) && (!qkv_bias || aligned16(qkv_bias)) && aligned16(q_out) && q_stride_b % 8 == 0 &&
                 q_stride_b >= (int64_t)Hq * 128,
             "decode_attention_shared_planned_rope: bad partial or query layout");
  SGLK_CHECK(launch_in_step >= 1, "decode_attention_shared_planned_rope: launches of a step are numbered from 1");
  QkvPartials qp;
  qp.partials = qkv_partials;
  qp.splits = splits;
  qp.part_stride = part_stride;
  qp.row_len = (int)row_len;
  qp.bias = reinterpret_cast<const uint16_t*>(qkv_bias);
  qp.cos_sin = cos_sin_cache;
  qp.positions = positions;
  qp.loc = loc;
  qp.loc_is64 = loc_is64;
  return launch_shared_prefix(q_out, nullptr, k_buffer, v_buffer, o, kv_indptr, kv_indices, nullptr, nullptr, 0, plan,
                              workspace, max_kv_splits, B, Hq, Hkv, D, Dv, q_stride_b, 128, k_stride_n, k_stride_h,
                              v_stride_n, v_stride_h, o_stride_b, o_stride_h, sm_scale, logit_cap, dtype, xsum_out, xs_stride,
                              stream, false, &qp, q_out, launch_in_step);
}
}
