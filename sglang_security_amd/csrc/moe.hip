// Mixture-of-experts routing helpers of the reference's ROCm op surface: token -> expert block alignment
// and softmax top-k gating.  Integer work and a row-wise reduction: LDS counters and wave ballots, no
// matrix cores.
#include "common.h"

namespace sglk {

static constexpr int kMoeMaxExperts = 512;
static constexpr int kMoeThreads = 1024;
static constexpr int kMoeWaves = kMoeThreads / 64;

// One block.  counts -> padded prefix sums -> expert id of every block -> stable placement: tokens of one
// expert keep their order of appearance (one of the orders the reference's atomic placement can produce,
// moe_align_kernel.cu:27-41, made deterministic).  Ids outside [0, E) are skipped.
template <typename I>
__global__ void __launch_bounds__(kMoeThreads) moe_align_kernel(const I* __restrict__ topk_ids, int64_t numel, int E,
                                                                int block_size, int32_t* __restrict__ sorted_token_ids,
                                                                int32_t* __restrict__ expert_ids,
                                                                int32_t* __restrict__ num_tokens_post_pad,
                                                                int32_t* __restrict__ cumsum_out) {
  extern __shared__ int32_t sm[];
  int32_t* counts = sm;                 // [E]
  int32_t* cumsum = counts + E;         // [E + 1]
  int32_t* base = cumsum + E + 1;       // [E] tokens of the expert placed by earlier chunks
  int32_t* wave_hist = base + E;        // [kMoeWaves][E]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int e = tid; e < E; e += kMoeThreads) {
    counts[e] = 0;
    base[e] = 0;
  }
  __syncthreads();
  for (int64_t i = tid; i < numel; i += kMoeThreads) {
    const int64_t e = (int64_t)topk_ids[i];
    if (e >= 0 && e < E) atomicAdd(&counts[e], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int32_t run = 0;
    cumsum[0] = 0;
    for (int e = 0; e < E; ++e) {
      run += (counts[e] + block_size - 1) / block_size * block_size;
      cumsum[e + 1] = run;
    }
    *num_tokens_post_pad = run;
  }
  __syncthreads();
  for (int e = tid; e < E; e += kMoeThreads) {
    for (int32_t i = cumsum[e]; i < cumsum[e + 1]; i += block_size) expert_ids[i / block_size] = e;
    if (cumsum_out) cumsum_out[e] = cumsum[e];
  }
  if (tid == 0 && cumsum_out) cumsum_out[E] = cumsum[E];

  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int64_t c0 = 0; c0 < numel; c0 += kMoeThreads) {
    for (int j = tid; j < kMoeWaves * E; j += kMoeThreads) wave_hist[j] = 0;
    __syncthreads();
    const int64_t i = c0 + tid;
    int64_t e64 = i < numel ? (int64_t)topk_ids[i] : -1;
    const bool valid = e64 >= 0 && e64 < E;
    const int e = valid ? (int)e64 : 0;
    // lanes of this wave that hold the same expert
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 9; ++bit) {
      const unsigned long long b = __ballot((e >> bit) & 1);
      same &= ((e >> bit) & 1) ? b : ~b;
    }
    if (!valid) same = 0;
    const int rank = __popcll(same & below);
    if (valid && rank == 0) wave_hist[w * E + e] = __popcll(same);
    __syncthreads();
    if (valid) {
      int pre = 0;
      for (int ww = 0; ww < w; ++ww) pre += wave_hist[ww * E + e];
      sorted_token_ids[cumsum[e] + base[e] + pre + rank] = (int32_t)i;
    }
    __syncthreads();
    for (int ee = tid; ee < E; ee += kMoeThreads) {
      int add = 0;
      for (int ww = 0; ww < kMoeWaves; ++ww) add += wave_hist[ww * E + ee];
      base[ee] += add;
    }
    __syncthreads();
  }
}

// One wave per token row: softmax over the experts, then k rounds of arg-max (ties: lowest expert id).
// token_expert_indices[row, j] = j * rows + row (moe_topk_softmax_kernels.cu:150-157).
static constexpr int kTopkMaxExperts = 1024;
template <typename T>
__global__ void __launch_bounds__(256) topk_softmax_kernel(const void* __restrict__ gating, float* __restrict__ weights,
                                                           int32_t* __restrict__ ids, int32_t* __restrict__ source_rows,
                                                           int rows, int E, int k, int64_t gating_stride) {
  extern __shared__ float rowbuf[];  // [4 waves][E]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= rows) return;  // no block-level synchronisation below
  float* v = rowbuf + w * E;
  float mx = -INFINITY;
  for (int e = lane; e < E; e += 64) {
    const float x = ld<T>(gating, (int64_t)row * gating_stride + e);
    v[e] = x;
    mx = fmaxf(mx, x);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int e = lane; e < E; e += 64) sum += expf(v[e] - mx);
  sum = wave_sum(sum);
  for (int j = 0; j < k; ++j) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int e = lane; e < E; e += 64) {
      const float x = v[e];
      if (x > best) {  // ascending e per lane: the first maximum stays
        best = x;
        bi = e;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    if (bi >= E) bi = 0;  // a row of NaNs / fewer experts than k: keep the index in range
    if (lane == 0) {
      weights[(int64_t)row * k + j] = expf(best - mx) / sum;
      ids[(int64_t)row * k + j] = bi;
      source_rows[(int64_t)row * k + j] = j * rows + row;
    }
    if (lane == (bi & 63)) v[bi] = -INFINITY;  // the owner lane retires the winner; same-wave LDS stays in order
  }
}

}  // namespace sglk

using namespace sglk;

extern "C" {

int sglk_moe_align_block_size(const void* topk_ids, int ids_is64, int64_t numel, int num_experts, int block_size,
                              int32_t* sorted_token_ids, int32_t* expert_ids, int32_t* num_tokens_post_pad,
                              int32_t* cumsum_buffer, void* stream) {
  SGLK_CHECK((topk_ids || numel == 0) && sorted_token_ids && expert_ids && num_tokens_post_pad,
             "moe_align_block_size: null pointer");
  SGLK_CHECK(numel >= 0 && numel < (1ll << 31), "moe_align_block_size: bad element count");
  SGLK_CHECK(num_experts > 0 && num_experts <= kMoeMaxExperts, "moe_align_block_size: num_experts=%d not in 1..%d",
             num_experts, kMoeMaxExperts);
  SGLK_CHECK(block_size > 0, "moe_align_block_size: block_size must be positive");
  const size_t lds = (size_t)((3 + kMoeWaves) * num_experts + 1) * sizeof(int32_t);
  hipStream_t s = (hipStream_t)stream;
  if (ids_is64)
    hipLaunchKernelGGL((moe_align_kernel<int64_t>), dim3(1), dim3(kMoeThreads), lds, s, (const int64_t*)topk_ids, numel,
                       num_experts, block_size, sorted_token_ids, expert_ids, num_tokens_post_pad, cumsum_buffer);
  else
    hipLaunchKernelGGL((moe_align_kernel<int32_t>), dim3(1), dim3(kMoeThreads), lds, s, (const int32_t*)topk_ids, numel,
                       num_experts, block_size, sorted_token_ids, expert_ids, num_tokens_post_pad, cumsum_buffer);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_topk_softmax(const void* gating_output, float* topk_weights, int32_t* topk_ids, int32_t* token_expert_indices,
                      int num_tokens, int num_experts, int topk, int64_t gating_stride, int dtype, void* stream) {
  SGLK_CHECK(num_tokens >= 0 && num_experts > 0 && num_experts <= kTopkMaxExperts, "topk_softmax: bad shape");
  SGLK_CHECK(num_tokens == 0 || (gating_output && topk_weights && topk_ids && token_expert_indices),
             "topk_softmax: null pointer");
  SGLK_CHECK(topk > 0 && topk <= num_experts, "topk_softmax: topk=%d not in 1..%d", topk, num_experts);
  SGLK_CHECK(gating_stride >= num_experts, "topk_softmax: row stride smaller than the row");
  if (num_tokens == 0) return SGLK_OK;
  const size_t lds = (size_t)4 * num_experts * sizeof(float);
  SGLK_DISPATCH_FLOAT(dtype, T,
                      hipLaunchKernelGGL((topk_softmax_kernel<T>), dim3((num_tokens + 3) / 4), dim3(256), lds,
                                         (hipStream_t)stream, gating_output, topk_weights, topk_ids, token_expert_indices,
                                         num_tokens, num_experts, topk, gating_stride));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}
}
