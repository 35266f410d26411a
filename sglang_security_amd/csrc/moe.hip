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

// Grouped, bias-corrected top-k gate of DeepSeek-V3 style routers (sgl-kernel/csrc/moe/moe_fused_gate.cu:52-245 ==
// biased_grouped_topk_impl, python/sglang/srt/layers/moe/topk.py:155-215): scores = sigmoid(logits); choice scores =
// scores + bias; a group's score is the sum of its two best choice scores; all but the topk_group best groups are
// dropped (lowest first, ties: the higher group goes first); the topk best choice scores of the groups kept win
// (ties: the lower expert id); weights are the plain scores, normalised by their sum.  With n_share > 0 the last of
// the topk slots names a shared expert (E + row % n_share) weighing sum / routed_scaling_factor before the
// normalisation.  Sigmoid, bias add and group sum are rounded to the input type, as the reference computes in it.
// One wave per row; the row lives in LDS.
static constexpr int kGateMaxExperts = 1024;
template <typename T>
__global__ void __launch_bounds__(256) moe_fused_gate_kernel(const void* __restrict__ logits, const void* __restrict__ bias,
                                                             float* __restrict__ out, int32_t* __restrict__ idx, int rows,
                                                             int E, int ng, int topk_group, int topk, int n_share,
                                                             float routed_scaling, int64_t in_stride) {
  extern __shared__ float gatebuf[];  // per wave: sig[E], sc[E], wout[64], alive[64]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= rows) return;  // no block-level synchronisation below
  float* sig = gatebuf + (size_t)w * (2 * E + 128);
  float* sc = sig + E;
  float* wout = sc + E;
  int* alive = reinterpret_cast<int*>(wout + 64);
  auto rnd = [](float v) { return Conv<T>::to_f32(Conv<T>::from_f32(v)); };
  for (int e = lane; e < E; e += 64) {
    const float x = ld<T>(logits, (int64_t)row * in_stride + e);
    const float s = rnd(1.0f / (1.0f + expf(-x)));
    sig[e] = s;
    sc[e] = rnd(s + ld<T>(bias, e));
  }
  // group scores: lane g scans group g
  const int gs = E / ng;
  float gsum = INFINITY;  // lanes without a group never win the arg-min
  if (lane < ng) {
    float m1 = -INFINITY, m2 = -INFINITY;
    for (int e = lane * gs; e < (lane + 1) * gs; ++e) {
      const float v = sc[e];
      if (v > m1) {
        m2 = m1;
        m1 = v;
      } else if (v > m2) {
        m2 = v;
      }
    }
    gsum = rnd(m1 + m2);
  }
  int keep = lane < ng ? 1 : 0;
  for (int r = 0; r < ng - topk_group; ++r) {
    float v = keep ? gsum : INFINITY;
    int gi = keep ? lane : -1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(gi, o, 64);
      if (ov < v || (ov == v && oi > gi)) {
        v = ov;
        gi = oi;
      }
    }
    if (lane == gi) keep = 0;
  }
  if (lane < ng) alive[lane] = keep;
  const int k_routed = topk - (n_share > 0 ? 1 : 0);
  float sum = 0.f;
  for (int j = 0; j < k_routed; ++j) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int e = lane; e < E; e += 64) {
      const float x = sc[e];
      if (alive[e / gs] && x > best) {  // ascending e per lane: the first maximum stays
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
    if (bi >= E) bi = 0;  // nothing left to choose from (NaN rows, topk beyond the kept groups): stay in range
    const float wgt = sig[bi];
    sum += wgt;
    if (lane == 0) {
      wout[j] = wgt;
      idx[(int64_t)row * topk + j] = bi;
    }
    if (lane == (bi & 63)) sc[bi] = -INFINITY;  // the owner lane retires the winner; same-wave LDS stays in order
  }
  if (n_share > 0 && lane == 0) {
    wout[k_routed] = sum / routed_scaling;
    idx[(int64_t)row * topk + k_routed] = E + row % n_share;
  }
  if (lane < topk) out[(int64_t)row * topk + lane] = wout[lane] / sum;
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

int sglk_moe_fused_gate(const void* input, const void* bias, float* output, int32_t* indices, int num_rows,
                        int num_experts, int num_expert_group, int topk_group, int topk, int n_share_experts_fusion,
                        float routed_scaling_factor, int64_t input_stride, int dtype, void* stream) {
  SGLK_CHECK(num_rows >= 0 && num_experts > 0 && num_experts <= kGateMaxExperts, "moe_fused_gate: num_experts=%d not in 1..%d",
             num_experts, kGateMaxExperts);
  SGLK_CHECK(num_expert_group > 0 && num_expert_group <= 64 && num_experts % num_expert_group == 0,
             "moe_fused_gate: num_expert_group=%d must divide num_experts and be at most 64", num_expert_group);
  SGLK_CHECK(topk_group > 0 && topk_group <= num_expert_group, "moe_fused_gate: bad topk_group");
  SGLK_CHECK(topk > 0 && topk <= 64 && n_share_experts_fusion >= 0, "moe_fused_gate: topk=%d not in 1..64", topk);
  SGLK_CHECK(topk - (n_share_experts_fusion > 0 ? 1 : 0) <= topk_group * (num_experts / num_expert_group) &&
                 topk - (n_share_experts_fusion > 0 ? 1 : 0) >= 1,
             "moe_fused_gate: topk exceeds the experts of the groups kept");
  SGLK_CHECK(n_share_experts_fusion == 0 || routed_scaling_factor != 0.f, "moe_fused_gate: routed_scaling_factor is 0");
  SGLK_CHECK(input_stride >= num_experts, "moe_fused_gate: row stride smaller than the row");
  if (num_rows == 0) return SGLK_OK;
  SGLK_CHECK(input && bias && output && indices, "moe_fused_gate: null pointer");
  const size_t lds = (size_t)4 * (2 * num_experts + 128) * sizeof(float);
  SGLK_DISPATCH_FLOAT(dtype, T,
                      hipLaunchKernelGGL((moe_fused_gate_kernel<T>), dim3((num_rows + 3) / 4), dim3(256), lds,
                                         (hipStream_t)stream, input, bias, output, indices, num_rows, num_experts,
                                         num_expert_group, topk_group, topk, n_share_experts_fusion,
                                         routed_scaling_factor, input_stride));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}
}
