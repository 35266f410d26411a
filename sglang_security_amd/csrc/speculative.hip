// Draft-tree helpers of EAGLE-style speculative decoding: tree layout from the selected draft tokens, greedy
// verification of a tree against the target model's picks, and target-only speculative sampling over a tree.
// Index work on a few dozen entries per request: one block per request, the request's rows held in LDS.
#include "common.h"

namespace sglk {
namespace {

constexpr int kMaxDraft = 1024;  // draft tokens per request (one thread each)

// Layout of one request's draft tree (sgl-kernel/csrc/speculative/eagle_utils.cu:32-114).  Token 0 is the verified
// root; token i > 0 is selected_index[i - 1], a flat index into the [depth, topk, topk] score list whose parent row
// (index / topk) names, through parent_list, the flat index of its parent.  Every thread resolves its own parent
// slot; children lists then fall out of two scans over the parent slots (first child = lowest token with that
// parent, next sibling = next higher token with the same parent), which is the order the reference's serial
// head insertions from the highest token down leave behind.
__global__ void __launch_bounds__(kMaxDraft) build_tree_kernel(const int64_t* __restrict__ parent_list,
                                                              const int64_t* __restrict__ selected_index,
                                                              const int32_t* __restrict__ verified_seq_len,
                                                              uint8_t* __restrict__ tree_mask,
                                                              int64_t* __restrict__ positions,
                                                              int64_t* __restrict__ retrive_index,
                                                              int64_t* __restrict__ retrive_next_token,
                                                              int64_t* __restrict__ retrive_next_sibling, int topk,
                                                              int depth, int n, int64_t mask_len) {
  __shared__ int sel[kMaxDraft];     // sel[i]: flat score index of token i + 1
  __shared__ int parent[kMaxDraft];  // parent[i]: parent token of token i; -1 for the root and for orphans
  __shared__ int64_t red[16];
  const int bid = blockIdx.x, tid = threadIdx.x;
  const int parent_pitch = topk * (depth - 1) + 1;

  // mask offset of this request: n * n per earlier request plus n * (their verified lengths)
  int64_t part = 0;
  for (int i = tid; i < bid; i += blockDim.x) part += verified_seq_len[i];
  part = wave_sum(part);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  if (tid < n - 1) sel[tid] = (int)selected_index[(int64_t)bid * (n - 1) + tid];
  __syncthreads();
  int64_t before = 0;
  for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) before += red[w];
  const int seq_len = verified_seq_len[bid];
  const int64_t base = (int64_t)n * n * bid + before * n;

  int par = -1;
  if (tid > 0 && tid < n) {
    const int row = sel[tid - 1] / topk;
    if (row == 0) {
      par = 0;
    } else if (row > 0 && row < parent_pitch) {
      const int want = (int)parent_list[(int64_t)bid * parent_pitch + row];
      for (int j = 0; j < n - 1; ++j)
        if (sel[j] == want) {
          par = j + 1;
          break;
        }
    }
  }
  if (tid < n) parent[tid] = par;
  __syncthreads();
  if (tid >= n || seq_len < 0 || base + (int64_t)(seq_len + n) * n > mask_len) return;  // mask too short: no writes

  // row `tid` of the mask: columns [seq_len + 1, seq_len + n) cleared, then the token's ancestors (itself included) set
  uint8_t* mrow = tree_mask + base + (int64_t)(seq_len + n) * tid + seq_len + 1;
  for (int i = 0; i < n - 1; ++i) mrow[i] = 0;
  int level = 0;
  for (int cur = tid; cur > 0 && level < n; cur = parent[cur]) {
    mrow[cur - 1] = 1;
    ++level;
  }
  positions[(int64_t)bid * n + tid] = seq_len + level;
  retrive_index[(int64_t)bid * n + tid] = (int64_t)bid * n + tid;

  // children lists; entries without a child / sibling keep the caller's -1
  int first_child = -1, sibling = -1;
  for (int j = n - 1; j > 0; --j) {
    if (parent[j] == tid) first_child = j;
    if (j > tid && tid > 0 && par >= 0 && parent[j] == par) sibling = j;
  }
  if (first_child >= 0) retrive_next_token[(int64_t)bid * n + tid] = first_child;
  if (sibling >= 0) retrive_next_sibling[(int64_t)bid * n + tid] = sibling;
}

// A link field as the walks use it: a slot of the request's row, or -1 (end of list; also for values outside the row,
// which a well-formed tree never holds).
__device__ __forceinline__ int link(int v, int n) { return (unsigned)v < (unsigned)n ? v : -1; }

// Walk of one request's tree along the target model's greedy picks (eagle_utils.cu:150-191): one thread per request.
__global__ void verify_tree_greedy_kernel(int32_t* __restrict__ predicts, int32_t* __restrict__ accept_index,
                                          int32_t* __restrict__ accept_token_num,
                                          const int32_t* __restrict__ candidates,
                                          const int32_t* __restrict__ retrive_index,
                                          const int32_t* __restrict__ retrive_next_token,
                                          const int32_t* __restrict__ retrive_next_sibling,
                                          const int32_t* __restrict__ target_predict, int bs, int num_spec, int n,
                                          int total) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= bs) return;
  const int64_t row = (int64_t)b * n;
  int last = retrive_index[row];
  accept_index[(int64_t)b * num_spec] = last;
  int accepted = 0, cur = 0;
  if ((unsigned)last >= (unsigned)total) {  // malformed index row: nothing can be addressed
    accept_token_num[b] = 0;
    return;
  }
  for (int j = 1; j < num_spec; ++j) {
    cur = link(retrive_next_token[row + cur], n);
    const int want = target_predict[last];
    for (int hops = 0; cur != -1 && candidates[row + cur] != want; ++hops)
      cur = hops < n ? link(retrive_next_sibling[row + cur], n) : -1;
    if (cur == -1) break;
    const int next = retrive_index[row + cur];
    if ((unsigned)next >= (unsigned)total) break;
    predicts[last] = want;
    ++accepted;
    last = next;
    accept_index[(int64_t)b * num_spec + accepted] = last;
  }
  accept_token_num[b] = accepted;
  predicts[last] = target_predict[last];
}

constexpr int kSpecThreads = 1024;

// Target-only speculative sampling over a tree (sgl-kernel/csrc/speculative/speculative_sampling.cuh:38-157): a
// candidate is taken when the request's coin is at most the accumulated target probability of the siblings tried so
// far (over threshold_acc), or when its own target probability reaches threshold_single; a rejected candidate's
// target probability is copied into draft_probs, and the token after the last accepted one is drawn by inverse CDF
// from relu(target - draft) of the row reached, with the coin of that row.
__global__ void __launch_bounds__(kSpecThreads) tree_spec_sampling_kernel(
    int32_t* __restrict__ predicts, int32_t* __restrict__ accept_index, int32_t* __restrict__ accept_token_num,
    const int32_t* __restrict__ candidates, const int32_t* __restrict__ retrive_index,
    const int32_t* __restrict__ retrive_next_token, const int32_t* __restrict__ retrive_next_sibling,
    const float* __restrict__ uniform_samples, const float* __restrict__ target_probs, float* draft_probs, int num_spec,
    int n, int d, int n_slots, float threshold_single, float threshold_acc) {
  __shared__ float scan[16];
  __shared__ int s_last, s_accepted, s_pick;
  __shared__ int64_t s_off;
  __shared__ float s_coin;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t row = (int64_t)b * n;
  if (tid == 0) {
    float acc = 0.f;
    int64_t off = row * d;
    float coin = uniform_samples[row];
    int last = retrive_index[row];
    accept_index[(int64_t)b * num_spec] = last;
    int accepted = 0, cur = 0;
    const bool ok = (unsigned)last < (unsigned)n_slots;
    for (int j = 1; ok && j < num_spec; ++j) {
      cur = link(retrive_next_token[row + cur], n);
      for (int hops = 0; cur != -1; ++hops) {
        const int tok = candidates[row + cur];
        const int next = retrive_index[row + cur];
        if ((unsigned)tok >= (unsigned)d || (unsigned)next >= (unsigned)n_slots || hops > n) {  // malformed tree
          cur = -1;
          break;
        }
        const float pt = target_probs[off + tok];
        acc += pt;
        if (coin <= acc / threshold_acc || pt >= threshold_single) {
          acc = 0.f;
          off = (row + cur) * d;
          coin = uniform_samples[row + cur];
          predicts[last] = tok;
          ++accepted;
          last = next;
          accept_index[(int64_t)b * num_spec + accepted] = last;
          break;
        }
        draft_probs[off + tok] = pt;
        cur = link(retrive_next_sibling[row + cur], n);
      }
      if (cur == -1) break;
    }
    accept_token_num[b] = accepted;
    s_last = ok ? last : -1;
    s_accepted = accepted;
    s_off = off;
    s_coin = coin;
    s_pick = d - 1;  // nothing left to draw from: the last token id
    __threadfence_block();
  }
  __syncthreads();
  const float* q = target_probs + s_off;
  const float* p = draft_probs + s_off;
  const bool bonus = s_accepted == num_spec - 1;  // the whole path was accepted: no draft distribution to subtract

  // thread t owns the contiguous chunk [t * per, (t + 1) * per): chunk sums, exclusive scan, then the owner of the
  // target walks its chunk
  const int per = (d + kSpecThreads - 1) / kSpecThreads;
  const int lo = min(tid * per, d), hi = min(lo + per, d);
  float mine = 0.f;
  for (int i = lo; i < hi; ++i) mine += fmaxf(q[i] - (bonus ? 0.f : p[i]), 0.f);
  float incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float v = __shfl_up(incl, o, 64);
    if ((tid & 63) >= o) incl += v;
  }
  if ((tid & 63) == 63) scan[tid >> 6] = incl;
  __syncthreads();
  float wave_base = 0.f, total = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    if (w < (tid >> 6)) wave_base += scan[w];
    total += scan[w];
  }
  const float start = wave_base + incl - mine;
  const float u = s_coin * total;
  if (mine > 0.f && start <= u && u < start + mine) {
    float run = start;
    int pick = -1, lastpos = -1;
    for (int i = lo; i < hi; ++i) {
      const float v = fmaxf(q[i] - (bonus ? 0.f : p[i]), 0.f);
      if (v > 0.f) {
        lastpos = i;
        run += v;
        if (run > u) {
          pick = i;
          break;
        }
      }
    }
    s_pick = pick >= 0 ? pick : lastpos;
  }
  __syncthreads();
  if (tid == 0 && s_last >= 0) predicts[s_last] = s_pick;
}

// Bytes of bits per segment, least significant bit first (numpy.packbits(..., bitorder="little") applied to every
// segment on its own): segment i reads x[input_indptr[i], input_indptr[i + 1]) and writes ceil(len / 8) bytes at
// y[output_indptr[i]].  One block per segment.
__global__ void __launch_bounds__(256) segment_packbits_kernel(const uint8_t* __restrict__ x,
                                                               const int32_t* __restrict__ input_indptr,
                                                               const int32_t* __restrict__ output_indptr,
                                                               uint8_t* __restrict__ y) {
  const int seg = blockIdx.x;
  const int64_t in0 = input_indptr[seg];
  const int len = input_indptr[seg + 1] - input_indptr[seg];
  const int64_t out0 = output_indptr[seg];
  const int room = output_indptr[seg + 1] - output_indptr[seg];
  const int nbytes = min((len + 7) / 8, room);
  for (int k = threadIdx.x; k < nbytes; k += 256) {
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (8 * k + j < len && x[in0 + 8 * k + j]) v |= 1u << j;
    y[out0 + k] = (uint8_t)v;
  }
}

}  // namespace
}  // namespace sglk

using namespace sglk;

extern "C" {

int sglk_build_tree_efficient(const int64_t* parent_list, const int64_t* selected_index,
                              const int32_t* verified_seq_len, uint8_t* tree_mask, int64_t* positions,
                              int64_t* retrive_index, int64_t* retrive_next_token, int64_t* retrive_next_sibling,
                              int topk, int depth, int draft_token_num, int bs, int64_t tree_mask_len, void* stream) {
  SGLK_CHECK(bs >= 0 && topk > 0 && depth > 0, "build_tree_efficient: bad shape");
  SGLK_CHECK(draft_token_num >= 1 && draft_token_num <= kMaxDraft, "build_tree_efficient: draft_token_num must be in [1, 1024]");
  if (bs == 0) return SGLK_OK;
  SGLK_CHECK(selected_index && verified_seq_len && tree_mask && positions && retrive_index && retrive_next_token &&
                 retrive_next_sibling,
             "build_tree_efficient: null pointer");
  SGLK_CHECK(parent_list || depth == 1, "build_tree_efficient: null parent_list");
  const int threads = ((draft_token_num + 63) / 64) * 64;
  hipLaunchKernelGGL(build_tree_kernel, dim3(bs), dim3(threads), 0, (hipStream_t)stream, parent_list, selected_index,
                     verified_seq_len, tree_mask, positions, retrive_index, retrive_next_token, retrive_next_sibling, topk,
                     depth, draft_token_num, tree_mask_len);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_verify_tree_greedy(int32_t* predicts, int32_t* accept_index, int32_t* accept_token_num,
                            const int32_t* candidates, const int32_t* retrive_index, const int32_t* retrive_next_token,
                            const int32_t* retrive_next_sibling, const int32_t* target_predict, int bs, int num_spec_step,
                            int num_draft_tokens, int total_draft_tokens, void* stream) {
  SGLK_CHECK(bs >= 0 && num_spec_step >= 1 && num_draft_tokens >= 1 && total_draft_tokens >= 0,
             "verify_tree_greedy: bad shape");
  if (bs == 0) return SGLK_OK;
  SGLK_CHECK(predicts && accept_index && accept_token_num && candidates && retrive_index && retrive_next_token &&
                 retrive_next_sibling && target_predict,
             "verify_tree_greedy: null pointer");
  hipLaunchKernelGGL(verify_tree_greedy_kernel, dim3((bs + 63) / 64), dim3(64), 0, (hipStream_t)stream, predicts,
                     accept_index, accept_token_num, candidates, retrive_index, retrive_next_token, retrive_next_sibling,
                     target_predict, bs, num_spec_step, num_draft_tokens, total_draft_tokens);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_tree_speculative_sampling_target_only(int32_t* predicts, int32_t* accept_index, int32_t* accept_token_num,
                                               const int32_t* candidates, const int32_t* retrive_index,
                                               const int32_t* retrive_next_token, const int32_t* retrive_next_sibling,
                                               const float* uniform_samples, const float* target_probs,
                                               float* draft_probs, int bs, int num_spec_step, int num_draft_tokens,
                                               int total_draft_tokens, int vocab, float threshold_single,
                                               float threshold_acc, void* stream) {
  SGLK_CHECK(bs >= 0 && num_spec_step >= 1 && num_draft_tokens >= 1 && vocab >= 1 && total_draft_tokens >= 0,
             "tree_speculative_sampling_target_only: bad shape");
  if (bs == 0) return SGLK_OK;
  SGLK_CHECK(predicts && accept_index && accept_token_num && candidates && retrive_index && retrive_next_token &&
                 retrive_next_sibling && uniform_samples && target_probs && draft_probs,
             "tree_speculative_sampling_target_only: null pointer");
  hipLaunchKernelGGL(tree_spec_sampling_kernel, dim3(bs), dim3(kSpecThreads), 0, (hipStream_t)stream, predicts,
                     accept_index, accept_token_num, candidates, retrive_index, retrive_next_token, retrive_next_sibling,
                     uniform_samples, target_probs, draft_probs, num_spec_step, num_draft_tokens, vocab, total_draft_tokens,
                     threshold_single,
                     fmaxf(threshold_acc, 1e-9f));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_segment_packbits(const uint8_t* x, const int32_t* input_indptr, const int32_t* output_indptr, uint8_t* y,
                          int batch, void* stream) {
  SGLK_CHECK(batch >= 0, "segment_packbits: bad batch");
  if (batch == 0) return SGLK_OK;
  SGLK_CHECK(x && input_indptr && output_indptr && y, "segment_packbits: null pointer");
  hipLaunchKernelGGL(segment_packbits_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, x, input_indptr,
                     output_indptr, y);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}
}
