// Probability filters and inverse-CDF sampling for the sampler's GPU path (top-k / top-p / min-p): one block per
// row, thresholds found by radix selection on the float bit pattern (exact k-th largest, smallest value whose
// cumulative mass reaches a target), no sort.  Byte / integer work plus row reductions: LDS histograms, no matrix
// cores.
#include "common.h"

namespace sglk {
namespace {

constexpr int kSampThreads = 1024;

struct SampShared {
  uint32_t cnt[256];
  float mass[256];
  float red[16];
  float scan[16];
  uint32_t prefix;
  float below;
  int remaining;
  int pick;
};

__device__ __forceinline__ float bsum(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) t += red[w];
  __syncthreads();
  return t;
}
__device__ __forceinline__ float bmax(float v, float* red) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) t = fmaxf(t, red[w]);
  __syncthreads();
  return t;
}

// Bits of the k-th largest value of x[0..n) (probabilities: non-negative, so the bit pattern orders like the value).
__device__ uint32_t kth_largest_bits(const float* __restrict__ x, int n, int k, SampShared& sh) {
  if (threadIdx.x == 0) {
    sh.prefix = 0;
    sh.remaining = k;
  }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (threadIdx.x < 256) sh.cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t prefix = sh.prefix;
    for (int i = threadIdx.x; i < n; i += kSampThreads) {
      const uint32_t b = __float_as_uint(fmaxf(x[i], 0.f));
      if (pass == 0 || (b >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&sh.cnt[(b >> shift) & 255], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int need = sh.remaining;
      int d = 255;
      for (; d > 0; --d) {  // from the top: the bin that holds the need-th largest candidate
        if ((int)sh.cnt[d] >= need) break;
        need -= (int)sh.cnt[d];
      }
      sh.prefix = prefix | ((uint32_t)d << shift);
      sh.remaining = need;
    }
    __syncthreads();
  }
  return sh.prefix;
}

// Bits of the smallest value v >= lo with F(v) >= target, F(v) = mass of the elements in [lo, v]: the first element kept
// when the ascending cumulative sum is cut at `target` (elements equal to it are all kept).
__device__ uint32_t mass_cut_bits(const float* __restrict__ x, int n, float lo, float target, SampShared& sh) {
  if (threadIdx.x == 0) {
    sh.prefix = 0;
    sh.below = 0.f;
  }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (threadIdx.x < 256) sh.mass[threadIdx.x] = 0.f;
    __syncthreads();
    const uint32_t prefix = sh.prefix;
    for (int i = threadIdx.x; i < n; i += kSampThreads) {
      const float v = fmaxf(x[i], 0.f);
      const uint32_t b = __float_as_uint(v);
      if (v >= lo && (pass == 0 || (b >> (shift + 8)) == (prefix >> (shift + 8)))) atomicAdd(&sh.mass[(b >> shift) & 255], v);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      // from the bottom: the bin in which the cumulative mass reaches the target; if rounding keeps the sum just
      // short of a target near the total, the highest occupied bin
      float below = sh.below, run = sh.below;
      int d = -1, last = 0;
      float below_last = sh.below;
      for (int j = 0; j < 256 && d < 0; ++j) {
        if (sh.mass[j] > 0.f) {
          last = j;
          below_last = run;
        }
        if (run + sh.mass[j] >= target && sh.mass[j] > 0.f) {
          d = j;
          below = run;
        }
        run += sh.mass[j];
      }
      if (d < 0) {
        d = last;
        below = below_last;
      }
      sh.prefix = prefix | ((uint32_t)d << shift);
      sh.below = below;
    }
    __syncthreads();
  }
  return sh.prefix;
}

struct RowParams {
  int top_k;      // <= 0 or >= n: no top-k filter
  float top_p;    // >= 1: no top-p filter
  float min_p;    // <= 0: no min-p filter
  int joint;      // 1: both filters on the original distribution; 0: top-p on the top-k renormalised one
};

// threshold of the kept set {x >= thr} for the row's filters
__device__ float row_threshold(const float* __restrict__ x, int n, const RowParams& rp, SampShared& sh) {
  float thr = 0.f;
  const bool use_k = rp.top_k > 0 && rp.top_k < n;
  if (use_k) thr = __uint_as_float(kth_largest_bits(x, n, rp.top_k, sh));
  if (rp.top_p < 1.0f) {
    const float lo = (use_k && !rp.joint) ? thr : 0.f;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += kSampThreads) {
      const float v = fmaxf(x[i], 0.f);
      s += v >= lo ? v : 0.f;
    }
    const float total = bsum(s, sh.red);
    const float cut = __uint_as_float(mass_cut_bits(x, n, lo, (1.0f - rp.top_p) * total, sh));
    thr = fmaxf(thr, cut);
  }
  if (rp.min_p > 0.f) {
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += kSampThreads) m = fmaxf(m, x[i]);
    thr = fmaxf(thr, rp.min_p * bmax(m, sh.red));
  }
  return thr;
}

__device__ __forceinline__ RowParams load_params(int row, const int32_t* k_arr, int k_val, const float* p_arr, float p_val,
                                                 const float* m_arr, float m_val, int joint) {
  RowParams rp;
  rp.top_k = k_arr ? k_arr[row] : k_val;
  rp.top_p = p_arr ? p_arr[row] : p_val;
  rp.min_p = m_arr ? m_arr[row] : m_val;
  rp.joint = joint;
  return rp;
}

__global__ void __launch_bounds__(kSampThreads) renorm_kernel(const float* __restrict__ probs, float* __restrict__ out,
                                                             int n, int64_t in_stride, int64_t out_stride,
                                                             const int32_t* k_arr, int k_val, const float* p_arr,
                                                             float p_val) {
  __shared__ SampShared sh;
  const int row = blockIdx.x;
  const float* x = probs + (int64_t)row * in_stride;
  const RowParams rp = load_params(row, k_arr, k_val, p_arr, p_val, nullptr, 0.f, 1);
  const float thr = row_threshold(x, n, rp, sh);
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += kSampThreads) s += x[i] >= thr ? x[i] : 0.f;
  const float inv = 1.0f / bsum(s, sh.red);
  float* o = out + (int64_t)row * out_stride;
  for (int i = threadIdx.x; i < n; i += kSampThreads) o[i] = x[i] >= thr ? x[i] * inv : 0.f;
}

// Inverse CDF over the kept elements in index order: the first kept index whose running sum exceeds u * total.
__global__ void __launch_bounds__(kSampThreads) sample_kernel(const float* __restrict__ probs,
                                                             const float* __restrict__ uniform, int32_t* __restrict__ out,
                                                             int n, int64_t in_stride, const int32_t* k_arr, int k_val,
                                                             const float* p_arr, float p_val, const float* m_arr,
                                                             float m_val, int joint) {
  __shared__ SampShared sh;
  const int row = blockIdx.x;
  const float* x = probs + (int64_t)row * in_stride;
  const RowParams rp = load_params(row, k_arr, k_val, p_arr, p_val, m_arr, m_val, joint);
  const float thr = row_threshold(x, n, rp, sh);
  // thread t owns the contiguous chunk [t * per, (t + 1) * per)
  const int per = (n + kSampThreads - 1) / kSampThreads;
  const int lo = min(threadIdx.x * per, n), hi = min(lo + per, n);
  float mine = 0.f;
  for (int i = lo; i < hi; ++i) mine += x[i] >= thr ? x[i] : 0.f;
  // exclusive scan of the chunk sums: within the wave, then over the 16 wave totals
  float incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float v = __shfl_up(incl, o, 64);
    if ((int)(threadIdx.x & 63) >= o) incl += v;
  }
  if ((threadIdx.x & 63) == 63) sh.scan[threadIdx.x >> 6] = incl;
  if (threadIdx.x == 0) sh.pick = -1;
  __syncthreads();
  float wave_base = 0.f, total = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    if (w < (int)(threadIdx.x >> 6)) wave_base += sh.scan[w];
    total += sh.scan[w];
  }
  const float before = wave_base + incl - mine;
  const float target = uniform[row] * total;
  if (mine > 0.f && before <= target && target < before + mine) {
    float run = before;
    int pick = -1, last = -1;
    for (int i = lo; i < hi; ++i)
      if (x[i] >= thr) {
        last = i;
        run += x[i];
        if (run > target) {
          pick = i;
          break;
        }
      }
    sh.pick = pick >= 0 ? pick : last;  // rounding at the chunk's end: its last kept element
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int pick = sh.pick;
    if (pick < 0) {  // u * total fell on a boundary lost to rounding: the last kept element of the row
      for (int i = n - 1; i >= 0 && pick < 0; --i)
        if (x[i] >= thr) pick = i;
      if (pick < 0) pick = 0;
    }
    out[row] = pick;
  }
}

}  // namespace
}  // namespace sglk

using namespace sglk;

extern "C" {

int sglk_top_k_renorm_probs(const float* probs, float* out, const int32_t* top_k_arr, int top_k_val, int batch, int vocab,
                            int64_t in_stride, int64_t out_stride, void* stream) {
  SGLK_CHECK(batch >= 0 && vocab > 0 && in_stride >= vocab && out_stride >= vocab, "top_k_renorm_probs: bad shape");
  if (batch == 0) return SGLK_OK;
  SGLK_CHECK(probs && out, "top_k_renorm_probs: null pointer");
  hipLaunchKernelGGL(renorm_kernel, dim3(batch), dim3(kSampThreads), 0, (hipStream_t)stream, probs, out, vocab, in_stride,
                     out_stride, top_k_arr, top_k_val, (const float*)nullptr, 1.0f);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_top_p_renorm_probs(const float* probs, float* out, const float* top_p_arr, float top_p_val, int batch, int vocab,
                            int64_t in_stride, int64_t out_stride, void* stream) {
  SGLK_CHECK(batch >= 0 && vocab > 0 && in_stride >= vocab && out_stride >= vocab, "top_p_renorm_probs: bad shape");
  if (batch == 0) return SGLK_OK;
  SGLK_CHECK(probs && out, "top_p_renorm_probs: null pointer");
  hipLaunchKernelGGL(renorm_kernel, dim3(batch), dim3(kSampThreads), 0, (hipStream_t)stream, probs, out, vocab, in_stride,
                     out_stride, (const int32_t*)nullptr, 0, top_p_arr, top_p_val);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_sampling_from_probs(const float* probs, const float* uniform, int32_t* out, const int32_t* top_k_arr,
                             int top_k_val, const float* top_p_arr, float top_p_val, const float* min_p_arr,
                             float min_p_val, int joint, int batch, int vocab, int64_t in_stride, void* stream) {
  SGLK_CHECK(batch >= 0 && vocab > 0 && in_stride >= vocab, "sampling_from_probs: bad shape");
  if (batch == 0) return SGLK_OK;
  SGLK_CHECK(probs && uniform && out, "sampling_from_probs: null pointer");
  hipLaunchKernelGGL(sample_kernel, dim3(batch), dim3(kSampThreads), 0, (hipStream_t)stream, probs, uniform, out, vocab,
                     in_stride, top_k_arr, top_k_val, top_p_arr, top_p_val, min_p_arr, min_p_val, joint ? 1 : 0);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}
}
