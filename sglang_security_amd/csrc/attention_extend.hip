// Extend (prefill with a cached prefix) attention.
//
// Per sequence b: queries are the rows qo_indptr[b] .. qo_indptr[b+1] of q_extend; they see the cached prefix
// (pool rows kv_indices[kv_indptr[b] .. kv_indptr[b+1]) of k_buffer / v_buffer, unmasked) and then the new
// tokens themselves (the same rows of k_extend / v_extend, under the causal mask).  One online softmax
// runs over both ranges (triton_ops/extend_attention.py:306-438 does the same in two loops).
//
// Fast path (16-bit, D = Dv = 128): one block per (sequence, query head, 64*NC query rows); the four waves
// share every 32-token K/V tile through LDS and each multiplies it against NC tiles of 16 query rows on
// the matrix cores.  Everything else takes a wave-per-(query row, head) kernel.
#include <type_traits>

#include "attention_tile.h"
#include "common.h"

namespace sglk {
namespace {

constexpr int kNC = 2;
#ifndef SGLK_EXTEND_PF
#define SGLK_EXTEND_PF 1
#endif
constexpr int kExtPF = SGLK_EXTEND_PF;  // K / V tiles in flight ahead of the one being multiplied

template <typename T, int NC, typename KV = T>
__global__ void __launch_bounds__(256) extend_attention_mfma128(
    const void* __restrict__ q_extend, const void* __restrict__ k_extend, const void* __restrict__ v_extend,
    void* __restrict__ o_extend, const void* __restrict__ k_buffer, const void* __restrict__ v_buffer,
    const int32_t* __restrict__ qo_indptr, const int32_t* __restrict__ kv_indptr,
    const int32_t* __restrict__ kv_indices, int Hq, int Hkv, int64_t q_stride_t, int64_t q_stride_h,
    int64_t ke_stride_t, int64_t ke_stride_h, int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t,
    int64_t o_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, float sm_scale,
    float logit_cap, int causal, int kv_splits, float* __restrict__ part_o, float* __restrict__ part_lse) {
  // kv_splits > 1 (new-token form only): blockIdx.x = query block * kv_splits + split; split s walks its slice of the
  // cached prefix, the last split also the new tokens; every split leaves a normalised fp32 partial + log-sum-exp per
  // (query row, head) in part_o [T, Hq, kv_splits, 128] / part_lse [T, Hq, kv_splits] for extend_merge_kernel.  A
  // single request's short extend over a long prefix has too few (query block, head) pairs to fill the chip otherwise.
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = qo_indptr[b];
  const int ext_len = qo_indptr[b + 1] - q0;
  const int split = kv_splits > 1 ? (int)blockIdx.x % kv_splits : 0;
  const int block_q0 = (kv_splits > 1 ? (int)blockIdx.x / kv_splits : (int)blockIdx.x) * 64 * NC;
  if (block_q0 >= ext_len) return;  // block-uniform
  // fp8 pool: code -> value table behind the two stages (visible after the barrier that opens every run())
  constexpr bool KV8 = !std::is_same<KV, T>::value;
  [[maybe_unused]] uint16_t* lut = reinterpret_cast<uint16_t*>(lds + kBlockAttnLds);
  if constexpr (KV8) lut[threadIdx.x] = Conv<T>::from_f32(KvElem<KV>::value(threadIdx.x));  // 256 threads, 256 codes
  const int pstart = kv_indptr[b];
  const int plen = kv_indptr[b + 1] - pstart;
  const int kvh = h / (Hq / Hkv);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int hl = lane & 15, g = lane >> 4;

  BlockAttn<T, NC> st;
  st.init();
  int qpos[NC], qrow[NC];
  bool col_ok[NC];
  int wave_qmax = 0;
  const uint16_t* qb = reinterpret_cast<const uint16_t*>(q_extend) + (int64_t)h * q_stride_h;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int tile = c * 4 + wid;  // interleaved so that every wave gets early and late rows
    const int row = block_q0 + tile * 16 + hl;
    col_ok[c] = row < ext_len;
    qpos[c] = min(row, ext_len - 1);
    qrow[c] = qpos[c];
    wave_qmax = max(wave_qmax, min(block_q0 + tile * 16 + 15, ext_len - 1));
    st.load_q(c, qb + (int64_t)(q0 + qpos[c]) * q_stride_t);
  }

  if (k_extend == nullptr) {
    // paged form: kv_indices lists ALL tokens of the sequence (the new ones are already in the pool); query row i
    // sees tokens 0 .. plen - ext_len + i
    const int shift = plen - ext_len;
#pragma unroll
    for (int c = 0; c < NC; ++c) qpos[c] += shift;
    // (an fp8 pool: byte pointers and byte strides, carried in the 16-bit pointer type of run())
    constexpr int kEl = KV8 ? 1 : 2;
    const uint16_t* kb = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(k_buffer) + (int64_t)kvh * k_stride_h * kEl);
    const uint16_t* vb = reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(v_buffer) + (int64_t)kvh * v_stride_h * kEl);
    if (causal) {
      const int t_end = min(plen, shift + block_q0 + 64 * NC);
      if (t_end > 0)
        st.template run<true, true, kExtPF, typename BlockAttn<T, NC>::NoHook, KV8>(
            lds, kb, vb, k_stride_n, v_stride_n, kv_indices + pstart, 0, t_end, qpos, wave_qmax + shift, sm_scale,
            logit_cap, typename BlockAttn<T, NC>::NoHook(), lut);
    } else if (plen > 0) {
      st.template run<false, true, kExtPF, typename BlockAttn<T, NC>::NoHook, KV8>(
          lds, kb, vb, k_stride_n, v_stride_n, kv_indices + pstart, 0, plen, qpos, 0, sm_scale, logit_cap,
          typename BlockAttn<T, NC>::NoHook(), lut);
    }
  } else {
  int p0 = 0, p1 = plen;
  if (kv_splits > 1) {
    const int per = (((plen + kv_splits - 1) / kv_splits) + 31) / 32 * 32;
    p0 = min(per * split, plen);
    p1 = min(p0 + per, plen);
  }
  if (p1 > p0) {
    constexpr int kEl = KV8 ? 1 : 2;
    st.template run<false, true, kExtPF, typename BlockAttn<T, NC>::NoHook, KV8>(
        lds, reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(k_buffer) + (int64_t)kvh * k_stride_h * kEl),
        reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(v_buffer) + (int64_t)kvh * v_stride_h * kEl),
        k_stride_n, v_stride_n, kv_indices + pstart, p0, p1, qpos, wave_qmax, sm_scale, logit_cap,
        typename BlockAttn<T, NC>::NoHook(), lut);
  }
  const uint16_t* ke = reinterpret_cast<const uint16_t*>(k_extend) + (int64_t)q0 * ke_stride_t + (int64_t)kvh * ke_stride_h;
  const uint16_t* ve = reinterpret_cast<const uint16_t*>(v_extend) + (int64_t)q0 * ve_stride_t + (int64_t)kvh * ve_stride_h;
  if (kv_splits > 1 && split != kv_splits - 1) {
    // the new tokens belong to the last split
  } else if (causal) {
    const int t_end = min(ext_len, block_q0 + 64 * NC);
    st.template run<true, false, kExtPF>(lds, ke, ve, ke_stride_t, ve_stride_t, nullptr, 0, t_end, qpos, wave_qmax, sm_scale,
                                 logit_cap);
  } else {
    st.template run<false, false, kExtPF>(lds, ke, ve, ke_stride_t, ve_stride_t, nullptr, 0, ext_len, qpos, wave_qmax, sm_scale,
                                  logit_cap);
  }
  }

  if (kv_splits > 1) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float tot = st.total(c);
      if (!col_ok[c]) continue;
      const float inv = tot > 0.f ? 1.0f / tot : 0.f;
      const int64_t row = ((int64_t)(q0 + qrow[c]) * Hq + h) * kv_splits + split;
      float* dst = part_o + row * 128;
#pragma unroll
      for (int n = 0; n < 8; ++n) *reinterpret_cast<float4v*>(dst + 16 * n + 4 * g) = st.oacc[c][n] * inv;
      if (g == 0) part_lse[row] = tot > 0.f ? (st.m[c] + log2f(tot)) * 0.6931471805599453f : -INFINITY;
    }
    return;
  }
  uint16_t* ob = reinterpret_cast<uint16_t*>(o_extend) + (int64_t)h * o_stride_h;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float inv = 1.0f / st.total(c);
    if (!col_ok[c]) continue;
    uint16_t* dst = ob + (int64_t)(q0 + qrow[c]) * o_stride_t;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      // lane (hl, g) holds O[row hl][16n + 4g + r], r = 0..3
      short4v pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) pk[r] = (short)Conv<T>::from_f32(st.oacc[c][n][r] * inv);
      *reinterpret_cast<short4v*>(dst + 16 * n + 4 * g) = pk;
    }
  }
}

// merges the kv_splits partials of one (query row, head) by log-sum-exp, in split order (merge_state arithmetic)
template <typename T>
__global__ void __launch_bounds__(128) extend_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_lse,
                                                         void* __restrict__ o, int Hq, int kv_splits, int64_t o_stride_t,
                                                         int64_t o_stride_h, const int32_t* __restrict__ qo_indptr, int B) {
  const int64_t t = blockIdx.x;
  if (t >= qo_indptr[B]) return;  // rows of an over-allocated q_extend hold no partials: leave their output alone
  const int h = blockIdx.y, d = threadIdx.x;
  const int64_t row0 = (t * Hq + h) * kv_splits;
  float e_max = -INFINITY, e_sum = 0.f, acc = 0.f;
  for (int s = 0; s < kv_splits; ++s) {
    const float l = part_lse[row0 + s];
    if (l > -INFINITY) {
      const float v = part_o[(row0 + s) * 128 + d];
      const float n_max = fmaxf(l, e_max);
      const float old = expf(e_max - n_max), w = expf(l - n_max);
      acc = acc * old + w * v;
      e_sum = e_sum * old + w;
      e_max = n_max;
    }
  }
  st<T>(o, t * o_stride_t + (int64_t)h * o_stride_h + d, e_sum > 0.f ? acc / e_sum : 0.f);
}

// one wave per (query row, head): any head size, any float type
template <typename T, typename KV, int DV_PER_LANE>
__global__ void __launch_bounds__(64) extend_attention_generic(
    const void* __restrict__ q_extend, const void* __restrict__ k_extend, const void* __restrict__ v_extend,
    void* __restrict__ o_extend, const void* __restrict__ k_buffer, const void* __restrict__ v_buffer,
    const int32_t* __restrict__ qo_indptr, const int32_t* __restrict__ kv_indptr,
    const int32_t* __restrict__ kv_indices, int Hq, int Hkv, int D, int Dv, int64_t q_stride_t, int64_t q_stride_h,
    int64_t ke_stride_t, int64_t ke_stride_h, int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t,
    int64_t o_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, float sm_scale,
    float logit_cap, int causal, const uint8_t* __restrict__ custom_mask, const int64_t* __restrict__ mask_indptr,
    int skip_prefix_mask, int window) {
  extern __shared__ float qs[];  // D floats
  const int b = blockIdx.z, h = blockIdx.y, qi = blockIdx.x;
  const int q0 = qo_indptr[b];
  const int ext_len = qo_indptr[b + 1] - q0;
  if (qi >= ext_len) return;
  const int pstart = kv_indptr[b];
  const int plen = kv_indptr[b + 1] - pstart;
  const int kvh = h / (Hq / Hkv);
  const int lane = threadIdx.x;
  for (int d = lane; d < D; d += 64) qs[d] = ld<T>(q_extend, (int64_t)(q0 + qi) * q_stride_t + (int64_t)h * q_stride_h + d);
  __syncthreads();
  const bool paged = k_extend == nullptr;   // kv_indices lists every token; row qi sees plen - ext_len + qi + 1 of them
  // with a custom mask (row qi of the request: plen + ext_len bytes, prefix columns first) the mask decides which new
  // tokens are visible, and, unless skip_prefix_mask, which prefix tokens
  const uint8_t* mrow = (custom_mask && !paged) ? custom_mask + mask_indptr[b] + (int64_t)qi * (plen + ext_len) : nullptr;
  const int vis = paged ? 0 : (mrow ? ext_len : (causal ? qi + 1 : ext_len));  // visible new tokens
  const int total = paged ? (causal ? plen - ext_len + qi + 1 : plen) : plen + vis;
  float m = -INFINITY, l = 0.f;
  float acc[DV_PER_LANE];
#pragma unroll
  for (int i = 0; i < DV_PER_LANE; ++i) acc[i] = 0.f;
  for (int t0 = 0; t0 < total; t0 += 64) {
    const int tok = t0 + lane;
    const bool valid = tok < total;
    // where this lane's token lives
    const bool in_prefix = paged || tok < plen;
    const void* kp = in_prefix ? k_buffer : k_extend;
    const void* vp = in_prefix ? v_buffer : v_extend;
    int64_t kb = 0, vb = 0;
    if (valid) {
      if (in_prefix) {
        const int64_t loc = kv_indices[pstart + tok];
        kb = loc * k_stride_n + (int64_t)kvh * k_stride_h;
        vb = loc * v_stride_n + (int64_t)kvh * v_stride_h;
      } else {
        const int64_t r = q0 + (tok - plen);
        kb = r * ke_stride_t + (int64_t)kvh * ke_stride_h;
        vb = r * ve_stride_t + (int64_t)kvh * ve_stride_h;
      }
    }
    float s = -INFINITY;
    // sliding window, as the reference masks it (extend_attention.py:179-184): only in the stage over the listed
    // prefix, row qi of the extend sees listed prefix token n iff qi <= n + window
    const bool in_window = !(window > 0 && in_prefix && !paged) || qi <= tok + window;
    const bool shown = valid && in_window && (!mrow || (in_prefix && skip_prefix_mask) || mrow[tok] != 0);
    if (shown) {
      s = 0.f;
      if (in_prefix) {
        for (int d = 0; d < D; ++d) s += qs[d] * KvElem<KV>::ld(kp, kb + d);
      } else {
        for (int d = 0; d < D; ++d) s += qs[d] * ld<T>(kp, kb + d);
      }
      s *= sm_scale;
      if (logit_cap > 0.f) s = logit_cap * tanhf(s / logit_cap);
    }
    const float m_new = fmaxf(m, wave_max(s));
    if (m_new == -INFINITY) continue;  // nothing visible so far (only possible under a custom mask)
    const float alpha = expf(m - m_new);
    const float p = shown ? expf(s - m_new) : 0.f;
    l = l * alpha + wave_sum(p);
    m = m_new;
#pragma unroll
    for (int i = 0; i < DV_PER_LANE; ++i) acc[i] *= alpha;
    const int cnt = min(64, total - t0);
    for (int j = 0; j < cnt; ++j) {
      const float pj = __shfl(p, j, 64);
      const int64_t vj = __shfl((long long)vb, j, 64);
      const bool from_pool = paged || (t0 + j) < plen;
#pragma unroll
      for (int i = 0; i < DV_PER_LANE; ++i) {
        const int d = lane + 64 * i;
        if (d < Dv) acc[i] += pj * (from_pool ? KvElem<KV>::ld(v_buffer, vj + d) : ld<T>(v_extend, vj + d));
      }
    }
    (void)vp;
  }
#pragma unroll
  for (int i = 0; i < DV_PER_LANE; ++i) {
    const int d = lane + 64 * i;
    if (d < Dv) st<T>(o_extend, (int64_t)(q0 + qi) * o_stride_t + (int64_t)h * o_stride_h + d, l > 0.f ? acc[i] / l : 0.f);
  }
}

}  // namespace
}  // namespace sglk

using namespace sglk;

static int extend_attention_impl(const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend,
                                 const void* k_buffer, const void* v_buffer, const int32_t* qo_indptr,
                                 const int32_t* kv_indptr, const int32_t* kv_indices, int B, int Hq, int Hkv, int D, int Dv,
                                 int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t, int64_t ke_stride_h,
                                 int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t, int64_t o_stride_h,
                                 int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
                                 int max_extend_len, float sm_scale, float logit_cap, int causal, int dtype, void* stream,
                                 bool paged, const uint8_t* custom_mask = nullptr, const int64_t* mask_indptr = nullptr,
                                 int skip_prefix_mask = 1, int window = 0, int kv_splits = 1, float* split_ws = nullptr,
                                 int64_t total_rows = 0) {
  const int kvf = dtype >> SGLK_KV_SHIFT;  // fp8 pool format, 0: pool elements of the query type
  dtype &= (1 << SGLK_KV_SHIFT) - 1;
  SGLK_CHECK(kvf == 0 || dtype == SGLK_F16 || dtype == SGLK_BF16, "extend_attention: an fp8 pool needs fp16 / bf16 queries");
  SGLK_CHECK(!custom_mask || (mask_indptr && !paged), "extend_attention: a custom mask needs mask_indptr and new K/V");
  SGLK_CHECK(q_extend && o_extend && qo_indptr && kv_indptr && (paged || (k_extend && v_extend)),
             "extend_attention: null pointer");
  SGLK_CHECK(B >= 0 && Hq > 0 && Hkv > 0 && D > 0 && Dv > 0, "extend_attention: bad shape");
  SGLK_CHECK(Hq % Hkv == 0, "extend_attention: Hq %d not a multiple of Hkv %d", Hq, Hkv);
  SGLK_CHECK(Dv <= 1024 && D <= 4096, "extend_attention: head size too large");
  SGLK_CHECK(max_extend_len >= 0, "extend_attention: negative max_extend_len");
  if (B == 0 || max_extend_len == 0) return SGLK_OK;
  // a batch without any cached prefix may pass no pool at all
  SGLK_CHECK(kvf == 0 || (k_buffer && v_buffer), "extend_attention: an fp8 pool format without a pool");
  if (!k_buffer) k_buffer = k_extend;
  if (!v_buffer) v_buffer = v_extend;
  if (paged) k_extend = v_extend = nullptr;  // the kernels take a null k_extend as "every token is in the pool"
  hipStream_t s = (hipStream_t)stream;
  auto mult8 = [](int64_t v) { return v % 8 == 0; };
  auto al8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
  // the pool side of the tile kernel's requirements: 16-byte row pieces of 16-bit elements, or 8-byte pieces of codes
  const bool pool_ok = (kvf ? (al8(k_buffer) && al8(v_buffer)) : (aligned16(k_buffer) && aligned16(v_buffer))) &&
                       mult8(k_stride_n) && mult8(k_stride_h) && mult8(v_stride_n) && mult8(v_stride_h);
  const bool fast = !custom_mask && window <= 0 && (dtype == SGLK_F16 || dtype == SGLK_BF16) && D == 128 && Dv == 128 &&
                    aligned16(q_extend) && (paged || (aligned16(k_extend) && aligned16(v_extend))) && pool_ok &&
                    ((uintptr_t)o_extend % 8 == 0) && mult8(q_stride_t) && mult8(q_stride_h) && mult8(ke_stride_t) &&
                    mult8(ke_stride_h) && mult8(ve_stride_t) && mult8(ve_stride_h) && o_stride_t % 4 == 0 &&
                    o_stride_h % 4 == 0;
  // One column tile per wave (64 query rows per block) keeps the kernel at two blocks per CU; two tiles per wave
  // halve the K/V re-reads but run at one wave per SIMD and measured 25 % slower (SGLK_EXTEND_NC=2 selects them).
  const char* env_nc = getenv("SGLK_EXTEND_NC");
  const size_t tile_lds = kBlockAttnLds + (kvf ? kKv8LutBytes : 0);
#define SGLK_EXT_TILE(NCV, GRID, SPLITS, PO, PL)                                                                       \
  SGLK_DISPATCH_HALF(dtype, T, {                                                                                       \
    SGLK_DISPATCH_KV(kvf, T, KV, {                                                                                     \
      hipLaunchKernelGGL((extend_attention_mfma128<T, NCV, KV>), GRID, dim3(256), tile_lds, s, q_extend, k_extend,      \
                         v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, Hq, Hkv, q_stride_t, \
                         q_stride_h, ke_stride_t, ke_stride_h, ve_stride_t, ve_stride_h, o_stride_t, o_stride_h,        \
                         k_stride_n, k_stride_h, v_stride_n, v_stride_h, sm_scale, logit_cap, causal, SPLITS, PO, PL);  \
    });                                                                                                                \
  })
  if (kv_splits > 1) {
    if (!fast || paged)
      return fail(SGLK_EUNSUP, "extend_attention_split: 16-bit types with head size 128 and contiguous new tokens only");
    float* part_o = split_ws;
    float* part_lse = split_ws + total_rows * Hq * kv_splits * 128;
    dim3 grid(((max_extend_len + 63) / 64) * kv_splits, Hq, B);
    SGLK_EXT_TILE(1, grid, kv_splits, part_o, part_lse);
    SGLK_LAUNCH_CHECK();
    SGLK_DISPATCH_HALF(dtype, T,
                       hipLaunchKernelGGL((extend_merge_kernel<T>), dim3((unsigned)total_rows, Hq), dim3(128), 0, s, part_o,
                                          part_lse, o_extend, Hq, kv_splits, o_stride_t, o_stride_h, qo_indptr, B));
  } else if (fast && !(env_nc && atoi(env_nc) == 2)) {
    dim3 grid((max_extend_len + 63) / 64, Hq, B);
    SGLK_EXT_TILE(1, grid, 1, (float*)nullptr, (float*)nullptr);
  } else if (fast) {
    dim3 grid((max_extend_len + 64 * kNC - 1) / (64 * kNC), Hq, B);
    SGLK_EXT_TILE(kNC, grid, 1, (float*)nullptr, (float*)nullptr);
  } else {
    dim3 grid(max_extend_len, Hq, B);
    const size_t lds = (size_t)D * sizeof(float);
    const int per_lane = (Dv + 63) / 64;
#define SGLK_GEN(N)                                                                                                   \
  hipLaunchKernelGGL((extend_attention_generic<T, KV, N>), grid, dim3(64), lds, s, q_extend, k_extend, v_extend, o_extend, \
                     k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, Hq, Hkv, D, Dv, q_stride_t, q_stride_h,     \
                     ke_stride_t, ke_stride_h, ve_stride_t, ve_stride_h, o_stride_t, o_stride_h, k_stride_n,           \
                     k_stride_h, v_stride_n, v_stride_h, sm_scale, logit_cap, causal, custom_mask, mask_indptr,        \
                     skip_prefix_mask, window)
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
#undef SGLK_EXT_TILE
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

extern "C" int sglk_extend_attention(const void* q_extend, const void* k_extend, const void* v_extend, void* o_extend,
                                     const void* k_buffer, const void* v_buffer, const int32_t* qo_indptr,
                                     const int32_t* kv_indptr, const int32_t* kv_indices, int B, int Hq, int Hkv, int D,
                                     int Dv, int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t,
                                     int64_t ke_stride_h, int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t,
                                     int64_t o_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n,
                                     int64_t v_stride_h, int max_extend_len, float sm_scale, float logit_cap, int causal,
                                     int dtype, void* stream) {
  return extend_attention_impl(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices,
                               B, Hq, Hkv, D, Dv, q_stride_t, q_stride_h, ke_stride_t, ke_stride_h, ve_stride_t, ve_stride_h,
                               o_stride_t, o_stride_h, k_stride_n, k_stride_h, v_stride_n, v_stride_h, max_extend_len,
                               sm_scale, logit_cap, causal, dtype, stream, false);
}

extern "C" int sglk_extend_attention_masked(const void* q_extend, const void* k_extend, const void* v_extend,
                                            void* o_extend, const void* k_buffer, const void* v_buffer,
                                            const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices,
                                            const uint8_t* custom_mask, const int64_t* mask_indptr,
                                            int skip_prefix_custom_mask, int B, int Hq, int Hkv, int D, int Dv,
                                            int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t,
                                            int64_t ke_stride_h, int64_t ve_stride_t, int64_t ve_stride_h,
                                            int64_t o_stride_t, int64_t o_stride_h, int64_t k_stride_n, int64_t k_stride_h,
                                            int64_t v_stride_n, int64_t v_stride_h, int max_extend_len, float sm_scale,
                                            float logit_cap, int dtype, void* stream) {
  SGLK_CHECK(custom_mask && mask_indptr, "extend_attention_masked: null mask");
  return extend_attention_impl(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices,
                               B, Hq, Hkv, D, Dv, q_stride_t, q_stride_h, ke_stride_t, ke_stride_h, ve_stride_t, ve_stride_h,
                               o_stride_t, o_stride_h, k_stride_n, k_stride_h, v_stride_n, v_stride_h, max_extend_len,
                               sm_scale, logit_cap, 1, dtype, stream, false, custom_mask, mask_indptr,
                               skip_prefix_custom_mask ? 1 : 0);
}

extern "C" int64_t sglk_extend_attention_split_workspace_bytes(int64_t total_rows, int Hq, int kv_splits) {
  if (total_rows <= 0 || Hq <= 0 || kv_splits <= 0) return 0;
  return total_rows * Hq * kv_splits * (128 + 1) * (int64_t)sizeof(float);
}

extern "C" int sglk_extend_attention_split(const void* q_extend, const void* k_extend, const void* v_extend,
                                           void* o_extend, const void* k_buffer, const void* v_buffer,
                                           const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices,
                                           int B, int Hq, int Hkv, int D, int Dv, int64_t q_stride_t, int64_t q_stride_h,
                                           int64_t ke_stride_t, int64_t ke_stride_h, int64_t ve_stride_t,
                                           int64_t ve_stride_h, int64_t o_stride_t, int64_t o_stride_h, int64_t k_stride_n,
                                           int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h, int max_extend_len,
                                           float sm_scale, float logit_cap, int causal, int kv_splits, float* workspace,
                                           int64_t total_rows, int dtype, void* stream) {
  SGLK_CHECK(kv_splits >= 1 && kv_splits <= 64, "extend_attention_split: kv_splits must be in 1..64");
  SGLK_CHECK(kv_splits == 1 || (workspace && aligned16(workspace) && total_rows > 0 && k_extend && v_extend),
             "extend_attention_split: needs a 16-byte aligned workspace, the row count and the new K / V");
  return extend_attention_impl(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices,
                               B, Hq, Hkv, D, Dv, q_stride_t, q_stride_h, ke_stride_t, ke_stride_h, ve_stride_t, ve_stride_h,
                               o_stride_t, o_stride_h, k_stride_n, k_stride_h, v_stride_n, v_stride_h, max_extend_len,
                               sm_scale, logit_cap, causal, dtype, stream, false, nullptr, nullptr, 1, 0, kv_splits,
                               workspace, total_rows);
}

extern "C" int sglk_extend_attention_window(const void* q_extend, const void* k_extend, const void* v_extend,
                                            void* o_extend, const void* k_buffer, const void* v_buffer,
                                            const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices,
                                            const uint8_t* custom_mask, const int64_t* mask_indptr,
                                            int skip_prefix_custom_mask, int sliding_window_size, int B, int Hq, int Hkv,
                                            int D, int Dv, int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t,
                                            int64_t ke_stride_h, int64_t ve_stride_t, int64_t ve_stride_h,
                                            int64_t o_stride_t, int64_t o_stride_h, int64_t k_stride_n, int64_t k_stride_h,
                                            int64_t v_stride_n, int64_t v_stride_h, int max_extend_len, float sm_scale,
                                            float logit_cap, int causal, int dtype, void* stream) {
  SGLK_CHECK(!custom_mask || mask_indptr, "extend_attention_window: a custom mask needs mask_indptr");
  return extend_attention_impl(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices,
                               B, Hq, Hkv, D, Dv, q_stride_t, q_stride_h, ke_stride_t, ke_stride_h, ve_stride_t, ve_stride_h,
                               o_stride_t, o_stride_h, k_stride_n, k_stride_h, v_stride_n, v_stride_h, max_extend_len,
                               sm_scale, logit_cap, custom_mask ? 1 : causal, dtype, stream, false, custom_mask, mask_indptr,
                               skip_prefix_custom_mask ? 1 : 0, sliding_window_size > 0 ? sliding_window_size : 0);
}

extern "C" int sglk_prefill_attention_paged(const void* q, void* o, const void* k_buffer, const void* v_buffer,
                                            const int32_t* qo_indptr, const int32_t* kv_indptr, const int32_t* kv_indices,
                                            int B, int Hq, int Hkv, int D, int Dv, int64_t q_stride_t, int64_t q_stride_h,
                                            int64_t o_stride_t, int64_t o_stride_h, int64_t k_stride_n, int64_t k_stride_h,
                                            int64_t v_stride_n, int64_t v_stride_h, int max_q_len, float sm_scale,
                                            float logit_cap, int causal, int dtype, void* stream) {
  SGLK_CHECK(k_buffer && v_buffer && kv_indices, "prefill_attention_paged: null pointer");
  return extend_attention_impl(q, nullptr, nullptr, o, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, B, Hq, Hkv, D,
                               Dv, q_stride_t, q_stride_h, 8, 8, 8, 8, o_stride_t, o_stride_h, k_stride_n, k_stride_h,
                               v_stride_n, v_stride_h, max_q_len, sm_scale, logit_cap, causal, dtype, stream, true);
}
