// Row-wise and elementwise kernels of the decode path: RMSNorm (plain / fused residual add),
// gated activations, rotary embedding, KV-pool row scatter, LSE merge of two partial attentions.
// All are HBM-bound byte movers: 16-byte accesses per lane, fp32 arithmetic, one rounding to the
// activation type where the reference rounds.
#include <algorithm>

#include "common.h"

namespace sglk {

// ---------------------------------------------------------------------------------------------
// block reduction (sum) for blocks of up to 1024 threads
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = (lane < nw) ? red[lane] : 0.f;
  t = wave_sum(t);
  __syncthreads();
  return t;
}

// RMSNorm: out = (x * rsqrt(mean(x^2) + eps)) * w     (layernorm.py:101-103 of the reference)
// FUSED:   s = x + residual (fp32); residual = round(s); out from the UNROUNDED s (layernorm.py:96-99)
template <typename T, bool FUSED, bool VEC>
__global__ void __launch_bounds__(256) rmsnorm_kernel(void* __restrict__ out, const void* __restrict__ xin,
                                                      void* __restrict__ residual, const void* __restrict__ w,
                                                      int hidden, int64_t x_stride, int64_t out_stride, float eps,
                                                      int gemma) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  using R = typename Raw<T>::type;
  const R* x = reinterpret_cast<const R*>(xin) + row * x_stride;
  R* o = reinterpret_cast<R*>(out) + row * out_stride;
  R* res = FUSED ? reinterpret_cast<R*>(residual) + row * x_stride : nullptr;
  const R* wr = reinterpret_cast<const R*>(w);
  float ss = 0.f;
  constexpr int MAXV = 4;  // 256 threads * 8 elems * 4 = 8192 elements kept in registers
  float keep[MAXV][8];
  if constexpr (VEC) {
    // rows of up to 8192 elements live in registers between the two passes (statically indexed);
    // the launcher routes longer rows to the scalar form.
    const int nvec = hidden >> 3;
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
      const int v = threadIdx.x + it * 256;
      if (v < nvec) {
        short8 xv = reinterpret_cast<const short8*>(x)[v];
#pragma unroll
        for (int j = 0; j < 8; ++j) keep[it][j] = Conv<T>::to_f32((uint16_t)xv[j]);
        if constexpr (FUSED) {
          short8 rv = reinterpret_cast<const short8*>(res)[v];
          short8 ro;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            keep[it][j] += Conv<T>::to_f32((uint16_t)rv[j]);
            ro[j] = (short)Conv<T>::from_f32(keep[it][j]);
          }
          reinterpret_cast<short8*>(res)[v] = ro;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += keep[it][j] * keep[it][j];
      }
    }
    const float tot = block_sum(ss, red);
    const float inv = rsqrtf(tot / (float)hidden + eps);
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
      const int v = threadIdx.x + it * 256;
      if (v < nvec) {
        short8 wv = reinterpret_cast<const short8*>(wr)[v];
        short8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float wf = Conv<T>::to_f32((uint16_t)wv[j]);
          if (gemma) wf += 1.0f;
          ov[j] = (short)Conv<T>::from_f32((keep[it][j] * inv) * wf);
        }
        reinterpret_cast<short8*>(o)[v] = ov;
      }
    }
  } else {
    for (int i = threadIdx.x; i < hidden; i += 256) {
      float f = Conv<T>::to_f32(x[i]);
      if constexpr (FUSED) f += Conv<T>::to_f32(res[i]);
      ss += f * f;
    }
    const float tot = block_sum(ss, red);
    const float inv = rsqrtf(tot / (float)hidden + eps);
    for (int i = threadIdx.x; i < hidden; i += 256) {
      float f = Conv<T>::to_f32(x[i]);
      if constexpr (FUSED) {
        f += Conv<T>::to_f32(res[i]);
        res[i] = Conv<T>::from_f32(f);
      }
      float wf = Conv<T>::to_f32(wr[i]);
      if (gemma) wf += 1.0f;
      o[i] = Conv<T>::from_f32((f * inv) * wf);
    }
  }
}

template <typename T>
static int launch_rmsnorm(void* out, const void* x, void* residual, const void* w, int64_t rows, int hidden,
                          int64_t x_stride, int64_t out_stride, float eps, int gemma, hipStream_t s) {
  if (rows == 0) return SGLK_OK;
  const bool fused = residual != nullptr;
  const bool vec = sizeof(typename Raw<T>::type) == 2 && (hidden % 8 == 0) && hidden <= 8192 && aligned16(x) &&
                   aligned16(out) &&
                   aligned16(w) && (x_stride % 8 == 0) && (out_stride % 8 == 0) && (!fused || aligned16(residual));
  dim3 grid((unsigned)rows), block(256);
  if (vec) {
    if (fused)
      hipLaunchKernelGGL((rmsnorm_kernel<T, true, true>), grid, block, 0, s, out, x, residual, w, hidden, x_stride,
                         out_stride, eps, gemma);
    else
      hipLaunchKernelGGL((rmsnorm_kernel<T, false, true>), grid, block, 0, s, out, x, residual, w, hidden, x_stride,
                         out_stride, eps, gemma);
  } else {
    if (fused)
      hipLaunchKernelGGL((rmsnorm_kernel<T, true, false>), grid, block, 0, s, out, x, residual, w, hidden, x_stride,
                         out_stride, eps, gemma);
    else
      hipLaunchKernelGGL((rmsnorm_kernel<T, false, false>), grid, block, 0, s, out, x, residual, w, hidden, x_stride,
                         out_stride, eps, gemma);
  }
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

// ---------------------------------------------------------------------------------------------
// Fusion forms of the decode layer (no counterpart in the reference, which runs these steps as separate ops):
//   * the input row may arrive as split-K partials of the preceding GEMM: x = round_T(sum_s partial[s][row][:] + bias),
//     the value that GEMM would have stored;
//   * the output row's sums over blocks of 128 elements (of the ROUNDED values, i.e. of what the next GEMM reads)
//     may be written to xsum[block][row] for the following fused AWQ GEMM.
// 16-bit types, hidden % 128 == 0, hidden <= 8192.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(1024) rmsnorm_fused_kernel(void* __restrict__ xio, void* __restrict__ residual,
                                                            const void* __restrict__ w, int hidden, float eps, int gemma,
                                                            const float* __restrict__ partials, int splits,
                                                            int64_t part_stride, const void* __restrict__ bias,
                                                            float* __restrict__ xsum_out, int64_t xs_stride,
                                                            float* __restrict__ zero_out, int64_t zero_count) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  // clears an accumulation buffer of a later kernel on the way (block sums gathered with atomic adds)
  for (int64_t i = row * blockDim.x + threadIdx.x; i < zero_count; i += (int64_t)gridDim.x * blockDim.x) zero_out[i] = 0.f;
  uint16_t* x = reinterpret_cast<uint16_t*>(xio) + row * hidden;
  uint16_t* res = residual ? reinterpret_cast<uint16_t*>(residual) + row * hidden : nullptr;
  const uint16_t* wr = reinterpret_cast<const uint16_t*>(w);
  const uint16_t* br = reinterpret_cast<const uint16_t*>(bias);
  constexpr int MAXV = 4;  // 8192 elements at 256 threads; one vector per thread at hidden/8 threads
  float keep[MAXV][8];
  float ss = 0.f;
  const int nvec = hidden >> 3;
  // everything that does not depend on the row statistics is requested up front (norm weights, residual), so that
  // it travels together with the partial sums instead of after the block reduction
  short8 wpre[MAXV], rpre[MAXV];
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int v = threadIdx.x + it * blockDim.x;
    if (v < nvec) {
      wpre[it] = reinterpret_cast<const short8*>(wr)[v];
      if (res) rpre[it] = reinterpret_cast<const short8*>(res)[v];
    }
  }
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int v = threadIdx.x + it * blockDim.x;
    if (v < nvec) {
      if (partials) {
        float acc[8];
        sum_partials(partials, splits, part_stride, row * hidden + 8 * v, acc);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float bj = br ? Conv<T>::to_f32(br[8 * v + j]) : 0.f;
          keep[it][j] = Conv<T>::to_f32(Conv<T>::from_f32(acc[j] + bj));  // what the GEMM would have stored
        }
      } else {
        const short8 xv = reinterpret_cast<const short8*>(x)[v];
#pragma unroll
        for (int j = 0; j < 8; ++j) keep[it][j] = Conv<T>::to_f32((uint16_t)xv[j]);
      }
      if (res) {
        const short8 rv = rpre[it];
        short8 ro;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          keep[it][j] += Conv<T>::to_f32((uint16_t)rv[j]);
          ro[j] = (short)Conv<T>::from_f32(keep[it][j]);
        }
        reinterpret_cast<short8*>(res)[v] = ro;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += keep[it][j] * keep[it][j];
    }
  }
  const float tot = block_sum(ss, red);
  const float inv = rsqrtf(tot / (float)hidden + eps);
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int v = threadIdx.x + it * blockDim.x;
    if (v < nvec) {  // uniform over each group of 16 lanes: hidden % 128 == 0
      const short8 wv = wpre[it];
      short8 ov;
      float bs = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float wf = Conv<T>::to_f32((uint16_t)wv[j]);
        if (gemma) wf += 1.0f;
        const uint16_t o = Conv<T>::from_f32((keep[it][j] * inv) * wf);
        ov[j] = (short)o;
        bs += Conv<T>::to_f32(o);
      }
      reinterpret_cast<short8*>(x)[v] = ov;
      if (xsum_out) {
#pragma unroll
        for (int o2 = 8; o2 > 0; o2 >>= 1) bs += __shfl_xor(bs, o2, 16);
        if ((threadIdx.x & 15) == 0) xsum_out[(int64_t)(v >> 4) * xs_stride + row] = bs;
      }
    }
  }
}

// act_and_mul that also writes the 128-element block sums of its (rounded) output rows
template <typename T, int ACT>
__global__ void __launch_bounds__(256) act_and_mul_xs_kernel(void* __restrict__ out, const void* __restrict__ xin,
                                                             int64_t rows, int d, float* __restrict__ xsum_out,
                                                             int64_t xs_stride);

// rope_store over a qkv row that is first assembled from split-K partials: the row is summed, rounded (the value
// the GEMM would have stored) and kept in LDS, then rotated; q goes to the qkv buffer, k and v to the pool rows
template <typename T>
__global__ void __launch_bounds__(1024) rope_store_partial_kernel(
    void* __restrict__ qkv, const float* __restrict__ partials, int splits, int64_t part_stride,
    const void* __restrict__ bias, const float* __restrict__ cos_sin, const int64_t* __restrict__ positions,
    void* __restrict__ k_buffer, void* __restrict__ v_buffer, const void* loc, int loc_is64, int q_heads, int kv_heads,
    int head_size, int rot, int64_t kbuf_stride_n, int64_t vbuf_stride_n, int interleave, int heads_per_block,
    const float* __restrict__ row_scale, const float* __restrict__ col_scale) {
  // block (t, part) owns heads [h0, h1) of token t's row (q heads, then k heads, then v heads): heads are independent,
  // so a row is spread over several blocks when there are fewer tokens than CUs
  extern __shared__ __attribute__((aligned(16))) uint16_t rowS[];  // heads_per_block * head_size
  const int64_t t = blockIdx.x;
  const int total_heads = q_heads + 2 * kv_heads;
  const int h0 = blockIdx.y * heads_per_block, h1 = min(h0 + heads_per_block, total_heads);
  if (h0 >= h1) return;
  const int row_len = total_heads * head_size;
  const int c0 = h0 * head_size, ncol = (h1 - h0) * head_size;
  // requested before the sums so that they travel together
  const int64_t pos = positions[t];
  const int64_t slot = ld_idx(loc, loc_is64, t);
  const uint16_t* br = reinterpret_cast<const uint16_t*>(bias);
  // raw sums of an fp8 product: scaled by the row's and the column's dequantisation scales, as that product's own
  // split sum does (scaled_sum)
  const float rs = row_scale ? row_scale[t] : 1.f;
  for (int v = threadIdx.x; v < (ncol >> 3); v += blockDim.x) {
    float acc[8];
    sum_partials(partials, splits, part_stride, t * row_len + c0 + 8 * v, acc);
    short8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float bj = br ? Conv<T>::to_f32(br[c0 + 8 * v + j]) : 0.f;
      o[j] = (short)Conv<T>::from_f32(col_scale ? scaled_sum(acc[j], rs, col_scale[c0 + 8 * v + j], bj) : acc[j] + bj);
    }
    reinterpret_cast<short8*>(rowS)[v] = o;
  }
  __syncthreads();
  uint16_t* qdst = reinterpret_cast<uint16_t*>(qkv) + t * row_len;
  const float* cs = cos_sin + pos * rot;
  const int half = rot / 2;
  uint16_t* kdst = reinterpret_cast<uint16_t*>(k_buffer) + slot * kbuf_stride_n;
  uint16_t* vdst = reinterpret_cast<uint16_t*>(v_buffer) + slot * vbuf_stride_n;
  const int rot_heads = max(0, min(h1, q_heads + kv_heads) - h0);  // q and k heads of this block
  for (int i = threadIdx.x; i < rot_heads * half; i += blockDim.x) {
    const int hl = i / half, p = i - hl * half, h = h0 + hl;
    const int a = interleave ? 2 * p : p, b = interleave ? 2 * p + 1 : p + half;
    const uint16_t* v = rowS + hl * head_size;
    const float c = cs[p], s = cs[half + p];
    const float xa = Conv<T>::to_f32(v[a]), xb = Conv<T>::to_f32(v[b]);
    float fa, fb;
    rope_pair(xa, xb, c, s, fa, fb);
    const uint16_t ra = Conv<T>::from_f32(fa), rb = Conv<T>::from_f32(fb);
    uint16_t* d = h < q_heads ? qdst + h * head_size : kdst + (h - q_heads) * head_size;
    d[a] = ra;
    d[b] = rb;
    if (h >= q_heads) {  // the qkv buffer keeps the rotated k as well, as the unfused op leaves it
      qdst[h * head_size + a] = ra;
      qdst[h * head_size + b] = rb;
    }
  }
  // pass-through dims (rot < head_size) of q and k
  if (rot < head_size) {
    const int pass = head_size - rot;
    for (int i = threadIdx.x; i < rot_heads * pass; i += blockDim.x) {
      const int hl = i / pass, p = rot + (i - hl * pass), h = h0 + hl;
      const uint16_t val = rowS[hl * head_size + p];
      qdst[h * head_size + p] = val;
      if (h >= q_heads) kdst[(h - q_heads) * head_size + p] = val;
    }
  }
  // v heads of this block
  const int v0 = max(h0, q_heads + kv_heads);
  for (int i = threadIdx.x; i < max(0, h1 - v0) * head_size; i += blockDim.x) {
    const int col = v0 * head_size + i;
    const uint16_t val = rowS[col - c0];
    vdst[col - (q_heads + kv_heads) * head_size] = val;
    qdst[col] = val;
  }
}

// ---------------------------------------------------------------------------------------------
// act_and_mul: out = act(x[:, :d]) * x[:, d:]; act rounded to T before the product
// (activation.cu:22-50 of the reference: silu returns T, then T*T)
// ---------------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
  if constexpr (ACT == 0) return v / (1.0f + __expf(-v));
  if constexpr (ACT == 1) {
    const float k = 0.7978845608028654f;  // sqrt(2/pi)
    return 0.5f * v * (1.0f + tanhf(k * (v + 0.044715f * v * v * v)));
  }
  return 0.5f * v * (1.0f + erff(v * 0.7071067811865476f));
}

template <typename T, int ACT, bool VEC>
__global__ void __launch_bounds__(256) act_and_mul_kernel(void* __restrict__ out, const void* __restrict__ xin,
                                                          int64_t rows, int d) {
  using R = typename Raw<T>::type;
  const R* x = reinterpret_cast<const R*>(xin);
  R* o = reinterpret_cast<R*>(out);
  if constexpr (VEC) {
    const int nv = d >> 3;
    const int64_t total = rows * nv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int64_t r = i / nv;
      const int c = (int)(i - r * nv);
      short8 a = reinterpret_cast<const short8*>(x + r * 2 * d)[c];
      short8 g = reinterpret_cast<const short8*>(x + r * 2 * d + d)[c];
      short8 ov;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float s = Conv<T>::to_f32(Conv<T>::from_f32(act_fn<ACT>(Conv<T>::to_f32((uint16_t)a[j]))));
        ov[j] = (short)Conv<T>::from_f32(s * Conv<T>::to_f32((uint16_t)g[j]));
      }
      reinterpret_cast<short8*>(o + r * d)[c] = ov;
    }
  } else {
    const int64_t total = rows * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int64_t r = i / d;
      const int c = (int)(i - r * d);
      const float s = Conv<T>::to_f32(Conv<T>::from_f32(act_fn<ACT>(Conv<T>::to_f32(x[r * 2 * d + c]))));
      o[r * d + c] = Conv<T>::from_f32(s * Conv<T>::to_f32(x[r * 2 * d + d + c]));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// rotary embedding (rope.cu:22-89 of the reference): cos_sin_cache[pos] = [cos(rot/2) | sin(rot/2)]
// one block per token; threads sweep (head, pair)
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void rope_rows(typename Raw<T>::type* base, int heads, int head_size, int half,
                                          const float* cs, int interleave) {
  const int per = heads * half;
  for (int i = threadIdx.x; i < per; i += blockDim.x) {
    const int h = i / half, p = i - h * half;
    const int a = interleave ? 2 * p : p, b = interleave ? 2 * p + 1 : p + half;
    typename Raw<T>::type* v = base + (int64_t)h * head_size;
    const float c = cs[p], s = cs[half + p];
    const float xa = Conv<T>::to_f32(v[a]), xb = Conv<T>::to_f32(v[b]);
    float fa, fb;
    rope_pair(xa, xb, c, s, fa, fb);
    v[a] = Conv<T>::from_f32(fa);
    v[b] = Conv<T>::from_f32(fb);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) rope_kernel(void* q, void* k, const float* __restrict__ cos_sin,
                                                   const int64_t* __restrict__ positions, int q_heads, int k_heads,
                                                   int head_size, int rot, int64_t q_stride_t, int64_t k_stride_t,
                                                   int interleave) {
  using R = typename Raw<T>::type;
  const int64_t t = blockIdx.x;
  const float* cs = cos_sin + positions[t] * rot;
  rope_rows<T>(reinterpret_cast<R*>(q) + t * q_stride_t, q_heads, head_size, rot / 2, cs, interleave);
  if (k) rope_rows<T>(reinterpret_cast<R*>(k) + t * k_stride_t, k_heads, head_size, rot / 2, cs, interleave);
}

// rope on the q and k slices of a fused qkv row, then scatter rotated k and v into pool rows loc[t]
template <typename T>
__global__ void __launch_bounds__(256) rope_store_kernel(void* qkv, const float* __restrict__ cos_sin,
                                                         const int64_t* __restrict__ positions, void* k_buffer,
                                                         void* v_buffer, const void* loc, int loc_is64, int q_heads,
                                                         int kv_heads, int head_size, int rot, int64_t kbuf_stride_n,
                                                         int64_t vbuf_stride_n, int interleave) {
  using R = typename Raw<T>::type;
  const int64_t t = blockIdx.x;
  const int64_t row = (int64_t)(q_heads + 2 * kv_heads) * head_size;
  R* base = reinterpret_cast<R*>(qkv) + t * row;
  const float* cs = cos_sin + positions[t] * rot;
  const int half = rot / 2;
  const int64_t slot = ld_idx(loc, loc_is64, t);
  R* kdst = reinterpret_cast<R*>(k_buffer) + slot * kbuf_stride_n;
  R* vdst = reinterpret_cast<R*>(v_buffer) + slot * vbuf_stride_n;
  // q heads and k heads are contiguous: rotate (q_heads + kv_heads) heads; k pairs also go to the pool
  const int nrot = (q_heads + kv_heads) * half;
  for (int i = threadIdx.x; i < nrot; i += 256) {
    const int h = i / half, p = i - h * half;
    const int a = interleave ? 2 * p : p, b = interleave ? 2 * p + 1 : p + half;
    R* v = base + (int64_t)h * head_size;
    const float c = cs[p], s = cs[half + p];
    const float xa = Conv<T>::to_f32(v[a]), xb = Conv<T>::to_f32(v[b]);
    float fa, fb;
    rope_pair(xa, xb, c, s, fa, fb);
    const R ra = Conv<T>::from_f32(fa), rb = Conv<T>::from_f32(fb);
    v[a] = ra;
    v[b] = rb;
    if (h >= q_heads) {
      R* kd = kdst + (int64_t)(h - q_heads) * head_size;
      kd[a] = ra;
      kd[b] = rb;
    }
  }
  // pass-through dims of k (rot < head_size) and all of v
  if (rot < head_size) {
    const int pass = head_size - rot;
    for (int i = threadIdx.x; i < kv_heads * pass; i += 256) {
      const int h = i / pass, p = rot + (i - h * pass);
      kdst[(int64_t)h * head_size + p] = base[(int64_t)(q_heads + h) * head_size + p];
    }
  }
  const R* vsrc = base + (int64_t)(q_heads + kv_heads) * head_size;
  const int nv = kv_heads * head_size;
  if ((nv % 8 == 0) && aligned16(vsrc) && aligned16(vdst)) {
    for (int i = threadIdx.x; i < (nv >> 3); i += 256)
      reinterpret_cast<short8*>(vdst)[i] = reinterpret_cast<const short8*>(vsrc)[i];
  } else {
    for (int i = threadIdx.x; i < nv; i += 256) vdst[i] = vsrc[i];
  }
}

// ---------------------------------------------------------------------------------------------
// KV pool row scatter (memory_pool.py:378-412): one block per token, 16-byte copies
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) set_kv_kernel(char* k_buffer, char* v_buffer, const char* k, const char* v,
                                                     const void* loc, int loc_is64, int64_t row_k_bytes,
                                                     int64_t row_v_bytes, int64_t kbuf_stride, int64_t vbuf_stride,
                                                     int64_t k_stride, int64_t v_stride, int vec) {
  const int64_t t = blockIdx.x;
  const int64_t slot = ld_idx(loc, loc_is64, t);
  char* kd = k_buffer + slot * kbuf_stride;
  char* vd = v_buffer + slot * vbuf_stride;
  const char* ks = k + t * k_stride;
  const char* vs = v + t * v_stride;
  if (vec) {
    for (int64_t i = threadIdx.x; i < (row_k_bytes >> 4); i += 256)
      reinterpret_cast<int4v*>(kd)[i] = reinterpret_cast<const int4v*>(ks)[i];
    for (int64_t i = threadIdx.x; i < (row_v_bytes >> 4); i += 256)
      reinterpret_cast<int4v*>(vd)[i] = reinterpret_cast<const int4v*>(vs)[i];
  } else {
    for (int64_t i = threadIdx.x; i < row_k_bytes; i += 256) kd[i] = ks[i];
    for (int64_t i = threadIdx.x; i < row_v_bytes; i += 256) vd[i] = vs[i];
  }
}

// The same scatter into an fp8 pool (memory_pool.py:389-400): optional division by a scale, evaluated as the product
// with the fp32 reciprocal and rounded to the source type (what `cache_k.div_(k_scale)` gives on the device), then
// the fp8 rounding of Tensor.to().  One block per token; a thread converts four neighbouring elements into one word
// when the row allows it.
template <typename T, int F>
__device__ __forceinline__ uint32_t kv8_from(const void* src, int64_t i, float inv, bool scaled) {
  float x = ld<T>(src, i);
  if (scaled) x = (float)Conv<T>::to_f32(Conv<T>::from_f32(__fmul_rn(x, inv)));
  return f32_to_kv8<F>(x);
}
template <typename T, int F>
__global__ void __launch_bounds__(256) set_kv_fp8_kernel(uint8_t* k_buffer, uint8_t* v_buffer, const void* k, const void* v,
                                                         const void* loc, int loc_is64, int row_k, int row_v,
                                                         int64_t kbuf_stride, int64_t vbuf_stride, int64_t k_stride,
                                                         int64_t v_stride, float k_inv, float v_inv, int k_scaled,
                                                         int v_scaled, int vec) {
  const int64_t t = blockIdx.x;
  const int64_t slot = ld_idx(loc, loc_is64, t);
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    uint8_t* dst = which ? v_buffer + slot * vbuf_stride : k_buffer + slot * kbuf_stride;
    const void* src = which ? v : k;
    const int64_t s0 = t * (which ? v_stride : k_stride);
    const int row = which ? row_v : row_k;
    const float inv = which ? v_inv : k_inv;
    const bool scaled = which ? v_scaled != 0 : k_scaled != 0;
    if (vec) {
      for (int i = threadIdx.x; i < (row >> 2); i += 256) {
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) w |= kv8_from<T, F>(src, s0 + 4 * i + e, inv, scaled) << (8 * e);
        reinterpret_cast<uint32_t*>(dst)[i] = w;
      }
    } else {
      for (int i = threadIdx.x; i < row; i += 256) dst[i] = (uint8_t)kv8_from<T, F>(src, s0 + i, inv, scaled);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// merge_state (merge_attn_states.cu:31-106 of the reference)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) merge_state_kernel(const void* __restrict__ v_a, const float* __restrict__ s_a,
                                                          const void* __restrict__ v_b, const float* __restrict__ s_b,
                                                          void* __restrict__ v_out, float* __restrict__ s_out,
                                                          int64_t nh, int d) {
  const int64_t total = nh * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / d;
    float pa = s_a[r], pb = s_b[r];
    pa = isinf(pa) ? -INFINITY : pa;
    pb = isinf(pb) ? -INFINITY : pb;
    const float mx = fmaxf(pa, pb);
    const float ea = expf(pa - mx), eb = expf(pb - mx);
    const float sum = ea + eb;
    const float wa = ea / sum, wb = eb / sum;
    st<T>(v_out, i, ld<T>(v_a, i) * wa + ld<T>(v_b, i) * wb);
    if (s_out != nullptr && i == r * d) s_out[r] = logf(sum) + mx;
  }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(256) act_and_mul_xs_kernel(void* __restrict__ out, const void* __restrict__ xin,
                                                             int64_t rows, int d, float* __restrict__ xsum_out,
                                                             int64_t xs_stride) {
  const uint16_t* x = reinterpret_cast<const uint16_t*>(xin);
  uint16_t* o = reinterpret_cast<uint16_t*>(out);
  const int nv = d >> 3;  // d % 128 == 0: every aligned group of 16 threads covers one 128-element block of one row
  const int64_t total = rows * nv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / nv;
    const int c = (int)(i - r * nv);
    const short8 a = reinterpret_cast<const short8*>(x + r * 2 * d)[c];
    const short8 g = reinterpret_cast<const short8*>(x + r * 2 * d + d)[c];
    short8 ov;
    float bs = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float sv = Conv<T>::to_f32(Conv<T>::from_f32(act_fn<ACT>(Conv<T>::to_f32((uint16_t)a[j]))));
      const uint16_t ob = Conv<T>::from_f32(sv * Conv<T>::to_f32((uint16_t)g[j]));
      ov[j] = (short)ob;
      bs += Conv<T>::to_f32(ob);
    }
    reinterpret_cast<short8*>(o + r * d)[c] = ov;
#pragma unroll
    for (int o2 = 8; o2 > 0; o2 >>= 1) bs += __shfl_xor(bs, o2, 16);
    if ((threadIdx.x & 15) == 0) xsum_out[(int64_t)(c >> 4) * xs_stride + r] = bs;
  }
}

// Greedy token: index of the first maximum of every row (torch.argmax, sampler.py greedy path).  One block per row,
// eight elements per load for 16-bit rows.  The scan is bound by vector instructions, so a thread's own elements (which
// it visits in ascending order) take one strict comparison each and NaNs are only noted; a row that holds one (a NaN
// wins, as in torch) is scanned again with the full rule.
template <typename T>
__global__ void __launch_bounds__(1024) argmax_rows_kernel(const void* __restrict__ x, int64_t* __restrict__ out, int cols,
                                                           int64_t row_stride) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int64_t row = blockIdx.x;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  bool any_nan = false;
  auto take = [&](float v, int i) {  // a NaN wins, as in torch; among equals (and among NaNs) the lowest index
    const bool vn = v != v, bn = best != best;
    if (vn ? (!bn || i < idx) : (!bn && (v > best || (v == best && i < idx)))) {
      best = v;
      idx = i;
    }
  };
  auto scan = [&](auto&& visit) {
    if constexpr (sizeof(typename Raw<T>::type) == 2) {
      const uint16_t* r = reinterpret_cast<const uint16_t*>(x) + row * row_stride;
      const int nv = ((row_stride % 8) == 0 && aligned16(x)) ? cols / 8 : 0;
      // four loads in flight per thread: a row of a large vocabulary is a few trips to memory instead of sixteen
      int v = threadIdx.x;
      for (; v + 3 * 1024 < nv; v += 4 * 1024) {
        short8 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = reinterpret_cast<const short8*>(r)[v + 1024 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) visit(Conv<T>::to_f32((uint16_t)q[u][j]), 8 * (v + 1024 * u) + j);
      }
      for (; v < nv; v += 1024) {
        const short8 q = reinterpret_cast<const short8*>(r)[v];
#pragma unroll
        for (int j = 0; j < 8; ++j) visit(Conv<T>::to_f32((uint16_t)q[j]), 8 * v + j);
      }
      for (int i = nv * 8 + threadIdx.x; i < cols; i += 1024) visit(Conv<T>::to_f32(r[i]), i);
    } else {
      for (int i = threadIdx.x; i < cols; i += 1024) visit(ld<T>(x, row * row_stride + i), i);
    }
  };
  // ascending indices within a thread: the first of equal values stays with a strict comparison (a row of nothing
  // but -inf keeps the sentinel and ends at index 0)
  scan([&](float v, int i) {
    any_nan |= v != v;
    if (v > best) {
      best = v;
      idx = i;
    }
  });
  if (__syncthreads_or(any_nan)) {  // block-uniform
    best = -INFINITY;
    idx = 0x7fffffff;
    scan(take);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    take(ov, oi);
  }
  if ((threadIdx.x & 63) == 0) {
    bv[threadIdx.x >> 6] = best;
    bi[threadIdx.x >> 6] = idx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) take(bv[w], bi[w]);
    out[row] = idx == 0x7fffffff ? 0 : idx;
  }
}

// Rotary step in front of the absorbed-MLA decode with fused rotation (triton_ops/rocm_mla_decode_rope.py:116-202):
// q_out[b, h] = q[b, h] with its rope part (elements lora .. lora + rope_dim) rotated at positions[b]; the rope part of
// the request's newest token (pool row kv_indices[kv_indptr[b + 1] - 1], stored unrotated by the caller) is rotated
// in place and copied to k_pe_out[b].  One block per request; a thread owns one (element, partner) pair.
template <typename T>
__global__ void __launch_bounds__(256) mla_rope_prepare_kernel(const void* __restrict__ q, void* __restrict__ q_out,
                                                               void* __restrict__ kv_buffer, void* __restrict__ k_pe_out,
                                                               const int32_t* __restrict__ kv_indptr,
                                                               const int32_t* __restrict__ kv_indices,
                                                               const float* __restrict__ cos_sin,
                                                               const int64_t* __restrict__ positions, int H, int lora,
                                                               int rope_dim, int rot_dim, int64_t q_stride_b,
                                                               int64_t q_stride_h, int64_t qo_stride_b, int64_t qo_stride_h,
                                                               int64_t kv_stride_n, int64_t kpe_stride_b,
                                                               int64_t cs_stride, int neox) {
  const int b = blockIdx.x;
  const int64_t pos = positions[b];
  const float* cs = cos_sin + pos * cs_stride;
  const int half = rot_dim / 2;
  // pair p < half: elements (p, p + half) in the neox layout, (2p, 2p + 1) interleaved; rows: heads then the key
  const int rows = H + 1;
  const int64_t last = (int64_t)kv_indptr[b + 1] - 1;
  const bool has_key = last >= kv_indptr[b];
  const int64_t loc = has_key ? kv_indices[last] : 0;
  for (int i = threadIdx.x; i < rows * half; i += blockDim.x) {
    const int r = i / half, p = i - r * half;
    const int e0 = neox ? p : 2 * p, e1 = neox ? p + half : 2 * p + 1;
    const float c = cs[p], sn = cs[half + p];
    if (r < H) {
      const int64_t src = (int64_t)b * q_stride_b + (int64_t)r * q_stride_h + lora;
      const int64_t dst = (int64_t)b * qo_stride_b + (int64_t)r * qo_stride_h + lora;
      const float x0 = ld<T>(q, src + e0), x1 = ld<T>(q, src + e1);
      st<T>(q_out, dst + e0, x0 * c - x1 * sn);
      st<T>(q_out, dst + e1, x1 * c + x0 * sn);
    } else if (has_key) {
      const int64_t src = loc * kv_stride_n + lora;
      const float x0 = ld<T>(kv_buffer, src + e0), x1 = ld<T>(kv_buffer, src + e1);
      const float y0 = x0 * c - x1 * sn, y1 = x1 * c + x0 * sn;
      st<T>(kv_buffer, src + e0, y0);
      st<T>(kv_buffer, src + e1, y1);
      st<T>(k_pe_out, (int64_t)b * kpe_stride_b + e0, y0);
      st<T>(k_pe_out, (int64_t)b * kpe_stride_b + e1, y1);
    }
  }
  // the rest: the latent part of q and rope elements past the rotary dimension, copied
  for (int i = threadIdx.x; i < H * (lora + rope_dim); i += blockDim.x) {
    const int h = i / (lora + rope_dim), e = i - h * (lora + rope_dim);
    if (e >= lora && e < lora + rot_dim) continue;
    st<T>(q_out, (int64_t)b * qo_stride_b + (int64_t)h * qo_stride_h + e,
          ld<T>(q, (int64_t)b * q_stride_b + (int64_t)h * q_stride_h + e));
  }
  if (has_key)
    for (int e = rot_dim + threadIdx.x; e < rope_dim; e += blockDim.x)
      st<T>(k_pe_out, (int64_t)b * kpe_stride_b + e, ld<T>(kv_buffer, loc * kv_stride_n + lora + e));
}

// Rotary step + cache write of the absorbed-MLA decode (models/deepseek_v2.py:1127-1193) in one launch.  Block
// (request b, part y): parts 0 .. gridDim.y - 2 deal the H heads' rope parts among themselves (q_pe -> rotated into
// q_out[..., lora ..]); the last part writes the latent row [k_nope | rotated k_pe] to pool row loc[b].
template <typename T>
__global__ void __launch_bounds__(256) mla_rope_store_kernel(const void* __restrict__ q_pe, void* __restrict__ q_out,
                                                             const void* __restrict__ latent, void* __restrict__ kv_buffer,
                                                             const void* __restrict__ loc, int loc_is64,
                                                             const float* __restrict__ cos_sin,
                                                             const int64_t* __restrict__ positions, int H, int lora,
                                                             int rope_dim, int rot_dim, int64_t qp_b, int64_t qp_h,
                                                             int64_t qo_b, int64_t qo_h, int64_t lat_b, int64_t kv_n,
                                                             int64_t cs_stride, int neox) {
  const int b = blockIdx.x;
  const float* cs = cos_sin + positions[b] * cs_stride;
  const int half = rot_dim / 2;
  if ((int)blockIdx.y + 1 < (int)gridDim.y) {
    const int parts = gridDim.y - 1;
    const int per = (H + parts - 1) / parts;
    const int h0 = blockIdx.y * per, h1 = min(h0 + per, H);
    for (int i = threadIdx.x; i < (h1 - h0) * half; i += blockDim.x) {
      const int h = h0 + i / half, p = i % half;
      const int e0 = neox ? p : 2 * p, e1 = neox ? p + half : 2 * p + 1;
      const float c = cs[p], sn = cs[half + p];
      const int64_t src = (int64_t)b * qp_b + (int64_t)h * qp_h, dst = (int64_t)b * qo_b + (int64_t)h * qo_h + lora;
      float y0, y1;
      rope_pair(ld<T>(q_pe, src + e0), ld<T>(q_pe, src + e1), c, sn, y0, y1);
      st<T>(q_out, dst + e0, y0);
      st<T>(q_out, dst + e1, y1);
    }
    for (int i = threadIdx.x; i < (h1 - h0) * (rope_dim - rot_dim); i += blockDim.x) {
      const int h = h0 + i / (rope_dim - rot_dim), e = rot_dim + i % (rope_dim - rot_dim);
      st<T>(q_out, (int64_t)b * qo_b + (int64_t)h * qo_h + lora + e, ld<T>(q_pe, (int64_t)b * qp_b + (int64_t)h * qp_h + e));
    }
  } else {
    const int64_t row = loc_is64 ? reinterpret_cast<const int64_t*>(loc)[b] : (int64_t) reinterpret_cast<const int32_t*>(loc)[b];
    const int64_t src = (int64_t)b * lat_b, dst = row * kv_n;
    for (int e = threadIdx.x; e < lora; e += blockDim.x) st<T>(kv_buffer, dst + e, ld<T>(latent, src + e));
    for (int p = threadIdx.x; p < half; p += blockDim.x) {
      const int e0 = neox ? p : 2 * p, e1 = neox ? p + half : 2 * p + 1;
      const float c = cs[p], sn = cs[half + p];
      float y0, y1;
      rope_pair(ld<T>(latent, src + lora + e0), ld<T>(latent, src + lora + e1), c, sn, y0, y1);
      st<T>(kv_buffer, dst + lora + e0, y0);
      st<T>(kv_buffer, dst + lora + e1, y1);
    }
    for (int e = rot_dim + threadIdx.x; e < rope_dim; e += blockDim.x)
      st<T>(kv_buffer, dst + lora + e, ld<T>(latent, src + lora + e));
  }
}

}  // namespace sglk

using namespace sglk;

extern "C" {

int sglk_rmsnorm(void* out, const void* x, const void* weight, int64_t rows, int hidden, int64_t x_stride,
                 int64_t out_stride, float eps, int gemma, int dtype, void* stream) {
  SGLK_CHECK(out && x && weight, "rmsnorm: null pointer");
  SGLK_CHECK(rows >= 0 && hidden > 0, "rmsnorm: bad shape rows=%lld hidden=%d", (long long)rows, hidden);
  SGLK_DISPATCH_FLOAT(dtype, T,
                      return launch_rmsnorm<T>(out, x, nullptr, weight, rows, hidden, x_stride, out_stride, eps, gemma,
                                               (hipStream_t)stream));
  return SGLK_OK;
}

int sglk_fused_add_rmsnorm(void* x, void* residual, const void* weight, int64_t rows, int hidden, float eps, int gemma,
                           int dtype, void* stream) {
  SGLK_CHECK(x && residual && weight, "fused_add_rmsnorm: null pointer");
  SGLK_CHECK(rows >= 0 && hidden > 0, "fused_add_rmsnorm: bad shape");
  SGLK_DISPATCH_FLOAT(dtype, T,
                      return launch_rmsnorm<T>(x, x, residual, weight, rows, hidden, hidden, hidden, eps, gemma,
                                               (hipStream_t)stream));
  return SGLK_OK;
}

int sglk_act_and_mul(void* out, const void* x, int64_t rows, int d, int act, int dtype, void* stream) {
  SGLK_CHECK(out && x, "act_and_mul: null pointer");
  SGLK_CHECK(rows >= 0 && d > 0, "act_and_mul: bad shape");
  SGLK_CHECK(act >= 0 && act <= 2, "act_and_mul: unknown activation %d", act);
  // the reference requires 16-byte aligned rows (python/sgl_kernel/elementwise.py:172-173)
  SGLK_CHECK(((int64_t)d * elem_size(dtype)) % 16 == 0, "The pointers must be multiple of 16 bytes.");
  if (rows == 0) return SGLK_OK;
  const bool vec = dtype != SGLK_F32 && aligned16(out) && aligned16(x);
  const int64_t work = vec ? rows * (d >> 3) : rows * d;
  int blocks = (int)((work + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t s = (hipStream_t)stream;
#define SGLK_ACT_LAUNCH(ACT)                                                                                      \
  if (vec)                                                                                                        \
    hipLaunchKernelGGL((act_and_mul_kernel<T, ACT, true>), dim3(blocks), dim3(256), 0, s, out, x, rows, d);       \
  else                                                                                                            \
    hipLaunchKernelGGL((act_and_mul_kernel<T, ACT, false>), dim3(blocks), dim3(256), 0, s, out, x, rows, d);
  SGLK_DISPATCH_FLOAT(dtype, T, {
    if (act == 0) {
      SGLK_ACT_LAUNCH(0)
    } else if (act == 1) {
      SGLK_ACT_LAUNCH(1)
    } else {
      SGLK_ACT_LAUNCH(2)
    }
  });
#undef SGLK_ACT_LAUNCH
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_rope(void* q, void* k, const float* cos_sin_cache, const int64_t* positions, int64_t tokens, int q_heads,
              int k_heads, int head_size, int rot_dim, int64_t q_stride_t, int64_t k_stride_t, int interleave,
              int dtype, void* stream) {
  SGLK_CHECK(q && cos_sin_cache && positions, "rope: null pointer");
  SGLK_CHECK(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, "rope: bad rotary dim %d (head_size %d)",
             rot_dim, head_size);
  if (tokens == 0) return SGLK_OK;
  SGLK_DISPATCH_FLOAT(dtype, T,
                      hipLaunchKernelGGL((rope_kernel<T>), dim3((unsigned)tokens), dim3(256), 0, (hipStream_t)stream, q,
                                         k, cos_sin_cache, positions, q_heads, k_heads, head_size, rot_dim, q_stride_t,
                                         k_stride_t, interleave));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_rope_and_store_kv(void* qkv, const float* cos_sin_cache, const int64_t* positions, void* k_buffer,
                           void* v_buffer, const void* loc, int loc_is64, int64_t tokens, int q_heads, int kv_heads,
                           int head_size, int rot_dim, int64_t kbuf_stride_n, int64_t vbuf_stride_n, int interleave,
                           int dtype, void* stream) {
  SGLK_CHECK(qkv && cos_sin_cache && positions && k_buffer && v_buffer && loc, "rope_and_store_kv: null pointer");
  SGLK_CHECK(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, "rope_and_store_kv: bad rotary dim");
  if (tokens == 0) return SGLK_OK;
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((rope_store_kernel<T>), dim3((unsigned)tokens), dim3(256), 0,
                                        (hipStream_t)stream, qkv, cos_sin_cache, positions, k_buffer, v_buffer, loc,
                                        loc_is64, q_heads, kv_heads, head_size, rot_dim, kbuf_stride_n, vbuf_stride_n,
                                        interleave));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_set_kv_buffer(void* k_buffer, void* v_buffer, const void* k, const void* v, const void* loc, int loc_is64,
                       int64_t tokens, int row_k, int row_v, int64_t kbuf_stride_n, int64_t vbuf_stride_n,
                       int64_t k_stride_t, int64_t v_stride_t, int elem_bytes, void* stream) {
  SGLK_CHECK(k_buffer && v_buffer && k && v && loc, "set_kv_buffer: null pointer");
  SGLK_CHECK(elem_bytes == 1 || elem_bytes == 2 || elem_bytes == 4, "set_kv_buffer: bad element size");
  if (tokens == 0) return SGLK_OK;
  const int64_t rk = (int64_t)row_k * elem_bytes, rv = (int64_t)row_v * elem_bytes;
  const int64_t kb = kbuf_stride_n * elem_bytes, vb = vbuf_stride_n * elem_bytes, ks = k_stride_t * elem_bytes,
                vs = v_stride_t * elem_bytes;
  const int vec = (rk % 16 == 0) && (rv % 16 == 0) && (kb % 16 == 0) && (vb % 16 == 0) && (ks % 16 == 0) &&
                  (vs % 16 == 0) && aligned16(k_buffer) && aligned16(v_buffer) && aligned16(k) && aligned16(v);
  hipLaunchKernelGGL(set_kv_kernel, dim3((unsigned)tokens), dim3(256), 0, (hipStream_t)stream, (char*)k_buffer,
                     (char*)v_buffer, (const char*)k, (const char*)v, loc, loc_is64, rk, rv, kb, vb, ks, vs, vec);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_set_kv_buffer_fp8(void* k_buffer, void* v_buffer, const void* k, const void* v, const void* loc, int loc_is64,
                           int64_t tokens, int row_k, int row_v, int64_t kbuf_stride_n, int64_t vbuf_stride_n,
                           int64_t k_stride_t, int64_t v_stride_t, float k_scale, float v_scale, int src_dtype,
                           int kv_format, void* stream) {
  SGLK_CHECK(k_buffer && v_buffer && k && v && loc, "set_kv_buffer_fp8: null pointer");
  SGLK_CHECK(tokens >= 0 && row_k > 0 && row_v > 0, "set_kv_buffer_fp8: bad shape");
  SGLK_CHECK(kbuf_stride_n >= row_k && vbuf_stride_n >= row_v && k_stride_t >= row_k && v_stride_t >= row_v,
             "set_kv_buffer_fp8: a stride is shorter than its row");
  if (tokens == 0) return SGLK_OK;
  const int vec = row_k % 4 == 0 && row_v % 4 == 0 && kbuf_stride_n % 4 == 0 && vbuf_stride_n % 4 == 0 &&
                  (uintptr_t)k_buffer % 4 == 0 && (uintptr_t)v_buffer % 4 == 0;
  const int ks = k_scale > 0.f, vs = v_scale > 0.f;
  const float ki = ks ? 1.0f / k_scale : 1.0f, vi = vs ? 1.0f / v_scale : 1.0f;
#define SGLK_SETKV8(F)                                                                                                 \
  hipLaunchKernelGGL((set_kv_fp8_kernel<T, F>), dim3((unsigned)tokens), dim3(256), 0, (hipStream_t)stream,              \
                     (uint8_t*)k_buffer, (uint8_t*)v_buffer, k, v, loc, loc_is64, row_k, row_v, kbuf_stride_n,          \
                     vbuf_stride_n, k_stride_t, v_stride_t, ki, vi, ks, vs, vec)
  SGLK_DISPATCH_FLOAT(src_dtype, T, {
    switch (kv_format) {
      case SGLK_KV_E4M3FN: SGLK_SETKV8(SGLK_KV_E4M3FN); break;
      case SGLK_KV_E5M2: SGLK_SETKV8(SGLK_KV_E5M2); break;
      case SGLK_KV_E5M2FNUZ: SGLK_SETKV8(SGLK_KV_E5M2FNUZ); break;
      default: return fail(SGLK_EINVAL, "set_kv_buffer_fp8: unknown KV pool format %d", kv_format);
    }
  });
#undef SGLK_SETKV8
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_merge_state(const void* v_a, const float* s_a, const void* v_b, const float* s_b, void* v_out, float* s_out,
                     int64_t n, int h, int d, int dtype, void* stream) {
  SGLK_CHECK(v_a && s_a && v_b && s_b && v_out, "merge_state: null pointer");
  const int64_t nh = n * h;
  if (nh == 0) return SGLK_OK;
  int blocks = (int)((nh * d + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  SGLK_DISPATCH_FLOAT(dtype, T,
                      hipLaunchKernelGGL((merge_state_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, v_a,
                                         s_a, v_b, s_b, v_out, s_out, nh, d));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_fused_add_rmsnorm_ex(void* x, void* residual, const void* weight, int64_t rows, int hidden, float eps, int gemma,
                              int dtype, const float* partials, int splits, int64_t part_stride, const void* bias,
                              float* xsum_out, int64_t xs_stride, float* zero_out, int64_t zero_count, void* stream) {
  SGLK_CHECK(zero_count >= 0 && (zero_count == 0 || zero_out), "fused_add_rmsnorm_ex: bad zero_out");
  SGLK_CHECK(x && weight, "fused_add_rmsnorm_ex: null pointer");
  SGLK_CHECK(rows >= 0 && hidden > 0 && hidden % 128 == 0 && hidden <= 8192,
             "fused_add_rmsnorm_ex: hidden=%d must be a multiple of 128, at most 8192", hidden);
  SGLK_CHECK(dtype == SGLK_F16 || dtype == SGLK_BF16, "fused_add_rmsnorm_ex: 16-bit types only");
  SGLK_CHECK(aligned16(x) && (!residual || aligned16(residual)) && aligned16(weight) && (!partials || aligned16(partials)),
             "fused_add_rmsnorm_ex: pointers must be 16-byte aligned");
  SGLK_CHECK(!partials || (splits >= 1 && part_stride >= rows * hidden && part_stride % 4 == 0),
             "fused_add_rmsnorm_ex: bad partial layout");
  SGLK_CHECK(!xsum_out || xs_stride >= rows, "fused_add_rmsnorm_ex: xs_stride smaller than the row count");
  if (rows == 0) return SGLK_OK;
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((rmsnorm_fused_kernel<T>), dim3((unsigned)rows),
                                        dim3(std::min(1024, std::max(256, (hidden / 8 + 63) / 64 * 64))), 0,
                                        (hipStream_t)stream,
                                        x, residual, weight, hidden, eps, gemma, partials, splits, part_stride, bias,
                                        xsum_out, xs_stride, zero_out, zero_count));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_act_and_mul_xs(void* out, const void* x, int64_t rows, int d, int act, int dtype, float* xsum_out,
                        int64_t xs_stride, void* stream) {
  SGLK_CHECK(out && x && xsum_out, "act_and_mul_xs: null pointer");
  SGLK_CHECK(rows >= 0 && d > 0 && d % 128 == 0, "act_and_mul_xs: d=%d must be a multiple of 128", d);
  SGLK_CHECK(act >= 0 && act <= 2, "act_and_mul_xs: unknown activation %d", act);
  SGLK_CHECK(dtype == SGLK_F16 || dtype == SGLK_BF16, "act_and_mul_xs: 16-bit types only");
  SGLK_CHECK(aligned16(out) && aligned16(x) && xs_stride >= rows, "act_and_mul_xs: alignment / stride");
  if (rows == 0) return SGLK_OK;
  int blocks = (int)((rows * (d >> 3) + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t s = (hipStream_t)stream;
#define SGLK_ACTX(ACT) \
  hipLaunchKernelGGL((act_and_mul_xs_kernel<T, ACT>), dim3(blocks), dim3(256), 0, s, out, x, rows, d, xsum_out, xs_stride)
  SGLK_DISPATCH_HALF(dtype, T, {
    if (act == 0) {
      SGLK_ACTX(0);
    } else if (act == 1) {
      SGLK_ACTX(1);
    } else {
      SGLK_ACTX(2);
    }
  });
#undef SGLK_ACTX
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_rope_and_store_kv_partial_scaled(void* qkv, const float* partials, int splits, int64_t part_stride,
                                          const void* bias, const float* row_scale, const float* col_scale,
                                          const float* cos_sin_cache, const int64_t* positions, void* k_buffer,
                                          void* v_buffer, const void* loc, int loc_is64, int64_t tokens, int q_heads,
                                          int kv_heads, int head_size, int rot_dim, int64_t kbuf_stride_n,
                                          int64_t vbuf_stride_n, int interleave, int dtype, void* stream) {
  SGLK_CHECK(qkv && partials, "rope_and_store_kv_partial: null pointer");
  SGLK_CHECK((row_scale == nullptr) == (col_scale == nullptr), "rope_and_store_kv_partial: both scale vectors or neither");
  const int64_t row_len = (int64_t)(q_heads + 2 * kv_heads) * head_size;
  SGLK_CHECK(splits >= 1 && row_len % 8 == 0 && part_stride >= tokens * row_len && part_stride % 4 == 0 &&
                 aligned16(partials) && aligned16(qkv),
             "rope_and_store_kv_partial: bad partial layout");
  SGLK_CHECK(dtype == SGLK_F16 || dtype == SGLK_BF16, "rope_and_store_kv_partial: 16-bit types only");
  if (tokens == 0) return SGLK_OK;
  SGLK_CHECK(cos_sin_cache && positions && k_buffer && v_buffer && loc, "rope_and_store_kv_partial: null pointer");
  SGLK_CHECK(rot_dim > 0 && rot_dim % 2 == 0 && rot_dim <= head_size, "rope_and_store_kv_partial: bad rotary dim");
  SGLK_CHECK(head_size % 8 == 0, "rope_and_store_kv_partial: head size must be a multiple of 8");
  // enough blocks to cover the CUs: a token's heads are spread over `parts` blocks
  const int total_heads = q_heads + 2 * kv_heads;
  const int parts = (int)std::max<int64_t>(1, std::min<int64_t>(total_heads, 256 / tokens));
  const int hpb = (total_heads + parts - 1) / parts;
  const int64_t cols = (int64_t)hpb * head_size;
  SGLK_CHECK(cols * 2 <= 64 * 1024, "rope_and_store_kv_partial: %lld elements per block do not fit in LDS", (long long)cols);
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((rope_store_partial_kernel<T>), dim3((unsigned)tokens, (unsigned)((total_heads + hpb - 1) / hpb)),
                                        dim3((unsigned)std::min<int64_t>(1024, std::max<int64_t>(64, (cols / 8 + 63) / 64 * 64))),
                                        (size_t)cols * 2, (hipStream_t)stream, qkv, partials, splits, part_stride, bias,
                                        cos_sin_cache, positions, k_buffer, v_buffer, loc, loc_is64, q_heads, kv_heads,
                                        head_size, rot_dim, kbuf_stride_n, vbuf_stride_n, interleave, hpb, row_scale,
                                        col_scale));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_rope_and_store_kv_partial(void* qkv, const float* partials, int splits, int64_t part_stride, const void* bias,
                                   const float* cos_sin_cache, const int64_t* positions, void* k_buffer, void* v_buffer,
                                   const void* loc, int loc_is64, int64_t tokens, int q_heads, int kv_heads,
                                   int head_size, int rot_dim, int64_t kbuf_stride_n, int64_t vbuf_stride_n,
                                   int interleave, int dtype, void* stream) {
  return sglk_rope_and_store_kv_partial_scaled(qkv, partials, splits, part_stride, bias, nullptr, nullptr, cos_sin_cache,
                                               positions, k_buffer, v_buffer, loc, loc_is64, tokens, q_heads, kv_heads,
                                               head_size, rot_dim, kbuf_stride_n, vbuf_stride_n, interleave, dtype, stream);
}

int sglk_argmax(const void* x, int64_t* out, int64_t rows, int cols, int64_t row_stride, int dtype, void* stream) {
  SGLK_CHECK(rows >= 0 && cols > 0 && row_stride >= cols, "argmax: bad shape");
  if (rows == 0) return SGLK_OK;
  SGLK_CHECK(x && out, "argmax: null pointer");
  SGLK_DISPATCH_FLOAT(dtype, T,
                      hipLaunchKernelGGL((argmax_rows_kernel<T>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream,
                                         x, out, cols, row_stride));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_mla_rope_and_store(const void* q_pe, void* q_out, const void* latent, void* kv_buffer, const void* loc,
                            int loc_is64, const float* cos_sin_cache, const int64_t* positions, int B, int H,
                            int kv_lora_rank, int rope_dim, int rotary_dim, int64_t qpe_stride_b, int64_t qpe_stride_h,
                            int64_t qo_stride_b, int64_t qo_stride_h, int64_t latent_stride_b, int64_t kv_stride_n,
                            int64_t cos_sin_stride, int is_neox, int dtype, void* stream) {
  SGLK_CHECK(B >= 0 && H > 0 && kv_lora_rank >= 0 && rope_dim > 0, "mla_rope_and_store: bad shape");
  SGLK_CHECK(rotary_dim > 0 && rotary_dim % 2 == 0 && rotary_dim <= rope_dim, "mla_rope_and_store: bad rotary dim");
  SGLK_CHECK(qpe_stride_h >= rope_dim && qo_stride_h >= kv_lora_rank + rope_dim && latent_stride_b >= kv_lora_rank + rope_dim &&
                 kv_stride_n >= kv_lora_rank + rope_dim && cos_sin_stride >= rotary_dim,
             "mla_rope_and_store: bad strides");
  if (B == 0) return SGLK_OK;
  SGLK_CHECK(q_pe && q_out && latent && kv_buffer && loc && cos_sin_cache && positions, "mla_rope_and_store: null pointer");
  const int parts = std::max(1, std::min(H, H * (rotary_dim / 2) / 1024));  // about four pairs a thread
  SGLK_DISPATCH_FLOAT(dtype, T,
                      hipLaunchKernelGGL((mla_rope_store_kernel<T>), dim3(B, parts + 1), dim3(256), 0, (hipStream_t)stream,
                                         q_pe, q_out, latent, kv_buffer, loc, loc_is64, cos_sin_cache, positions, H,
                                         kv_lora_rank, rope_dim, rotary_dim, qpe_stride_b, qpe_stride_h, qo_stride_b,
                                         qo_stride_h, latent_stride_b, kv_stride_n, cos_sin_stride, is_neox));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_mla_rope_prepare(const void* q, void* q_out, void* kv_buffer, void* k_pe_out, const int32_t* kv_indptr,
                          const int32_t* kv_indices, const float* cos_sin_cache, const int64_t* positions, int B, int H,
                          int kv_lora_rank, int rope_dim, int rotary_dim, int64_t q_stride_b, int64_t q_stride_h,
                          int64_t qo_stride_b, int64_t qo_stride_h, int64_t kv_stride_n, int64_t kpe_stride_b,
                          int64_t cos_sin_stride, int is_neox, int dtype, void* stream) {
  SGLK_CHECK(B >= 0 && H > 0 && kv_lora_rank >= 0 && rope_dim > 0, "mla_rope_prepare: bad shape");
  SGLK_CHECK(rotary_dim > 0 && rotary_dim % 2 == 0 && rotary_dim <= rope_dim, "mla_rope_prepare: bad rotary dim");
  SGLK_CHECK(cos_sin_stride >= rotary_dim && kpe_stride_b >= rope_dim && kv_stride_n >= kv_lora_rank + rope_dim,
             "mla_rope_prepare: bad strides");
  if (B == 0) return SGLK_OK;
  SGLK_CHECK(q && q_out && kv_buffer && k_pe_out && kv_indptr && kv_indices && cos_sin_cache && positions,
             "mla_rope_prepare: null pointer");
  SGLK_DISPATCH_FLOAT(dtype, T,
                      hipLaunchKernelGGL((mla_rope_prepare_kernel<T>), dim3(B), dim3(256), 0, (hipStream_t)stream, q, q_out,
                                         kv_buffer, k_pe_out, kv_indptr, kv_indices, cos_sin_cache, positions, H,
                                         kv_lora_rank, rope_dim, rotary_dim, q_stride_b, q_stride_h, qo_stride_b,
                                         qo_stride_h, kv_stride_n, kpe_stride_b, cos_sin_stride, is_neox ? 1 : 0));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}
}
