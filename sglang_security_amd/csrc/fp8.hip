// fp8 (OCP e4m3fn) activation quantisation and scaled matmuls on the gfx950 fp8 matrix instructions.
//
//   sglk_per_token_quant_fp8        <- sgl-kernel/csrc/gemm/per_token_quant_fp8.cu:10-106
//   sglk_per_token_group_quant_fp8  <- sgl-kernel/csrc/gemm/per_token_group_quant_8bit.cu:19-197
//   sglk_per_tensor_quant_fp8       <- sgl-kernel/csrc/gemm/per_tensor_quant_fp8.cu:10-122
//   sglk_fp8_scaled_mm              <- sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1071-1146
//   sglk_fp8_blockwise_scaled_mm    <- sgl-kernel/csrc/gemm/fp8_blockwise_gemm_kernel.cu:345-423
//
// Quantisation is bit-exact integer-valued work (one correctly rounded f32 op per step, then the
// hardware round-to-nearest-even f32 -> e4m3 conversion of a value already clamped to +-448).
// The matmuls accumulate in fp32 on v_mfma_f32_16x16x32_fp8_fp8.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace sglk {
namespace {

constexpr float kFp8Max = 448.0f;

template <typename T>
__device__ __forceinline__ void load8(const void* base, int64_t i, float (&v)[8]) {
  if constexpr (sizeof(typename Raw<T>::type) == 2) {
    const uint4 raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + i);
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = Conv<T>::to_f32((uint16_t)(w[j] & 0xffffu));
      v[2 * j + 1] = Conv<T>::to_f32((uint16_t)(w[j] >> 16));
    }
  } else {
    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + i);
    const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + i + 4);
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
  }
}

__device__ __forceinline__ float clamp_sym(float v, float lo, float hi) { return fmaxf(fminf(v, hi), lo); }

// two floats -> two e4m3 bytes in the low half of the result
__device__ __forceinline__ uint32_t cvt2(float a, float b) {
  return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}
__device__ __forceinline__ uint8_t cvt1(float a) { return (uint8_t)(cvt2(a, 0.f) & 0xff); }

__device__ __forceinline__ uint2 cvt8(const float (&v)[8]) {
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
  return make_uint2((uint32_t)lo, (uint32_t)hi);
}

__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return v;
}

// ---- per token: one block per row ------------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void per_token_quant_kernel(const void* __restrict__ x, uint8_t* __restrict__ q,
                                                              float* __restrict__ scales, int hidden) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const int64_t base = row * hidden;
  float amax = 0.f;
  if constexpr (VEC) {
    for (int i = threadIdx.x * 8; i < hidden; i += 256 * 8) {
      float v[8];
      load8<T>(x, base + i, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
    }
  } else {
    for (int i = threadIdx.x; i < hidden; i += 256) amax = fmaxf(amax, fabsf(ld<T>(x, base + i)));
  }
  amax = block_max_256(amax, red);
  const float scale = amax / kFp8Max;
  if (threadIdx.x == 0) scales[row] = scale;
  const float inv = 1.0f / scale;
  if constexpr (VEC) {
    for (int i = threadIdx.x * 8; i < hidden; i += 256 * 8) {
      float v[8];
      load8<T>(x, base + i, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = clamp_sym(v[j] * inv, -kFp8Max, kFp8Max);
      *reinterpret_cast<uint2*>(q + base + i) = cvt8(v);
    }
  } else {
    for (int i = threadIdx.x; i < hidden; i += 256)
      q[base + i] = cvt1(clamp_sym(ld<T>(x, base + i) * inv, -kFp8Max, kFp8Max));
  }
}

// ---- producers that quantise on the way out: the 16-bit tensor the plain op sequence would write in between is
// formed (and rounded) in registers only.  One block of 1024 threads per row, up to four 8-element vectors a thread.
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) t += red[w];
  __syncthreads();
  return t;
}
__device__ __forceinline__ float block_max_1024(float v, float* red) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) t = fmaxf(t, red[w]);
  __syncthreads();
  return t;
}

// per (token, 128-element group) scales, as sgl_per_token_group_quant_fp8 forms them (scale = max(eps, amax) / 448,
// q = clamp(y / scale)): a group is sixteen consecutive 8-element vectors, i.e. sixteen consecutive lanes of a wave
// (vector v of thread t: v = t + 1024 it, and 1024 is a multiple of 16), all of them live when the row length is a
// multiple of 128
constexpr float kGroupEps = 1e-10f;
__device__ __forceinline__ void group_quant_store(const float (&y)[8], uint8_t* q, float* scales, int64_t row, int len,
                                                  int v) {
  float amax = kGroupEps;
#pragma unroll
  for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(y[j]));
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 16));
  const float ys = amax / kFp8Max;
  if ((v & 15) == 0) scales[row * (len >> 7) + (v >> 4)] = ys;
  float o8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o8[j] = clamp_sym(y[j] / ys, -kFp8Max, kFp8Max);
  *reinterpret_cast<uint2*>(q + row * len + 8 * v) = cvt8(o8);
}

// residual += x (when given), y = round(rmsnorm(residual) * w), then per-token quantisation of y:
// fused_add_rmsnorm (norm.cu) followed by sgl_per_token_quant_fp8 (per_token_quant_fp8.cu) with the same roundings
// GROUP: per (token, 128-group) scales instead (sgl_per_token_group_quant_fp8), scales [rows, hidden / 128]
template <typename T, bool GROUP = false>
__global__ __launch_bounds__(1024) void rmsnorm_quant_kernel(const void* __restrict__ xin, void* __restrict__ residual,
                                                             const void* __restrict__ w, uint8_t* __restrict__ q,
                                                             float* __restrict__ scales, int hidden, float eps, int gemma,
                                                             const float* __restrict__ partials, int splits,
                                                             int64_t part_stride, const float* __restrict__ row_scale,
                                                             const float* __restrict__ col_scale,
                                                             const void* __restrict__ bias, int64_t x_stride) {
  // partials: the input row arrives as the raw split sums of the preceding fp8 product, partials[s][row][:], and is
  // formed here as that product's own split sum forms it: round_T(sum * row_scale[row] * col_scale[n] + bias[n]).
  // row_scale may be the very array `scales` (the product's activation scales, replaced by this kernel's own): it is
  // read before the block's first barrier and written after its last.
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const float rsc = (partials && row_scale) ? row_scale[row] : 1.f;
  const uint16_t* bsrc = reinterpret_cast<const uint16_t*>(bias);
  const uint16_t* x = reinterpret_cast<const uint16_t*>(xin) + row * x_stride;
  uint16_t* res = residual ? reinterpret_cast<uint16_t*>(residual) + row * hidden : nullptr;
  const uint16_t* wr = reinterpret_cast<const uint16_t*>(w);
  const int nvec = hidden >> 3;
  constexpr int MAXV = 4;
  float keep[MAXV][8];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int v = threadIdx.x + it * 1024;
    if (v < nvec) {
      if (partials) {
        float acc[8];
        sum_partials(partials, splits, part_stride, row * hidden + 8 * v, acc);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float bj = bsrc ? Conv<T>::to_f32(bsrc[8 * v + j]) : 0.f;
          const float val = col_scale ? scaled_sum(acc[j], rsc, col_scale[8 * v + j], bj) : acc[j] + bj;
          keep[it][j] = Conv<T>::to_f32(Conv<T>::from_f32(val));  // the value the product would have stored
        }
      } else {
        const short8 xv = reinterpret_cast<const short8*>(x)[v];
#pragma unroll
        for (int j = 0; j < 8; ++j) keep[it][j] = Conv<T>::to_f32((uint16_t)xv[j]);
      }
      if (res) {
        const short8 rv = reinterpret_cast<const short8*>(res)[v];
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
  const float inv = rsqrtf(block_sum_1024(ss, red) / (float)hidden + eps);
  float amax = 0.f;
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int v = threadIdx.x + it * 1024;
    if (v < nvec) {
      const short8 wv = reinterpret_cast<const short8*>(wr)[v];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float wf = Conv<T>::to_f32((uint16_t)wv[j]);
        if (gemma) wf += 1.0f;
        keep[it][j] = Conv<T>::to_f32(Conv<T>::from_f32((keep[it][j] * inv) * wf));  // the value rmsnorm stores
        amax = fmaxf(amax, fabsf(keep[it][j]));
      }
      if constexpr (GROUP) group_quant_store(keep[it], q, scales, row, hidden, v);
    }
  }
  if constexpr (GROUP) return;
  amax = block_max_1024(amax, red);
  const float scale = amax / kFp8Max;
  if (threadIdx.x == 0) scales[row] = scale;
  const float sinv = 1.0f / scale;
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int v = threadIdx.x + it * 1024;
    if (v < nvec) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = clamp_sym(keep[it][j] * sinv, -kFp8Max, kFp8Max);
      *reinterpret_cast<uint2*>(q + row * hidden + 8 * v) = cvt8(o);
    }
  }
}

// y = round(round(silu(gate)) * up) per element of x = [gate | up], then per-token quantisation of y:
// silu_and_mul (activation.cu:22-50) followed by sgl_per_token_quant_fp8
template <typename T, bool GROUP = false>
__global__ __launch_bounds__(1024) void silu_mul_quant_kernel(const void* __restrict__ xin, uint8_t* __restrict__ q,
                                                              float* __restrict__ scales, int d,
                                                              const float* __restrict__ partials, int splits,
                                                              int64_t part_stride, const float* __restrict__ row_scale,
                                                              const float* __restrict__ col_scale) {
  // partials: x arrives as the raw split sums of the gate_up fp8 product, partials[s][row][2 d], and is formed as that
  // product's own split sum forms it (round_T(sum * row_scale[row] * col_scale[n])).  row_scale may be the array
  // `scales`: read before the block's barrier, written after it.
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const uint16_t* x = reinterpret_cast<const uint16_t*>(xin) + row * 2 * d;
  const float rsc = (partials && row_scale) ? row_scale[row] : 1.f;
  const int nvec = d >> 3;
  constexpr int MAXV = 4;
  float keep[MAXV][8];
  float amax = 0.f;
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int v = threadIdx.x + it * 1024;
    if (v < nvec) {
      float ga[8], ua[8];
      if (partials) {
        sum_partials(partials, splits, part_stride, row * 2 * d + 8 * v, ga);
        sum_partials(partials, splits, part_stride, row * 2 * d + d + 8 * v, ua);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float cg = col_scale ? col_scale[8 * v + j] : 1.f, cu = col_scale ? col_scale[d + 8 * v + j] : 1.f;
          ga[j] = Conv<T>::to_f32(Conv<T>::from_f32(scaled_sum(ga[j], rsc, cg, 0.f)));  // what the product would store
          ua[j] = Conv<T>::to_f32(Conv<T>::from_f32(scaled_sum(ua[j], rsc, cu, 0.f)));
        }
      } else {
        const short8 a = reinterpret_cast<const short8*>(x)[v];
        const short8 g = reinterpret_cast<const short8*>(x + d)[v];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ga[j] = Conv<T>::to_f32((uint16_t)a[j]);
          ua[j] = Conv<T>::to_f32((uint16_t)g[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float av = ga[j];
        const float sv = Conv<T>::to_f32(Conv<T>::from_f32(av / (1.0f + __expf(-av))));
        keep[it][j] = Conv<T>::to_f32(Conv<T>::from_f32(sv * ua[j]));
        amax = fmaxf(amax, fabsf(keep[it][j]));
      }
      if constexpr (GROUP) group_quant_store(keep[it], q, scales, row, d, v);
    }
  }
  if constexpr (GROUP) return;
  amax = block_max_1024(amax, red);
  const float scale = amax / kFp8Max;
  if (threadIdx.x == 0) scales[row] = scale;
  const float sinv = 1.0f / scale;
#pragma unroll
  for (int it = 0; it < MAXV; ++it) {
    const int v = threadIdx.x + it * 1024;
    if (v < nvec) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = clamp_sym(keep[it][j] * sinv, -kFp8Max, kFp8Max);
      *reinterpret_cast<uint2*>(q + row * d + 8 * v) = cvt8(o);
    }
  }
}

// ---- per (token, group): 16 lanes per group ------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void per_group_quant_kernel(const void* __restrict__ x, uint8_t* __restrict__ q,
                                                              float* __restrict__ scales, int64_t rows, int hidden,
                                                              int group, float eps, float fp8_min, float fp8_max,
                                                              int col_major) {
  const int ng = hidden / group;
  const int64_t total = rows * ng;
  const int sub = threadIdx.x & 15;
  const int64_t gid = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = gid < total;
  const int64_t g = live ? gid : total - 1;  // every lane stays in the shuffles
  const int64_t base = g * group;            // groups tile the row-major tensor contiguously
  float amax = eps;
  if constexpr (VEC) {
    for (int i = sub * 8; i < group; i += 128) {
      float v[8];
      load8<T>(x, base + i, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
    }
  } else {
    for (int i = sub; i < group; i += 16) amax = fmaxf(amax, fabsf(ld<T>(x, base + i)));
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 16));
  const float ys = amax / fp8_max;
  if (!live) return;
  if (sub == 0) {
    const int64_t r = g / ng, c = g % ng;
    scales[col_major ? c * rows + r : g] = ys;
  }
  if constexpr (VEC) {
    for (int i = sub * 8; i < group; i += 128) {
      float v[8];
      load8<T>(x, base + i, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = clamp_sym(v[j] / ys, fp8_min, fp8_max);
      *reinterpret_cast<uint2*>(q + base + i) = cvt8(v);
    }
  } else {
    for (int i = sub; i < group; i += 16) q[base + i] = cvt1(clamp_sym(ld<T>(x, base + i) / ys, fp8_min, fp8_max));
  }
}

// ---- per tensor ----------------------------------------------------------------------------------------
// x is a [rows, cols] view (row pitch xs, the columns consecutive), q likewise (row pitch qs); VEC: eight columns a
// thread and step (cols, both pitches and both base addresses multiples of eight elements).
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void per_tensor_absmax_kernel(const void* __restrict__ x, float* __restrict__ scale,
                                                                int64_t rows, int64_t cols, int64_t xs) {
  __shared__ float red[4];
  float amax = 0.f;
  const int64_t per = VEC ? cols / 8 : cols, n = rows * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / per, c = i - r * per;
    if constexpr (VEC) {
      float v[8];
      load8<T>(x, r * xs + 8 * c, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
    } else {
      amax = fmaxf(amax, fabsf(ld<T>(x, r * xs + c)));
    }
  }
  amax = block_max_256(amax, red);
  // non-negative floats order like their bit patterns
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<int*>(scale), __float_as_int(amax / kFp8Max));
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void per_tensor_quant_kernel(const void* __restrict__ x, uint8_t* __restrict__ q,
                                                               const float* __restrict__ scale, int64_t rows,
                                                               int64_t cols, int64_t xs, int64_t qs) {
  const float inv = 1.0f / scale[0];
  const int64_t per = VEC ? cols / 8 : cols, n = rows * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / per, c = i - r * per;
    if constexpr (VEC) {
      float v[8];
      load8<T>(x, r * xs + 8 * c, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = clamp_sym(v[j] * inv, -kFp8Max, kFp8Max);
      *reinterpret_cast<uint2*>(q + r * qs + 8 * c) = cvt8(v);
    } else {
      q[r * qs + c] = cvt1(clamp_sym(ld<T>(x, r * xs + c) * inv, -kFp8Max, kFp8Max));
    }
  }
}

// ---- scaled matmul -------------------------------------------------------------------------------------
//
// Block = 4 waves on one (16*MT rows) x (32 cols) output tile; the waves split K in 128-byte blocks
// (wave w takes blocks w, w+4, ...) and their fp32 partials are summed through LDS.  Operands go
// straight from global memory to the MFMA registers: inside a 128-byte K block lane group g = lane>>4
// owns bytes [32g, 32g+32) of its row for BOTH operands, so any fixed assignment of those bytes to the
// four MFMA k-slots gives the same dot product.  BLOCKWISE applies a_scale[m,kb] * b_scale[n/128,kb]
// per K block; otherwise the epilogue applies scales_a[m] * scales_b[n] (+ bias).
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MmArgs {
  const uint8_t* a;
  const uint8_t* b;
  const float* sa;
  const float* sb;
  const void* bias;
  void* out;
  int M, N, K;
  int64_t a_stride, b_stride, out_stride;
  int64_t sa_m, sa_k, sb_n, sb_k;
  int64_t a_batch, b_batch, out_batch;  // element strides between the matrices of a batch (blockIdx.z)
  int scalar_scales;                    // one scale per operand for the whole batch (bmm_fp8)
  float* partial;                       // split-K form: blockIdx.z is the K split, raw fp32 sums go to partial[z][M][N]
  int kb_per_split;
};

// operand formats of the 16x16x32 MFMA: e4m3 x e4m3, e5m2 x e4m3, e4m3 x e5m2
template <int FMT>
__device__ __forceinline__ f32x4 mma_f8(long a, long b, f32x4 c) {
  if constexpr (FMT == 1)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf8_fp8(a, b, c, 0, 0, 0);
  else if constexpr (FMT == 2)
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(a, b, c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void load32(const uint8_t* row, int k, int K, long (&r)[4]) {
  // K % 16 == 0: 16-byte chunks are all-or-nothing
  int4v lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
  if (k + 16 <= K) lo = *reinterpret_cast<const int4v*>(row + k);
  if (k + 32 <= K) hi = *reinterpret_cast<const int4v*>(row + k + 16);
  r[0] = (long)(uint32_t)lo.x | ((long)lo.y << 32);
  r[1] = (long)(uint32_t)lo.z | ((long)lo.w << 32);
  r[2] = (long)(uint32_t)hi.x | ((long)hi.y << 32);
  r[3] = (long)(uint32_t)hi.z | ((long)hi.w << 32);
}

// NW waves split the block's K range (4, or 8 where the grid leaves CUs without a block)
template <typename TO, int MT, bool BLOCKWISE, int FMT = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW) void fp8_mm_kernel(MmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float red_raw[];
  float(*red)[MT * 2][64][4] = reinterpret_cast<float(*)[MT * 2][64][4]>(red_raw);  // [NW][MT * 2][64][4]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 16 * MT;
  const uint8_t* arow[MT];
  int mrow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    mrow[i] = min(m0 + 16 * i + l16, p.M - 1);
    arow[i] = p.a + (int64_t)blockIdx.z * p.a_batch + (int64_t)mrow[i] * p.a_stride;
  }
  const uint8_t* brow[2];
  int ncol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    ncol[j] = min(n0 + 16 * j + l16, p.N - 1);
    brow[j] = p.b + (int64_t)blockIdx.z * p.b_batch + (int64_t)ncol[j] * p.b_stride;
  }
  f32x4 tot[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int kblocks = (p.K + 127) / 128, kb_first = w;
  if (p.partial) {  // block-uniform: this block's K split
    kb_first = blockIdx.z * p.kb_per_split + w;
    kblocks = min(kblocks, (int)(blockIdx.z + 1) * p.kb_per_split);
  }
  auto body = [&](int kb) {
    const int k = kb * 128 + 32 * g;
    float sc[MT][2][4];
    if constexpr (BLOCKWISE) {
      // the block's scales are requested before the operands, one product per output element
      float bs[2], as[MT][4];
#pragma unroll
      for (int j = 0; j < 2; ++j) bs[j] = p.sb[(int64_t)(ncol[j] / 128) * p.sb_n + (int64_t)kb * p.sb_k];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          as[i][r] = p.sa[(int64_t)min(m0 + 16 * i + 4 * g + r, p.M - 1) * p.sa_m + (int64_t)kb * p.sa_k];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[i][j][r] = as[i][r] * bs[j];
    }
    long af[MT][4], bf[2][4];
#pragma unroll
    for (int i = 0; i < MT; ++i) load32(arow[i], k, p.K, af[i]);
#pragma unroll
    for (int j = 0; j < 2; ++j) load32(brow[j], k, p.K, bf[j]);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 acc = BLOCKWISE ? f32x4{0.f, 0.f, 0.f, 0.f} : tot[i][j];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mma_f8<FMT>(af[i][s], bf[j][s], acc);
        if constexpr (BLOCKWISE) {
          // C layout: column = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
          for (int r = 0; r < 4; ++r) tot[i][j][r] += acc[r] * sc[i][j][r];
        } else {
          tot[i][j] = acc;
        }
      }
  };
  for (int kb = kb_first; kb < kblocks; kb += NW) body(kb);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *reinterpret_cast<f32x4*>(&red[w][i * 2 + j][lane][0]) = tot[i][j];
  __syncthreads();
  // wave w finishes tiles w, w+4, ...
  for (int t = w; t < MT * 2; t += NW) {
    const int i = t >> 1, j = t & 1;
    f32x4 s = *reinterpret_cast<f32x4*>(&red[0][t][lane][0]);
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) {
      const f32x4 o = *reinterpret_cast<f32x4*>(&red[ww][t][lane][0]);
      s += o;
    }
    const int n = n0 + 16 * j + l16;
    if (n >= p.N) continue;
    if (p.partial) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * i + 4 * g + r;
        if (m < p.M) p.partial[((int64_t)blockIdx.z * p.M + m) * p.N + n] = s[r];
      }
      continue;
    }
    float sbn = 1.f, bn = 0.f;
    if constexpr (!BLOCKWISE) {
      sbn = p.sb[p.scalar_scales ? 0 : n];
      if (p.bias) bn = ld<TO>(p.bias, n);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 16 * i + 4 * g + r;
      if (m >= p.M) continue;
      float v = s[r];
      if constexpr (!BLOCKWISE) v = v * p.sa[p.scalar_scales ? 0 : m] * sbn + bn;
      st<TO>(p.out, (int64_t)blockIdx.z * p.out_batch + (int64_t)m * p.out_stride + n, v);
    }
  }
}

// out[m][n] = (sum over splits, in split order, of partial[z][m][n]) * sa[m] * sb[n] (+ bias); sa / sb optional
template <typename TO>
__global__ __launch_bounds__(256) void fp8_splitk_reduce_kernel(const float* __restrict__ partial, int splits,
                                                               const float* __restrict__ sa, const float* __restrict__ sb,
                                                               const void* __restrict__ bias, void* __restrict__ out, int M,
                                                               int N, int64_t out_stride) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int nv = N / 4;
  if (idx >= (int64_t)M * nv) return;
  const int m = (int)(idx / nv), n = (int)(idx - (int64_t)m * nv) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int z0 = 0; z0 < splits; z0 += 4) {  // four splits' loads in flight
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = *reinterpret_cast<const f32x4*>(partial + ((int64_t)min(z0 + u, splits - 1) * M + m) * N + n);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (z0 + u < splits) acc += v[u];
  }
  const float am = sa ? sa[m] : 1.f;  // no scales: the partials of the block-scaled form are final sums
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float bn = bias ? ld<TO>(bias, n + j) : 0.f;
    st<TO>(out, (int64_t)m * out_stride + n + j, scaled_sum(acc[j], am, sb ? sb[n + j] : 1.f, bn));
  }
}

// K splits of the two-launch form: narrow outputs leave most of the chip idle otherwise
static int fp8_splits(int M, int N, int K) {
  const char* env = getenv("SGLK_FP8_SPLITS");
  const int kblocks = (K + 127) / 128;
  if (env && atoi(env) > 0) return std::min(atoi(env), kblocks);
  if (N % 4 != 0) return 1;
  const int blocks = (N + 31) / 32 * ((M + 63) / 64);
  if (blocks >= 200) return 1;
  return std::max(1, std::min({(512 + blocks - 1) / blocks, kblocks / 8, 16}));
}

template <typename TO, bool BLOCKWISE, int FMT = 0>
int launch_mm(const MmArgs& p, hipStream_t s, int batch = 1) {
  const dim3 block(256);
  const unsigned nb = (p.N + 31) / 32, z = (unsigned)batch;
  auto lds = [](int mt, int nw) { return (size_t)nw * mt * 2 * 64 * 4 * sizeof(float); };
  if (p.M <= 16) {
    hipLaunchKernelGGL((fp8_mm_kernel<TO, 1, BLOCKWISE, FMT>), dim3(nb, (p.M + 15) / 16, z), block, lds(1, 4), s, p);
  } else if (p.M <= 32) {
    hipLaunchKernelGGL((fp8_mm_kernel<TO, 2, BLOCKWISE, FMT>), dim3(nb, (p.M + 31) / 32, z), block, lds(2, 4), s, p);
  } else {
    const unsigned mb = (p.M + 63) / 64;
    const char* env = getenv("SGLK_FP8_WAVES");  // 4 / 8: forces the K-waves per block (tests, timing)
    const int kblocks = (p.K + 127) / 128;
    const bool eight = env ? atoi(env) == 8 : (nb * mb * z <= 256 && kblocks >= 32);
    if (eight && FMT == 0)
      hipLaunchKernelGGL((fp8_mm_kernel<TO, 4, BLOCKWISE, 0, 8>), dim3(nb, mb, z), dim3(512), lds(4, 8), s, p);
    else
      hipLaunchKernelGGL((fp8_mm_kernel<TO, 4, BLOCKWISE, FMT>), dim3(nb, mb, z), block, lds(4, 4), s, p);
  }
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}


// ---- streaming form of the per-token / per-channel scaled product at decode batch sizes --------------------------
// out[M, N] = (x[M, K] . W[N, K]^T) * sa[m] * sb[n] (+ bias) with the weight re-laid-out once at load time
// (sglk_fp8_repack) so that every wave instruction reads 1 KiB of consecutive memory:
//   packed = [N / 16 strips][K / 128 k-blocks][2 halves][64 lanes][16 bytes],
//   piece (strip, kb, h, lane = 16 g + i16) = W[16 strip + i16][128 kb + 32 g + 16 h .. + 16]
// (inside a 128-byte k-block lane group g owns bytes [32 g, 32 g + 32) of its row for both operands, as in the
// kernel above).  A wave streams its strip straight into the matrix-core operand registers, PF k-blocks ahead; the
// quantised activations, which every wave re-reads, go through an LDS tile (64 rows x 128 bytes per k-block, 16-byte
// chunks XOR-swizzled by (row >> 1) & 7: the 64 lanes of an operand read fall on every bank group four times, the
// minimum).  Narrow outputs split K over blockIdx.y and leave raw fp32 sums in partial[split][M][N], written through.
__device__ __forceinline__ int fs_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }
__device__ __forceinline__ long lo64(int4v v) { return (long)(uint32_t)v.x | ((long)v.y << 32); }
__device__ __forceinline__ long hi64(int4v v) { return (long)(uint32_t)v.z | ((long)v.w << 32); }

// BW: the block-scaled form (DeepSeek): sa[m, kb] (strides sa_m, sa_k) and sb[n / 128, kb] (strides sb_n, sb_k) scale
// every 128-k block's product before it is added up; the activation scales of a k-block travel with its tile, the
// weight scale with the weight pieces.
template <typename TO, int NS, int PF, bool BW = false, int XPF = PF>
__global__ void __launch_bounds__(256, 2) fp8_stream_kernel(const uint8_t* __restrict__ x, const uint8_t* __restrict__ wp,
                                                            const float* __restrict__ sa, const float* __restrict__ sb,
                                                            const void* __restrict__ bias, void* __restrict__ out,
                                                            float* __restrict__ partial, int M, int N, int K,
                                                            int kb_per_split, int64_t x_stride, int64_t out_stride,
                                                            int64_t sa_m = 0, int64_t sa_k = 0, int64_t sb_n = 0,
                                                            int64_t sb_k = 0) {
  __shared__ __attribute__((aligned(16))) char xs[2 * 64 * 128];
  __shared__ __attribute__((aligned(16))) float sas[2][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int kblocks = K / 128;
  const int m0 = blockIdx.z * 64;
  const int kb0 = blockIdx.y * kb_per_split;
  const int nkb = min(kb_per_split, kblocks - kb0);
  const int n0 = (blockIdx.x * 4 + wid) * 16 * NS + i16;

  // activations: thread (row xr, quarter xq) moves bytes [32 xq, 32 xq + 32) of its row; a k-block's bytes are
  // requested XPF blocks ahead and parked in LDS one block ahead (a step of the loop is shorter than the way to the
  // L2 and back: a request made only a step before its use would set the pace)
  const int xr = threadIdx.x >> 2, xq = threadIdx.x & 3;
  // rows past M re-read row M - 1 (their products are never stored): no exec-masked branch in the loop
  const int64_t xrow = min(m0 + xr, M - 1);
  const uint8_t* xsrc = x + xrow * x_stride + 32 * xq;
  int4v xreg[XPF][2];
  [[maybe_unused]] float sareg[XPF] = {};
  auto x_fetch = [&](int4v(&r2)[2], int kb, [[maybe_unused]] float& sr) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
      r2[c] = *reinterpret_cast<const int4v*>(xsrc + 128 * (int64_t)kb + 16 * c);
    if constexpr (BW) sr = sa[xrow * sa_m + (int64_t)kb * sa_k];  // every thread of a row: no branch
  };
  auto x_park = [&](const int4v(&r2)[2], int buf, [[maybe_unused]] float sr) {
#pragma unroll
    for (int c = 0; c < 2; ++c) *reinterpret_cast<int4v*>(xs + buf * 8192 + fs_off(xr, 2 * xq + c)) = r2[c];
    if constexpr (BW)
      if (xq == 0) sas[buf][xr] = sr;
  };

  const uint8_t* wsrc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int strip = min((n0 + 16 * s) >> 4, (N >> 4) - 1);
    wsrc[s] = wp + ((int64_t)strip * kblocks + kb0) * 2048 + 16 * lane;
  }
  auto wload = [](const uint8_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const int4v*>(p)); };
  int4v wq[PF][NS][2];
  [[maybe_unused]] float wsc[PF][NS];
  [[maybe_unused]] const float* sbrow[NS];
  if constexpr (BW) {
#pragma unroll
    for (int s = 0; s < NS; ++s) sbrow[s] = sb + (int64_t)(min(n0 + 16 * s, N - 1) >> 7) * sb_n + (int64_t)kb0 * sb_k;
  }
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (p < nkb) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int h = 0; h < 2; ++h) wq[p % PF][s][h] = wload(wsrc[s] + 2048 * p + 1024 * h);
        if constexpr (BW) wsc[p % PF][s] = sbrow[s][(int64_t)p * sb_k];
      }
    }

  f32x4 oacc[NS][4];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) oacc[s][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int U = PF > XPF ? PF : XPF;  // the loop is unrolled so that register slots are compile-time
  static_assert(PF % 2 == 0 && XPF % 2 == 0 && U % PF == 0 && U % XPF == 0,
                "the LDS tiles alternate with the parity of the k-block");
#pragma unroll
  for (int p = 0; p < XPF; ++p)
    if (p < nkb) x_fetch(xreg[p], kb0 + p, sareg[p]);
  x_park(xreg[0], 0, sareg[0]);
  __syncthreads();

  for (int base = 0; base < nkb; base += U) {
#pragma unroll
    for (int p = 0; p < U; ++p) {  // block j: weight slot j % PF = p % PF, activation slot p % XPF
      const int j = base + p;
      if (j < nkb) {
        const int buf = j & 1;
        // block j's own registers were parked a step ago
        if (j + XPF < nkb) x_fetch(xreg[p % XPF], kb0 + j + XPF, sareg[p % XPF]);
        if constexpr (BW) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int4v af0 = *reinterpret_cast<const int4v*>(xs + buf * 8192 + fs_off(16 * mt + i16, 2 * g));
            const int4v af1 = *reinterpret_cast<const int4v*>(xs + buf * 8192 + fs_off(16 * mt + i16, 2 * g + 1));
            const f32x4 as4 = *reinterpret_cast<const f32x4*>(&sas[buf][16 * mt + 4 * g]);  // rows 16 mt + 4 g + r
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
              acc = mma_f8<0>(lo64(af0), lo64(wq[p % PF][s][0]), acc);
              acc = mma_f8<0>(hi64(af0), hi64(wq[p % PF][s][0]), acc);
              acc = mma_f8<0>(lo64(af1), lo64(wq[p % PF][s][1]), acc);
              acc = mma_f8<0>(hi64(af1), hi64(wq[p % PF][s][1]), acc);
#pragma unroll
              for (int r = 0; r < 4; ++r) oacc[s][mt][r] += acc[r] * (as4[r] * wsc[p % PF][s]);
            }
          }
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
              const int4v af = *reinterpret_cast<const int4v*>(xs + buf * 8192 + fs_off(16 * mt + i16, 2 * g + h));
#pragma unroll
              for (int s = 0; s < NS; ++s) {
                oacc[s][mt] = mma_f8<0>(lo64(af), lo64(wq[p % PF][s][h]), oacc[s][mt]);
                oacc[s][mt] = mma_f8<0>(hi64(af), hi64(wq[p % PF][s][h]), oacc[s][mt]);
              }
            }
        }
        if (j + PF < nkb) {
#pragma unroll
          for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int h = 0; h < 2; ++h) wq[p % PF][s][h] = wload(wsrc[s] + 2048 * (int64_t)(j + PF) + 1024 * h);
            if constexpr (BW) wsc[p % PF][s] = sbrow[s][(int64_t)(j + PF) * sb_k];
          }
        }
        if (j + 1 < nkb) x_park(xreg[(p + 1) % XPF], buf ^ 1, sareg[(p + 1) % XPF]);
        __syncthreads();
      }
    }
  }

  // lane (i16, g) holds rows m0 + 16 mt + 4 g + r of column n0 + 16 s
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int n = n0 + 16 * s;
    if (n >= N) continue;
    const float sbn = (partial || BW) ? 1.f : sb[n];
    const float bn = (!partial && bias) ? ld<TO>(bias, n) : 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * mt + 4 * g + r;
        if (m >= M) continue;
        if (partial)
          __hip_atomic_store(&partial[((int64_t)blockIdx.y * M + m) * N + n], oacc[s][mt][r], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        else if constexpr (BW)
          st<TO>(out, (int64_t)m * out_stride + n, oacc[s][mt][r]);
        else
          st<TO>(out, (int64_t)m * out_stride + n, oacc[s][mt][r] * sa[m] * sbn + bn);
      }
  }
}

// row-major e4m3 [N, K] -> the streaming layout above
__global__ void __launch_bounds__(256) fp8_repack_kernel(const uint8_t* __restrict__ w, uint8_t* __restrict__ packed, int N,
                                                         int K, int64_t w_stride) {
  const int64_t piece = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int kblocks = K / 128;
  if (piece >= (int64_t)(N / 16) * kblocks * 128) return;
  const int lane = (int)(piece & 63), h = (int)((piece >> 6) & 1);
  const int64_t sk = piece >> 7;
  const int kb = (int)(sk % kblocks), strip = (int)(sk / kblocks);
  const int i16 = lane & 15, g = lane >> 4;
  *reinterpret_cast<int4v*>(packed + piece * 16) =
      *reinterpret_cast<const int4v*>(w + (int64_t)(16 * strip + i16) * w_stride + 128 * kb + 32 * g + 16 * h);
}

// K splits of the streaming form (measured on the 8B shapes, one strip per wave): outputs of 256 blocks or more run
// unsplit (no reduction launch); narrower ones split until about 512 blocks, at most 8 ways, at least 1024 of k each
static int fp8_stream_splits(int M, int N, int K) {
  const char* env = getenv("SGLK_FP8_STREAM_SPLITS");
  const int kblocks = K / 128;
  if (env && atoi(env) > 0) return std::min(atoi(env), kblocks);
  if (N % 4 != 0) return 1;
  const int blocks = (N + 63) / 64 * ((M + 63) / 64);
  if (blocks >= 256) return 1;
  return std::max(1, std::min({(512 + blocks - 1) / blocks, std::max(K / 1024, 1), 8}));
}

}  // namespace
}  // namespace sglk

using namespace sglk;

extern "C" {

int sglk_per_token_quant_fp8(const void* x, void* q, float* scales, int64_t rows, int hidden, int dtype,
                             void* stream) {
  SGLK_CHECK(x && q && scales, "per_token_quant_fp8: null pointer");
  SGLK_CHECK(rows >= 0 && hidden > 0, "per_token_quant_fp8: bad shape rows=%lld hidden=%d", (long long)rows, hidden);
  if (rows == 0) return SGLK_OK;
  const bool vec = hidden % 8 == 0 && aligned16(x) && ((uintptr_t)q & 7) == 0;
  hipStream_t s = (hipStream_t)stream;
  SGLK_DISPATCH_FLOAT(dtype, T, {
    if (vec)
      hipLaunchKernelGGL((per_token_quant_kernel<T, true>), dim3((unsigned)rows), dim3(256), 0, s, x, (uint8_t*)q, scales,
                         hidden);
    else
      hipLaunchKernelGGL((per_token_quant_kernel<T, false>), dim3((unsigned)rows), dim3(256), 0, s, x, (uint8_t*)q,
                         scales, hidden);
  });
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_per_token_group_quant_fp8(const void* x, void* q, float* scales, int64_t rows, int hidden, int group_size,
                                   float eps, float fp8_min, float fp8_max, int scale_col_major, int dtype,
                                   void* stream) {
  SGLK_CHECK(x && q && scales, "per_token_group_quant_fp8: null pointer");
  SGLK_CHECK(rows >= 0 && hidden > 0 && group_size > 0, "per_token_group_quant_fp8: bad shape");
  SGLK_CHECK(hidden % group_size == 0, "per_token_group_quant_fp8: hidden %d not a multiple of group_size %d", hidden,
             group_size);
  SGLK_CHECK(fp8_max > 0.f && fp8_min < 0.f, "per_token_group_quant_fp8: bad clamp range");
  if (rows == 0) return SGLK_OK;
  const int64_t groups = rows * (hidden / group_size);
  const bool vec = group_size % 8 == 0 && aligned16(x) && ((uintptr_t)q & 7) == 0;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((groups + 15) / 16));
  SGLK_DISPATCH_FLOAT(dtype, T, {
    if (vec)
      hipLaunchKernelGGL((per_group_quant_kernel<T, true>), grid, dim3(256), 0, s, x, (uint8_t*)q, scales, rows, hidden,
                         group_size, eps, fp8_min, fp8_max, scale_col_major);
    else
      hipLaunchKernelGGL((per_group_quant_kernel<T, false>), grid, dim3(256), 0, s, x, (uint8_t*)q, scales, rows, hidden,
                         group_size, eps, fp8_min, fp8_max, scale_col_major);
  });
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_rmsnorm_quant_fp8(const void* x, void* residual, const void* weight, void* q, float* scales, int64_t rows,
                           int hidden, float eps, int gemma, int dtype, void* stream) {
  SGLK_CHECK(x && weight && q && scales, "rmsnorm_quant_fp8: null pointer");
  SGLK_CHECK(rows >= 0 && hidden > 0, "rmsnorm_quant_fp8: bad shape");
  SGLK_CHECK(hidden % 8 == 0 && hidden <= 4 * 1024 * 8, "rmsnorm_quant_fp8: hidden=%d must be a multiple of 8, at most 32768",
             hidden);
  SGLK_CHECK(aligned16(x) && aligned16(weight) && (!residual || aligned16(residual)) && ((uintptr_t)q & 7) == 0,
             "rmsnorm_quant_fp8: tensors must be 16-byte aligned");
  if (rows == 0) return SGLK_OK;
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((rmsnorm_quant_kernel<T>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream,
                                        x, residual, weight, (uint8_t*)q, scales, hidden, eps, gemma,
                                        (const float*)nullptr, 0, (int64_t)0, (const float*)nullptr, (const float*)nullptr,
                                        (const void*)nullptr, (int64_t)hidden));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_rmsnorm_group_quant_fp8(const void* x, void* residual, const void* weight, void* q, float* scales, int64_t rows,
                                 int hidden, int64_t x_row_stride, float eps, int gemma, int dtype, void* stream) {
  SGLK_CHECK(x && weight && q && scales, "rmsnorm_group_quant_fp8: null pointer");
  SGLK_CHECK(rows >= 0 && hidden > 0, "rmsnorm_group_quant_fp8: bad shape");
  SGLK_CHECK(hidden % 128 == 0 && hidden <= 4 * 1024 * 8,
             "rmsnorm_group_quant_fp8: hidden=%d must be a multiple of 128, at most 32768", hidden);
  SGLK_CHECK(x_row_stride >= hidden && x_row_stride % 8 == 0, "rmsnorm_group_quant_fp8: bad row pitch");
  SGLK_CHECK(aligned16(x) && aligned16(weight) && (!residual || aligned16(residual)) && ((uintptr_t)q & 7) == 0,
             "rmsnorm_group_quant_fp8: tensors must be 16-byte aligned");
  if (rows == 0) return SGLK_OK;
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((rmsnorm_quant_kernel<T, true>), dim3((unsigned)rows), dim3(1024), 0,
                                        (hipStream_t)stream, x, residual, weight, (uint8_t*)q, scales, hidden, eps, gemma,
                                        (const float*)nullptr, 0, (int64_t)0, (const float*)nullptr, (const float*)nullptr,
                                        (const void*)nullptr, x_row_stride));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_silu_and_mul_group_quant_fp8(const void* x, void* q, float* scales, int64_t rows, int d, int dtype, void* stream) {
  SGLK_CHECK(x && q && scales, "silu_and_mul_group_quant_fp8: null pointer");
  SGLK_CHECK(rows >= 0 && d > 0 && d % 128 == 0 && d <= 4 * 1024 * 8,
             "silu_and_mul_group_quant_fp8: d=%d must be a multiple of 128, at most 32768", d);
  SGLK_CHECK(aligned16(x) && ((uintptr_t)q & 7) == 0, "silu_and_mul_group_quant_fp8: tensors must be 16-byte aligned");
  if (rows == 0) return SGLK_OK;
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((silu_mul_quant_kernel<T, true>), dim3((unsigned)rows), dim3(1024), 0,
                                        (hipStream_t)stream, x, (uint8_t*)q, scales, d, (const float*)nullptr, 0, (int64_t)0,
                                        (const float*)nullptr, (const float*)nullptr));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_rmsnorm_quant_fp8_partial(const float* partials, int splits, int64_t part_stride, const float* row_scale,
                                   const float* col_scale, const void* bias, void* residual, const void* weight, void* q,
                                   float* scales, int64_t rows, int hidden, float eps, int gemma, int dtype,
                                   void* stream) {
  SGLK_CHECK(partials && weight && q && scales, "rmsnorm_quant_fp8_partial: null pointer");
  SGLK_CHECK((row_scale == nullptr) == (col_scale == nullptr), "rmsnorm_quant_fp8_partial: both scale vectors or neither");
  SGLK_CHECK(rows >= 0 && hidden > 0 && hidden % 8 == 0 && hidden <= 4 * 1024 * 8,
             "rmsnorm_quant_fp8_partial: hidden=%d must be a multiple of 8, at most 32768", hidden);
  SGLK_CHECK(splits >= 1 && part_stride >= rows * hidden && part_stride % 4 == 0 && aligned16(partials),
             "rmsnorm_quant_fp8_partial: bad partial layout");
  SGLK_CHECK(aligned16(weight) && (!residual || aligned16(residual)) && ((uintptr_t)q & 7) == 0,
             "rmsnorm_quant_fp8_partial: tensors must be 16-byte aligned");
  if (rows == 0) return SGLK_OK;
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((rmsnorm_quant_kernel<T>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream,
                                        (const void*)nullptr, residual, weight, (uint8_t*)q, scales, hidden, eps, gemma,
                                        partials, splits, part_stride, row_scale, col_scale, bias, (int64_t)hidden));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_silu_and_mul_quant_fp8(const void* x, void* q, float* scales, int64_t rows, int d, int dtype, void* stream) {
  SGLK_CHECK(x && q && scales, "silu_and_mul_quant_fp8: null pointer");
  SGLK_CHECK(rows >= 0 && d > 0, "silu_and_mul_quant_fp8: bad shape");
  SGLK_CHECK(d % 8 == 0 && d <= 4 * 1024 * 8, "silu_and_mul_quant_fp8: d=%d must be a multiple of 8, at most 32768", d);
  SGLK_CHECK(aligned16(x) && ((uintptr_t)q & 7) == 0, "silu_and_mul_quant_fp8: tensors must be 16-byte aligned");
  if (rows == 0) return SGLK_OK;
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((silu_mul_quant_kernel<T>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream,
                                        x, (uint8_t*)q, scales, d, (const float*)nullptr, 0, (int64_t)0,
                                        (const float*)nullptr, (const float*)nullptr));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_silu_and_mul_quant_fp8_partial(const float* partials, int splits, int64_t part_stride, const float* row_scale,
                                        const float* col_scale, void* q, float* scales, int64_t rows, int d, int dtype,
                                        void* stream) {
  SGLK_CHECK(partials && q && scales, "silu_and_mul_quant_fp8_partial: null pointer");
  SGLK_CHECK((row_scale == nullptr) == (col_scale == nullptr), "silu_and_mul_quant_fp8_partial: both scale vectors or neither");
  SGLK_CHECK(rows >= 0 && d > 0 && d % 8 == 0 && d <= 4 * 1024 * 8,
             "silu_and_mul_quant_fp8_partial: d=%d must be a multiple of 8, at most 32768", d);
  SGLK_CHECK(splits >= 1 && part_stride >= rows * 2 * d && part_stride % 4 == 0 && aligned16(partials) &&
                 ((uintptr_t)q & 7) == 0,
             "silu_and_mul_quant_fp8_partial: bad partial layout");
  if (rows == 0) return SGLK_OK;
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((silu_mul_quant_kernel<T>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream,
                                        (const void*)nullptr, (uint8_t*)q, scales, d, partials, splits, part_stride,
                                        row_scale, col_scale));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_per_tensor_quant_fp8_rows(const void* x, void* q, float* scale, int64_t rows, int64_t cols, int64_t x_row_stride,
                                   int64_t q_row_stride, int is_static, int dtype, void* stream) {
  SGLK_CHECK(x && q && scale, "per_tensor_quant_fp8: null pointer");
  SGLK_CHECK(rows >= 0 && cols >= 0, "per_tensor_quant_fp8: negative size");
  SGLK_CHECK(rows <= 1 || (x_row_stride >= cols && q_row_stride >= cols), "per_tensor_quant_fp8: row pitch below the row");
  if (rows == 0 || cols == 0) return SGLK_OK;
  hipStream_t s = (hipStream_t)stream;
  const int esz = dtype == SGLK_F32 ? 4 : 2;
  const bool vec = cols % 8 == 0 && x_row_stride % 8 == 0 && q_row_stride % 8 == 0 && aligned16(x) &&
                   (reinterpret_cast<uintptr_t>(q) & 7) == 0 && (esz == 2 || (reinterpret_cast<uintptr_t>(x) & 31) == 0);
  const int64_t work = vec ? rows * (cols / 8) : rows * cols;
  const unsigned blocks = (unsigned)std::min<int64_t>((work + 255) / 256, 2048);
  // every block of the maximum pass ends in one atomic on the same word: few blocks, more elements a thread
  const unsigned mblocks = std::min(blocks, 256u);
  SGLK_DISPATCH_FLOAT(dtype, T, {
    if (vec) {
      if (!is_static)
        hipLaunchKernelGGL((per_tensor_absmax_kernel<T, true>), dim3(mblocks), dim3(256), 0, s, x, scale, rows, cols,
                           x_row_stride);
      hipLaunchKernelGGL((per_tensor_quant_kernel<T, true>), dim3(blocks), dim3(256), 0, s, x, (uint8_t*)q, scale, rows,
                         cols, x_row_stride, q_row_stride);
    } else {
      if (!is_static)
        hipLaunchKernelGGL((per_tensor_absmax_kernel<T, false>), dim3(mblocks), dim3(256), 0, s, x, scale, rows, cols,
                           x_row_stride);
      hipLaunchKernelGGL((per_tensor_quant_kernel<T, false>), dim3(blocks), dim3(256), 0, s, x, (uint8_t*)q, scale, rows,
                         cols, x_row_stride, q_row_stride);
    }
  });
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_per_tensor_quant_fp8(const void* x, void* q, float* scale, int64_t numel, int is_static, int dtype,
                              void* stream) {
  SGLK_CHECK(numel >= 0, "per_tensor_quant_fp8: negative size");
  return sglk_per_tensor_quant_fp8_rows(x, q, scale, 1, numel, numel, numel, is_static, dtype, stream);
}

int sglk_fp8_scaled_mm(const void* a, const void* b_nk, const float* scales_a, const float* scales_b, const void* bias,
                       void* out, int M, int N, int K, int64_t a_stride, int64_t b_stride, int64_t out_stride,
                       int out_dtype, void* stream) {
  SGLK_CHECK(a && b_nk && scales_a && scales_b && out, "fp8_scaled_mm: null pointer");
  SGLK_CHECK(M >= 0 && N > 0 && K > 0, "fp8_scaled_mm: bad shape M=%d N=%d K=%d", M, N, K);
  // fp8_gemm_kernel.cu:1086-1091: both inner dimensions must be multiples of 16 bytes
  SGLK_CHECK(K % 16 == 0, "fp8_scaled_mm: K=%d must be a multiple of 16", K);
  SGLK_CHECK(N % 16 == 0, "fp8_scaled_mm: N=%d must be a multiple of 16", N);
  SGLK_CHECK(a_stride % 16 == 0 && b_stride % 16 == 0 && aligned16(a) && aligned16(b_nk),
             "fp8_scaled_mm: operand rows must be 16-byte aligned");
  SGLK_CHECK(a_stride >= K && b_stride >= K && out_stride >= N, "fp8_scaled_mm: stride smaller than the row");
  if (M == 0) return SGLK_OK;
  MmArgs p{(const uint8_t*)a, (const uint8_t*)b_nk, scales_a, scales_b, bias, out, M, N, K, a_stride, b_stride,
           out_stride,        0,                    0,        0,        0};
  SGLK_DISPATCH_HALF(out_dtype, TO, return (launch_mm<TO, false>(p, (hipStream_t)stream)));
  return SGLK_OK;
}

int64_t sglk_fp8_mm_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int splits = fp8_splits(M, N, K);
  return splits > 1 ? (int64_t)splits * M * N * (int64_t)sizeof(float) : 0;
}

int sglk_fp8_repack(const void* w_nk, void* packed, int N, int K, int64_t w_stride, void* stream) {
  SGLK_CHECK(w_nk && packed, "fp8_repack: null pointer");
  SGLK_CHECK(N > 0 && K > 0 && N % 16 == 0 && K % 128 == 0, "fp8_repack: N=%d must be a multiple of 16, K=%d of 128", N, K);
  SGLK_CHECK(aligned16(w_nk) && aligned16(packed) && w_stride % 16 == 0 && w_stride >= K,
             "fp8_repack: rows must be 16-byte aligned");
  const int64_t pieces = (int64_t)(N / 16) * (K / 128) * 128;
  hipLaunchKernelGGL(fp8_repack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)w_nk, (uint8_t*)packed, N, K, w_stride);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int64_t sglk_fp8_mm_packed_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 128 != 0 || N % 16 != 0) return 0;
  const int splits = fp8_stream_splits(M, N, K);
  return splits > 1 ? (int64_t)splits * M * N * (int64_t)sizeof(float) : 0;
}

int sglk_fp8_scaled_mm_packed(const void* a, const void* packed, const float* scales_a, const float* scales_b,
                              const void* bias, void* out, int M, int N, int K, int64_t a_stride, int64_t out_stride,
                              void* workspace, int64_t workspace_bytes, int out_dtype, void* stream) {
  SGLK_CHECK(a && packed && scales_a && scales_b && out, "fp8_scaled_mm_packed: null pointer");
  SGLK_CHECK(M > 0 && N > 0 && K > 0 && N % 16 == 0 && K % 128 == 0,
             "fp8_scaled_mm_packed: N=%d must be a multiple of 16, K=%d of 128", N, K);
  SGLK_CHECK(aligned16(a) && aligned16(packed) && a_stride % 16 == 0 && a_stride >= K && out_stride >= N,
             "fp8_scaled_mm_packed: rows must be 16-byte aligned");
  int splits = fp8_stream_splits(M, N, K);
  const int kblocks = K / 128;
  if (splits > 1 && (!workspace || workspace_bytes < (int64_t)splits * M * N * (int64_t)sizeof(float) ||
                     !aligned16(workspace)))
    splits = 1;
  const int per = (kblocks + splits - 1) / splits;
  const int used = (kblocks + per - 1) / per;
  const int mblocks = (M + 63) / 64;
  float* partial = used > 1 ? (float*)workspace : nullptr;
  static const int env_ns = getenv("SGLK_FP8_STREAM_NS") ? atoi(getenv("SGLK_FP8_STREAM_NS")) : 0;
  const bool two = env_ns == 2;  // two strips per wave measured slower than one on every decode shape of the 8B model
  hipStream_t s = (hipStream_t)stream;
  SGLK_DISPATCH_HALF(out_dtype, TO, {
    if (two)
      hipLaunchKernelGGL((fp8_stream_kernel<TO, 2, 4>), dim3((N + 127) / 128, used, mblocks), dim3(256), 0, s,
                         (const uint8_t*)a, (const uint8_t*)packed, scales_a, scales_b, bias, out, partial, M, N, K, per,
                         a_stride, out_stride);
    else
      hipLaunchKernelGGL((fp8_stream_kernel<TO, 1, 4>), dim3((N + 63) / 64, used, mblocks), dim3(256), 0, s,
                         (const uint8_t*)a, (const uint8_t*)packed, scales_a, scales_b, bias, out, partial, M, N, K, per,
                         a_stride, out_stride);
    if (used > 1) {
      const int64_t vecs = (int64_t)M * (N / 4);
      hipLaunchKernelGGL((fp8_splitk_reduce_kernel<TO>), dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, s,
                         (const float*)workspace, used, scales_a, scales_b, bias, out, M, N, out_stride);
    }
  });
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_fp8_blockwise_scaled_mm_packed(const void* a, const void* packed, const float* scales_a, const float* scales_b,
                                        void* out, int M, int N, int K, int64_t sa_stride_m, int64_t sa_stride_k,
                                        int64_t sb_stride_n, int64_t sb_stride_k, void* workspace, int64_t workspace_bytes,
                                        int out_dtype, void* stream) {
  SGLK_CHECK(a && packed && scales_a && scales_b && out, "fp8_blockwise_scaled_mm_packed: null pointer");
  SGLK_CHECK(M > 0 && N > 0 && K > 0 && N % 16 == 0 && K % 128 == 0,
             "fp8_blockwise_scaled_mm_packed: N=%d must be a multiple of 16, K=%d of 128", N, K);
  SGLK_CHECK(aligned16(a) && aligned16(packed), "fp8_blockwise_scaled_mm_packed: operands must be 16-byte aligned");
  int splits = fp8_stream_splits(M, N, K);
  const int kblocks = K / 128;
  if (splits > 1 && (!workspace || workspace_bytes < (int64_t)splits * M * N * (int64_t)sizeof(float) ||
                     !aligned16(workspace)))
    splits = 1;
  const int per = (kblocks + splits - 1) / splits;
  const int used = (kblocks + per - 1) / per;
  float* partial = used > 1 ? (float*)workspace : nullptr;
  hipStream_t s = (hipStream_t)stream;
  SGLK_DISPATCH_HALF(out_dtype, TO, {
    // block-scaled: the scaling holds more registers; the deeper activation request does not pay here (measured)
    hipLaunchKernelGGL((fp8_stream_kernel<TO, 1, 4, true, 2>), dim3((N + 63) / 64, used, (M + 63) / 64), dim3(256), 0, s,
                       (const uint8_t*)a, (const uint8_t*)packed, scales_a, scales_b, (const void*)nullptr, out, partial, M,
                       N, K, per, (int64_t)K, (int64_t)N, sa_stride_m, sa_stride_k, sb_stride_n, sb_stride_k);
    if (used > 1) {
      const int64_t vecs = (int64_t)M * (N / 4);
      hipLaunchKernelGGL((fp8_splitk_reduce_kernel<TO>), dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, s,
                         (const float*)workspace, used, (const float*)nullptr, (const float*)nullptr,
                         (const void*)nullptr, out, M, N, (int64_t)N);
    }
  });
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_fp8_mm_packed_partial_splits(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 128 != 0 || N % 16 != 0) return 0;
  const int splits = fp8_stream_splits(M, N, K), kblocks = K / 128;
  const int per = (kblocks + splits - 1) / splits;
  return (kblocks + per - 1) / per;
}

int sglk_fp8_mm_packed_partial(const void* a, const void* packed, float* partial, int M, int N, int K, int64_t a_stride,
                               void* stream) {
  SGLK_CHECK(a && packed && partial, "fp8_mm_packed_partial: null pointer");
  SGLK_CHECK(M > 0 && N > 0 && K > 0 && N % 16 == 0 && K % 128 == 0,
             "fp8_mm_packed_partial: N=%d must be a multiple of 16, K=%d of 128", N, K);
  SGLK_CHECK(aligned16(a) && aligned16(packed) && aligned16(partial) && a_stride % 16 == 0 && a_stride >= K,
             "fp8_mm_packed_partial: rows must be 16-byte aligned");
  const int splits = fp8_stream_splits(M, N, K), kblocks = K / 128;
  const int per = (kblocks + splits - 1) / splits;
  const int used = (kblocks + per - 1) / per;
  hipLaunchKernelGGL((fp8_stream_kernel<bf16_t, 1, 4>), dim3((N + 63) / 64, used, (M + 63) / 64), dim3(256), 0,
                     (hipStream_t)stream, (const uint8_t*)a, (const uint8_t*)packed, (const float*)nullptr,
                     (const float*)nullptr, (const void*)nullptr, (void*)nullptr, partial, M, N, K, per, a_stride,
                     (int64_t)0);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_fp8_scaled_mm_ws(const void* a, const void* b_nk, const float* scales_a, const float* scales_b, const void* bias,
                          void* out, int M, int N, int K, int64_t a_stride, int64_t b_stride, int64_t out_stride,
                          void* workspace, int64_t workspace_bytes, int out_dtype, void* stream) {
  const int splits = (M > 0 && N > 0 && K > 0) ? fp8_splits(M, N, K) : 1;
  if (splits <= 1 || !workspace || workspace_bytes < sglk_fp8_mm_workspace_bytes(M, N, K) || !aligned16(workspace))
    return sglk_fp8_scaled_mm(a, b_nk, scales_a, scales_b, bias, out, M, N, K, a_stride, b_stride, out_stride, out_dtype,
                              stream);
  SGLK_CHECK(a && b_nk && scales_a && scales_b && out, "fp8_scaled_mm: null pointer");
  SGLK_CHECK(K % 16 == 0 && N % 16 == 0, "fp8_scaled_mm: N=%d and K=%d must be multiples of 16", N, K);
  SGLK_CHECK(a_stride % 16 == 0 && b_stride % 16 == 0 && aligned16(a) && aligned16(b_nk),
             "fp8_scaled_mm: operand rows must be 16-byte aligned");
  SGLK_CHECK(a_stride >= K && b_stride >= K && out_stride >= N, "fp8_scaled_mm: stride smaller than the row");
  const int kblocks = (K + 127) / 128;
  const int per = (kblocks + splits - 1) / splits;
  const int used = (kblocks + per - 1) / per;
  MmArgs p{};
  p.a = (const uint8_t*)a;
  p.b = (const uint8_t*)b_nk;
  p.sa = scales_a;
  p.sb = scales_b;
  p.out = out;
  p.M = M;
  p.N = N;
  p.K = K;
  p.a_stride = a_stride;
  p.b_stride = b_stride;
  p.out_stride = out_stride;
  p.partial = (float*)workspace;
  p.kb_per_split = per;
  hipStream_t s = (hipStream_t)stream;
  SGLK_DISPATCH_HALF(out_dtype, TO, {
    const int rc = launch_mm<TO, false>(p, s, used);
    if (rc != SGLK_OK) return rc;
    const int64_t vecs = (int64_t)M * (N / 4);
    hipLaunchKernelGGL((fp8_splitk_reduce_kernel<TO>), dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, s,
                       (const float*)workspace, used, scales_a, scales_b, bias, out, M, N, out_stride);
  });
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_bmm_fp8(const void* a, const void* b_nk, const float* scale_a, const float* scale_b, void* out, int batch, int M,
                 int N, int K, int64_t a_batch_stride, int64_t a_stride, int64_t b_batch_stride, int64_t b_stride,
                 int64_t out_batch_stride, int64_t out_stride, int a_e5m2, int b_e5m2, int out_dtype, void* stream) {
  SGLK_CHECK(batch >= 0 && M >= 0 && N > 0 && K > 0, "bmm_fp8: bad shape batch=%d M=%d N=%d K=%d", batch, M, N, K);
  SGLK_CHECK(batch == 0 || M == 0 || (a && b_nk && scale_a && scale_b && out), "bmm_fp8: null pointer");
  SGLK_CHECK(!(a_e5m2 && b_e5m2), "bmm_fp8: both operands e5m2 is not a valid combination");
  SGLK_CHECK(K % 16 == 0, "bmm_fp8: K=%d must be a multiple of 16", K);
  SGLK_CHECK(a_stride % 16 == 0 && b_stride % 16 == 0 && a_batch_stride % 16 == 0 && b_batch_stride % 16 == 0 &&
                 aligned16(a) && aligned16(b_nk),
             "bmm_fp8: operand rows must be 16-byte aligned");
  SGLK_CHECK(a_stride >= K && b_stride >= K && out_stride >= N, "bmm_fp8: stride smaller than the row");
  SGLK_CHECK(batch <= 65535, "bmm_fp8: batch %d too large", batch);
  if (M == 0 || batch == 0) return SGLK_OK;
  MmArgs p{};
  p.a = (const uint8_t*)a;
  p.b = (const uint8_t*)b_nk;
  p.sa = scale_a;
  p.sb = scale_b;
  p.bias = nullptr;
  p.out = out;
  p.M = M;
  p.N = N;
  p.K = K;
  p.a_stride = a_stride;
  p.b_stride = b_stride;
  p.out_stride = out_stride;
  p.a_batch = a_batch_stride;
  p.b_batch = b_batch_stride;
  p.out_batch = out_batch_stride;
  p.scalar_scales = 1;
  hipStream_t s = (hipStream_t)stream;
  SGLK_DISPATCH_HALF(out_dtype, TO, {
    if (a_e5m2) return (launch_mm<TO, false, 1>(p, s, batch));
    if (b_e5m2) return (launch_mm<TO, false, 2>(p, s, batch));
    return (launch_mm<TO, false, 0>(p, s, batch));
  });
  return SGLK_OK;
}

int sglk_fp8_blockwise_scaled_mm(const void* a, const void* b_nk, const float* scales_a, const float* scales_b,
                                 void* out, int M, int N, int K, int64_t sa_stride_m, int64_t sa_stride_k,
                                 int64_t sb_stride_n, int64_t sb_stride_k, int out_dtype, void* stream) {
  SGLK_CHECK(a && b_nk && scales_a && scales_b && out, "fp8_blockwise_scaled_mm: null pointer");
  SGLK_CHECK(M >= 0 && N > 0 && K > 0, "fp8_blockwise_scaled_mm: bad shape M=%d N=%d K=%d", M, N, K);
  // fp8_blockwise_gemm_kernel.cu:372-380: K and N multiples of 16 bytes, 128x128 weight blocks
  SGLK_CHECK(K % 16 == 0 && N % 16 == 0, "fp8_blockwise_scaled_mm: N=%d and K=%d must be multiples of 16", N, K);
  SGLK_CHECK(aligned16(a) && aligned16(b_nk), "fp8_blockwise_scaled_mm: operands must be 16-byte aligned");
  if (M == 0) return SGLK_OK;
  MmArgs p{(const uint8_t*)a, (const uint8_t*)b_nk, scales_a,    scales_b,    nullptr,     out,        M, N, K, K, K,
           N,                 sa_stride_m,          sa_stride_k, sb_stride_n, sb_stride_k};
  SGLK_DISPATCH_HALF(out_dtype, TO, return (launch_mm<TO, true>(p, (hipStream_t)stream)));
  return SGLK_OK;
}

int sglk_fp8_blockwise_scaled_mm_ws(const void* a, const void* b_nk, const float* scales_a, const float* scales_b,
                                    void* out, int M, int N, int K, int64_t sa_stride_m, int64_t sa_stride_k,
                                    int64_t sb_stride_n, int64_t sb_stride_k, void* workspace, int64_t workspace_bytes,
                                    int out_dtype, void* stream) {
  const int splits = (M > 0 && N > 0 && K > 0) ? fp8_splits(M, N, K) : 1;
  if (splits <= 1 || !workspace || workspace_bytes < sglk_fp8_mm_workspace_bytes(M, N, K) || !aligned16(workspace))
    return sglk_fp8_blockwise_scaled_mm(a, b_nk, scales_a, scales_b, out, M, N, K, sa_stride_m, sa_stride_k, sb_stride_n,
                                        sb_stride_k, out_dtype, stream);
  SGLK_CHECK(a && b_nk && scales_a && scales_b && out, "fp8_blockwise_scaled_mm: null pointer");
  SGLK_CHECK(K % 16 == 0 && N % 16 == 0, "fp8_blockwise_scaled_mm: N=%d and K=%d must be multiples of 16", N, K);
  SGLK_CHECK(aligned16(a) && aligned16(b_nk), "fp8_blockwise_scaled_mm: operands must be 16-byte aligned");
  const int kblocks = (K + 127) / 128;
  const int per = (kblocks + splits - 1) / splits;
  const int used = (kblocks + per - 1) / per;
  MmArgs p{(const uint8_t*)a, (const uint8_t*)b_nk, scales_a,    scales_b,    nullptr,     out,        M, N, K, K, K,
           N,                 sa_stride_m,          sa_stride_k, sb_stride_n, sb_stride_k};
  p.partial = (float*)workspace;
  p.kb_per_split = per;
  hipStream_t s = (hipStream_t)stream;
  SGLK_DISPATCH_HALF(out_dtype, TO, {
    const int rc = launch_mm<TO, true>(p, s, used);
    if (rc != SGLK_OK) return rc;
    const int64_t vecs = (int64_t)M * (N / 4);
    hipLaunchKernelGGL((fp8_splitk_reduce_kernel<TO>), dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, s,
                       (const float*)workspace, used, (const float*)nullptr, (const float*)nullptr, (const void*)nullptr,
                       out, M, N, (int64_t)N);
  });
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

}  // extern "C"
