// Shared device/host helpers for libsglk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "sglk.h"

namespace sglk {

// ---- error plumbing -------------------------------------------------------------------------
char* err_buf();
int fail(int code, const char* fmt, ...);

#define SGLK_CHECK(cond, ...)                                  \
  do {                                                         \
    if (!(cond)) return ::sglk::fail(SGLK_EINVAL, __VA_ARGS__); \
  } while (0)

#define SGLK_HIP(call)                                                                      \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess)                                                                  \
      return ::sglk::fail(SGLK_EHIP, "%s failed: %s", #call, hipGetErrorString(e__));        \
  } while (0)

#define SGLK_LAUNCH_CHECK()                                                                       \
  do {                                                                                            \
    hipError_t e__ = hipGetLastError();                                                           \
    if (e__ != hipSuccess) return ::sglk::fail(SGLK_EHIP, "kernel launch failed: %s", hipGetErrorString(e__)); \
  } while (0)

__host__ __device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- 16-bit float types ---------------------------------------------------------------------
struct bf16_t {
  uint16_t x;
};
struct f16_t {
  uint16_t x;
};

typedef short short8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
typedef int int2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfloat8 __attribute__((ext_vector_type(8)));

template <typename T>
struct Conv;

template <>
struct Conv<bf16_t> {
  static __device__ __forceinline__ float to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
  static __device__ __forceinline__ uint16_t from_f32(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
    return __builtin_bit_cast(uint16_t, b);
  }
};

template <>
struct Conv<f16_t> {
  static __device__ __forceinline__ float to_f32(uint16_t v) {
    return (float)__builtin_bit_cast(_Float16, v);
  }
  static __device__ __forceinline__ uint16_t from_f32(float f) {
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
  }
};

template <>
struct Conv<float> {
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float f) { return f; }
};

template <typename T>
struct Raw {
  using type = uint16_t;
};
template <>
struct Raw<float> {
  using type = float;
};

template <typename T>
__device__ __forceinline__ float ld(const void* p, int64_t i) {
  using R = typename Raw<T>::type;
  return Conv<T>::to_f32(reinterpret_cast<const R*>(p)[i]);
}
template <typename T>
__device__ __forceinline__ void st(void* p, int64_t i, float v) {
  using R = typename Raw<T>::type;
  reinterpret_cast<R*>(p)[i] = Conv<T>::from_f32(v);
}

// ---- fp8 KV cache formats (SGLK_KV_* of sglk.h) ------------------------------------------------
// The pool bytes are what torch's Tensor.to(float8 type) writes (mem_cache/memory_pool.py:389-400) and the kernels read
// them back as `.to(query type)` does (triton_ops/decode_attention.py).  Both directions are integer arithmetic on
// the fp32 bit pattern, the same for the three formats (gfx950's conversion instructions know the two OCP formats only,
// and the reference's pool type on ROCm is the fnuz one, model_runner.py:847-851).
template <int F>
struct Fp8 {  // tag type: pool elements of format F
  static constexpr int M = F == SGLK_KV_E4M3FN ? 3 : 2;                             // mantissa bits
  static constexpr int B = F == SGLK_KV_E4M3FN ? 7 : (F == SGLK_KV_E5M2 ? 15 : 16);  // exponent bias
};

template <int F>
__device__ __forceinline__ float kv8_to_f32(uint32_t b) {
  constexpr int M = Fp8<F>::M, B = Fp8<F>::B;
  const uint32_t mag = b & 0x7Fu;
  float v;
  if ((mag >> M) != 0) {
    uint32_t bits = (mag << (23 - M)) + ((uint32_t)(127 - B) << 23);
    if (F == SGLK_KV_E4M3FN && mag == 0x7Fu) bits = 0x7FC00000u;
    if (F == SGLK_KV_E5M2 && mag >= 0x7Cu) bits = 0x7F800000u | ((mag & 3u) << 21);
    v = __uint_as_float(bits);
  } else {
    v = (float)mag * __uint_as_float((uint32_t)(127 + 1 - B - M) << 23);  // denormals: mantissa * 2^(1 - B - M)
  }
  if (F == SGLK_KV_E5M2FNUZ) return b == 0x80u ? __uint_as_float(0x7FC00000u) : ((b & 0x80u) ? -v : v);
  return (b & 0x80u) ? -v : v;
}

template <int F>
__device__ __forceinline__ uint32_t f32_to_kv8(float x) {
  constexpr int M = Fp8<F>::M, B = Fp8<F>::B, shift = 23 - M;
  const uint32_t bits = __float_as_uint(x), sign = bits >> 31, a = bits & 0x7FFFFFFFu;
  uint32_t code;
  if (a < ((uint32_t)(1 - B + 127) << 23)) {
    // below the smallest normal: the integer nearest to |x| / 2^(1 - B - M), ties to even, read off the mantissa of
    // |x| + 2^(24 - B - M)
    const float magic = __uint_as_float((uint32_t)(24 - B - M + 127) << 23);
    code = __float_as_uint(__fadd_rn(__uint_as_float(a), magic)) - __float_as_uint(magic);
  } else {
    const uint32_t r = a + ((1u << (shift - 1)) - 1u) + ((a >> shift) & 1u);
    code = (r >> shift) - ((uint32_t)(127 - B) << M);
  }
  if (F == SGLK_KV_E5M2FNUZ) {
    if (code >= 0x80u || a >= 0x7F800000u) return 0x80u;  // out of range and NaN: the one NaN code
    return code ? (code | (sign << 7)) : 0u;              // no negative zero
  }
  code = code < (F == SGLK_KV_E4M3FN ? 0x7Fu : 0x7Cu) ? code : (F == SGLK_KV_E4M3FN ? 0x7Fu : 0x7Cu);
  if (a > 0x7F800000u) code = 0x7Fu;
  return code | (sign << 7);
}

// element loads of a K / V pool: KV is the query type T (16- or 32-bit floats) or Fp8<F> (one byte per element)
template <typename KV>
struct KvElem {
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return ::sglk::ld<KV>(p, i); }
};
template <int F>
struct KvElem<Fp8<F>> {
  static __device__ __forceinline__ float ld(const void* p, int64_t i) {
    return kv8_to_f32<F>(reinterpret_cast<const uint8_t*>(p)[i]);
  }
  static __device__ __forceinline__ float value(uint32_t code) { return kv8_to_f32<F>(code); }
};

// kvf: 0 (pool elements of type T) or SGLK_KV_*; binds KV accordingly
#define SGLK_DISPATCH_KV(kvf, T, KV, ...)                                   \
  switch (kvf) {                                                            \
    case 0: {                                                               \
      using KV = T;                                                         \
      __VA_ARGS__;                                                          \
    } break;                                                                \
    case SGLK_KV_E4M3FN: {                                                  \
      using KV = ::sglk::Fp8<SGLK_KV_E4M3FN>;                               \
      __VA_ARGS__;                                                          \
    } break;                                                                \
    case SGLK_KV_E5M2: {                                                    \
      using KV = ::sglk::Fp8<SGLK_KV_E5M2>;                                 \
      __VA_ARGS__;                                                          \
    } break;                                                                \
    case SGLK_KV_E5M2FNUZ: {                                                \
      using KV = ::sglk::Fp8<SGLK_KV_E5M2FNUZ>;                             \
      __VA_ARGS__;                                                          \
    } break;                                                                \
    default:                                                                \
      return ::sglk::fail(SGLK_EINVAL, "unknown KV pool format %d", kvf);   \
  }

// ---- split-K partial sums handed to a consumer kernel -------------------------------------------
// acc[0..7] = sum over splits (in split order) of the 8 floats at partials[s][off .. off+7]; up to eight splits'
// loads are in flight together
__device__ __forceinline__ void sum_partials(const float* __restrict__ partials, int splits, int64_t part_stride,
                                             int64_t off, float (&acc)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int s0 = 0; s0 < splits; s0 += 8) {
    float4v a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* pr = partials + (int64_t)min(s0 + u, splits - 1) * part_stride + off;
      a[u] = *reinterpret_cast<const float4v*>(pr);
      b[u] = *reinterpret_cast<const float4v*>(pr + 4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (s0 + u < splits) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j] += a[u][j];
          acc[4 + j] += b[u][j];
        }
      }
    }
  }
}

// value an fp8 product's split sum stores: raw sum * scale of the activation row * scale of the weight column + bias,
// in this order of operations wherever it is formed (the reduction launch and the consumer kernels give the same bits)
__device__ __forceinline__ float scaled_sum(float acc, float row_scale, float col_scale, float bias) {
  return __fmaf_rn(__fmul_rn(acc, row_scale), col_scale, bias);
}

// ---- wave (64 lanes) reductions ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// One rotary pair in fp32 with a fixed operation order (two roundings per output), so that every kernel that rotates
// the same values produces the same bits whatever the compiler would contract elsewhere.
__device__ __forceinline__ void rope_pair(float xa, float xb, float c, float s, float& ra, float& rb) {
  ra = __fmaf_rn(xa, c, -__fmul_rn(xb, s));
  rb = __fmaf_rn(xb, c, __fmul_rn(xa, s));
}

__device__ __forceinline__ int64_t ld_idx(const void* p, int is64, int64_t i) {
  return is64 ? reinterpret_cast<const int64_t*>(p)[i] : (int64_t) reinterpret_cast<const int32_t*>(p)[i];
}

// dtype dispatch for 16-bit/32-bit float element kernels
#define SGLK_DISPATCH_FLOAT(dtype, T, ...)                                   \
  switch (dtype) {                                                           \
    case SGLK_F32: {                                                         \
      using T = float;                                                       \
      __VA_ARGS__;                                                           \
    } break;                                                                 \
    case SGLK_F16: {                                                         \
      using T = ::sglk::f16_t;                                               \
      __VA_ARGS__;                                                           \
    } break;                                                                 \
    case SGLK_BF16: {                                                        \
      using T = ::sglk::bf16_t;                                              \
      __VA_ARGS__;                                                           \
    } break;                                                                 \
    default:                                                                 \
      return ::sglk::fail(SGLK_EINVAL, "unsupported dtype code %d", dtype);  \
  }

#define SGLK_DISPATCH_HALF(dtype, T, ...)                                              \
  switch (dtype) {                                                                     \
    case SGLK_F16: {                                                                   \
      using T = ::sglk::f16_t;                                                         \
      __VA_ARGS__;                                                                     \
    } break;                                                                           \
    case SGLK_BF16: {                                                                  \
      using T = ::sglk::bf16_t;                                                        \
      __VA_ARGS__;                                                                     \
    } break;                                                                           \
    default:                                                                           \
      return ::sglk::fail(SGLK_EINVAL, "dtype code %d: expected fp16 or bf16", dtype); \
  }

inline int elem_size(int dtype) { return dtype == SGLK_F32 ? 4 : (dtype == SGLK_FP8_E4M3 ? 1 : 2); }

}  // namespace sglk
