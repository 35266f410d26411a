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
