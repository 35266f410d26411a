// Pieces of the AWQ GEMM kernels (awq.hip) kept apart from the kernels themselves.
#pragma once
#include "common.h"

namespace sglk {

// (OFF + q) as a 16-bit float: the nibble ORed into the mantissa of OFF = 128 (bf16) / 1024 (fp16)
template <typename T>
struct AwqMagic;
template <>
struct AwqMagic<bf16_t> {
  static constexpr uint32_t kOr = 0x43004300u;
  static constexpr float kOff = 128.f;
  static __device__ __forceinline__ float4v mma(int4v a, int4v b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfloat8, a), __builtin_bit_cast(bfloat8, b), c, 0,
                                                   0, 0);
  }
  static __device__ __forceinline__ float16v mma32(int4v a, int4v b, float16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfloat8, a), __builtin_bit_cast(bfloat8, b), c, 0,
                                                   0, 0);
  }
  // acc + lo + hi of a packed pair
  static __device__ __forceinline__ float pair_sum(uint32_t v, float acc) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, v), __builtin_bit_cast(bf2, 0x3F803F80u), acc, false);
  }
};
template <>
struct AwqMagic<f16_t> {
  static constexpr uint32_t kOr = 0x64006400u;
  static constexpr float kOff = 1024.f;
  static __device__ __forceinline__ float4v mma(int4v a, int4v b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0,
                                                  0);
  }
  static __device__ __forceinline__ float16v mma32(int4v a, int4v b, float16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0,
                                                  0);
  }
  static __device__ __forceinline__ float pair_sum(uint32_t v, float acc) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, v), __builtin_bit_cast(h2, 0x3C003C00u), acc, false);
  }
};


static constexpr int kGemmBM = 64;  // activation rows per block

// Work split of the 128-column x 512-k tile kernels: K in 512-k stages, `splits` blocks per column tile.
struct WidePlan {
  bool ok;
  int tiles, splits, stages_per_split, mblocks;
};
WidePlan plan_awq_wide(int M, int N, int K, int group_size, bool partial_mode = false);

}  // namespace sglk
