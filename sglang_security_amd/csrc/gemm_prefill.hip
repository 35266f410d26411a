// Large-M GEMM for the extend (prefill) step: out[M, N] = x[M, K] @ W (+ bias), 16-bit activations, for
//   * AWQ int4 weights in the packed layout of sglk_awq_repack, dequantised on the way into LDS
//     ((q - z) * scale rounded to the activation type: the very weights the reference's
//     awq_dequantize-then-matmul multiplies by, python/sglang/srt/layers/quantization/awq.py:197-215), and
//   * dense 16-bit weights [N, K] (bf16 / fp16 checkpoints: layers/linear.py UnquantizedLinearMethod).
//
// A block of four waves owns a 256 x 128 output tile and walks K in steps of 32 through two LDS stages
// (A: 256 rows x 64 bytes, B: 128 columns x 64 bytes, k contiguous in both, 16-byte chunks XOR-swizzled by
// (row >> 2) & 3 so that the eight lanes a 16-byte LDS read serves at a time fall on eight different bank
// groups).  Wave (wm, wn) multiplies rows [128 wm, +128) by columns [64 wn, +64): 4 x 2 tiles of
// v_mfma_f32_32x32x16, 128 accumulator registers, six operand reads (6 KiB) per eight matrix instructions.
// The next stage's rows travel global -> registers while the current one is multiplied, and are parked (the
// int4 words dequantised first) before the one barrier per stage.  The output tile is transposed through LDS
// so that it leaves in 16-byte row pieces.
#include <algorithm>

#include "common.h"

namespace sglk {

template <typename T>
struct Mfma32;
template <>
struct Mfma32<bf16_t> {
  static __device__ __forceinline__ float16v mma(int4v a, int4v b, float16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfloat8, a), __builtin_bit_cast(bfloat8, b), c, 0,
                                                   0, 0);
  }
};
template <>
struct Mfma32<f16_t> {
  static __device__ __forceinline__ float16v mma(int4v a, int4v b, float16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0,
                                                  0);
  }
};

static constexpr int kPfBM = 256, kPfBN = 128, kPfBK = 32;
static constexpr int kPfABytes = kPfBM * 64, kPfBBytes = kPfBN * 64;
static constexpr int kPfStage = kPfABytes + kPfBBytes;  // 24 KiB
static constexpr int kPfLds = 2 * kPfStage;             // 48 KiB: two blocks per CU

// 16-byte chunk c (k = 8c .. 8c + 7 of the stage) of row / column r
__device__ __forceinline__ int pf_off(int r, int c) { return r * 64 + ((c ^ ((r >> 2) & 3)) << 4); }

template <typename T, bool AWQ>
__global__ void __launch_bounds__(256, 2) prefill_gemm_kernel(
    const uint16_t* __restrict__ x, const void* __restrict__ wptr, const uint16_t* __restrict__ scales,
    const int32_t* __restrict__ qzeros, const uint16_t* __restrict__ bias, uint16_t* __restrict__ out, int M, int N, int K,
    int group_size, int64_t x_stride, int64_t w_stride, int64_t out_stride) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.x * kPfBM, n0 = blockIdx.y * kPfBN;
  const int nk = K / kPfBK;

  // ---- global -> register roles
  // A: chunk q = tid + 256 i (i < 4): row q >> 2, chunk q & 3
  const uint16_t* asrc[4];
  bool aok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + 256 * i, row = q >> 2, c = q & 3;
    aok[i] = m0 + row < M;
    asrc[i] = x + (int64_t)(aok[i] ? m0 + row : 0) * x_stride + 8 * c;
  }
  // B: (column q >> 2, chunk q & 3) for q = tid + 256 i (i < 2)
  const uint16_t* bsrc[2] = {nullptr, nullptr};  // dense: row n0 + col of W [N, K]
  const int32_t* psrc[2] = {nullptr, nullptr};   // AWQ: the lane's four words of a 128-k block
  int bcol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = tid + 256 * i, col = q >> 2, c = q & 3;
    bcol[i] = n0 + col;
    if (AWQ) {
      // word d of lane (i16, g) of strip s holds k = 128 kb + 32 d + 8 g + e of column 16 s + i16: chunk c <-> g
      psrc[i] = reinterpret_cast<const int32_t*>(wptr) + ((int64_t)(bcol[i] >> 4) * 64 + (bcol[i] & 15) + 16 * c) * 4;
    } else {
      bsrc[i] = reinterpret_cast<const uint16_t*>(wptr) + (int64_t)bcol[i] * w_stride + 8 * c;
    }
  }
  const int nstrips = N >> 4;

  int4v areg[4], breg[2];
  int4v wq[2];         // AWQ: the words of the current 128-k block
  float sc[2];         // ... scale and zero point of this thread's two columns in the current group
  int zp[2];
  auto fetch_a = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      areg[i] = aok[i] ? *reinterpret_cast<const int4v*>(asrc[i] + kPfBK * kt) : int4v{0, 0, 0, 0};
  };
  auto fetch_b = [&](int kt) {
    if (AWQ) {
      if ((kt & 3) == 0) {
        const int kb = kt >> 2;
        const int gq = (128 * kb) / group_size;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          wq[i] = *reinterpret_cast<const int4v*>(psrc[i] + (int64_t)kb * nstrips * 256);
          sc[i] = Conv<T>::to_f32(scales[(int64_t)gq * N + bcol[i]]);
          const uint32_t zw = (uint32_t)qzeros[(int64_t)gq * (N >> 3) + (bcol[i] >> 3)];
          const int j = bcol[i] & 7;
          zp[i] = (int)((zw >> (4 * ((j >> 1) + 4 * (j & 1)))) & 0xF);  // AWQ nibble order {0, 4, 1, 5, 2, 6, 3, 7}
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) breg[i] = *reinterpret_cast<const int4v*>(bsrc[i] + kPfBK * kt);
    }
  };
  auto park = [&](int kt, int buf) {
    char* a = lds + buf * kPfStage;
    char* b = a + kPfABytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + 256 * i;
      *reinterpret_cast<int4v*>(a + pf_off(q >> 2, q & 3)) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + 256 * i;
      int4v v;
      if (AWQ) {
        const uint32_t w = (uint32_t)wq[i][kt & 3];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          // k offsets e = 2p (nibble p) and e = 2p + 1 (nibble p + 4)
          const float v0 = (float)((int)((w >> (4 * p)) & 0xF) - zp[i]) * sc[i];
          const float v1 = (float)((int)((w >> (4 * p + 16)) & 0xF) - zp[i]) * sc[i];
          v[p] = (int)((uint32_t)Conv<T>::from_f32(v0) | ((uint32_t)Conv<T>::from_f32(v1) << 16));
        }
      } else {
        v = breg[i];
      }
      *reinterpret_cast<int4v*>(b + pf_off(q >> 2, q & 3)) = v;
    }
  };

  float16v acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  fetch_a(0);
  fetch_b(0);
  park(0, 0);
  __syncthreads();

  const int r32 = lane & 31, h = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {
      fetch_a(kt + 1);
      fetch_b(kt + 1);
    }
    const char* a = lds + buf * kPfStage;
    const char* b = a + kPfABytes;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      int4v af[4], bf[2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        af[mt] = *reinterpret_cast<const int4v*>(a + pf_off(128 * wm + 32 * mt + r32, 2 * ks + h));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        bf[nt] = *reinterpret_cast<const int4v*>(b + pf_off(64 * wn + 32 * nt + r32, 2 * ks + h));
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = Mfma32<T>::mma(af[mt], bf[nt], acc[mt][nt]);
    }
    if (kt + 1 < nk) park(kt + 1, buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: every wave transposes its 32 x 32 tiles through 2.5 KiB of LDS of its own (rows padded to 80 bytes)
  // and stores 16-byte row pieces.  Accumulator register r of lane (c, h) is D[(r & 3) + 8 (r >> 2) + 4 h][c].
  char* tl = lds + wid * 2560;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col0 = n0 + 64 * wn + 32 * nt;
      const float bv = bias ? Conv<T>::to_f32(bias[min(col0 + r32, N - 1)]) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        *reinterpret_cast<uint16_t*>(tl + row * 80 + r32 * 2) = Conv<T>::from_f32(acc[mt][nt][r] + bv);
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int q = lane + 64 * i, row = q >> 2, c = q & 3;
        const int m = m0 + 128 * wm + 32 * mt + row, n = col0 + 8 * c;
        const int4v v = *reinterpret_cast<const int4v*>(tl + row * 80 + c * 16);
        if (m < M && n < N) *reinterpret_cast<int4v*>(out + (int64_t)m * out_stride + n) = v;
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

static int prefill_checks(const void* x, const void* w, const void* out, int M, int N, int K, int64_t x_stride,
                          int64_t out_stride, const char* what) {
  SGLK_CHECK(x && w && out, "%s: null pointer", what);
  SGLK_CHECK(M > 0 && N > 0 && K > 0, "%s: bad shape", what);
  SGLK_CHECK(K % 128 == 0 && N % 128 == 0, "%s: K=%d and N=%d must be multiples of 128", what, K, N);
  SGLK_CHECK(aligned16(x) && aligned16(w) && aligned16(out) && x_stride % 8 == 0 && out_stride % 8 == 0,
             "%s: rows must be 16-byte aligned", what);
  return SGLK_OK;
}

}  // namespace sglk

using namespace sglk;

extern "C" {

int sglk_awq_gemm_prefill(const void* x, const int32_t* packed, const void* scales, const int32_t* qzeros,
                          const void* bias, void* out, int M, int N, int K, int group_size, int64_t x_stride,
                          int64_t out_stride, int dtype, void* stream) {
  if (int rc = prefill_checks(x, packed, out, M, N, K, x_stride, out_stride, "awq_gemm_prefill")) return rc;
  SGLK_CHECK(scales && qzeros, "awq_gemm_prefill: null pointer");
  SGLK_CHECK(group_size > 0 && group_size % 128 == 0 && K % group_size == 0,
             "awq_gemm_prefill: group size %d must be a multiple of 128 that divides K", group_size);
  dim3 grid((M + kPfBM - 1) / kPfBM, N / kPfBN);
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((prefill_gemm_kernel<T, true>), grid, dim3(256), kPfLds, (hipStream_t)stream,
                                        (const uint16_t*)x, (const void*)packed, (const uint16_t*)scales, qzeros,
                                        (const uint16_t*)bias, (uint16_t*)out, M, N, K, group_size, x_stride, (int64_t)0,
                                        out_stride));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_dense_gemm_prefill(const void* x, const void* w, const void* bias, void* out, int M, int N, int K,
                            int64_t x_stride, int64_t w_stride, int64_t out_stride, int dtype, void* stream) {
  if (int rc = prefill_checks(x, w, out, M, N, K, x_stride, out_stride, "dense_gemm_prefill")) return rc;
  SGLK_CHECK(w_stride % 8 == 0, "dense_gemm_prefill: weight rows must be 16-byte aligned");
  dim3 grid((M + kPfBM - 1) / kPfBM, N / kPfBN);
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((prefill_gemm_kernel<T, false>), grid, dim3(256), kPfLds, (hipStream_t)stream,
                                        (const uint16_t*)x, w, (const uint16_t*)nullptr, (const int32_t*)nullptr,
                                        (const uint16_t*)bias, (uint16_t*)out, M, N, K, 128, x_stride, w_stride,
                                        out_stride));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}
}
