// AWQ int4: dequantise (reference layout), one-time repack, and the fused dequant + MFMA GEMM.
//
// HBM layouts
//   reference (python/sglang/srt/layers/quantization/awq.py:98-151):
//     qweight int32 [K, N/8]  -- 8 nibbles per word along N, AWQ order {0,4,1,5,2,6,3,7}
//     scales  half  [K/G, N]
//     qzeros  int32 [K/G, N/8]
//   repacked for the matrix cores (sglk_awq_repack), same byte count as qweight:
//     packed int32 [N/16][K/128][64 lanes][4]
//     word d of lane (i = lane & 15, g = lane >> 4) in tile (nb, kb) holds the 8 weights
//     k = 128*kb + 32*d + 8*g + e (e = 0..7) of column n = 16*nb + i, with e = 2p in nibble p and
//     e = 2p+1 in nibble p+4.  ((w >> 4p) & 0x000F000F) | MAGIC is then, as it stands, register p of
//     the 16x16x32 MFMA operand holding (OFF + q) for k-slots 2p, 2p+1, OFF = 128 (bf16) / 1024 (fp16).
//     A wave streams one 16-column strip as consecutive 1-KiB pieces, 16 bytes per lane.
//
// GEMM arithmetic, per 128-row block kb of K with scale group gq = 128*kb / G:
//     acc[m,n]  = sum_k x[m,k] * (OFF + q[k,n])                       (matrix cores, fp32)
//     out[m,n] += s[gq,n] * acc[m,n] - s[gq,n] * (OFF + z[gq,n]) * X[kb,m],   X[kb,m] = sum_k x[m,k]
// i.e. out = x @ ((q - z) * s) with the product (q - z) * s never rounded to half precision.
#include "common.h"

namespace sglk {

__constant__ int kAwqOrd[8] = {0, 4, 1, 5, 2, 6, 3, 7};

// ---------------------------------------------------------------------------------------------
// awq_dequantize (awq_kernel.cu:126-182 of the reference): one thread = one int32 = 8 outputs
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) awq_dequant_kernel(const int32_t* __restrict__ qweight,
                                                          const uint16_t* __restrict__ scales,
                                                          const int32_t* __restrict__ qzeros,
                                                          uint16_t* __restrict__ out, int rows, int qcols,
                                                          int group_size) {
  const int64_t total = (int64_t)rows * qcols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / qcols), c = (int)(i - (int64_t)r * qcols);
    const int gq = r / group_size;
    const uint32_t w = (uint32_t)qweight[i];
    const uint32_t z = (uint32_t)qzeros[(int64_t)gq * qcols + c];
    const short8 sv = *reinterpret_cast<const short8*>(scales + ((int64_t)gq * qcols + c) * 8);
    short8 ov;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int sh = 4 * ((j & 1) * 4 + (j >> 1));  // AWQ order 0,4,1,5,2,6,3,7
      const float diff = (float)((int)((w >> sh) & 0xF) - (int)((z >> sh) & 0xF));
      // exact in fp32, one rounding to the scale type: == half-precision sub then mul of the reference
      ov[j] = (short)Conv<T>::from_f32(diff * Conv<T>::to_f32((uint16_t)sv[j]));
    }
    *reinterpret_cast<short8*>(out + i * 8) = ov;
  }
}

// ---------------------------------------------------------------------------------------------
// repack: one thread per packed word
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) awq_repack_kernel(const int32_t* __restrict__ qweight,
                                                         int32_t* __restrict__ packed, int K, int qcols) {
  const int64_t total = (int64_t)K * qcols;  // words
  const int kblocks = K / 128;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int d = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    const int64_t tile = idx >> 8;
    const int kb = (int)(tile % kblocks);
    const int nb = (int)(tile / kblocks);
    const int i = lane & 15, g = lane >> 4;
    const int n = 16 * nb + i;
    const int sh = 4 * kAwqOrd[n & 7];
    const int k0 = 128 * kb + 32 * d + 8 * g;
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t q = ((uint32_t)qweight[(int64_t)(k0 + e) * qcols + (n >> 3)] >> sh) & 0xF;
      const int pos = (e & 1) ? (e >> 1) + 4 : (e >> 1);
      w |= q << (4 * pos);
    }
    packed[idx] = (int32_t)w;
  }
}

// X[kb][m] = sum over the 128 activations of block kb, fp32, laid out [K/128][Mpad]
template <typename T>
__global__ void __launch_bounds__(256) awq_xsum_kernel(const uint16_t* __restrict__ x, float* __restrict__ xsum,
                                                       int M, int Mpad, int K, int64_t x_stride) {
  // one 16-lane group per (m, kb): 16 lanes x 8 elements = 128
  const int kblocks = K / 128;
  const int64_t item = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  if (item >= (int64_t)Mpad * kblocks) return;
  const int m = (int)(item % Mpad), kb = (int)(item / Mpad);
  float s = 0.f;
  if (m < M) {
    const short8 v = *reinterpret_cast<const short8*>(x + (int64_t)m * x_stride + 128 * kb + 8 * sub);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += Conv<T>::to_f32((uint16_t)v[j]);
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  s += __shfl_xor(s, 8, 64);
  if (sub == 0) xsum[(int64_t)kb * Mpad + m] = s;
}

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
};
template <>
struct AwqMagic<f16_t> {
  static constexpr uint32_t kOr = 0x64006400u;
  static constexpr float kOff = 1024.f;
  static __device__ __forceinline__ float4v mma(int4v a, int4v b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0,
                                                  0);
  }
};

static constexpr int kGemmBM = 64;          // activation rows per block (4 MFMA row tiles)
static constexpr int kXTileBytes = 64 * 256;  // 64 rows x 128 k x 2 B

// x tile in LDS: 16-byte chunk c of row r at slot c ^ (r & 15) (conflict-free row reads, as the K tile
// of the attention kernel)
__device__ __forceinline__ int x_off(int r, int c) { return r * 256 + ((c ^ (r & 15)) << 4); }

// TN: 16-column strips per wave.  Block = 4 waves = 64*TN columns x 64 rows x (K / splits).
template <typename T, int TN>
__global__ void __launch_bounds__(256, 2) awq_gemm_kernel(
    const uint16_t* __restrict__ x, const int32_t* __restrict__ packed, const uint16_t* __restrict__ scales,
    const int32_t* __restrict__ qzeros, const uint16_t* __restrict__ bias, const float* __restrict__ xsum,
    uint16_t* __restrict__ out, float* __restrict__ partial, int M, int Mpad, int N, int K, int group_size,
    int kb_per_split, int64_t x_stride, int64_t out_stride) {
  __shared__ __attribute__((aligned(16))) char xs[2 * kXTileBytes];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int kblocks = K / 128;
  const int m0 = blockIdx.z * kGemmBM;
  const int kb0 = blockIdx.y * kb_per_split;
  const int nkb = min(kb_per_split, kblocks - kb0);
  const int nb0 = (blockIdx.x * 4 + wid) * TN;  // first 16-column strip of this wave
  const int qcols = N / 8;

  // ---- x staging: thread (row = tid / 4, quarter = tid % 4) moves 64 bytes per 128-k block
  const int xr = threadIdx.x >> 2, xq = threadIdx.x & 3;
  const bool xrow_ok = (m0 + xr) < M;
  const uint16_t* xsrc = x + (int64_t)(m0 + xr) * x_stride + 32 * xq;
  int4v xreg[4];
  auto x_fetch = [&](int kb) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      xreg[c] = xrow_ok ? *reinterpret_cast<const int4v*>(xsrc + 128 * kb + 8 * c) : int4v{0, 0, 0, 0};
  };
  auto x_park = [&](int buf) {
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<int4v*>(xs + buf * kXTileBytes + x_off(xr, 4 * xq + c)) = xreg[c];
  };

  // ---- weight stream: strip t of this wave, block kb: 16 bytes per lane
  const int4v* wsrc[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t)
    wsrc[t] = reinterpret_cast<const int4v*>(packed) + ((int64_t)(nb0 + t) * kblocks + kb0) * 64 + lane;
  constexpr int PF = 4;  // 128-k blocks of weights in flight per wave
  int4v wq[PF][TN];
#pragma unroll
  for (int p = 0; p < PF; ++p)
#pragma unroll
    for (int t = 0; t < TN; ++t)
      if (p < nkb) wq[p][t] = wsrc[t][(int64_t)p * 64];

  float4v oacc[TN][4];
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) oacc[t][mt] = float4v{0.f, 0.f, 0.f, 0.f};

  x_fetch(kb0);
  x_park(0);
  __syncthreads();

  for (int base = 0; base < nkb; base += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int j = base + p;  // block index within this split (wave-uniform)
      if (j < nkb) {
        const int kb = kb0 + j;
        const int buf = j & 1;
        if (j + 1 < nkb) x_fetch(kb + 1);
        // per-column scale and zero of this block's group
        const int gq = (128 * kb) / group_size;
        float sc[TN], c1[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int n = 16 * (nb0 + t) + i16;
          const float s = Conv<T>::to_f32(scales[(int64_t)gq * N + n]);
          const uint32_t zw = (uint32_t)qzeros[(int64_t)gq * qcols + (n >> 3)];
          const float z = (float)((zw >> (4 * kAwqOrd[n & 7])) & 0xF);
          sc[t] = s;
          c1[t] = -s * (AwqMagic<T>::kOff + z);
        }
        float4v xs4[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          xs4[mt] = *reinterpret_cast<const float4v*>(xsum + (int64_t)kb * Mpad + m0 + 16 * mt + 4 * g);

        float4v acc[TN][4];
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[t][mt] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          int4v af[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            af[mt] = *reinterpret_cast<const int4v*>(xs + buf * kXTileBytes + x_off(16 * mt + i16, 4 * ks + g));
#pragma unroll
          for (int t = 0; t < TN; ++t) {
            const uint32_t w = (uint32_t)wq[p][t][ks];
            int4v bf;
#pragma unroll
            for (int e = 0; e < 4; ++e) bf[e] = (int)(((w >> (4 * e)) & 0x000F000Fu) | AwqMagic<T>::kOr);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[t][mt] = AwqMagic<T>::mma(af[mt], bf, acc[t][mt]);
          }
        }
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) oacc[t][mt] += acc[t][mt] * sc[t] + xs4[mt] * c1[t];
        // refill this ring slot, hand the next x tile to LDS
        if (j + PF < nkb) {
#pragma unroll
          for (int t = 0; t < TN; ++t) wq[p][t] = wsrc[t][(int64_t)(j + PF) * 64];
        }
        if (j + 1 < nkb) x_park(buf ^ 1);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: lane (i16, g) holds rows 16*mt + 4*g + r of column 16*(nb0+t) + i16
  const bool direct = partial == nullptr;
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int n = 16 * (nb0 + t) + i16;
    const float bv = (direct && bias) ? Conv<T>::to_f32(bias[n]) : 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * mt + 4 * g + r;
        if (direct) {
          if (m < M) out[(int64_t)m * out_stride + n] = Conv<T>::from_f32(oacc[t][mt][r] + bv);
        } else {
          partial[((int64_t)blockIdx.y * Mpad + m) * N + n] = oacc[t][mt][r];
        }
      }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ partial,
                                                            const uint16_t* __restrict__ bias,
                                                            uint16_t* __restrict__ out, int M, int Mpad, int N,
                                                            int splits, int64_t out_stride) {
  const int nv = N >> 2;
  const int64_t total = (int64_t)M * nv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / nv), c = (int)(i - (int64_t)m * nv);
    float4v a = float4v{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s) a += *reinterpret_cast<const float4v*>(partial + ((int64_t)s * Mpad + m) * N + 4 * c);
    short4v o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float b = bias ? Conv<T>::to_f32(bias[4 * c + j]) : 0.f;
      o[j] = (short)Conv<T>::from_f32(a[j] + b);
    }
    *reinterpret_cast<short4v*>(out + (int64_t)m * out_stride + 4 * c) = o;
  }
}

struct GemmPlan {
  int tn, splits, kb_per_split, mblocks, mpad;
};

static GemmPlan plan_awq_gemm(int M, int N, int K) {
  GemmPlan p;
  p.mblocks = (M + kGemmBM - 1) / kGemmBM;
  p.mpad = p.mblocks * kGemmBM;
  p.tn = 1;
  const int kblocks = K / 128;
  const int nblocks = N / (64 * p.tn);
  // enough blocks to cover 256 CUs twice, but keep >= 4 k-blocks (512 k) per split
  int splits = 1;
  while (nblocks * p.mblocks * splits < 256 && kblocks / (splits * 2) >= 4) splits *= 2;
  p.kb_per_split = (kblocks + splits - 1) / splits;
  p.splits = (kblocks + p.kb_per_split - 1) / p.kb_per_split;
  return p;
}

}  // namespace sglk

using namespace sglk;

extern "C" {

int sglk_awq_dequantize(const int32_t* qweight, const void* scales, const int32_t* qzeros, void* out, int rows,
                        int qcols, int group_size, int dtype, void* stream) {
  SGLK_CHECK(qweight && scales && qzeros && out, "awq_dequantize: null pointer");
  SGLK_CHECK(rows > 0 && qcols > 0 && group_size > 0 && rows % group_size == 0, "awq_dequantize: bad shape");
  SGLK_CHECK(aligned16(scales) && aligned16(out), "awq_dequantize: scales/out must be 16-byte aligned");
  int blocks = (int)(((int64_t)rows * qcols + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((awq_dequant_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                                        qweight, (const uint16_t*)scales, qzeros, (uint16_t*)out, rows, qcols,
                                        group_size));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_awq_repack(const int32_t* qweight, int32_t* packed, int rows, int qcols, void* stream) {
  SGLK_CHECK(qweight && packed, "awq_repack: null pointer");
  SGLK_CHECK(rows % 128 == 0, "awq_repack: K=%d must be a multiple of 128", rows);
  SGLK_CHECK(qcols % 2 == 0, "awq_repack: N=%d must be a multiple of 16", qcols * 8);
  int blocks = (int)(((int64_t)rows * qcols + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(awq_repack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, qweight, packed, rows, qcols);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int64_t sglk_awq_gemm_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 128) return 0;
  const GemmPlan p = plan_awq_gemm(M, N, K);
  int64_t bytes = (int64_t)(K / 128) * p.mpad * 4;  // X block sums
  bytes = (bytes + 255) & ~(int64_t)255;
  if (p.splits > 1) bytes += (int64_t)p.splits * p.mpad * N * 4;
  return bytes;
}

int sglk_awq_gemm(const void* x, const int32_t* packed, const void* scales, const int32_t* qzeros, const void* bias,
                  void* out, float* workspace, int M, int N, int K, int group_size, int64_t x_stride,
                  int64_t out_stride, int dtype, void* stream) {
  SGLK_CHECK(x && packed && scales && qzeros && out && workspace, "awq_gemm: null pointer");
  SGLK_CHECK(M > 0 && N > 0 && K > 0, "awq_gemm: bad shape");
  SGLK_CHECK(K % 128 == 0, "awq_gemm: K=%d must be a multiple of 128", K);
  SGLK_CHECK(N % 64 == 0, "awq_gemm: N=%d must be a multiple of 64", N);
  SGLK_CHECK(group_size % 128 == 0 && K % group_size == 0, "awq_gemm: group_size=%d must be a multiple of 128 dividing K",
             group_size);
  SGLK_CHECK(aligned16(x) && x_stride % 8 == 0, "awq_gemm: x rows must be 16-byte aligned");
  SGLK_CHECK(out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0, "awq_gemm: out rows must be 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const GemmPlan p = plan_awq_gemm(M, N, K);
  float* xsum = workspace;
  int64_t off = (int64_t)(K / 128) * p.mpad * 4;
  off = (off + 255) & ~(int64_t)255;
  float* partial = p.splits > 1 ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + off) : nullptr;
  const int64_t items = (int64_t)p.mpad * (K / 128);
  dim3 grid(N / (64 * p.tn), p.splits, p.mblocks);
  SGLK_DISPATCH_HALF(dtype, T, {
    hipLaunchKernelGGL((awq_xsum_kernel<T>), dim3((unsigned)((items * 16 + 255) / 256)), dim3(256), 0, s,
                       (const uint16_t*)x, xsum, M, p.mpad, K, x_stride);
    hipLaunchKernelGGL((awq_gemm_kernel<T, 1>), grid, dim3(256), 0, s, (const uint16_t*)x, packed,
                       (const uint16_t*)scales, qzeros, (const uint16_t*)bias, xsum, (uint16_t*)out, partial, M,
                       p.mpad, N, K, group_size, p.kb_per_split, x_stride, out_stride);
    if (partial) {
      int blocks = (int)(((int64_t)M * (N / 4) + 255) / 256);
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, s, partial, (const uint16_t*)bias,
                         (uint16_t*)out, M, p.mpad, N, p.splits, out_stride);
    }
  });
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}
}
