// Skinny dense GEMM  out[M,N] = x[M,K] @ W[N,K]^T (+bias)  for 16-bit types: the unquantised lm_head
// and the bf16 configuration at decode batch sizes.  HBM-bound on W: every wave streams a strip of 16
// weight rows straight into MFMA operand registers (per load instruction the four lanes of a row
// cover 64 contiguous bytes, so one instruction touches one cache line per row; no LDS round trip
// for the operand that is read once), while the activations, which every wave
// re-reads, are staged through an XOR-swizzled LDS tile shared by the block.
//
// Weight layouts.  Row-major [N, K] (the reference's): an instruction touches 16 rows, K * 2 bytes apart, 64 bytes
// each.  Packed (sglk_dense_repack, once at load time): [N / 16 strips][K / 128 k-blocks][4 k-steps][64 lanes]
// [16 bytes], i.e. every instruction of a wave reads 1 KiB of consecutive memory and a wave walks its strip as one
// sequential stream -- the row-major form leaves a fifth of the HBM bandwidth unused on the lm_head (rows 8 KiB
// apart fall on few channels at a time).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace sglk {

template <typename T>
struct DenseMma;
template <>
struct DenseMma<bf16_t> {
  static __device__ __forceinline__ float4v mma(int4v a, int4v b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfloat8, a), __builtin_bit_cast(bfloat8, b), c, 0,
                                                   0, 0);
  }
};
template <>
struct DenseMma<f16_t> {
  static __device__ __forceinline__ float4v mma(int4v a, int4v b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0,
                                                  0);
  }
};

static constexpr int kDenseXTile = 64 * 256;
__device__ __forceinline__ int dx_off(int r, int c) { return r * 256 + ((c ^ (r & 15)) << 4); }

template <typename T, int NS, int PF, bool PACKED = false>
__global__ void __launch_bounds__(256, 2) dense_gemm_nt_kernel(const uint16_t* __restrict__ x,
                                                               const uint16_t* __restrict__ w,
                                                               const uint16_t* __restrict__ bias,
                                                               uint16_t* __restrict__ out, float* __restrict__ partial,
                                                               int M, int Mpad, int N, int K, int kb_per_split,
                                                               int64_t x_stride, int64_t w_stride, int64_t out_stride) {
  __shared__ __attribute__((aligned(16))) char xs[2 * kDenseXTile];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  const int kblocks = K / 128;
  const int m0 = blockIdx.z * 64;
  const int kb0 = blockIdx.y * kb_per_split;
  const int nkb = min(kb_per_split, kblocks - kb0);
  const int n0 = (blockIdx.x * 4 + wid) * 16 * NS + i16;  // weight row (output column) of this lane, strip 0

  const int xr = threadIdx.x >> 2, xq = threadIdx.x & 3;
  // rows past M re-read row M - 1 (their products land in rows nobody reads): no exec-masked branch in the loop
  const uint16_t* xsrc = x + (int64_t)min(m0 + xr, M - 1) * x_stride + 32 * xq;
  // Activations are requested two k-blocks ahead: loads return in order, so the wait before a tile is
  // parked leaves only the weight loads issued after its request in flight, and that has to be the
  // whole weight prefetch depth.
  int4v xreg[2][4];
  auto x_fetch = [&](int4v(&xr4)[4], int kb) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      xr4[c] = *reinterpret_cast<const int4v*>(xsrc + 128 * kb + 8 * c);
  };
  auto x_park = [&](const int4v(&xr4)[4], int buf) {
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<int4v*>(xs + buf * kDenseXTile + dx_off(xr, 4 * xq + c)) = xr4[c];
  };

  // lane (i16, g) owns the 16-byte chunks g, 4+g, 8+g, 12+g of its weight row in every 128-k block
  // packed: strip-major, 2048 elements (4 KiB) per (strip, k-block), 512 per k-step, 8 per lane
  const uint16_t* wsrc[NS];
  constexpr int kbs = PACKED ? 2048 : 128;  // elements from one k-block to the next
  constexpr int cs = PACKED ? 512 : 32;     // ... from one k-step to the next
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int n = n0 + 16 * s;
    if (PACKED) {
      const int strip = min(n >> 4, (N >> 4) - 1);
      wsrc[s] = w + ((int64_t)strip * kblocks + kb0) * 2048 + 8 * lane;
    } else {
      wsrc[s] = w + (int64_t)(n < N ? n : 0) * w_stride + 128 * (int64_t)kb0 + 8 * g;
    }
  }
  // packed weights are read once, whole lines at a time: non-temporal; row-major lines are shared by four loads
  auto wload = [](const uint16_t* p) {
    return PACKED ? __builtin_nontemporal_load(reinterpret_cast<const int4v*>(p)) : *reinterpret_cast<const int4v*>(p);
  };
  int4v wq[PF][NS][4];
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (p < nkb) {
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          wq[p][s][c] = wload(wsrc[s] + kbs * p + cs * c);
    }

  float4v oacc[NS][4];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) oacc[s][mt] = float4v{0.f, 0.f, 0.f, 0.f};

  static_assert(PF % 2 == 0, "the activation registers alternate with the parity of the k-block");
  x_fetch(xreg[0], kb0);
  if (nkb > 1) x_fetch(xreg[1], kb0 + 1);
  x_park(xreg[0], 0);
  __syncthreads();

  for (int base = 0; base < nkb; base += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int j = base + p;
      if (j < nkb) {
        const int buf = j & 1;
        if (j + 2 < nkb) x_fetch(xreg[p & 1], kb0 + j + 2);  // block j's own registers were parked a step ago
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          // k-slots of step ks: k = 128*kb + 32*ks + 8*g + e  <->  activation chunk 4*ks + g
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int4v af = *reinterpret_cast<const int4v*>(xs + buf * kDenseXTile + dx_off(16 * mt + i16, 4 * ks + g));
#pragma unroll
            for (int s = 0; s < NS; ++s) oacc[s][mt] = DenseMma<T>::mma(af, wq[p][s][ks], oacc[s][mt]);
          }
        }
        if (j + PF < nkb) {
#pragma unroll
          for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              wq[p][s][c] = wload(wsrc[s] + kbs * (int64_t)(j + PF) + cs * c);
        }
        if (j + 1 < nkb) x_park(xreg[(p + 1) & 1], buf ^ 1);
        __syncthreads();
      }
    }
  }

  const bool direct = partial == nullptr;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int n = n0 + 16 * s;
    if (n >= N) continue;
    const float bv = (direct && bias) ? Conv<T>::to_f32(bias[n]) : 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * mt + 4 * g + r;
        if (direct) {
          if (m < M) out[(int64_t)m * out_stride + n] = Conv<T>::from_f32(oacc[s][mt][r] + bv);
        } else if (m < Mpad) {
          // written through (agent scope): megabytes of partials left dirty in the L2s would be written back at the
          // kernel's end, inside its duration (see the wide int4 GEMM)
          __hip_atomic_store(&partial[((int64_t)blockIdx.y * Mpad + m) * N + n], oacc[s][mt][r], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
      }
  }
}

// out[m][n] = sum over splits (in split order) of partial[s][m][n] (+ bias), eight outputs per thread
template <typename T>
__global__ void __launch_bounds__(256) dense_reduce_kernel(const float* __restrict__ partial, const uint16_t* __restrict__ bias,
                                                           uint16_t* __restrict__ out, int M, int Mpad, int N, int splits,
                                                           int64_t out_stride) {
  const int nv = N / 8;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)M * nv) return;
  const int m = (int)(idx / nv), n = (int)(idx - (int64_t)m * nv) * 8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int s0 = 0; s0 < splits; s0 += 4) {  // four splits' loads in flight
    float4v a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* pr = partial + ((int64_t)min(s0 + u, splits - 1) * Mpad + m) * N + n;
      a[u] = *reinterpret_cast<const float4v*>(pr);
      b[u] = *reinterpret_cast<const float4v*>(pr + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (s0 + u < splits) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j] += a[u][j];
          acc[4 + j] += b[u][j];
        }
      }
  }
  short8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (short)Conv<T>::from_f32(acc[j] + (bias ? Conv<T>::to_f32(bias[n + j]) : 0.f));
  *reinterpret_cast<short8*>(out + (int64_t)m * out_stride + n) = o;
}

// The same sum for a gate_up product, with the gated activation applied on the way out: out[m][j] =
// round(round(act(round(gate[m][j]))) * round(up[m][j])) for gate = columns [0, d), up = columns [d, 2d) -- the values
// the reduction above followed by act_and_mul stores (activation.cu:22-50: the activation returns T, then T * T).
template <typename T, int ACT>
__global__ void __launch_bounds__(256) dense_reduce_act_kernel(const float* __restrict__ partial, uint16_t* __restrict__ out,
                                                               int M, int Mpad, int N, int splits, int64_t out_stride) {
  const int d = N / 2, nv = d / 8;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)M * nv) return;
  const int m = (int)(idx / nv), n = (int)(idx - (int64_t)m * nv) * 8;
  float ga[8], ua[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ga[j] = ua[j] = 0.f;
  for (int s0 = 0; s0 < splits; s0 += 2) {  // two splits' loads in flight
    float4v a[2][2], b[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float* pr = partial + ((int64_t)min(s0 + u, splits - 1) * Mpad + m) * N + n;
      a[u][0] = *reinterpret_cast<const float4v*>(pr);
      a[u][1] = *reinterpret_cast<const float4v*>(pr + 4);
      b[u][0] = *reinterpret_cast<const float4v*>(pr + d);
      b[u][1] = *reinterpret_cast<const float4v*>(pr + d + 4);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (s0 + u < splits) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          ga[j] += a[u][0][j];
          ga[4 + j] += a[u][1][j];
          ua[j] += b[u][0][j];
          ua[4 + j] += b[u][1][j];
        }
      }
  }
  short8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float gv = Conv<T>::to_f32(Conv<T>::from_f32(ga[j])), uv = Conv<T>::to_f32(Conv<T>::from_f32(ua[j]));
    float av;
    if constexpr (ACT == 0) {
      av = gv / (1.0f + __expf(-gv));
    } else {
      const float k = 0.7978845608028654f;
      av = ACT == 1 ? 0.5f * gv * (1.0f + tanhf(k * (gv + 0.044715f * gv * gv * gv))) : 0.5f * gv * (1.0f + erff(gv * 0.7071067811865476f));
    }
    o[j] = (short)Conv<T>::from_f32(Conv<T>::to_f32(Conv<T>::from_f32(av)) * uv);
  }
  *reinterpret_cast<short8*>(out + (int64_t)m * out_stride + n) = o;
}

// row-major [N, K] -> packed: 16-byte piece (strip, kb, c, lane = 16 g + i16) = w[16 strip + i16][128 kb + 32 c + 8 g ..]
__global__ void __launch_bounds__(256) dense_repack_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ packed,
                                                           int N, int K, int64_t w_stride) {
  const int64_t piece = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int kblocks = K / 128;
  if (piece >= (int64_t)(N / 16) * kblocks * 256) return;
  const int lane = (int)(piece & 63), c = (int)((piece >> 6) & 3);
  const int64_t sk = piece >> 8;
  const int kb = (int)(sk % kblocks), strip = (int)(sk / kblocks);
  const int i16 = lane & 15, g = lane >> 4;
  *reinterpret_cast<int4v*>(packed + piece * 8) =
      *reinterpret_cast<const int4v*>(w + (int64_t)(16 * strip + i16) * w_stride + 128 * kb + 32 * c + 8 * g);
}

// K-splits of the two-launch form: enough blocks to cover the chip about twice, at least 512 of k per block
static int dense_splits(int M, int N, int K) {
  const char* env = getenv("SGLK_DENSE_SPLITS");
  const int kblocks = K / 128;
  if (env && atoi(env) > 0) return std::min(atoi(env), kblocks);
  if (N % 8 != 0) return 1;
  const int blocks = (N + 63) / 64 * ((M + 63) / 64);
  if (blocks >= 768) return 1;
  int splits = std::min((1024 + blocks - 1) / blocks, std::max(kblocks / 4, 1));
  return std::max(std::min(splits, 8), 1);  // measured on the 8B shapes: 8 x 1792 k beats 16 x 896 k on the down projection
}

}  // namespace sglk

using namespace sglk;

extern "C" {

// Dense split-K needs scratch; this entry point keeps the weights' whole K in one block when N gives
// enough blocks (lm_head: 1002 blocks of 128 rows) and otherwise runs unsplit as well: the strips are
// 16 rows of W, so even N = 4096 yields 64 blocks x 4 waves.  (The AWQ path owns the split-K machinery.)
int64_t sglk_dense_gemm_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 128 != 0) return 0;
  const int splits = dense_splits(M, N, K);
  return splits > 1 ? (int64_t)splits * ((M + 63) / 64 * 64) * N * (int64_t)sizeof(float) : 0;
}

int sglk_dense_gemm_nt_ws(const void* x, const void* w, const void* bias, void* out, int M, int N, int K,
                          int64_t x_stride, int64_t w_stride, int64_t out_stride, void* workspace,
                          int64_t workspace_bytes, int dtype, void* stream) {
  SGLK_CHECK(x && w && out, "dense_gemm_nt: null pointer");
  SGLK_CHECK(M > 0 && N > 0 && K > 0, "dense_gemm_nt: bad shape");
  SGLK_CHECK(K % 128 == 0, "dense_gemm_nt: K=%d must be a multiple of 128", K);
  const int splits = dense_splits(M, N, K);
  const int64_t need = sglk_dense_gemm_workspace_bytes(M, N, K);
  if (splits <= 1 || !workspace || workspace_bytes < need || !aligned16(workspace) || !aligned16(out) ||
      out_stride % 8 != 0)
    return sglk_dense_gemm_nt(x, w, bias, out, M, N, K, x_stride, w_stride, out_stride, dtype, stream);
  SGLK_CHECK(aligned16(x) && aligned16(w) && x_stride % 8 == 0 && w_stride % 8 == 0,
             "dense_gemm_nt: rows must be 16-byte aligned");
  const int mblocks = (M + 63) / 64, kblocks = K / 128;
  const int per = (kblocks + splits - 1) / splits;
  const int used = (kblocks + per - 1) / per;  // every split non-empty
  dim3 grid((N + 63) / 64, used, mblocks);
  hipStream_t s = (hipStream_t)stream;
  SGLK_DISPATCH_HALF(dtype, T, {
    hipLaunchKernelGGL((dense_gemm_nt_kernel<T, 1, 4>), grid, dim3(256), 0, s, (const uint16_t*)x, (const uint16_t*)w,
                       (const uint16_t*)nullptr, (uint16_t*)nullptr, (float*)workspace, M, mblocks * 64, N, K, per,
                       x_stride, w_stride, out_stride);
    const int64_t vecs = (int64_t)M * (N / 8);
    hipLaunchKernelGGL((dense_reduce_kernel<T>), dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, s,
                       (const float*)workspace, (const uint16_t*)bias, (uint16_t*)out, M, mblocks * 64, N, used,
                       out_stride);
  });
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_dense_repack(const void* w, void* packed, int N, int K, int64_t w_stride, void* stream) {
  SGLK_CHECK(w && packed, "dense_repack: null pointer");
  SGLK_CHECK(N > 0 && K > 0 && N % 16 == 0 && K % 128 == 0, "dense_repack: N=%d must be a multiple of 16, K=%d of 128", N, K);
  SGLK_CHECK(aligned16(w) && aligned16(packed) && w_stride % 8 == 0, "dense_repack: rows must be 16-byte aligned");
  const int64_t pieces = (int64_t)(N / 16) * (K / 128) * 256;
  hipLaunchKernelGGL(dense_repack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)w, (uint16_t*)packed, N, K, w_stride);
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_dense_gemm_packed(const void* x, const void* packed, const void* bias, void* out, int M, int N, int K,
                           int64_t x_stride, int64_t out_stride, int dtype, void* stream) {
  SGLK_CHECK(x && packed && out, "dense_gemm_packed: null pointer");
  SGLK_CHECK(M > 0 && N > 0 && K > 0, "dense_gemm_packed: bad shape");
  SGLK_CHECK(N % 16 == 0 && K % 128 == 0, "dense_gemm_packed: N=%d must be a multiple of 16, K=%d of 128", N, K);
  SGLK_CHECK(aligned16(x) && aligned16(packed) && x_stride % 8 == 0, "dense_gemm_packed: rows must be 16-byte aligned");
  const int mblocks = (M + 63) / 64;
  if (N >= 16 * 1024) {
    dim3 grid((N + 127) / 128, 1, mblocks);
    SGLK_DISPATCH_HALF(dtype, T,
                       hipLaunchKernelGGL((dense_gemm_nt_kernel<T, 2, 2, true>), grid, dim3(256), 0, (hipStream_t)stream,
                                          (const uint16_t*)x, (const uint16_t*)packed, (const uint16_t*)bias,
                                          (uint16_t*)out, (float*)nullptr, M, mblocks * 64, N, K, K / 128, x_stride,
                                          (int64_t)0, out_stride));
  } else {
    dim3 grid((N + 63) / 64, 1, mblocks);
    SGLK_DISPATCH_HALF(dtype, T,
                       hipLaunchKernelGGL((dense_gemm_nt_kernel<T, 1, 4, true>), grid, dim3(256), 0, (hipStream_t)stream,
                                          (const uint16_t*)x, (const uint16_t*)packed, (const uint16_t*)bias,
                                          (uint16_t*)out, (float*)nullptr, M, mblocks * 64, N, K, K / 128, x_stride,
                                          (int64_t)0, out_stride));
  }
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

// ---- split-K forms of the packed product, for weights of a few thousand rows (the decoder layers of a 16-bit model):
// one block per 64 columns would leave most CUs idle, so K is split and the fp32 slabs partial[split][M][N] are summed
// either by a second launch here (_ws) or by the kernel that consumes them (_partial: sglk_fused_add_rmsnorm_ex,
// sglk_rope_and_store_kv_partial add the splits up in split order and round once, as the int4 path's layer fusion does).
int sglk_dense_gemm_partial_splits(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 128 != 0 || N % 16 != 0) return 0;
  const int splits = dense_splits(M, N, K), kblocks = K / 128;
  const int per = (kblocks + splits - 1) / splits;
  return (kblocks + per - 1) / per;  // every split non-empty
}

static int dense_packed_split_launch(const void* x, const void* packed, float* partial, int M, int Mpad, int N, int K,
                                     int64_t x_stride, int dtype, hipStream_t s) {
  const int splits = dense_splits(M, N, K), kblocks = K / 128;
  const int per = (kblocks + splits - 1) / splits;
  const int used = (kblocks + per - 1) / per;
  static const int env_ns = getenv("SGLK_DENSE_NS") ? atoi(getenv("SGLK_DENSE_NS")) : 0;
  // two strips per wave (128 columns per block) halve the activation re-reads; not where that leaves a fractional
  // round of blocks on the 256 CUs below two full rounds (measured on the 8B shapes: o and down 256 blocks, gate_up
  // 672: faster; qkv 384: slower than 768 blocks of 64 columns)
  const int blocks2 = (N + 127) / 128 * used * ((M + 63) / 64);
  const bool two = env_ns ? env_ns == 2 : (N % 128 == 0 && (blocks2 % 256 == 0 || blocks2 >= 512));
  if (two) {
    dim3 grid((N + 127) / 128, used, (M + 63) / 64);
    SGLK_DISPATCH_HALF(dtype, T,
                       hipLaunchKernelGGL((dense_gemm_nt_kernel<T, 2, 2, true>), grid, dim3(256), 0, s, (const uint16_t*)x,
                                          (const uint16_t*)packed, (const uint16_t*)nullptr, (uint16_t*)nullptr, partial,
                                          M, Mpad, N, K, per, x_stride, (int64_t)0, (int64_t)0));
  } else {
    dim3 grid((N + 63) / 64, used, (M + 63) / 64);
    SGLK_DISPATCH_HALF(dtype, T,
                       hipLaunchKernelGGL((dense_gemm_nt_kernel<T, 1, 4, true>), grid, dim3(256), 0, s, (const uint16_t*)x,
                                          (const uint16_t*)packed, (const uint16_t*)nullptr, (uint16_t*)nullptr, partial,
                                          M, Mpad, N, K, per, x_stride, (int64_t)0, (int64_t)0));
  }
  SGLK_LAUNCH_CHECK();
  return used;
}

int sglk_dense_gemm_packed_partial(const void* x, const void* packed, float* partial, int M, int N, int K,
                                   int64_t x_stride, int dtype, void* stream) {
  SGLK_CHECK(x && packed && partial, "dense_gemm_packed_partial: null pointer");
  SGLK_CHECK(M > 0 && N > 0 && K > 0 && N % 16 == 0 && K % 128 == 0,
             "dense_gemm_packed_partial: N=%d must be a multiple of 16, K=%d of 128", N, K);
  SGLK_CHECK(aligned16(x) && aligned16(packed) && aligned16(partial) && x_stride % 8 == 0,
             "dense_gemm_packed_partial: rows must be 16-byte aligned");
  const int rc = dense_packed_split_launch(x, packed, partial, M, M, N, K, x_stride, dtype, (hipStream_t)stream);
  return rc < 0 ? rc : SGLK_OK;
}

int sglk_dense_gemm_packed_ws(const void* x, const void* packed, const void* bias, void* out, int M, int N, int K,
                              int64_t x_stride, int64_t out_stride, void* workspace, int64_t workspace_bytes, int dtype,
                              void* stream) {
  SGLK_CHECK(x && packed && out, "dense_gemm_packed: null pointer");
  SGLK_CHECK(M > 0 && N > 0 && K > 0, "dense_gemm_packed: bad shape");
  const int splits = (N % 16 == 0 && K % 128 == 0) ? sglk_dense_gemm_partial_splits(M, N, K) : 1;
  const int64_t need = (int64_t)splits * M * N * (int64_t)sizeof(float);
  if (splits <= 1 || !workspace || workspace_bytes < need || !aligned16(workspace) || !aligned16(out) || N % 8 != 0 ||
      out_stride % 8 != 0)
    return sglk_dense_gemm_packed(x, packed, bias, out, M, N, K, x_stride, out_stride, dtype, stream);
  SGLK_CHECK(aligned16(x) && aligned16(packed) && x_stride % 8 == 0, "dense_gemm_packed: rows must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int used = dense_packed_split_launch(x, packed, (float*)workspace, M, M, N, K, x_stride, dtype, s);
  if (used < 0) return used;
  const int64_t vecs = (int64_t)M * (N / 8);
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((dense_reduce_kernel<T>), dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, s,
                                        (const float*)workspace, (const uint16_t*)bias, (uint16_t*)out, M, M, N, used,
                                        out_stride));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_dense_gemm_packed_act(const void* x, const void* packed, void* out, int M, int N, int K, int64_t x_stride,
                               int64_t out_stride, int act, void* workspace, int64_t workspace_bytes, int dtype,
                               void* stream) {
  SGLK_CHECK(x && packed && out && workspace, "dense_gemm_packed_act: null pointer");
  SGLK_CHECK(M > 0 && N > 0 && K > 0 && N % 32 == 0 && K % 128 == 0,
             "dense_gemm_packed_act: N=%d must be a multiple of 32, K=%d of 128", N, K);
  SGLK_CHECK(act >= 0 && act <= 2, "dense_gemm_packed_act: act is 0 (silu), 1 (gelu tanh) or 2 (gelu)");
  const int splits = sglk_dense_gemm_partial_splits(M, N, K);
  SGLK_CHECK(splits >= 1 && workspace_bytes >= (int64_t)splits * M * N * (int64_t)sizeof(float),
             "dense_gemm_packed_act: workspace too small (%lld bytes needed)", (long long)splits * M * N * 4);
  SGLK_CHECK(aligned16(x) && aligned16(packed) && aligned16(workspace) && aligned16(out) && x_stride % 8 == 0 &&
                 out_stride % 8 == 0,
             "dense_gemm_packed_act: rows must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int used = dense_packed_split_launch(x, packed, (float*)workspace, M, M, N, K, x_stride, dtype, s);
  if (used < 0) return used;
  const int64_t vecs = (int64_t)M * (N / 16);
  const dim3 grid((unsigned)((vecs + 255) / 256));
#define SGLK_RA(A)                                                                                                  \
  hipLaunchKernelGGL((dense_reduce_act_kernel<T, A>), grid, dim3(256), 0, s, (const float*)workspace, (uint16_t*)out, M, \
                     M, N, used, out_stride)
  SGLK_DISPATCH_HALF(dtype, T, {
    if (act == 0) {
      SGLK_RA(0);
    } else if (act == 1) {
      SGLK_RA(1);
    } else {
      SGLK_RA(2);
    }
  });
#undef SGLK_RA
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_dense_gemm_nt(const void* x, const void* w, const void* bias, void* out, int M, int N, int K,
                       int64_t x_stride, int64_t w_stride, int64_t out_stride, int dtype, void* stream) {
  SGLK_CHECK(x && w && out, "dense_gemm_nt: null pointer");
  SGLK_CHECK(M > 0 && N > 0 && K > 0, "dense_gemm_nt: bad shape");
  SGLK_CHECK(K % 128 == 0, "dense_gemm_nt: K=%d must be a multiple of 128", K);
  SGLK_CHECK(aligned16(x) && aligned16(w) && x_stride % 8 == 0 && w_stride % 8 == 0,
             "dense_gemm_nt: rows must be 16-byte aligned");
  const int mblocks = (M + 63) / 64;
  // two strips per wave halve the activation re-reads; kept for wide outputs only, where the grid still
  // fills the chip several times over
  const char* env = getenv("SGLK_DENSE_NS");  // 1 / 2 forces the strips per wave (tests)
  const int force = env ? atoi(env) : 0;
  const bool two = force ? force == 2 : N >= 64 * 1024;
  if (two) {
    dim3 grid((N + 127) / 128, 1, mblocks);
    SGLK_DISPATCH_HALF(dtype, T,
                       hipLaunchKernelGGL((dense_gemm_nt_kernel<T, 2, 2>), grid, dim3(256), 0, (hipStream_t)stream,
                                          (const uint16_t*)x, (const uint16_t*)w, (const uint16_t*)bias,
                                          (uint16_t*)out, (float*)nullptr, M, mblocks * 64, N, K, K / 128, x_stride,
                                          w_stride, out_stride));
  } else {
    dim3 grid((N + 63) / 64, 1, mblocks);
    SGLK_DISPATCH_HALF(dtype, T,
                       hipLaunchKernelGGL((dense_gemm_nt_kernel<T, 1, 4>), grid, dim3(256), 0, (hipStream_t)stream,
                                          (const uint16_t*)x, (const uint16_t*)w, (const uint16_t*)bias,
                                          (uint16_t*)out, (float*)nullptr, M, mblocks * 64, N, K, K / 128, x_stride,
                                          w_stride, out_stride));
  }
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}
}
