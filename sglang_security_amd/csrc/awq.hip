// AWQ int4: dequantise (reference layout), one-time repack, and the fused dequant + MFMA GEMM.
//
// HBM layouts
//   reference (python/sglang/srt/layers/quantization/awq.py:98-151):
//     qweight int32 [K, N/8]  -- 8 nibbles per word along N, AWQ order {0,4,1,5,2,6,3,7}
//     scales  half  [K/G, N]
//     qzeros  int32 [K/G, N/8]
//   repacked for the matrix cores (sglk_awq_repack), same byte count as qweight:
//     packed int32 [K/128][N/16][64 lanes][4]   (k-block major: the strips that all waves stream at the
//     same time are contiguous, so concurrent reads spread over every HBM channel)
//     word d of lane (i = lane & 15, g = lane >> 4) in tile (nb, kb) holds the 8 weights
//     k = 128*kb + 32*d + 8*g + e (e = 0..7) of column n = 16*nb + i, with e = 2p in nibble p and
//     e = 2p+1 in nibble p+4.  ((w >> 4p) & 0x000F000F) | MAGIC is then, as it stands, register p of
//     the 16x16x32 MFMA operand holding (OFF + q) for k-slots 2p, 2p+1, OFF = 128 (bf16) / 1024 (fp16).
//     A wave streams one 16-column strip as 1-KiB pieces (16 bytes per lane), N/16 KiB apart.
//
// GEMM arithmetic, per 128-row block kb of K with scale group gq = 128*kb / G:
//     acc[m,n]  = sum_k x[m,k] * (OFF + q[k,n])                       (matrix cores, fp32)
//     out[m,n] += s[gq,n] * acc[m,n] - s[gq,n] * (OFF + z[gq,n]) * X[kb,m],   X[kb,m] = sum_k x[m,k]
// i.e. out = x @ ((q - z) * s) with the product (q - z) * s never rounded to half precision.
#include <cstdlib>

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
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int d = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    const int64_t tile = idx >> 8;
    const int nstrips = qcols / 2;
    const int nb = (int)(tile % nstrips);
    const int kb = (int)(tile / nstrips);
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
  static __device__ __forceinline__ float pair_sum(uint32_t v, float acc) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, v), __builtin_bit_cast(h2, 0x3C003C00u), acc, false);
  }
};

static constexpr int kGemmBM = 64;  // activation rows per block (4 MFMA row tiles)

// x tile in LDS: 16-byte chunk c of row r at slot c ^ (r & 15) (conflict-free row reads, as the K tile
// of the attention kernel)
__device__ __forceinline__ int x_off(int r, int c) { return r * 256 + ((c ^ (r & 15)) << 4); }

typedef __attribute__((address_space(3))) void* lds_void_ptr;
// LDS-DMA: 64 lanes x SIZE bytes land at lds_base + lane * SIZE; the source address is per lane
// (the size operand must be a literal: one helper per width)
__device__ __forceinline__ void glds16(const void* src, char* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (lds_void_ptr)lds_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* src, char* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (lds_void_ptr)lds_base, 4, 0, 0);
}

template <int NW, int TN, int WR>
struct GemmGeom {
  static constexpr int kXRing = 3;                 // x stages in LDS: one being multiplied, two in flight
  static constexpr int kXBytes = 64 * 256;         // 64 rows x 128 k x 2 B
  static constexpr int kXPieces = 16 / NW;         // 1-KiB x pieces per wave per stage
  static constexpr int kWBytes = NW * TN * 1024;   // NW waves x TN strips x 1 KiB
  static constexpr int kSzBytes = NW * 256;        // per wave: scale pairs + zero words of its columns
  static constexpr int kWStage = kWBytes + kSzBytes;
  static constexpr int kWOff = kXRing * kXBytes;
  static constexpr int kXsumOff = kWOff + WR * kWStage;  // float [2][64]
  static constexpr int kFlagOff = kXsumOff + 2 * 64 * 4;
  static constexpr int kLds = kFlagOff + 16;
  // DMA instructions of a wave that may stay outstanding when stage j must have landed:
  // w(j+WR-3), x(j+1), w(j+WR-2) in issue order behind x(j)
  static constexpr int kWaitCount = 2 * (TN + 1) + kXPieces;
  static_assert(WR >= 4, "weight ring too shallow for the wait arithmetic");
};

// Block = NW waves x TN strips of 16 columns = 16*NW*TN columns, 64 activation rows, kb_per_split 128-k blocks.
//
// Everything the block reads arrives by LDS-DMA (global_load_lds), one stage = one 128-k block:
//   x  : 64 x 128 tile, 16 one-KiB pieces (4 rows each) spread over the waves, XOR-swizzled through the
//        per-lane source address; L2-resident after first touch, so a ring of 3 stages covers its latency;
//   w  : TN one-KiB pieces per wave (its own strips), lane-linear; first-touch HBM reads whose latency under
//        load is microseconds, hence a separate, deeper ring of WR stages (WR-1 in flight);
//   sz : per wave one 256-byte piece per weight stage holding its columns' scales and zero words.
// A wave waits with a counted vmcnt (the younger stages stay outstanding), then one barrier per stage
// publishes the tile.  Compute per stage and wave: 4 k-steps x (4 A fragments by ds_read_b128, TN x (7 VALU
// to turn a packed word into an MFMA operand + 4 MFMA)).  Row tiles are visited in the order wid, wid+1, ..
// so the first A fragment of waves 0..3 is row tile wid: they add those rows up (v_dot2c) into the block sums
// X[m]; the scale / zero-point update  out += s * acc + c1 * X  of stage j runs after the next barrier, when X
// is complete.
// Split-K: each split stores its fp32 tile to a slab in register order (256-byte contiguous stores); the
// split that draws the last ticket of a tile sums the slabs and writes the 16-bit result.
template <typename T, int NW, int TN, int WR>
__global__ void __launch_bounds__(64 * NW, NW / 4) awq_gemm_kernel(
    const uint16_t* __restrict__ x, const int32_t* __restrict__ packed, const uint16_t* __restrict__ scales,
    const int32_t* __restrict__ qzeros, const uint16_t* __restrict__ bias, uint16_t* __restrict__ out,
    float* __restrict__ partial, int* __restrict__ tickets, int M, int N, int K, int group_size, int kb_per_split,
    int splits, int64_t x_stride, int64_t out_stride) {
  using G = GemmGeom<NW, TN, WR>;
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xsumS = reinterpret_cast<float*>(smem + G::kXsumOff);
  int* flag = reinterpret_cast<int*>(smem + G::kFlagOff);
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  const int kblocks = K / 128;
  const int m0 = blockIdx.z * kGemmBM;
  const int kb0 = blockIdx.y * kb_per_split;
  const int nkb = min(kb_per_split, kblocks - kb0);
  const int nb0 = (blockIdx.x * NW + wid) * TN;  // first 16-column strip of this wave
  const int n0 = 16 * nb0;
  const int qcols = N / 8;
  const int64_t nstrips = N / 16;

  // ---- per-lane DMA sources
  // x piece q of this wave covers rows 4*(kXPieces*wid+q) + (lane >> 4); rows past M re-read row M-1
  const uint16_t* xsrc[G::kXPieces];
#pragma unroll
  for (int q = 0; q < G::kXPieces; ++q) {
    const int row = 4 * (G::kXPieces * wid + q) + g;
    xsrc[q] = x + (int64_t)min(m0 + row, M - 1) * x_stride + 8 * (i16 ^ (row & 15));
  }
  const int4v* wsrc[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t)
    wsrc[t] = reinterpret_cast<const int4v*>(packed) + ((int64_t)kb0 * nstrips + nb0 + t) * 64 + lane;
  // sz piece: lanes [0, 8TN) fetch scale pairs, lanes [8TN, 10TN) zero words, the rest repeat lane 0
  const bool sz_is_zero = lane >= 8 * TN && lane < 10 * TN;
  const int sz_col = lane < 8 * TN ? 2 * lane : 0;

  if (threadIdx.x < 128) xsumS[threadIdx.x] = 0.f;  // before any DMA is in flight

  // loads past the split's end re-read its last block: every iteration issues the same number of DMAs
  auto issue_x = [&](int jj, int slot) {
    const int kb = kb0 + min(jj, nkb - 1);
    char* st = smem + slot * G::kXBytes;
#pragma unroll
    for (int q = 0; q < G::kXPieces; ++q) glds16(xsrc[q] + 128 * kb, st + (G::kXPieces * wid + q) * 1024);
    // the counted waits assume this issue order: pin it
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto issue_w = [&](int jj, int slot) {
    const int j = min(jj, nkb - 1);
    const int kb = kb0 + j;
    char* st = smem + G::kWOff + slot * G::kWStage;
#pragma unroll
    for (int t = 0; t < TN; ++t) glds16(wsrc[t] + (int64_t)j * nstrips * 64, st + (wid * TN + t) * 1024);
    const int gq = (128 * kb) / group_size;
    const void* szsrc = sz_is_zero ? (const void*)(qzeros + (int64_t)gq * qcols + (n0 >> 3) + (lane - 8 * TN))
                                   : (const void*)(scales + (int64_t)gq * N + n0 + sz_col);
    glds4(szsrc, st + G::kWBytes + wid * 256);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  float4v oacc[TN][4], acc[TN][4];
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      oacc[t][q] = float4v{0.f, 0.f, 0.f, 0.f};
      acc[t][q] = float4v{0.f, 0.f, 0.f, 0.f};
    }
  float sc[TN], c1[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) sc[t] = c1[t] = 0.f;
  int zsh[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) zsh[t] = 4 * kAwqOrd[(n0 + 16 * t + i16) & 7];

  // prologue in the steady-state issue order: w(0..WR-4), then [x0, w(WR-3)], [x1, w(WR-2)]
#pragma unroll
  for (int p = 0; p < WR - 3; ++p) issue_w(p, p);
  issue_x(0, 0);
  issue_w(WR - 3, WR - 3);
  issue_x(1, 1);
  issue_w(WR - 2, WR - 2);

  // per-lane LDS offset of the A fragment of k-step ks inside a row tile (the tile adds q * 4096, an immediate)
  int aoff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) aoff[ks] = x_off(i16, 4 * ks + g);
  int aown[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) aown[ks] = aoff[ks] + (wid & 3) * 4096;
  // v_and_or_b32 cannot take two literals: keep the OR constant in a VGPR and the mask in an SGPR (laundered
  // through empty asm so the compiler does not fold them back into literals and split the instruction)
  uint32_t or_const = AwqMagic<T>::kOr;
  uint32_t nib_mask = 0x000F000Fu;
  asm volatile("" : "+v"(or_const));
  asm volatile("" : "+s"(nib_mask));

  int wslot = 0;
  const int npad = (nkb + 2) / 3 * 3;  // the x ring slot is the unroll index; padded stages contribute zero
  for (int j0 = 0; j0 < npad; j0 += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int j = j0 + u;
      // stage j has landed once only the younger stages of this wave's DMA are outstanding
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G::kWaitCount) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // scale / zero-point update of stage j-1 (its block sums are complete now)
      {
        const float* xp = xsumS + ((j & 1) ^ 1) * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int mq = q;
          const float4v xs4 = *reinterpret_cast<const float4v*>(xp + 16 * mq + 4 * g);
#pragma unroll
          for (int t = 0; t < TN; ++t) {
            oacc[t][q] = __builtin_elementwise_fma(acc[t][q], float4v{sc[t], sc[t], sc[t], sc[t]}, oacc[t][q]);
            oacc[t][q] = __builtin_elementwise_fma(xs4, float4v{c1[t], c1[t], c1[t], c1[t]}, oacc[t][q]);
          }
        }
      }
      // the slots multiplied in iteration j-1 are free: refill them
      {
        issue_x(j + 2, (u + 2) % 3);
        int ws_next = wslot + WR - 1;
        ws_next = ws_next >= WR ? ws_next - WR : ws_next;
        issue_w(j + WR - 1, ws_next);
      }
      const bool live = j < nkb;
      const char* st = smem + u * G::kXBytes;
      const char* wt = smem + G::kWOff + wslot * G::kWStage;
      // this stage's per-column scale and zero
      {
        const char* szp = wt + G::kWBytes + wid * 256;
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int c = 16 * t + i16;
          const float s = live ? Conv<T>::to_f32(*reinterpret_cast<const uint16_t*>(szp + 2 * c)) : 0.f;
          const uint32_t zw = *reinterpret_cast<const uint32_t*>(szp + 32 * TN + 4 * (c >> 3));
          const float z = (float)((zw >> zsh[t]) & 0xF);
          sc[t] = s;
          c1[t] = -s * (AwqMagic<T>::kOff + z);
        }
      }
      int4v wv[TN];
#pragma unroll
      for (int t = 0; t < TN; ++t) wv[t] = *reinterpret_cast<const int4v*>(wt + (wid * TN + t) * 1024 + lane * 16);
      float xpart = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        int4v af[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) af[q] = *reinterpret_cast<const int4v*>(st + q * 4096 + aoff[ks]);
        // waves 0..3 add up row tile wid for the block sums: a second read of their own fragment keeps every
        // register index static
        {
          const int4v own = *reinterpret_cast<const int4v*>(st + aown[ks]);
#pragma unroll
          for (int e = 0; e < 4; ++e) xpart = AwqMagic<T>::pair_sum((uint32_t)own[e], xpart);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const uint32_t w = (uint32_t)wv[t][ks];
          int4v bf;
#pragma unroll
          for (int e = 0; e < 4; ++e) bf[e] = (int)(((w >> (4 * e)) & nib_mask) | or_const);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[t][q] = AwqMagic<T>::mma(af[q], bf, ks == 0 ? float4v{0.f, 0.f, 0.f, 0.f} : acc[t][q]);
        }
      }
      // block sums of row tile wid (waves 0..3): reduce over the four k-groups of lanes
      xpart += __shfl_xor(xpart, 16, 64);
      xpart += __shfl_xor(xpart, 32, 64);
      if (g == 0 && wid < 4) xsumS[(j & 1) * 64 + 16 * wid + i16] = xpart;
      wslot = wslot + 1 >= WR ? 0 : wslot + 1;
    }
  }
  // drain the look-ahead DMA (it targets this block's LDS), publish the last block sums, final update
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    const float* xp = xsumS + ((npad & 1) ^ 1) * 64;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mq = q;
      const float4v xs4 = *reinterpret_cast<const float4v*>(xp + 16 * mq + 4 * g);
#pragma unroll
      for (int t = 0; t < TN; ++t) oacc[t][q] += acc[t][q] * sc[t] + xs4 * c1[t];
    }
  }

  // ---- epilogue: lane (i16, g) holds rows 16*mq + 4*g + r of column n0 + 16*t + i16, mq = (q + wid) & 3
  if (splits == 1) {
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int n = n0 + 16 * t + i16;
      const float bv = bias ? Conv<T>::to_f32(bias[n]) : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mq = q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + 16 * mq + 4 * g + r;
          if (m < M) out[(int64_t)m * out_stride + n] = Conv<T>::from_f32(oacc[t][q][r] + bv);
        }
      }
    }
    return;
  }
  // split-K: slab[tile][split] holds the block's 64 x 64*TN fp32 tile in register order
  constexpr int kTileFloats = NW * TN * 16 * 64;
  const int tile = blockIdx.z * gridDim.x + blockIdx.x;
  float* slab = partial + ((int64_t)tile * splits + blockIdx.y) * kTileFloats;
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mq = q;
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(((wid * TN + t) * 4 + mq) * 4 + r) * 64 + lane] = oacc[t][q][r];
    }
  // publish: every wave drains its stores, then one lane releases at agent scope and draws a ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int ticket = __hip_atomic_fetch_add(&tickets[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == splits - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tickets[tile] = 0;  // leave the counter zero for the next launch
    }
    *flag = last;
  }
  __syncthreads();
  if (*flag == 0) return;
  const float* tslab = partial + (int64_t)tile * splits * kTileFloats;
  for (int v = threadIdx.x; v < kTileFloats / 4; v += NT) {
    float4v a = float4v{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s) a += *reinterpret_cast<const float4v*>(tslab + (int64_t)s * kTileFloats + 4 * v);
    // element 4v + e: lane = (4v + e) & 63, reg = (4v) >> 6 -> (w, t, mt, r)
    const int l0 = (4 * v) & 63, reg = (4 * v) >> 6;
    const int r = reg & 3, mt = (reg >> 2) & 3, wt = reg >> 4;  // wt = w * TN + t
    const int m = m0 + 16 * mt + 4 * (l0 >> 4) + r;
    const int n = 16 * ((blockIdx.x * NW) * TN + wt) + (l0 & 15);
    if (m < M) {
      short4v o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float b = bias ? Conv<T>::to_f32(bias[n + e]) : 0.f;
        o[e] = (short)Conv<T>::from_f32(a[e] + b);
      }
      *reinterpret_cast<short4v*>(out + (int64_t)m * out_stride + n) = o;
    }
  }
}

struct GemmPlan {
  int nw, tn, splits, kb_per_split, mblocks, tiles;
  int64_t slab_floats;
};

static GemmPlan plan_awq_gemm(int M, int N, int K) {
  static const int env_cfg = getenv("SGLK_AWQ_CFG") ? atoi(getenv("SGLK_AWQ_CFG")) : 0;  // 82, 42, 41
  static const int env_blocks = getenv("SGLK_AWQ_MIN_BLOCKS") ? atoi(getenv("SGLK_AWQ_MIN_BLOCKS")) : 224;
  GemmPlan p;
  p.mblocks = (M + kGemmBM - 1) / kGemmBM;
  const int kblocks = K / 128;
  int cfg = env_cfg;
  if (cfg == 82 && N % 256) cfg = 0;
  if (cfg == 42 && N % 128) cfg = 0;
  if (cfg == 0) cfg = (N % 128 == 0 && N >= 8192) ? 42 : 41;
  p.nw = cfg / 10;
  p.tn = cfg % 10;
  const int nblocks = N / (16 * p.nw * p.tn);
  p.tiles = nblocks * p.mblocks;
  // enough blocks to cover the 256 CUs, but keep >= 4 k-blocks (512 k) per split
  int splits = 1;
  while (p.tiles * splits < env_blocks && kblocks / (splits * 2) >= 4) splits *= 2;
  p.kb_per_split = (kblocks + splits - 1) / splits;
  p.splits = (kblocks + p.kb_per_split - 1) / p.kb_per_split;
  p.slab_floats = (int64_t)p.nw * p.tn * 16 * 64;
  return p;
}

static constexpr int kMaxTiles = 4096;  // ticket words at the head of the workspace

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
  int64_t bytes = (int64_t)kMaxTiles * 4;
  if (p.splits > 1) bytes += (int64_t)p.tiles * p.splits * p.slab_floats * 4;
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
  SGLK_CHECK(aligned16(workspace), "awq_gemm: workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const GemmPlan p = plan_awq_gemm(M, N, K);
  SGLK_CHECK(p.tiles <= kMaxTiles, "awq_gemm: too many output tiles (%d)", p.tiles);
  int* tickets = reinterpret_cast<int*>(workspace);
  float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + (int64_t)kMaxTiles * 4);
  dim3 grid(N / (16 * p.nw * p.tn), p.splits, p.mblocks);
#define SGLK_AWQ_LAUNCH(NW, TN, WR)                                                                               \
  {                                                                                                               \
    using GG = GemmGeom<NW, TN, WR>;                                                                              \
    static bool attr_set = false;                                                                                 \
    if (!attr_set) {                                                                                              \
      SGLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&awq_gemm_kernel<T, NW, TN, WR>),                \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, GG::kLds));                        \
      attr_set = true;                                                                                            \
    }                                                                                                             \
    hipLaunchKernelGGL((awq_gemm_kernel<T, NW, TN, WR>), grid, dim3(64 * NW), GG::kLds, s, (const uint16_t*)x,    \
                       packed, (const uint16_t*)scales, qzeros, (const uint16_t*)bias, (uint16_t*)out, partial,   \
                       tickets, M, N, K, group_size, p.kb_per_split, p.splits, x_stride, out_stride);             \
  }
  SGLK_DISPATCH_HALF(dtype, T, {
    if (p.nw == 8) {
      SGLK_AWQ_LAUNCH(8, 2, 6)
    } else if (p.tn == 2) {
      SGLK_AWQ_LAUNCH(4, 2, 6)
    } else {
      SGLK_AWQ_LAUNCH(4, 1, 6)
    }
  });
#undef SGLK_AWQ_LAUNCH
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}
}
