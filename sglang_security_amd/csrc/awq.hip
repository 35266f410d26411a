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
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "awq_common.h"

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

// dequantise from the re-laid-out copy (the prefill path when the reference-layout tensor has been dropped):
// one thread per packed word = 8 weights k0 .. k0+7 of one column
template <typename T>
__global__ void __launch_bounds__(256) awq_dequant_packed_kernel(const int32_t* __restrict__ packed,
                                                                 const uint16_t* __restrict__ scales,
                                                                 const int32_t* __restrict__ qzeros,
                                                                 uint16_t* __restrict__ out, int K, int N,
                                                                 int group_size) {
  const int qcols = N / 8, nstrips = N / 16;
  const int64_t total = (int64_t)K * qcols;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int d = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    const int64_t tile = idx >> 8;
    const int nb = (int)(tile % nstrips), kb = (int)(tile / nstrips);
    const int n = 16 * nb + (lane & 15);
    const int k0 = 128 * kb + 32 * d + 8 * (lane >> 4);
    const int gq = k0 / group_size;
    const uint32_t w = (uint32_t)packed[idx];
    const uint32_t z = ((uint32_t)qzeros[(int64_t)gq * qcols + (n >> 3)] >> (4 * kAwqOrd[n & 7])) & 0xF;
    const float s = Conv<T>::to_f32(scales[(int64_t)gq * N + n]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int pos = (e & 1) ? (e >> 1) + 4 : (e >> 1);
      const float diff = (float)((int)((w >> (4 * pos)) & 0xF) - (int)z);
      out[(int64_t)(k0 + e) * N + n] = Conv<T>::from_f32(diff * s);
    }
  }
}

// GPTQ int4 checkpoints (qweight int32 [K/8, N]: eight consecutive k of one column per word, nibble e = k % 8;
// qzeros int32 [K/G, N/8]: eight consecutive columns per word in plain order, stored minus one in the v1 format) into
// the layouts the kernels above stream: the packed weight word of (k-octet, column) is the GPTQ word of the same
// octet with its nibbles reordered, and the zero words take the AWQ nibble order with the stored offset added back.
// A zero point of 16 (stored nibble 15 in the v1 format) does not fit four bits: *overflow is set instead.
__global__ void __launch_bounds__(256) gptq_repack_kernel(const int32_t* __restrict__ qweight,
                                                          const int32_t* __restrict__ qzeros,
                                                          int32_t* __restrict__ packed, int32_t* __restrict__ zeros_out,
                                                          int* __restrict__ overflow, int K, int N, int groups,
                                                          int zero_offset) {
  const int qcols = N / 8, nstrips = N / 16;
  const int64_t total = (int64_t)K * qcols;
  const int64_t zwords = (int64_t)groups * qcols;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int d = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    const int64_t tile = idx >> 8;
    const int nb = (int)(tile % nstrips), kb = (int)(tile / nstrips);
    const int n = 16 * nb + (lane & 15);
    const int k0 = 128 * kb + 32 * d + 8 * (lane >> 4);
    const uint32_t src = (uint32_t)qweight[(int64_t)(k0 >> 3) * N + n];
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int pos = (e & 1) ? (e >> 1) + 4 : (e >> 1);
      w |= ((src >> (4 * e)) & 0xF) << (4 * pos);
    }
    packed[idx] = (int32_t)w;
    if (idx < zwords) {
      const uint32_t zsrc = (uint32_t)qzeros[idx];
      uint32_t z = 0;
      bool over = false;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t v = ((zsrc >> (4 * j)) & 0xF) + (uint32_t)zero_offset;
        over |= v > 15u;
        z |= (v & 0xF) << (4 * ((j >> 1) + 4 * (j & 1)));
      }
      zeros_out[idx] = (int32_t)z;
      if (over) atomicOr(overflow, 1);
    }
  }
}

// The same over whole tiles, with row-contiguous stores: a wave takes one 128-k group of 64 columns (four 1-KiB strip
// tiles), parks the packed words in LDS with the eight column words of every (k-octet, 8-column half) side by side,
// and each lane then builds 8 neighbouring columns of one row per step: two 16-byte LDS reads, one 16-byte store, a
// row's 128 bytes from eight adjacent lanes.  Needs N % 64 == 0 and group_size % 128 == 0.
template <typename T>
__global__ void __launch_bounds__(256) awq_dequant_packed_tile_kernel(const int32_t* __restrict__ packed,
                                                                      const uint16_t* __restrict__ scales,
                                                                      const int32_t* __restrict__ qzeros,
                                                                      uint16_t* __restrict__ out, int K, int N,
                                                                      int group_size) {
  __shared__ __attribute__((aligned(16))) uint32_t words[4][4 * 256];  // per wave: [strip][(g * 4 + d) * 16 + i]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nchunks = N / 64, nstrips = N / 16, qcols = N / 8;
  const int64_t ntiles = (int64_t)(K / 128) * nchunks;
  const int i16 = lane & 15, g4 = lane >> 4;
  const int cg = lane & 7, r = lane >> 3;  // output role: 8 columns 8 * cg .. + 7 of the chunk, rows r + 8 * it
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wid; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    const int kb = (int)(tile / nchunks), chunk = (int)(tile % nchunks);
    uint32_t* w = words[wid];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int4v v = *reinterpret_cast<const int4v*>(packed + (((int64_t)kb * nstrips + chunk * 4 + t) * 64 + lane) * 4);
#pragma unroll
      for (int d = 0; d < 4; ++d) w[t * 256 + (g4 * 4 + d) * 16 + i16] = (uint32_t)v[d];
    }
    const int gq = (128 * kb) / group_size;
    const int n0 = chunk * 64 + 8 * cg;
    const int4v sv = *reinterpret_cast<const int4v*>(scales + (int64_t)gq * N + n0);
    const uint32_t zw = (uint32_t)qzeros[(int64_t)gq * qcols + (n0 >> 3)];
    float s[8];
    int z[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s[j] = Conv<T>::to_f32((uint16_t)(((uint32_t)sv[j >> 1]) >> (16 * (j & 1))));
      z[j] = (int)((zw >> (4 * ((j >> 1) + 4 * (j & 1)))) & 0xF);  // AWQ nibble order {0, 4, 1, 5, 2, 6, 3, 7}
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int pos = (r & 1) ? (r >> 1) + 4 : (r >> 1);  // nibble of row e = r inside a word
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int g = it & 3, d = it >> 2;
      const uint32_t* src = w + (cg >> 1) * 256 + (g * 4 + d) * 16 + 8 * (cg & 1);
      const int4v a = *reinterpret_cast<const int4v*>(src);
      const int4v b = *reinterpret_cast<const int4v*>(src + 4);
      int4v o;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const uint32_t w0 = (uint32_t)(j < 4 ? a[j] : b[j - 4]), w1 = (uint32_t)(j < 4 ? a[j + 1] : b[j - 3]);
        const float v0 = (float)((int)((w0 >> (4 * pos)) & 0xF) - z[j]) * s[j];
        const float v1 = (float)((int)((w1 >> (4 * pos)) & 0xF) - z[j + 1]) * s[j + 1];
        o[j >> 1] = (int)((uint32_t)Conv<T>::from_f32(v0) | ((uint32_t)Conv<T>::from_f32(v1) << 16));
      }
      const int k = 128 * kb + 32 * d + 8 * g + r;
      *reinterpret_cast<int4v*>(out + (int64_t)k * N + n0) = o;
    }
    __builtin_amdgcn_wave_barrier();  // the next tile's words overwrite this wave's LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}


// x tile in LDS: 16-byte chunk c of row r at slot c ^ (r & 15) (conflict-free row reads, as the K tile
// of the attention kernel)
__device__ __forceinline__ int x_off(int r, int c) { return r * 256 + ((c ^ (r & 15)) << 4); }

typedef __attribute__((address_space(3))) void* lds_void_ptr;
// LDS-DMA: 64 lanes x SIZE bytes land at lds_base + lane * SIZE; the source address is per lane
// (the size operand must be a literal: one helper per width)
__device__ __forceinline__ void glds16(const void* src, char* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (lds_void_ptr)lds_base, 16, 0, 0);
}
// same with an instruction offset, which the hardware adds to BOTH addresses: pieces that are OFF bytes apart in LDS
// share one LDS base (one M0 write) when the source pointer is moved back by OFF
template <int OFF>
__device__ __forceinline__ void glds16_off(const void* src_minus_off, char* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_minus_off, (lds_void_ptr)lds_base,
                                   16, OFF, 0);
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

// ---------------------------------------------------------------------------------------------
// Wide kernel: 8 waves = 4 k-groups x 2 column chunks of 64, one block per CU (two waves per SIMD).
//
// The narrow kernel above spends ~15 instructions per matrix instruction on 16 columns per wave; issue slots,
// not HBM, bound it.  Here a wave owns 64 columns x 64 rows of one 128-k group per stage and multiplies with
// v_mfma_f32_32x32x16: one dequantised B operand (7 vector instructions for 8 weights per lane) feeds two matrix
// instructions of 32 cycles each, and a matrix instruction holds the vector issue port for 8 cycles whatever its
// size.  The four k-waves of a column chunk work on four consecutive groups (a stage is 512 k) and add their
// tiles up through LDS at the end, so a block needs no split-K partner for K <= 4096.
//
//   x   : stage = 4 tiles of 64 rows x 128 k (64 KiB) by LDS-DMA, two stages;
//   sz  : per wave and stage one 256-byte LDS-DMA piece (scale pairs + zero words), three stages;
//   w   : 4 x 16 bytes per lane straight into registers, two stages ahead (three register sets).  The
//         repacked layout serves the 32-column operand as it stands: lane (r = lane & 31, h = lane >> 5) of
//         column tile ct takes the 16-byte pieces of strip-lanes (r & 15) + 16 h and + 32 of strip 2 ct + (r >> 4),
//         whose word d holds k-step 2 d and 2 d + 1 respectively;
//   X   : block sums of x per (group, row), from the producer of x or computed here once per stage.
// The zero-point term -(OFF + z) * X of a group is one more k-step on the matrix cores (16-bit floats with eight
// significant bits, bf16): X split exactly into three such values in k-slots 0..2 of the A operand, -(OFF + z)
// (exact) in the same slots of B.  Types with a narrower exponent (fp16) keep that term on the vector unit.
// Per-wave issue order inside a stage is x(i+1), sz(i+2), w(i+2): waiting for x(i+1) leaves the five younger
// operations outstanding.
// ---------------------------------------------------------------------------------------------
struct Wide {
  static constexpr int kStageK = 512;
  static constexpr int kXStage = 4 * 64 * 256;          // 64 KiB
  static constexpr int kSzOff = 2 * kXStage;            // [3][8 waves][256 B]
  static constexpr int kXsumOff = kSzOff + 3 * 8 * 256;  // float [2][4][64] computed here, or [2][8 waves][64] fetched
  static constexpr int kFlagOff = kXsumOff + 2 * 8 * 64 * 4;
  static constexpr int kLds = kFlagOff + 16;
  static constexpr int kTileFloats = 64 * 128;
};

// Development builds only (tools/build_abl.sh): SGLK_ABL bit 64 compiles per-phase clock stamps into the wide kernel
// (tools/bench_gemm.py --fused --stamps reads them), the other bits remove one component of its main loop each
// (1 matrix instructions, 2 dequantisation, 4 LDS fragment reads, 16 rescale, 32 stage barrier, 128 activation DMA,
// 256 weight loads, 512 half of the activation DMA, 32768 plain instead of write-through output stores) for timing
// experiments.  The product build has SGLK_ABL == 0.
#ifndef SGLK_ABL
#define SGLK_ABL 0
#endif
#if SGLK_ABL & 1024
// per-wave cycle timeline of one steady-state stage (shader clock, s_memtime): [block][wave][16], word 15 = wall clock
__device__ unsigned int g_timeline[256 * 8 * 16];
#define SGLK_TL(k) tl[k] = (unsigned int)__builtin_readcyclecounter()
#else
#define SGLK_TL(k) \
  do {             \
  } while (0)
#endif
#if SGLK_ABL & 64
__device__ long long g_stamps[256 * 24];
#define SGLK_STAMP(k)                                                                             \
  do {                                                                                            \
    if (tid == 0 && blin < 256 && (k) < 24) g_stamps[blin * 24 + (k)] = (long long)wall_clock64(); \
  } while (0)
#else
#define SGLK_STAMP(k) \
  do {                \
  } while (0)
#endif
template <typename T>
struct ZpOnMatrixCores {
  static constexpr bool value = false;
};
template <>
struct ZpOnMatrixCores<bf16_t> {
  static constexpr bool value = true;
};

// XS_EXT: the block sums X[group][row] come from the kernel that produced x (xsum_in, row stride xs_stride) instead
// of being computed here: one barrier per stage instead of two.  partial_mode: every split stores its fp32 tile
// row-major to partial[split][M][N] and returns; the consumer kernel adds the splits up.
// MT: 32-row tiles of the 64-row block that are multiplied (batches of up to 32 rows skip the matrix work of the
// rows they do not have; staging, reduction and stores keep the 64-row shape, the skipped tile is zeros).
//
// One projection is a WideJob.  awq_gemm_wide_kernel runs one job, a block per (column tile, split, row block).
// awq_chain_kernel (CHAIN) runs a list of jobs in ONE launch of persistent blocks, job j + 1 reading what job j wrote:
// block b takes (column tile b % grid_x, split b / grid_x) of every job with nblocks > b.  What a block needs from
// the next job that does not depend on the previous one -- its first two stages of weights, scales and zeros -- is
// requested before the block finishes the current job (the in-block reduction, the stores, the hand-over), so the
// fill of job j + 1 runs under the drain of job j instead of behind a kernel boundary.  Hand-over: outputs are stored
// write-through, a block counts itself done once its stores are acknowledged (one of eight counters per job, spread
// over the memory channels), and a block starts the activation fetch of job j + 1 when the counters of job j add up
// to all its blocks.  Relaxed agent-scope accesses only, never a fence (a fence writes every L2 back).  The grid must
// be resident at once (one block per CU: the launcher checks the CU count), and the wait is bounded all the same: a
// block that sees no progress for two seconds raises the error word and goes on instead of hanging the GPU.
// A chain must not read an address that another block wrote earlier IN THE SAME LAUNCH unless nobody on the reader's
// XCD read it before the write (the L2s of the XCDs are not coherent with each other inside a launch): every buffer a
// job hands to the next is written once and read after.
struct WideJob {
  const uint16_t* x;
  const int32_t* packed;
  const uint16_t* scales;
  const int32_t* qzeros;
  const uint16_t* bias;
  uint16_t* out;
  float* partial;
  int* tickets;
  const float* xsum_in;
  float* xsum_out;
  int64_t x_stride, out_stride, xs_stride, xs_out_stride;
  int N, K, stages_per_split, splits, partial_mode, act_mode;
  int grid_x, nblocks;  // chain only: column tiles, blocks that take part (grid_x * splits)
};
static constexpr int kChainMaxJobs = 8;
struct ChainJobs {
  WideJob job[kChainMaxJobs];
  int njobs;
};
// sync words of a chain, 128 bytes apart: done[job][8], then the exit counter, then the error word
static constexpr int kChainSyncStride = 32;
static constexpr int kChainExitWord = kChainMaxJobs * 8 * kChainSyncStride;
static constexpr int kChainErrorWord = kChainExitWord + kChainSyncStride;
static constexpr int kChainSyncInts = kChainErrorWord + kChainSyncStride;
static constexpr uint64_t kChainWaitTicks = 200000000ull;  // 2 s of the 100 MHz wall clock

// Block-wide: returns once every block of job `job - 1` has counted itself done (job 0 waits for nothing).
__device__ __forceinline__ void chain_wait(int* __restrict__ sync, const WideJob* __restrict__ jobs, int job, char* smem) {
  if (job > 0 && threadIdx.x < 64) {
    const int target = jobs[job - 1].nblocks;
    int* const word = sync + ((job - 1) * 8 + ((int)threadIdx.x & 7)) * kChainSyncStride;
    const uint64_t t_start = wall_clock64();
    uint32_t polls = 0;
    while (true) {
      int v = (int)threadIdx.x < 8 ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      if (__builtin_amdgcn_readfirstlane(v) >= target) break;
      if ((++polls & 255u) == 0 && wall_clock64() - t_start > kChainWaitTicks) {
        if (threadIdx.x == 0) __hip_atomic_store(sync + kChainErrorWord, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// A block's last act: the block that leaves last clears the counters for the next launch.
__device__ __forceinline__ void chain_leave(int* __restrict__ sync, int njobs) {
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(sync + kChainExitWord, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (int)gridDim.x - 1) {
      for (int i = 0; i < njobs * 8; ++i)
        __hip_atomic_store(sync + i * kChainSyncStride, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sync + kChainExitWord, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <typename T, bool XS_EXT, int MT, bool CHAIN>
__device__ __forceinline__ void wide_body(const WideJob* __restrict__ jobs, int njobs, int M, int* __restrict__ sync,
                                          char* smem) {
  constexpr bool ZPM = ZpOnMatrixCores<T>::value;
  float* xsumS = reinterpret_cast<float*>(smem + Wide::kXsumOff);
  int* flag = reinterpret_cast<int*>(smem + Wide::kFlagOff);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kw = wid & 3, nw = wid >> 2;
  const int i16 = lane & 15, g = lane >> 4;
  const int r32 = lane & 31, h = lane >> 5;
  const int m0 = CHAIN ? 0 : blockIdx.z * kGemmBM;
  // ---- the job this block works on (set_job), under the names the single-job kernel's parameters had
  const uint16_t* x = nullptr;
  const int32_t* packed = nullptr;
  const uint16_t* scales = nullptr;
  const int32_t* qzeros = nullptr;
  const float* xsum_in = nullptr;
  int64_t x_stride = 0, xs_stride = 0;
  int N = 0, bx = 0, by = 0;
  int st0 = 0, nst = 0, n0 = 0, qcols = 0;
  int64_t nstrips = 0;
  uint32_t xoff[8];
  uint32_t wvoff = 0;
  [[maybe_unused]] const int blin = CHAIN ? (int)blockIdx.x : (int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
  auto set_job = [&](int jj) {
    const WideJob& J = jobs[jj];
    x = J.x;
    packed = J.packed;
    scales = J.scales;
    qzeros = J.qzeros;
    xsum_in = J.xsum_in;
    x_stride = J.x_stride;
    xs_stride = J.xs_stride;
    N = J.N;
    if constexpr (CHAIN) {
      bx = (int)blockIdx.x % J.grid_x;
      by = (int)blockIdx.x / J.grid_x;
    } else {
      bx = blockIdx.x;
      by = blockIdx.y;
    }
    const int total_stages = J.K / Wide::kStageK;
    st0 = by * J.stages_per_split;
    nst = min(J.stages_per_split, total_stages - st0);
    n0 = bx * 128 + nw * 64;
    nstrips = N / 16;
    qcols = N / 8;
    // ---- x DMA: wave w fills pieces 8w .. 8w+7 of a stage; piece p = (tile p >> 4, rows 4 * (p & 15) + g)
    // byte offset of this lane's 16-byte source chunk inside stage 0, biased so that "uniform base + 32-bit lane offset
    // + instruction offset" addresses it: piece j is issued with instruction offset 1024 * (j & 3) (which also moves
    // its LDS destination), the base is moved back by 3072 bytes, and the lane offset carries 3072 - 1024 * (j & 3)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = 8 * wid + j;
      const int row = 4 * (p & 15) + g;
      xoff[j] = (uint32_t)(2 * (min(m0 + row, M - 1) * x_stride + 128 * (p >> 4) + 8 * (i16 ^ (row & 15))) + 3072 -
                           1024 * (j & 3));
    }
    // ---- weights: strip t of group gq is 1 KiB at ((gq * nstrips) + n0/16 + t) * 1024 (see load_w)
    wvoff = (uint32_t)((((int64_t)kw * nstrips + (n0 >> 4) + (r32 >> 4)) * 64 + i16 + 16 * h) * 16);
  };
  // first job with work for this block at or after `from` (a single-job launch: job 0)
  auto next_job = [&](int from) {
    if constexpr (CHAIN) {
      int jj = from;
      while (jj < njobs && (int)blockIdx.x >= jobs[jj].nblocks) ++jj;
      return jj;
    } else {
      return from;
    }
  };
  int job = next_job(0);
  if (CHAIN && job >= njobs) {
    chain_leave(sync, njobs);
    return;
  }
  set_job(job);
  SGLK_STAMP(0);

  // four 1-KiB pieces as raw instructions in the scalar-base form (uniform 64-bit base in SGPRs, 32-bit lane offset,
  // instruction offset): through the builtin every piece costs two 64-bit vector adds for its address
  // Look-ahead requests past the split's end are issued with every lane switched off (live = 0): the instruction
  // still counts in vmcnt, so the counted waits hold for every stage, but nothing is fetched.  A short split (the o
  // projection runs one stage per block) would otherwise fetch up to three stages for each one it multiplies, through
  // an address unit that is the prologue's limit, and wait for them before its reduction.
  auto live_mask = [&](int stage) -> uint64_t {
    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane(((SGLK_ABL & 16384) || stage < nst) ? -1 : 0);
    uint64_t m = ((uint64_t)c << 32) | c;
    asm volatile("" : "+s"(m));  // in a register pair, never a 64-bit literal
    return m;
  };
  auto glds_quad = [&](const char* src, char* dst, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, uint64_t live) {
    const uint32_t lds_addr = (uint32_t)(uintptr_t)(lds_void_ptr)dst;
    uint64_t saved;
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_and_saveexec_b64 %0, %7\n\t"
        "global_load_lds_dwordx4 %2, %6\n\t"
        "global_load_lds_dwordx4 %3, %6 offset:1024\n\t"
        "global_load_lds_dwordx4 %4, %6 offset:2048\n\t"
        "global_load_lds_dwordx4 %5, %6 offset:3072\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(saved)
        : "s"(lds_addr), "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(src), "s"(live)
        : "memory", "m0", "scc");
  };
  // one dword per lane to lds_dst + 4 * lane, per-lane 64-bit source address
  auto glds4_live = [&](const void* src, char* dst, uint64_t live) {
    const uint32_t lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)(lds_void_ptr)dst);
    uint64_t saved;
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_and_saveexec_b64 %0, %3\n\t"
        "global_load_lds_dword %2, off\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(saved)
        : "s"(lds_addr), "v"(src), "s"(live)
        : "memory", "m0", "scc");
  };
  auto issue_x = [&](int stage, int buf) {
    const char* src = reinterpret_cast<const char*>(x + (int64_t)(st0 + min(stage, nst - 1)) * Wide::kStageK) - 3072;
    char* dst = smem + buf * Wide::kXStage + wid * 8 * 1024;
    // eight 1-KiB pieces, two LDS bases (the instruction offset reaches 4095)
    glds_quad(src, dst, xoff[0], xoff[1], xoff[2], xoff[3], live_mask(0));
    glds_quad(src, dst + 4096, xoff[4], xoff[5], xoff[6], xoff[7], live_mask(0));
    if constexpr (XS_EXT) {
      // this wave's group of the stage: 64 row sums, one per lane, into the wave's own slot
      const int gq = (st0 + min(stage, nst - 1)) * 4 + kw;
      glds4(xsum_in + (int64_t)gq * xs_stride + min(m0 + lane, M - 1), smem + Wide::kXsumOff + (buf * 8 + wid) * 256);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // the same in two halves, issued between the k-steps of the stage being multiplied: the CU's address unit takes 16
  // cycles per 1-KiB instruction, so eight waves issuing everything at the top of a stage stall each other
  auto issue_x_half = [&](int stage, int buf, int half) {
    const char* src = reinterpret_cast<const char*>(x + (int64_t)(st0 + min(stage, nst - 1)) * Wide::kStageK) - 3072;
    char* dst = smem + buf * Wide::kXStage + wid * 8 * 1024 + half * 4096;
    const uint64_t live = live_mask(stage);
    glds_quad(src, dst, xoff[4 * half + 0], xoff[4 * half + 1], xoff[4 * half + 2], xoff[4 * half + 3], live);
    if constexpr (XS_EXT) {
      if (half == 1) {
        const int gq = (st0 + min(stage, nst - 1)) * 4 + kw;
        glds4_live(xsum_in + (int64_t)gq * xs_stride + min(m0 + lane, M - 1), smem + Wide::kXsumOff + (buf * 8 + wid) * 256,
                   live);
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- sz DMA: lanes [0, 32) scale pairs of the wave's 64 columns, [32, 40) zero words, the rest repeat
  const bool sz_is_zero = lane >= 32 && lane < 40;
  const int sz_col = lane < 32 ? 2 * lane : 0;
  auto issue_sz = [&](int stage, int slot) {
    const int gq = (st0 + min(stage, nst - 1)) * 4 + kw;
    const void* src = sz_is_zero ? (const void*)(qzeros + (int64_t)gq * qcols + (n0 >> 3) + (lane - 32))
                                 : (const void*)(scales + (int64_t)gq * N + n0 + sz_col);
    glds4_live(src, smem + Wide::kSzOff + (slot * 8 + wid) * 256, live_mask(stage));
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- weights: strip t of group gq is 1 KiB at ((gq * nstrips) + n0/16 + t) * 1024.  Load 2 ct + j (j = 0, 1) of a
  // stage fetches, for column tile ct, the piece of strip-lane (r & 15) + 16 h + 32 j of strip 2 ct + (r >> 4): word d of
  // it is this lane's B operand source of k-step 2 d + j.  Issued as raw instructions so that the compiler neither
  // counts them nor waits for them: the counted vmcnt at each stage boundary (bound to the registers through its
  // operands) is the only wait.
  auto load_w = [&](int stage, int4v(&w)[4]) {
    const char* src = reinterpret_cast<const char*>(packed) + (int64_t)(st0 + min(stage, nst - 1)) * 4 * nstrips * 1024;
    uint64_t saved;
    const uint64_t live = live_mask(stage);
    asm volatile(
        "s_and_saveexec_b64 %4, %7\n\t"
        "global_load_dwordx4 %0, %5, %6\n\t"
        "global_load_dwordx4 %1, %5, %6 offset:512\n\t"
        "global_load_dwordx4 %2, %5, %6 offset:2048\n\t"
        "global_load_dwordx4 %3, %5, %6 offset:2560\n\t"
        "s_mov_b64 exec, %4"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&s"(saved)
        : "v"(wvoff), "s"(src), "s"(live)
        : "memory", "scc");
    __builtin_amdgcn_sched_barrier(0);
  };
  // x(next) has landed once only sz and w issued after it (five operations) are outstanding; the weights named
  // here were issued before it, so they have landed too
  auto stage_wait = [&](int4v(&w)[4]) {
    asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3])::"memory");
  };

  // One accumulator set: R[ct][rt] (column tile, row tile; lane = column, registers = rows) holds the running sum
  // in units of the current group's scale,
  //   R = sum_{g' <= g} (s_g' / s_g) * (x . (OFF + q_g') - (OFF + z_g') * X_g'),
  // so the matrix cores add straight into it; moving to the next group multiplies R by s_g / s_{g+1}.
  // A column whose scale is zero keeps its previous unit (see below); non-finite scales propagate.
  float16v R[2][2];
  float s_cur[2] = {1.f, 1.f};
  auto reset_acc = [&]() {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) R[ct][rt][e] = 0.f;
    s_cur[0] = s_cur[1] = 1.f;
  };
  // AWQ order {0,4,1,5,2,6,3,7} in arithmetic: a table lookup here would be a load the stage pipeline waits for
  const int zsh = 4 * (((r32 & 7) >> 1) + 4 * (r32 & 1));
  // A operand of k-step s, row tile rt: 16 bytes at row 32 rt + r, chunk 2 s + h of this wave's x tile
  int aoff[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) aoff[s] = kw * 16384 + x_off(r32, 2 * s + h);
  uint32_t or_const = AwqMagic<T>::kOr;
  uint32_t nib_mask = 0x000F000Fu;
  asm volatile("" : "+v"(or_const));
  asm volatile("" : "+s"(nib_mask));

  // block sums of the landed x stage: thread (row = tid >> 3, tile = (tid & 7) >> 1, half = tid & 1)
  auto xsum_pass = [&](int buf) {
    const int row = tid >> 3, tile = (tid & 7) >> 1, half = tid & 1;
    const char* tp = smem + buf * Wide::kXStage + tile * 16384;
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int4v v = *reinterpret_cast<const int4v*>(tp + x_off(row, 8 * half + j));
#pragma unroll
      for (int e = 0; e < 4; ++e) part = AwqMagic<T>::pair_sum((uint32_t)v[e], part);
    }
    part += __shfl_xor(part, 1, 64);
    if (half == 0) xsumS[(buf * 4 + tile) * 64 + row] = part;
  };

  // Per stage and wave, before its k-steps: read the group's scales and zeros, move R to the new unit, add the
  // zero-point term.  Everything it reads was fetched by this wave's own DMA (scale / zero piece, block sums), so it
  // needs no barrier, only the wave's own counted wait: waves 4-7 run it at the END of the previous stage (after that
  // wait, before the barrier) and waves 0-3 at the start of the stage.  SIMD partners (waves w and w + 4) are then out
  // of phase: one multiplies while the other rescales, instead of both idling the matrix pipe after each barrier.
  uint32_t msk[2] = {nib_mask, nib_mask}, orc[2] = {or_const, or_const};  // dequantisation constants per column tile
  auto prep = [&](int ws, int buf) {
    int4v bz[2];   // B operand of the zero-point k-step
    float nzo[2];  // -(OFF + z), vector-unit form
    const char* szp = smem + Wide::kSzOff + (ws * 8 + wid) * 256;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int c = 32 * ct + r32;
      const float s_in = Conv<T>::to_f32(*reinterpret_cast<const uint16_t*>(szp + 2 * c));
      const uint32_t zw = *reinterpret_cast<const uint32_t*>(szp + 128 + 4 * (c >> 3));
      // A zero (or denormal) scale cannot become the unit of R: that column keeps its unit and this group
      // contributes nothing to it (its B operand and zero-point term are zeroed).
      const bool ok = fabsf(s_in) >= 1.17549435e-38f;
      const float sg = ok ? s_in : s_cur[ct];
      const float ratio = s_cur[ct] * __builtin_amdgcn_rcpf(sg);
      s_cur[ct] = sg;
      const uint32_t z = (zw >> zsh) & 0xF;
      msk[ct] = ok ? nib_mask : 0u;
      orc[ct] = ok ? or_const : 0u;
      if constexpr (ZPM) {
        // -(OFF + z) as the 16-bit pattern of the operand type: sign bit on the dequantisation constant
        const uint32_t nzb = ((AwqMagic<T>::kOr & 0xFFFFu) | 0x8000u) | z;
        const bool okh = ok && h == 0;  // k-slots 0..2 belong to the lower half-wave
        bz[ct] = int4v{okh ? (int)(nzb * 0x10001u) : 0, okh ? (int)nzb : 0, 0, 0};
      } else {
        nzo[ct] = ok ? -(AwqMagic<T>::kOff + (float)z) : 0.f;
      }
#pragma unroll
      for (int rt = 0; rt < MT; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if constexpr ((SGLK_ABL & 8192) != 0) {
            // one plain multiply per register: the compiler otherwise pairs them into packed fp32 multiplies
            float v = R[ct][rt][e];
            asm("v_mul_f32 %0, %1, %0" : "+v"(v) : "v"(ratio));
            R[ct][rt][e] = v;
          } else if constexpr (!(SGLK_ABL & 16)) {
            R[ct][rt][e] *= ratio;
          }
        }
      if constexpr (SGLK_ABL & 16) asm volatile("" ::"v"(ratio));
    }
    const float* xp = xsumS + (XS_EXT ? (buf * 8 + wid) : (buf * 4 + kw)) * 64;
    if constexpr (ZPM) {
      // zero-point k-step: X[row] = a + b + c exactly, three values of the operand type
#pragma unroll
      for (int rt = 0; rt < MT; ++rt) {
        const float X = xp[32 * rt + r32];
        const uint32_t a = __float_as_uint(X) & 0xFFFF0000u;
        const float r1 = X - __uint_as_float(a);
        const uint32_t b = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(b);
        const int4v ax = int4v{(int)((a >> 16) | b), (int)(__float_as_uint(r2) >> 16), 0, 0};
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) R[ct][rt] = AwqMagic<T>::mma32(ax, bz[ct], R[ct][rt]);
      }
    } else {
      // zero-point term with this group's block sums on the vector unit: register e of a tile is row
      // (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
      for (int rt = 0; rt < MT; ++rt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4v xs4 = *reinterpret_cast<const float4v*>(xp + 32 * rt + 8 * j + 4 * h);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) R[ct][rt][4 * j + e] = fmaf(xs4[e], nzo[ct], R[ct][rt][4 * j + e]);
        }
    }
    asm volatile("" ::: "memory");
  };
#if SGLK_ABL & 4096
  const bool late = false;
#else
  const bool late = XS_EXT && wid >= 4;
#endif

  int4v wreg[3][4];
  if constexpr (CHAIN) {
    // what the first job needs and no other job writes: its first two stages of scales, zeros and weights
    issue_sz(0, 0);
    load_w(0, wreg[0]);
    issue_sz(1, 1);
    load_w(1, wreg[1]);
  }
  while (true) {
  reset_acc();
  msk[0] = msk[1] = nib_mask;
  orc[0] = orc[1] = or_const;
  if constexpr (CHAIN) {
    // the activations of this job are what the job before it wrote: wait until all of its blocks have counted
    // themselves done, then fetch the first stage; everything older (the prefetched weights) lands with it
    chain_wait(sync, jobs, job, smem);
    issue_x(0, 0);
    SGLK_STAMP(19);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(wreg[0][0]), "+v"(wreg[0][1]), "+v"(wreg[0][2]), "+v"(wreg[0][3]),
                 "+v"(wreg[1][0]), "+v"(wreg[1][1]), "+v"(wreg[1][2]), "+v"(wreg[1][3])::"memory");
  } else {
  // prologue: sz(0), w(0), x(0), sz(1), w(1); then X(0)
  issue_sz(0, 0);
  load_w(0, wreg[0]);
  issue_x(0, 0);
  issue_sz(1, 1);
  load_w(1, wreg[1]);
  SGLK_STAMP(19);
  if constexpr ((SGLK_ABL & 64) != 0) {
    if constexpr (XS_EXT) asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    SGLK_STAMP(20);
  }
  stage_wait(wreg[0]);  // sz(0), w(0), x(0) landed; sz(1) and w(1) fly on
  }
  SGLK_STAMP(21);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if constexpr (!XS_EXT) {
    xsum_pass(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  if (late) prep(0, 0);
  SGLK_STAMP(1);
  // register set = unroll index mod 3, x buffer = unroll index mod 2: unrolled six times, every index is static
  const int npad = (nst + 5) / 6 * 6;
  for (int i0 = 0; i0 < npad; i0 += 6) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = i0 + u;
      constexpr int kUnused = 0;
      (void)kUnused;
      const int buf = u & 1;
      const int ws = u % 3;
      [[maybe_unused]] unsigned int tl[16];
#if SGLK_ABL & 1024
      tl[14] = (unsigned int)wall_clock64();
#endif
      SGLK_TL(0);
      {
        const char* xt = smem + buf * Wide::kXStage;
        if (!late) prep(ws, buf);
        SGLK_TL(1);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          int4v af[MT];
#pragma unroll
          for (int rt = 0; rt < MT; ++rt) {
            if constexpr (SGLK_ABL & 4) {
              af[rt] = int4v{aoff[s], aoff[s] + rt, aoff[s] + 2, aoff[s] + 3};
            } else {
              af[rt] = *reinterpret_cast<const int4v*>(xt + rt * 8192 + aoff[s]);
            }
          }
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            const uint32_t w = (uint32_t)wreg[ws][2 * ct + (s & 1)][s >> 1];
            int4v bf;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if constexpr (SGLK_ABL & 2) {
                bf[e] = (int)(w | orc[ct]);
              } else {
                bf[e] = (int)(((w >> (4 * e)) & msk[ct]) | orc[ct]);
              }
            }
#pragma unroll
            for (int rt = 0; rt < MT; ++rt) {
              if constexpr (SGLK_ABL & 1) {
                asm volatile("" ::"v"(af[rt]), "v"(bf));
              } else {
                R[ct][rt] = AwqMagic<T>::mma32(af[rt], bf, R[ct][rt]);
              }
            }
          }
          // this stage's memory instructions go out between its k-steps, in the order the counted wait assumes:
          // x(i+1) in two halves, then sz(i+2) and w(i+2)
          if constexpr ((SGLK_ABL & 128) != 0) {
            if (s == 4) {
              issue_sz(i + 2, (u + 2) % 3);
              if constexpr ((SGLK_ABL & 256) != 0) load_w(0, wreg[(u + 2) % 3]); else load_w(i + 2, wreg[(u + 2) % 3]);
            }
          } else if (s == 0 || (s == 2 && !(SGLK_ABL & 512))) {
            if constexpr (SGLK_ABL & 8) {
              issue_x_half(0, buf ^ 1, s >> 1);  // the same stage again: L1-resident after the first round
            } else {
              issue_x_half(i + 1, buf ^ 1, s >> 1);
            }
          } else if (s == 4) {
            issue_sz(i + 2, (u + 2) % 3);
            load_w(i + 2, wreg[(u + 2) % 3]);
          }
          SGLK_TL(2 + s);
        }
      }
      SGLK_STAMP(2 + i);
      if (i + 1 >= nst) goto stages_done;  // block-uniform: the look-ahead loads are drained below
      if constexpr ((SGLK_ABL & 64) != 0) {
        if (i == 2) {
          if constexpr (XS_EXT) asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
          SGLK_STAMP(16);
        }
      }
      SGLK_TL(10);
      stage_wait(wreg[(u + 1) % 3]);  // x(i+1) and w(i+1) landed; sz(i+2), w(i+2) fly on
      if (late) prep((u + 1) % 3, buf ^ 1);
      SGLK_TL(11);
      if (i == 2) SGLK_STAMP(17);
      if constexpr (!(SGLK_ABL & 32)) __builtin_amdgcn_s_barrier();
      SGLK_TL(12);
      if (i == 2) SGLK_STAMP(18);
      asm volatile("" ::: "memory");
#if SGLK_ABL & 1024
      if (i == 3 && lane == 0 && blin < 256) {
        tl[15] = (unsigned int)wall_clock64();
        for (int k = 0; k < 16; ++k) g_timeline[(blin * 8 + wid) * 16 + k] = tl[k];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
#endif
      if constexpr (!XS_EXT) {
        xsum_pass(buf ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
  }
stages_done:
  // drain the look-ahead loads (the DMA targets this block's LDS) before the stages are reused
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  SGLK_STAMP(12);
  // ---- what the epilogue needs of this job; in a chain the next job's set-up follows at once
  const WideJob& EJ = jobs[job];
  const int ebx = bx, eby = by, eN = N;
  const int done_job = job;
  if constexpr (CHAIN) {
    job = next_job(job + 1);
    if (job < njobs) {
      set_job(job);
      issue_sz(0, 0);
      load_w(0, wreg[0]);
      issue_sz(1, 1);
      load_w(1, wreg[1]);
    }
  }
  [&]() {
  uint16_t* const out = EJ.out;
  float* const partial = EJ.partial;
  int* const tickets = EJ.tickets;
  const uint16_t* const bias = EJ.bias;
  float* const xsum_out = EJ.xsum_out;
  const int64_t out_stride = EJ.out_stride, xs_out_stride = EJ.xs_out_stride;
  const int splits = EJ.splits, partial_mode = EJ.partial_mode, act_mode = EJ.act_mode;
  const int N = eN;  // (shadows the job variable, which may already name the next job)
  // ---- add the four k-waves of each column chunk up through LDS.  red[wave][tile = 2 ct + rt][j][lane] holds
  // registers 4 j .. 4 j + 3 of a wave's tile (128 KiB: the x stages); wave w then finishes the block's tile
  // (column tile cc = w >> 1 of four, row tile w & 1).
  float4v* red = reinterpret_cast<float4v*>(smem);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4v v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = R[ct][rt][4 * j + e] * s_cur[ct];
        red[((wid * 4 + 2 * ct + rt) * 4 + j) * 64 + lane] = v;
      }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  SGLK_STAMP(22);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  SGLK_STAMP(23);
  // sum of the four k-waves' registers 4 j .. 4 j + 3 of tile (column tile cc of the block, row tile rt), in k order
  auto ksum = [&](int cc, int rt, int j) {
    float4v a = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) a += red[((((cc >> 1) * 4 + k2) * 4 + 2 * (cc & 1) + rt) * 4 + j) * 64 + lane];
    return a;
  };
  if (act_mode) {
    // Gated-activation epilogue (single split only).  The tile's first 64 columns are gate columns, its last 64 the
    // matching up columns (the caller interleaved the weight that way), so out[m, 64*tile + j] =
    // round(round(silu(round(gate))) * round(up)): the values the unfused GEMM + silu_and_mul sequence stores.
    // Wave w finishes registers 8 hf .. 8 hf + 7 (hf = w & 1) of output tile (column tile cc2 = w >> 2, row tile
    // rt = (w >> 1) & 1); the block's row sums (for the next projection's zero-point term) go to xsum_out, two
    // blocks per 128-column group, by one atomic add per row and block (two addends: order-independent).
    float* rs = reinterpret_cast<float*>(smem + Wide::kXsumOff);  // [2 column tiles][64 rows]: the block-sum slots are idle now
    const int cc2 = wid >> 2, rt = (wid >> 1) & 1, hf = wid & 1;
    const int ncol = ebx * 64 + 32 * cc2 + r32;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * hf + jj;
      const float4v gs = ksum(cc2, rt, j), us = ksum(cc2 + 2, rt, j);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gb = Conv<T>::to_f32(Conv<T>::from_f32(gs[e]));
        const float ub = Conv<T>::to_f32(Conv<T>::from_f32(us[e]));
        const float sv = Conv<T>::to_f32(Conv<T>::from_f32(gb / (1.0f + __expf(-gb))));
        const uint16_t ob = Conv<T>::from_f32(sv * ub);
        const int row = 32 * rt + e + 8 * j + 4 * h;
        if constexpr ((SGLK_ABL & 32768) != 0) {
          if (m0 + row < M) out[(int64_t)(m0 + row) * out_stride + ncol] = ob;
        } else {  // write-through, as the partials below
          if (m0 + row < M)
            __hip_atomic_store(&out[(int64_t)(m0 + row) * out_stride + ncol], ob, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (xsum_out) {
          float v = Conv<T>::to_f32(ob);
#pragma unroll
          for (int o2 = 16; o2 > 0; o2 >>= 1) v += __shfl_xor(v, o2, 32);
          if (r32 == 0) rs[cc2 * 64 + row] = v;
        }
      }
    }
    if (xsum_out) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (tid < 64 && m0 + tid < M)
        atomicAdd(&xsum_out[(int64_t)(ebx >> 1) * xs_out_stride + m0 + tid], rs[tid] + rs[64 + tid]);
    }
    return;
  }
  SGLK_STAMP(13);
  const int cc = wid >> 1, frt = wid & 1;
  float4v fin[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) fin[j] = ksum(cc, frt, j);
  // lane (r, h) of wave w now holds rows 32 frt + e + 8 j + 4 h (fin[j][e]) of column ebx * 128 + 32 cc + r
  const int n = ebx * 128 + 32 * cc + r32;
  const int mrow = m0 + 32 * frt + 4 * h;
  if constexpr ((SGLK_ABL & 64) != 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(fin[0]), "v"(fin[1]), "v"(fin[2]), "v"(fin[3]) : "memory");
    SGLK_STAMP(15);
  }
  if (partial_mode) {
    float* dst = partial + (int64_t)eby * M * N;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = mrow + 8 * j + e;
        // write-through (agent scope): the partials are the next kernel's input, and 6-8 MB of them left dirty in
        // the L2s lengthen this kernel's end by the write-back (measured 0.6-0.7 us per launch); non-temporal
        // stores shorten it as much but make the consumer's reads slower
        if constexpr ((SGLK_ABL & 32768) != 0) {
          if (m < M) dst[(int64_t)m * N + n] = fin[j][e];
        } else {
          if (m < M) __hip_atomic_store(&dst[(int64_t)m * N + n], fin[j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SGLK_STAMP(14);
    return;
  }
  if (splits == 1) {
    const float bv = bias ? Conv<T>::to_f32(bias[n]) : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = mrow + 8 * j + e;
        if (m < M) out[(int64_t)m * out_stride + n] = Conv<T>::from_f32(fin[j][e] + bv);
      }
    return;
  }
  // split-K: slab[tile][split] holds the block's 64 x 128 fp32 tile as [wave w][j][lane] float4.
  // Slabs are written and read with 16-byte agent-scope (write-through / cache-bypassing) accesses, so that no
  // whole-cache write-back or invalidate is needed around the ticket.
  const int tile = (CHAIN ? 0 : (int)(blockIdx.z * gridDim.x)) + ebx;
  float* const tslab = partial + (int64_t)tile * splits * Wide::kTileFloats;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(tslab, 0, splits * Wide::kTileFloats * 4, 0x00020000);
  constexpr int kSc1 = 16;  // cache policy bit of the buffer instructions: agent scope
  const int voff = ((wid * 4) * 64 + lane) * 16;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, fin[j]), rsrc, voff + j * 1024,
                                           eby * Wide::kTileFloats * 4, kSc1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every store of this wave has been acknowledged
  __builtin_amdgcn_s_barrier();
  if (tid == 0) {
    const int ticket = __hip_atomic_fetch_add(&tickets[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == splits - 1;
    if (last) __hip_atomic_store(&tickets[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero for the next launch
    *flag = last;
  }
  __syncthreads();
  if (*flag == 0) return;
  {
    float4v a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = float4v{0.f, 0.f, 0.f, 0.f};
    // two slabs' worth of loads in flight at a time (each one is a trip to memory); the sum runs in slab order
    for (int s0 = 0; s0 < splits; s0 += 2) {
      float4v v[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int s2 = min(s0 + q, splits - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          v[q][j] = __builtin_bit_cast(
              float4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + j * 1024, s2 * Wide::kTileFloats * 4, kSc1));
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float keep = (s0 + q < splits) ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] += keep * v[q][j];
      }
    }
    const float bv = bias ? Conv<T>::to_f32(bias[n]) : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = mrow + 8 * j + e;
        if (m < M) out[(int64_t)m * out_stride + n] = Conv<T>::from_f32(a[j][e] + bv);
      }
  }
  }();
  if constexpr (!CHAIN) break;
  // ---- hand-over: every store of this block acknowledged, then one count; the next job starts at the top
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (tid == 0)
    __hip_atomic_fetch_add(sync + (done_job * 8 + ((int)blockIdx.x & 7)) * kChainSyncStride, 1, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  if (job >= njobs) break;
  }  // while (true): jobs
  if constexpr (CHAIN) chain_leave(sync, njobs);
}

template <typename T, bool XS_EXT, int MT = 2>
__global__ void __launch_bounds__(512, 2) awq_gemm_wide_kernel(
    const uint16_t* __restrict__ x, const int32_t* __restrict__ packed, const uint16_t* __restrict__ scales,
    const int32_t* __restrict__ qzeros, const uint16_t* __restrict__ bias, uint16_t* __restrict__ out,
    float* __restrict__ partial, int* __restrict__ tickets, int M, int N, int K, int stages_per_split, int splits,
    int64_t x_stride, int64_t out_stride, const float* __restrict__ xsum_in, int64_t xs_stride, int partial_mode,
    int act_mode, float* __restrict__ xsum_out, int64_t xs_out_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const WideJob job{x,         packed,     scales,    qzeros,        bias, out, partial,          tickets, xsum_in,
                    xsum_out,  x_stride,   out_stride, xs_stride,    xs_out_stride, N,   K,       stages_per_split,
                    splits,    partial_mode, act_mode, 0,            0};
  wide_body<T, XS_EXT, MT, false>(&job, 1, M, nullptr, smem);
}

// The jobs of a chain, one persistent block per CU (see wide_body).  Block sums always come from the producers.
template <typename T, int MT>
__global__ void __launch_bounds__(512, 2) awq_chain_kernel(const ChainJobs cj, int M, int* __restrict__ sync) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  wide_body<T, true, MT, true>(cj.job, cj.njobs, M, sync, smem);
}

struct GemmPlan {
  int nw, tn, wr, splits, kb_per_split, mblocks, tiles;
  int64_t slab_floats;
};

static GemmPlan plan_awq_gemm(int M, int N, int K) {
  // SGLK_AWQ_CFG = 100 * waves + 10 * strips per wave + weight ring depth (tuning aid)
  static const int env_cfg = getenv("SGLK_AWQ_CFG") ? atoi(getenv("SGLK_AWQ_CFG")) : 0;
  static const int env_blocks = getenv("SGLK_AWQ_MIN_BLOCKS") ? atoi(getenv("SGLK_AWQ_MIN_BLOCKS")) : 224;
  static const int env_minkb = getenv("SGLK_AWQ_MIN_KB") ? atoi(getenv("SGLK_AWQ_MIN_KB")) : 4;
  GemmPlan p;
  p.mblocks = (M + kGemmBM - 1) / kGemmBM;
  const int kblocks = K / 128;
  int cfg = env_cfg;
  if (cfg && N % (16 * (cfg / 100) * ((cfg / 10) % 10))) cfg = 0;
  if (cfg == 0) cfg = (N % 128 == 0 && N >= 8192) ? 426 : 416;
  p.nw = cfg / 100;
  p.tn = (cfg / 10) % 10;
  p.wr = cfg % 10;
  const int nblocks = N / (16 * p.nw * p.tn);
  p.tiles = nblocks * p.mblocks;
  // enough blocks to cover the 256 CUs, but keep >= env_minkb k-blocks per split
  int splits = 1;
  while (p.tiles * splits < env_blocks && kblocks / (splits * 2) >= env_minkb) splits *= 2;
  p.kb_per_split = (kblocks + splits - 1) / splits;
  p.splits = (kblocks + p.kb_per_split - 1) / p.kb_per_split;
  p.slab_floats = (int64_t)p.nw * p.tn * 16 * 64;
  return p;
}

static constexpr int kMaxTiles = 4096;  // ticket words at the head of the workspace

// The wide kernel needs K in whole 512-k stages, N in whole 128-column tiles and 128-k scale groups.
WidePlan plan_awq_wide(int M, int N, int K, int group_size, bool partial_mode) {
  // 0: never, 1: where it is expected to be faster, 2: wherever the shape allows (read per call: a test hook)
  const char* env_wide_s = getenv("SGLK_AWQ_WIDE");
  const int env_wide = env_wide_s ? atoi(env_wide_s) : 1;
  const char* env_sps_s = getenv("SGLK_AWQ_WIDE_SPS");  // stages per split, read per call (test hook)
  const int env_sps = env_sps_s ? atoi(env_sps_s) : 0;
  WidePlan p{};
  p.ok = env_wide && group_size == 128 && K % Wide::kStageK == 0 && N % 128 == 0;
  if (!p.ok) return p;
  p.mblocks = (M + kGemmBM - 1) / kGemmBM;
  p.tiles = (N / 128) * p.mblocks;
  const int total = K / Wide::kStageK;
  // one block per CU: rounds of 256 blocks, each costing its stages plus about two stages of fill and drain,
  // plus the last split's reduction over all slabs
  int best_sps = total;
  float best = 1e30f;
  for (int sps = 1; sps <= total; ++sps) {
    const int splits = (total + sps - 1) / sps;
    const int rounds = (p.tiles * splits + 255) / 256;
    const float cost = rounds * (sps + 2.0f) + (splits > 1 && !partial_mode ? 0.25f * splits : 0.f);
    if (cost < best) {
      best = cost;
      best_sps = sps;
    }
  }
  if (env_sps > 0 && env_sps <= total) best_sps = env_sps;
  p.stages_per_split = best_sps;
  p.splits = (total + best_sps - 1) / best_sps;
  // a block that would run a single stage is all fill and drain: the narrow kernel is faster there
  if (best_sps < 2 && env_wide < 2 && !partial_mode) p.ok = false;
  return p;
}

}  // namespace sglk

using namespace sglk;

static constexpr int kMaxDevices = 64;

// Raises a kernel's dynamic-LDS limit once per (kernel, device): the attribute belongs to the device's copy of the
// code object.  One flag array per kernel, shared by the launch sites and by sglk_awq_gemm_static_init().
template <auto Kernel>
static int ensure_lds(int bytes) {
  static bool done[kMaxDevices];
  int dev = 0;
  SGLK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices || !done[dev]) {
    SGLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (dev >= 0 && dev < kMaxDevices) done[dev] = true;
  }
  return SGLK_OK;
}

// Everything the entry point would otherwise do on its first call on a device: the array's address and the LDS
// limits of every kernel the call can dispatch to.
template <typename T>
static int static_prepare_kernels() {
  if (int rc = ensure_lds<&awq_gemm_wide_kernel<T, false, 1>>(Wide::kLds)) return rc;
  if (int rc = ensure_lds<&awq_gemm_wide_kernel<T, false, 2>>(Wide::kLds)) return rc;
  if (int rc = ensure_lds<&awq_gemm_wide_kernel<T, true, 1>>(Wide::kLds)) return rc;
  if (int rc = ensure_lds<&awq_gemm_wide_kernel<T, true, 2>>(Wide::kLds)) return rc;
  if (int rc = ensure_lds<&awq_gemm_kernel<T, 4, 2, 6>>(GemmGeom<4, 2, 6>::kLds)) return rc;
  if (int rc = ensure_lds<&awq_gemm_kernel<T, 4, 1, 6>>(GemmGeom<4, 1, 6>::kLds)) return rc;
  if (int rc = ensure_lds<&awq_gemm_kernel<T, 4, 1, 4>>(GemmGeom<4, 1, 4>::kLds)) return rc;
  if (int rc = ensure_lds<&awq_gemm_kernel<T, 8, 1, 6>>(GemmGeom<8, 1, 6>::kLds)) return rc;
  if (int rc = ensure_lds<&awq_gemm_kernel<T, 8, 1, 4>>(GemmGeom<8, 1, 4>::kLds)) return rc;
  return SGLK_OK;
}


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

int sglk_awq_dequantize_packed(const int32_t* packed, const void* scales, const int32_t* qzeros, void* out, int K, int N,
                               int group_size, int dtype, void* stream) {
  SGLK_CHECK(packed && scales && qzeros && out, "awq_dequantize_packed: null pointer");
  SGLK_CHECK(K > 0 && N > 0 && K % 128 == 0 && N % 16 == 0, "awq_dequantize_packed: K %% 128 and N %% 16 must be 0");
  SGLK_CHECK(group_size > 0 && K % group_size == 0 && group_size % 8 == 0, "awq_dequantize_packed: bad group size");
  if (N % 64 == 0 && group_size % 128 == 0) {  // whole tiles, row-contiguous stores
    int64_t tiles = (int64_t)(K / 128) * (N / 64);
    int blocks = (int)((tiles + 3) / 4);
    if (blocks > 8192) blocks = 8192;
    SGLK_DISPATCH_HALF(dtype, T,
                       hipLaunchKernelGGL((awq_dequant_packed_tile_kernel<T>), dim3(blocks), dim3(256), 0,
                                          (hipStream_t)stream, packed, (const uint16_t*)scales, qzeros, (uint16_t*)out, K,
                                          N, group_size));
    SGLK_LAUNCH_CHECK();
    return SGLK_OK;
  }
  int blocks = (int)(((int64_t)K * (N / 8) + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  SGLK_DISPATCH_HALF(dtype, T,
                     hipLaunchKernelGGL((awq_dequant_packed_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                                        packed, (const uint16_t*)scales, qzeros, (uint16_t*)out, K, N, group_size));
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_gptq_repack(const int32_t* qweight, const int32_t* qzeros, int32_t* packed, int32_t* qzeros_out, int* overflow,
                     int K, int N, int group_size, int zero_offset, void* stream) {
  SGLK_CHECK(qweight && qzeros && packed && qzeros_out && overflow, "gptq_repack: null pointer");
  SGLK_CHECK(K > 0 && N > 0 && K % 128 == 0 && N % 16 == 0, "gptq_repack: K %% 128 and N %% 16 must be 0");
  SGLK_CHECK(group_size > 0 && K % group_size == 0 && group_size % 8 == 0, "gptq_repack: bad group size");
  SGLK_CHECK(zero_offset == 0 || zero_offset == 1, "gptq_repack: zero_offset is 1 (v1 checkpoints) or 0 (v2)");
  SGLK_CHECK((int64_t)K * (N / 8) >= (int64_t)(K / group_size) * (N / 8), "gptq_repack: bad shape");
  int blocks = (int)(((int64_t)K * (N / 8) + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(gptq_repack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, qweight, qzeros, packed,
                     qzeros_out, overflow, K, N, K / group_size, zero_offset);
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
  int64_t bytes = 0;
  if (p.splits > 1) bytes = (int64_t)p.tiles * p.splits * p.slab_floats * 4;
  const WidePlan wp = plan_awq_wide(M, N, K, 128);
  if (wp.ok && wp.splits > 1) bytes = std::max<int64_t>(bytes, (int64_t)wp.tiles * wp.splits * Wide::kTileFloats * 4);
  return bytes + (int64_t)kMaxTiles * 4;
}

static int launch_awq_wide(const WidePlan& wp, const void* x, const int32_t* packed, const void* scales,
                           const int32_t* qzeros, const void* bias, void* out, float* partial, int* tickets, int M, int N,
                           int K, int64_t x_stride, int64_t out_stride, int dtype, const float* xsum, int64_t xs_stride,
                           int partial_mode, hipStream_t s, int act_mode = 0, float* xsum_out = nullptr,
                           int64_t xs_out_stride = 0) {
  dim3 wgrid(N / 128, wp.splits, wp.mblocks);
#define SGLK_WIDE_LAUNCH(XS, MQ)                                                                                   \
  {                                                                                                                \
    if (int rc = ensure_lds<&awq_gemm_wide_kernel<T, XS, MQ>>(Wide::kLds)) return rc;                              \
    hipLaunchKernelGGL((awq_gemm_wide_kernel<T, XS, MQ>), wgrid, dim3(512), Wide::kLds, s, (const uint16_t*)x,     \
                       packed, (const uint16_t*)scales, qzeros, (const uint16_t*)bias, (uint16_t*)out, partial,    \
                       tickets, M, N, K, wp.stages_per_split, wp.splits, x_stride, out_stride, xsum, xs_stride,    \
                       partial_mode, act_mode, xsum_out, xs_out_stride);                                           \
  }
#define SGLK_WIDE_LAUNCH_MQ(XS)            \
  {                                        \
    if (M <= 32) SGLK_WIDE_LAUNCH(XS, 1)   \
    else SGLK_WIDE_LAUNCH(XS, 2)           \
  }
  SGLK_DISPATCH_HALF(dtype, T, {
    if (xsum) SGLK_WIDE_LAUNCH_MQ(true) else SGLK_WIDE_LAUNCH_MQ(false)
  });
#undef SGLK_WIDE_LAUNCH_MQ
#undef SGLK_WIDE_LAUNCH
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_awq_gemm_partial_splits(int M, int N, int K, int group_size) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const WidePlan wp = plan_awq_wide(M, N, K, group_size, true);
  return wp.ok ? wp.splits : 0;
}

int sglk_awq_gemm_partial(const void* x, const int32_t* packed, const void* scales, const int32_t* qzeros,
                          const float* xsum, int64_t xs_stride, float* partial, int M, int N, int K, int group_size,
                          int64_t x_stride, int dtype, void* stream) {
  SGLK_CHECK(x && packed && scales && qzeros && partial, "awq_gemm_partial: null pointer");
  SGLK_CHECK(M > 0 && N > 0 && K > 0, "awq_gemm_partial: bad shape");
  SGLK_CHECK(aligned16(x) && x_stride % 8 == 0, "awq_gemm_partial: x rows must be 16-byte aligned");
  SGLK_CHECK(!xsum || xs_stride >= M, "awq_gemm_partial: xs_stride smaller than M");
  const WidePlan wp = plan_awq_wide(M, N, K, group_size, true);
  if (!wp.ok)
    return fail(SGLK_EUNSUP, "awq_gemm_partial: needs K %% 512 == 0, N %% 128 == 0 and group size 128 (K=%d N=%d G=%d)", K,
                N, group_size);
  return launch_awq_wide(wp, x, packed, scales, qzeros, nullptr, nullptr, partial, nullptr, M, N, K, x_stride, N, dtype,
                         xsum, xs_stride, 1, (hipStream_t)stream);
}

int sglk_awq_gemm_silu_mul(const void* x, const int32_t* packed, const void* scales, const int32_t* qzeros, void* out,
                           int M, int N, int K, int group_size, int64_t x_stride, int64_t out_stride, int dtype,
                           const float* xsum, int64_t xs_stride, float* xsum_out, int64_t xs_out_stride, void* stream) {
  SGLK_CHECK(x && packed && scales && qzeros && out, "awq_gemm_silu_mul: null pointer");
  SGLK_CHECK(M > 0 && N > 0 && K > 0 && N % 256 == 0, "awq_gemm_silu_mul: N=%d must be a multiple of 256", N);
  SGLK_CHECK(aligned16(x) && x_stride % 8 == 0, "awq_gemm_silu_mul: x rows must be 16-byte aligned");
  SGLK_CHECK(out_stride >= N / 2, "awq_gemm_silu_mul: out_stride smaller than N/2");
  SGLK_CHECK(!xsum || xs_stride >= M, "awq_gemm_silu_mul: xs_stride smaller than M");
  SGLK_CHECK(!xsum_out || xs_out_stride >= M, "awq_gemm_silu_mul: xs_out_stride smaller than M");
  WidePlan wp = plan_awq_wide(M, N, K, group_size, true);
  if (!wp.ok)
    return fail(SGLK_EUNSUP, "awq_gemm_silu_mul: needs K %% 512 == 0 and group size 128 (K=%d G=%d)", K, group_size);
  // the activation needs whole sums: one split, whatever the planner would pick
  wp.stages_per_split = K / Wide::kStageK;
  wp.splits = 1;
  return launch_awq_wide(wp, x, packed, scales, qzeros, nullptr, out, nullptr, nullptr, M, N, K, x_stride, out_stride,
                         dtype, xsum, xs_stride, 0, (hipStream_t)stream, 1, xsum_out, xs_out_stride);
}

// ---- several projections in one launch of persistent blocks (see wide_body) ------------------------------------
int64_t sglk_awq_chain_sync_bytes(void) { return (int64_t)kChainSyncInts * 4; }

int sglk_awq_chain_status(const void* sync) {
  if (!sync) return fail(SGLK_EINVAL, "awq_chain_status: null pointer");
  int word = 0;
  SGLK_HIP(hipMemcpy(&word, reinterpret_cast<const int*>(sync) + kChainErrorWord, sizeof(int), hipMemcpyDeviceToHost));
  return word != 0 ? 1 : 0;
}

int sglk_awq_chain(const sglk_chain_job* jobs, int njobs, void* sync, int M, int group_size, int dtype, void* stream) {
  SGLK_CHECK(jobs && sync, "awq_chain: null pointer");
  SGLK_CHECK(njobs >= 1 && njobs <= kChainMaxJobs, "awq_chain: 1..%d jobs", kChainMaxJobs);
  SGLK_CHECK(M > 0 && M <= kGemmBM, "awq_chain: at most %d rows (got %d)", kGemmBM, M);
  SGLK_CHECK(dtype == SGLK_F16 || dtype == SGLK_BF16, "awq_chain: 16-bit activations only");
  static int cu_count = 0;
  if (!cu_count) {
    int dev = 0;
    SGLK_HIP(hipGetDevice(&dev));
    SGLK_HIP(hipDeviceGetAttribute(&cu_count, hipDeviceAttributeMultiprocessorCount, dev));
  }
  ChainJobs cj{};
  cj.njobs = njobs;
  int grid = 0;
  for (int i = 0; i < njobs; ++i) {
    const sglk_chain_job& a = jobs[i];
    SGLK_CHECK(a.x && a.packed && a.scales && a.qzeros && a.xsum_in, "awq_chain: job %d: null pointer", i);
    SGLK_CHECK(a.kind == SGLK_CHAIN_PARTIAL || a.kind == SGLK_CHAIN_SILU_MUL, "awq_chain: job %d: unknown kind %d", i, a.kind);
    SGLK_CHECK(a.N > 0 && a.K > 0, "awq_chain: job %d: bad shape", i);
    SGLK_CHECK(aligned16(a.x) && a.x_stride % 8 == 0, "awq_chain: job %d: x rows must be 16-byte aligned", i);
    SGLK_CHECK(a.xs_stride >= M, "awq_chain: job %d: xs_stride smaller than M", i);
    WidePlan wp = plan_awq_wide(M, a.N, a.K, group_size, true);
    if (!wp.ok)
      return fail(SGLK_EUNSUP, "awq_chain: job %d needs K %% 512 == 0, N %% 128 == 0 and group size 128 (K=%d N=%d G=%d)", i,
                  a.K, a.N, group_size);
    WideJob& j = cj.job[i];
    j = WideJob{};
    if (a.kind == SGLK_CHAIN_SILU_MUL) {
      SGLK_CHECK(a.out && a.N % 256 == 0 && a.out_stride >= a.N / 2, "awq_chain: job %d: gated activation needs out and N %% 256 == 0", i);
      SGLK_CHECK(!a.xsum_out || a.xs_out_stride >= M, "awq_chain: job %d: xs_out_stride smaller than M", i);
      wp.stages_per_split = a.K / Wide::kStageK;  // the activation needs whole sums: one split
      wp.splits = 1;
      j.act_mode = 1;
    } else {
      SGLK_CHECK(a.partial, "awq_chain: job %d: no room for the partial sums", i);
      SGLK_CHECK(a.splits_expected == 0 || a.splits_expected == wp.splits,
                 "awq_chain: job %d: the caller expects %d splits, the plan has %d", i, a.splits_expected, wp.splits);
      j.partial_mode = 1;
    }
    j.x = (const uint16_t*)a.x;
    j.packed = a.packed;
    j.scales = (const uint16_t*)a.scales;
    j.qzeros = a.qzeros;
    j.out = (uint16_t*)a.out;
    j.partial = a.partial;
    j.xsum_in = a.xsum_in;
    j.xsum_out = a.xsum_out;
    j.x_stride = a.x_stride;
    j.out_stride = a.kind == SGLK_CHAIN_SILU_MUL ? a.out_stride : a.N;
    j.xs_stride = a.xs_stride;
    j.xs_out_stride = a.xs_out_stride;
    j.N = a.N;
    j.K = a.K;
    j.stages_per_split = wp.stages_per_split;
    j.splits = wp.splits;
    j.grid_x = a.N / 128;
    j.nblocks = j.grid_x * wp.splits;
    grid = std::max(grid, j.nblocks);
  }
  // every block waits for blocks of the same launch: all of them must be resident at once (one per CU)
  if (grid > cu_count)
    return fail(SGLK_EUNSUP, "awq_chain: %d blocks do not fit the %d CUs of this device at once", grid, cu_count);
  hipStream_t s = (hipStream_t)stream;
#define SGLK_CHAIN_LAUNCH(MQ)                                                                                       \
  {                                                                                                                 \
    if (int rc = ensure_lds<&awq_chain_kernel<T, MQ>>(Wide::kLds)) return rc;                                       \
    hipLaunchKernelGGL((awq_chain_kernel<T, MQ>), dim3(grid), dim3(512), Wide::kLds, s, cj, M, (int*)sync);         \
  }
  SGLK_DISPATCH_HALF(dtype, T, {
    if (M <= 32) SGLK_CHAIN_LAUNCH(1) else SGLK_CHAIN_LAUNCH(2)
  });
#undef SGLK_CHAIN_LAUNCH
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_awq_gemm_xs(const void* x, const int32_t* packed, const void* scales, const int32_t* qzeros, const void* bias,
                     void* out, float* workspace, int M, int N, int K, int group_size, int64_t x_stride,
                     int64_t out_stride, int dtype, const float* xsum, int64_t xs_stride, void* stream) {
  SGLK_CHECK(x && packed && scales && qzeros && out && workspace, "awq_gemm: null pointer");
  SGLK_CHECK(M > 0 && N > 0 && K > 0, "awq_gemm: bad shape");
  SGLK_CHECK(K % 128 == 0, "awq_gemm: K=%d must be a multiple of 128", K);
  SGLK_CHECK(N % 64 == 0, "awq_gemm: N=%d must be a multiple of 64", N);
  SGLK_CHECK(group_size % 128 == 0 && K % group_size == 0, "awq_gemm: group_size=%d must be a multiple of 128 dividing K",
             group_size);
  SGLK_CHECK(aligned16(x) && x_stride % 8 == 0, "awq_gemm: x rows must be 16-byte aligned");
  SGLK_CHECK(out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0, "awq_gemm: out rows must be 8-byte aligned");
  SGLK_CHECK(aligned16(workspace), "awq_gemm: workspace must be 16-byte aligned");
  SGLK_CHECK(!xsum || xs_stride >= M, "awq_gemm: xs_stride smaller than M");
  hipStream_t s = (hipStream_t)stream;
  int* tickets = reinterpret_cast<int*>(workspace);
  float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + (int64_t)kMaxTiles * 4);
  const WidePlan wp = plan_awq_wide(M, N, K, group_size);
  if (wp.ok && wp.tiles <= kMaxTiles && aligned16(out) && out_stride % 8 == 0)
    return launch_awq_wide(wp, x, packed, scales, qzeros, bias, out, partial, tickets, M, N, K, x_stride, out_stride,
                           dtype, xsum, xs_stride, 0, s);
  const GemmPlan p = plan_awq_gemm(M, N, K);
  SGLK_CHECK(p.tiles <= kMaxTiles, "awq_gemm: too many output tiles (%d)", p.tiles);
  dim3 grid(N / (16 * p.nw * p.tn), p.splits, p.mblocks);
#define SGLK_AWQ_LAUNCH(NW, TN, WR)                                                                               \
  {                                                                                                               \
    using GG = GemmGeom<NW, TN, WR>;                                                                              \
    if (int rc = ensure_lds<&awq_gemm_kernel<T, NW, TN, WR>>(GG::kLds)) return rc;                                \
    hipLaunchKernelGGL((awq_gemm_kernel<T, NW, TN, WR>), grid, dim3(64 * NW), GG::kLds, s, (const uint16_t*)x,    \
                       packed, (const uint16_t*)scales, qzeros, (const uint16_t*)bias, (uint16_t*)out, partial,   \
                       tickets, M, N, K, group_size, p.kb_per_split, p.splits, x_stride, out_stride);             \
  }
  const int code = p.nw * 100 + p.tn * 10 + p.wr;
  SGLK_DISPATCH_HALF(dtype, T, {
    switch (code) {
      case 426: SGLK_AWQ_LAUNCH(4, 2, 6) break;
      case 416: SGLK_AWQ_LAUNCH(4, 1, 6) break;
      case 414: SGLK_AWQ_LAUNCH(4, 1, 4) break;
      case 816: SGLK_AWQ_LAUNCH(8, 1, 6) break;
      case 814: SGLK_AWQ_LAUNCH(8, 1, 4) break;
      default: return fail(SGLK_EINVAL, "awq_gemm: no kernel for configuration %d", code);
    }
  });
#undef SGLK_AWQ_LAUNCH
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_awq_gemm(const void* x, const int32_t* packed, const void* scales, const int32_t* qzeros, const void* bias,
                  void* out, float* workspace, int M, int N, int K, int group_size, int64_t x_stride,
                  int64_t out_stride, int dtype, void* stream) {
  return sglk_awq_gemm_xs(x, packed, scales, qzeros, bias, out, workspace, M, N, K, group_size, x_stride, out_stride,
                          dtype, nullptr, 0, stream);
}

// ---- sglk_awq_gemm for callers that cannot allocate: the split-K workspace (ticket words + fp32 slabs) lives in a
// static device array of the code object.  Every GPU the library runs on holds its own copy at its own address; the
// address is looked up once per device (the array's name is only a host-side handle, never a device pointer).
static constexpr int64_t kStaticWsBytes = (int64_t)kMaxTiles * 4 + ((int64_t)7 << 20);
static __device__ __align__(256) unsigned char scratch[kStaticWsBytes];
static void* g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_mu;

// The array's address on the CURRENT device.  `operand` (may be null) is a device pointer of the call: its owner
// must be the current device, because the address handed to the kernel is that device's copy of the array.
// `stream` (with `launching`) is the stream the call is about to launch on: the first resolution on a device takes a
// lock and asks the runtime for the symbol, which must not happen while that stream is being captured.
static int static_scratch(void** ptr, const void* operand = nullptr, hipStream_t stream = nullptr,
                          bool launching = false) {
  int dev = -1;
  SGLK_HIP(hipGetDevice(&dev));
  SGLK_CHECK(dev >= 0 && dev < kMaxDevices, "awq_gemm_static: device ordinal %d out of range", dev);
  if (operand) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, operand) == hipSuccess && attr.type == hipMemoryTypeDevice)
      SGLK_CHECK(attr.device == dev,
                 "awq_gemm_static: the operands live on device %d but device %d is current (hipSetDevice to the "
                 "operands' device before the call: each GPU has its own static workspace)",
                 attr.device, dev);
    else
      (void)hipGetLastError();  // not a pointer the runtime knows: leave the judgement to the launch
  }
  void* known = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (!known) {
    if (launching) {
      hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
      SGLK_HIP(hipStreamIsCapturing(stream, &st));
      SGLK_CHECK(st == hipStreamCaptureStatusNone,
                 "awq_gemm_static: first use on device %d while the stream is being captured: call "
                 "sglk_awq_gemm_static_init() on this device before the capture begins",
                 dev);
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
      void* p = nullptr;
      SGLK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
      SGLK_CHECK(p != nullptr, "awq_gemm_static: no device address for the static workspace");
      __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);
    }
    known = g_scratch_ptr[dev];
  }
  *ptr = known;
  return SGLK_OK;
}

#if SGLK_ABL & 1024
__attribute__((visibility("default"))) int sglk_debug_read_timeline(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_timeline), sizeof(unsigned int) * 256 * 8 * 16);
}
#endif
#if SGLK_ABL & 64
__attribute__((visibility("default"))) int sglk_debug_read_stamps(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), sizeof(long long) * 256 * 24);
}
__attribute__((visibility("default"))) int sglk_debug_clear_stamps(void) {
  static long long zeros[256 * 24];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof(zeros));
}
#endif

int64_t sglk_awq_gemm_static_max_workspace_bytes(void) { return kStaticWsBytes; }

int sglk_awq_gemm_static_init(void) {
  void* p = nullptr;
  if (int rc = static_scratch(&p)) return rc;
  if (int rc = static_prepare_kernels<bf16_t>()) return rc;
  return static_prepare_kernels<f16_t>();
}

int sglk_awq_gemm_static(const void* x, const int32_t* packed, const void* scales, const int32_t* qzeros,
                         const void* bias, void* out, int M, int N, int K, int group_size, int64_t x_stride,
                         int64_t out_stride, int dtype, void* stream) {
  SGLK_CHECK(M > 0 && N > 0 && K > 0 && K % 128 == 0, "awq_gemm_static: bad shape");
  const int64_t need = sglk_awq_gemm_workspace_bytes(M, N, K);
  if (need > kStaticWsBytes)
    return fail(SGLK_EINVAL,
                "awq_gemm_static: M=%d N=%d K=%d needs a %lld-byte workspace, the static one holds %lld: use "
                "sglk_awq_gemm with a caller-provided workspace",
                M, N, K, (long long)need, (long long)kStaticWsBytes);
  void* ws = nullptr;
  const int rc = static_scratch(&ws, out, (hipStream_t)stream, true);
  if (rc != SGLK_OK) return rc;
  // the array keeps the previous call's contents: the ticket words must start at zero
  SGLK_HIP(hipMemsetAsync(ws, 0, (size_t)kMaxTiles * 4, (hipStream_t)stream));
  return sglk_awq_gemm_xs(x, packed, scales, qzeros, bias, out, reinterpret_cast<float*>(ws), M, N, K, group_size,
                          x_stride, out_stride, dtype, nullptr, 0, stream);
}
}
