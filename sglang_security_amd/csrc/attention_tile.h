// Matrix-core attention building blocks for head size 128 and 16-bit types (gfx950).
//
// Tiles of 32 KV tokens live in LDS as 32 rows x 256 bytes, XOR-swizzled so that both the row reads
// of the K operand (ds_read_b128) and the transposed reads of the V operand (ds_read_b64_tr_b16)
// spread over all banks.  Scores are produced transposed, S^T = K . Q^T, so that one 16-column
// operand tile holds 16 independent query rows ("columns") and the score accumulators are, as they
// stand, the P^T operand of O^T += V^T . P^T.
#pragma once
#include "common.h"

namespace sglk {

template <typename T>
struct Mfma16;
template <>
struct Mfma16<bf16_t> {
  static __device__ __forceinline__ float4v mma(short8 a, short8 b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfloat8, a), __builtin_bit_cast(bfloat8, b), c, 0,
                                                   0, 0);
  }
};
template <>
struct Mfma16<f16_t> {
  static __device__ __forceinline__ float4v mma(short8 a, short8 b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0,
                                                  0);
  }
};

typedef __attribute__((address_space(3))) short4v* lds_short4_ptr;

__device__ __forceinline__ short4v lds_read_tr16(const char* p) {
  // ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of 16-bit elements, delivered
  // column-major (lane i of the group receives column i of the four rows).  EXEC must be full.
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_ptr)(p));
}

static constexpr int kTile = 32;                      // tokens per tile
static constexpr int kRowBytes = 256;                 // 128 x 16-bit
static constexpr int kTileBytes = kTile * kRowBytes;  // 8 KiB each for K and V

// K tile: 16-byte chunk c of row r sits at chunk slot c ^ (r & 15): the row reads of one MFMA
// operand (16 rows, same chunk) then fall on 16 different slots of the 256-byte bank row.
__device__ __forceinline__ int k_off(int r, int c) { return r * kRowBytes + ((c ^ (r & 15)) << 4); }
// V tile: 32-byte granule g of row r sits at granule slot g ^ (r & 7): the transposed reads of one
// 32-lane half (8 rows, same granule) fall on 8 different 32-byte slots.
__device__ __forceinline__ int v_off(int r, int byte_in_row) {
  const int g = byte_in_row >> 5;
  return r * kRowBytes + (((g ^ (r & 7)) << 5) | (byte_in_row & 31));
}

// ---------------------------------------------------------------------------------------------------
// The qkv projection of a fused decode layer as its GEMM leaves it: split-K partial sums in fp32, not yet added up,
// biased, rounded or rotated.  The value the attention kernels work on is
//   round_T(rope(round_T(sum_s partial[s][token][col] + bias[col])))   (rotation of full 128-element heads, pairs
// (d, d + 64), cos_sin row = cos[64] | sin[64] at the token's position): the bits the separate
// "sum, round, rotate, store" kernel produces.
// ---------------------------------------------------------------------------------------------------
struct QkvPartials {
  const float* partials;     // [splits][tokens][row_len]
  int splits;
  int64_t part_stride;       // floats between splits
  int row_len;               // (q heads + 2 kv heads) * 128
  const uint16_t* bias;      // row_len values of the 16-bit type, or nullptr
  const float* cos_sin;      // [positions][128]
  const int64_t* positions;  // [tokens]
  const void* loc;           // pool slot of every token's new K / V row
  int loc_is64;
};

// sum over splits (in split order, four splits' loads in flight together) of the 8 floats at p + s * stride
__device__ __forceinline__ void sum_splits8(const float* p, int splits, int64_t stride, float (&acc)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int s0 = 0; s0 < splits; s0 += 4) {
    float4v a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* pr = p + (int64_t)min(s0 + u, splits - 1) * stride;
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
}

// The rotated Q^T operand of one column (a 128-element head starting at element col_off of row `token`) in the
// register layout of load_q: lane (hl, g) keeps elements 32 ks + 8 g .. + 8, and the rotation partner of element d,
// d + 64, is element (ks + 2, same g) of the same lane.
template <typename T>
__device__ __forceinline__ void load_q_rotated(short8 (&qf)[4], const QkvPartials& qp, int token, int64_t pos, int col_off,
                                               bool ok) {
  const int g = (threadIdx.x & 63) >> 4;
  if (!ok) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = short8{0, 0, 0, 0, 0, 0, 0, 0};
    return;
  }
  const float* cs = qp.cos_sin + pos * 128;
  const float* base = qp.partials + (int64_t)token * qp.row_len + col_off + 8 * g;
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    float a[8], b[8];
    sum_splits8(base + 32 * pr, qp.splits, qp.part_stride, a);
    sum_splits8(base + 32 * pr + 64, qp.splits, qp.part_stride, b);
    const float4v c0 = *reinterpret_cast<const float4v*>(cs + 32 * pr + 8 * g);
    const float4v c1 = *reinterpret_cast<const float4v*>(cs + 32 * pr + 8 * g + 4);
    const float4v s0 = *reinterpret_cast<const float4v*>(cs + 64 + 32 * pr + 8 * g);
    const float4v s1 = *reinterpret_cast<const float4v*>(cs + 64 + 32 * pr + 8 * g + 4);
    short8 ba = short8{0, 0, 0, 0, 0, 0, 0, 0}, bb = ba;
    if (qp.bias) {
      ba = *reinterpret_cast<const short8*>(qp.bias + col_off + 32 * pr + 8 * g);
      bb = *reinterpret_cast<const short8*>(qp.bias + col_off + 32 * pr + 8 * g + 64);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float bja = qp.bias ? Conv<T>::to_f32((uint16_t)ba[e]) : 0.f;
      const float bjb = qp.bias ? Conv<T>::to_f32((uint16_t)bb[e]) : 0.f;
      const float xa = Conv<T>::to_f32(Conv<T>::from_f32(a[e] + bja));
      const float xb = Conv<T>::to_f32(Conv<T>::from_f32(b[e] + bjb));
      float ra, rb;
      rope_pair(xa, xb, e < 4 ? c0[e & 3] : c1[e & 3], e < 4 ? s0[e & 3] : s1[e & 3], ra, rb);
      qf[pr][e] = (short)Conv<T>::from_f32(ra);
      qf[pr + 2][e] = (short)Conv<T>::from_f32(rb);
    }
  }
}

// One wave finishes a token's new K row (rotated) and V row (as it is) of one kv head and writes them to the pool:
// lane l owns elements l and l + 64.  The stores are left in flight: a reader waits for this wave's "s_waitcnt
// vmcnt(0)" (and a barrier, if it is another wave) before it loads the rows.
template <typename T>
__device__ __forceinline__ void store_new_kv(const QkvPartials& qp, int token, int64_t pos, int k_col, int v_col,
                                             uint16_t* krow, uint16_t* vrow) {
  const int l = threadIdx.x & 63;
  const float* cs = qp.cos_sin + pos * 128;
  const float* pk = qp.partials + (int64_t)token * qp.row_len + k_col + l;
  const float* pv = qp.partials + (int64_t)token * qp.row_len + v_col + l;
  float ka = 0.f, kb = 0.f, va = 0.f, vb = 0.f;
  for (int s0 = 0; s0 < qp.splits; s0 += 4) {
    float t[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t o = (int64_t)min(s0 + u, qp.splits - 1) * qp.part_stride;
      t[u][0] = pk[o];
      t[u][1] = pk[o + 64];
      t[u][2] = pv[o];
      t[u][3] = pv[o + 64];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (s0 + u < qp.splits) {
        ka += t[u][0];
        kb += t[u][1];
        va += t[u][2];
        vb += t[u][3];
      }
  }
  auto bias_at = [&](int col) { return qp.bias ? Conv<T>::to_f32(qp.bias[col]) : 0.f; };
  const float xa = Conv<T>::to_f32(Conv<T>::from_f32(ka + bias_at(k_col + l)));
  const float xb = Conv<T>::to_f32(Conv<T>::from_f32(kb + bias_at(k_col + l + 64)));
  float ra, rb;
  rope_pair(xa, xb, cs[l], cs[64 + l], ra, rb);
  krow[l] = Conv<T>::from_f32(ra);
  krow[l + 64] = Conv<T>::from_f32(rb);
  vrow[l] = Conv<T>::from_f32(va + bias_at(v_col + l));
  vrow[l + 64] = Conv<T>::from_f32(vb + bias_at(v_col + l + 64));
}

// ---------------------------------------------------------------------------------------------------
// Block-cooperative streaming attention: the 4 waves of a 256-thread block share each K/V tile
// (double-buffered in LDS, one barrier per tile) and every wave multiplies it against its own NC
// column tiles of 16 query rows.  State is carried across run() calls, so a caller can stream a
// gathered range (pool slots through an index list) and then a contiguous range under a causal
// mask into the same accumulators.
// ---------------------------------------------------------------------------------------------------
static constexpr int kBlockAttnLds = 2 * 2 * kTileBytes;  // two stages of (K tile + V tile)

// fp8 pools: eight one-byte codes -> eight values of the 16-bit type through a 256-entry table in LDS
// (lut[code] = the code's exact value: every fp8 value, denormal, infinity and NaN has one in fp16 and in bf16)
static constexpr int kKv8LutBytes = 512;
__device__ __forceinline__ int4v kv8_widen(const uint16_t* lut, int lo, int hi) {
  int4v out;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const uint32_t word = (uint32_t)(p < 2 ? lo : hi) >> (16 * (p & 1));
    out[p] = (int)((uint32_t)lut[word & 0xFFu] | ((uint32_t)lut[(word >> 8) & 0xFFu] << 16));
  }
  return out;
}

template <typename T, int NC>
struct BlockAttn {
  float m[NC];      // running maximum per column (log2 domain), replicated over the four lane groups
  float lsum[NC];   // this lane's share of the running sum
  float4v oacc[NC][8];
  short8 qf[NC][4];

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      m[c] = -INFINITY;
      lsum[c] = 0.f;
#pragma unroll
      for (int n = 0; n < 8; ++n) oacc[c][n] = float4v{0.f, 0.f, 0.f, 0.f};
    }
  }

  // Q^T operand of column tile c: this lane's column is the 128-element row at qrow (nullptr: zeros);
  // lane (hl, g) keeps elements 32*ks + 8*g .. +8 for ks = 0..3.
  __device__ __forceinline__ void load_q(int c, const uint16_t* qrow) {
    const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (qrow)
        qf[c][ks] = *reinterpret_cast<const short8*>(qrow + 32 * ks + 8 * g);
      else
        qf[c][ks] = short8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  // a 64-element query row in half `half` of the 128-wide operand, zeros in the other half (pairs of 64-wide kv heads
  // walked as one 128-wide row: the product then is the row's own head's)
  __device__ __forceinline__ void load_q_half(int c, const uint16_t* qrow, int half) {
    const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (qrow && (ks >> 1) == half)
        qf[c][ks] = *reinterpret_cast<const short8*>(qrow + 32 * (ks & 1) + 8 * g);
      else
        qf[c][ks] = short8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  // Stream tokens [t_begin, t_end) (block-uniform, non-empty).  GATHER: token t lives in pool row idx[t],
  // else in row t of kbase/vbase.  CAUSAL: column tile c of this lane sees token t only if t <= qpos[c];
  // wave_qmax is the largest qpos of the wave (tiles wholly above it are skipped by the wave).
  // PF: tiles in flight ahead of the one being multiplied (registers: 4 x 16-byte per tile and thread).
  struct NoHook {
    __device__ __forceinline__ void operator()() const {}
  };
  // after_fetch runs once the first tiles' loads are on their way (e.g. to assemble the Q operand in their shadow)
  // KV8: the rows are 128 one-byte codes of an fp8 pool (kbase / vbase byte pointers, strides in bytes): they travel
  // as 8-byte pieces and are widened through `lut` when parked.
  // pre_rows: the pool rows of the first PF tiles, already in registers (pre_rows[2 * p + i] = the row of token
  // t_begin + p * kTile + frow + 16 * i, clamped as fetch_idx clamps): a caller that could fetch them together with
  // its own metadata saves the tile loop its first trip to memory.
  template <bool CAUSAL, bool GATHER, int PF = 1, typename Hook = NoHook, bool KV8 = false>
  __device__ __forceinline__ void run(char* lds, const uint16_t* kbase, const uint16_t* vbase, int64_t k_stride_n,
                                      int64_t v_stride_n, const int32_t* idx, int t_begin, int t_end,
                                      const int (&qpos)[NC], int wave_qmax, float sm_scale, float logit_cap,
                                      Hook after_fetch = Hook(), const uint16_t* lut = nullptr,
                                      const int* pre_rows = nullptr) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int hl = lane & 15, g = lane >> 4;
    const int frow = tid >> 4, chunk = tid & 15;  // this thread parks rows frow and frow + 16
    const float LOG2E = 1.4426950408889634f;
    const float qk_scale = (logit_cap > 0.f) ? sm_scale : sm_scale * LOG2E;

    // Pool rows are found through the index list, so a tile costs two trips to memory: its indices, then its rows.
    // The counters that order returning loads are in issue order, so an index load issued after row loads waits for
    // them; the indices are therefore fetched one round (PF tiles) ahead of the rows that need them.
    int4v kreg[PF][2], vreg[PF][2];
    int64_t rowi[PF][2];
    auto fetch_idx = [&](int64_t(&ri)[2], int t0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int tok = t0 + frow + 16 * i;
        tok = tok < t_end ? tok : t_begin;  // rows past the end re-read a valid row; their scores are masked
        ri[i] = GATHER ? (int64_t)idx[tok] : (int64_t)tok;
      }
    };
    auto fetch = [&](int4v(&kr)[2], int4v(&vr)[2], const int64_t(&ri)[2]) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr (KV8) {
          const int2v a = *reinterpret_cast<const int2v*>(reinterpret_cast<const uint8_t*>(kbase) + ri[i] * k_stride_n + chunk * 8);
          const int2v b = *reinterpret_cast<const int2v*>(reinterpret_cast<const uint8_t*>(vbase) + ri[i] * v_stride_n + chunk * 8);
          kr[i] = int4v{a[0], a[1], 0, 0};
          vr[i] = int4v{b[0], b[1], 0, 0};
        } else {
          kr[i] = *reinterpret_cast<const int4v*>(kbase + ri[i] * k_stride_n + chunk * 8);
          vr[i] = *reinterpret_cast<const int4v*>(vbase + ri[i] * v_stride_n + chunk * 8);
        }
      }
    };
    // One K / V fragment read from LDS feeds all NC column tiles of the wave.
    auto compute = [&](int t0, const char* kl, const char* vl) {
      float4v sacc[NC][2];
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) sacc[c][hh] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const short8 kf = *reinterpret_cast<const short8*>(kl + k_off(16 * hh + hl, 4 * ks + g));
#pragma unroll
          for (int c = 0; c < NC; ++c) sacc[c][hh] = Mfma16<T>::mma(kf, qf[c][ks], sacc[c][hh]);
        }
      short8 pf[NC];  // P^T operands: k-slot j <-> token (j < 4 ? 4g + j : 16 + 4g + j - 4), as the S^T registers stand
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float sv[8];
        float tmax = -INFINITY;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float s = sacc[c][hh][r] * qk_scale;
            if (logit_cap > 0.f) s = logit_cap * tanhf(s / logit_cap) * LOG2E;
            const int tok = t0 + 16 * hh + 4 * g + r;
            bool ok = tok < t_end;
            if (CAUSAL) ok = ok && tok <= qpos[c];
            s = ok ? s : -INFINITY;
            sv[hh * 4 + r] = s;
            tmax = fmaxf(tmax, s);
          }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m[c], tmax);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // nothing visible yet: keep everything at zero
        const float alpha = exp2f(m[c] - m_use);
        m[c] = m_new;
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = exp2f(sv[j] - m_use);
          psum += p;
          pf[c][j] = (short)Conv<T>::from_f32(p);
        }
        lsum[c] = lsum[c] * alpha + psum;
#pragma unroll
        for (int n = 0; n < 8; ++n) oacc[c][n] *= alpha;
      }
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const int byte_in_row = 32 * n + 8 * (hl & 3);
        const short4v lo = lds_read_tr16(vl + v_off(4 * g + (hl >> 2), byte_in_row));
        const short4v hi = lds_read_tr16(vl + v_off(16 + 4 * g + (hl >> 2), byte_in_row));
        const short8 vf = short8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int c = 0; c < NC; ++c) oacc[c][n] = Mfma16<T>::mma(vf, pf[c], oacc[c][n]);
      }
    };

    __syncthreads();  // a previous run() may still be reading the stages
#pragma unroll
    for (int p = 0; p < PF; ++p)
      if (t_begin + p * kTile < t_end) {
        if (pre_rows) {
          rowi[p][0] = pre_rows[2 * p];
          rowi[p][1] = pre_rows[2 * p + 1];
        } else {
          fetch_idx(rowi[p], t_begin + p * kTile);
        }
      }
#pragma unroll
    for (int p = 0; p < PF; ++p)
      if (t_begin + p * kTile < t_end) fetch(kreg[p], vreg[p], rowi[p]);
#pragma unroll
    for (int p = 0; p < PF; ++p)
      if (t_begin + (PF + p) * kTile < t_end) fetch_idx(rowi[p], t_begin + (PF + p) * kTile);
    after_fetch();
    int stage = 0;
    for (int tb = t_begin; tb < t_end; tb += kTile * PF) {
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        const int t0 = tb + p * kTile;
        if (t0 < t_end) {  // block-uniform
          char* kl = lds + stage * 2 * kTileBytes;
          char* vl = kl + kTileBytes;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int r = frow + 16 * i;
            if constexpr (KV8) {
              *reinterpret_cast<int4v*>(kl + k_off(r, chunk)) = kv8_widen(lut, kreg[p][i][0], kreg[p][i][1]);
              *reinterpret_cast<int4v*>(vl + v_off(r, chunk * 16)) = kv8_widen(lut, vreg[p][i][0], vreg[p][i][1]);
            } else {
              *reinterpret_cast<int4v*>(kl + k_off(r, chunk)) = kreg[p][i];
              *reinterpret_cast<int4v*>(vl + v_off(r, chunk * 16)) = vreg[p][i];
            }
          }
          __syncthreads();
          if (t0 + PF * kTile < t_end) {  // flies while tiles are multiplied
            fetch(kreg[p], vreg[p], rowi[p]);
            if (t0 + 2 * PF * kTile < t_end) fetch_idx(rowi[p], t0 + 2 * PF * kTile);
          }
          if (!(CAUSAL && t0 > wave_qmax)) compute(t0, kl, vl);  // wave-uniform skip
          stage ^= 1;
        }
      }
    }
  }

  // total of the running sum over the four lane groups
  __device__ __forceinline__ float total(int c) const {
    float s = lsum[c];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    return s;
  }
};

// ---------------------------------------------------------------------------------------------------
// Wave-private streaming attention: one wave walks a gathered token range through its own K/V tile
// pair in LDS (no block-level synchronisation) and multiplies it against one column tile of 16
// query rows.  Rows of the next tile fly in registers while the current one is multiplied.
// ---------------------------------------------------------------------------------------------------
static constexpr int kWaveAttnLds = 2 * kTileBytes;  // K tile + V tile per wave

template <typename T>
struct WaveAttn {
  float m;     // running maximum (log2 domain), replicated over the four lane groups
  float lsum;  // this lane's share of the running sum
  float4v oacc[8];
  short8 qf[4];

  __device__ __forceinline__ void init() {
    m = -INFINITY;
    lsum = 0.f;
#pragma unroll
    for (int n = 0; n < 8; ++n) oacc[n] = float4v{0.f, 0.f, 0.f, 0.f};
  }

  // Q^T operand: this lane's column is the 128-element row at qrow (nullptr: zeros); lane (hl, g)
  // keeps elements 32*ks + 8*g .. +8 for ks = 0..3.
  __device__ __forceinline__ void load_q(const uint16_t* qrow) {
    const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (qrow)
        qf[ks] = *reinterpret_cast<const short8*>(qrow + 32 * ks + 8 * g);
      else
        qf[ks] = short8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  // a 64-element query row in half `half` of the 128-wide operand, zeros in the other half
  __device__ __forceinline__ void load_q_half(const uint16_t* qrow, int half) {
    const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (qrow && (ks >> 1) == half)
        qf[ks] = *reinterpret_cast<const short8*>(qrow + 32 * (ks & 1) + 8 * g);
      else
        qf[ks] = short8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  // Rows of one tile on their way from the pool to LDS: instruction i covers rows 4i + g.
  struct Rows {
    int4v k[8], v[8];
  };
  Rows rows;  // the tile run() keeps in flight

  // Pool slots of tokens base .. base + 63 of a list [s0, s1): lane l holds the slot of token base + l (tokens past the
  // end re-read a valid slot; their scores are masked).  One load serves two tiles, and it is issued well before the
  // row loads that need it: the counters that order returning loads are in issue order.
  static __device__ __forceinline__ int load_slots(const int32_t* idx, int base, int s0, int s1) {
    int tok = base + (int)(threadIdx.x & 63);
    tok = tok < s1 ? tok : s0;
    return idx[tok];
  }

  // rows of the tile that starts `off` tokens into the 64 tokens whose slots are held in `slots`
  static __device__ __forceinline__ void fetch(Rows& r, const uint16_t* kbase, const uint16_t* vbase, int64_t k_stride_n,
                                               int64_t v_stride_n, int slots, int off) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, chunk = lane & 15;  // chunk: 16-byte piece of the row this lane fetches
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t slot = __shfl(slots, off + 4 * i + g, 64);
      r.k[i] = *reinterpret_cast<const int4v*>(kbase + slot * k_stride_n + chunk * 8);
      r.v[i] = *reinterpret_cast<const int4v*>(vbase + slot * v_stride_n + chunk * 8);
    }
  }

  // fetched rows -> this wave's K / V tile in LDS (swizzled)
  static __device__ __forceinline__ void park(const Rows& r, char* kl, char* vl) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, chunk = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 4 * i + g;
      *reinterpret_cast<int4v*>(kl + k_off(row, chunk)) = r.k[i];
      *reinterpret_cast<int4v*>(vl + v_off(row, chunk * 16)) = r.v[i];
    }
  }

  // One parked tile (tokens t0 .. t0 + 31 of a list that ends at s1) into the running state.  col_ok: this lane's
  // query column takes part in the tile (a column that does not keeps its state).
  __device__ __forceinline__ void step(const char* kl, const char* vl, int t0, int s1, bool col_ok, float sm_scale,
                                       float logit_cap) {
    const int lane = threadIdx.x & 63;
    const int hl = lane & 15, g = lane >> 4;
    const float LOG2E = 1.4426950408889634f;
    const float qk_scale = (logit_cap > 0.f) ? sm_scale : sm_scale * LOG2E;
    // ---- S^T = K . Q^T : two 16-token halves, lane (hl, g) gets tokens 16*hh + 4*g + r of column hl
    float4v sacc[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      sacc[hh] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const short8 kf = *reinterpret_cast<const short8*>(kl + k_off(16 * hh + hl, 4 * ks + g));
        sacc[hh] = Mfma16<T>::mma(kf, qf[ks], sacc[hh]);
      }
    }
    // ---- scale / cap / mask, online softmax (exp2 domain)
    float sv[8];
    float tmax = -INFINITY;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = sacc[hh][r] * qk_scale;
        if (logit_cap > 0.f) s = logit_cap * tanhf(s / logit_cap) * LOG2E;
        const int tok = t0 + 16 * hh + 4 * g + r;
        s = (col_ok && tok < s1) ? s : -INFINITY;
        sv[hh * 4 + r] = s;
        tmax = fmaxf(tmax, s);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // nothing seen yet: everything stays at zero
    const float alpha = exp2f(m - m_use);
    m = m_new;
    float psum = 0.f;
    short8 pf;  // P^T operand: k-slot j <-> token (j < 4 ? 4g + j : 16 + 4g + j - 4), as the S^T registers stand
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float p = exp2f(sv[j] - m_use);
      psum += p;
      pf[j] = (short)Conv<T>::from_f32(p);
    }
    lsum = lsum * alpha + psum;
#pragma unroll
    for (int n = 0; n < 8; ++n) oacc[n] *= alpha;
    // ---- O^T += V^T . P^T : eight 16-column blocks of V, read transposed
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int byte_in_row = 32 * n + 8 * (hl & 3);
      const short4v lo = lds_read_tr16(vl + v_off(4 * g + (hl >> 2), byte_in_row));
      const short4v hi = lds_read_tr16(vl + v_off(16 + 4 * g + (hl >> 2), byte_in_row));
      const short8 vf = short8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      oacc[n] = Mfma16<T>::mma(vf, pf, oacc[n]);
    }
  }

  // Stream tokens [s0, s1) of the slot list idx (wave-uniform range, may be empty); one tile in flight.
  __device__ __forceinline__ void run(char* kl, char* vl, const uint16_t* kbase, const uint16_t* vbase,
                                      int64_t k_stride_n, int64_t v_stride_n, const int32_t* idx, int s0, int s1,
                                      float sm_scale, float logit_cap) {
    if (s1 <= s0) return;
    int slots = load_slots(idx, s0, s0, s1);
    int slots_next = s0 + 2 * kTile < s1 ? load_slots(idx, s0 + 2 * kTile, s0, s1) : 0;
    fetch(rows, kbase, vbase, k_stride_n, v_stride_n, slots, 0);
    for (int t0 = s0; t0 < s1; t0 += 2 * kTile) {
      park(rows, kl, vl);
      if (t0 + kTile < s1) fetch(rows, kbase, vbase, k_stride_n, v_stride_n, slots, kTile);  // flies during the step
      step(kl, vl, t0, s1, true, sm_scale, logit_cap);
      if (t0 + kTile >= s1) break;
      park(rows, kl, vl);
      slots = slots_next;
      if (t0 + 2 * kTile < s1) {
        fetch(rows, kbase, vbase, k_stride_n, v_stride_n, slots, 0);
        if (t0 + 4 * kTile < s1) slots_next = load_slots(idx, t0 + 4 * kTile, s0, s1);
      }
      step(kl, vl, t0 + kTile, s1, true, sm_scale, logit_cap);
    }
  }

  // ---- fp8 pools ---------------------------------------------------------------------------------------------
  // A pool row of 128 one-byte elements travels as 8-byte pieces (16 lanes per row) and is widened to T on its way into
  // the LDS tile, through a 256-entry table the block builds once (lut[code] = the code's exact value in T: every
  // fp8 value, its denormals, infinities and NaNs have one), so the tile step above runs unchanged.
  struct Rows8 {
    int2v k[8], v[8];
  };

  static __device__ __forceinline__ void fetch8(Rows8& r, const uint8_t* kbase, const uint8_t* vbase, int64_t k_stride_n,
                                                int64_t v_stride_n, int slots, int off) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, chunk = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t slot = __shfl(slots, off + 4 * i + g, 64);
      r.k[i] = *reinterpret_cast<const int2v*>(kbase + slot * k_stride_n + chunk * 8);
      r.v[i] = *reinterpret_cast<const int2v*>(vbase + slot * v_stride_n + chunk * 8);
    }
  }

  static __device__ __forceinline__ void park8(const Rows8& r, char* kl, char* vl, const uint16_t* lut) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, chunk = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 4 * i + g;
      *reinterpret_cast<int4v*>(kl + k_off(row, chunk)) = kv8_widen(lut, r.k[i][0], r.k[i][1]);
      *reinterpret_cast<int4v*>(vl + v_off(row, chunk * 16)) = kv8_widen(lut, r.v[i][0], r.v[i][1]);
    }
  }

  // run() over an fp8 pool (strides in bytes)
  __device__ __forceinline__ void run8(char* kl, char* vl, const uint16_t* lut, const uint8_t* kbase, const uint8_t* vbase,
                                       int64_t k_stride_n, int64_t v_stride_n, const int32_t* idx, int s0, int s1,
                                       float sm_scale, float logit_cap) {
    if (s1 <= s0) return;
    Rows8 r;
    int slots = load_slots(idx, s0, s0, s1);
    int slots_next = s0 + 2 * kTile < s1 ? load_slots(idx, s0 + 2 * kTile, s0, s1) : 0;
    fetch8(r, kbase, vbase, k_stride_n, v_stride_n, slots, 0);
    for (int t0 = s0; t0 < s1; t0 += 2 * kTile) {
      park8(r, kl, vl, lut);
      if (t0 + kTile < s1) fetch8(r, kbase, vbase, k_stride_n, v_stride_n, slots, kTile);  // flies during the step
      step(kl, vl, t0, s1, true, sm_scale, logit_cap);
      if (t0 + kTile >= s1) break;
      park8(r, kl, vl, lut);
      slots = slots_next;
      if (t0 + 2 * kTile < s1) {
        fetch8(r, kbase, vbase, k_stride_n, v_stride_n, slots, 0);
        if (t0 + 4 * kTile < s1) slots_next = load_slots(idx, t0 + 4 * kTile, s0, s1);
      }
      step(kl, vl, t0 + kTile, s1, true, sm_scale, logit_cap);
    }
  }

  // ---- K straight into operand registers -------------------------------------------------------------------
  // The K operand of S^T = K . Q^T is K-contiguous per lane: lane (hl, g) of half hh and k-step ks needs elements
  // 32 ks + 8 g .. + 8 of token 16 hh + hl, 16 contiguous bytes of that token's pool row.  So K needs no LDS at all;
  // only V (contracted over tokens, stored along the head dimension) goes through an LDS tile for its transposed
  // reads.  A wave then owns 8 KiB of LDS instead of 16.
  struct RowsKD {
    short8 k[2][4];  // operand of half hh, k-step ks
    int4v v[8];      // rows 4i + g on their way to the V tile
  };

  static __device__ __forceinline__ void fetch_kd(RowsKD& r, const uint16_t* kbase, const uint16_t* vbase,
                                                  int64_t k_stride_n, int64_t v_stride_n, int slots, int off) {
    const int lane = threadIdx.x & 63;
    const int hl = lane & 15, g = lane >> 4, chunk = lane & 15;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int64_t slot = __shfl(slots, off + 16 * hh + hl, 64);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        r.k[hh][ks] = *reinterpret_cast<const short8*>(kbase + slot * k_stride_n + 32 * ks + 8 * g);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t slot = __shfl(slots, off + 4 * i + g, 64);
      r.v[i] = *reinterpret_cast<const int4v*>(vbase + slot * v_stride_n + chunk * 8);
    }
  }

  static __device__ __forceinline__ void park_v(const RowsKD& r, char* vl) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, chunk = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<int4v*>(vl + v_off(4 * i + g, chunk * 16)) = r.v[i];
  }

  // step() with the K operands taken from r (must be called before r is refilled)
  __device__ __forceinline__ float4v score_half(const RowsKD& r, int hh) const {
    float4v acc = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = Mfma16<T>::mma(r.k[hh][ks], qf[ks], acc);
    return acc;
  }

  // softmax + O update from the two score halves and the parked V tile
  __device__ __forceinline__ void finish_step(const float4v (&sacc)[2], const char* vl, int t0, int s1, bool col_ok,
                                              float sm_scale, float logit_cap) {
    const int lane = threadIdx.x & 63;
    const int hl = lane & 15, g = lane >> 4;
    const float LOG2E = 1.4426950408889634f;
    const float qk_scale = (logit_cap > 0.f) ? sm_scale : sm_scale * LOG2E;
    float sv[8];
    float tmax = -INFINITY;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = sacc[hh][r] * qk_scale;
        if (logit_cap > 0.f) s = logit_cap * tanhf(s / logit_cap) * LOG2E;
        const int tok = t0 + 16 * hh + 4 * g + r;
        s = (col_ok && tok < s1) ? s : -INFINITY;
        sv[hh * 4 + r] = s;
        tmax = fmaxf(tmax, s);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m, tmax);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // nothing seen yet: everything stays at zero
    const float alpha = exp2f(m - m_use);
    m = m_new;
    float psum = 0.f;
    short8 pf;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float p = exp2f(sv[j] - m_use);
      psum += p;
      pf[j] = (short)Conv<T>::from_f32(p);
    }
    lsum = lsum * alpha + psum;
#pragma unroll
    for (int n = 0; n < 8; ++n) oacc[n] *= alpha;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int byte_in_row = 32 * n + 8 * (hl & 3);
      const short4v lo = lds_read_tr16(vl + v_off(4 * g + (hl >> 2), byte_in_row));
      const short4v hi = lds_read_tr16(vl + v_off(16 + 4 * g + (hl >> 2), byte_in_row));
      const short8 vf = short8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      oacc[n] = Mfma16<T>::mma(vf, pf, oacc[n]);
    }
  }

  // Stream tokens [s0, s1) of the slot list idx with K straight into registers and V through the 8 KiB tile at vl.
  struct KdSlots {
    int first, next;
  };
  // the slot loads of run_kd's first tiles, for a caller with work to place between them and the row loads
  static __device__ __forceinline__ KdSlots begin_kd(const int32_t* idx, int s0, int s1) {
    KdSlots k{0, 0};
    if (s1 <= s0) return k;
    k.first = load_slots(idx, s0, s0, s1);
    k.next = s0 + 2 * kTile < s1 ? load_slots(idx, s0 + 2 * kTile, s0, s1) : 0;
    return k;
  }
  __device__ __forceinline__ void run_kd(char* vl, const uint16_t* kbase, const uint16_t* vbase, int64_t k_stride_n,
                                         int64_t v_stride_n, const int32_t* idx, int s0, int s1, float sm_scale,
                                         float logit_cap) {
    run_kd(vl, kbase, vbase, k_stride_n, v_stride_n, idx, s0, s1, sm_scale, logit_cap, begin_kd(idx, s0, s1));
  }
  __device__ __forceinline__ void run_kd(char* vl, const uint16_t* kbase, const uint16_t* vbase, int64_t k_stride_n,
                                         int64_t v_stride_n, const int32_t* idx, int s0, int s1, float sm_scale,
                                         float logit_cap, KdSlots ks) {
    if (s1 <= s0) return;
    RowsKD r;
    int slots = ks.first;
    int slots_next = ks.next;
    fetch_kd(r, kbase, vbase, k_stride_n, v_stride_n, slots, 0);
    int off = 0;
    for (int t0 = s0; t0 < s1; t0 += kTile) {
      park_v(r, vl);
      float4v sacc[2];
      sacc[0] = score_half(r, 0);
      sacc[1] = score_half(r, 1);
      // the next tile's rows fly while this one is finished
      off += kTile;
      if (off == 2 * kTile) {
        off = 0;
        slots = slots_next;
      }
      if (t0 + kTile < s1) {
        fetch_kd(r, kbase, vbase, k_stride_n, v_stride_n, slots, off);
        if (off == 0 && t0 + 3 * kTile < s1) slots_next = load_slots(idx, t0 + 3 * kTile, s0, s1);
      }
      finish_step(sacc, vl, t0, s1, true, sm_scale, logit_cap);
    }
  }

  // total of the running sum over the four lane groups
  __device__ __forceinline__ float total() const {
    float s = lsum;
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    return s;
  }
};

}  // namespace sglk
