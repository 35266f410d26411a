/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's algorithms for the
 * RadixAttention + quantized-matmul decode path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product (libsglk.so) never does.
 *
 * Plain C, scalar loops, fp32 arithmetic unless the reference rounds to half precision at a given
 * point (then the rounding is restated here).  Every function cites the reference file:line whose
 * arithmetic it follows.  Pinned against the compiled reference CPU kernels (oracle/_ref), the
 * reference's Triton kernels run under the CPU interpreter, and the torch formulas held in the
 * reference's own tests -- see tests/test_oracle_golden.py and oracle/gen_golden.py.
 *
 * Build: gcc -O2 -fPIC -shared -fopenmp oracle/oracle.c -o oracle/liboracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ---------- half-precision helpers (round-to-nearest-even) ---------- */

static inline float bf16_to_f32(uint16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static inline uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); /* NaN stays NaN */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

static inline float f16_to_f32(uint16_t h) {
  const int s = h >> 15, e = (h >> 10) & 0x1f, m = h & 0x3ff;
  float r;
  if (e == 0)
    r = ldexpf((float)m, -24);
  else if (e == 31)
    r = m ? NAN : INFINITY;
  else
    r = ldexpf((float)(m | 0x400), e - 25);
  return s ? -r : r;
}

/* IEEE binary16, round-to-nearest-even, overflow to inf, gradual underflow */
static inline uint16_t f32_to_f16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return sign | 0x7e00;  /* NaN */
  if (a >= 0x47800000u) return sign | 0x7c00; /* >= 65536 -> inf (65520 rounds up below) */
  if (a < 0x33000000u) return sign;           /* < 2^-25 -> 0 */
  const float af = fabsf(f);
  if (a < 0x38800000u) { /* subnormal half: units of 2^-24 */
    const float r = nearbyintf(ldexpf(af, 24));
    return sign | (uint16_t)r;
  }
  uint32_t v = a + 0xfffu + ((a >> 13) & 1u); /* RNE on the 13 dropped bits */
  v = (v >> 13) - (112u << 10);               /* rebias 127 -> 15 */
  if (v >= 0x7c00u) return sign | 0x7c00;
  return sign | (uint16_t)v;
}

/* dtype codes shared with the Python wrapper: 0 = fp32, 1 = fp16, 2 = bf16 */
static inline float round_to(float v, int dtype) {
  if (dtype == 1) return f16_to_f32(f32_to_f16(v));
  if (dtype == 2) return bf16_to_f32(f32_to_bf16(v));
  return v;
}

ORACLE_API void oracle_round(float* x, int64_t n, int dtype) {
  for (int64_t i = 0; i < n; ++i) x[i] = round_to(x[i], dtype);
}

/* ---------- A0: kv index gather ----------
 * python/sglang/srt/layers/attention/utils.py:5-41 (create_flashinfer_kv_indices_triton):
 * kv_indices[kv_indptr[b] + j] = req_to_token[req_pool_indices[b], kv_start[b] + j],
 * j in [0, page_kernel_lens[b]).  kv_start may be NULL (zero). */
ORACLE_API void oracle_create_kv_indices(const int32_t* req_to_token, int64_t row_stride,
                                         const int64_t* req_pool_indices, const int64_t* lens,
                                         const int32_t* kv_indptr, const int64_t* kv_start,
                                         int32_t* kv_indices, int bs) {
  for (int b = 0; b < bs; ++b) {
    int64_t st = kv_start ? kv_start[b] : 0;
    const int32_t* row = req_to_token + req_pool_indices[b] * row_stride + st;
    int32_t* out = kv_indices + kv_indptr[b];
    for (int64_t j = 0; j < lens[b]; ++j) out[j] = row[j];
  }
}

/* python/sglang/srt/layers/attention/triton_backend.py:23-72 (get_num_kv_splits_triton).
 * All integer arithmetic except the log2 heuristic, which the reference does in fp32. */
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

ORACLE_API void oracle_get_num_kv_splits(int32_t* num_kv_splits, const int64_t* seq_lens,
                                         int num_seq, int num_group, int num_head, int num_kv_head,
                                         int max_kv_splits, int device_core_count) {
  int64_t max_len = 0, min_len;
  for (int i = 0; i < num_seq; ++i)
    if (seq_lens[i] > max_len) max_len = seq_lens[i];
  min_len = max_len;
  for (int i = 0; i < num_seq; ++i)
    if (seq_lens[i] < min_len) min_len = seq_lens[i];
  if (max_len * 8 < min_len * 10) min_len = max_len;
  int64_t splits1 = cdiv64(max_len, min_len);
  if (splits1 > max_kv_splits) splits1 = max_kv_splits;
  int64_t chunk1 = cdiv64(max_len, splits1);

  float ext_seq_len = (float)max_len / 64.0f;
  float lg = log2f(ext_seq_len);
  if (!(lg > 1.0f)) lg = 1.0f;
  int32_t ext_cores = (int32_t)((float)device_core_count * lg);
  int64_t block_h = 16, group = num_head / num_kv_head, token_grid;
  if (group == 1) {
    token_grid = (int64_t)num_seq * num_group * num_head;
  } else {
    if (group < block_h) block_h = group;
    token_grid = (int64_t)num_seq * num_group * cdiv64(num_head, block_h);
  }
  int64_t splits2 = cdiv64(ext_cores, token_grid);
  if (splits2 > max_kv_splits) splits2 = max_kv_splits;
  int64_t chunk2 = cdiv64(max_len, splits2);
  for (int i = 0; i < num_seq; ++i) {
    int64_t a = cdiv64(seq_lens[i], chunk1), b = cdiv64(seq_lens[i], chunk2);
    int32_t v = (int32_t)(a > b ? a : b);
    for (int g = 0; g < num_group; ++g) num_kv_splits[i * num_group + g] = v;
  }
}

/* ---------- A1: paged decode attention ----------
 * python/sglang/srt/layers/attention/triton_ops/decode_attention.py:50-174 (stage 1, MHA),
 * :246-406 (grouped), :497-557 (stage 2); sgl-kernel/csrc/cpu/decode.cpp:622-825,827-1051.
 * o[b,h,:] = softmax(scale * q[b,h] . K[idx]^T [tanh cap]) . V[idx], head h reads kv head
 * h / (Hq/Hkv).  Split-KV with the reference's split length rule (rounded up to 32) and an
 * LSE merge, so the summation order matches the reference structure; all math fp32.
 * q:[B,Hq,D] k_buf:[N,Hkv,D] v_buf:[N,Hkv,Dv] (contiguous fp32 copies), o:[B,Hq,Dv].
 * If lse_out != NULL it receives the final log-sum-exp per (b,h). */
ORACLE_API void oracle_decode_attention(const float* q, const float* k_buf, const float* v_buf,
                                        float* o, float* lse_out, const int32_t* kv_indptr,
                                        const int32_t* kv_indices, const int32_t* num_kv_splits,
                                        int B, int Hq, int Hkv, int D, int Dv, float sm_scale,
                                        float logit_cap) {
  const int group = Hq / Hkv;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int b = 0; b < B; ++b) {
    for (int h = 0; h < Hq; ++h) {
      const int kvh = h / group;
      const int32_t start = kv_indptr[b];
      const int32_t len = kv_indptr[b + 1] - start;
      int splits = num_kv_splits ? num_kv_splits[b] : 1;
      if (splits < 1) splits = 1;
      int per_split = (int)(cdiv64(cdiv64(len, splits), 32) * 32);
      float* acc = (float*)malloc(sizeof(float) * (size_t)Dv);
      float* part = (float*)malloc(sizeof(float) * (size_t)Dv);
      float g_max = -INFINITY, g_sum = 0.f;
      for (int d = 0; d < Dv; ++d) acc[d] = 0.f;
      const float* qv = q + ((size_t)b * Hq + h) * D;
      for (int s = 0; s < splits; ++s) {
        int s0 = per_split * s, s1 = s0 + per_split;
        if (s1 > len) s1 = len;
        if (s1 <= s0) continue;
        float e_max = -INFINITY, e_sum = 0.f;
        for (int d = 0; d < Dv; ++d) part[d] = 0.f;
        for (int t = s0; t < s1; ++t) {
          const int64_t loc = kv_indices[start + t];
          const float* kr = k_buf + ((size_t)loc * Hkv + kvh) * D;
          const float* vr = v_buf + ((size_t)loc * Hkv + kvh) * Dv;
          float qk = 0.f;
          for (int d = 0; d < D; ++d) qk += qv[d] * kr[d];
          qk *= sm_scale;
          if (logit_cap > 0.f) qk = logit_cap * tanhf(qk / logit_cap);
          float n_max = qk > e_max ? qk : e_max;
          float re = expf(e_max - n_max), p = expf(qk - n_max);
          for (int d = 0; d < Dv; ++d) part[d] = part[d] * re + p * vr[d];
          e_sum = e_sum * re + p;
          e_max = n_max;
        }
        /* stage 2 merge (decode_attention.py:534-551) on the normalised partial */
        float lse = e_max + logf(e_sum);
        float n_max = lse > g_max ? lse : g_max;
        float old = expf(g_max - n_max), w = expf(lse - n_max);
        for (int d = 0; d < Dv; ++d) acc[d] = acc[d] * old + w * (part[d] / e_sum);
        g_sum = g_sum * old + w;
        g_max = n_max;
      }
      float* orow = o + ((size_t)b * Hq + h) * Dv;
      for (int d = 0; d < Dv; ++d) orow[d] = acc[d] / g_sum;
      if (lse_out) lse_out[(size_t)b * Hq + h] = g_max + logf(g_sum);
      free(acc);
      free(part);
    }
  }
}

/* ---------- A2: extend (prefill with cached prefix) attention ----------
 * python/sglang/srt/layers/attention/triton_ops/extend_attention.py:42-303 and
 * sgl-kernel/csrc/cpu/extend.cpp:241-449: stage 1 over the cached prefix through kv_indices,
 * stage 2 causal over the contiguous new tokens.  q_ext:[T,Hq,D] k_ext:[T,Hkv,D]
 * v_ext:[T,Hkv,Dv]; qo_indptr:[B+1]; prefix described by kv_indptr/kv_indices. */
ORACLE_API void oracle_extend_attention(const float* q_ext, const float* k_ext, const float* v_ext,
                                        float* o_ext, const float* k_buf, const float* v_buf,
                                        const int32_t* qo_indptr, const int32_t* kv_indptr,
                                        const int32_t* kv_indices, int B, int Hq, int Hkv, int D,
                                        int Dv, float sm_scale, float logit_cap, int causal) {
  const int group = Hq / Hkv;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int b = 0; b < B; ++b) {
    for (int h = 0; h < Hq; ++h) {
      const int kvh = h / group;
      const int q0 = qo_indptr[b], nq = qo_indptr[b + 1] - q0;
      const int p0 = kv_indptr[b], np = kv_indptr[b + 1] - p0;
      float* acc = (float*)malloc(sizeof(float) * (size_t)Dv);
      for (int i = 0; i < nq; ++i) {
        const float* qv = q_ext + ((size_t)(q0 + i) * Hq + h) * D;
        float e_max = -INFINITY, e_sum = 0.f;
        for (int d = 0; d < Dv; ++d) acc[d] = 0.f;
        const int n_new = causal ? i + 1 : nq;
        for (int t = 0; t < np + n_new; ++t) {
          const float *kr, *vr;
          if (t < np) {
            const int64_t loc = kv_indices[p0 + t];
            kr = k_buf + ((size_t)loc * Hkv + kvh) * D;
            vr = v_buf + ((size_t)loc * Hkv + kvh) * Dv;
          } else {
            kr = k_ext + ((size_t)(q0 + t - np) * Hkv + kvh) * D;
            vr = v_ext + ((size_t)(q0 + t - np) * Hkv + kvh) * Dv;
          }
          float qk = 0.f;
          for (int d = 0; d < D; ++d) qk += qv[d] * kr[d];
          qk *= sm_scale;
          if (logit_cap > 0.f) qk = logit_cap * tanhf(qk / logit_cap);
          float n_max = qk > e_max ? qk : e_max;
          float re = expf(e_max - n_max), p = expf(qk - n_max);
          for (int d = 0; d < Dv; ++d) acc[d] = acc[d] * re + p * vr[d];
          e_sum = e_sum * re + p;
          e_max = n_max;
        }
        float* orow = o_ext + ((size_t)(q0 + i) * Hq + h) * Dv;
        for (int d = 0; d < Dv; ++d) orow[d] = acc[d] / e_sum;
      }
      free(acc);
    }
  }
}

/* ---------- A3: merge_state_v2 ----------
 * sgl-kernel/csrc/attention/merge_attn_states.cu:31-106: inf-safe LSE merge of two partial
 * attention outputs.  v:[n,h,d] s:[n,h]. */
ORACLE_API void oracle_merge_state(const float* v_a, const float* s_a, const float* v_b,
                                   const float* s_b, float* v_out, float* s_out, int64_t nh,
                                   int d) {
  for (int64_t i = 0; i < nh; ++i) {
    float pa = s_a[i], pb = s_b[i];
    if (isinf(pa)) pa = -INFINITY; /* merge_attn_states.cu:68-69 maps +-inf to -inf */
    if (isinf(pb)) pb = -INFINITY;
    float mx = pa > pb ? pa : pb;
    float ea = expf(pa - mx), eb = expf(pb - mx);
    float sum = ea + eb;
    float wa = ea / sum, wb = eb / sum;
    for (int j = 0; j < d; ++j) v_out[i * d + j] = v_a[i * d + j] * wa + v_b[i * d + j] * wb;
    if (s_out) s_out[i] = logf(sum) + mx;
  }
}

/* ---------- G1: AWQ int4 dequant (+ matmul) ----------
 * sgl-kernel/csrc/gemm/awq_kernel.cu:126-219 and the torch statement in
 * sgl-kernel/tests/test_awq_dequant.py:9-57.
 * out[r, 8c+j] = half( (nib(qweight[r,c], ORD[j]) - nib(qzeros[r/G,c], ORD[j])) * scales[r/G, 8c+j] )
 * with ORD = {0,4,1,5,2,6,3,7}: the subtraction is exact, the product is rounded once to the
 * scale dtype (awq_kernel.cu:149-156 fp16, :173 bf16).  scales passed as raw 16-bit words. */
static const int AWQ_ORD[8] = {0, 4, 1, 5, 2, 6, 3, 7};

ORACLE_API void oracle_awq_dequantize(const int32_t* qweight, const uint16_t* scales,
                                      const int32_t* qzeros, uint16_t* out, int rows, int qcols,
                                      int group_size, int dtype /*1 fp16, 2 bf16*/) {
#pragma omp parallel for
  for (int r = 0; r < rows; ++r) {
    const int g = r / group_size;
    for (int c = 0; c < qcols; ++c) {
      const uint32_t w = (uint32_t)qweight[(size_t)r * qcols + c];
      const uint32_t z = (uint32_t)qzeros[(size_t)g * qcols + c];
      for (int j = 0; j < 8; ++j) {
        const int sh = 4 * AWQ_ORD[j];
        const float diff = (float)((int)((w >> sh) & 0xF) - (int)((z >> sh) & 0xF));
        const uint16_t sraw = scales[(size_t)g * qcols * 8 + 8 * c + j];
        const float s = dtype == 1 ? f16_to_f32(sraw) : bf16_to_f32(sraw);
        const float p = diff * s; /* exact in fp32: 5-bit int times <=11-bit significand */
        out[(size_t)r * qcols * 8 + 8 * c + j] = dtype == 1 ? f32_to_f16(p) : f32_to_bf16(p);
      }
    }
  }
}

/* AWQLinearMethod.apply, python/sglang/srt/layers/quantization/awq.py:197-215:
 * out = x @ awq_dequantize(qweight, scales, qzeros) (+ bias), fp32 accumulation, rounded once
 * to the activation dtype.  x:[M,K] fp32 copy, w: dequantised [K,N] as 16-bit words. */
ORACLE_API void oracle_half_matmul(const float* x, const uint16_t* w, const float* bias,
                                   float* out, int M, int K, int N, int dtype) {
#pragma omp parallel for
  for (int m = 0; m < M; ++m) {
    float* acc = (float*)calloc((size_t)N, sizeof(float));
    for (int k = 0; k < K; ++k) {
      const float xv = x[(size_t)m * K + k];
      const uint16_t* wr = w + (size_t)k * N;
      if (dtype == 1)
        for (int n = 0; n < N; ++n) acc[n] += xv * f16_to_f32(wr[n]);
      else
        for (int n = 0; n < N; ++n) acc[n] += xv * bf16_to_f32(wr[n]);
    }
    for (int n = 0; n < N; ++n) out[(size_t)m * N + n] = round_to(acc[n] + (bias ? bias[n] : 0.f), dtype);
    free(acc);
  }
}

/* ---------- E1: elementwise ----------
 * rmsnorm / fused_add_rmsnorm: python/sglang/srt/layers/layernorm.py:88-107 (forward_native) and
 * sgl-kernel/tests/test_norm.py:8-33:
 *   x32 = x.float(); with a residual: x32 += residual.float(), residual = x32.to(dtype);
 *   x32 *= rsqrt(mean(x32^2) + eps); out = (x32 * weight.float()).to(dtype)
 * i.e. the weight multiply runs in fp32 on the unrounded normalised row (layernorm.py:105) and the
 * result is rounded once. */
ORACLE_API void oracle_rmsnorm(const float* x, const float* w, float* out, int64_t rows, int hidden,
                               float eps, int dtype) {
  for (int64_t r = 0; r < rows; ++r) {
    const float* xr = x + r * hidden;
    double ss = 0.0;
    for (int i = 0; i < hidden; ++i) ss += (double)xr[i] * xr[i];
    const float inv = 1.0f / sqrtf((float)(ss / hidden) + eps);
    for (int i = 0; i < hidden; ++i) out[r * hidden + i] = round_to(xr[i] * inv * w[i], dtype);
  }
}

ORACLE_API void oracle_fused_add_rmsnorm(float* x, float* residual, const float* w, int64_t rows,
                                         int hidden, float eps, int dtype) {
  for (int64_t r = 0; r < rows; ++r) {
    float* xr = x + r * hidden;
    float* rr = residual + r * hidden;
    double ss = 0.0;
    for (int i = 0; i < hidden; ++i) {
      const float s = xr[i] + rr[i];
      rr[i] = round_to(s, dtype);
      /* the reference normalises the UNROUNDED fp32 sum (layernorm.py:99-103) */
      xr[i] = s;
      ss += (double)s * s;
    }
    const float inv = 1.0f / sqrtf((float)(ss / hidden) + eps);
    for (int i = 0; i < hidden; ++i) xr[i] = round_to(xr[i] * inv * w[i], dtype);
  }
}

/* sgl-kernel/csrc/elementwise/activation.cu:22-50: out = silu(x[..., :d]) * x[..., d:],
 * silu(v) = v / (1 + exp(-v)) computed in fp32; the product silu(v)*g is formed on values already
 * rounded to the activation type: silu() returns T, then `T * T` (flashinfer act_and_mul_kernel). */
ORACLE_API void oracle_silu_and_mul(const float* x, float* out, int64_t rows, int d, int dtype) {
  for (int64_t r = 0; r < rows; ++r)
    for (int i = 0; i < d; ++i) {
      const float v = x[r * 2 * d + i], g = x[r * 2 * d + d + i];
      const float s = round_to(v / (1.0f + expf(-v)), dtype);
      out[r * d + i] = round_to(s * g, dtype);
    }
}

/* sgl-kernel/csrc/elementwise/rope.cu:22-89 + torch statement in
 * sgl-kernel/tests/test_rotary_embedding.py:30-84 (RotaryEmbedding.forward_native):
 * cos_sin_cache:[max_pos, rot] fp32, cos first half, sin second half.
 * neox: pairs (i, i+rot/2); interleaved: pairs (2i, 2i+1).  Dims >= rot pass through. */
ORACLE_API void oracle_rope(float* x, const int64_t* positions, const float* cos_sin, int64_t tokens,
                            int heads, int head_size, int rot, int interleave, int dtype) {
  const int half = rot / 2;
  for (int64_t t = 0; t < tokens; ++t) {
    const float* cs = cos_sin + positions[t] * rot;
    for (int h = 0; h < heads; ++h) {
      float* v = x + (t * heads + h) * head_size;
      for (int i = 0; i < half; ++i) {
        const int a = interleave ? 2 * i : i, b = interleave ? 2 * i + 1 : i + half;
        const float c = cs[i], s = cs[half + i];
        const float xa = v[a], xb = v[b];
        v[a] = round_to(xa * c - xb * s, dtype);
        v[b] = round_to(xb * c + xa * s, dtype);
      }
    }
  }
}

/* ---------- P1: set_kv_buffer ----------
 * python/sglang/srt/mem_cache/memory_pool.py:378-412: k_buffer[layer][loc] = cache_k (index put). */
ORACLE_API void oracle_set_kv_buffer(float* k_buf, float* v_buf, const float* k, const float* v,
                                     const int64_t* loc, int64_t tokens, int row_k, int row_v) {
  for (int64_t t = 0; t < tokens; ++t) {
    memcpy(k_buf + loc[t] * row_k, k + t * row_k, sizeof(float) * (size_t)row_k);
    memcpy(v_buf + loc[t] * row_v, v + t * row_v, sizeof(float) * (size_t)row_v);
  }
}

/* ---------- G2/G3: fp8 (OCP e4m3fn) quantisation and scaled matmul ---------- */

/* float -> e4m3fn, round-to-nearest-even, saturating to +-448 (the reference clamps before the
 * cast: per_token_quant_fp8.cu:60-71), NaN -> 0x7f. */
static uint8_t f32_to_e4m3(float f) {
  if (isnan(f)) return 0x7f;
  uint8_t sign = signbit(f) ? 0x80 : 0;
  float a = fabsf(f);
  if (a >= 448.f) return sign | 0x7e;
  if (a < 0.0009765625f) return sign; /* below half of the smallest subnormal 2^-9 -> 0 */
  int e;
  float m = frexpf(a, &e); /* a = m * 2^e, m in [0.5,1) */
  int exp = e - 1;         /* a = (2m) * 2^exp, 2m in [1,2) */
  if (exp < -6) {
    /* subnormal: value = k * 2^-9, k in 0..7 */
    float k = a * 512.f;
    float r = nearbyintf(k);
    if (r >= 8.f) return sign | 0x08;
    return sign | (uint8_t)r;
  }
  float frac = (2.f * m - 1.f) * 8.f; /* 3 mantissa bits */
  float r = nearbyintf(frac);
  int mant = (int)r;
  if (mant == 8) {
    mant = 0;
    exp += 1;
  }
  int be = exp + 7;
  if (be > 15 || (be == 15 && mant == 7)) return sign | 0x7e;
  return sign | (uint8_t)((be << 3) | mant);
}

static float e4m3_to_f32(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 0xF, m = v & 7;
  float r;
  if (e == 15 && m == 7) return NAN;
  if (e == 0)
    r = ldexpf((float)m, -9);
  else
    r = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -r : r;
}

ORACLE_API void oracle_e4m3_decode(const uint8_t* q, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = e4m3_to_f32(q[i]);
}
ORACLE_API void oracle_e4m3_encode(const float* x, uint8_t* q, int64_t n) {
  for (int64_t i = 0; i < n; ++i) q[i] = f32_to_e4m3(x[i]);
}

/* sgl-kernel/csrc/gemm/per_token_quant_fp8.cu:10-106: scale = absmax(row) / FP8_MAX,
 * q = cast(clamp(x / scale ... )).  The kernel multiplies by scale_inv = 1/scale (:56-58). */
ORACLE_API void oracle_per_token_quant_fp8(const float* x, uint8_t* q, float* s, int64_t rows,
                                           int hidden, float fp8_max) {
  for (int64_t r = 0; r < rows; ++r) {
    float amax = 0.f;
    for (int i = 0; i < hidden; ++i) amax = fmaxf(amax, fabsf(x[r * hidden + i]));
    const float scale = amax / fp8_max;
    s[r] = scale;
    const float inv = 1.0f / scale;
    for (int i = 0; i < hidden; ++i) {
      float v = x[r * hidden + i] * inv;
      v = fmaxf(fminf(v, fp8_max), -fp8_max);
      q[r * hidden + i] = f32_to_e4m3(v);
    }
  }
}

/* sgl-kernel/csrc/gemm/per_token_group_quant_8bit.cu:19-197: per (token, group):
 * absmax = max(eps, |x|), y_s = absmax / fp8_max, q = clamp(x / y_s, fp8_min, fp8_max). */
ORACLE_API void oracle_per_token_group_quant_fp8(const float* x, uint8_t* q, float* s, int64_t rows,
                                                 int hidden, int group, float eps, float fp8_min,
                                                 float fp8_max) {
  const int ng = hidden / group;
  for (int64_t r = 0; r < rows; ++r)
    for (int g = 0; g < ng; ++g) {
      const float* xr = x + r * hidden + g * group;
      float amax = eps;
      for (int i = 0; i < group; ++i) amax = fmaxf(amax, fabsf(xr[i]));
      const float ys = amax / fp8_max;
      s[r * ng + g] = ys;
      for (int i = 0; i < group; ++i) {
        float v = xr[i] / ys;
        v = fmaxf(fminf(v, fp8_max), fp8_min);
        q[r * hidden + g * group + i] = f32_to_e4m3(v);
      }
    }
}

/* sgl-kernel/tests/test_fp8_gemm.py:6-14 (torch_scaled_mm):
 * out = ((A.f32 @ B.f32) * scale_a[:,None] * scale_b[None,:] + bias).to(out_dtype).
 * a:[M,K] e4m3 row-major; b given as [N,K] row-major (= the reference's column-major [K,N]). */
ORACLE_API void oracle_fp8_scaled_mm(const uint8_t* a, const uint8_t* b_nk, const float* sa,
                                     const float* sb, const float* bias, float* out, int M, int N,
                                     int K, int dtype) {
  float lut[256];
  for (int i = 0; i < 256; ++i) lut[i] = e4m3_to_f32((uint8_t)i);
#pragma omp parallel for
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      float acc = 0.f;
      const uint8_t* ar = a + (size_t)m * K;
      const uint8_t* br = b_nk + (size_t)n * K;
      for (int k = 0; k < K; ++k) acc += lut[ar[k]] * lut[br[k]];
      float v = acc * sa[m] * sb[n];
      if (bias) v += bias[n];
      out[(size_t)m * N + n] = round_to(v, dtype);
    }
}

/* sgl-kernel/tests/test_fp8_blockwise_gemm.py:19-59 (native_w8a8_block_fp8_matmul):
 * per 128-wide K block: C[:, nblk] += (A_blk.f32 @ B_blk.f32^T) * As[:, kb] * Bs[nb, kb].
 * a:[M,K], as:[M,K/bk]; b:[N,K], bs:[N/bn, K/bk]. */
ORACLE_API void oracle_fp8_blockwise_mm(const uint8_t* a, const uint8_t* b_nk, const float* as_,
                                        const float* bs, float* out, int M, int N, int K, int bn,
                                        int bk, int dtype) {
  float lut[256];
  for (int i = 0; i < 256; ++i) lut[i] = e4m3_to_f32((uint8_t)i);
  const int kt = (K + bk - 1) / bk;
#pragma omp parallel for
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      float tot = 0.f;
      for (int kb = 0; kb < kt; ++kb) {
        float acc = 0.f;
        int k1 = (kb + 1) * bk;
        if (k1 > K) k1 = K;
        for (int k = kb * bk; k < k1; ++k) acc += lut[a[(size_t)m * K + k]] * lut[b_nk[(size_t)n * K + k]];
        tot += acc * as_[(size_t)m * kt + kb] * bs[(size_t)(n / bn) * kt + kb];
      }
      out[(size_t)m * N + n] = round_to(tot, dtype);
    }
}

/* ---------- C1: all-reduce ----------
 * sgl-kernel/csrc/allreduce/custom_all_reduce_hip.cuh:260-282: every rank sums all ranks'
 * buffers in fp32 in rank order 0..W-1 and casts once. bufs: W pointers to fp32 copies. */
ORACLE_API void oracle_allreduce_sum(const float* const* bufs, int world, float* out, int64_t n,
                                     int dtype) {
  for (int64_t i = 0; i < n; ++i) {
    float acc = 0.f;
    for (int w = 0; w < world; ++w) acc += bufs[w][i];
    out[i] = round_to(acc, dtype);
  }
}


/* ------------------------------------------------------------------------------------------------
 * MoE routing helpers.
 * moe_align_block_size: sgl-kernel/csrc/moe/moe_align_kernel.cu:43-100 (counts, padded prefix sums, expert id per
 * block) with the placement of :27-41 made deterministic: slots of one expert in order of appearance (parity
 * unpinned for that order: the reference's is whatever its atomics produce).
 * ------------------------------------------------------------------------------------------------ */
void oracle_moe_align_block_size(const int64_t* topk_ids, int64_t numel, int num_experts, int block_size,
                                 int32_t* sorted_token_ids, int32_t* expert_ids, int32_t* num_tokens_post_pad) {
  int32_t* counts = (int32_t*)calloc((size_t)num_experts, sizeof(int32_t));
  int32_t* cumsum = (int32_t*)calloc((size_t)num_experts + 1, sizeof(int32_t));
  for (int64_t i = 0; i < numel; ++i)
    if (topk_ids[i] >= 0 && topk_ids[i] < num_experts) counts[topk_ids[i]]++;
  for (int e = 0; e < num_experts; ++e)
    cumsum[e + 1] = cumsum[e] + (counts[e] + block_size - 1) / block_size * block_size;
  *num_tokens_post_pad = cumsum[num_experts];
  for (int e = 0; e < num_experts; ++e)
    for (int32_t i = cumsum[e]; i < cumsum[e + 1]; i += block_size) expert_ids[i / block_size] = e;
  for (int e = 0; e < num_experts; ++e) counts[e] = 0;
  for (int64_t i = 0; i < numel; ++i) {
    const int64_t e = topk_ids[i];
    if (e >= 0 && e < num_experts) sorted_token_ids[cumsum[e] + counts[e]++] = (int32_t)i;
  }
  free(counts);
  free(cumsum);
}

/* topk_softmax: softmax over the experts, k rounds of arg-max, ties to the lowest expert id
 * (sgl-kernel/csrc/moe/moe_topk_softmax_kernels.cu:104-160). */
void oracle_topk_softmax(const float* gating, float* weights, int32_t* ids, int32_t* source_rows, int rows,
                         int num_experts, int k) {
  float* p = (float*)malloc((size_t)num_experts * sizeof(float));
  for (int r = 0; r < rows; ++r) {
    const float* g = gating + (int64_t)r * num_experts;
    float mx = -INFINITY, sum = 0.f;
    for (int e = 0; e < num_experts; ++e) mx = fmaxf(mx, g[e]);
    for (int e = 0; e < num_experts; ++e) sum += expf(g[e] - mx);
    for (int e = 0; e < num_experts; ++e) p[e] = g[e];
    for (int j = 0; j < k; ++j) {
      int best = 0;
      for (int e = 1; e < num_experts; ++e)
        if (p[e] > p[best]) best = e;
      weights[(int64_t)r * k + j] = expf(p[best] - mx) / sum;
      ids[(int64_t)r * k + j] = best;
      source_rows[(int64_t)r * k + j] = j * rows + r;
      p[best] = -INFINITY;
    }
  }
  free(p);
}
