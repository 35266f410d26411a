// Micro-benchmark: do VALU instructions of one wave run in the shadow of another wave's (or their own) MFMAs?
// 512 threads per block = 2 waves per SIMD, one block per CU.  Per iteration a wave issues 16 MFMAs
// (16 independent accumulators) and / or 48 integer VALU instructions shaped like the int4 dequantisation
// (shift, and-or), in blocks of 4 MFMA + 12 VALU.
// build: hipcc --offload-arch=gfx950 -O3 tools/coissue.hip -o tools/coissue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef int int4v __attribute__((ext_vector_type(4)));

template <int DO_MFMA, int DO_VALU, int PIPE = 0>
__global__ void __launch_bounds__(512) k_mix(float* out, const int* in, int iters) {
  bf8 a;
  for (int i = 0; i < 8; ++i) a[i] = (__bf16)(float)((threadIdx.x + i) & 7);
  float4v c[16];
  for (int i = 0; i < 16; ++i) c[i] = float4v{0, 0, 0, 0};
  unsigned w0 = in[threadIdx.x], w1 = in[threadIdx.x + 512], w2 = in[threadIdx.x + 1024], w3 = in[threadIdx.x + 1536];
  unsigned mask = 0x000F000Fu, orc = 0x43004300u;
  asm volatile("" : "+v"(mask), "+v"(orc));
  int4v b = {(int)orc, (int)orc, (int)orc, (int)orc};
  if (!PIPE) {
  for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (DO_VALU) {
          const unsigned w = t == 0 ? w0 : t == 1 ? w1 : t == 2 ? w2 : w3;
#pragma unroll
          for (int r = 0; r < 2; ++r) {  // 2 x (3 shifts + 4 and-or) ~ 14 VALU per strip
            b[0] = (int)(((w >> (0 + r)) & mask) | orc);
            b[1] = (int)(((w >> (4 + r)) & mask) | orc);
            b[2] = (int)(((w >> (8 + r)) & mask) | orc);
            b[3] = (int)(((w >> (12 + r)) & mask) | orc);
            asm volatile("" : "+v"(b));
          }
          w0 += 0x11; w1 += 0x13; w2 += 0x17; w3 += 0x19;
        }
        if (DO_MFMA) {
          bf8 bb = __builtin_bit_cast(bf8, b);
#pragma unroll
          for (int q = 0; q < 4; ++q) c[4 * t + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, c[4 * t + q], 0, 0, 0);
        }
      }
    }
  } else {
    // software-pipelined: the operand of strip t + 1 is built while the MFMAs of strip t (operand from the previous
    // step) are in flight, one VALU pair after each MFMA
    int4v bn = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const unsigned w = t == 0 ? w0 : t == 1 ? w1 : t == 2 ? w2 : w3;
        bf8 bb = __builtin_bit_cast(bf8, b);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          c[4 * t + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, c[4 * t + q], 0, 0, 0);
          bn[q] = (int)(((w >> (4 * q)) & mask) | orc);
          bn[(q + 1) & 3] ^= (int)(((w >> (4 * q + 1)) & mask) | orc);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        b = bn;
        w0 += 0x11; w1 += 0x13; w2 += 0x17; w3 += 0x19;
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s + (float)(b[0] + b[3] + (int)w0);
}

template <int M, int V, int P = 0>
static void run(const char* name, float* out, int* in, int blocks) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int iters = 20000;
  k_mix<M, V, P><<<blocks, 512>>>(out, in, 100);
  hipEventRecord(a);
  k_mix<M, V, P><<<blocks, 512>>>(out, in, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  printf("%-22s blocks=%d  %8.1f ns per iteration (16 MFMA and/or ~64 VALU per wave, 2 waves per SIMD)\n", name, blocks,
         ms * 1e6 / iters);
}

int main() {
  float* out;
  int* in;
  hipMalloc(&out, 1024 * 512 * 4);
  hipMalloc(&in, 2048 * 4);
  hipMemset(in, 0x5a, 2048 * 4);
  for (int blocks : {256}) {
    run<1, 0>("mfma only", out, in, blocks);
    run<0, 1>("valu only", out, in, blocks);
    run<1, 1>("mfma + valu", out, in, blocks);
    run<1, 1, 1>("mfma + valu pipelined", out, in, blocks);
  }
  return 0;
}
