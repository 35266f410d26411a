// Calibration micro-benchmarks: MFMA issue rate, VALU issue rate, s_barrier cost, LDS-DMA round trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr;

__global__ void __launch_bounds__(256) k_mfma(float* out, int iters) {
  bf8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i); }
  float4v c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void __launch_bounds__(256) k_valu(float* out, int iters) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  for (int i = 0; i < iters; ++i) {
    x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}
__global__ void __launch_bounds__(256) k_barrier(long long* out, int iters) {
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_barrier();
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
// one wave issues `depth` 1-KiB LDS-DMA pieces then waits for all: cycles per round
__global__ void __launch_bounds__(256) k_glds(const char* src, long long* out, int iters, int depth, long long stride) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const char* p = src + ((long long)blockIdx.x * 4 + wid) * stride + lane * 16;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    for (int d = 0; d < depth; ++d)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + ((long long)i * depth + d) * 1024),
                                       (lds_ptr)(lds + (wid * 16 + (d & 15)) * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 4 + wid] = t1 - t0;
}
int main() {
  float* out; hipMalloc(&out, 4 << 20);
  long long* lo; hipMalloc(&lo, 1 << 20);
  char* src; size_t sbytes = (size_t)2 << 30; hipMalloc(&src, sbytes); hipMemset(src, 1, sbytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float ms;
  for (int blocks : {256, 512, 1024}) {
    int iters = 20000;
    k_mfma<<<blocks, 256>>>(out, 100);
    hipEventRecord(a); k_mfma<<<blocks, 256>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    double fl = (double)blocks * 4 * iters * 4 * 16 * 16 * 32 * 2;
    printf("mfma  blocks=%d  %.3f ms  %.1f TFLOP/s  (%.2f cycles@2.4GHz per mfma per wave)\n", blocks, ms, fl / ms / 1e9,
           ms * 1e-3 * 2.4e9 / (iters * 4.0) / ((blocks + 255) / 256));
  }
  for (int blocks : {256, 512, 2048}) {
    int iters = 100000;
    hipEventRecord(a); k_valu<<<blocks, 256>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("valu  blocks=%d  %.3f ms  %.2f cycles@2.4GHz per fma per wave (waves/SIMD=%d)\n", blocks, ms,
           ms * 1e-3 * 2.4e9 / (iters * 4.0), (blocks + 255) / 256);
  }
  {
    int iters = 10000;
    k_barrier<<<256, 256>>>(lo, iters); hipDeviceSynchronize();
    long long h; hipMemcpy(&h, lo, 8, hipMemcpyDeviceToHost);
    printf("barrier: %.1f memtime ticks per s_barrier (4 waves)\n", (double)h / iters);
  }
  for (int depth : {1, 4, 16}) {
    for (int mode = 0; mode < 2; ++mode) {
      int iters = 200; int blocks = 256;
      long long stride = mode == 0 ? 0 : (long long)iters * depth * 1024;  // 0: all waves same lines (L2 hits)
      if ((long long)blocks * 4 * stride > (long long)sbytes) continue;
      hipEventRecord(a);
      k_glds<<<blocks, 256, 65536>>>(src, lo, iters, depth, stride);
      hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      std::vector<long long> h(blocks * 4); hipMemcpy(h.data(), lo, blocks * 4 * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (auto v : h) avg += v; avg /= h.size();
      printf("glds depth=%2d %s: %.0f ticks per round (%.3f ms total, %.1f GB/s)\n", depth, mode ? "HBM stream" : "same lines",
             avg / iters, ms, (double)blocks * 4 * iters * depth * 1024 / ms / 1e6);
    }
  }
  return 0;
}
