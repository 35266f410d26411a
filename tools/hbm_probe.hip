// Achievable HBM bandwidth on this GPU with hand-written streaming kernels (16 bytes per lane, grid-stride, eight
// independent loads in flight per lane):  read-only sweep, read-only sweep with non-temporal loads, and copy.
// Buffers are far larger than the 256 MiB Infinity Cache.  Prints one JSON line.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_probe.bin tools/hbm_probe.hip && tools/hbm_probe.bin [GiB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float float4v __attribute__((ext_vector_type(4)));

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                       \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <bool NT>
__global__ void __launch_bounds__(512) read_kernel(const float4v* __restrict__ src, size_t n16, float* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * 512;
  size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
  float4v acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + 7 * stride < n16; i += 8 * stride) {
    float4v v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; i < n16; i += stride) acc += src[i];
  const float s = acc[0] + acc[1] + acc[2] + acc[3];
  if (s == 123456.789f) sink[0] = s;  // keeps the loads alive; never true on the fill used here
}

__global__ void __launch_bounds__(512) copy_kernel(const float4v* __restrict__ src, float4v* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 512;
  size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    float4v v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) dst[i + u * stride] = v[u];
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 4.0;
  const size_t bytes = (size_t)(gib * (1ull << 30)) / 16 * 16;
  const size_t n16 = bytes / 16;
  float4v *a, *b;
  float* sink;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 0x11, bytes));
  CK(hipMemset(b, 0x22, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 10;
  printf("{\"probe\": \"hbm\", \"buffer_gib\": %.2f", gib);
  for (int blocks_per_cu : {2, 4}) {
    const int grid = 256 * blocks_per_cu;  // 512-thread blocks: 16 or 32 waves per CU
    auto time = [&](auto&& launch) {
      launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < iters; ++i) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      return ms / iters;
    };
    const float rd = time([&] { hipLaunchKernelGGL(read_kernel<false>, dim3(grid), dim3(512), 0, 0, a, n16, sink); });
    const float rdnt = time([&] { hipLaunchKernelGGL(read_kernel<true>, dim3(grid), dim3(512), 0, 0, a, n16, sink); });
    const float cp = time([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(512), 0, 0, a, b, n16); });
    printf(", \"waves_per_cu_%d\": {\"read_TBps\": %.3f, \"read_nt_TBps\": %.3f, \"copy_TBps_read_plus_write\": %.3f}",
           8 * blocks_per_cu, bytes / rd / 1e9, bytes / rdnt / 1e9, 2.0 * bytes / cp / 1e9);
  }
  printf("}\n");
  return 0;
}
