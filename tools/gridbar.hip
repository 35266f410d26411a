// Micro-benchmark: cost of a grid-wide barrier between co-resident blocks (one agent-scope counter, bounded polling),
// with and without a release / acquire fence pair around it, next to the cost of a dependent kernel boundary.
// Every poll loop is bounded and watches an abort word, so the grid drains even if a block never arrives.
// build: hipcc --offload-arch=gfx950 -O3 tools/gridbar.hip -o gpurun_out/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) k_gridbar(unsigned* counter, unsigned* abort_word, int iters, int fence, float* sink) {
  __shared__ int stop;
  if (threadIdx.x == 0) stop = 0;
  __syncthreads();
  float acc = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    acc = acc * 1.0001f + 1.0f;
    if (fence) __threadfence();  // release this block's earlier stores to the agent
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(i + 1) * gridDim.x;
      int polls = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++polls > 2000000 || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          stop = 1;
          break;
        }
      }
    }
    __syncthreads();
    if (fence) __threadfence();  // acquire
    if (stop) break;
  }
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) k_empty(float* sink) { sink[blockIdx.x * 256 + threadIdx.x] += 1.0f; }

int main() {
  unsigned* words;
  hipMalloc(&words, 256);
  float* sink;
  hipMalloc(&sink, 1024 * 256 * 4);
  hipMemset(sink, 0, 1024 * 256 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float ms;
  const int iters = 2000;
  for (int blocks : {64, 256, 512}) {
    for (int fence : {0, 1}) {
      hipMemset(words, 0, 256);
      hipDeviceSynchronize();
      hipEventRecord(a);
      k_gridbar<<<blocks, 256>>>(words, words + 16, iters, fence, sink);
      hipEventRecord(b);
      hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
      unsigned ab = 0;
      hipMemcpy(&ab, words + 16, 4, hipMemcpyDeviceToHost);
      printf("grid barrier  blocks=%4d fence=%d  %7.3f us per barrier%s\n", blocks, fence, ms * 1e3 / iters, ab ? "  (ABORTED)" : "");
      if (ab) return 1;
    }
  }
  // dependent kernel boundaries, replayed from a graph
  hipStream_t s;
  hipStreamCreate(&s);
  for (int blocks : {256, 1024}) {
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 200; ++i) k_empty<<<blocks, 256, 0, s>>>(sink);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("kernel boundary blocks=%4d (graph of 200 dependent launches)  %7.3f us per launch\n", blocks, ms * 1e3 / 2000);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
  }
  return 0;
}
