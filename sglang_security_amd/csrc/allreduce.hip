// Peer-to-peer all-reduce over xGMI on IPC-mapped buffers (one process per GPU).
//
// Replaces the ROCm op set of sgl-kernel/csrc/torch_extension_rocm.cc:25-55 (device code
// custom_all_reduce_hip.cuh:155-575).  Every rank maps every other rank's signal block and registered
// input buffers into its own address space; a reduction is then one kernel per rank:
//
//   one-stage  : barrier, every rank sums all ranks' inputs (fp32, rank order 0..W-1, one rounding)
//                into its own output, barrier (inputs may be overwritten afterwards).
//   two-stage  : barrier, rank r reduces slice r of all inputs into its own staging area (behind its
//                signal block), barrier with release/acquire, every rank gathers all slices from the
//                peers' staging areas.
//
// Because every element is summed in the same order by whichever rank owns it, all ranks end with
// bit-identical results.
//
// Barriers: per (block, peer) 32-bit tickets in uncached memory.  Barrier number n of block b writes n
// into slot [n & 1][b][my rank] of every peer and waits for n in its own slots; two slot sets keep a
// fast rank's next barrier from overwriting a ticket a slow rank has not read yet.
//
// A wait that sees no peer within the communicator's time limit (wall clock, 100 MHz counter) does not hang the GPU
// and does not return a sum either: the block records itself in `stuck`, raises the rank's sticky `failed` word and
// a word in host-pinned memory, and fills its share of the output with NaN (all bits set).  Every later launch on
// that communicator poisons its output at once without waiting, and the host entry points return SGLK_ESTUCK as
// soon as they see the host word.  Peers waiting for the failed rank time out the same way, so the failure reaches
// every rank.  Recovery is a new communicator.
#include <array>
#include <cstdlib>
#include <cstring>
#include <map>
#include <cstddef>
#include <mutex>
#include <vector>

#include "common.h"

namespace sglk {
namespace {

constexpr int kMaxRanks = 8;
constexpr int kMaxBlocks = 64;       // tickets and slots per rank: the most a launch may use
constexpr int kDefaultBlocks = 16;   // what a launch uses: the reference's ROCm block_limit (custom_all_reduce_hip.cuh:507-509)
constexpr int kThreads = 512;
constexpr int64_t kDefaultTimeoutMs = 30000;
constexpr int64_t kTicksPerMs = 100000;  // s_memrealtime counts at 100 MHz

struct Signal {
  alignas(128) uint32_t ticket[kMaxBlocks];                // barriers this block has passed (owner only)
  alignas(128) uint32_t slot[2][kMaxBlocks][kMaxRanks];    // written by peers
  alignas(128) uint32_t stuck[kMaxBlocks];                 // non-zero: a wait of this block timed out
  alignas(128) uint32_t failed;                            // sticky: some wait on this rank timed out
};

// distributed/custom_all_reduce.py (stuck_detail) reads these words through the meta buffer after a time-out
static_assert(offsetof(Signal, slot) == 256 && offsetof(Signal, stuck) == 4352 && offsetof(Signal, failed) == 4608,
              "Signal layout is read by the Python diagnostics");

struct Watch {
  uint32_t* host_flag;  // host-pinned word, device address
  uint64_t ticks;       // time limit of one wait
};

struct Peers {
  Signal* sig[kMaxRanks];
};
struct Inputs {
  const void* ptr[kMaxRanks];
};

// False (for the whole block) when a peer did not arrive in time or the communicator has failed before.
template <bool RELEASE_ACQUIRE>
__device__ __forceinline__ bool barrier_all(const Peers& peers, Signal* self, int rank, int world, const Watch& watch) {
  __shared__ uint32_t bad;
  if (threadIdx.x == 0) bad = __hip_atomic_load(&self->failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();  // also: all of the block's prior stores are issued
  if (bad) return false;
  const uint32_t n = self->ticket[blockIdx.x] + 1;
  if ((int)threadIdx.x < world) {
    uint32_t* remote = &peers.sig[threadIdx.x]->slot[n & 1][blockIdx.x][rank];
    if (RELEASE_ACQUIRE)
      __hip_atomic_store(remote, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    else
      __hip_atomic_store(remote, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t* mine = &self->slot[n & 1][blockIdx.x][threadIdx.x];
    const uint64_t t0 = wall_clock64();
    uint32_t polls = 0;
    // The polls are relaxed loads of uncached memory; the acquire (an invalidate of this GPU's caches) is issued
    // once, after the peer's ticket has been seen.  An acquire per poll would keep invalidating the caches under
    // whatever else runs on the chip for as long as the wait lasts (ranks that share one GPU slow the straggler
    // they wait for).  After a thousand quick polls the wait backs off to long sleeps.
    while (true) {
      const uint32_t v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      // tickets only grow; a peer that is already one barrier ahead wrote the other slot set
      if (v == n) break;
      if ((++polls & 63u) == 0 && wall_clock64() - t0 > watch.ticks) {
        self->stuck[blockIdx.x] = n;
        __hip_atomic_store(&self->failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(watch.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        bad = 1;
        break;
      }
      if (polls < 1024u)
        __builtin_amdgcn_s_sleep(2);
      else
        __builtin_amdgcn_s_sleep(64);
    }
    if (RELEASE_ACQUIRE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // "" = system scope
  }
  __syncthreads();
  if (bad) return false;
  if (threadIdx.x == 0) self->ticket[blockIdx.x] = n;
  __syncthreads();
  return true;
}

// all bits set: NaN in fp32, fp16 and bf16
__device__ __forceinline__ void poison(void* out, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * kThreads)
    reinterpret_cast<uint4*>(out)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
}

template <typename T>
__device__ __forceinline__ void add_vec(float (&acc)[8], const uint4& v) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  if constexpr (sizeof(typename Raw<T>::type) == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[2 * j] += Conv<T>::to_f32((uint16_t)(w[j] & 0xffffu));
      acc[2 * j + 1] += Conv<T>::to_f32((uint16_t)(w[j] >> 16));
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += __uint_as_float(w[j]);
  }
}

template <typename T>
__device__ __forceinline__ uint4 pack_vec(const float (&acc)[8]) {
  uint32_t w[4];
  if constexpr (sizeof(typename Raw<T>::type) == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      w[j] = (uint32_t)Conv<T>::from_f32(acc[2 * j]) | ((uint32_t)Conv<T>::from_f32(acc[2 * j + 1]) << 16);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = __float_as_uint(acc[j]);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// sum of 16-byte vector `i` over all ranks, rank order, fp32.  W > 0: the group size at compile time, so that the W
// loads (W - 1 of them cross a link, a couple of microseconds each) are in flight together before the first add;
// W = 0 takes `world` at run time (one load per trip of the loop).
template <typename T, int W>
__device__ __forceinline__ uint4 reduce_vec(const Inputs& in, int world, int64_t i) {
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (W > 0) {
    uint4 v[W];
#pragma unroll
    for (int r = 0; r < W; ++r) v[r] = reinterpret_cast<const uint4*>(in.ptr[r])[i];
#pragma unroll
    for (int r = 0; r < W; ++r) add_vec<T>(acc, v[r]);
  } else {
    for (int r = 0; r < world; ++r) add_vec<T>(acc, reinterpret_cast<const uint4*>(in.ptr[r])[i]);
  }
  return pack_vec<T>(acc);
}

// The peer pointers of an input live in device memory (`slot`, one Inputs record per registered buffer) and not in
// the kernel arguments: a launch captured into a graph names a record that register_graph_buffers fills in only
// after the capture has ended (custom_all_reduce_hip.cuh:453-476), so the record is read when the kernel runs.
template <typename T, int W>
__global__ void __launch_bounds__(kThreads) allreduce_one_stage(const Inputs* __restrict__ slot, Peers peers, Signal* self,
                                                                void* out, int rank, int world, int64_t nvec, Watch watch) {
  const Inputs in = *slot;
  if (!barrier_all<true>(peers, self, rank, world, watch)) return poison(out, nvec);
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * kThreads)
    reinterpret_cast<uint4*>(out)[i] = reduce_vec<T, W>(in, world, i);
  if (!barrier_all<false>(peers, self, rank, world, watch)) poison(out, nvec);
}

template <typename T, int W>
__global__ void __launch_bounds__(kThreads) allreduce_two_stage(const Inputs* __restrict__ slot, Peers peers, Signal* self,
                                                                void* out, int rank, int world, int64_t nvec, Watch watch) {
  const Inputs in = *slot;
  const int64_t part = nvec / world;  // the last rank also takes the remainder
  const int64_t lo = part * rank, hi = rank == world - 1 ? nvec : lo + part;
  uint4* stage = reinterpret_cast<uint4*>(self + 1);
  if (!barrier_all<true>(peers, self, rank, world, watch)) return poison(out, nvec);
  for (int64_t i = lo + (int64_t)blockIdx.x * kThreads + threadIdx.x; i < hi; i += (int64_t)gridDim.x * kThreads)
    stage[i - lo] = reduce_vec<T, W>(in, world, i);
  if (!barrier_all<true>(peers, self, rank, world, watch)) return poison(out, nvec);
  if constexpr (W > 0) {
    // one vector of every rank's slice per trip: W loads in flight (W - 1 over the links), then W stores; the own
    // slice first and the ring order after it, so that the links are used evenly
    const int64_t longest = nvec - part * (W - 1);
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < longest; i += (int64_t)gridDim.x * kThreads) {
      uint4 v[W];
#pragma unroll
      for (int r = 0; r < W; ++r) {
        const int src = (rank + r) % W;
        if (i < part || src == W - 1) v[r] = reinterpret_cast<const uint4*>(peers.sig[src] + 1)[i];
      }
#pragma unroll
      for (int r = 0; r < W; ++r) {
        const int src = (rank + r) % W;
        if (i < part || src == W - 1) reinterpret_cast<uint4*>(out)[part * src + i] = v[r];
      }
    }
  } else {
    for (int r = 0; r < world; ++r) {
      const int src = (rank + r) % world;
      const int64_t slo = part * src, shi = src == world - 1 ? nvec : slo + part;
      const uint4* from = reinterpret_cast<const uint4*>(peers.sig[src] + 1);
      for (int64_t i = slo + (int64_t)blockIdx.x * kThreads + threadIdx.x; i < shi; i += (int64_t)gridDim.x * kThreads)
        reinterpret_cast<uint4*>(out)[i] = from[i - slo];
    }
  }
}

// out[r * nvec + i] = rank r's input vector i: every rank's registered input, concatenated in rank order.
// W as in reduce_vec: with the group size known, one vector from every rank is in flight per trip.
template <int W>
__global__ void __launch_bounds__(kThreads) allgather_kernel(const Inputs* __restrict__ slot, Peers peers, Signal* self,
                                                             void* out, int rank, int world, int64_t nvec, Watch watch) {
  const Inputs in = *slot;
  if (!barrier_all<true>(peers, self, rank, world, watch)) return poison(out, nvec * world);
  if constexpr (W > 0) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * kThreads) {
      uint4 v[W];
#pragma unroll
      for (int r = 0; r < W; ++r) v[r] = reinterpret_cast<const uint4*>(in.ptr[r])[i];
#pragma unroll
      for (int r = 0; r < W; ++r) reinterpret_cast<uint4*>(out)[(int64_t)r * nvec + i] = v[r];
    }
  } else {
    for (int r = 0; r < world; ++r) {
      const int src = (rank + r) % world;
      const uint4* from = reinterpret_cast<const uint4*>(in.ptr[src]);
      uint4* to = reinterpret_cast<uint4*>(out) + (int64_t)src * nvec;
      for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * kThreads)
        to[i] = from[i];
    }
  }
  if (!barrier_all<false>(peers, self, rank, world, watch)) poison(out, nvec * world);
}

using IpcKey = std::array<uint8_t, sizeof(hipIpcMemHandle_t)>;

// Blocks per launch.  Every block spins in the barriers until all ranks have arrived, so a launch must stay a small
// part of the chip: with the group size known at compile time each thread keeps one 16-byte load per rank in flight,
// and 16 blocks of 512 threads hold 1 MiB in flight, more than seven links take.  SGLK_AR_BLOCKS (1..64) overrides.
int block_limit() {
  static const int limit = [] {
    const char* env = getenv("SGLK_AR_BLOCKS");
    const int v = env ? atoi(env) : kDefaultBlocks;
    return v >= 1 && v <= kMaxBlocks ? v : kDefaultBlocks;
  }();
  return limit;
}

struct Comm {
  int rank = 0, world = 1, full_xgmi = 1;
  int64_t staging_bytes = 0;
  Peers peers{};
  Signal* self = nullptr;
  // Records of peer pointers in device memory: the caller's rank_data buffer (init_custom_ar's second argument,
  // custom_all_reduce.py:289-294) or, for the pointer form of init, an allocation of this communicator.
  Inputs* slot_next = nullptr;
  Inputs* slot_end = nullptr;
  void* own_slots = nullptr;
  std::map<const void*, Inputs*> buffers;   // registered input -> its record
  std::vector<void*> graph_unreg;           // inputs seen under capture, in launch order; records reserved
  std::vector<Inputs*> graph_slots;
  std::map<IpcKey, char*> ipc;              // handles this communicator opened (closed in dispose)
  std::mutex mu;
  uint32_t* host_flag = nullptr;  // pinned; the kernels write it through watch.host_flag
  Watch watch{};
  ~Comm() {
    if (host_flag) (void)hipHostFree(host_flag);
    for (auto& kv : ipc)
      if (kv.second) (void)hipIpcCloseMemHandle(kv.second);
    if (own_slots) (void)hipFree(own_slots);
  }
  // open_ipc_handle (custom_all_reduce_hip.cuh:402-411): one mapping per distinct handle
  char* open(const void* handle) {
    IpcKey key;
    std::memcpy(key.data(), handle, sizeof(hipIpcMemHandle_t));
    auto it = ipc.find(key);
    if (it != ipc.end()) return it->second;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return nullptr;
    ipc[key] = static_cast<char*>(p);
    return static_cast<char*>(p);
  }
  Inputs* take_slot() {
    if (!slot_next || slot_next >= slot_end) return nullptr;
    return slot_next++;
  }
};

constexpr int64_t kOwnSlots = 4096;  // records the pointer form of init allocates for itself

int check_alive(Comm* c, const char* what) {
  if (__atomic_load_n(c->host_flag, __ATOMIC_RELAXED) != 0)
    return fail(SGLK_ESTUCK,
                "%s: a peer did not reach a barrier within the time limit on this communicator; its outputs since "
                "then are NaN-filled and it takes no more work (make a new one on every rank)",
                what);
  return SGLK_OK;
}

}  // namespace
}  // namespace sglk

using namespace sglk;

extern "C" {

int64_t sglk_ar_meta_size(void) { return (int64_t)sizeof(Signal); }

int64_t sglk_ar_init(void* const* meta_ptrs, int rank, int world, int full_xgmi, int64_t staging_bytes) {
  if (!meta_ptrs || world < 1 || world > kMaxRanks || rank < 0 || rank >= world || staging_bytes < 0) {
    fail(SGLK_EINVAL, "ar_init: bad arguments (world=%d rank=%d; at most %d ranks)", world, rank, kMaxRanks);
    return 0;
  }
  for (int r = 0; r < world; ++r)
    if (!meta_ptrs[r]) {
      fail(SGLK_EINVAL, "ar_init: null signal block for rank %d", r);
      return 0;
    }
  Comm* c = new Comm();
  c->rank = rank;
  c->world = world;
  c->full_xgmi = full_xgmi;
  c->staging_bytes = staging_bytes;
  for (int r = 0; r < world; ++r) c->peers.sig[r] = reinterpret_cast<Signal*>(meta_ptrs[r]);
  c->self = c->peers.sig[rank];
  void* dev_view = nullptr;
  hipError_t e = hipHostMalloc((void**)&c->host_flag, sizeof(uint32_t), hipHostMallocMapped);
  if (e == hipSuccess) {
    *c->host_flag = 0;
    e = hipHostGetDevicePointer(&dev_view, c->host_flag, 0);
  }
  if (e != hipSuccess) {
    fail(SGLK_EHIP, "ar_init: pinned status word: %s", hipGetErrorString(e));
    delete c;
    return 0;
  }
  c->watch.host_flag = static_cast<uint32_t*>(dev_view);
  int64_t ms = kDefaultTimeoutMs;
  if (const char* env = getenv("SGLK_AR_TIMEOUT_MS")) {
    const long long v = atoll(env);
    if (v > 0) ms = v;
  }
  c->watch.ticks = (uint64_t)ms * kTicksPerMs;
  if (hipMalloc(&c->own_slots, sizeof(Inputs) * kOwnSlots) != hipSuccess) {
    fail(SGLK_EHIP, "ar_init: record area: %s", hipGetErrorString(hipGetLastError()));
    c->own_slots = nullptr;
    delete c;
    return 0;
  }
  c->slot_next = static_cast<Inputs*>(c->own_slots);
  c->slot_end = c->slot_next + kOwnSlots;
  return (int64_t) reinterpret_cast<intptr_t>(c);
}

int64_t sglk_ar_init_handles(void* meta, void* rank_data, int64_t rank_data_bytes, const void* const* handles,
                             const int64_t* offsets, int rank, int world, int full_xgmi, int64_t staging_bytes) {
  if (!meta || !handles || !offsets || world < 1 || world > kMaxRanks || rank < 0 || rank >= world) {
    fail(SGLK_EINVAL, "ar_init_handles: bad arguments (world=%d rank=%d; at most %d ranks)", world, rank, kMaxRanks);
    return 0;
  }
  if (rank_data && rank_data_bytes < (int64_t)sizeof(Inputs)) {
    fail(SGLK_EINVAL, "ar_init_handles: rank_data holds no record (%lld bytes)", (long long)rank_data_bytes);
    return 0;
  }
  // the own block first, so that a Comm exists to keep the mappings opened for the peers
  void* metas[kMaxRanks];
  for (int r = 0; r < world; ++r) metas[r] = meta;
  const int64_t fa = sglk_ar_init(metas, rank, world, full_xgmi, staging_bytes);
  if (!fa) return 0;
  Comm* c = reinterpret_cast<Comm*>((intptr_t)fa);
  for (int r = 0; r < world; ++r) {
    if (r == rank) continue;
    char* base = handles[r] ? c->open(handles[r]) : nullptr;
    if (!base) {
      fail(SGLK_EHIP, "ar_init_handles: cannot map the signal block of rank %d", r);
      delete c;
      return 0;
    }
    c->peers.sig[r] = reinterpret_cast<Signal*>(base + offsets[r]);
  }
  if (rank_data) {  // the caller's buffer takes the place of the own record area
    (void)hipFree(c->own_slots);
    c->own_slots = nullptr;
    c->slot_next = static_cast<Inputs*>(rank_data);
    c->slot_end = c->slot_next + rank_data_bytes / (int64_t)sizeof(Inputs);
  }
  return fa;
}

int sglk_ar_set_timeout_ms(int64_t fa, int64_t ms) {
  Comm* c = reinterpret_cast<Comm*>((intptr_t)fa);
  SGLK_CHECK(c && ms > 0, "ar_set_timeout_ms: null handle or non-positive time");
  c->watch.ticks = (uint64_t)ms * kTicksPerMs;
  return SGLK_OK;
}

int sglk_ar_dispose(int64_t fa) {
  delete reinterpret_cast<Comm*>((intptr_t)fa);
  return SGLK_OK;
}

int sglk_ar_register_buffer(int64_t fa, void* local, void* const* data_ptrs) {
  Comm* c = reinterpret_cast<Comm*>((intptr_t)fa);
  SGLK_CHECK(c && local && data_ptrs, "ar_register_buffer: null argument");
  Inputs in{};
  for (int r = 0; r < c->world; ++r) {
    SGLK_CHECK(data_ptrs[r], "ar_register_buffer: null mapping for rank %d", r);
    in.ptr[r] = data_ptrs[r];
  }
  SGLK_CHECK(data_ptrs[c->rank] == local, "ar_register_buffer: data_ptrs[rank] must be the local buffer");
  std::lock_guard<std::mutex> g(c->mu);
  auto it = c->buffers.find(local);
  Inputs* slot = it != c->buffers.end() ? it->second : c->take_slot();
  SGLK_CHECK(slot, "ar_register_buffer: the record area is full (custom_all_reduce_hip.cuh:436-440)");
  SGLK_HIP(hipMemcpy(slot, &in, sizeof(Inputs), hipMemcpyHostToDevice));
  c->buffers[local] = slot;
  return SGLK_OK;
}

int sglk_ar_register_buffer_handles(int64_t fa, void* local, const void* const* handles, const int64_t* offsets) {
  Comm* c = reinterpret_cast<Comm*>((intptr_t)fa);
  SGLK_CHECK(c && local && handles && offsets, "ar_register_buffer_handles: null argument");
  void* ptrs[kMaxRanks];
  {
    std::lock_guard<std::mutex> g(c->mu);
    for (int r = 0; r < c->world; ++r) {
      if (r == c->rank) {
        ptrs[r] = local;
        continue;
      }
      char* base = handles[r] ? c->open(handles[r]) : nullptr;
      SGLK_CHECK(base, "ar_register_buffer_handles: cannot map the buffer of rank %d", r);
      ptrs[r] = base + offsets[r];
    }
  }
  return sglk_ar_register_buffer(fa, local, ptrs);
}

int64_t sglk_ar_graph_buffer_count(int64_t fa) {
  Comm* c = reinterpret_cast<Comm*>((intptr_t)fa);
  if (!c) return fail(SGLK_EINVAL, "ar_graph_buffer_count: null handle");
  std::lock_guard<std::mutex> g(c->mu);
  return (int64_t)c->graph_unreg.size();
}

int sglk_ar_get_graph_buffer_ipc_meta(int64_t fa, void* handles_out, int64_t* offsets_out, int64_t cap) {
  Comm* c = reinterpret_cast<Comm*>((intptr_t)fa);
  SGLK_CHECK(c, "ar_get_graph_buffer_ipc_meta: null handle");
  std::lock_guard<std::mutex> g(c->mu);
  const int64_t n = (int64_t)c->graph_unreg.size();
  SGLK_CHECK(n <= cap && (n == 0 || (handles_out && offsets_out)), "ar_get_graph_buffer_ipc_meta: room for %lld records needed",
             (long long)n);
  for (int64_t i = 0; i < n; ++i) {
    // the handle names the whole allocation: share its base and the pointer's offset in it
    // (custom_all_reduce_hip.cuh:413-434)
    void* base = nullptr;
    size_t size = 0;
    SGLK_HIP(hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)c->graph_unreg[(size_t)i]));
    hipIpcMemHandle_t h;
    SGLK_HIP(hipIpcGetMemHandle(&h, base));
    std::memcpy(static_cast<char*>(handles_out) + i * (int64_t)sizeof(h), &h, sizeof(h));
    offsets_out[i] = (int64_t)((char*)c->graph_unreg[(size_t)i] - (char*)base);
  }
  return SGLK_OK;
}

int sglk_ar_register_graph_buffers(int64_t fa, const void* const* handles, const int64_t* const* offsets, int64_t count) {
  Comm* c = reinterpret_cast<Comm*>((intptr_t)fa);
  SGLK_CHECK(c && (count == 0 || (handles && offsets)), "ar_register_graph_buffers: null argument");
  std::lock_guard<std::mutex> g(c->mu);
  SGLK_CHECK(count == (int64_t)c->graph_unreg.size(),
             "ar_register_graph_buffers: %lld records given, %lld inputs were captured", (long long)count,
             (long long)c->graph_unreg.size());
  // addresses are not deduplicated, as in the reference (:453-476): every captured launch owns its record
  for (int64_t i = 0; i < count; ++i) {
    Inputs in{};
    for (int r = 0; r < c->world; ++r) {
      if (r == c->rank) {
        in.ptr[r] = c->graph_unreg[(size_t)i];
        continue;
      }
      SGLK_CHECK(handles[r] && offsets[r], "ar_register_graph_buffers: null list for rank %d", r);
      char* base = c->open(static_cast<const char*>(handles[r]) + i * (int64_t)sizeof(hipIpcMemHandle_t));
      SGLK_CHECK(base, "ar_register_graph_buffers: cannot map input %lld of rank %d", (long long)i, r);
      in.ptr[r] = base + offsets[r][i];
    }
    SGLK_HIP(hipMemcpy(c->graph_slots[(size_t)i], &in, sizeof(Inputs), hipMemcpyHostToDevice));
  }
  c->graph_unreg.clear();
  c->graph_slots.clear();
  return SGLK_OK;
}

int sglk_ar_all_reduce(int64_t fa, const void* inp, void* out, int64_t numel, int dtype, int algo, void* stream) {
  Comm* c = reinterpret_cast<Comm*>((intptr_t)fa);
  SGLK_CHECK(c && inp && out, "ar_all_reduce: null argument");
  SGLK_CHECK(numel >= 0, "ar_all_reduce: negative size");
  SGLK_CHECK(dtype == SGLK_F32 || dtype == SGLK_F16 || dtype == SGLK_BF16, "ar_all_reduce: unsupported dtype %d", dtype);
  if (numel == 0) return SGLK_OK;
  if (int rc = check_alive(c, "ar_all_reduce")) return rc;
  const int64_t bytes = numel * elem_size(dtype);
  // custom_all_reduce_hip.cuh:480-484: the payload must be a whole number of 16-byte packs
  SGLK_CHECK(bytes % 16 == 0, "ar_all_reduce: byte size %lld is not a multiple of 16", (long long)bytes);
  SGLK_CHECK(aligned16(inp) && aligned16(out), "ar_all_reduce: buffers must be 16-byte aligned");
  const Inputs* in = nullptr;
  {
    std::lock_guard<std::mutex> g(c->mu);
    auto it = c->buffers.find(inp);
    if (it != c->buffers.end()) {
      in = it->second;
    } else {
      // custom_all_reduce_hip.cuh:517-533: under capture an unknown input gets a record that
      // register_graph_buffers fills in before the graph is replayed; outside capture it is an error
      hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
      SGLK_HIP(hipStreamIsCapturing((hipStream_t)stream, &st));
      SGLK_CHECK(st == hipStreamCaptureStatusActive,
                 "ar_all_reduce: input %p is not a registered buffer (and the stream is not capturing)", inp);
      Inputs* slot = c->take_slot();
      SGLK_CHECK(slot, "ar_all_reduce: the record area is full (custom_all_reduce_hip.cuh:436-440)");
      c->graph_unreg.push_back(const_cast<void*>(inp));
      c->graph_slots.push_back(slot);
      in = slot;
    }
  }
  const int64_t nvec = bytes / 16;
  if (algo == 0) {
    // custom_all_reduce_hip.cuh:543-553
    if (c->world <= 2)
      algo = 1;
    else if (c->full_xgmi && ((c->world <= 4 && bytes < 512 * 1024) || bytes < 256 * 1024))
      algo = 1;
    else
      algo = 2;
  }
  SGLK_CHECK(algo == 1 || algo == 2, "ar_all_reduce: algo must be 0 (auto), 1 or 2");
  if (algo == 2) {
    const int64_t need = (nvec / c->world + nvec % c->world) * 16;
    SGLK_CHECK(need <= c->staging_bytes, "ar_all_reduce: two-stage needs %lld staging bytes, %lld were given at init",
               (long long)need, (long long)c->staging_bytes);
  }
  const int64_t work = algo == 2 ? (nvec + c->world - 1) / c->world : nvec;
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(block_limit(), (work + kThreads - 1) / kThreads));
  hipStream_t s = (hipStream_t)stream;
#define SGLK_AR_LAUNCH(W)                                                                                              \
  do {                                                                                                                 \
    if (algo == 1)                                                                                                     \
      hipLaunchKernelGGL((allreduce_one_stage<T, W>), dim3(blocks), dim3(kThreads), 0, s, in, c->peers, c->self, out,  \
                         c->rank, c->world, nvec, c->watch);                                                           \
    else                                                                                                               \
      hipLaunchKernelGGL((allreduce_two_stage<T, W>), dim3(blocks), dim3(kThreads), 0, s, in, c->peers, c->self, out,  \
                         c->rank, c->world, nvec, c->watch);                                                           \
  } while (0)
  // the group sizes the reference dispatches on (custom_all_reduce_hip.cuh:555-565) are compiled with the size known;
  // any other size runs the same kernels with the loops over ranks at run time
  static const bool generic = getenv("SGLK_AR_GENERIC") != nullptr;
  SGLK_DISPATCH_FLOAT(dtype, T, {
    switch (generic ? 0 : c->world) {
      case 2: SGLK_AR_LAUNCH(2); break;
      case 4: SGLK_AR_LAUNCH(4); break;
      case 6: SGLK_AR_LAUNCH(6); break;
      case 8: SGLK_AR_LAUNCH(8); break;
      default: SGLK_AR_LAUNCH(0); break;
    }
  });
#undef SGLK_AR_LAUNCH
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_ar_all_gather(int64_t fa, const void* inp, void* out, int64_t bytes_per_rank, void* stream) {
  Comm* c = reinterpret_cast<Comm*>((intptr_t)fa);
  SGLK_CHECK(c && inp && out, "ar_all_gather: null argument");
  SGLK_CHECK(bytes_per_rank >= 0, "ar_all_gather: negative size");
  if (bytes_per_rank == 0) return SGLK_OK;
  if (int rc = check_alive(c, "ar_all_gather")) return rc;
  SGLK_CHECK(bytes_per_rank % 16 == 0, "ar_all_gather: byte size %lld is not a multiple of 16", (long long)bytes_per_rank);
  SGLK_CHECK(aligned16(inp) && aligned16(out), "ar_all_gather: buffers must be 16-byte aligned");
  const Inputs* in = nullptr;
  {
    std::lock_guard<std::mutex> g(c->mu);
    auto it = c->buffers.find(inp);
    SGLK_CHECK(it != c->buffers.end(), "ar_all_gather: input %p is not a registered buffer", inp);
    in = it->second;
  }
  const int64_t nvec = bytes_per_rank / 16;
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(block_limit(), (nvec + kThreads - 1) / kThreads));
#define SGLK_AG_LAUNCH(W)                                                                                          \
  hipLaunchKernelGGL(allgather_kernel<W>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, in, c->peers, c->self, \
                     out, c->rank, c->world, nvec, c->watch)
  static const bool generic = getenv("SGLK_AR_GENERIC") != nullptr;
  switch (generic ? 0 : c->world) {
    case 2: SGLK_AG_LAUNCH(2); break;
    case 4: SGLK_AG_LAUNCH(4); break;
    case 6: SGLK_AG_LAUNCH(6); break;
    case 8: SGLK_AG_LAUNCH(8); break;
    default: SGLK_AG_LAUNCH(0); break;
  }
#undef SGLK_AG_LAUNCH
  SGLK_LAUNCH_CHECK();
  return SGLK_OK;
}

int sglk_ar_all_reduce_unreg(int64_t fa, const void* inp, void* reg_buffer, void* out, int64_t numel, int dtype, int algo,
                             void* stream) {
  SGLK_CHECK(inp && reg_buffer, "ar_all_reduce_unreg: null argument");
  SGLK_CHECK(numel >= 0, "ar_all_reduce_unreg: negative size");
  SGLK_HIP(hipMemcpyAsync(reg_buffer, inp, (size_t)numel * elem_size(dtype), hipMemcpyDeviceToDevice,
                          (hipStream_t)stream));
  return sglk_ar_all_reduce(fa, reg_buffer, out, numel, dtype, algo, stream);
}

int sglk_ar_stuck(int64_t fa) {
  Comm* c = reinterpret_cast<Comm*>((intptr_t)fa);
  if (!c) return fail(SGLK_EINVAL, "ar_stuck: null handle");
  return __atomic_load_n(c->host_flag, __ATOMIC_RELAXED) != 0 ? 1 : 0;
}

// ---- IPC plumbing --------------------------------------------------------------------------------------
int sglk_ipc_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

int sglk_ipc_get_handle(void* dev_ptr, void* handle_out, int64_t* offset_out) {
  SGLK_CHECK(dev_ptr && handle_out, "ipc_get_handle: null argument");
  void* base = nullptr;
  size_t size = 0;
  SGLK_HIP(hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)dev_ptr));
  hipIpcMemHandle_t h;
  SGLK_HIP(hipIpcGetMemHandle(&h, base));
  std::memcpy(handle_out, &h, sizeof(h));
  if (offset_out) *offset_out = (int64_t)((char*)dev_ptr - (char*)base);
  return SGLK_OK;
}

void* sglk_ipc_open_handle(const void* handle) {
  if (!handle) {
    fail(SGLK_EINVAL, "ipc_open_handle: null handle");
    return nullptr;
  }
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, sizeof(h));
  void* p = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    fail(SGLK_EHIP, "hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
    return nullptr;
  }
  return p;
}

int sglk_ipc_close(void* mapped) {
  SGLK_CHECK(mapped, "ipc_close: null pointer");
  SGLK_HIP(hipIpcCloseMemHandle(mapped));
  return SGLK_OK;
}

void* sglk_alloc_uncached(int64_t bytes) {
  if (bytes <= 0) {
    fail(SGLK_EINVAL, "alloc_uncached: bad size");
    return nullptr;
  }
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    fail(SGLK_EHIP, "hipExtMallocWithFlags(uncached) failed: %s", hipGetErrorString(e));
    return nullptr;
  }
  e = hipMemset(p, 0, (size_t)bytes);
  if (e != hipSuccess) {
    fail(SGLK_EHIP, "hipMemset failed: %s", hipGetErrorString(e));
    (void)hipFree(p);
    return nullptr;
  }
  return p;
}

int sglk_free_uncached(void* p) {
  if (p) SGLK_HIP(hipFree(p));
  return SGLK_OK;
}

}  // extern "C"
