// Gradient exchange of the data-parallel step behind the C ABI (lvae_allreduce_{init,enqueue,wait,destroy}): a private RCCL
// communicator, and the fork / join of a side stream around each bucket's ncclAllReduce, so that the exchange of a bucket overlaps the
// backward kernels still being issued on the launch stream (reference: none — the reference is single-process).
//
// Why not torch.distributed's collectives: ProcessGroupNCCL keeps a Work object, events and a polling watchdog per collective, which
// races with hipGraph capture (round 3: "operation not permitted on an event last recorded in a capturing stream"). A communicator of
// our own leaves nothing to poll: ncclAllReduce is a plain launch on the stream we pass and is captured like any other kernel.
//
// librccl is resolved at RUN TIME from the path the caller gives (the librccl.so torch itself loaded: "nccl" IS RCCL on ROCm), never at
// link time: liblvae_hip.so must load, and export every symbol, on a box without a GPU runtime behind it (tests/test_cabi.py).
#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <string>

#include "lvae_common.h"

namespace lvae {

// device copy of the out-of-place rehearsal form: a KERNEL, not hipMemcpyAsync — a memcpy NODE on a side branch of a captured graph was
// suspected of the 0.57 ms each bucket cost in the one-rank rehearsal (tools/forced_ab.sh); n % 4 == 0 or the tail is copied by scalars
__global__ __launch_bounds__(256) void allreduce_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[i];
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = src[(n4 << 2) + threadIdx.x];
}

struct RcclId {
  char internal[128];  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES)
};

struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

struct AllReduceHandle {
  RcclApi api;
  void* comm = nullptr;
  hipEvent_t fork_ev = nullptr, join_ev = nullptr;
  int world = 0, rank = 0;
};

// One handle per process: the library the caller names is first looked up among the objects ALREADY mapped (RTLD_NOLOAD — in a
// torch process that is the librccl ProcessGroupNCCL uses, so no second copy of RCCL is injected); only if it is not mapped yet is it
// loaded, with local symbol visibility. The handle is cached (and intentionally never closed: the communicators outlive any call).
static int rccl_open(const char* path, RcclApi& a) {
  LVAE_REQUIRE(path != nullptr, LVAE_EINVAL, "lvae_allreduce: null librccl path");
  static std::mutex mu;
  static void* cached = nullptr;
  static std::string cached_path;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (cached == nullptr || cached_path != path) {
      void* h = dlopen(path, RTLD_NOW | RTLD_NOLOAD);
      if (h == nullptr) h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
      LVAE_REQUIRE(h != nullptr, LVAE_EINVAL, "lvae_allreduce: dlopen(%s) failed: %s", path, dlerror());
      cached = h;
      cached_path = path;
    }
    a.lib = cached;
  }
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
  a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.lib, "ncclAllReduce"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
  LVAE_REQUIRE(a.GetUniqueId && a.CommInitRank && a.AllReduce && a.CommDestroy && a.GetErrorString, LVAE_EINVAL,
               "lvae_allreduce: %s does not export the ncclCommInitRank / ncclAllReduce family", path);
  return 0;
}

#define LVAE_RCCL_CHECK(api, rc, what)                                                      \
  do {                                                                                      \
    const int rc__ = (rc);                                                                  \
    if (rc__ != 0) {                                                                        \
      lvae::set_error("%s failed: %s", what, (api).GetErrorString(rc__));                   \
      return 1000 + rc__;                                                                   \
    }                                                                                       \
  } while (0)

#define LVAE_HIP_CHECK(expr, what)                                                          \
  do {                                                                                      \
    const hipError_t e__ = (expr);                                                          \
    if (e__ != hipSuccess) {                                                                \
      lvae::set_error("%s failed: %s", what, hipGetErrorString(e__));                       \
      return (int)e__;                                                                      \
    }                                                                                       \
  } while (0)

}  // namespace lvae

using namespace lvae;

extern "C" int lvae_allreduce_unique_id(const char* librccl_path, void* id128) {
  LVAE_REQUIRE(id128 != nullptr, LVAE_EINVAL, "lvae_allreduce_unique_id: null id buffer");
  RcclApi a;
  int rc = rccl_open(librccl_path, a);
  if (rc) return rc;
  RcclId id;
  LVAE_RCCL_CHECK(a, a.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int lvae_allreduce_init(const char* librccl_path, const void* id128, int32_t world, int32_t rank, void** handle) {
  LVAE_REQUIRE(id128 != nullptr && handle != nullptr && world >= 1 && rank >= 0 && rank < world, LVAE_EINVAL,
               "lvae_allreduce_init: bad id / handle / world / rank");
  AllReduceHandle* h = new AllReduceHandle();
  int rc = rccl_open(librccl_path, h->api);
  if (rc) {
    delete h;
    return rc;
  }
  RcclId id;
  memcpy(&id, id128, sizeof(id));
  const int nrc = h->api.CommInitRank(&h->comm, world, id, rank);
  if (nrc != 0) {
    set_error("ncclCommInitRank failed: %s", h->api.GetErrorString(nrc));
    delete h;
    return 1000 + nrc;
  }
  hipError_t e = hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->join_ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    set_error("lvae_allreduce_init: hipEventCreate failed: %s", hipGetErrorString(e));
    h->api.CommDestroy(h->comm);
    delete h;
    return (int)e;
  }
  h->world = world;
  h->rank = rank;
  *handle = h;
  return 0;
}

extern "C" int lvae_allreduce_enqueue(void* handle, float* buf, int64_t n, float* scratch, void* launch_stream, void* side_stream) {
  AllReduceHandle* h = static_cast<AllReduceHandle*>(handle);
  LVAE_REQUIRE(h != nullptr && h->comm != nullptr && buf != nullptr && n > 0, LVAE_EINVAL, "lvae_allreduce_enqueue: bad handle / buffer / count");
  hipStream_t ls = (hipStream_t)launch_stream, ss = (hipStream_t)side_stream;
  if (ls != ss) {  // fork: the side stream continues behind everything issued on the launch stream so far
    LVAE_HIP_CHECK(hipEventRecord(h->fork_ev, ls), "lvae_allreduce_enqueue: hipEventRecord");
    LVAE_HIP_CHECK(hipStreamWaitEvent(ss, h->fork_ev, 0), "lvae_allreduce_enqueue: hipStreamWaitEvent");
  }
  if (scratch != nullptr) {
    // out of place + copy back: with ONE rank an in-place all-reduce enqueues nothing at all, so a single-GPU rehearsal of the captured
    // exchange would capture an empty branch; this form always puts RCCL's kernel (its copy, for one rank) and a device copy on the stream
    LVAE_RCCL_CHECK(h->api, h->api.AllReduce(buf, scratch, (size_t)n, 7 /* ncclFloat32 */, 0 /* ncclSum */, h->comm, ss), "ncclAllReduce");
    if ((reinterpret_cast<uintptr_t>(buf) & 15) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0) {
      hipLaunchKernelGGL(allreduce_copy_kernel, dim3(grid_for(n >> 2, 256, 1024)), dim3(256), 0, ss, scratch, buf, n);
      LVAE_LAUNCH_CHECK("allreduce_copy");
    } else {
      LVAE_HIP_CHECK(hipMemcpyAsync(buf, scratch, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, ss), "lvae_allreduce_enqueue: hipMemcpyAsync");
    }
  } else {
    LVAE_RCCL_CHECK(h->api, h->api.AllReduce(buf, buf, (size_t)n, 7, 0, h->comm, ss), "ncclAllReduce");
  }
  return 0;
}

extern "C" int lvae_allreduce_wait(void* handle, void* launch_stream, void* side_stream) {
  AllReduceHandle* h = static_cast<AllReduceHandle*>(handle);
  LVAE_REQUIRE(h != nullptr, LVAE_EINVAL, "lvae_allreduce_wait: null handle");
  hipStream_t ls = (hipStream_t)launch_stream, ss = (hipStream_t)side_stream;
  if (ls == ss) return 0;
  LVAE_HIP_CHECK(hipEventRecord(h->join_ev, ss), "lvae_allreduce_wait: hipEventRecord");
  LVAE_HIP_CHECK(hipStreamWaitEvent(ls, h->join_ev, 0), "lvae_allreduce_wait: hipStreamWaitEvent");
  return 0;
}

extern "C" int lvae_allreduce_destroy(void* handle) {
  AllReduceHandle* h = static_cast<AllReduceHandle*>(handle);
  if (h == nullptr) return 0;
  if (h->comm) h->api.CommDestroy(h->comm);
  if (h->fork_ev) (void)hipEventDestroy(h->fork_ev);
  if (h->join_ev) (void)hipEventDestroy(h->join_ev);
  delete h;
  return 0;
}
