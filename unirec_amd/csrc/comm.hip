// ur_comm_*: the gradient all-reduce of the data-parallel step behind the C ABI.
//
// The reference trains on one device (training/train_item_individual_token_joint.py:755-773 hands the model to the HF
// Trainer; no torch.distributed call anywhere), so there is no reference line to match: this is the north-star's pure data
// parallelism -- one process per GPU, a SUM all-reduce per gradient bucket over RCCL / xGMI, overlapped with the backward.
//
// Ownership and ordering (no host synchronisation anywhere):
//   * the library owns the RCCL communicator, ONE side stream, a `ready` event and a ring of completion events; the caller owns every buffer;
//   * ur_comm_allreduce_async(buf, producer_stream): records `ready` on the producer stream (the stream whose kernels wrote
//     buf), makes the side stream wait for it, and queues an in-place sum all-reduce on the side stream.  Buckets queue in call
//     order on that one stream -- the same order on every rank, which is what RCCL requires;
//   * ur_comm_wait(consumer_stream): the consumer stream (the optimizer's) waits for everything queued so far;
//     ur_comm_wait_ticket(ticket, consumer_stream) for ONE bucket and everything before it (every all-reduce records its own event of
//     a ring), so a consumer can start on the first buckets while the last are still on the wire;
//   * every call runs on the communicator's device and restores the caller's current device.
//   The buffer must stay allocated and untouched by other streams until a ur_comm_wait has been issued.
//
// RCCL is resolved at run time (dlopen of librccl.so.1, preferring the copy the process has already loaded -- PyTorch-ROCm
// ships one): the library itself loads, and every other entry point works, on a box without RCCL; ur_comm_* then fail with a
// message instead of the loader failing.
#include "common.hip.h"
#include "unirec_hip.h"

#include <algorithm>
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char why[256] = {0};
};

RcclApi g_api;
std::once_flag g_api_once;

void load_api() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {          // the copy this process already holds, if any (one RCCL per process)
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (h) break;
  }
  for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    snprintf(g_api.why, sizeof(g_api.why), "librccl.so.1 not found (%s)", dlerror());
    return;
  }
  g_api.GetUniqueId = reinterpret_cast<decltype(g_api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  g_api.CommInitRank = reinterpret_cast<decltype(g_api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  g_api.CommDestroy = reinterpret_cast<decltype(g_api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  g_api.AllReduce = reinterpret_cast<decltype(g_api.AllReduce)>(dlsym(h, "ncclAllReduce"));
  g_api.GetErrorString = reinterpret_cast<decltype(g_api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!g_api.GetUniqueId || !g_api.CommInitRank || !g_api.CommDestroy || !g_api.AllReduce || !g_api.GetErrorString) {
    snprintf(g_api.why, sizeof(g_api.why), "librccl.so.1 lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce");
    return;
  }
  g_api.handle = h;
}

const RcclApi* api() {
  std::call_once(g_api_once, load_api);
  return g_api.handle ? &g_api : nullptr;
}

constexpr uint32_t COMM_MAGIC = 0x55524343u;   // "URCC"

struct Comm {
  uint32_t magic = COMM_MAGIC;
  ncclComm_t comm = nullptr;
  hipStream_t side = nullptr;
  hipEvent_t ready = nullptr, done[UR_COMM_RING] = {};      // done[(ticket - 1) % UR_COMM_RING]: completion of the all-reduce with that ticket
  int rank = 0, world = 1, device = 0;
  long long queued = 0;
};

// the communicator's device for the duration of a call (a caller thread may have another one current)
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  explicit DeviceScope(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};

Comm* as_comm(void* p) {
  Comm* c = static_cast<Comm*>(p);
  return (c && c->magic == COMM_MAGIC) ? c : nullptr;
}

}  // namespace

#define UR_HIP_OK(call, what)                                                                     \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) UR_FAIL((int)e_, "%s: %s failed: %s", fn, what, hipGetErrorString(e_)); \
  } while (0)
#define UR_RCCL_OK(call, what)                                                                                  \
  do {                                                                                                          \
    ncclResult_t r_ = (call);                                                                                   \
    if (r_ != ncclSuccess) UR_FAIL(1000 + (int)r_, "%s: %s failed: %s", fn, what, a->GetErrorString(r_));       \
  } while (0)

extern "C" int ur_comm_unique_id(void* id_out) {
  static const char* fn = "ur_comm_unique_id";
  UR_REQUIRE(id_out != nullptr, "%s: id_out is null", fn);
  static_assert(sizeof(ncclUniqueId) == UR_COMM_ID_BYTES, "RCCL unique id size");
  const RcclApi* a = api();
  if (!a) UR_FAIL(-2, "%s: %s", fn, g_api.why);
  ncclUniqueId id;
  UR_RCCL_OK(a->GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

extern "C" int ur_comm_init(void** comm_out, int32_t rank, int32_t world, const void* unique_id, int32_t device) {
  static const char* fn = "ur_comm_init";
  UR_REQUIRE(comm_out != nullptr && unique_id != nullptr, "%s: null argument", fn);
  *comm_out = nullptr;
  UR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "%s: rank %d outside a world of %d", fn, rank, world);
  UR_REQUIRE(device >= 0, "%s: device %d", fn, device);
  const RcclApi* a = api();
  if (!a) UR_FAIL(-2, "%s: %s", fn, g_api.why);
  DeviceScope scope(device);
  {
    int cur = -1;
    UR_HIP_OK(hipGetDevice(&cur), "hipGetDevice");
    UR_REQUIRE(cur == device, "%s: cannot select device %d", fn, device);
  }
  Comm* c = new Comm();
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclResult_t r = a->CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    UR_FAIL(1000 + (int)r, "%s: ncclCommInitRank(rank %d of %d) failed: %s", fn, rank, world, a->GetErrorString(r));
  }
  hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
  for (int i = 0; e == hipSuccess && i < UR_COMM_RING; ++i) e = hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming);
  if (e != hipSuccess) {
    for (int i = 0; i < UR_COMM_RING; ++i) if (c->done[i]) (void)hipEventDestroy(c->done[i]);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->side) (void)hipStreamDestroy(c->side);
    a->CommDestroy(c->comm);
    delete c;
    UR_FAIL((int)e, "%s: side stream / events: %s", fn, hipGetErrorString(e));
  }
  *comm_out = c;
  return 0;
}

extern "C" int ur_comm_allreduce_async(void* comm, void* buf, int64_t count, int32_t dtype, void* producer_stream) {
  static const char* fn = "ur_comm_allreduce_async";
  Comm* c = as_comm(comm);
  UR_REQUIRE(c != nullptr, "%s: not a communicator from ur_comm_init", fn);
  UR_REQUIRE(count >= 0, "%s: count %lld", fn, (long long)count);
  UR_REQUIRE(dtype == UR_COMM_F32 || dtype == UR_COMM_BF16, "%s: dtype %d (UR_COMM_F32 = 0, UR_COMM_BF16 = 1)", fn, dtype);
  if (count == 0) return 0;
  UR_REQUIRE(buf != nullptr, "%s: buf is null", fn);
  UR_REQUIRE((((uintptr_t)buf) & (dtype == UR_COMM_F32 ? 3 : 1)) == 0, "%s: buf is not aligned to its element size", fn);
  const RcclApi* a = api();
  if (!a) UR_FAIL(-2, "%s: %s", fn, g_api.why);
  DeviceScope scope(c->device);
  hipStream_t prod = static_cast<hipStream_t>(producer_stream);
  UR_HIP_OK(hipEventRecord(c->ready, prod), "hipEventRecord(ready)");
  UR_HIP_OK(hipStreamWaitEvent(c->side, c->ready, 0), "hipStreamWaitEvent(side)");
  UR_RCCL_OK(a->AllReduce(buf, buf, (size_t)count, dtype == UR_COMM_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, c->comm, c->side), "ncclAllReduce");
  UR_HIP_OK(hipEventRecord(c->done[c->queued % UR_COMM_RING], c->side), "hipEventRecord(done)");
  c->queued += 1;
  return 0;
}

extern "C" int64_t ur_comm_ticket(void* comm) {
  Comm* c = as_comm(comm);
  return c ? (int64_t)c->queued : -1;
}

extern "C" int ur_comm_wait_ticket(void* comm, int64_t ticket, void* consumer_stream) {
  static const char* fn = "ur_comm_wait_ticket";
  Comm* c = as_comm(comm);
  UR_REQUIRE(c != nullptr, "%s: not a communicator from ur_comm_init", fn);
  UR_REQUIRE(ticket >= 0 && ticket <= c->queued, "%s: ticket %lld outside 0..%lld", fn, (long long)ticket, c->queued);
  if (ticket == 0) return 0;
  // an event that has left the ring was re-recorded by a LATER all-reduce of the same in-order stream: waiting for the oldest one kept covers it
  const long long t = std::max<long long>(ticket, c->queued - UR_COMM_RING + 1);
  DeviceScope scope(c->device);
  UR_HIP_OK(hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), c->done[(t - 1) % UR_COMM_RING], 0), "hipStreamWaitEvent(consumer)");
  return 0;
}

extern "C" int ur_comm_wait(void* comm, void* consumer_stream) {
  Comm* c = as_comm(comm);
  if (c == nullptr) UR_FAIL(-1, "ur_comm_wait: not a communicator from ur_comm_init");
  return ur_comm_wait_ticket(comm, (int64_t)c->queued, consumer_stream);
}

extern "C" int ur_comm_destroy(void* comm) {
  static const char* fn = "ur_comm_destroy";
  if (comm == nullptr) return 0;
  Comm* c = as_comm(comm);
  UR_REQUIRE(c != nullptr, "%s: not a communicator from ur_comm_init", fn);
  const RcclApi* a = api();
  DeviceScope scope(c->device);
  hipError_t e = hipStreamSynchronize(c->side);          // tear-down only: nothing of ours may still be queued
  if (a && c->comm) a->CommDestroy(c->comm);
  (void)hipEventDestroy(c->ready);
  for (int i = 0; i < UR_COMM_RING; ++i) (void)hipEventDestroy(c->done[i]);
  (void)hipStreamDestroy(c->side);
  c->magic = 0;
  delete c;
  if (e != hipSuccess) UR_FAIL((int)e, "%s: side stream: %s", fn, hipGetErrorString(e));
  return 0;
}
