// comm.cpp — the gradient collective inside the C-ABI: RCCL (over xGMI) called directly, no torch.distributed on the data path.
//
// Replaces what torch.nn.parallel.DistributedDataParallel does under the reference's train.py:113-114 (process group from
// train.py:58-61): one broadcast of rank 0's parameters / buffers at construction, and a bucketed MEAN all-reduce of the
// gradients overlapped with backward.  Here a communicator is attached to an executor context
// (mi355_resnet50_set_comm); mi355_resnet50_backward then issues, whenever the last segment of a bucket of the flat
// gradient array has been enqueued, ONE ncclAllReduce(ncclAvg) over the bucket's contiguous slice on a side HIP stream the
// communicator owns, chained by events behind the kernels (of both executor streams) that produce the bucket; the
// caller's stream waits for the last all-reduce at the end of the call.  No host thread, no host wait.
//
// RCCL is bound at run time (dlopen of librccl.so.1, preferring the copy already in the process — PyTorch-ROCm ships
// one with the same soname), so the library loads on single-GPU / GPU-less hosts without it.
#include <dlfcn.h>

#include <mutex>
#include <vector>

#include "common.h"
#include "comm.h"

namespace mi355 {
namespace {

// the slice of rccl.h this file needs (ABI-stable C interface; values from /opt/rocm/include/rccl/rccl.h)
typedef struct ncclComm* ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclFloat32 = 7 };
enum { ncclSum = 0, ncclAvg = 4 };

struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // the copy already loaded (torch's), if any
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.Broadcast = (decltype(r.Broadcast))dlsym(h, "ncclBroadcast");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Broadcast && r.GetErrorString) r.h = h;
  });
  return r.h ? &r : nullptr;
}

#define MI355_RCCL(expr)                                                                               \
  do {                                                                                                 \
    int rc_ = (expr);                                                                                  \
    if (rc_ != ncclSuccess) {                                                                          \
      ::mi355::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, rccl()->GetErrorString(rc_));   \
      return MI355_E_HIP;                                                                              \
    }                                                                                                  \
  } while (0)

}  // namespace
}  // namespace mi355

using namespace mi355;

struct mi355_comm {
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0, device = 0;
  hipStream_t stream = nullptr;  // the collective stream
  hipEvent_t done = nullptr;     // last all-reduce of a backward call
  hipEvent_t ev[2] = {nullptr, nullptr};
  // what has been issued through this communicator (mi355_comm_stats): kind 0 = bucket mean all-reduce of the flat gradient
  // array (begin / end in elements of it), 1 = broadcast, 2 = whole-buffer mean all-reduce (begin 0, end n)
  struct Op {
    int kind;
    size_t begin, end;
  };
  std::vector<Op> log;
  void note(int kind, size_t b, size_t e) {
    if (log.size() < 4096) log.push_back({kind, b, e});
  }
};

namespace mi355 {

// ---- used by resnet_exec.cpp ------------------------------------------------------------------------------------------
// the bucket [begin, end) of `grads` is complete once everything enqueued so far on `s` (and on `side`, if non-null) has
// run: chain ONE mean all-reduce behind both on the communicator's stream
int comm_allreduce_bucket(mi355_comm* cm, float* grads, size_t begin, size_t end, hipStream_t s, hipStream_t side) {
  Rccl* r = rccl();
  if (!r || !cm || !cm->comm) {
    set_error("allreduce_bucket: no communicator");
    return MI355_E_STATE;
  }
  MI355_HIP(hipEventRecord(cm->ev[0], s));
  MI355_HIP(hipStreamWaitEvent(cm->stream, cm->ev[0], 0));
  if (side) {
    MI355_HIP(hipEventRecord(cm->ev[1], side));
    MI355_HIP(hipStreamWaitEvent(cm->stream, cm->ev[1], 0));
  }
  MI355_RCCL(r->AllReduce(grads + begin, grads + begin, end - begin, ncclFloat32, ncclAvg, cm->comm, cm->stream));
  cm->note(0, begin, end);
  return 0;
}
// everything the communicator's stream has been given becomes visible to `s`
int comm_join(mi355_comm* cm, hipStream_t s) {
  MI355_HIP(hipEventRecord(cm->done, cm->stream));
  MI355_HIP(hipStreamWaitEvent(s, cm->done, 0));
  return 0;
}

}  // namespace mi355

// (see mi355_comm_standin) every wave polls the 100 MHz wall clock with s_sleep between reads: a CU slot held, no memory traffic
__global__ void standin_kernel(long long ticks) {
  extern __shared__ char standin_lds[];
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (ticks < 0) standin_lds[threadIdx.x] = 0;  // (keeps the LDS allocation)
}

extern "C" {

int mi355_comm_available(void) { return rccl() ? 1 : 0; }

int mi355_comm_unique_id(void* id_out) {
  MI355_ARG(id_out, "comm_unique_id: null pointer");
  Rccl* r = rccl();
  if (!r) {
    set_error("comm: librccl.so.1 not found");
    return MI355_E_STATE;
  }
  ncclUniqueId id;
  MI355_RCCL(r->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

int mi355_comm_create(mi355_comm** out, const void* id, int nranks, int rank, int device) {
  MI355_ARG(out && id && nranks >= 1 && rank >= 0 && rank < nranks && device >= 0, "comm_create: bad arguments");
  Rccl* r = rccl();
  if (!r) {
    set_error("comm: librccl.so.1 not found");
    return MI355_E_STATE;
  }
  MI355_HIP(hipSetDevice(device));
  mi355_comm* c = new mi355_comm();
  c->nranks = nranks; c->rank = rank; c->device = device;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  int rc = r->CommInitRank(&c->comm, nranks, uid, rank);
  if (rc != ncclSuccess) {
    set_error("comm_create: ncclCommInitRank -> %s", r->GetErrorString(rc));
    delete c;
    return MI355_E_HIP;
  }
  bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->done, hipEventDisableTiming) == hipSuccess;
  for (auto& e : c->ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    set_error("comm_create: stream / events -> %s", hipGetErrorString(hipGetLastError()));
    mi355_comm_destroy(c);
    return MI355_E_HIP;
  }
  *out = c;
  return 0;
}

int mi355_comm_destroy(mi355_comm* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
  for (auto e : c->ev)
    if (e) (void)hipEventDestroy(e);
  if (c->done) (void)hipEventDestroy(c->done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

int mi355_comm_nranks(const mi355_comm* c) { return c ? c->nranks : 0; }

int mi355_comm_broadcast(mi355_comm* c, float* buf, size_t n, int root, void* stream) {
  MI355_ARG(c && buf && root >= 0 && root < c->nranks, "comm_broadcast: bad arguments");
  if (n == 0) return 0;
  MI355_RCCL(rccl()->Broadcast(buf, buf, n, ncclFloat32, root, c->comm, (hipStream_t)stream));
  c->note(1, 0, n);
  return 0;
}

int mi355_comm_allreduce_mean(mi355_comm* c, float* buf, size_t n, void* stream) {
  MI355_ARG(c && buf, "comm_allreduce_mean: bad arguments");
  if (n == 0) return 0;
  MI355_RCCL(rccl()->AllReduce(buf, buf, n, ncclFloat32, ncclAvg, c->comm, (hipStream_t)stream));
  c->note(2, 0, n);
  return 0;
}

int mi355_comm_stats(mi355_comm* c, int reset, int cap, int* n_out, int* kinds, size_t* begins, size_t* ends) {
  MI355_ARG(c && n_out, "comm_stats: bad arguments");
  *n_out = (int)c->log.size();
  for (int i = 0; i < (int)c->log.size() && i < cap; ++i) {
    if (kinds) kinds[i] = c->log[i].kind;
    if (begins) begins[i] = c->log[i].begin;
    if (ends) ends[i] = c->log[i].end;
  }
  if (reset) c->log.clear();
  return 0;
}

// Measurement stand-in for a collective's CU footprint on ONE GPU (no curve can be measured without a node: this only shows what the
// executor's grids pay when `workgroups` CUs' worth of another library's persistent kernels sit beside them for `usec` microseconds —
// RCCL's ring kernels are one 256-thread workgroup per channel that spins on flags).  tools/reserve_cus_ab.py uses it.
int mi355_comm_standin(int workgroups, int usec, void* stream) {
  MI355_ARG(workgroups >= 1 && workgroups <= 256 && usec >= 1 && usec <= 100000, "comm_standin: workgroups=%d usec=%d", workgroups, usec);
  hipLaunchKernelGGL(standin_kernel, dim3(workgroups), dim3(256), 16 * 1024, (hipStream_t)stream, (long long)usec * 100);  // s_memrealtime: 100 MHz
  MI355_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
