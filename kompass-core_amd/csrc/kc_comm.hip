// Multi-GPU exchange of the sampling controller inside libkompass_hip.so: one
// process per GPU, each context scores its shard of the sample list, ONE
// ncclAllReduce(1 x int64, ncclMin) over RCCL / xGMI merges the packed
// (cost, global index) keys (LowestCost::combine, datatypes/
// trajectory.h:621-644).  RCCL is opened with dlopen on first use: a single-GPU
// user neither links nor loads it.  The caller moves the 128-byte unique id
// between its processes (any transport: MPI, a file, torch.distributed ...).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "kc_internal.h"

using namespace kc;

namespace {

struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  std::string why;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.h) break;
    }
    if (!r.h) {
      const char *e = dlerror();
      r.why = std::string("librccl.so not found: ") + (e ? e : "");
      return;
    }
    auto sym = [&](const char *n) { return dlsym(r.h, n); };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString;
    if (!r.ok) r.why = "librccl.so lacks a required symbol";
  });
  return r;
}

}  // namespace

struct kc_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

#define KC_NCCL(expr)                                                              \
  do {                                                                             \
    ncclResult_t _r = (expr);                                                      \
    if (_r != ncclSuccess) {                                                       \
      ::kc::set_error("%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(_r), __FILE__, __LINE__); \
      return KC_ERR_HIP;                                                           \
    }                                                                              \
  } while (0)

namespace kc {
// used by kc_dwa.hip: in-place all-reduce of `count` int64 at `dev` on `stream`
int comm_allreduce_i64(kc_comm *m, long long *dev, size_t count, bool sum, hipStream_t stream) {
  if (!m || !m->comm) KC_FAIL(KC_ERR_INVALID, "null communicator");
  KC_NCCL(rccl().AllReduce(dev, dev, count, ncclInt64, sum ? ncclSum : ncclMin, m->comm, stream));
  return KC_OK;
}
int comm_world(const kc_comm *m) { return m ? m->world : 1; }
int comm_device(const kc_comm *m) { return m ? m->device : -1; }
}  // namespace kc

extern "C" {

int kc_comm_unique_id(uint8_t id_out[KC_COMM_ID_BYTES]) {
  if (!id_out) KC_FAIL(KC_ERR_INVALID, "null argument");
  static_assert(KC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  if (!rccl().ok) KC_FAIL(KC_ERR_HIP, "RCCL unavailable: %s", rccl().why.c_str());
  ncclUniqueId id;
  KC_NCCL(rccl().GetUniqueId(&id));
  std::memcpy(id_out, id.internal, KC_COMM_ID_BYTES);
  return KC_OK;
}

int kc_comm_create(int rank, int world, const uint8_t id_in[KC_COMM_ID_BYTES], int device, kc_comm **out) {
  if (!id_in || !out) KC_FAIL(KC_ERR_INVALID, "null argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) KC_FAIL(KC_ERR_RANGE, "rank %d outside world %d", rank, world);
  if (!rccl().ok) KC_FAIL(KC_ERR_HIP, "RCCL unavailable: %s", rccl().why.c_str());
  KC_HIP(hipSetDevice(device));
  ncclUniqueId id;
  std::memcpy(id.internal, id_in, KC_COMM_ID_BYTES);
  auto *m = new kc_comm();
  m->rank = rank;
  m->world = world;
  m->device = device;
  ncclResult_t r = rccl().CommInitRank(&m->comm, world, id, rank);
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank failed: %s", rccl().GetErrorString(r));
    delete m;
    return KC_ERR_HIP;
  }
  *out = m;
  return KC_OK;
}

void kc_comm_destroy(kc_comm *m) {
  if (!m) return;
  if (m->comm && rccl().ok) {
    hipError_t e = hipSetDevice(m->device);
    (void)e;
    (void)rccl().CommDestroy(m->comm);
  }
  delete m;
}

int kc_comm_rank(const kc_comm *m) { return m ? m->rank : -1; }
int kc_comm_world(const kc_comm *m) { return m ? m->world : 0; }

}  // extern "C"
