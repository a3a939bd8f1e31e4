// Multi-GPU exchange of the sampling controller inside libkompass_hip.so: one
// process per GPU, each context scores its shard of the sample list, ONE
// ncclAllReduce(int64 x (2 + world * words), ncclMin) over RCCL / xGMI merges the
// packed (cost, global index) keys (LowestCost::combine, datatypes/
// trajectory.h:621-644) and carries, in the same call, the error word and every
// rank's admissible bitmap (kc_shard.h: the exchange record).  RCCL is opened
// with dlopen on first use: a single-GPU user neither links nor loads it.  The
// caller moves the 128-byte unique id between its processes (any transport: MPI,
// a file, torch.distributed ...).
//
// Second transport, for rehearsals and tests on boxes with fewer GPUs than ranks
// (RCCL refuses two ranks on one device): kc_comm_create_shm -- the ranks are
// processes of one host that meet in a POSIX shared-memory segment; the same
// records are reduced by the host cores (D2H, min, H2D in stream order).  It
// exists so that every line of kc_dwa_cycle_sharded except the ncclAllReduce call
// itself runs with a world > 1 on a one-GPU box; it is not the multi-GPU product
// path and kc_comm_transport() says which one a communicator uses.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdlib>
#include <mutex>
#include <thread>

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
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;      // what the communicator itself reports
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;   // (kc_comm_query)
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
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
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
    r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(sym("ncclCommCuDevice"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString && r.CommCount &&
           r.CommUserRank && r.CommCuDevice;
    if (!r.ok) r.why = "librccl.so lacks a required symbol";
  });
  return r;
}

}  // namespace

// shared-memory transport: one segment per communicator
//   header | per rank: two slots (sequence parity) of {flag, count, words[kShmMaxWords]}
constexpr size_t kShmMaxWords = 16384;
struct ShmSlot {
  std::atomic<long long> flag;  // sequence number of the record in `words`
  long long count;
  long long words[kShmMaxWords];
};
struct ShmRank {
  ShmSlot slot[2];
};
struct ShmHeader {
  std::atomic<long long> magic;
  long long world;
  std::atomic<long long> attached, detached;
};
constexpr long long kShmMagic = 0x6b635f73686d3031ll;  // "kc_shm01"

struct kc_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  // shm transport
  bool shm = false;
  std::string shm_name;
  void *shm_base = nullptr;
  size_t shm_bytes = 0;
  long long shm_seq = 0;
  int shm_timeout_ms = 20000;
  long long *h_stage = nullptr;  // pinned, 2 x kShmMaxWords
};

#define KC_NCCL(expr)                                                              \
  do {                                                                             \
    ncclResult_t _r = (expr);                                                      \
    if (_r != ncclSuccess) {                                                       \
      ::kc::set_error("%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(_r), __FILE__, __LINE__); \
      return KC_ERR_HIP;                                                           \
    }                                                                              \
  } while (0)

namespace {
inline ShmHeader *shm_header(kc_comm *m) { return static_cast<ShmHeader *>(m->shm_base); }
inline ShmRank *shm_rank(kc_comm *m, int r) {
  return reinterpret_cast<ShmRank *>(static_cast<char *>(m->shm_base) + 4096) + r;
}

#ifndef KC_WITHOUT_REHEARSAL_TRANSPORT  // (make HIPFLAGS_EXTRA=-DKC_WITHOUT_REHEARSAL_TRANSPORT: a product build without it)
// host-side all-reduce over the segment: every rank publishes its record under the next sequence
// number, waits for the same number from every peer, reduces.  Two slots by sequence parity: a
// rank that is one exchange ahead writes the other slot; it cannot be two ahead, because the
// exchange in between needs this rank's record.
int shm_allreduce(kc_comm *m, const long long *send_dev, long long *recv_dev, size_t count, bool sum,
                  hipStream_t stream) {
  if (count > kShmMaxWords) KC_FAIL(KC_ERR_RANGE, "shm transport: %zu words exceed %zu", count, kShmMaxWords);
  long long *mine = m->h_stage, *out = m->h_stage + kShmMaxWords;
  KC_HIP(hipMemcpyAsync(mine, send_dev, count * sizeof(long long), hipMemcpyDeviceToHost, stream));
  KC_HIP(hipStreamSynchronize(stream));
  const long long seq = ++m->shm_seq;
  ShmSlot &s = shm_rank(m, m->rank)->slot[seq & 1];
  s.count = static_cast<long long>(count);
  std::memcpy(s.words, mine, count * sizeof(long long));
  s.flag.store(seq, std::memory_order_release);
  std::memcpy(out, mine, count * sizeof(long long));
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < m->world; ++r) {
    if (r == m->rank) continue;
    ShmSlot &p = shm_rank(m, r)->slot[seq & 1];
    for (long spins = 0; p.flag.load(std::memory_order_acquire) != seq; ++spins) {
      if ((spins & 255) == 255) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(m->shm_timeout_ms))
          KC_FAIL(KC_ERR_HIP, "shm transport: rank %d never arrived at exchange %lld (waited %d ms)", r, seq,
                  m->shm_timeout_ms);
        std::this_thread::yield();
      }
    }
    if (p.count != static_cast<long long>(count))
      KC_FAIL(KC_ERR_STATE, "shm transport: rank %d sent %lld words, this rank %zu", r, p.count, count);
    for (size_t i = 0; i < count; ++i) {
      const long long v = p.words[i];
      if (sum) out[i] += v;
      else if (v < out[i]) out[i] = v;
    }
  }
  KC_HIP(hipMemcpyAsync(recv_dev, out, count * sizeof(long long), hipMemcpyHostToDevice, stream));
  return KC_OK;
}
#else
int shm_allreduce(kc_comm *, const long long *, long long *, size_t, bool, hipStream_t) {
  KC_FAIL(KC_ERR_UNSUPPORTED, "this build has no rehearsal transport");
}
#endif
}  // namespace

namespace kc {
// used by kc_dwa.hip: all-reduce (min, or sum) of `count` int64 from `send` into `recv` (may be
// the same address) on `stream`
int comm_allreduce_i64(kc_comm *m, const long long *send, long long *recv, size_t count, bool sum,
                       hipStream_t stream) {
  if (!m) KC_FAIL(KC_ERR_INVALID, "null communicator");
  if (m->shm) return shm_allreduce(m, send, recv, count, sum, stream);
  if (!m->comm) KC_FAIL(KC_ERR_INVALID, "null communicator");
  KC_NCCL(rccl().AllReduce(send, recv, count, ncclInt64, sum ? ncclSum : ncclMin, m->comm, stream));
  return KC_OK;
}
int comm_world(const kc_comm *m) { return m ? m->world : 1; }
int comm_rank(const kc_comm *m) { return m ? m->rank : 0; }
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
  // the communicator's own view has to agree with the arguments (a stale unique id, a rank started twice)
  int cnt = -1, ur = -1, dev = -1;
  if (rccl().CommCount(m->comm, &cnt) != ncclSuccess || rccl().CommUserRank(m->comm, &ur) != ncclSuccess ||
      rccl().CommCuDevice(m->comm, &dev) != ncclSuccess || cnt != world || ur != rank || dev != device) {
    set_error("RCCL communicator reports %d ranks / rank %d / device %d, asked for %d / %d / %d", cnt, ur, dev, world, rank, device);
    (void)rccl().CommDestroy(m->comm);
    delete m;
    return KC_ERR_STATE;
  }
  *out = m;
  return KC_OK;
}

int kc_comm_create_shm(int rank, int world, const char *name, int device, kc_comm **out) {
#ifdef KC_WITHOUT_REHEARSAL_TRANSPORT
  (void)rank; (void)world; (void)name; (void)device;
  if (out) *out = nullptr;
  KC_FAIL(KC_ERR_UNSUPPORTED, "built with KC_WITHOUT_REHEARSAL_TRANSPORT: ranks exchange through RCCL only");
#else
  if (!name || !out) KC_FAIL(KC_ERR_INVALID, "null argument");
  *out = nullptr;
  if (world < 1 || world > 64 || rank < 0 || rank >= world)
    KC_FAIL(KC_ERR_RANGE, "rank %d outside world %d (shm transport: at most 64 ranks)", rank, world);
  KC_HIP(hipSetDevice(device));
  auto *m = new kc_comm();
  m->rank = rank;
  m->world = world;
  m->device = device;
  m->shm = true;
  m->shm_name = std::string("/kc_comm_") + name;
  if (const char *e = std::getenv("KC_SHM_TIMEOUT_MS")) m->shm_timeout_ms = std::max(100, std::atoi(e));
  m->shm_bytes = 4096 + sizeof(ShmRank) * static_cast<size_t>(world);
  // sys: a system call failed (errno says why); otherwise a peer did not show up in time.  Rank 0 takes the
  // name away on every failure path: a later communicator of the same name must not find this segment.
  auto fail = [&](const char *what, bool sys) {
    if (sys) set_error("shm transport: %s(%s) failed: %s", what, m->shm_name.c_str(), std::strerror(errno));
    else set_error("shm transport: %s on %s timed out after %d ms", what, m->shm_name.c_str(), m->shm_timeout_ms);
    if (m->shm_base) munmap(m->shm_base, m->shm_bytes);
    if (rank == 0) shm_unlink(m->shm_name.c_str());
    delete m;
    return KC_ERR_HIP;
  };
  // rank 0 creates and sizes the segment (a fresh file is all zero: every flag 0) and stores the magic word;
  // the others look the name up until they hold a segment that is (a) large enough, (b) carries the magic word
  // and (c) is still waiting for ranks.  A segment whose ranks have all attached is a LEFT-OVER of an earlier
  // communicator of this name (its rank 0 died between attach and unlink): never joined, looked up again.
  const auto t0 = std::chrono::steady_clock::now();
  auto late = [&] { return std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(m->shm_timeout_ms); };
  ShmHeader *h = nullptr;
  if (rank == 0) {
    shm_unlink(m->shm_name.c_str());
    const int fd = shm_open(m->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return fail("shm_open", true);
    if (ftruncate(fd, static_cast<off_t>(m->shm_bytes)) != 0) {
      close(fd);
      return fail("ftruncate", true);
    }
    m->shm_base = mmap(nullptr, m->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m->shm_base == MAP_FAILED) {
      m->shm_base = nullptr;
      return fail("mmap", true);
    }
    h = shm_header(m);
    h->world = world;
    h->magic.store(kShmMagic, std::memory_order_release);
  } else {
    for (;;) {
      const int fd = shm_open(m->shm_name.c_str(), O_RDWR, 0600);
      struct stat st {};
      if (fd >= 0 && fstat(fd, &st) == 0 && static_cast<size_t>(st.st_size) >= m->shm_bytes) {
        void *base = mmap(nullptr, m->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (base == MAP_FAILED) return fail("mmap", true);
        ShmHeader *hh = static_cast<ShmHeader *>(base);
        if (hh->magic.load(std::memory_order_acquire) == kShmMagic &&
            hh->attached.load(std::memory_order_acquire) < hh->world) {
          m->shm_base = base;
          h = hh;
          break;
        }
        munmap(base, m->shm_bytes);  // not sized / not signed yet, or stale
      } else if (fd >= 0) {
        close(fd);
      }
      if (late()) return fail("waiting for rank 0's segment", false);
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    if (h->world != world) {
      set_error("shm transport: segment %s was created for %lld ranks, not %d", m->shm_name.c_str(), h->world, world);
      munmap(m->shm_base, m->shm_bytes);
      delete m;
      return KC_ERR_INVALID;
    }
  }
  // everybody is attached before anybody returns (rank 0 unlinks the name then: the segment lives
  // as long as a process maps it and cannot collide with a later communicator of the same name)
  h->attached.fetch_add(1, std::memory_order_acq_rel);
  while (h->attached.load(std::memory_order_acquire) < world) {
    if (late()) return fail("attach", false);
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  if (rank == 0) shm_unlink(m->shm_name.c_str());
  hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&m->h_stage), 2 * kShmMaxWords * sizeof(long long),
                               hipHostMallocDefault);
  if (e != hipSuccess) {
    set_error("hipHostMalloc failed: %s", hipGetErrorString(e));
    munmap(m->shm_base, m->shm_bytes);
    delete m;
    return KC_ERR_HIP;
  }
  *out = m;
  return KC_OK;
#endif
}

void kc_comm_destroy(kc_comm *m) {
  if (!m) return;
  if (m->comm && rccl().ok) {
    hipError_t e = hipSetDevice(m->device);
    (void)e;
    (void)rccl().CommDestroy(m->comm);
  }
  if (m->h_stage) {
    hipError_t e = hipHostFree(m->h_stage);
    (void)e;
  }
  if (m->shm_base) munmap(m->shm_base, m->shm_bytes);
  delete m;
}

int kc_comm_rank(const kc_comm *m) { return m ? m->rank : -1; }
int kc_comm_world(const kc_comm *m) { return m ? m->world : 0; }
int kc_comm_transport(const kc_comm *m) { return m ? (m->shm ? KC_COMM_SHM : KC_COMM_RCCL) : -1; }

// What the TRANSPORT reports, not what the constructor was told: ncclCommCount / ncclCommUserRank /
// ncclCommCuDevice of the RCCL communicator; the attach count of the segment for the shared-memory transport.
int kc_comm_query(kc_comm *m, int *n_ranks, int *user_rank, int *device) {
  if (!m) KC_FAIL(KC_ERR_INVALID, "null communicator");
  int cnt = 0, ur = 0, dev = 0;
  if (m->shm) {
    cnt = m->shm_base ? static_cast<int>(shm_header(m)->attached.load(std::memory_order_acquire)) : 0;
    ur = m->rank;
    dev = m->device;
  } else {
    if (!m->comm || !rccl().ok) KC_FAIL(KC_ERR_STATE, "no RCCL communicator");
    KC_NCCL(rccl().CommCount(m->comm, &cnt));
    KC_NCCL(rccl().CommUserRank(m->comm, &ur));
    KC_NCCL(rccl().CommCuDevice(m->comm, &dev));
  }
  if (n_ranks) *n_ranks = cnt;
  if (user_rank) *user_rank = ur;
  if (device) *device = dev;
  return KC_OK;
}

}  // extern "C"
