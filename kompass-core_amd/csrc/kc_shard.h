// Sharding of the sample list over the ranks of a communicator and the exchange
// record of a sharded cycle.  Part of kc_dwa.hip.
//
// ONE all-reduce(int64, min) per cycle carries everything the ranks owe each other:
//
//   X[0]                      packed key of the rank's best sample, GLOBAL raw index
//                             (KEY_NONE: nothing admissible) -- the minimum over the ranks is
//                             LowestCost::combine (datatypes/trajectory.h:630-636)
//   X[1]                      0; -1 when this rank's device error word is set; -2 when the rank
//                             failed before the exchange.  The minimum makes
//                             the failure collective: every rank fails the SAME cycle and none
//                             is a collective out of step
//   X[2 + r*rw ... + rw)      rank r's admissible bitmap by shard-local sample id (bit i of
//                             word j = local sample 64 j + i); every other rank contributes
//                             INT64_MAX there, so the minimum delivers r's words unchanged
//
// With the bitmaps every rank forms the reference's admissible-only index of the
// winner (samples in front of it in GENERATION order, trajectory_sampler.cpp:
// 207-217,256-272) and the global admissible count by itself: no second collective.
// rw = ceil(largest shard / 64): 1 KB per rank for an 8192-sample shard -- the message
// stays in the latency-bound range of RCCL's LL protocol.
#pragma once

#include <algorithm>

namespace kc {

enum { X_KEY = 0, X_ERR = 1, X_REGIONS = 2 };

struct ShardLayout {
  int mode = -1;  // -1: explicit kc_dwa_set_shard; KC_SHARD_BLOCKS; KC_SHARD_ROWS
  int rank = 0, world = 1;
  size_t n_total = 0;
  std::vector<size_t> first, count;        // per rank (ROWS: count only)
  std::vector<std::vector<int32_t>> gids;  // ROWS: ascending global sample ids per rank
  std::vector<int32_t> rows_seen;          // ROWS: the trig-row pattern `gids` was dealt from

  size_t max_count() const {
    size_t m = 0;
    for (size_t c : count) m = std::max(m, c);
    return m;
  }
  // samples of rank r with a global id below `raw`
  size_t before(int r, int64_t raw) const {
    if (raw <= 0) return 0;
    if (mode == KC_SHARD_ROWS) {
      const auto &g = gids[static_cast<size_t>(r)];
      return static_cast<size_t>(std::lower_bound(g.begin(), g.end(), static_cast<int32_t>(std::min<int64_t>(raw, INT32_MAX))) -
                                 g.begin());
    }
    const int64_t f = static_cast<int64_t>(first[static_cast<size_t>(r)]);
    const int64_t c = static_cast<int64_t>(count[static_cast<size_t>(r)]);
    return static_cast<size_t>(std::min(std::max<int64_t>(raw - f, 0), c));
  }
  // local id of global sample `raw` on rank r, or -1
  int64_t local_of(int r, int64_t raw) const {
    if (raw < 0) return -1;
    if (mode == KC_SHARD_ROWS) {
      const auto &g = gids[static_cast<size_t>(r)];
      const auto it = std::lower_bound(g.begin(), g.end(), static_cast<int32_t>(std::min<int64_t>(raw, INT32_MAX)));
      return (it != g.end() && *it == raw) ? static_cast<int64_t>(it - g.begin()) : -1;
    }
    const int64_t f = static_cast<int64_t>(first[static_cast<size_t>(r)]);
    return (raw >= f && raw < f + static_cast<int64_t>(count[static_cast<size_t>(r)])) ? raw - f : -1;
  }
};

// KC_SHARD_BLOCKS: rank r owns the contiguous block [n r / W, n (r + 1) / W) of the list.
inline void shard_blocks(size_t n, int world, ShardLayout &L) {
  L.first.assign(static_cast<size_t>(world), 0);
  L.count.assign(static_cast<size_t>(world), 0);
  L.gids.clear();
  L.rows_seen.clear();
  for (int r = 0; r < world; ++r) {
    const size_t f = n * static_cast<size_t>(r) / static_cast<size_t>(world);
    const size_t l = n * (static_cast<size_t>(r) + 1) / static_cast<size_t>(world);
    L.first[static_cast<size_t>(r)] = f;
    L.count[static_cast<size_t>(r)] = l - f;
  }
}

// KC_SHARD_ROWS: samples are dealt by TRIG ROW (distinct omega), so that a rank evaluates 1 / W
// of the host's cos / sin table instead of all of it (a contiguous block of the vx-major lattice
// holds every omega).  Row a (in order of first appearance; `rows` holds each sample's row) goes
// to rank (k mod W), k counting the rows dealt so far; a row that holds more than twice the mean
// (the omega = 0 row of an omni lattice: every (vx, vy, 0) sample) is dealt sample by sample
// instead.  owner[g] = rank of sample g.  Pure function of (rows, world): every rank computes the
// same deal from the same list.
inline void shard_rows_owner(const int32_t *rows, size_t n, int world, int32_t *owner) {
  int32_t A = 0;
  for (size_t g = 0; g < n; ++g) A = std::max(A, rows[g] + 1);
  std::vector<size_t> m(static_cast<size_t>(A), 0);
  for (size_t g = 0; g < n; ++g) ++m[static_cast<size_t>(rows[g])];
  const size_t mean = A > 0 ? (n + static_cast<size_t>(A) - 1) / static_cast<size_t>(A) : 0;
  std::vector<int32_t> row_rank(static_cast<size_t>(A), -2);  // -2: not met yet, -1: dealt by sample
  size_t k = 0, heavy = 0;
  for (size_t g = 0; g < n; ++g) {
    int32_t &rr = row_rank[static_cast<size_t>(rows[g])];
    if (rr == -2) {
      if (m[static_cast<size_t>(rows[g])] > 2 * mean && world > 1) rr = -1;
      else rr = static_cast<int32_t>(k++ % static_cast<size_t>(world));
    }
    owner[g] = rr >= 0 ? rr : static_cast<int32_t>(heavy++ % static_cast<size_t>(world));
  }
}

inline void shard_rows(const std::vector<int32_t> &rows, int world, ShardLayout &L) {
  const size_t n = rows.size();
  std::vector<int32_t> owner(n);
  shard_rows_owner(rows.data(), n, world, owner.data());
  L.first.assign(static_cast<size_t>(world), 0);
  L.count.assign(static_cast<size_t>(world), 0);
  L.gids.assign(static_cast<size_t>(world), {});
  for (size_t g = 0; g < n; ++g) L.gids[static_cast<size_t>(owner[g])].push_back(static_cast<int32_t>(g));
  for (int r = 0; r < world; ++r) L.count[static_cast<size_t>(r)] = L.gids[static_cast<size_t>(r)].size();
  L.rows_seen = rows;
}

inline int popcount_prefix(const long long *words, size_t nbits) {
  int c = 0;
  const size_t full = nbits >> 6;
  for (size_t j = 0; j < full; ++j) c += __builtin_popcountll(static_cast<unsigned long long>(words[j]));
  if (nbits & 63) c += __builtin_popcountll(static_cast<unsigned long long>(words[full]) & ((1ull << (nbits & 63)) - 1ull));
  return c;
}

// The reduced record -> the result every rank returns.  *failed: some rank reported an error.
inline void merge_exchange(const ShardLayout &L, const long long *X, size_t rw, kc_result *out, bool *failed) {
  kc_result r{};
  *failed = X[X_ERR] < 0;
  const long long key = X[X_KEY];
  long long nadm = 0;
  for (int q = 0; q < L.world; ++q)
    nadm += popcount_prefix(X + X_REGIONS + static_cast<size_t>(q) * rw, L.count[static_cast<size_t>(q)]);
  r.n_admissible = nadm;
  r.n_samples = static_cast<int64_t>(L.n_total);
  if (key == KEY_NONE || *failed) {
    r.found = 0;
    r.cost = 0.0f;
    r.index = -1;
    r.raw_index = -1;
  } else {
    r.found = 1;
    r.cost = sortable_float(static_cast<int32_t>(key >> 32));
    r.raw_index = static_cast<int64_t>(static_cast<uint32_t>(key & 0xFFFFFFFFll));
    long long idx = 0;
    for (int q = 0; q < L.world; ++q)
      idx += popcount_prefix(X + X_REGIONS + static_cast<size_t>(q) * rw, L.before(q, r.raw_index));
    r.index = idx;
  }
  *out = r;
}

// ---- device side ---------------------------------------------------------------------------
struct PackArgs {
  const long long *result;  // device record of the cycle (R_KEY, R_NADM)
  const uint8_t *flags;     // [n] admissible flags by shard-local id
  int n;                    // shard size (the flags are indexed by shard-local id)
  const int32_t *gid;       // local-lattice id -> global id (null: identity)
  long long *xs;            // send record
  int rank, rw;
  int host_key;             // 1: the key comes from the host (kc_dwa_exchange_best: host-side costs were added to the
  long long key;            //    device totals), already with the GLOBAL raw index; the device record is not read
};
// one workgroup: this rank's words of the send record from the cycle's device record + flags
#ifdef KC_TU_SHARD  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(1024) void xchg_pack_kernel(PackArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long *region = a.xs + X_REGIONS + static_cast<size_t>(a.rank) * a.rw;
  for (int j = wave; j < a.rw; j += static_cast<int>(blockDim.x >> 6)) {
    const int i = j * 64 + lane;
    const bool f = i < a.n && a.flags[i] != 0;
    const unsigned long long bal = __ballot(f);
    if (lane == 0) region[j] = static_cast<long long>(bal);
  }
  if (threadIdx.x == 0 && a.host_key) {
    a.xs[X_KEY] = a.key;
    a.xs[X_ERR] = 0ll;
  } else if (threadIdx.x == 0) {
    long long key = a.result[R_KEY];
    const bool err = a.result[R_NADM] < 0;
    if (key != KEY_NONE && a.gid) {
      const uint32_t lat = static_cast<uint32_t>(key & 0xFFFFFFFFll);
      key = (key & ~0xFFFFFFFFll) | static_cast<long long>(static_cast<uint32_t>(a.gid[lat]));
    }
    a.xs[X_KEY] = err ? KEY_NONE : key;
    a.xs[X_ERR] = err ? -1ll : 0ll;
  }
}
#endif  // KC_TU_SHARD
// this rank failed before the exchange: it still takes part, with the error word set
#ifdef KC_TU_SHARD  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ void xchg_fail_kernel(long long *xs, int rank, int rw) {
  for (int j = threadIdx.x; j < rw; j += blockDim.x) xs[X_REGIONS + static_cast<size_t>(rank) * rw + j] = 0;
  if (threadIdx.x == 0) {
    xs[X_KEY] = KEY_NONE;
    xs[X_ERR] = -2ll;
  }
}
#endif  // KC_TU_SHARD

__host__ __device__ inline unsigned long long xchg_word_mix(long long w, unsigned i) {
  return rec_mix(static_cast<unsigned long long>(w) + 0x9E3779B97F4A7C15ull * (static_cast<unsigned long long>(i) + 1ull));
}
// the reduced record into pinned host memory (plain stores) + a 5-word record {checksum of the
// words, length, sequence} the host polls; the words are accepted when their checksum adds up
#ifdef KC_TU_SHARD  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(256) void xchg_publish_kernel(const long long *xr, int len, long long *host_vec,
                                                           long long *host_rec, long long seq) {
  __shared__ unsigned long long wsum[4];
  unsigned long long s = 0ull;
  for (int i = threadIdx.x; i < len; i += 256) {
    const long long w = xr[i];
    host_vec[i] = w;
    s += xchg_word_mix(w, static_cast<unsigned>(i));
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    store_host_record(host_rec, static_cast<long long>(t), static_cast<long long>(len), seq, 0);
  }
}
#endif  // KC_TU_SHARD

}  // namespace kc
