// Device functions of the single-launch controller cycle (the CycleTail form of
// rollout_collide_kernel, kc_rollout_kernels.h): cost tables into LDS, cost of a
// workgroup's survivors from the poses it still holds, arrival ticket + final
// reduction by the workgroup that arrives last.  Part of kc_dwa.hip.
//
// Inter-workgroup visibility (MI355X: per-XCD L2s are not coherent with each
// other): everything one workgroup hands to the last arriver -- its key, its
// survivor mask, its best row -- is stored with agent-scope (sc1, write-through)
// stores, every storing wave drains them (s_waitcnt vmcnt(0)) in front of the
// workgroup barrier, ONE lane then takes the ticket with an agent-scope atomic
// add, and the last arriver reads with agent-scope (sc1) loads behind its own
// add and a workgroup barrier: the counter-ordered hand-off of
// MI355X_MICROARCH.md ("Workgroup dispatch, XCD placement & inter-workgroup
// visibility"), with no L2 write-back or invalidate on the critical path.
#pragma once

namespace kc {

struct CycleTail;  // kc_rollout_kernels.h
constexpr int kTeamMaxSurvivors = 4;  // one or two survivors: half the workgroup each, eight lanes per
                                      // trajectory point; three or four: a quarter each, four lanes per
                                      // point (one pass either way); more: one per wavefront

// LDS layout of the cost tables behind tab_off (the host sizes it the same way:
// cycle_table_bytes in kc_dwa_cycle.hip); every table starts on a 16-byte boundary
struct CycleTabs {
  float4 *xy, *za;  // [seg_pairs_padded] pair records of the segment points (struct SegPairs)
  float *cap;       // [8][nch] capsules, then [4][nsup] spheres, then [8][nsup] super-chunk capsules
  int *cells;       // [ncell + 1], padded to 4
  uint8_t *skip;    // [ncell], padded to 16
  float *mind;      // [4][P] team scratch
};
__device__ __forceinline__ CycleTabs cycle_tabs(const CostArgs &c, unsigned char *smem, unsigned tab_off) {
  CycleTabs t;
  const int ncell = c.b.W * c.b.H;
  const int npp = c.use_seg ? seg_pairs_padded(c.nch, c.seg_chunk) : 0;
  t.xy = reinterpret_cast<float4 *>(smem + tab_off);
  t.za = t.xy + npp;
  t.cap = reinterpret_cast<float *>(t.za + npp);
  t.cells = reinterpret_cast<int *>(t.cap + (c.use_seg ? (8 * c.nch + 12 * c.nsup + 3) & ~3 : 0));
  t.skip = reinterpret_cast<uint8_t *>(t.cells + (c.use_obs ? (ncell + 1 + 3) & ~3 : 0));
  t.mind = reinterpret_cast<float *>(t.skip + (c.use_obs ? ((ncell + 15) & ~15) : 0));
  return t;
}

// the plain copy loops (tables beyond what CycleTabRegs holds)
template <class Tail>
__device__ __forceinline__ void cycle_fill_tables(const Tail &tail, unsigned char *smem, int tid, int nthreads) {
  const CostArgs &c = tail.c;
  const CycleTabs t = cycle_tabs(c, smem, tail.tab_off);
  if (c.use_seg) {
    const int npp = seg_pairs_padded(c.nch, c.seg_chunk);
    for (int k = tid; k < npp; k += nthreads) seg_pair_from_rows(c.sx, c.sy, c.szz, c.acc_seg, c.S, k, t.xy[k], t.za[k]);
    const float *gc = c.sx + seg_cap_offset(c.S);
    for (int j = tid; j < 8 * c.nch + 12 * c.nsup; j += nthreads) t.cap[j] = gc[j];
  }
  if (c.use_obs) {
    const int ncell = c.b.W * c.b.H;
    // (both tables are allocated with room for the last vector: kc_dwa_sensor.hip)
    const int4 *gcell = reinterpret_cast<const int4 *>(c.b.cell_start);
    int4 *lcell = reinterpret_cast<int4 *>(t.cells);
    for (int j = tid; j < (ncell + 1 + 3) / 4; j += nthreads) lcell[j] = gcell[j];
    const uint4 *gs = reinterpret_cast<const uint4 *>(c.b.skip);
    uint4 *ls = reinterpret_cast<uint4 *>(t.skip);
    for (int j = tid; j < (ncell + 15) / 16; j += nthreads) ls[j] = gs[j];
  }
}

// The same in two halves -- every global load first (into registers), the LDS
// stores later -- so that ONE memory latency is paid for all of them instead of one
// per copy loop.  Capacity: one segment pair, two capsule words, two vectors of four
// cell words and one of sixteen skip bytes per thread; `ok` false: the plain loops above.
// The threads that copy are the caller's choice (tid 0 .. nthreads - 1).
typedef int CycleV4i __attribute__((ext_vector_type(4)));  // (native vectors: the registers of a thread, never memory)
template <int kBlock>
struct CycleTabRegs {
  float4 sxy, sza;
  float cap[2];
  CycleV4i cells[2];
  CycleV4i skip;
  bool ok;
};
template <int kBlock, class Tail>
__device__ __forceinline__ void cycle_tables_load(const Tail &tail, int tid, int nthreads, CycleTabRegs<kBlock> &r) {
  const CostArgs &c = tail.c;
  const int ncell = c.use_obs ? c.b.W * c.b.H : 0;
  const int capw = c.use_seg ? 8 * c.nch + 12 * c.nsup : 0;
  const int npp = c.use_seg ? seg_pairs_padded(c.nch, c.seg_chunk) : 0;
  r.ok = npp <= nthreads && capw <= 2 * nthreads && (ncell + 1 + 3) / 4 <= 2 * nthreads && (ncell + 15) / 16 <= nthreads;
  if (!r.ok) return;
  if (c.use_seg) {
    if (tid < npp) seg_pair_from_rows(c.sx, c.sy, c.szz, c.acc_seg, c.S, tid, r.sxy, r.sza);
    const float *gc = c.sx + seg_cap_offset(c.S);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = tid + u * nthreads;
      if (j < capw) r.cap[u] = gc[j];
    }
  }
  if (c.use_obs) {
    const CycleV4i *gcell = reinterpret_cast<const CycleV4i *>(c.b.cell_start);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = tid + u * nthreads;
      r.cells[u] = CycleV4i{0, 0, 0, 0};
      if (j < (ncell + 1 + 3) / 4) r.cells[u] = gcell[j];
    }
    r.skip = CycleV4i{0, 0, 0, 0};
    if (tid < (ncell + 15) / 16) r.skip = reinterpret_cast<const CycleV4i *>(c.b.skip)[tid];
  }
}
template <int kBlock, class Tail>
__device__ __forceinline__ void cycle_tables_store(const Tail &tail, unsigned char *smem, int tid, int nthreads,
                                                   const CycleTabRegs<kBlock> &r) {
  const CostArgs &c = tail.c;
  if (!r.ok) {
    cycle_fill_tables(tail, smem, tid, nthreads);
    return;
  }
  const CycleTabs t = cycle_tabs(c, smem, tail.tab_off);
  const int ncell = c.use_obs ? c.b.W * c.b.H : 0;
  if (c.use_seg) {
    const int capw = 8 * c.nch + 12 * c.nsup;
    if (tid < seg_pairs_padded(c.nch, c.seg_chunk)) {
      t.xy[tid] = r.sxy;
      t.za[tid] = r.sza;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = tid + u * nthreads;
      if (j < capw) t.cap[j] = r.cap[u];
    }
  }
  if (c.use_obs) {
    CycleV4i *lcell = reinterpret_cast<CycleV4i *>(t.cells);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = tid + u * nthreads;
      if (j < (ncell + 1 + 3) / 4) lcell[j] = r.cells[u];
    }
    if (tid < (ncell + 15) / 16) reinterpret_cast<CycleV4i *>(t.skip)[tid] = r.skip;
  }
}

// Cost of the R survivors of this workgroup (slots lsurv[0..R), ascending).  Few
// survivors: two at a time, half the workgroup each, eight lanes per point
// (team_sample_search); many: one per wavefront, pulled from an LDS counter
// (wave_sample_total).  Same arithmetic as the stand-alone cost kernels.  Returns
// the workgroup's best key (uniform) and the slot of that sample.
template <int kSamples, int kBlock, class Tail>
__device__ __forceinline__ long long cycle_costs(const RollArgs &a, const Tail &tail, unsigned char *smem,
                                                 const double2 *lpos, int PP, const int *lperm,
                                                 const int *lsurv, int R, int tid, int *best_slot) {
  const CostArgs &c = tail.c;
  __shared__ long long s_key;
  __shared__ unsigned long long s_ob[kBlock / 64];
  __shared__ float s_goal[4], s_end[4];
  __shared__ int s_next, s_bslot;
  const CycleTabs t = cycle_tabs(c, smem, tail.tab_off);
  const float sz_end = (c.use_seg && c.S > 0) ? c.sz[c.S - 1] : 0.0f;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    s_key = KEY_NONE;
    s_next = 0;
    s_bslot = -1;
  }
#ifdef KC_PHASE_STAMPS
  if (a.dbg && tid == 0 && blockIdx.x < 512) a.dbg[(size_t)blockIdx.x * 32 + 21] = static_cast<unsigned long long>(R) + 1ull;  // survivors + 1
#endif
  const SegPairs seg{t.xy, t.za};
  // what the wavefront obstacle term reads: the scan block when there is a scan's near table, the buckets' arrays otherwise
  const float *const wobx = tail.t.onear != nullptr ? tail.t.osx : c.b.bx;
  const float *const woby = tail.t.onear != nullptr ? tail.t.osy : c.b.by;
  const bool teams = R <= tail.team_max;
  if (teams) {
    // every survivor at once: R teams (two halves or four quarters of the workgroup)
    const bool quarters = R > 2;
    const int team = quarters ? kBlock / 4 : kBlock / 2;
    const int h = tid / team, tt = tid - h * team;
    const bool active = h < R;
    if (tt == 0) s_ob[h] = static_cast<unsigned long long>(__double_as_longlong(DBL_MAX));
    __syncthreads();  // (also s_key)
    const int s = active ? lsurv[h] : 0;
    const PosePts pts{lpos + s * PP, PP - 1};
    // The obstacle term of a team's sample is formed by the team's LAST wavefront, a lane a point, the way the
    // wavefront-per-sample path forms it (wave_obstacle_term: near table of a scan / one scan of the union rectangle /
    // cooperative ring walk) -- beside the segment search of the other wavefronts when that wavefront holds no point of it
    // (halves: 400 of 512 lanes search), behind its share otherwise (quarters).  The block walk a point apiece that the
    // teams ran before took as many trips as the longest row of any point of a wavefront: quarters were 16 us late in
    // clutter (cfg2, 38-step horizon).  Trajectories of more than 64 points keep the walk.
    const bool wave_obs = c.use_obs && c.P <= 64;
    const int lanes_per = quarters ? 4 : 8;
    const bool last_wave = tt >= team - 64;
    const bool holds_points = (team - 64) / lanes_per < c.P;  // (uniform: the last wavefront's first point slot)
    if (active && !(wave_obs && last_wave && !holds_points)) {
      if (quarters)
        team_sample_search<kBlock / 4, SegPairs, PosePts, 4>(c, seg, sz_end, t.cells, t.skip, c.b.bx, c.b.by, pts, tt,
                                                            t.mind + h * c.P, &s_goal[h], &s_end[h], &s_ob[h],
                                                            t.cap, t.cap + 8 * c.nch, wave_obs);
      else
        team_sample_search<kBlock / 2, SegPairs, PosePts, 8>(c, seg, sz_end, t.cells, t.skip, c.b.bx, c.b.by, pts, tt,
                                                            t.mind + h * c.P, &s_goal[h], &s_end[h], &s_ob[h],
                                                            t.cap, t.cap + 8 * c.nch, wave_obs);
    }
    if (active && wave_obs && last_wave) {
      const bool live = lane < c.P;
      const int p = live ? lane : c.P - 1;
      double ubound2 = DBL_MAX;
      wave_obstacle_term(c, tail.t, t.cells, t.skip, wobx, woby, pts.x(p), pts.y(p), live, lane, &s_ob[h], ubound2);
    }
    KC_RSTAMP(10);
    __syncthreads();
    KC_RSTAMP(11);
    if (active && tt < 64) {
      const int n = lperm[s];
      const float total = team_sample_total(c, n, lane, t.mind + h * c.P, s_goal[h], s_end[h], s_ob[h]);
      if (lane == 0) {
        c.costs[n] = total;
        if (total < FLT_MAX)  // `total_cost < minCost`, minCost starts at FLT_MAX
          atomicMin(&s_key, key_pack(total, static_cast<uint32_t>(c.first + n)));
      }
    }
    KC_RSTAMP(12);
  } else {
    __syncthreads();  // s_next, s_key
    const float *cap = t.cap, *sup = t.cap + 8 * c.nch;
    long long wkey = KEY_NONE;
    for (;;) {
      int q = 0;
      if (lane == 0) q = atomicAdd(&s_next, 1);
      q = __builtin_amdgcn_readfirstlane(q);
      if (q >= R) break;
      const int s = lsurv[q];
      const int n = lperm[s];
      const PosePts pts{lpos + s * PP, PP - 1};
      const float total = wave_sample_total(c, tail.t, seg, cap, sup, sz_end, t.cells, t.skip,
                                            wobx, woby, pts, n, lane, &s_ob[wave], false);
      if (lane == 0) c.costs[n] = total;
      if (total < FLT_MAX) {
        const long long k = key_pack(total, static_cast<uint32_t>(c.first + n));
        wkey = k < wkey ? k : wkey;
      }
    }
    if (lane == 0 && wkey != KEY_NONE) atomicMin(&s_key, wkey);
  }
  __syncthreads();
  const long long key = s_key;
  if (key != KEY_NONE && tid < R) {
    const int s = lsurv[tid];
    if (static_cast<uint32_t>(c.first + lperm[s]) == static_cast<uint32_t>(key & 0xFFFFFFFFll)) s_bslot = s;
  }
  __syncthreads();
  *best_slot = s_bslot;
  return key;
}

// Single-GPU epilogue: this workgroup's 32-byte slot + its best row into pinned
// host memory, nothing else (see CycleTail::host_slots).  mask: survivors by slot.
template <int kBlock, class Tail>
__device__ __forceinline__ void cycle_epilogue_host(const RollArgs &a, const Tail &tail, long long key,
                                                    unsigned long long mask, int best_slot,
                                                    const double2 *best_row, int tid) {
  const int P = a.P;
  const unsigned b = blockIdx.x;
  // wavefront 0 alone (the row lies in LDS behind the barrier of the cost phase): row out, its check word by a
  // wave reduction, the slot -- no barrier, no LDS atomic on the tail of the workgroup
  if (tid >= 64) return;
  const bool have_row = best_slot >= 0 && tail.host_rows != nullptr;
  unsigned int x = 0u;
  if (have_row) {
    uint32_t *dst = tail.host_rows + (size_t)b * 2 * P;
    for (int k = tid; k < 2 * P; k += 64) {
      const int p = k < P ? k : k - P;
      const double2 r = best_row[p == 0 ? (P | 1) - 1 : p - 1];  // pose 0 sits in the spare slot of the row
      const uint32_t w = __float_as_uint(static_cast<float>(k < P ? r.x : r.y));
      dst[k] = w;
      x ^= w * (2u * static_cast<unsigned>(k) + 1u);
    }
    for (int off = 32; off > 0; off >>= 1) x ^= __shfl_xor(x, off, 64);
  }
  if (tid == 0) {
    const long long w1 = static_cast<long long>((mask & 0xFFFFFFFFull) | (static_cast<unsigned long long>(x) << 32));
    const long long w2 = tail.seq | (have_row ? (1ll << 61) : 0ll);
    longlong2 *v = reinterpret_cast<longlong2 *>(tail.host_slots + 4 * b);
    longlong2 lo, hi;
    lo.x = key;
    lo.y = w1;
    hi.x = w2;
    hi.y = record_check(key, w1, w2, static_cast<long long>(b));
    v[0] = lo;
    v[1] = hi;
  }
}

// Arrival ticket; the workgroup that arrives last publishes the cycle.
// R survivors of this workgroup (slots lsurv[0..R)) are OR-ed into the device-wide
// bitmap of admissible local ids first.
template <int kBlock, class Tail>
__device__ __forceinline__ void cycle_epilogue(const RollArgs &a, const Tail &tail, long long key, int R,
                                               int best_slot, const double2 *best_row, const int *lperm,
                                               const int *lsurv, int tid) {
  const int P = a.P;
  const unsigned b = blockIdx.x, G = gridDim.x;
  __shared__ int s_last, s_bw;
  __shared__ long long s_wkey[kBlock / 64];
  __shared__ int s_wadm[kBlock / 64], s_wcnt[kBlock / 64];
  __shared__ unsigned int s_rowx, s_winx;
  if (tid == 0) {
    s_rowx = 0u;
    s_bw = -1;
  }
  if (tid < R) {
    const int id = lperm[lsurv[tid]];
    __hip_atomic_fetch_or(tail.adm_bits + (id >> 5), 1u << (id & 31), __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_AGENT);
  }
  // This workgroup's best row -- the floats the roll-out would have stored -- goes
  // straight into its slot of the pinned host buffer (posted writes: nobody on
  // the device waits for them -- they are issued BEHIND the ticket); a position-
  // weighted xor of the words goes with the key, the host checks the row of the
  // winning workgroup against it.
  __syncthreads();  // s_rowx
  const bool have_row = best_slot >= 0 && tail.host_rows != nullptr;
  uint32_t roww[2] = {0u, 0u};  // this lane's words of the row (2 P <= 2 * kBlock)
  if (have_row) {
    unsigned int x = 0u;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = tid + u * kBlock;
      if (k < 2 * P) {
        const int p = k < P ? k : k - P;
        const double2 r = best_row[p == 0 ? (P | 1) - 1 : p - 1];  // pose 0 sits in the spare slot of the row
        roww[u] = __float_as_uint(static_cast<float>(k < P ? r.x : r.y));
        x ^= roww[u] * (2u * static_cast<unsigned>(k) + 1u);
      }
    }
    for (int off = 32; off > 0; off >>= 1) x ^= __shfl_xor(x, off, 64);
    if ((tid & 63) == 0 && x) atomicXor(&s_rowx, x);
    __syncthreads();
  }
  if (tid == 0) {
    // key + row word as one 16-byte agent-scope record of this workgroup
    st_agent(tail.block_keys + 2 * b, key);
    st_agent(tail.block_keys + 2 * b + 1, static_cast<long long>(s_rowx));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave drains its atomics / the record
  __syncthreads();
  if (tid == 0) {
    const unsigned long long t = __hip_atomic_fetch_add(
        reinterpret_cast<unsigned long long *>(tail.result + W_TICKET), 1ull, __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == static_cast<unsigned long long>(G) - 1ull) ? 1 : 0;
  }
  if (have_row) {  // behind the ticket: posted PCIe writes nothing here waits for
    uint32_t *dst = tail.host_rows + (size_t)b * 2 * P;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = tid + u * kBlock;
      if (k < 2 * P) dst[k] = roww[u];
    }
  }
  __syncthreads();
  KC_RSTAMP(13);
  if (!s_last) return;
  // ---- last arriver: every other workgroup's stores are behind its ticket ----
  // one round trip: the keys, the bitmap words and the error word, kept in registers
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int kMaxKeys = 2, kMaxWords = 2;  // per lane: <= 2048 workgroups, <= 65536 samples (host check)
  long long kreg[kMaxKeys], xreg[kMaxKeys];
  uint32_t wreg[kMaxWords];
  const unsigned nwords = (static_cast<unsigned>(a.n) + 31u) >> 5;
  long long err = 0;
  if (tid == 0) err = static_cast<long long>(ld_agent(reinterpret_cast<const unsigned long long *>(a.dev_err)));
#pragma unroll
  for (int u = 0; u < kMaxKeys; ++u) {
    const unsigned g = tid + u * kBlock;
    kreg[u] = g < G ? ld_agent(tail.block_keys + 2 * g) : KEY_NONE;
    xreg[u] = g < G ? ld_agent(tail.block_keys + 2 * g + 1) : 0;
  }
#pragma unroll
  for (int u = 0; u < kMaxWords; ++u) {
    const unsigned w = tid + u * kBlock;
    wreg[u] = w < nwords ? ld_agent(tail.adm_bits + w) : 0u;
  }
  long long k = kreg[0] < kreg[1] ? kreg[0] : kreg[1];
  int nadm = __popc(wreg[0]) + __popc(wreg[1]);
#pragma unroll
  for (int u = 0; u < kMaxWords; ++u) {  // the bitmap is clear again for the next cycle
    const unsigned w = tid + u * kBlock;
    if (w < nwords && wreg[u]) tail.adm_bits[w] = 0u;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const long long o = __shfl_xor(k, off, 64);
    k = o < k ? o : k;
    nadm += __shfl_xor(nadm, off, 64);
  }
  if (lane == 0) {
    s_wkey[wave] = k;
    s_wadm[wave] = nadm;
  }
  __syncthreads();
  long long fkey = s_wkey[0];
  int na = s_wadm[0];
  for (int w = 1; w < kBlock / 64; ++w) {
    fkey = s_wkey[w] < fkey ? s_wkey[w] : fkey;
    na += s_wadm[w];
  }
  // the reference's index counts the admissible samples in front of the winner
  // (generation order = local id order)
  int cnt = 0;
  if (fkey != KEY_NONE) {
    const unsigned lim = static_cast<uint32_t>(fkey & 0xFFFFFFFFll) - static_cast<unsigned>(a.first);
#pragma unroll
    for (int u = 0; u < kMaxWords; ++u) {
      const unsigned w0 = (tid + u * kBlock) << 5;  // first id of this word
      if (w0 + 32u <= lim) cnt += __popc(wreg[u]);
      else if (w0 < lim) cnt += __popc(wreg[u] & ((1u << (lim - w0)) - 1u));
    }
#pragma unroll
    for (int u = 0; u < kMaxKeys; ++u)
      if (kreg[u] == fkey) {  // one owner: indices are unique
        s_bw = tid + u * kBlock;
        s_winx = static_cast<unsigned int>(xreg[u]);
      }
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  if (lane == 0) s_wcnt[wave] = cnt;
  if (tail.xs) {
    // the rank's words of the exchange record: its bitmap (the words this workgroup holds in registers)
    uint32_t *region = reinterpret_cast<uint32_t *>(tail.xs + X_REGIONS + static_cast<size_t>(tail.xrank) * tail.xrw);
#pragma unroll
    for (int u = 0; u < kMaxWords; ++u) {
      const unsigned w = tid + u * kBlock;
      if (w < 2u * static_cast<unsigned>(tail.xrw)) region[w] = w >= nwords ? 0u : wreg[u];
    }
  }
  __syncthreads();
  KC_RSTAMP(14);
  if (tid == 0 && tail.xs) {
    long long key = fkey;
    if (key != KEY_NONE && tail.xgid) {
      const uint32_t lat = static_cast<uint32_t>(key & 0xFFFFFFFFll);
      key = (key & ~0xFFFFFFFFll) | static_cast<long long>(static_cast<uint32_t>(tail.xgid[lat]));
    }
    tail.xs[X_KEY] = err ? KEY_NONE : key;
    tail.xs[X_ERR] = err ? -1ll : 0ll;
  }
  if (tid == 0) {
    int s = 0;
    for (int w = 0; w < kBlock / 64; ++w) s += s_wcnt[w];
    if (fkey == KEY_NONE) s = -1;
    const long long na_pub = err ? -1 : na;
    const long long w1 = (na_pub << 32) | static_cast<long long>(static_cast<uint32_t>(s));
    // row word: check value of the winner's row, the workgroup slot it lies in, presence bit
    const bool has_row = fkey != KEY_NONE && tail.host_rows != nullptr;
    const long long w4 = has_row ? ((static_cast<long long>(s_winx) << 32) |
                                    (static_cast<long long>(static_cast<uint32_t>(s_bw)) << 1) | 1ll)
                                 : 0ll;
    if (tail.host_pub) store_host_record(tail.host_pub, fkey, w1, tail.seq, w4);
    tail.result[R_KEY] = fkey;
    tail.result[R_NADM] = na_pub;
    tail.result[R_COMPACT] = s;
    tail.result[W_KEY] = KEY_NONE;
    tail.result[W_NADM] = 0;   // device error word
    tail.result[W_TICKET] = 0;
    tail.result[W_LIST] = 0;
  }
}

}  // namespace kc
