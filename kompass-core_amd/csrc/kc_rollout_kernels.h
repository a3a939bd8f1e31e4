// Roll-out and collision kernels (A2-A4): split path, sensor-bitmap dilation,
// fused roll-out + pose gate, pose-parallel collision pass.  Part of kc_dwa.hip.
#pragma once

#include "kc_sincostab.h"
#include "kc_trig_exact.h"

namespace kc {

// ===========================================================================
// K0: the (step x omega row) table of cos / sin(yaw_k) on the device, for the kernels of the split
// path (rollout_kernel, collision_kernel, collision_tilted_kernel) -- what the host's libm produced
// every cycle in rounds 1-3 -- and as a job riding in another launch (below).
// ===========================================================================
// The same table as a JOB that rides in another launch (the sensor build of a controller cycle with fresh
// inputs: a few extra workgroups on CUs that launch leaves idle).  A lane per entry, yaw_k by the lane's own
// repeated additions; consecutive lanes take consecutive rows of one step (coalesced stores).  `tab` is the
// context's copy of the 440 table values in device memory.
struct TrigJob {
  double yaw0, dt;
  const double *omega, *tab;
  double2 *out;  // [P][A]
  int A, P;
  int nblk;      // workgroups of the carrying launch that belong to the job (the LAST nblk); 0: no job
};
template <int kBlock>
__device__ __forceinline__ void trig_job_block(const TrigJob &j, int blk) {
  const int total = j.A * j.P;
  for (int i = blk * kBlock + static_cast<int>(threadIdx.x); i < total; i += j.nblk * kBlock) {
    const int k = i / j.A, r = i - k * j.A;
    const double w = j.omega[r] * j.dt;
    double yaw = j.yaw0;
    int q = 0;
    for (; q + 4 <= k; q += 4) yaw = (((yaw + w) + w) + w) + w;
    for (; q < k; ++q) yaw += w;
    double sn, cs;
    trig::sincos_exact(yaw, &sn, &cs, j.tab);
    j.out[i] = make_double2(cs, sn);
  }
}

// stand-alone: every workgroup of the launch belongs to the job (any horizon: no LDS)
constexpr int kTrigBlock = 256;
#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(kTrigBlock) void trig_table_kernel(TrigJob j) {
  trig_job_block<kTrigBlock>(j, static_cast<int>(blockIdx.x));
}
#endif  // KC_TU_CYCLE

// ===========================================================================
// K1a: roll-out.  One lane per sample: the recurrence x_{k+1} = x_k + (...) is
// serial in k and keeps the reference's addition order (path.h:24-30).  The
// float path leaves through an LDS tile so the sample-major rows are written as
// whole contiguous lines; the double poses go out step-major (coalesced) for
// the collision pass.
// ===========================================================================
constexpr int kRollBlock = 64;

#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(kRollBlock) void rollout_kernel(RollArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int P1 = a.P | 1;  // odd row pitch: conflict-free column writes
  float *tx = reinterpret_cast<float *>(smem);
  float *ty = tx + (size_t)kRollBlock * P1;

  const int tid = threadIdx.x;
  const int base = blockIdx.x * kRollBlock;
  const int n = base + tid;

  if (n < a.n) {
    const double vx = sample_vx(a, n);
    const double vy = sample_vy(a, n);
    const int r = a.row[a.first + n];
    double x = a.x0, y = a.y0;
    const float fx0 = static_cast<float>(x), fy0 = static_cast<float>(y);
    if (a.stage) {
      tx[tid * P1] = fx0;
      ty[tid * P1] = fy0;
    } else {
      a.px[(size_t)n * a.P] = fx0;
      a.py[(size_t)n * a.P] = fy0;
    }
    const bool want_pos = a.c.enabled != 0;
    // all trig rows of (up to) 64 steps are requested at once and held in
    // registers, so the serial recurrence pays the memory latency once
    constexpr int CH = 64;
    double2 tr[CH];
    const int steps = a.P - 1;
    for (int k0 = 0; k0 < steps; k0 += CH) {
#pragma unroll
      for (int j = 0; j < CH; ++j)
        tr[j] = a.trig[(size_t)min(k0 + j, steps - 1) * a.A + r];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int k = k0 + j;
        if (k < steps) {
          const double2 cs = tr[j];
          // Path::State::update, datatypes/path.h:24-30
          x += (vx * cs.x - vy * cs.y) * a.dt;
          y += (vx * cs.y + vy * cs.x) * a.dt;
          if (want_pos) a.pos[(size_t)(k + 1) * a.n + n] = make_double2(x, y);
          const float fx = static_cast<float>(x), fy = static_cast<float>(y);
          if (a.stage) {
            tx[tid * P1 + k + 1] = fx;
            ty[tid * P1 + k + 1] = fy;
          } else {
            a.px[(size_t)n * a.P + k + 1] = fx;
            a.py[(size_t)n * a.P + k + 1] = fy;
          }
        }
      }
    }
    a.flags[n] = 1;
    if (a.first_hit) a.first_hit[n] = 0x7FFFFFFF;
  }

  if (a.stage) {
    __syncthreads();
    // the block's 64 rows are one contiguous [64*P] range of each plane
    const int rows = min(kRollBlock, a.n - base);
    const int total = rows * a.P;
    float *gx = a.px + (size_t)base * a.P;
    float *gy = a.py + (size_t)base * a.P;
    int s = 0, k = tid;
    while (k >= a.P) {
      k -= a.P;
      ++s;
    }
    for (int i = tid; i < total; i += kRollBlock) {
      gx[i] = tx[s * P1 + k];
      gy[i] = ty[s * P1 + k];
      k += kRollBlock;
      while (k >= a.P) {
        k -= a.P;
        ++s;
      }
    }
  }
}
#endif  // KC_TU_CYCLE

// ---------------------------------------------------------------------------
// Dilated occupancy masks (once per sensor update).  For a pose in cell c and
// an occupied cell at integer offset (i, j) the clamped distance d used by the
// exact tests obeys  res*hypot((|i|-1)+, (|j|-1)+) <= d <= res*hypot(i, j),
// so with rho = radius / res (in cells, 1e-6 of slack for the rounding of the
// pose's own cell index):
//   inner: hypot(i, j) <= rho_in - 1e-6            -> collision certain
//   outer: hypot((|i|-1)+, (|j|-1)+) <= rho_out + 1e-6 -> collision possible
// Both sets are runs per row offset j (half widths win[|j|], wout[|j|]; -1 =
// empty).  One thread per output word; a row is dilated horizontally from the
// three words around the output word (half widths <= 31).
// ---------------------------------------------------------------------------
constexpr int kMaxDil = 32;
struct DilArgs {
  const uint32_t *g;
  uint32_t *inner, *outer;
  int H, wpr, R;
  signed char win[kMaxDil + 1], wout[kMaxDil + 1];
};
// bits of the middle word within w (<= 31) of a set bit of the 96-bit string left | mid | right: the OR of the
// shifts 0..w towards higher x (high word of mid:left) and towards lower x (low word of right:mid), by doubling --
// a value that holds the shifts 0..c, OR-ed with itself shifted by s <= c + 1, holds 0..c + s
__device__ __forceinline__ uint32_t hdilate(uint32_t left, uint32_t mid, uint32_t right, int w) {
  unsigned long long up = (static_cast<unsigned long long>(mid) << 32) | left;
  unsigned long long dn = (static_cast<unsigned long long>(right) << 32) | mid;
  for (int cover = 0; cover < w;) {
    const int s = min(cover + 1, w - cover);
    up |= up << s;
    dn |= dn >> s;
    cover += s;
  }
  return static_cast<uint32_t>(up >> 32) | static_cast<uint32_t>(dn);
}
#ifdef KC_TU_SENSOR  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(256) void dilate_kernel(DilArgs a) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.H * a.wpr) return;
  const int y = t / a.wpr, w = t - y * a.wpr;
  uint32_t in_acc = 0u, out_acc = 0u;
  for (int j = -a.R; j <= a.R; ++j) {
    const int yy = y + j;
    if (yy < 0 || yy >= a.H) continue;
    const uint32_t *row = a.g + (size_t)yy * a.wpr;
    const uint32_t mid = row[w];
    const uint32_t left = w > 0 ? row[w - 1] : 0u;
    const uint32_t right = w + 1 < a.wpr ? row[w + 1] : 0u;
    if ((mid | left | right) == 0u) continue;
    const int aj = j < 0 ? -j : j;
    if (a.win[aj] >= 0) in_acc |= hdilate(left, mid, right, a.win[aj]);
    if (a.wout[aj] >= 0) out_acc |= hdilate(left, mid, right, a.wout[aj]);
  }
  a.inner[t] = in_acc;
  a.outer[t] = out_acc;
}
#endif  // KC_TU_SENSOR

// ===========================================================================
// K1 (fused): roll-out + collision gate of 32 samples per workgroup, no global
// round trip in between.  512 lanes: (A) copy the occupancy bits of the
// reachable window into LDS (word aligned with the sensor bitmap) and fetch the trig
// rows into LDS (every load in flight at once), (B) wavefront 0 runs the 64
// serial recurrences LDS -> LDS (pose k+1 replaces trig row k in place),
// (C) all lanes convert the poses to the float sample-major rows (coalesced)
// and test one pose each against the LDS bits; a hit marks the sample.
// ===========================================================================
//
// With a CycleTail the same kernel is the whole controller cycle in ONE launch
// (reference shape dwa.h:215-229): the survivors of a
// workgroup are costed right here, from the double poses still in LDS (the
// float of a pose is the one the roll-out would have stored), the workgroup's
// best key goes through an arrival ticket, and the workgroup whose ticket comes
// last reduces the keys, counts the admissible samples in front of the winner
// and hands record + winner row to the host through pinned memory.  Nothing is
// materialised unless asked for (write_paths).  The host deals the samples to
// the workgroups with a skewed stride over the trig-row order (build_perm):
// survivors cluster in a few omega rows, contiguous blocks of that order would
// leave the cost phase to a few workgroups.
// Device trig inside the fused kernel (kc_trig_exact.h).  The samples of a workgroup share a few omega rows and
// rows are non-decreasing in slot order (build_perm: row-sorted order, dealt in runs): the first slot of a
// run is its LEADER.  Phase A, round 4:
//   (0) every kernarg line is asked for at once and the arguments are read through the pointer that block hands
//       back (kernargs_touched, kc_collision_dev.h: scalar reloads at the use sites instead of SGPR spills -- the
//       kernel is VALU-issue bound inside its phases, and a spilled scalar is a vector instruction);
//   (1) the SMALL loads: sample ids, the 440-entry table sincos reads, the omega values, the value tables of the
//       x / y axes -> LDS; wavefront 0 lists the leaders; barrier;
//   (2) everybody issues its window words; then by wavefront role: the waves that own trig entries -- a lane per
//       (leader, step) -- form yaw_k by repeated addition from yaw0 (path.h:30: the additions of the steps in
//       front of k, in the lane itself), evaluate sincos and leave the increments of the run's samples (short
//       runs) or {cos, sin} in the leader's LDS pose row, from LDS alone; the waves behind them fetch the cost
//       tables of the cycle's last phase meanwhile (16-byte copies);
//   (3) the window words go to LDS; barrier.
constexpr int kTrigOmegaLds = 384;  // omega values kept in LDS (more rows: read from global memory)

struct NoTail {};
struct CycleTail {
  CostArgs c;
  DcArgs t;
  unsigned tab_off;               // byte offset of the cost tables in dynamic LDS (16-aligned)
  int write_paths;                // also store the float rows (debugging samples)
  int team_max;                   // survivors of a workgroup up to which they are costed by teams (<= kTeamMaxSurvivors)
  long long *block_keys;          // [grid][2] best key + row check word per workgroup (sc1 stores / loads)
  uint32_t *adm_bits;             // [n / 32 + 1] admissible samples by local id (agent-scope atomic OR;
                                  // zero at launch, cleared again by the last workgroup)
  long long *result;              // device record (R_* / W_* slots)
  long long *host_pub;            // pinned: {key, n_adm << 32 | compact, seq, check, row word}
  uint32_t *host_rows;            // pinned: [grid][2 P] best row of every workgroup (x | y float bits)
  long long seq;                  // sequence number of this cycle's record
  // Single GPU: no device-side reduction at all.  Every workgroup leaves a 32-byte
  // slot {key, survivor mask | row check << 32, seq | error << 62, checksum} in
  // pinned host memory (posted writes) and ends; the HOST waits for the G slots of
  // this sequence number and reduces them (min key, popcounts, compacted index
  // from the dealt order it built itself).  No drain, no ticket, no round trip on
  // the device.  Null: the ticket epilogue (device record for the all-reduce).
  long long *host_slots;          // pinned: [grid][4]
  // kc_dwa_cycle_sharded: the last workgroup writes this rank's part of the exchange record (kc_shard.h: best
  // key with the GLOBAL raw index, error word, admissible bitmap by share-local id) -- what xchg_pack_kernel
  // does behind the three-kernel cycle.  Null: no exchange.
  long long *xs;
  const int32_t *xgid;            // share-local lattice id -> global id (null: identity)
  int xrank, xrw;
};

template <int kFusedSamples, int kFusedBlock, class Tail = NoTail>
__global__ __launch_bounds__(kFusedBlock) void rollout_collide_kernel(RollArgs a_, Tail tail_) {
  constexpr bool kCycle = !std::is_same<Tail, NoTail>::value;
  // (the arguments as read behind the touch of every kernarg line: see kernargs_touched)
  const KernargPair<RollArgs, Tail> *ka_ = kernargs_touched<KernargPair<RollArgs, Tail>>();
  const RollArgs &a = ka_->a;
  const Tail &tail = kCycle ? ka_->b : tail_;  // (NoTail is empty: nothing to read)
  extern __shared__ __align__(16) unsigned char smem[];
  const int PP = a.P | 1;  // pitch of a sample's row in 16-byte slots
  double2 *lpos = reinterpret_cast<double2 *>(smem);
  uint32_t *lbits = reinterpret_cast<uint32_t *>(
      smem + (size_t)kFusedSamples * PP * sizeof(double2));
  const int nwin = a.c.enabled ? a.c.H * a.c.wpr : 0;
  uint32_t *linner = lbits + nwin;                 // a.c.dil only
  uint32_t *louter = linner + (a.c.dil ? nwin : 0);
  int *lcand = reinterpret_cast<int *>(louter + (a.c.dil ? nwin : 0));  // [samples * P]
  __shared__ int ncand, ncand2;
  __shared__ int lhit[kFusedSamples];
  __shared__ int lperm[kFusedSamples];  // local sample id of slot s
  __shared__ int lrow[kFusedSamples];   // its trig row
  __shared__ uint32_t lvi[kFusedSamples];  // its value indices (vx | vy << 16)

  const int tid = threadIdx.x;
  const int base = blockIdx.x * kFusedSamples;
  const int rows = min(kFusedSamples, a.n - base);
  const int steps = a.P - 1;

  KC_RSTAMP(0);
  // ---- A: window bits, cost tables, trig rows ------------------------------------------------
  // A table that is there already (formed in the sensor update's launch, or -- fallback -- by the host): one round
  // of loads (window, cost tables), the sample ids used behind the copies.
  // Device trig: the SMALL loads first (sample ids, the table sincos reads, omega values), a barrier, then the
  // bulk loads are issued and the trig entries are formed from LDS while they are in flight.
  int my_id = 0, my_row = 0;
  uint32_t my_vi = 0u;
  if (tid < rows) {
    my_id = a.perm[base + tid];
    my_row = a.prow[base + tid];
    my_vi = a.pvi[base + tid];
  }
  __shared__ int lfirst[kFusedSamples];  // device trig: slot -> its leader (the first slot with the same trig row)
  __shared__ int llead[kFusedSamples];   // ... the leaders, ascending
  __shared__ int nlead;
  __shared__ int lrun[kFusedSamples];    // ... samples in the run of leader l
  __shared__ int runs_short;             // ... every run has at most kRunMax samples: the entry lanes write the increments
  constexpr int kRunMax = 4;
  __shared__ double ltab[440];           // ... sin / cos (k / 128): what sincos reads, as FOUR ROWS of 110 (sn | ssn |
                                         // cs | ccs): the lanes of a wavefront look up different k -- entries of four
                                         // doubles side by side put every lane on one of 8 bank groups (LDS conflict
                                         // share of the kernel 0.11 -> 0.29 when the table came into LDS, r3), rows of
                                         // 110 doubles spread them over 32
  __shared__ double lom[kTrigOmegaLds];  // ... omega of the trig rows
  constexpr int kVxLds = 128, kVyLds = 64;
  __shared__ double lvx[kVxLds], lvy[kVyLds];  // ... the value tables of the x / y axes (the head of longer ones)
  const bool box = a.c.enabled && a.c.shape == KC_BOX;
  constexpr int kTabPer = (440 + kFusedBlock - 1) / kFusedBlock, kOmPer = (kTrigOmegaLds + kFusedBlock - 1) / kFusedBlock;
  // the per-slot words of the workgroup (+ device trig: the leaders of the row runs)
  auto slot_words = [&]() {
    if (a.trig_dev && tid < 64) {
      // leaders: a slot whose row differs from the slot in front of it (rows come in runs)
      const int prev = __shfl_up(my_row, 1, 64);
      const bool lead = tid < rows && (tid == 0 || prev != my_row);
      const unsigned long long bal = __ballot(lead);
      const unsigned long long upto = bal & (~0ull >> (63 - tid));  // leaders at or in front of this slot
      if (tid < rows) lfirst[tid] = 63 - __clzll(static_cast<long long>(upto));
      const unsigned long long above = bal & ~(~0ull >> (63 - tid));  // leaders behind this slot
      const int run = (above ? __ffsll(static_cast<long long>(above)) - 1 : rows) - tid;
      if (lead) {
        llead[__popcll(upto) - 1] = tid;
        lrun[__popcll(upto) - 1] = run;
      }
      const unsigned long long longrun = __ballot(lead && run > kRunMax);
      if (tid == 0) {
        nlead = __popcll(bal);
        runs_short = longrun == 0ull ? 1 : 0;
      }
    }
    if (tid < kFusedSamples) {
      lhit[tid] = a.freeze ? 0x7FFFFFFF : 0;  // freeze mode: the FIRST colliding pose index of the sample (minimum)
      lperm[tid] = my_id;
      lrow[tid] = my_row;
      lvi[tid] = my_vi;
      if constexpr (kCycle) lpos[tid * PP + PP - 1] = make_double2(a.x0, a.y0);  // spare slot of the row: pose 0
    }
    if (tid == 0) {
      ncand = 0;
      ncand2 = 0;
    }
  };
  if (a.trig_dev) {
    double tabv[kTabPer], omv[kOmPer], vxv = 0.0, vyv = 0.0;
#pragma unroll
    for (int u = 0; u < kTabPer; ++u) {
      const int j = tid + u * kFusedBlock;
      tabv[u] = j < 440 ? a.sincostab[j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kOmPer; ++u) {
      const int j = tid + u * kFusedBlock;
      omv[u] = (j < kTrigOmegaLds && j < a.A) ? a.omega_values[j] : 0.0;
    }
    // (the value tables of the axes from the far end of the workgroup: the lanes in front hold sample ids)
    const int jv = kFusedBlock - 1 - tid;
    if (jv < kVxLds && jv < a.nvx) vxv = a.vxt[jv];
    if (jv >= kVxLds && jv < kVxLds + kVyLds && jv - kVxLds < a.nvy) vyv = a.vyt[jv - kVxLds];
#pragma unroll
    for (int u = 0; u < kTabPer; ++u) {
      const int j = tid + u * kFusedBlock;
      if (j < 440) ltab[(j & 3) * 110 + (j >> 2)] = tabv[u];
    }
#pragma unroll
    for (int u = 0; u < kOmPer; ++u) {
      const int j = tid + u * kFusedBlock;
      if (j < kTrigOmegaLds) lom[j] = omv[u];
    }
    if (jv < kVxLds) lvx[jv] = vxv;
    else if (jv < kVxLds + kVyLds) lvy[jv - kVxLds] = vyv;
    slot_words();
    __syncthreads();
  }
  KC_RSTAMP(16);
  // Behind the barrier the wavefronts split the rest of the phase (device trig): the ones that own trig entries
  // -- a lane per (leader, step), the first waves of the workgroup -- form them from LDS alone; the waves behind
  // them fetch the cost tables of the cycle's last phase meanwhile (loads, the wait, the LDS stores).  Everybody
  // brings in the window words (issued first: vector loads return in order, and the window is what phase C
  // needs).  Without device trig every thread copies.
  const int L = a.trig_dev ? nlead : 0;
  int sh = 0;
  while ((1 << sh) < L) ++sh;
  const int ktop = steps + (box ? 1 : 0);  // (boxes: yaw of the last pose too, for the exact tests)
  const int nentry = a.trig_dev ? (ktop << sh) : 0;
  // first thread of the table copy: behind the entry lanes, but at least a quarter of the workgroup copies
  const int tab_t0 = min((nentry + 63) & ~63, kFusedBlock - kFusedBlock / 4);
  const bool win = a.c.enabled != 0;
  const int nwords = a.c.enabled ? a.c.H * a.c.wpr : 0;
  const int w0 = a.c.enabled ? (a.c.kx0 - a.c.gkx0) >> 5 : 0;  // exact: difference is a multiple of 32
  // window origin is word aligned with the sensor bitmap: whole-word copies (up to three
  // words of each mask per thread held in registers: all loads first, then the stores)
  auto win_load = [&](int i0, uint32_t(&v)[3], uint32_t(&vi)[3], uint32_t(&vo)[3]) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = i0 + tid + u * kFusedBlock;
      v[u] = vi[u] = vo[u] = 0u;
      if (i < nwords) {
        // i / wpr by the host's reciprocal (CollDev::wpr_magic)
        const int cy = a.c.wpr > 1 ? static_cast<int>(__umulhi(static_cast<uint32_t>(i), a.c.wpr_magic)) : i;
        const int w = i - cy * a.c.wpr;
        const int gy = a.c.ky0 + cy - a.c.gky0, gw = w0 + w;
        if (gy >= 0 && gy < a.c.gH && gw >= 0 && gw < a.c.gwpr) {
          const size_t g = (size_t)gy * a.c.gwpr + gw;
          v[u] = a.c.gbits[g];
          if (a.c.dil) {
            vi[u] = a.c.ginner[g];
            vo[u] = a.c.gouter[g];
          }
        }
      }
    }
  };
  auto win_store = [&](int i0, const uint32_t(&v)[3], const uint32_t(&vi)[3], const uint32_t(&vo)[3]) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = i0 + tid + u * kFusedBlock;
      if (i < nwords) {
        lbits[i] = v[u];
        if (a.c.dil) {
          linner[i] = vi[u];
          louter[i] = vo[u];
        }
      }
    }
  };
  uint32_t wv[3], wvi[3], wvo[3];
  if (win) win_load(0, wv, wvi, wvo);
  KC_RSTAMP(17);
  if constexpr (kCycle)
    if (tid >= tab_t0) {
      CycleTabRegs<kFusedBlock> tabregs;
      cycle_tables_load<kFusedBlock>(tail, tid - tab_t0, kFusedBlock - tab_t0, tabregs);
      cycle_tables_store<kFusedBlock>(tail, smem, tid - tab_t0, kFusedBlock - tab_t0, tabregs);
    }
  KC_RSTAMP(20);
  // device trig: the sample's velocity (for the increments) and {cos, sin}(yaw_k) of every distinct row into
  // the LDS pose row of its leader, a lane per entry -- LDS and VALU only
  const int s = tid & (kFusedSamples - 1);
  const bool mine = s < rows;
  double vx = 0.0, vy = 0.0;
  const bool direct_inc = a.trig_dev && runs_short != 0;  // (read behind the early barrier; uniform)
  // (the LDS copy is read unconditionally and a value beyond it replaces the result: `c ? lds[j] : global[j]` would
  // select between the two ADDRESSES and load once through a flat pointer -- a load that counts as a vector load and
  // waits, in order, behind every window word still in flight)
  auto axis_vx = [&](uint32_t vi) {
    const int j = static_cast<int>(vi & 0xFFFFu);
    double v = lvx[j & (kVxLds - 1)];
    if (j >= kVxLds) v = *static_cast<const volatile double *>(a.vxt + j);  // (volatile: never merged with the LDS read)
    return v;
  };
  auto axis_vy = [&](uint32_t vi) {
    const int j = static_cast<int>(vi >> 16);
    double v = lvy[j & (kVyLds - 1)];
    if (j >= kVyLds) v = *static_cast<const volatile double *>(a.vyt + j);
    return v;
  };
  if (a.trig_dev) {
    if (mine && !direct_inc) {
      const uint32_t vi = lvi[s];
      vx = axis_vx(vi);
      vy = axis_vy(vi);
    }
    for (int i = tid; i < nentry; i += kFusedBlock) {
      const int l = i & ((1 << sh) - 1), k = i >> sh;
      if (l >= L) continue;
      const int sl = llead[l], r = lrow[sl];
      // short runs (the dealt order of the cycle: four samples per row): this lane also forms the increments of
      // the run's samples from its entry
      // (every read unconditional -- slots beyond the run are read and not used: four LDS reads in flight, then
      // eight, instead of a wait per sample)
      const int run = direct_inc ? lrun[l] : 0;
      uint32_t rvi[kRunMax];
#pragma unroll
      for (int u = 0; u < kRunMax; ++u) rvi[u] = lvi[min(sl + u, kFusedSamples - 1)];
      double rvx[kRunMax], rvy[kRunMax];
#pragma unroll
      for (int u = 0; u < kRunMax; ++u) {
        rvx[u] = axis_vx(rvi[u]);
        rvy[u] = axis_vy(rvi[u]);
      }
      double om = lom[min(r, kTrigOmegaLds - 1)];
      if (r >= kTrigOmegaLds) om = *static_cast<const volatile double *>(a.omega_values + r);
      const double w = om * a.dt;
      double yaw = a.yaw0;
      int q = 0;
      for (; q + 4 <= k; q += 4) yaw = (((yaw + w) + w) + w) + w;
      for (; q < k; ++q) yaw += w;
      double sn, cs;
      trig::sincos_exact(yaw, &sn, &cs, trig::TabRows{ltab});
      if (k < steps) {
        if (direct_inc) {
          //   x += (vx*cos - vy*sin) * dt;  y += (vx*sin + vy*cos) * dt   (datatypes/path.h:24-30)
#pragma unroll
          for (int u = 0; u < kRunMax; ++u)
            if (u < run) {
              const double tx = rvx[u] * cs - rvy[u] * sn;
              const double ty = rvx[u] * sn + rvy[u] * cs;
              lpos[(sl + u) * PP + k] = make_double2(tx * a.dt, ty * a.dt);
            }
        } else {
          lpos[sl * PP + k] = make_double2(cs, sn);
        }
      }
      if (box) a.trig_out[(size_t)k * a.A + r] = make_double2(cs, sn);
    }
  }
  KC_RSTAMP(18);
  if (win) {
    win_store(0, wv, wvi, wvo);
    for (int i0 = 3 * kFusedBlock; i0 < nwords; i0 += 3 * kFusedBlock) {
      win_load(i0, wv, wvi, wvo);
      win_store(i0, wv, wvi, wvo);
    }
  }
  KC_RSTAMP(19);
  if (!a.trig_dev) slot_words();
  KC_RSTAMP(1);
  if (direct_inc) {
    // (the entry lanes have written the increments: nothing to do in front of the serial sums)
    KC_RSTAMP(2);
  } else if (a.trig_dev) {
    // every sample forms its increments from its leader's entries -- followers first, the leaders in place
    // behind a barrier
    __syncthreads();
    KC_RSTAMP(2);
    const int f = mine ? lfirst[s] : 0;
    //   x += (vx*cos - vy*sin) * dt;  y += (vx*sin + vy*cos) * dt   (datatypes/path.h:24-30)
    if (mine && f != s)
      for (int k = tid / kFusedSamples; k < steps; k += kFusedBlock / kFusedSamples) {
        const double2 cs = lpos[f * PP + k];
        const double tx = vx * cs.x - vy * cs.y;
        const double ty = vx * cs.y + vy * cs.x;
        lpos[s * PP + k] = make_double2(tx * a.dt, ty * a.dt);
      }
    __syncthreads();
    if (mine && f == s)
      for (int k = tid / kFusedSamples; k < steps; k += kFusedBlock / kFusedSamples) {
        const double2 cs = lpos[s * PP + k];
        const double tx = vx * cs.x - vy * cs.y;
        const double ty = vx * cs.y + vy * cs.x;
        lpos[s * PP + k] = make_double2(tx * a.dt, ty * a.dt);
      }
  } else {
    __syncthreads();  // lrow, lperm
    const int s = tid & (kFusedSamples - 1);
    if (s < rows) {
      const int r = lrow[s];
      const uint32_t vi = lvi[s];
      const double vx = a.vxt[vi & 0xFFFFu], vy = a.vyt[vi >> 16];
      for (int k = tid / kFusedSamples; k < steps; k += kFusedBlock / kFusedSamples) {
        const double2 cs = a.trig[(size_t)k * a.A + r];
        const double tx = vx * cs.x - vy * cs.y;
        const double ty = vx * cs.y + vy * cs.x;
        lpos[s * PP + k] = make_double2(tx * a.dt, ty * a.dt);
      }
    }
  }
  __syncthreads();
  KC_RSTAMP(3);
  KC_RSTAMP(7);
  // ---- B: the serial sums ------------------------------------------------------
  // (the x sums of the samples by wavefront 0, the y sums by wavefront 1: two independent chains per sample,
  // half the LDS traffic and half the additions per wavefront)
  if (tid < 128 && (tid & 63) < rows) {
    // sixteen increments per register chunk: one LDS latency per chunk instead
    // of one per step
    constexpr int kChunk = 16;
    const int comp = tid >> 6;
    double acc = comp ? a.y0 : a.x0;
    double *mine = reinterpret_cast<double *>(lpos + (tid & 63) * PP) + comp;  // (.x or .y of the row's entries)
    int k = 0;
    for (; k + kChunk <= steps; k += kChunk) {
      double v[kChunk];
#pragma unroll
      for (int j = 0; j < kChunk; ++j) v[j] = mine[2 * (k + j)];
#pragma unroll
      for (int j = 0; j < kChunk; ++j) {
        acc += v[j];
        v[j] = acc;  // pose k + j + 1
      }
#pragma unroll
      for (int j = 0; j < kChunk; ++j) mine[2 * (k + j)] = v[j];
    }
    for (; k < steps; ++k) {
      acc += mine[2 * k];
      mine[2 * k] = acc;
    }
  }
  __syncthreads();
  KC_RSTAMP(4);
  // ---- C: float rows out; poses classified with the dilated masks, the
  // undecided ones queued and tested exactly by densely packed lanes ----------
  {
    bool store_row = true;
    if constexpr (kCycle) store_row = tail.write_paths != 0;
    // one pose: float row out (when rows are kept), cell of the dilated masks, queue
    auto classify = [&](int s, int k) {
      double2 p;
      if (k == 0) p = make_double2(a.x0, a.y0);
      else p = lpos[s * PP + k - 1];
      if (store_row) {
        const size_t o = (size_t)lperm[s] * a.P + k;  // sample-major rows
        a.px[o] = static_cast<float>(p.x);
        a.py[o] = static_cast<float>(p.y);
      }
      bool exact = a.c.enabled && k > 0;
      if (exact && a.c.dil) {
        const double dx = p.x - a.c.tx, dy = p.y - a.c.ty;
        const double xf = a.c.r00 * dx + a.c.r10 * dy;
        const double yf = a.c.r01 * dx + a.c.r11 * dy;
        const int cx = static_cast<int>(floor(xf * a.c.inv)) - a.c.kx0;
        const int cy = static_cast<int>(floor(yf * a.c.inv)) - a.c.ky0;
        if ((a.c.cover & 0xFF) > 1) {
          // a long box: circles along its long axis (CollDev::cover) -- any inner hit decides, all outer misses clear
          const size_t e = (size_t)k * a.A + lrow[s];  // yaw_k
          double cw, sw;
          if (a.trig_dev) {
            const double *tg = reinterpret_cast<const double *>(a.trig);
            cw = __hip_atomic_load(tg + 2 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sw = __hip_atomic_load(tg + 2 * e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            cw = a.trig[e].x;
            sw = a.trig[e].y;
          }
          double ux = a.c.r00 * cw + a.c.r10 * sw, uy = a.c.r01 * cw + a.c.r11 * sw;  // the box's x axis, octree frame
          if (a.c.cover >> 8) {
            const double t = ux;
            ux = -uy;
            uy = t;
          }
          const int nc = a.c.cover & 0xFF;
          const double step = 2.0 * fmax(a.c.a, a.c.b) / static_cast<double>(nc);
          bool in = false, out = true;
          for (int i = 0; i < nc; ++i) {
            const double o = (static_cast<double>(i) - 0.5 * static_cast<double>(nc - 1)) * step;
            const int qx = static_cast<int>(floor((xf + o * ux) * a.c.inv)) - a.c.kx0;
            const int qy = static_cast<int>(floor((yf + o * uy) * a.c.inv)) - a.c.ky0;
            if (qx >= 0 && qx < a.c.W && qy >= 0 && qy < a.c.H) {
              const int w = qy * a.c.wpr + (qx >> 5);
              const uint32_t bit = 1u << (qx & 31);
              in = in || (linner[w] & bit) != 0u;
              out = out && (louter[w] & bit) == 0u;
            } else {
              out = false;  // (outside the window: nothing known, the exact test decides)
            }
          }
          if (in) {
            if (a.freeze) atomicMin(&lhit[s], k);
            else lhit[s] = 1;
            exact = false;
          } else if (out) {
            exact = false;
          }
        } else if (cx >= 0 && cx < a.c.W && cy >= 0 && cy < a.c.H) {
          const int w = cy * a.c.wpr + (cx >> 5);
          const uint32_t bit = 1u << (cx & 31);
          if (linner[w] & bit) {
            if (a.freeze) atomicMin(&lhit[s], k);
            else lhit[s] = 1;  // every writer stores the same value
            exact = false;
          } else if (!(louter[w] & bit)) {
            exact = false;
          }
        }
      }
      // queue slots: one LDS atomic per wavefront (hundreds of undecided poses per workgroup in clutter)
      const unsigned long long bal = __ballot(exact);
      if (bal) {
        const int lane = tid & 63;
        int base = 0;
        if (lane == __ffsll(static_cast<long long>(bal)) - 1) base = atomicAdd(&ncand, __popcll(bal));
        base = __shfl(base, __ffsll(static_cast<long long>(bal)) - 1, 64);
        if (exact) lcand[base + __popcll(bal & ((1ull << lane) - 1ull))] = (s << 16) | k;
      }
    };
    if (!store_row) {
      // sample = low bits of the thread id, poses strided: no division, and the LDS reads of a
      // wavefront go to different samples' rows (odd pitch: conflict free)
      const int s = tid % kFusedSamples;
      if (s < rows)
        for (int k = tid / kFusedSamples; k < a.P; k += kFusedBlock / kFusedSamples) classify(s, k);
    } else {
      // rows are written: consecutive threads take consecutive poses of a sample (coalesced stores)
      const int total = rows * a.P;
      const float inv_p = 1.0f / static_cast<float>(a.P);
      for (int i = tid; i < total; i += kFusedBlock) {
        int s = min(static_cast<int>(static_cast<float>(i) * inv_p), rows - 1);
        int k = i - s * a.P;
        if (k < 0) {
          --s;
          k += a.P;
        } else if (k >= a.P) {
          ++s;
          k -= a.P;
        }
        classify(s, k);
      }
    }
  }
  __syncthreads();
  KC_RSTAMP(13);
  {
    int nc = ncand;
    constexpr int kMine = 8;  // queue entries a thread holds across the barrier
    if (nc * 8 > kFusedBlock && nc <= kMine * kFusedBlock) {
      // A long queue is mostly poses of samples that an inner-mask hit has decided meanwhile
      // (dense clutter): squeeze those out first, so that the live ones get several lanes each.
      int mine[kMine];
      int cnt = 0;
      for (int i = tid; i < nc; i += kFusedBlock) mine[cnt++] = lcand[i];
      __syncthreads();
      for (int q = 0; q < cnt; ++q) {
        // (freeze mode: a pose behind a known hit of its sample cannot be the first one)
        const bool keep = a.freeze ? lhit[mine[q] >> 16] > (mine[q] & 0xFFFF) : !lhit[mine[q] >> 16];
        const unsigned long long bal = __ballot(keep);
        if (bal) {
          const int lane = tid & 63, lead = __ffsll(static_cast<long long>(bal)) - 1;
          int base = 0;
          if (lane == lead) base = atomicAdd(&ncand2, __popcll(bal));
          base = __shfl(base, lead, 64);
          if (keep) lcand[base + __popcll(bal & ((1ull << lane) - 1ull))] = mine[q];
        }
      }
      __syncthreads();
      nc = ncand2;
    }
#ifdef KC_PHASE_STAMPS
    if (a.dbg && tid == 0 && blockIdx.x < 512) a.dbg[(size_t)blockIdx.x * 32 + 15] = static_cast<unsigned long long>(nc);
#endif
    // 2, 4 or 8 lanes per undecided pose when the queue is short enough for that (the rows of the
    // voxel window go round the lanes: each exact test is a chain of dependent LDS reads and f64
    // arithmetic per row, and any lane's hit decides the sample)
    int lanes_per = 1;
    while (lanes_per < 8 && nc * lanes_per * 2 <= kFusedBlock) lanes_per *= 2;
    const int sub = tid & (lanes_per - 1);
    for (int i = tid / lanes_per; i < nc; i += kFusedBlock / lanes_per) {
      const int s = lcand[i] >> 16, k = lcand[i] & 0xFFFF;
      if (a.freeze ? lhit[s] <= k : lhit[s] != 0) continue;  // already decided (stale reads only cost work)
      const double2 p = lpos[s * PP + k - 1];
      bool hit;
      if (a.c.shape == KC_BOX) {
        const size_t e = (size_t)k * a.A + lrow[s];  // yaw_k
        double2 t;
        if (a.trig_dev) {
          // written by lanes of this workgroup in the trig phase (barriers in between); read at L2
          const double *tg = reinterpret_cast<const double *>(a.trig);
          t.x = __hip_atomic_load(tg + 2 * e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          t.y = __hip_atomic_load(tg + 2 * e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          t = a.trig[e];
        }
        hit = hit_box(a.c, lbits, p.x, p.y, t.x, t.y, sub, lanes_per);
      } else {
        hit = hit_round(a.c, lbits, p.x, p.y, sub, lanes_per);
      }
      if (hit) {
        if (a.freeze) atomicMin(&lhit[s], k);
        else lhit[s] = 1;
      }
    }
  }
  __syncthreads();
  KC_RSTAMP(5);
  __shared__ int lsurv[kFusedSamples];  // cycle: slots of the survivors, ascending
  __shared__ int nsurv;
  __shared__ unsigned long long lmask;  // ... as a mask over the slots
  __shared__ int lfrz[kFusedSamples];   // freeze mode: first zero-velocity step of a frozen sample, else 0
  if (tid < 64) {  // wavefront 0: publish the flags, append the survivors
    bool ok = tid < rows && !lhit[tid < kFusedSamples ? tid : 0];
    if (a.freeze) {
      // trajectory_sampler.cpp:147-168: collision at loop step i = k - 1 -> last_free_index = i - 1 (i > 0);
      // kept when last_free_index > numCtrlPoints_ (and < P - 1, which it always is)
      const int kc = lhit[tid < kFusedSamples ? tid : 0];
      int fstep = 0;
      if (kc != 0x7FFFFFFF) {
        const int i = kc - 1;
        if (i >= 1 && (i - 1) > a.num_ctrl) fstep = i;
      }
      ok = tid < rows && (kc == 0x7FFFFFFF || fstep > 0);
      if (tid < kFusedSamples) lfrz[tid] = tid < rows ? fstep : 0;
      if (tid < rows) {
        const int id = lperm[tid];
        float fs = 0.0f, fj = 0.0f;
        if (fstep > 0)
          frozen_velocity_sums(static_cast<float>(a.vxt[lvi[tid] & 0xFFFFu]), static_cast<float>(a.vyt[lvi[tid] >> 16]),
                               static_cast<float>(a.omega_values[lrow[tid]]), fstep, a.P - 1, a.acc0, a.acc1, a.acc2,
                               &fs, &fj);
        a.freeze_step[id] = fstep;
        a.frz_smooth[id] = fs;
        a.frz_jerk[id] = fj;
      }
    }
    if (tid < rows) a.flags[lperm[tid]] = ok ? 1 : 0;
    const unsigned long long bal = __ballot(ok);
    const int cnt = __popcll(bal);
    if constexpr (kCycle) {
      if (ok) lsurv[__popcll(bal & ((1ull << tid) - 1ull))] = tid;
      if (tid == 0) {
        nsurv = cnt;
        lmask = bal;
      }
    } else {
      int start = 0;
      if (tid == 0 && cnt)
        start = static_cast<int>(atomicAdd(
            reinterpret_cast<unsigned long long *>(a.adm_count),
            static_cast<unsigned long long>(cnt)));
      start = __shfl(start, 0, 64);
      if (ok) a.adm_list[start + __popcll(bal & ((1ull << tid) - 1ull))] = lperm[tid];
    }
  }
  KC_RSTAMP(6);
  if (a.freeze) {
    // frozen samples: points i + 1 .. P - 1 repeat point i - 1 -- in the LDS rows the cost phase reads and in
    // the float rows (the sums above went to global memory: visible to this workgroup behind the barrier)
    __syncthreads();
    bool store_row = true;
    if constexpr (kCycle) store_row = tail.write_paths != 0;
    const int s = tid % kFusedSamples;
    const int i = s < rows ? lfrz[s] : 0;
    if (i > 0) {
      const double2 fp = (i - 1 == 0) ? make_double2(a.x0, a.y0) : lpos[s * PP + i - 2];
      for (int k = i + 1 + tid / kFusedSamples; k < a.P; k += kFusedBlock / kFusedSamples) {
        lpos[s * PP + k - 1] = fp;
        if (store_row) {
          const size_t o = (size_t)lperm[s] * a.P + k;
          a.px[o] = static_cast<float>(fp.x);
          a.py[o] = static_cast<float>(fp.y);
        }
      }
    }
  }
  if constexpr (kCycle) {
    __syncthreads();
    int best_slot = -1;
    const long long key = cycle_costs<kFusedSamples, kFusedBlock>(a, tail, smem, lpos, PP, lperm, lsurv,
                                                                  nsurv, tid, &best_slot);
    KC_RSTAMP(8);
    if (tail.host_slots)
      cycle_epilogue_host<kFusedBlock>(a, tail, key, lmask, best_slot,
                                       lpos + (best_slot < 0 ? 0 : best_slot) * PP, tid);
    else
      cycle_epilogue<kFusedBlock>(a, tail, key, nsurv, best_slot, lpos + (best_slot < 0 ? 0 : best_slot) * PP,
                                  lperm, lsurv, tid);
    KC_RSTAMP(9);
  }
}

// ===========================================================================
// K1b: collision gate.  A sample is dropped as soon as ANY of its poses
// collides (trajectory_sampler.cpp:147-152 with drop_samples_ == true), so the
// (step, sample) pairs are independent: one lane per pose, occupancy bits of
// the reachable window staged in LDS, a hit clears the sample's flag (every
// writer stores the same 0).
// ===========================================================================
constexpr int kCollBlock = 256;

#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(kCollBlock) void collision_kernel(RollArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint32_t *lbits = reinterpret_cast<uint32_t *>(smem);
  if (a.c.lds) {
    const int nwords = a.c.H * a.c.wpr;
    for (int i = threadIdx.x; i < nwords; i += kCollBlock) lbits[i] = a.c.bits[i];
    __syncthreads();
  }
  const long t = (long)blockIdx.x * kCollBlock + threadIdx.x;
  if (t >= (long)a.n * (a.P - 1)) return;
  const int k = static_cast<int>(t / a.n) + 1;  // pose index 1..P-1
  const int n = static_cast<int>(t - (long)(k - 1) * a.n);
  const double2 p = a.pos[(size_t)k * a.n + n];
  bool hit;
  if (a.c.shape == KC_BOX) {
    const double2 cs = a.trig[(size_t)k * a.A + a.row[a.first + n]];  // yaw_k
    hit = a.c.lds ? hit_box(a.c, lbits, p.x, p.y, cs.x, cs.y)
                  : hit_box(a.c, a.c.bits, p.x, p.y, cs.x, cs.y);
  } else {
    hit = a.c.lds ? hit_round(a.c, lbits, p.x, p.y)
                  : hit_round(a.c, a.c.bits, p.x, p.y);
  }
  if (hit) {
    a.flags[n] = 0;
    if (a.first_hit) atomicMin(&a.first_hit[n], k);  // drop_samples = false: which pose collides FIRST decides
  }
}
#endif  // KC_TU_CYCLE

// split path, drop_samples = false: from the first colliding pose of every sample decide what
// trajectory_sampler.cpp:157-168 decides, freeze the float rows of the kept samples (the row already holds the
// float of the point it repeats) and leave the frozen profile's smoothness / jerk sums.  One lane per sample.
#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(256) void freeze_fixup_kernel(RollArgs a) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= a.n) return;
  const int kc = a.first_hit[n];
  int fstep = 0;
  float fs = 0.0f, fj = 0.0f;
  if (kc != 0x7FFFFFFF) {
    const int i = kc - 1;  // loop step of the collision: last_free_index = i - 1 (when i > 0)
    if (i >= 1 && (i - 1) > a.num_ctrl) {
      fstep = i;
      float *rx = a.px + (size_t)n * a.P, *ry = a.py + (size_t)n * a.P;
      const float lx = rx[i - 1], ly = ry[i - 1];
      for (int k = i + 1; k < a.P; ++k) {
        rx[k] = lx;
        ry[k] = ly;
      }
      frozen_velocity_sums(static_cast<float>(sample_vx(a, n)), static_cast<float>(sample_vy(a, n)),
                           static_cast<float>(a.omega_values[a.row[a.first + n]]), i, a.P - 1, a.acc0, a.acc1, a.acc2,
                           &fs, &fj);
      a.flags[n] = 1;
    }
  }
  a.freeze_step[n] = fstep;
  a.frz_smooth[n] = fs;
  a.frz_jerk[n] = fj;
}
#endif  // KC_TU_CYCLE

// batch pose check (CollisionChecker::checkCollisions for arbitrary poses):
// occupancy bits read from global memory, cos/sin(yaw) from the host table
#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ void pose_check_kernel(CollDev c, const double2 *__restrict__ pos,
                                  const double2 *__restrict__ cs, int n,
                                  uint8_t *__restrict__ hit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double2 p = pos[i];
  bool h;
  if (c.shape == KC_BOX) {
    const double2 t = cs[i];
    h = hit_box(c, c.bits, p.x, p.y, t.x, t.y);
  } else {
    h = hit_round(c, c.bits, p.x, p.y);
  }
  hit[i] = h ? 1 : 0;
}
#endif  // KC_TU_CYCLE

}  // namespace kc
