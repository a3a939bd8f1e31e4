// Cost kernels (A5-A10): both sample-cost kernels, the publish kernel and the
// small list / count helpers.  Part of kc_dwa.hip.
#pragma once

namespace kc {

// ===========================================================================
// K2: cost of every admissible sample + per-workgroup argmin.  One WAVEFRONT
// per sample (sixteen samples in flight per workgroup), one lane per
// trajectory point; the search tables sit in LDS, filled once per workgroup.
//
// Tracked segment (pathCostFunc inner loops, cost_evaluator.cpp:120-130, and
// goalCostFunc's closest-point search, :157-166): d2 = dx*dx + (dy*dy + dz*dz)
// in float, the lowest index winning ties (the reference's strict `<` in index
// order); min_j sqrt(d2_j) == sqrt(min_j d2_j) for the correctly rounded sqrt,
// so one sqrt per point; the END point's search also yields the goal cost
// ((a-b)^2 == (b-a)^2 bit for bit).  Instead of all S points a lane evaluates
// (1) the first point of every super-chunk (8 chunks of 16 points), (2) a
// bounding-sphere test per super-chunk against that upper bound
// (|q - p_j| >= |q - c| - r), (3) the first points of the chunks of the
// surviving super-chunks, (4) a capsule test per such chunk
// (|q - p_j| >= dist(q, chord AB) - max_j dist(p_j, AB): on a smooth path the
// band of chunks that can hold the minimum is one or two wide, where a sphere
// leaves sqrt(2 d r) of path), and (5) every remaining point of the chunks
// that may hold something at least as close -- the same minimum and the same
// lowest index as the full scan.  Bounds come from the host (computed around
// the stored float values, rounded up, 1e-4 relative slack in the tests;
// non-finite chunks always qualify).
//
// Obstacles (TrajectoryPath::minDist2D, trajectory.h:218-235): float
// difference, squares and sum in double, rounded to float once; the rounding
// is monotonic, so the minimum is taken in double and rounded later.  Instead
// of the reference's brute force over all O obstacles the points are bucketed
// on a uniform grid (host, once per sensor update) with a Chebyshev distance
// table to the nearest non-empty cell.  A lane searches square rings of cells
// outwards from the first ring that can hold a point.  A block of half-width m
// contains every obstacle closer than m*g, so once the best squared distance
// is below (m*g)^2 (1e-6 relative guard, four orders above the float rounding
// of the differences) nothing outside can beat it.  Only the minimum over the
// whole trajectory is used (obstaclesDistCostFunc, cost_evaluator.cpp:179-184),
// so the lanes of a sample share their best distance: a lane whose unvisited
// cells are all farther than what another lane already found stops, and so
// does one that has covered max_obstacles_dist (those distances cost 0).
//
// The weighted total follows the reference's accumulation order
// (cost_evaluator.cpp:59-100: float total, each += a double multiply-add
// rounded once; the path-cost sum walks the points in order with
// v_readlane).  Every workgroup leaves its best (cost, index) key for
// publish_kernel.
// ===========================================================================
constexpr int kCostBlock = 1024;  // 16 wavefronts = 16 samples in flight, one workgroup per CU
constexpr int kCostWaves = kCostBlock / 64;
constexpr int kCostGrid = 256;    // one workgroup per CU
// LDS the search tables of a workgroup may take
constexpr size_t kCostLdsBudget = 150 * 1024;
constexpr int kSegChunkMin = 16;
constexpr int kCoopMinSkip = 3;   // empty cells around every point from which the obstacle search
                                  // of a sample is done point by point by the whole wavefront
// Longest list the workgroup-per-sample kernel gets by itself.  Round 2 measured a crossover near 650 samples; since then
// the wavefront-per-sample kernel got the near table, the union-rectangle scan and the folded publish, and round 4's
// density sweep finds it ahead at every list length (cfg3: 0 / 123 / 489 admissible: 10.4 / 24.2 / 25.7 us against
// 7.3 + 6.5 / 27.7 + 6.6 / 36.7 + 6.6 with the publish kernel the block kernel needs; cfg2's three-kernel cycle at 393:
// 18.9 against 16.4 + 6.6).  The block kernel stays behind option cost_kernel = 1.
constexpr long long kBlockKernelMaxAdm = -1;

struct BucketDev {
  int W, H;            // cells
  double gx0, gy0;     // origin
  double g, inv_g;     // cell edge
  double cap;          // max_obstacles_dist * 1.001 (search never needs more)
  const int *cell_start;   // [W*H + 1]
  const float *bx, *by;    // obstacle coordinates in cell order
  const uint8_t *skip;     // [W*H] Chebyshev distance (cells, saturated at 255)
                           // to the nearest non-empty cell: the first block
                           // searched is the smallest that can contain a point
  int nobs;                // finite obstacles in bx/by
};

// device result record (long long slots)
enum { R_KEY = 0, R_NADM = 1, R_COMPACT = 2, R_SPARE = 3,   // published
       W_KEY = 4, W_NADM = 5, W_TICKET = 6, W_LIST = 7,     // working area
       R_SCRATCH = 8, R_TRIGSEQ = 9,  // host-written (BAR): sequence of the trig table in d_trig
       R_SLOTS = 10 };

struct CostArgs {
  int n, first, P, S, O;
  int use_seg, use_obs, have_vel;
  const float *px, *py;
  const uint8_t *flags;
  const int *adm_list;            // admissible local sample ids (any order)
  const long long *adm_count;     // device-side count (result[W_LIST])
  int identity_n;                 // > 0: caller-provided batch, every sample admissible -- the list is 0 .. identity_n - 1
                                  // and neither adm_list nor adm_count is read (no compact_kernel in front)
  const float *sx, *sy, *sz, *szz, *acc_seg;  // contiguous rows [5][S], then the chunk capsules
                                               // (from a 16-byte boundary) [nch] records of 8 floats
                                               // (struct Capsule), then the super-chunk spheres
                                               // [4][nsup]: cx cy cz r, then [nsup] super-chunk capsules
  int seg_chunk, nch;             // points per chunk, chunk count (<= 64)
  float seg_len, ref_len;
  BucketDev b;
  const float *vvx, *vvy, *vom;   // [n][P-1] when have_vel
  const float *vsum_smooth, *vsum_jerk;  // [n] ordered sums of velocity_sums_kernel, or null: formed in the cost kernel
  const float *frz_smooth, *frz_jerk;    // [n] drop_samples = false: the sums of a frozen sample's profile (0 for the
                                         // others), left by the roll-out; null: every sample has a constant profile
  int defer_vel;                  // 1: velocity_sums_kernel runs beside this kernel (second stream); the totals
                                  // stop in front of smoothness / jerk, velocity_finish_kernel adds them and forms the keys
  float max_obs_dist;
  float acc0, acc1, acc2;
  double w_path, w_goal, w_obs, w_smooth, w_jerk;
  float *costs;
  long long *result;    // R_* published record + W_* working area
  long long *block_keys;  // [gridDim.x] best key of every workgroup (publish_kernel reduces)
  unsigned long long *dbg;  // diagnostic build only (KC_DEBUG_STAMPS): per-block phase clocks
  int seg_flat;             // every z of the tracked segment is +0.0f (the common case: planar paths)
  int nsup;                 // super-chunks of 8 chunks (kept at the end: the workgroup-per-sample
                            // kernel lost 5 us when this field sat next to nch -- its scalar
                            // argument loads are sensitive to the layout above)
};

// second argument of the long-list kernel only (the layout of CostArgs is left
// alone: the workgroup-per-sample kernel is sensitive to it)
struct DcArgs {
  // Near table of the tracked segment (segment_near_kernel, kc_segment_kernels.h), or null: for
  // every cell of a grid over the reachable box, the range of chunks that can hold the nearest
  // segment point of ANY point of the cell, and the index of the point nearest to its centre:
  // clo | chi << 8 | j* << 16
  const uint32_t *near;
  float nx0, ny0, ninv;     // origin and 1 / cell edge (floats: the kernels index with float arithmetic,
                            // the slack for that is in the table)
  int nW, nH;
  // Near table of the OBSTACLES of a laser scan (obs_near_kernel), or null.  Consecutive beams of a
  // scan are a polyline: `ocs` consecutive obstacles form a chunk (at most 64 chunks) with a bounding
  // box; per cell of a grid over the reachable box the table holds {mask lo, mask hi, seed, floor}:
  // the chunks that can hold the obstacle nearest to ANY point of the cell, an obstacle nearest to the
  // cell centre, and a lower bound of the distance of any point of the cell to the obstacle set.
  const uint4 *onear;
  float ox0, oy0, oinv;
  int oW, oH;
  const float *osx, *osy;   // [on] obstacle coordinates in SCAN order (what setPointScan computes)
  const float *oaabb;       // [4][64] xmin | xmax | ymin | ymax of the chunks (empty chunks: +inf boxes)
  int on, ocs, onch;
  // long scans (chunks of 32 obstacles and more): every chunk in four quarters of `oscs` obstacles with boxes of
  // their own, [4][256] behind the chunk boxes (quarter q of chunk c: entry 4 c + q); 0: no quarters
  int oscs;
  double ocap;              // max_obstacles_dist
  int ounion;               // > 0: obstacle_union_scan for rectangles of at most this many obstacles
};


#ifdef KC_PHASE_STAMPS
#define KC_STAMP(slot)                                                     \
  do {                                                                     \
    if (a.dbg && threadIdx.x == 0 && blockIdx.x < 512)                                    \
      a.dbg[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define KC_STAMP(slot) do { } while (0)
#endif

// Chunk / super-chunk capsules are records of eight floats (two 16-byte reads, one address):
// ax ay abx aby | 1/|ab|^2 eps az abz  -- a planar segment needs the first six only.  In the
// tracked-segment table they start at a 16-byte boundary behind the five rows.
__host__ __device__ inline int seg_cap_offset(int S) { return (5 * S + 3) & ~3; }
// floats of a scan block (DcArgs::osx): x | y of the scan's obstacles, the chunk boxes [4][64], the quarter boxes [4][256]
__host__ __device__ inline int scan_block_floats(int on, int oscs) { return 2 * on + 256 + (oscs > 0 ? 1024 : 0); }
struct Capsule {
  float ax, ay, abx, aby, inv, eps, az, abz;
};
__device__ __forceinline__ Capsule load_capsule(const float *tab, int c) {
  const float4 u = *reinterpret_cast<const float4 *>(tab + 8 * c);
  const float4 v = *reinterpret_cast<const float4 *>(tab + 8 * c + 4);
  return Capsule{u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
}
// squared distance of (x, y, 0) to the chord, and the slack the comparison needs
__device__ __forceinline__ void capsule_dist2(const Capsule &k, float x, float y, bool flat, float &d2,
                                              float &mag) {
  // (a pruning bound, not part of the exactness contract: fused multiply-adds are fine here,
  // the slack of the comparison covers any rounding)
  const float qx = x - k.ax, qy = y - k.ay;
  if (flat) {
    float t = __builtin_fmaf(qx, k.abx, qy * k.aby) * k.inv;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float ex = __builtin_fmaf(-t, k.abx, qx), ey = __builtin_fmaf(-t, k.aby, qy);
    d2 = __builtin_fmaf(ex, ex, ey * ey);
    mag = fabsf(qx) + fabsf(qy);
  } else {
    const float qz = 0.0f - k.az;
    float t = __builtin_fmaf(qx, k.abx, __builtin_fmaf(qy, k.aby, qz * k.abz)) * k.inv;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float ex = __builtin_fmaf(-t, k.abx, qx), ey = __builtin_fmaf(-t, k.aby, qy),
                ez = __builtin_fmaf(-t, k.abz, qz);
    d2 = __builtin_fmaf(ex, ex, __builtin_fmaf(ey, ey, ez * ez));
    mag = fabsf(qx) + fabsf(qy) + fabsf(qz);
  }
}
__device__ __forceinline__ bool capsule_may_hold(const Capsule &k, float x, float y, float thr, bool flat) {
  float d2, mag;
  capsule_dist2(k, x, y, flat, d2, mag);
  const float lim = __builtin_fmaf(4e-7f, mag, thr + k.eps);
  return !(d2 > lim * lim * 1.0001f);  // NaN compares false: qualifies
}

__device__ __forceinline__ float accum(float total, double w, float c) {
  return static_cast<float>(static_cast<double>(total) +
                            w * static_cast<double>(c));
}
__device__ __forceinline__ float sq_over(float total, float d, float lim) {
  // smoothness_cost += std::pow(delta, 2) / accLimits_[i]  (double, then float)
  const double dd = static_cast<double>(d);
  return static_cast<float>(static_cast<double>(total) +
                            (dd * dd) / static_cast<double>(lim));
}
// The pinned result record {key, counts, seq, check, row word}: plain stores, no
// `volatile` (each volatile store to host memory is followed by a wait for its
// PCIe completion -- five of them cost microseconds); the words may arrive in
// any order, the host accepts the record only when record_check adds up.
__device__ __forceinline__ void store_host_record(long long *hp, long long w0, long long w1, long long seq,
                                                  long long w4) {
  const long long chk = record_check(w0, w1, seq, w4);
  longlong2 *v = reinterpret_cast<longlong2 *>(hp);
  longlong2 a, b;
  a.x = w0;
  a.y = w1;
  b.x = seq;
  b.y = chk;
  hp[4] = w4;
  v[0] = a;
  v[1] = b;
}

// Wave-wide unsigned minimum on the DPP path (four cross-lane ALU steps inside
// each row of 16, then four scalar row reads) instead of six ds_bpermute round
// trips: the searches are latency chains, and a bpermute costs about as much
// as an LDS access.  All 64 lanes must be active; the result is wave-uniform.
template <int kCtrl>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
  return static_cast<uint32_t>(
      __builtin_amdgcn_update_dpp(0, static_cast<int>(v), kCtrl, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  v = min(v, dpp_u32<0xB1>(v));   // quad_perm [1,0,3,2]
  v = min(v, dpp_u32<0x4E>(v));   // quad_perm [2,3,0,1]
  v = min(v, dpp_u32<0x141>(v));  // row_half_mirror
  v = min(v, dpp_u32<0x140>(v));  // row_mirror
  const uint32_t r0 = __builtin_amdgcn_readlane(static_cast<int>(v), 0);
  const uint32_t r1 = __builtin_amdgcn_readlane(static_cast<int>(v), 16);
  const uint32_t r2 = __builtin_amdgcn_readlane(static_cast<int>(v), 32);
  const uint32_t r3 = __builtin_amdgcn_readlane(static_cast<int>(v), 48);
  return min(min(r0, r1), min(r2, r3));
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
  v |= dpp_u32<0xB1>(v);
  v |= dpp_u32<0x4E>(v);
  v |= dpp_u32<0x141>(v);
  v |= dpp_u32<0x140>(v);
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 0)) |
         static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 16)) |
         static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 32)) |
         static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 48));
}
// the same inside each aligned group of eight lanes (three DPP steps; every
// lane of the group ends up with the group's minimum)
__device__ __forceinline__ uint32_t group8_min_u32(uint32_t v) {
  v = min(v, dpp_u32<0xB1>(v));   // quad_perm [1,0,3,2]
  v = min(v, dpp_u32<0x4E>(v));   // quad_perm [2,3,0,1]
  v = min(v, dpp_u32<0x141>(v));  // row_half_mirror
  return v;
}
// ... and inside each aligned group of kL = 4 or 8 lanes (quads need no third step)
template <int kL>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t v) {
  static_assert(kL == 4 || kL == 8 || kL == 16, "four, eight or sixteen lanes per group");
  v = min(v, dpp_u32<0xB1>(v));   // quad_perm [1,0,3,2]
  v = min(v, dpp_u32<0x4E>(v));   // quad_perm [2,3,0,1]
  if (kL >= 8) v = min(v, dpp_u32<0x141>(v));   // row_half_mirror
  if (kL >= 16) v = min(v, dpp_u32<0x140>(v));  // row_mirror
  return v;
}
template <int kL>
__device__ __forceinline__ uint32_t group_or_u32(uint32_t v) {
  v |= dpp_u32<0xB1>(v);
  v |= dpp_u32<0x4E>(v);
  if (kL >= 8) v |= dpp_u32<0x141>(v);
  if (kL >= 16) v |= dpp_u32<0x140>(v);
  return v;
}
template <int kL>
__device__ __forceinline__ double group_min_nonneg(double v) {
  const uint64_t u = static_cast<uint64_t>(__double_as_longlong(v));
  const uint32_t hi = static_cast<uint32_t>(u >> 32), lo = static_cast<uint32_t>(u);
  const uint32_t mh = group_min_u32<kL>(hi);
  const uint32_t ml = group_min_u32<kL>(hi == mh ? lo : 0xFFFFFFFFu);
  return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(mh) << 32) | ml));
}
__device__ __forceinline__ double group8_min_nonneg(double v) {
  const uint64_t u = static_cast<uint64_t>(__double_as_longlong(v));
  const uint32_t hi = static_cast<uint32_t>(u >> 32), lo = static_cast<uint32_t>(u);
  const uint32_t mh = group8_min_u32(hi);
  const uint32_t ml = group8_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
  return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(mh) << 32) | ml));
}
// minimum of non-negative, non-NaN doubles (their bit patterns order like the
// values): high words first, then the low words of the lanes that tie
__device__ __forceinline__ double wave_min_nonneg(double v) {
  const uint64_t u = static_cast<uint64_t>(__double_as_longlong(v));
  const uint32_t hi = static_cast<uint32_t>(u >> 32), lo = static_cast<uint32_t>(u);
  const uint32_t mh = wave_min_u32(hi);
  const uint32_t ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
  return __longlong_as_double(static_cast<long long>((static_cast<uint64_t>(mh) << 32) | ml));
}
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// Ordered float sum ((carry + v0) + v1) + ... + v63 of one value per lane, the order in which
// pathCostFunc adds them (cost_evaluator.cpp:111-141): lane k takes lane k-1's partial sum through
// the DPP wave shift and adds its own value, 63 dependent v_add_f32_dpp instead of 64 v_readlane +
// 64 v_add.  All 64 lanes active; v >= +0 or NaN (so 0 + v == v bit for bit in lane 0, which reads
// 0 from beyond the wave); wave-uniform result.
__device__ __forceinline__ float wave_ordered_sum(float carry, float v, int lane) {
  const float m = lane == 0 ? carry + v : v;
  float r = m;
#pragma unroll
  for (int i = 0; i < 63; ++i) {
    const float prev =
        __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
    r = prev + m;
  }
  return lane_value(r, 63);
}

// ---------------------------------------------------------------------------
// K2b: the same costs for SHORT admissible lists (a cluttered scene leaves a few
// hundred samples): one workgroup per sample, eight lanes per trajectory point,
// so that a sample's searches are spread over eight wavefronts instead of being
// one long chain in a single one.  Brute-force segment scan (eight lanes per
// point), block search around the query cell, same arithmetic and the same
// results as sample_cost_kernel; the host picks the kernel from the admissible
// count of the previous cycle.
// ---------------------------------------------------------------------------
constexpr int kBlkCostBlock = 512;  // 8 wavefronts, two workgroups per CU
constexpr size_t kBlkLdsBudget = 78 * 1024;

// Where the float points of one sample come from: the sample-major rows in
// global memory (split path, kc_cost_evaluate) or the double poses the fused
// cycle kernel still holds in LDS (the float is the one the roll-out would have
// stored).
struct RowPts {
  const float *rx, *ry;
  __device__ __forceinline__ float x(int p) const { return rx[p]; }
  __device__ __forceinline__ float y(int p) const { return ry[p]; }
};
struct PosePts {
  const double2 *row;  // pose k at row[k - 1]; the start pose in the spare slot row[start] (the rows have an
  int start;           // odd pitch >= P: a plain index select, nothing of this struct has to live in memory)
  __device__ __forceinline__ float x(int p) const { return static_cast<float>(row[p == 0 ? start : p - 1].x); }
  __device__ __forceinline__ float y(int p) const { return static_cast<float>(row[p == 0 ? start : p - 1].y); }
};
// Tracked segment as five rows (x | y | z | z^2 | acc: the table in global memory, or the same rows
// in LDS) or, in LDS, as PAIR records: points 2k and 2k+1 as (x0, x1, y0, y1) and (zz0, zz1, acc0, acc1).
// A pair is what one packed-fp32 instruction works on (v_pk_add_f32 / v_pk_mul_f32: two IEEE single
// operations per lane, each rounded like the scalar one), so a scan costs five packed operations and one
// three-way minimum per TWO segment points.  The pair table is padded with x = +inf behind the last point
// (their distance is +inf: above FLT_MAX, never a minimum) up to the end of the last chunk and four pairs
// beyond, so the scans read whole batches without clamping.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__host__ __device__ inline int seg_pairs_padded(int nch, int chunk) { return (nch * chunk) / 2 + 4; }
struct SegRows {
  const float *sx, *sy, *szz, *sacc;
  int S;
  static constexpr bool kPadded = false;
  __device__ __forceinline__ float2 xy1(int j) const { return make_float2(sx[j], sy[j]); }
  __device__ __forceinline__ float zz1(int j) const { return szz[j]; }
  __device__ __forceinline__ float acc(int j) const { return sacc[j]; }
  // (a pair that reaches beyond the table repeats the last point: changes no minimum)
  __device__ __forceinline__ float4 pair_xy(int k) const {
    const int j0 = min(2 * k, S - 1), j1 = min(2 * k + 1, S - 1);
    return make_float4(sx[j0], sx[j1], sy[j0], sy[j1]);
  }
  __device__ __forceinline__ f32x2 pair_zz(int k) const {
    const int j0 = min(2 * k, S - 1), j1 = min(2 * k + 1, S - 1);
    return f32x2{szz[j0], szz[j1]};
  }
};
struct SegPairs {
  const float4 *xy, *za;
  static constexpr bool kPadded = true;
  __device__ __forceinline__ float2 xy1(int j) const {
    const float *f = reinterpret_cast<const float *>(xy) + 4 * (j >> 1) + (j & 1);
    return make_float2(f[0], f[2]);
  }
  __device__ __forceinline__ float zz1(int j) const {
    return (reinterpret_cast<const float *>(za) + 4 * (j >> 1) + (j & 1))[0];
  }
  __device__ __forceinline__ float acc(int j) const {
    return (reinterpret_cast<const float *>(za) + 4 * (j >> 1) + (j & 1))[2];
  }
  __device__ __forceinline__ float4 pair_xy(int k) const { return xy[k]; }
  __device__ __forceinline__ f32x2 pair_zz(int k) const {
    const float2 v = *reinterpret_cast<const float2 *>(za + k);
    return f32x2{v.x, v.y};
  }
};
// pair k of the table in global memory, padded as above
__device__ __forceinline__ void seg_pair_from_rows(const float *sx, const float *sy, const float *szz,
                                                   const float *acc, int S, int k, float4 &xy, float4 &za) {
  const int j0 = 2 * k, j1 = j0 + 1;
  const float inf = __builtin_inff();
  xy = make_float4(j0 < S ? sx[j0] : inf, j1 < S ? sx[j1] : inf, j0 < S ? sy[j0] : 0.0f, j1 < S ? sy[j1] : 0.0f);
  za = make_float4(j0 < S ? szz[j0] : 0.0f, j1 < S ? szz[j1] : 0.0f, j0 < S ? acc[j0] : 0.0f,
                   j1 < S ? acc[j1] : 0.0f);
}
// squared distances of (x, y, 0) to the two points of pair k: dx*dx + (dy*dy + z^2) each, Eigen's
// a + (b + c); a flat segment (every z^2 == +0) leaves dy*dy + 0 == dy*dy out, bit for bit
template <class Seg>
__device__ __forceinline__ f32x2 pair_d2(const Seg &seg, int k, float x, float y, bool flat) {
  const float4 q = seg.pair_xy(k);
  const f32x2 dx = f32x2{q.x, q.y} - x, dy = f32x2{q.z, q.w} - y;
  const f32x2 xx = dx * dx, yy = dy * dy;
  if (flat) return xx + yy;
  return xx + (yy + seg.pair_zz(k));
}
// minimum (as bits) over the pairs [k0, k1), four pairs per batch (a padded table lets the last batch
// run into the next chunk or the padding; an unpadded one repeats its last point)
template <class Seg>
__device__ __forceinline__ uint32_t pairs_min_bits(const Seg &seg, int k0, int k1, float x, float y, bool flat,
                                                   uint32_t bestb) {
  for (int kb = k0; kb < k1; kb += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x2 d = pair_d2(seg, kb + u, x, y, flat);
      bestb = min(bestb, min(__float_as_uint(d.x), __float_as_uint(d.y)));
    }
  }
  return bestb;
}

// The searches of ONE sample by a team of kTeam lanes (a multiple of 64), eight
// lanes per trajectory point: brute-force segment scan, block search around the
// query cell.  Leaves s_mind[P], *s_goal, *s_end and the sample-wide minimum in
// *s_obest (armed by the caller, behind a barrier).  tid = lane id inside the team.
// cap / sup (chunk capsules and super-chunk spheres): the pruned segment search;
// null: the plain scan, eight lanes wide.
template <class Seg, int kL>
__device__ __forceinline__ void group_segment_search(const CostArgs &a, const Seg &seg, const float *cap,
                                                     const float *sup, float x, float y, int sub,
                                                     float &best_out, int &arg_out);

template <int kTeam, class Seg, class Pts, int kL = 8>
__device__ __forceinline__ void team_sample_search(const CostArgs &a, const Seg &seg, float sz_end,
                                                   const int *cells, const uint8_t *skip,
                                                   const float *obx, const float *oby, const Pts pts,
                                                   int tid, float *s_mind, float *s_goal, float *s_end,
                                                   unsigned long long *s_obest, const float *cap = nullptr,
                                                   const float *sup = nullptr, bool skip_obs = false) {
  const BucketDev &b = a.b;
  const int sub = tid & (kL - 1);
  for (int p0 = 0; p0 < a.P; p0 += kTeam / kL) {
    // Idle groups (beyond P) work on a clamped point and write nothing, so
    // the cross-lane steps always see active lanes.
    const int pp = p0 + tid / kL;
    const bool live = pp < a.P;
    const int p = live ? pp : a.P - 1;
    const float x = pts.x(p), y = pts.y(p);
    if (a.use_seg) {
      float best = FLT_MAX;
      int arg = 0;
      if (cap) {
        group_segment_search<Seg, kL>(a, seg, cap, sup, x, y, sub, best, arg);
      } else {
        const bool flat = a.seg_flat != 0;
#pragma unroll 2
        for (int k = sub; 2 * k < a.S; k += kL) {  // j ascending per lane
          const f32x2 d = pair_d2(seg, k, x, y, flat);
          if (d.x < best) {
            best = d.x;
            arg = 2 * k;
          }
          if (d.y < best) {  // (a repeat of the last point or the padding never is)
            best = d.y;
            arg = 2 * k + 1;
          }
        }
        // non-negative floats order like their bit patterns; ties go to the
        // lowest segment index (the reference's strict `<` in index order)
        const uint32_t mine = __float_as_uint(best);
        const uint32_t mbits = group_min_u32<kL>(mine);
        arg = static_cast<int>(
            group_min_u32<kL>(mine == mbits ? static_cast<uint32_t>(arg) : 0xFFFFFFFFu));
        best = __uint_as_float(mbits);
      }
      if (sub == 0 && live) {
        s_mind[p] = kc::sqrt_rn(best);
        if (p == a.P - 1) {
          // goalCostFunc, cost_evaluator.cpp:168-176, from the search above
          const float arc = kc::div_rn(a.ref_len - seg.acc(arg), a.ref_len);
          *s_goal = arc + kc::div_rn(kc::sqrt_rn(best), a.ref_len);
          // end-point term of pathCostFunc, cost_evaluator.cpp:131-136
          const float2 qe = seg.xy1(a.S - 1);
          const float dx = x - qe.x, dy = y - qe.y, dz = 0.0f - sz_end;
          const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
          *s_end = kc::div_rn(kc::sqrt_rn(xx + (yy + zz)), a.seg_len);
        }
      }
    }
    if (a.use_obs && !skip_obs) {  // (skip_obs: a wavefront of the team forms the term, wave_obstacle_term)
      // query cell (clamped: a query outside the grid searches from the
      // border and the guarantee radius shrinks by its distance to the grid)
      const double fx = (static_cast<double>(x) - b.gx0) * b.inv_g;
      const double fy = (static_cast<double>(y) - b.gy0) * b.inv_g;
      int cx = static_cast<int>(floor(fx)), cy = static_cast<int>(floor(fy));
      double off = 0.0;
      if (fx < 0.0) off = fmax(off, -fx);
      if (fy < 0.0) off = fmax(off, -fy);
      if (fx > b.W) off = fmax(off, fx - b.W);
      if (fy > b.H) off = fmax(off, fy - b.H);
      cx = min(max(cx, 0), b.W - 1);
      cy = min(max(cy, 0), b.H - 1);
      double best = DBL_MAX;
      const int mmax = max(b.W, b.H);
      // first block: just large enough to contain the nearest non-empty cell;
      // following blocks: just large enough to prove the best distance found
      const int sk = static_cast<int>(skip[cy * b.W + cx]);
      int m = max(1, sk);
      // Cells nearer (Chebyshev) than sk are empty: nothing is closer than lb.  A point whose lb lies
      // beyond max_obstacles_dist costs nothing, and one whose lb lies beyond what another point of the
      // sample has already found cannot lower the sample's minimum: neither walks any block (in open
      // space that was every point, each scanning a block out to the cap).
      bool search = true;
      {
        const double lb = (static_cast<double>(sk - 1) - off) * b.g;
        if (lb >= b.cap) search = false;
        else if (lb > 0.0) {
          const double seen = __longlong_as_double(static_cast<long long>(
              *const_cast<volatile unsigned long long *>(s_obest)));
          if (lb * lb * (1.0 - 1e-6) >= seen) search = false;
        }
      }
      while (search) {  // uniform within the group of eight, divergent between groups
        const int y0 = max(cy - m, 0), y1 = min(cy + m, b.H - 1);
        const int x0 = max(cx - m, 0), x1 = min(cx + m, b.W - 1);
        // a row of the block is a contiguous run of the cell-ordered obstacle
        // list: two lanes per row, kL / 2 rows per pass
        for (int row = y0 + (sub >> 1); row <= y1; row += kL / 2) {
          const int beg = cells[row * b.W + x0];
          const int end = cells[row * b.W + x1 + 1];
          for (int j = beg + (sub & 1); j < end; j += 2) {
            const double dx = static_cast<double>(obx[j] - x);
            const double dy = static_cast<double>(oby[j] - y);
            const double dd = dx * dx + dy * dy;
            best = dd < best ? dd : best;
          }
        }
        best = group_min_nonneg<kL>(best);
        // Only the minimum over the whole sample is used (trajectory.h:218-235
        // inside obstaclesDistCostFunc), so the points of a sample share
        // their best distance: a point stops as soon as everything it has
        // not visited yet is farther than what some point already found.
        if (sub == 0 && live)
          atomicMin(s_obest, static_cast<unsigned long long>(__double_as_longlong(best)));
        const double shared = __longlong_as_double(static_cast<long long>(
            *const_cast<volatile unsigned long long *>(s_obest)));
        // every obstacle closer than `reach` (true distance) was visited
        const double reach = (static_cast<double>(m) - off) * b.g;
        if (reach > 0.0) {
          const double r2 = reach * reach * (1.0 - 1e-6);
          if (shared < r2) break;
          if (reach >= b.cap) break;
        }
        if (m >= mmax) break;  // whole grid visited
        // next half-width: enough cells to cover sqrt(shared) (+ guard), or
        // the cap radius when nothing has been found yet (any over-estimate
        // only visits more cells: float sqrt is enough)
        const double need =
            shared < DBL_MAX ? static_cast<double>(__builtin_sqrtf(static_cast<float>(shared)) * 1.0001f)
                             : b.cap;
        const double mm = ceil(fmin(need, b.cap * 1.001) * b.inv_g + off) + 1.0;
        m = max(m + 1, static_cast<int>(fmin(mm, static_cast<double>(mmax))));
      }
    }
  }
}

// OR inside each aligned group of eight lanes (every lane gets the group's OR)
__device__ __forceinline__ uint32_t group8_or_u32(uint32_t v) {
  v |= dpp_u32<0xB1>(v);   // quad_perm [1,0,3,2]
  v |= dpp_u32<0x4E>(v);   // quad_perm [2,3,0,1]
  v |= dpp_u32<0x141>(v);  // row_half_mirror
  return v;
}

// Nearest tracked-segment point of q = (x, y) by a group of kL (four or eight)
// lanes (sub = lane id in the group) with the chunk hierarchy of
// wave_sample_total spread over the lanes: (0) the heads of the (<= 8)
// super-chunks, (1) their bounding spheres, (2) the heads of the chunks of the
// surviving super-chunks, (3) the capsules of those chunks, (4) every other
// point of the chunks that may hold something at least as close, kL points per
// step.  Same minimum of d2 = dx*dx + (dy*dy + z^2) and same lowest index as the
// full scan (cost_evaluator.cpp:120-130 / :157-166).  All lanes of the group must
// be active; best / arg come back group-uniform.
template <class Seg, int kL>
__device__ __forceinline__ void group_segment_search(const CostArgs &a, const Seg &seg, const float *cap,
                                                     const float *sup, float x, float y, int sub,
                                                     float &best_out, int &arg_out) {
  const bool flat = a.seg_flat != 0;
  auto d2_to = [&](int j) {  // (chunk heads: even indices, the first half of a pair record)
    const float4 q = seg.pair_xy(j >> 1);
    const float dx = q.x - x;
    const float dy = q.z - y;
    const float xx = dx * dx;
    const float yy = dy * dy;
    if (flat) return xx + yy;
    return xx + (yy + seg.pair_zz(j >> 1).x);  // Eigen order a + (b + c)
  };
  auto merge = [&](float &best, int &arg) {
    // non-negative floats order like their bit patterns (NaN above everything:
    // `d < best` never took one); ties go to the lowest segment index
    const uint32_t mine = __float_as_uint(best);
    const uint32_t mbits = group_min_u32<kL>(mine);
    arg = static_cast<int>(group_min_u32<kL>(mine == mbits ? static_cast<uint32_t>(arg) : 0xFFFFFFFFu));
    best = __uint_as_float(mbits);
  };
  const int sup_pts = 8 * a.seg_chunk;
  float best = FLT_MAX;
  int arg = 0;  // (a lane that finds nothing below FLT_MAX keeps index 0, like the reference's scan)
  // (0) super-chunk heads (ascending per lane)
#pragma unroll
  for (int s = sub; s < 8; s += kL) {
    if (s < a.nsup) {
      const float dd = d2_to(s * sup_pts);
      if (dd < best) {
        best = dd;
        arg = s * sup_pts;
      }
    }
  }
  merge(best, arg);
  // (1) spheres: |q - c| - r <= thr on the squares (NaN compares false: qualifies)
  float thr = __builtin_sqrtf(best) * 1.0001f;
  uint32_t keep = 0u;
#pragma unroll
  for (int s = sub; s < 8; s += kL) {
    if (s < a.nsup) {
      const float dx = sup[s] - x, dy = sup[a.nsup + s] - y, dz = sup[2 * a.nsup + s];
      const float d2 = dx * dx + dy * dy + dz * dz;
      const float lim = thr + sup[3 * a.nsup + s];
      if (!(d2 > lim * lim * 1.00001f)) keep |= 1u << s;
    }
  }
  const uint32_t smask = group_or_u32<kL>(keep);
  // (2) heads of their chunks (chunk 8 s + u, u over the lanes)
  for (uint32_t m = smask; m;) {
    const int s8 = (__ffs(static_cast<int>(m)) - 1) * 8;
    m &= m - 1u;
#pragma unroll
    for (int u = sub; u < 8; u += kL) {
      const int c = s8 + u;
      if (c < a.nch) {
        const int j = c * a.seg_chunk;
        const float dd = d2_to(j);
        if (dd < best || (dd == best && j < arg)) {
          best = dd;
          arg = j;
        }
      }
    }
  }
  merge(best, arg);
  // (3) capsules of those chunks: distance to the chord minus the largest
  // deviation of the chunk's points from it
  thr = __builtin_sqrtf(best) * 1.0001f;
  uint32_t clo = 0u, chi = 0u;
  for (uint32_t m = smask; m;) {
    const int s8 = (__ffs(static_cast<int>(m)) - 1) * 8;
    m &= m - 1u;
#pragma unroll
    for (int u = sub; u < 8; u += kL) {
      const int c = s8 + u;
      if (c < a.nch) {
        if (capsule_may_hold(load_capsule(cap, c), x, y, thr, flat)) {
          if (c < 32) clo |= 1u << c;
          else chi |= 1u << (c - 32);
        }
      }
    }
  }
  clo = group_or_u32<kL>(clo);
  chi = group_or_u32<kL>(chi);
  // (4) the remaining points of those chunks, kL per step
  for (unsigned long long cand = (static_cast<unsigned long long>(chi) << 32) | clo; cand;) {
    const int c = __ffsll(static_cast<long long>(cand)) - 1;
    cand &= cand - 1ull;
    const int j1 = min((c + 1) * a.seg_chunk, a.S);
    for (int k = ((c * a.seg_chunk) >> 1) + sub; 2 * k < j1; k += kL) {  // (chunks hold an even number of points)
      const f32x2 d = pair_d2(seg, k, x, y, flat);
      const int ja = 2 * k, jb = min(2 * k + 1, a.S - 1);
      if (d.x < best || (d.x == best && ja < arg)) {
        best = d.x;
        arg = ja;
      }
      if (jb != ja && (d.y < best || (d.y == best && jb < arg))) {
        best = d.y;
        arg = jb;
      }
    }
  }
  merge(best, arg);
  best_out = best;
  arg_out = arg;
}

// smoothness + jerk of caller-provided velocity profiles (kc_cost_evaluate),
// by one full wavefront: cost += pow(delta, 2) / accLimit in float, each term a
// double quotient rounded into the running float (cost_evaluator.cpp:187-233).
// The quotients do not depend on the running sum, so the lanes form 64 steps'
// worth of them at once (the f64 divisions are the expensive part); only the
// additions, whose order fixes the rounding, walk the lanes in step order.
__device__ __forceinline__ double lane_value_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <bool kJerk>
__device__ __forceinline__ float velocity_cost_sum(const CostArgs &a, const float *vx, const float *vy,
                                                   const float *om, int nv, int lane) {
  constexpr int k_first = kJerk ? 2 : 1;
  float c = 0.0f;
  for (int k0 = k_first; k0 < nv; k0 += 64) {
    const int k = k0 + lane;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    if (k < nv) {
      auto term = [&](const float *v, float lim) {
        const float d = kJerk ? v[k] - 2 * v[k - 1] + v[k - 2] : v[k] - v[k - 1];
        const double dd = static_cast<double>(d);
        return (dd * dd) / static_cast<double>(lim);
      };
      if (a.acc0 > 0) t0 = term(vx, a.acc0);
      if (a.acc1 > 0) t1 = term(vy, a.acc1);
      if (a.acc2 > 0) t2 = term(om, a.acc2);
    }
    // The running float takes the three quotients of a step one after the other (each sum formed in
    // double and rounded back, cost_evaluator.cpp:187-233), step after step: lane j continues from lane
    // j-1's value, handed on by the DPP wave shift (lane 0: the carry of the previous 64 steps) -- nine
    // dependent instructions per step where v_readlane of the double quotients took fifteen.  Steps
    // beyond nv and axes without a limit add +0.0, which leaves the non-negative sum as it is.
    float r = 0.0f;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
      const float prev = __int_as_float(
          __builtin_amdgcn_update_dpp(__float_as_int(c), __float_as_int(r), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
      float v = static_cast<float>(static_cast<double>(prev) + t0);
      v = static_cast<float>(static_cast<double>(v) + t1);
      r = static_cast<float>(static_cast<double>(v) + t2);
    }
    c = lane_value(r, 63);
  }
  return c;
}
__device__ __forceinline__ float add_velocity_costs(const CostArgs &a, int n, float total, int lane) {
  const int nv = a.P - 1;
  const float *vx = a.vvx + (size_t)n * nv;
  const float *vy = a.vvy + (size_t)n * nv;
  const float *om = a.vom + (size_t)n * nv;
  const float div = static_cast<float>(3L * nv);
  // (large batches: the sums come from velocity_sums_kernel, several samples per wavefront)
  if (a.w_smooth > 0.0)  // cost_evaluator.cpp:187-206
    total = accum(total, a.w_smooth,
                  kc::div_rn(a.vsum_smooth ? a.vsum_smooth[n] : velocity_cost_sum<false>(a, vx, vy, om, nv, lane), div));
  if (a.w_jerk > 0.0)  // cost_evaluator.cpp:209-233
    total = accum(total, a.w_jerk,
                  kc::div_rn(a.vsum_jerk ? a.vsum_jerk[n] : velocity_cost_sum<true>(a, vx, vy, om, nv, lane), div));
  return total;
}

// The same sums for LARGE batches of velocity profiles (the reference's CostEvaluator_5k workload: 5001
// profiles of 999 steps).  The additions of one profile are a serial chain -- nine dependent instructions per
// step however many lanes watch -- so a wavefront per sample spends 9 wave instructions per step (5001 x 999 x
// 9 x 2 = 90 M of the 164 M the whole evaluation issued, VALU-issue bound).  Here a wavefront carries
// 64 / kLanes samples at once: kLanes consecutive steps of a sample sit in neighbouring lanes, the running float
// walks them by a DPP rotation inside the group (row_ror for 16 lanes, quad_perm for 4), and the last lane of a
// group keeps the sum between tiles -- lane 0 reads it back through the rotation, so no carry is handed
// around.  Lanes a rotation has not reached yet compute on stale values; only the last lane's value after
// kLanes rounds is used.  Fewer, longer-running wavefronts: a lone wavefront issues one of these dependent f64
// instructions every ~14-26 cycles (SQ_WAVE_CYCLES / SQ_INSTS_VALU), six per SIMD hide that -- 4 samples per
// wavefront pay from ~5 chains per SIMD on (cost5k: 222 -> 115 us for both sums), 16 only for batches near 10^5.
// blockIdx.y: 0 smoothness, 1 jerk (when both are asked for).
struct VelSumArgs {
  const float *vx, *vy, *om;  // [n][nv]
  int n, nv;
  float acc0, acc1, acc2;
  float *out[2];              // [n] smoothness / jerk sums (the float the reference divides by 3 (P - 1))
  int first_kind;             // 0: blockIdx.y = 0 is smoothness; 1: only jerk is asked for
};
template <bool kJerk, int kLanes>
__device__ __forceinline__ void velocity_sums_group(const VelSumArgs &a, float *out) {
  constexpr int k_first = kJerk ? 2 : 1;
  constexpr int kGroups = 64 / kLanes;
  constexpr int kRotate = kLanes == 4 ? 0x93 /* quad_perm:[3,0,1,2] */ : 0x121 /* row_ror:1 */;
  const int lane = threadIdx.x & 63;
  const int j = lane % kLanes;
  const int n = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kGroups + lane / kLanes;
  const bool valid = n < a.n;
  const float *vx = a.vx + static_cast<size_t>(valid ? n : 0) * a.nv;
  const float *vy = a.vy + static_cast<size_t>(valid ? n : 0) * a.nv;
  const float *om = a.om + static_cast<size_t>(valid ? n : 0) * a.nv;
  // The quotients of a tile (f64 divisions, ~100 instructions, independent of the running sum) are formed one
  // tile AHEAD of the additions that consume them and without a branch (steps beyond the row and axes without
  // a limit read a valid entry and select +0.0, which leaves the non-negative sum as it is): one basic block,
  // so the divisions of tile i + 1 fill the issue slots the dependent chain of tile i leaves empty.
  auto quotients = [&](int k0, double &t0, double &t1, double &t2) {
    const int k = k0 + j;
    const bool in = valid && k < a.nv;
    const int kk = in ? k : k_first;
    auto term = [&](const float *v, float lim) {
      const float d = kJerk ? v[kk] - 2 * v[kk - 1] + v[kk - 2] : v[kk] - v[kk - 1];
      const double dd = static_cast<double>(d);
      const double q = (dd * dd) / static_cast<double>(lim);
      return (in && lim > 0) ? q : 0.0;
    };
    t0 = term(vx, a.acc0);
    t1 = term(vy, a.acc1);
    t2 = term(om, a.acc2);
  };
  float r = 0.0f;
  double t0, t1, t2;
  quotients(k_first, t0, t1, t2);
  for (int k0 = k_first; k0 < a.nv; k0 += kLanes) {
    double n0, n1, n2;
    quotients(k0 + kLanes, n0, n1, n2);
#pragma unroll
    for (int q = 0; q < kLanes; ++q) {
      const float prev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), kRotate, 0xf, 0xf, true));
      float v = static_cast<float>(static_cast<double>(prev) + t0);
      v = static_cast<float>(static_cast<double>(v) + t1);
      r = static_cast<float>(static_cast<double>(v) + t2);
    }
    t0 = n0;
    t1 = n1;
    t2 = n2;
  }
  if (valid && j == kLanes - 1) out[n] = r;
}
constexpr int kVelBlock = 256;
template <int kLanes>
__global__ __launch_bounds__(kVelBlock) void velocity_sums_kernel(VelSumArgs a) {
  if (blockIdx.y + a.first_kind == 0)
    velocity_sums_group<false, kLanes>(a, a.out[0]);
  else
    velocity_sums_group<true, kLanes>(a, a.out[1]);
}

// drop_samples = false: the profile of a frozen sample steps to zero once, its two sums come from the roll-out
__device__ __forceinline__ float add_frozen_costs(const CostArgs &a, int n, float total) {
  const float div = static_cast<float>(3L * (a.P - 1));
  if (a.w_smooth > 0.0) total = accum(total, a.w_smooth, kc::div_rn(a.frz_smooth[n], div));  // cost_evaluator.cpp:187-206
  if (a.w_jerk > 0.0) total = accum(total, a.w_jerk, kc::div_rn(a.frz_jerk[n], div));        // :209-233
  return total;
}

// Behind a cost kernel that ran with defer_vel (beside velocity_sums_kernel on a second stream): the last two
// terms of getMinTrajectoryCost (cost_evaluator.cpp:49-109: ... smoothness, jerk) onto the stored totals, and
// the per-workgroup keys for publish_kernel.
struct VelFinishArgs {
  const int *adm_list;
  const long long *adm_count;
  int identity_n;               // as in CostArgs
  float *costs;                 // [n] in: total in front of smoothness; out: total
  const float *vsum_smooth, *vsum_jerk;  // null: weight 0
  double w_smooth, w_jerk;
  float div;                    // 3 (P - 1)
  int first;
  long long *block_keys;        // [gridDim.x]
};

// obstaclesDistCostFunc, cost_evaluator.cpp:179-184, from the minimum squared distance (double)
__device__ __forceinline__ float obstacle_cost_from(const CostArgs &a, double best) {
  const float min_d2 = static_cast<float>(best);
  const float dist = static_cast<float>(kc::dsqrt_rn(static_cast<double>(min_d2)));
  float v = a.max_obs_dist - dist;
  v = v < 0.0f ? 0.0f : v;
  return kc::div_rn(v, a.max_obs_dist);
}

// weighted total of the sample a team has searched, by ONE wavefront (uniform result)
__device__ __forceinline__ float team_sample_total(const CostArgs &a, int n, int lane,
                                                   const float *s_mind, float s_goal, float s_end,
                                                   unsigned long long s_obest) {
  float total = 0.0f;
  if (a.ref_len > 0.0f) {
    if (a.w_goal > 0.0) total = accum(total, a.w_goal, s_goal);
    if (a.w_path > 0.0) {
      // pathCostFunc, cost_evaluator.cpp:111-141: ordered float sum
      // (lanes beyond the last point hold +0.0f: adding it leaves the non-negative
      // sum as it is, so the 64 additions of a tile need no loop or branch)
      float sum = 0.0f;
      for (int base = 0; base < a.P; base += 64) {
        const int cnt = min(64, a.P - base);
        const float v = (lane < cnt) ? s_mind[base + lane] : 0.0f;
        sum = wave_ordered_sum(sum, v, lane);
      }
      const float c = kc::div_rn(
          kc::div_rn(sum, static_cast<float>(a.P)) + s_end, 2.0f);
      total = accum(total, a.w_path, c);
    }
  }
  if (a.O > 0 && a.w_obs > 0.0)
    total = accum(total, a.w_obs,
                  obstacle_cost_from(a, __longlong_as_double(static_cast<long long>(s_obest))));
  if (a.have_vel) total = add_velocity_costs(a, n, total, lane);
  else if (a.frz_smooth) total = add_frozen_costs(a, n, total);
  // constant-velocity samples: both terms are exactly 0 and `total += w*0`
  // leaves total unchanged, so nothing to do when !have_vel.
  return total;
}

template <bool kLds, bool kObsLds>
__global__ __launch_bounds__(kBlkCostBlock, 4) void sample_cost_block_kernel(CostArgs a_) {
  const CostArgs &a = *kernargs_touched<CostArgs>();  // (the arguments as read behind the touch of every kernarg line)
  extern __shared__ __align__(16) unsigned char smem[];
  float *s_mind = reinterpret_cast<float *>(smem);               // [P]
  float *s_px = s_mind + a.P;                                    // [P]
  float *s_py = s_px + a.P;                                      // [P]
  __shared__ float s_goal, s_end;
  __shared__ long long s_key;
  __shared__ unsigned long long s_obest;  // sample-wide min squared obstacle distance (double bits)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform
    const int na = a.identity_n > 0 ? a.identity_n : static_cast<int>(*a.adm_count);
  // blocks beyond the admissible count have nothing to do and take no ticket
  const unsigned working = static_cast<unsigned>(min(static_cast<int>(gridDim.x), max(na, 1)));
  if (blockIdx.x >= working) {
    if (threadIdx.x == 0) a.block_keys[blockIdx.x] = KEY_NONE;
    return;
  }
  if (threadIdx.x == 0) s_key = KEY_NONE;

  const BucketDev &b = a.b;
  // LDS layout after the per-sample arrays: five segment rows, the cell table,
  // the skip table (padded to words), the obstacle coordinates.  The pointers
  // are chosen at compile time so that the LDS variants issue ds_read, not
  // flat loads.
  const int ncell = b.W * b.H;
  float *const l_seg = s_py + a.P;
  int *const l_cells = reinterpret_cast<int *>(l_seg + (a.use_seg ? 5 * a.S : 0));
  uint8_t *const l_skip = reinterpret_cast<uint8_t *>(l_cells + (a.use_obs ? ncell + 1 : 0));
  float *const l_obs = reinterpret_cast<float *>(l_skip + (a.use_obs ? ((ncell + 3) & ~3) : 0));
  const int *const cells = kLds ? l_cells : b.cell_start;
  const uint8_t *const skip = kLds ? l_skip : b.skip;
  const float *const obx = kObsLds ? l_obs : b.bx;
  const float *const oby = kObsLds ? l_obs + b.nobs : b.by;
  const float *const sx = kLds ? l_seg : a.sx;
  const float *const sy = kLds ? l_seg + a.S : a.sy;
  const float *const sz = kLds ? l_seg + 2 * a.S : a.sz;
  const float *const szz = kLds ? l_seg + 3 * a.S : a.szz;
  const float *const sacc = kLds ? l_seg + 4 * a.S : a.acc_seg;
  if (kLds) {
    if (a.use_seg) {
      // the five rows are contiguous in global memory too (d_seg)
#pragma unroll 4
      for (int j = threadIdx.x; j < 5 * a.S; j += kBlkCostBlock) l_seg[j] = a.sx[j];
    }
    if (a.use_obs) {
#pragma unroll 4
      for (int j = threadIdx.x; j <= ncell; j += kBlkCostBlock) l_cells[j] = b.cell_start[j];
      // the skip table is padded to a multiple of 4 bytes on the host
      const uint32_t *gs = reinterpret_cast<const uint32_t *>(b.skip);
      uint32_t *ls = reinterpret_cast<uint32_t *>(l_skip);
      for (int j = threadIdx.x; j < (ncell + 3) / 4; j += kBlkCostBlock) ls[j] = gs[j];
      if (kObsLds) {
#pragma unroll 4
        for (int j = threadIdx.x; j < b.nobs; j += kBlkCostBlock) {
          l_obs[j] = b.bx[j];
          l_obs[b.nobs + j] = b.by[j];
        }
      }
    }
  }
  const SegRows seg{sx, sy, szz, sacc, a.S};
  const RowPts pts{s_px, s_py};

  for (int i = blockIdx.x; i < na; i += gridDim.x) {
    const int n = a.identity_n > 0 ? i : a.adm_list[i];
    if (threadIdx.x == 0)
      s_obest = static_cast<unsigned long long>(__double_as_longlong(DBL_MAX));
    if (threadIdx.x < a.P) {
      s_px[threadIdx.x] = a.px[(size_t)n * a.P + threadIdx.x];
      s_py[threadIdx.x] = a.py[(size_t)n * a.P + threadIdx.x];
    }
    for (int k = kBlkCostBlock + threadIdx.x; k < a.P; k += kBlkCostBlock) {
      s_px[k] = a.px[(size_t)n * a.P + k];
      s_py[k] = a.py[(size_t)n * a.P + k];
    }
    __syncthreads();  // also covers the structure copy above
    team_sample_search<kBlkCostBlock>(a, seg, (a.use_seg && a.S > 0) ? sz[a.S - 1] : 0.0f, cells, skip,
                                      obx, oby, pts, threadIdx.x, s_mind, &s_goal, &s_end, &s_obest);
    __syncthreads();
    // ---- wavefront 0: weighted total of this sample ---------------------------
    if (wave == 0) {
      const float total = team_sample_total(a, n, lane, s_mind, s_goal, s_end, s_obest);
      if (lane == 0) {
        a.costs[n] = total;
        if (total < FLT_MAX) {  // `total_cost < minCost`, minCost starts at FLT_MAX
          const long long k = key_pack(total, static_cast<uint32_t>(a.first + n));
          if (k < s_key) s_key = k;
        }
      }
    }
    __syncthreads();  // LDS minima are reused by the next sample
  }

  // ---- block epilogue: the block's best key, for publish_kernel ----------------
  if (threadIdx.x == 0) a.block_keys[blockIdx.x] = s_key;
}


// ---- obstacle term of a sample near clutter: one scan of the union block ------------------
// The private ring walks below run in lock-step: every lane pays for the longest walk of the
// wavefront, and each walk re-reads cell runs that its neighbours read too.  Only the minimum
// over the WHOLE trajectory counts (trajectory.h:218-235), so where the trajectory runs through
// occupied cells the wavefront does this instead:
//   (1) every point takes a seed, the first and last obstacle of the block of (2 s + 1)^2
//       bucket cells around its own (s <= 2 = the smallest skip value of the trajectory, found
//       by ballots: at least one such block is non-empty): a bound B above an ATTAINED distance;
//   (2) points whose empty neighbourhood already reaches beyond sqrt(B) drop out, the others
//       need the obstacles within R = sqrt(B) (1 + 1e-4) of themselves: all of those lie in
//       the bucket cells [cell(x - R), cell(x + R)] x [cell(y - R), cell(y + R)] (the cell map
//       is the monotone one the sensor build sorts the obstacles with), and the union of these
//       rectangles over the points is ONE rectangle of cells;
//   (3) its obstacles (row runs of the cell-sorted list, a row per lane, flattened to one
//       obstacle per lane and load) are broadcast one at a time to all lanes: every lane forms
//       the exact distance of ITS point, in the arithmetic of the ring walks.
// Every value formed is a true distance of a trajectory point, and the pair that attains the
// minimum is inside the rectangle: the result is the minimum of the full scan.  A rectangle
// with more than `limit` obstacles (walls of a dense scan, far clutter) returns false with B
// left in *obest, and the ring walks / the cooperative pass take over.  Grids of at most
// 64 x 64 cells (the host checks).  Returns wave-uniform.
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t wave_pk_min_u16(uint32_t v) {
  v = pk_min_u16(v, dpp_u32<0xB1>(v));
  v = pk_min_u16(v, dpp_u32<0x4E>(v));
  v = pk_min_u16(v, dpp_u32<0x141>(v));
  v = pk_min_u16(v, dpp_u32<0x140>(v));
  const uint32_t r0 = __builtin_amdgcn_readlane(static_cast<int>(v), 0);
  const uint32_t r1 = __builtin_amdgcn_readlane(static_cast<int>(v), 16);
  const uint32_t r2 = __builtin_amdgcn_readlane(static_cast<int>(v), 32);
  const uint32_t r3 = __builtin_amdgcn_readlane(static_cast<int>(v), 48);
  const uint32_t a = min(min(r0 >> 16, r1 >> 16), min(r2 >> 16, r3 >> 16));
  const uint32_t b = min(min(r0 & 0xFFFFu, r1 & 0xFFFFu), min(r2 & 0xFFFFu, r3 & 0xFFFFu));
  return (a << 16) | b;
}
__device__ __forceinline__ uint32_t wave_add_u32(uint32_t v) {
  v += dpp_u32<0xB1>(v);
  v += dpp_u32<0x4E>(v);
  v += dpp_u32<0x141>(v);
  v += dpp_u32<0x140>(v);
  return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 0)) +
         static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 16)) +
         static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 32)) +
         static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), 48));
}

constexpr int kScanTransposeCost = 5;  // (3 and 8 measured: 105.5 / 108.8 us of cfg5's cost kernel with a scan against 104.4)  // scan near table: a chunk of n obstacles is taken a lane an obstacle for up to n / 5 points
constexpr int kUnionSeedMax = 24;  // largest seed block of obstacle_union_scan, cells of half-width
__device__ __forceinline__ bool obstacle_union_scan(const BucketDev &b, int limit, const int *cells,
                                                    const uint8_t *skip, const float *obx, const float *oby,
                                                    float x, float y, bool live, int lane,
                                                    unsigned long long *obest) {
  const bool fin = live && __builtin_isfinite(x) && __builtin_isfinite(y);  // (the others never win a minimum)
  if (__ballot(fin) == 0ull) return true;
  // Cell coordinates in double like the sensor build's cell map, then float: everything up to the exact
  // distances of step (3) is a BOUND, kept conservative by explicit slack (the kernel is VALU-issue bound and a
  // double operation costs two float ones).
  const float fx = static_cast<float>((static_cast<double>(x) - b.gx0) * b.inv_g);
  const float fy = static_cast<float>((static_cast<double>(y) - b.gy0) * b.inv_g);
  const float Wf = static_cast<float>(b.W), Hf = static_cast<float>(b.H);
  const float off = fmaxf(fmaxf(fmaxf(-fx, fx - Wf), fmaxf(-fy, fy - Hf)), 0.0f);  // cells outside the grid
  const float ferr = (fabsf(fx) + fabsf(fy)) * 2.4e-7f + 1e-5f;                  // of fx, fy, off (cells)
  const int cx = fin ? min(max(static_cast<int>(floorf(fx)), 0), b.W - 1) : 0;
  const int cy = fin ? min(max(static_cast<int>(floorf(fy)), 0), b.H - 1) : 0;
  const int sk = fin ? static_cast<int>(skip[cy * b.W + cx]) : 255;
  const float gdn = static_cast<float>(b.g) * (1.0f - 2e-7f), gup = static_cast<float>(b.inv_g) * (1.0f + 2e-7f);
  const float capf = static_cast<float>(b.cap) * (1.0f + 2e-7f);
  // cells nearer (Chebyshev) than sk to (cx, cy) are empty: nothing is closer than this to the point, metres
  const float lbm = (static_cast<float>(sk - 1) - off - ferr) * gdn;
  const bool near = fin && !(lbm >= capf);  // (beyond max_obstacles_dist: costs nothing)
  if (__ballot(near) == 0ull) return true;
  if (__ballot(near && sk >= 255) != 0ull) return false;  // (a saturated skip value inside the cap: the walks)
  auto dd_f32 = [&](int j) {
    const float dx = obx[j] - x, dy = oby[j] - y;
    return __builtin_fmaf(dx, dx, dy * dy);
  };
  // (1) seeds: the smallest block around every point in which some point finds an obstacle
  // (the block of (2 s + 1)^2 cells around a point holds an obstacle exactly when its skip value is <= s)
  // s = the smallest skip value of the sample's points.  (Round 4: it used to stop at 2 -- "nothing close to any point:
  // the walks" -- and a sparse scene, a few obstacles two metres from every trajectory, sent all 8192 samples of cfg2 down
  // the ring walks: 73 us a cycle against 37 with four times the obstacles.  Inside max_obstacles_dist a skip value is
  // a dozen cells at most; the seed pass below costs 2 s + 1 rows a lane.)
  const int s = static_cast<int>(wave_min_u32(fin ? static_cast<uint32_t>(sk) : 255u));
  if (s > kUnionSeedMax) return false;  // (long walks: the cooperative pass)
  float sf = __builtin_inff();
  for (int dy = -s; dy <= s; ++dy) {
    const int row = cy + dy;
    if (fin && row >= 0 && row < b.H) {
      const int beg = cells[row * b.W + max(cx - s, 0)], end = cells[row * b.W + min(cx + s, b.W - 1) + 1];
      if (beg < end) sf = fminf(sf, fminf(dd_f32(beg), dd_f32(end - 1)));  // (NaN never wins)
    }
  }
  if (__ballot(sf < 3.0e38f) == 0ull) return false;  // (cannot happen: the skip table said otherwise)
  // B: a float ABOVE the exact squared distance of the best seed (and of an earlier tile's minimum)
  const double prev = __longlong_as_double(static_cast<long long>(*const_cast<volatile unsigned long long *>(obest)));
  const float bseed = __uint_as_float(wave_min_u32(__float_as_uint(sf))) * 1.000002f;
  const float bf = fminf(bseed, static_cast<float>(prev) * 1.000001f) + 1e-37f;
  // (2) the points that can still lower it, the rectangle of cells they need
  const bool cont = near && !(lbm > 0.0f && lbm * lbm * (1.0f - 1e-6f) > bf);
  if (__ballot(cont) == 0ull) return true;
  const float Rg = fminf(__builtin_sqrtf(bf) * 1.0001f + 1e-9f, capf * 1.001f) * gup;  // radius, cells
  uint32_t lo = 0xFFFFFFFFu, hi = 0xFFFFFFFFu;
  if (cont) {
    const float e = Rg + ferr;
    const uint32_t xl = static_cast<uint32_t>(min(max(static_cast<int>(floorf(fx - e)), 0), b.W - 1));
    const uint32_t yl = static_cast<uint32_t>(min(max(static_cast<int>(floorf(fy - e)), 0), b.H - 1));
    const uint32_t xh = static_cast<uint32_t>(min(max(static_cast<int>(floorf(fx + e)), 0), b.W - 1));
    const uint32_t yh = static_cast<uint32_t>(min(max(static_cast<int>(floorf(fy + e)), 0), b.H - 1));
    lo = (xl << 16) | yl;
    hi = ((63u - xh) << 16) | (63u - yh);
  }
  lo = wave_pk_min_u16(lo);
  hi = wave_pk_min_u16(hi);
  const int x0 = static_cast<int>(lo >> 16), y0 = static_cast<int>(lo & 0xFFFFu);
  const int x1 = 63 - static_cast<int>(hi >> 16), y1 = 63 - static_cast<int>(hi & 0xFFFFu);
  // (3) its rows, one per lane
  const int nr = y1 - y0 + 1;
  int rb = 0;
  uint32_t cnt = 0u;
  if (lane < nr) {
    rb = cells[(y0 + lane) * b.W + x0];
    cnt = static_cast<uint32_t>(cells[(y0 + lane) * b.W + x1 + 1] - rb);
  }
  const unsigned long long rows = __ballot(cnt != 0u);
  double best = DBL_MAX;
  uint32_t T = 0xFFFFFFFFu;  // obstacles in the rectangle: known after the first flattening
  for (uint32_t base = 0u; base < T; base += 64u) {
    // obstacle base + lane of the rectangle
    const uint32_t i = base + static_cast<uint32_t>(lane);
    int j = -1;
    uint32_t acc = 0u;
    for (unsigned long long rm = rows; rm;) {
      const int r = __ffsll(static_cast<long long>(rm)) - 1;
      rm &= rm - 1ull;
      const uint32_t cr = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cnt), r));
      const int br = __builtin_amdgcn_readlane(rb, r);
      const uint32_t rel = i - acc;
      if (rel < cr) j = br + static_cast<int>(rel);
      acc += cr;
    }
    T = acc;
    if (T > static_cast<uint32_t>(limit)) return false;
    const float ox = j >= 0 ? obx[j] : 0.0f, oy = j >= 0 ? oby[j] : 0.0f;
    const int m = static_cast<int>(min(T - base, 64u));
    // A float filter in front of the exact distance: dx and dy are the float differences the exact form
    // converts, their squares are exact in double, so the float sum of squares is within 3 ulp (float) of the
    // exact value -- an obstacle whose float value lies above B cannot lower the minimum for ANY lane, and
    // most of the rectangle is like that.  The best seed passes (it defines B) and lies in the rectangle.
    for (int u0 = 0; u0 < m; u0 += 4) {
      float dxf[4], dyf[4];
      bool any = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int u = min(u0 + k, m - 1);  // (a repeat of the last one changes nothing)
        dxf[k] = lane_value(ox, u) - x;
        dyf[k] = lane_value(oy, u) - y;
        const float f = __builtin_fmaf(dxf[k], dxf[k], dyf[k] * dyf[k]);
        any = any || !(f > bf);  // (NaN: evaluated exactly, where it never wins)
      }
      if (__ballot(any) == 0ull) continue;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double dx = static_cast<double>(dxf[k]), dy = static_cast<double>(dyf[k]);
        best = __builtin_fmin(dx * dx + dy * dy, best);  // (NaN distances never win)
      }
    }
  }
  const double found = wave_min_nonneg(best);
  if (lane == 0 && found < DBL_MAX) atomicMin(obest, static_cast<unsigned long long>(__double_as_longlong(found)));
  return true;
}

// The obstacle term of ONE tile of a sample by a wavefront, a lane a trajectory point (x, y; `live` false: an idle lane
// shadowing a point): the minimum squared distance to the obstacles goes into *obest (armed with DBL_MAX by the
// caller), by the scan's near table, one scan of the union rectangle of bucket cells, or the ring walk.  ubound2 carries
// a bound from tile to tile of a long trajectory.  Part of wave_sample_total; the teams of the cycle kernel call it
// too (cycle_costs: their own block walk a point apiece runs four lanes a point and as many trips as its longest row).
__device__ __forceinline__ void wave_obstacle_term(const CostArgs &a, const DcArgs &t, const int *cells, const uint8_t *skip,
                                                   const float *obx, const float *oby, float x, float y, bool live, int lane,
                                                   unsigned long long *obest, double &ubound2) {
  const BucketDev &b = a.b;
  if (a.use_obs && t.onear != nullptr) {
    // Laser scan: the obstacles are a polyline in beam order, cut into <= 64 chunks with bounding boxes, and
    // the near table of the scan names, per cell of the reachable box, the chunks that can hold the nearest
    // obstacle of ANY point of the cell, an obstacle nearest to the cell centre (an attained upper bound) and
    // a lower bound of the distance of the cell's points.  Only the minimum over the whole trajectory counts
    // (trajectory.h:218-235): the seeds give a tight bound on it, lanes whose floor lies above that bound
    // drop out, and the candidate chunks of the others are scanned one at a time by ALL lanes (a lane a
    // trajectory point, a loop over the chunk's obstacles: every value formed is a true distance, so nobody
    // needs masking and the minimum is the one of the full scan).
    // (with a scan, obx / oby are the caller's view of the scan block -- x | y in scan order | chunk boxes, DcArgs::osx:
    // in LDS in the stand-alone cost kernels, where every chunk scanned was a chain of round trips to L2)
    const float *const oaabb = obx + 2 * t.on;
    const float fx = (x - t.ox0) * t.oinv, fy = (y - t.oy0) * t.oinv;
    const bool inside = fx >= 0.0f && fy >= 0.0f && fx < static_cast<float>(t.oW) && fy < static_cast<float>(t.oH);
    uint4 e = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u);  // outside the table: every chunk, no seed
    if (inside) e = t.onear[static_cast<int>(fy) * t.oW + static_cast<int>(fx)];
    auto exact_dd = [&](int j) {
      const double dx = static_cast<double>(obx[j] - x);
      const double dy = static_cast<double>(oby[j] - y);
      return dx * dx + dy * dy;
    };
    double best = DBL_MAX;
    if (live && e.z != 0xFFFFFFFFu) best = exact_dd(static_cast<int>(e.z));
    // The bound the lanes share is only ever COMPARED with: a float above the sample's minimum so far (a wave
    // minimum of 32-bit keys and float tests, where the exact double took a 64-bit reduction behind every quarter
    // scanned and double products per chunk: the reductions were as long as the scans); the exact minimum is
    // formed once, at the end.
    auto above = [](double v) { return static_cast<float>(v) * 1.0000003f + 1e-37f; };  // >= v (+inf stays)
    auto wave_min_f32 = [](float v) { return __uint_as_float(wave_min_u32(__float_as_uint(v))); };  // v >= 0
    float shared = fminf(above(ubound2), wave_min_f32(above(best)));
    const float lbk = __uint_as_float(e.w);
    const float lbk2 = lbk * lbk * (1.0f - 1e-6f);  // (below lbk^2: dropping a lane needs lbk^2 >= the bound for certain)
    uint32_t mlo = live ? e.x : 0u, mhi = live ? e.y : 0u;
    bool cont = live && (mlo | mhi) != 0u && static_cast<double>(lbk) < t.ocap && !(x != x) && !(y != y);
    for (int half = 0; half < 2; ++half) {
      // (re-evaluated per half: the bound only falls)
      uint32_t U = wave_or_u32((cont && lbk2 < shared) ? (half ? mhi : mlo) : 0u);
      while (U) {
        const int cb = __ffs(static_cast<int>(U)) - 1;
        U &= U - 1u;
        const int c = cb + 32 * half;
        if (c >= t.onch) break;
        // box of the chunk against this lane's point (float, 1e-4 of slack on the compared square)
        const float bx0 = oaabb[c], bx1 = oaabb[64 + c], by0 = oaabb[128 + c], by1 = oaabb[192 + c];
        const float gx = fmaxf(fmaxf(bx0 - x, x - bx1), 0.0f), gy = fmaxf(fmaxf(by0 - y, y - by1), 0.0f);
        const float lb2 = (gx * gx + gy * gy) * (1.0f - 1e-4f);
        const bool part = cont && (((half ? mhi : mlo) >> cb) & 1u) && lbk2 < shared && !(lb2 >= shared);
        const unsigned long long pm = __ballot(part);
        if (pm == 0ull) continue;
        const int j0 = c * t.ocs, j1 = min(j0 + t.ocs, t.on);
        if (__popcll(pm) * kScanTransposeCost <= j1 - j0) {
          // Few points want this chunk (two of fifty on average in a room: the others' floors lie above the
          // bound): one of them at a time with a LANE AN OBSTACLE -- one pass and one reduction per point
          // instead of every lane walking the whole chunk for the sake of two.  Same differences, same products.
          for (unsigned long long m = pm; m; m &= m - 1ull) {
            const int L = __ffsll(static_cast<long long>(m)) - 1;
            const float xq = lane_value(x, L), yq = lane_value(y, L);
            double d = DBL_MAX;
            for (int j = j0 + lane; j < j1; j += 64) {
              const double dx = static_cast<double>(obx[j] - xq);
              const double dy = static_cast<double>(oby[j] - yq);
              d = __builtin_fmin(dx * dx + dy * dy, d);  // (NaN distances never win)
            }
            const double mn = wave_min_nonneg(d);
            if (lane == L) best = __builtin_fmin(mn, best);
          }
          shared = fminf(shared, wave_min_f32(above(best)));
          continue;
        }
        if (t.oscs > 0) {
          // a quarter of the chunk at a time: the same test against the quarter's own box, the bound refreshed
          // behind every quarter that was scanned (a 4096-beam scan has chunks of 64 obstacles: most of a chunk
          // the table names lies beyond what a neighbouring quarter has already found)
          const float *sb = oaabb + 256;
          for (int q = 0; q < 4; ++q) {
            const int s0 = j0 + q * t.oscs, s1 = min(s0 + t.oscs, j1);
            if (s0 >= s1) break;
            const int e4 = 4 * c + q;
            const float qx0 = sb[e4], qx1 = sb[256 + e4], qy0 = sb[512 + e4], qy1 = sb[768 + e4];
            const float hx = fmaxf(fmaxf(qx0 - x, x - qx1), 0.0f), hy = fmaxf(fmaxf(qy0 - y, y - qy1), 0.0f);
            const float lq2 = (hx * hx + hy * hy) * (1.0f - 1e-4f);
            if (__ballot(part && !(lq2 >= shared)) == 0ull) continue;
            for (int jb = s0; jb < s1; jb += 4) {
              double d[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) d[u] = exact_dd(min(jb + u, s1 - 1));
#pragma unroll
              for (int u = 0; u < 4; ++u) best = __builtin_fmin(d[u], best);
            }
            shared = fminf(shared, wave_min_f32(above(best)));
          }
          continue;
        }
        for (int jb = j0; jb < j1; jb += 4) {
          double d[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) d[u] = exact_dd(min(jb + u, j1 - 1));  // (a repeat of the last one changes nothing)
#pragma unroll
          for (int u = 0; u < 4; ++u) best = __builtin_fmin(d[u], best);    // (NaN distances never win)
        }
        shared = fminf(shared, wave_min_f32(above(best)));
      }
    }
    ubound2 = fmin(ubound2, wave_min_nonneg(best));  // (exact; carried to the next tile of a long trajectory)
    if (lane == 0) atomicMin(obest, static_cast<unsigned long long>(__double_as_longlong(ubound2)));
  } else if (a.use_obs && t.ounion > 0 &&
             obstacle_union_scan(b, t.ounion, cells, skip, obx, oby, x, y, live, lane, obest)) {
    // (one scan of the union block did it)
  } else if (a.use_obs) {
    // query cell (clamped: a query outside the grid searches from the
    // border and the guarantee radius shrinks by its distance to the grid)
    const double fx = (static_cast<double>(x) - b.gx0) * b.inv_g;
    const double fy = (static_cast<double>(y) - b.gy0) * b.inv_g;
    int cx = static_cast<int>(floor(fx)), cy = static_cast<int>(floor(fy));
    double off = 0.0;
    if (fx < 0.0) off = fmax(off, -fx);
    if (fy < 0.0) off = fmax(off, -fy);
    if (fx > b.W) off = fmax(off, fx - b.W);
    if (fy > b.H) off = fmax(off, fy - b.H);
    cx = min(max(cx, 0), b.W - 1);
    cy = min(max(cy, 0), b.H - 1);
    const int mmax = max(b.W, b.H);
    const int sk = static_cast<int>(skip[cy * b.W + cx]);
    // cells closer (Chebyshev) than sk are empty: the first ring is sk, and
    // nothing is closer than (sk - 1 - off) cells
    int pm = sk - 1;           // half-width of the block known to be empty / visited
    int m = max(1, sk);
    double best = DBL_MAX;
    bool active = live && !(isnan(fx) || isnan(fy));
    if ((static_cast<double>(pm) - off) * b.g >= b.cap) active = false;  // all of it costs 0
    // lanes with an empty neighbourhood wait for the cooperative pass below
    bool far = active && sk >= kCoopMinSkip && sk < 255;
    if (far) active = false;
    double lb0 = fmax((static_cast<double>(pm) - off) * b.g, 0.0);
    double ubp = DBL_MAX;
    while (__ballot(active)) {
      if (active) {
        const int y0 = max(cy - m, 0), y1 = min(cy + m, b.H - 1);
        const int x0 = max(cx - m, 0), x1 = min(cx + m, b.W - 1);
        for (int row = y0; row <= y1; ++row) {
          // rows inside the visited block only add the two side runs
          const bool inner = pm >= 0 && row >= cy - pm && row <= cy + pm;
          int beg = cells[row * b.W + x0];
          int end = inner ? cells[row * b.W + max(cx - pm, x0)]
                          : cells[row * b.W + x1 + 1];
          for (int pass = 0; pass < 2; ++pass) {
            // (a run of a thinned scene holds one or two obstacles: a first batch of two, fours behind it)
            if (beg < end) {
              const int j1 = min(beg + 1, end - 1);
              const float ox0 = obx[beg], oy0 = oby[beg], ox1 = obx[j1], oy1 = oby[j1];
              const double dx0 = static_cast<double>(ox0 - x), dy0 = static_cast<double>(oy0 - y);
              const double dx1 = static_cast<double>(ox1 - x), dy1 = static_cast<double>(oy1 - y);
              const double d0 = dx0 * dx0 + dy0 * dy0, d1 = dx1 * dx1 + dy1 * dy1;
              best = __builtin_fmin(d0, best);  // (NaN distances never win either way)
              best = __builtin_fmin(d1, best);
            }
            for (int jb = beg + 2; jb < end; jb += 4) {
              float ox[4], oy[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int j = min(jb + u, end - 1);  // repeats of the last one change nothing
                ox[u] = obx[j];
                oy[u] = oby[j];
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const double dx = static_cast<double>(ox[u] - x);
                const double dy = static_cast<double>(oy[u] - y);
                const double dd = dx * dx + dy * dy;
                best = __builtin_fmin(dd, best);
              }
            }
            if (!inner) break;
            beg = cells[row * b.W + min(cx + pm, x1) + 1];
            end = cells[row * b.W + x1 + 1];
          }
        }
        atomicMin(obest, static_cast<unsigned long long>(__double_as_longlong(best)));
      }
      // (all lanes: the LDS queue of a wavefront is in order, the read sees every lane's minimum)
      const double shared = __longlong_as_double(static_cast<long long>(
          *const_cast<volatile unsigned long long *>(obest)));
      if (active) {
        // every obstacle closer than `reach` (true distance) was visited
        const double reach = (static_cast<double>(m) - off) * b.g;
        bool done = m >= mmax;  // whole grid visited
        if (reach > 0.0) {
          const double r2 = reach * reach * (1.0 - 1e-6);
          if (shared < r2) done = true;
          if (reach >= b.cap) done = true;
        }
        if (done) {
          active = false;
        } else {
          // next half-width: enough cells to cover sqrt(shared) (+ guard),
          // or the cap radius when nothing has been found yet (any
          // over-estimate only visits more cells: float sqrt is enough)
          const double need =
              shared < DBL_MAX ? static_cast<double>(__builtin_sqrtf(static_cast<float>(shared)) * 1.0001f)
                               : b.cap;
          const double mm = ceil(fmin(need, b.cap * 1.001) * b.inv_g + off) + 1.0;
          pm = m;
          m = max(m + 1, static_cast<int>(fmin(mm, static_cast<double>(mmax))));
        }
      }
    }
    // Far obstacles (points with an empty neighbourhood of kCoopMinSkip
    // cells): a private ring walk per lane is long and mostly wasted,
    // because only the trajectory minimum counts and the distance to the
    // obstacle set is 1-Lipschitz along the trajectory.  After the near
    // points have left their distances in the shared bound, the wavefront
    // evaluates the far points one at a time TOGETHER (ring rows over the
    // lanes), always the one with the smallest lower bound, and every exact
    // distance raises the lower bounds of the others by the triangle
    // inequality; points whose bound exceeds the best distance found are
    // never evaluated.  The values that survive are exact, so the minimum
    // is the one of the full scan.
    if (__ballot(far)) {
      // lower bound of this lane's distance (cells nearer than sk are empty;
      // the centre table when there is one)
      double lbk = lb0;
      {
        // the smallest upper bound bounds the trajectory minimum: points whose
        // lower bound lies above it are never evaluated
        const double u = wave_min_nonneg(far ? ubp : DBL_MAX);
        if (u < 1.0e150) ubound2 = fmin(ubound2, u * u);
      }
      for (int guard = 0; guard < 64; ++guard) {
        const double ub2 = fmin(ubound2, __longlong_as_double(static_cast<long long>(
            *const_cast<volatile unsigned long long *>(obest))));
        // lanes that can still lower the minimum
        const bool cont = far && lbk * lbk < ub2 * (1.0 - 1e-6) && lbk < b.cap;
        const unsigned long long cm = __ballot(cont);
        if (cm == 0ull) break;
        // the one with the smallest lower bound (float key, ties by lane)
        const uint32_t key = cont ? __float_as_uint(static_cast<float>(lbk)) : 0xFFFFFFFFu;
        const uint32_t kmin = wave_min_u32(key);
        const int q = __ffsll(static_cast<long long>(__ballot(cont && key == kmin))) - 1;
        const float xq = lane_value(x, q), yq = lane_value(y, q);
        const int cxq = __builtin_amdgcn_readlane(cx, q), cyq = __builtin_amdgcn_readlane(cy, q);
        const int skq = __builtin_amdgcn_readlane(sk, q);
        const double offq = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(off), q),
                                             __builtin_amdgcn_readlane(__double2loint(off), q));
        // exact search for (xq, yq): ring rows over the lanes; with a bound on
        // the answer the first block is already the one that proves it
        int pmq = skq - 1, mq = max(1, skq);
        if (ub2 < 1.0e300) {
          const double need0 = static_cast<double>(__builtin_sqrtf(static_cast<float>(ub2)) * 1.0001f);
          const double mm0 = ceil(fmin(need0, b.cap * 1.001) * b.inv_g + offq) + 1.0;
          mq = max(mq, static_cast<int>(fmin(mm0, static_cast<double>(mmax))));
        }
        double found = DBL_MAX;   // wave-uniform after every stage
        double proven = 0.0;      // everything closer than this was visited
        for (;;) {
          const int y0 = max(cyq - mq, 0), y1 = min(cyq + mq, b.H - 1);
          const int x0 = max(cxq - mq, 0), x1 = min(cxq + mq, b.W - 1);
          double part = DBL_MAX;
          auto eval = [&](int j) {
            const double dx = static_cast<double>(obx[j] - xq);
            const double dy = static_cast<double>(oby[j] - yq);
            const double dd = dx * dx + dy * dy;
            part = dd < part ? dd : part;
          };
          for (int row0 = y0; row0 <= y1; row0 += 64) {
            // the (up to two) runs of this lane's row: [b1, e1) and, in rows of the visited square, [b2, e2)
            const int row = row0 + lane;
            int b1 = 0, e1 = 0, b2 = 0, e2 = 0;
            if (row <= y1) {
              const bool inner = pmq >= 0 && row >= cyq - pmq && row <= cyq + pmq;
              if (!inner) {
                b1 = cells[row * b.W + x0];
                e1 = cells[row * b.W + x1 + 1];
              } else {  // rows of the visited square: the two side runs
                const int lb_ = min(x1, cxq - pmq - 1), ra = max(x0, cxq + pmq + 1);
                if (x0 <= lb_) {
                  b1 = cells[row * b.W + x0];
                  e1 = cells[row * b.W + lb_ + 1];
                }
                if (ra <= x1) {
                  b2 = cells[row * b.W + ra];
                  e2 = cells[row * b.W + x1 + 1];
                }
              }
            }
            // A wall seen by a dense scan puts hundreds of points into one row of the block (a room, 1440 /
            // 4096 beams: 216 / 440 us per cfg2-sized cycle with one lane per row): runs beyond kLongRun points
            // are walked by the whole wavefront (107 / 111 us), the others by their lane alone (sparse clutter:
            // runs of one to five points; a lower threshold costs the mid-density costmap scene 10 us)
            constexpr int kLongRun = 16;
            const bool long1 = e1 - b1 > kLongRun, long2 = e2 - b2 > kLongRun;
            if (!long1)
              for (int j = b1; j < e1; ++j) eval(j);
            if (!long2)
              for (int j = b2; j < e2; ++j) eval(j);
            for (int pass = 0; pass < 2; ++pass) {
              unsigned long long lm = __ballot(pass == 0 ? long1 : long2);
              while (lm) {
                const int r = __ffsll(static_cast<long long>(lm)) - 1;
                lm &= lm - 1ull;
                const int rb = __builtin_amdgcn_readlane(pass == 0 ? b1 : b2, r);
                const int re = __builtin_amdgcn_readlane(pass == 0 ? e1 : e2, r);
                for (int j = rb + lane; j < re; j += 64) eval(j);
              }
            }
          }
          const double stage = wave_min_nonneg(part);
          found = stage < found ? stage : found;
          const double sh = found < ub2 ? found : ub2;
          const double reach = (static_cast<double>(mq) - offq) * b.g;
          bool done = mq >= mmax;
          if (reach > 0.0) {
            proven = reach;
            if (sh < reach * reach * (1.0 - 1e-6)) done = true;
            if (reach >= b.cap) done = true;
          }
          if (done) break;
          const double need = sh < DBL_MAX
                                  ? static_cast<double>(__builtin_sqrtf(static_cast<float>(sh)) * 1.0001f)
                                  : b.cap;
          const double mm = ceil(fmin(need, b.cap * 1.001) * b.inv_g + offq) + 1.0;
          pmq = mq;
          mq = max(mq + 1, static_cast<int>(fmin(mm, static_cast<double>(mmax))));
        }
        if (lane == 0)
          atomicMin(obest, static_cast<unsigned long long>(__double_as_longlong(found)));
        // what is now known about the distance of point q: it is `found` when
        // that lies inside the proven radius, at least the proven radius
        // otherwise (the whole grid visited: nothing else exists)
        double dq = kc::dsqrt_rn(found);
        if (!(found < proven * proven) && mq < mmax) dq = proven;
        if (lane == q) far = false;
        // triangle inequality: d(p) >= d(q) - |p - q| (slack for the rounding)
        const double ddx = static_cast<double>(x) - static_cast<double>(xq);
        const double ddy = static_cast<double>(y) - static_cast<double>(yq);
        const double sep = kc::dsqrt_rn(ddx * ddx + ddy * ddy);
        const double lb = dq * (1.0 - 1e-6) - sep * (1.0 + 1e-6) - 1e-9;
        lbk = lb > lbk ? lb : lbk;
      }
    }
  }
}

// The wavefront-per-sample evaluation of ONE sample (one lane per trajectory
// point, tiles of 64 points): see the comment on top of this file.  `seg`
// gives the (x, y, z^2) of a segment point and its accumulated length, `cap` /
// `sup` the chunk capsules [8][nch] and super-chunk spheres [4][nsup], `pts`
// the sample's float points; *obest is a 64-bit LDS word of this wavefront.
// Returns the weighted total (wave-uniform).
// kBatched (sample_cost_batched_kernel): the per-POINT part only -- every point's distance to the segment goes
// to `bs.mind`, what the end point's goal term needs to `bs.xe ...`, the obstacle minimum to *obest -- and the
// per-SAMPLE part (ordered sum, the end point's index, the weighted total) is left to batch_totals, which does
// it for 64 samples at once, a lane a sample.
struct BatchSlot {
  float *mind;                 // [P] this sample's row
  float *xe, *ye;              // end point
  uint32_t *be;                // bits of its minimum squared distance
  unsigned long long *cand;    // the chunks its search scanned
};
template <class Seg, class Pts, bool kBatched = false>
__device__ __forceinline__ float wave_sample_total(const CostArgs &a, const DcArgs &t,
                                                   const Seg &seg, const float *cap, const float *sup,
                                                   float sz_end, const int *cells, const uint8_t *skip,
                                                   const float *obx, const float *oby, const Pts pts,
                                                   int n, int lane, unsigned long long *obest,
                                                   bool stamp, const BatchSlot bs = BatchSlot{}) {
  if (lane == 0) *obest = static_cast<unsigned long long>(__double_as_longlong(DBL_MAX));
  float sum = 0.0f;            // ordered path-cost sum, carried over the point tiles
  float goal = 0.0f, endc = 0.0f;
  double ubound2 = DBL_MAX;    // square of an upper bound of the sample's obstacle distance (not attained)
  for (int p0 = 0; p0 < a.P; p0 += 64) {
    const int pp = p0 + lane;
    const bool live = pp < a.P;
    const int p = live ? pp : a.P - 1;  // idle lanes shadow the last point, write nothing
    const float x = pts.x(p), y = pts.y(p);
    float mind = 0.0f, goal_l = 0.0f, end_l = 0.0f;
    const bool st = stamp && p0 == 0;
    if (st) KC_STAMP(7);
    if (a.use_seg) {
      // Only the END point needs to know WHICH segment point is nearest (goal
      // cost); every point needs the nearest distance.  So the lanes track the
      // minimum alone -- as the bits of the non-negative float: unsigned order =
      // float order, NaN and +inf sit above FLT_MAX and never win, exactly like
      // `dist < minDist` -- and the end point's index is found afterwards by the
      // whole wavefront (lowest index among the minima).
      uint32_t bestb = 0x7F7FFFFFu;  // FLT_MAX
      const bool flat = a.seg_flat != 0;  // every z of the segment is +0: the z terms vanish exactly
      auto d2_bits = [&](int j) {
        const float2 q = seg.xy1(j);
        const float dx = q.x - x;
        const float dy = q.y - y;
        const float xx = dx * dx;
        const float yy = dy * dy;
        if (flat) return __float_as_uint(xx + yy);      // yy + (+0) == yy
        return __float_as_uint(xx + (yy + seg.zz1(j)));  // Eigen order a + (b + c)
      };
      unsigned long long cand = 0ull;
      if (t.near != nullptr) {
        // Near table: the cell of this point names the chunks that can hold its nearest
        // segment point (a contiguous range along the path, conservative for every point of
        // the cell) and a seed, the segment point nearest to the cell centre: an attained
        // upper bound within a fraction of the cell of the answer.  What is left is the
        // capsule test of that range against the seed's distance and the scan of the one
        // or two chunks that pass.  A point outside the table (the host lays it over
        // everything a roll-out can reach) tests every chunk.
        const float fx = (x - t.nx0) * t.ninv, fy = (y - t.ny0) * t.ninv;
        const bool inside = fx >= 0.0f && fy >= 0.0f && fx < static_cast<float>(t.nW) &&
                            fy < static_cast<float>(t.nH);  // NaN: outside
        uint32_t e = static_cast<uint32_t>(a.nch - 1) << 8;   // clo 0, chi nch - 1, seed 0
        if (inside) e = t.near[static_cast<int>(fy) * t.nW + static_cast<int>(fx)];
        bestb = d2_bits(static_cast<int>(e >> 16));
        if (st) KC_STAMP(8);
        const float thr = __builtin_sqrtf(__uint_as_float(bestb)) * 1.0001f;
        const int chi = static_cast<int>((e >> 8) & 0xFFu);
        for (int c = static_cast<int>(e & 0xFFu); c <= chi; ++c)
          if (capsule_may_hold(load_capsule(cap, c), x, y, thr, flat)) cand |= 1ull << c;
      } else {
        const int sup_pts = 8 * a.seg_chunk;
        // (1) the first point of every super-chunk
        for (int s = 0; s < a.nsup; ++s) bestb = min(bestb, d2_bits(s * sup_pts));
        // (2) super-chunks that may hold something at least as close:
        // |q - c| - r <= thr on the squares; 1e-4 relative slack on the bound,
        // 1e-5 on the compared square (NaN compares false: qualifies)
        float thr = __builtin_sqrtf(__uint_as_float(bestb)) * 1.0001f;
        unsigned smask = 0u;
        for (int s = 0; s < a.nsup; ++s) {
          const float dx = sup[s] - x, dy = sup[a.nsup + s] - y;
          float d2 = dx * dx + dy * dy;
          if (!flat) {
            const float dz = sup[2 * a.nsup + s];
            d2 += dz * dz;
          }
          const float lim = thr + sup[3 * a.nsup + s];
          if (!(d2 > lim * lim * 1.00001f)) smask |= 1u << s;
        }
        // ... and whose capsule (chord of the whole super-chunk + largest deviation) does: on a
        // smooth path a far query keeps one or two super-chunks where the spheres keep them all
        {
          const float *sc = sup + 4 * a.nsup;  // [nsup] capsule records
          for (unsigned m = smask; m;) {
            const int s = __ffs(static_cast<int>(m)) - 1;
            m &= m - 1u;
            if (!capsule_may_hold(load_capsule(sc, s), x, y, thr, flat)) smask &= ~(1u << s);
          }
        }
        if (st) KC_STAMP(8);
        // (3) the first points of their other chunks
        for (unsigned m = smask; m;) {
          const int s = __ffs(static_cast<int>(m)) - 1;
          m &= m - 1u;
#pragma unroll
          for (int u = 1; u < 8; ++u) {
            const int c = s * 8 + u;
            bestb = min(bestb, d2_bits(min(c, a.nch - 1) * a.seg_chunk));  // a repeat of the last chunk changes nothing
          }
        }
        // (4) capsule test of their chunks: distance to the chord minus the
        // largest deviation of the chunk's points from it
        thr = __builtin_sqrtf(__uint_as_float(bestb)) * 1.0001f;
        for (unsigned m = smask; m;) {
          const int s = __ffs(static_cast<int>(m)) - 1;
          m &= m - 1u;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int c = min(s * 8 + u, a.nch - 1);
            if (capsule_may_hold(load_capsule(cap, c), x, y, thr, flat)) cand |= 1ull << c;
          }
        }
      }
      if (st) KC_STAMP(9);
      const unsigned long long cand0 = cand;  // (the end point's search below looks at its chunks again)
      // (5) the points of those chunks, two per packed operation
      while (cand) {
        const int c = __ffsll(static_cast<long long>(cand)) - 1;
        cand &= cand - 1ull;
        const int k0 = (c * a.seg_chunk) >> 1;  // (chunks hold an even number of points)
        bestb = pairs_min_bits(seg, k0, k0 + (a.seg_chunk >> 1), x, y, flat, bestb);
      }
      if (st) KC_STAMP(10);
      const float best = __uint_as_float(bestb);
      mind = kc::sqrt_rn(best);
      if (kBatched) {
        if (pp == a.P - 1) {
          *bs.xe = x;
          *bs.ye = y;
          *bs.be = bestb;
          *bs.cand = cand0;
        }
      } else if (p0 + 64 >= a.P) {
        // The end point (lane a.P - 1 - p0 of this tile): goalCostFunc,
        // cost_evaluator.cpp:157-176.  Its nearest segment point = the LOWEST
        // index whose squared distance equals the minimum found above (the
        // reference's strict `<` in index order; index 0 when nothing is below
        // FLT_MAX), looked for by all lanes, 64 segment points per step.
        const int le = a.P - 1 - p0;
        const float xe = lane_value(x, le), ye = lane_value(y, le);
        const uint32_t be = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(bestb), le));
        // Points as close as the minimum lie in the chunks the end point scanned (everything else was
        // proven farther): eight lanes per candidate chunk, eight chunks per pass.
        unsigned long long ce =
            (static_cast<unsigned long long>(static_cast<uint32_t>(
                 __builtin_amdgcn_readlane(static_cast<int>(cand0 >> 32), le))) << 32) |
            static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cand0), le));
        uint32_t arg_l = 0xFFFFFFFFu;
        if (be < 0x7F7FFFFFu) {
          const int hp = a.seg_chunk >> 1;
          while (ce) {
            int myc = -1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              if (ce) {
                const int c = __ffsll(static_cast<long long>(ce)) - 1;
                ce &= ce - 1ull;
                if ((lane >> 3) == i) myc = c;
              }
            }
            if (myc >= 0) {
              const int k0 = (myc * a.seg_chunk) >> 1;
              for (int k = k0 + (lane & 7); k < k0 + hp && 2 * k < a.S && arg_l == 0xFFFFFFFFu; k += 8) {
                const f32x2 d = pair_d2(seg, k, xe, ye, flat);
                if (__float_as_uint(d.x) == be) arg_l = static_cast<uint32_t>(2 * k);
                else if (__float_as_uint(d.y) == be) arg_l = static_cast<uint32_t>(2 * k + 1);
              }
            }
          }
        }
        const uint32_t argm = wave_min_u32(arg_l);
        const int arg = argm == 0xFFFFFFFFu ? 0 : static_cast<int>(argm);
        if (pp == a.P - 1) {
          const float acc_arg = seg.acc(arg);
          const float arc = kc::div_rn(a.ref_len - acc_arg, a.ref_len);
          goal_l = arc + kc::div_rn(mind, a.ref_len);
          // end-point term of pathCostFunc, cost_evaluator.cpp:131-136
          const int e = a.S - 1;
          const float2 qe = seg.xy1(e);
          const float dx = x - qe.x, dy = y - qe.y, dz = 0.0f - sz_end;
          const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
          end_l = kc::div_rn(kc::sqrt_rn(xx + (yy + zz)), a.seg_len);
        }
      }
    }
    if (st) KC_STAMP(11);
    wave_obstacle_term(a, t, cells, skip, obx, oby, x, y, live, lane, obest, ubound2);
    if (st) KC_STAMP(12);
    // ordered path-cost sum of this tile (pathCostFunc, cost_evaluator.cpp:111-141)
    if (kBatched) {
      if (live && a.use_seg) bs.mind[pp] = mind;
    } else if (a.use_seg) {
      // (idle lanes contribute +0.0f, which leaves the non-negative sum as it is:
      // 64 straight-line additions instead of a counted loop)
      const float mv = live ? mind : 0.0f;
      sum = wave_ordered_sum(sum, mv, lane);
      if (p0 + 64 >= a.P) {
        goal = lane_value(goal_l, a.P - 1 - p0);
        endc = lane_value(end_l, a.P - 1 - p0);
      }
    }
  }
  if (kBatched) return 0.0f;
  // ---- weighted total (uniform over the wavefront) ---------------------------
  float total = 0.0f;
  if (a.ref_len > 0.0f) {
    if (a.w_goal > 0.0) total = accum(total, a.w_goal, goal);
    if (a.w_path > 0.0) {
      const float c = kc::div_rn(
          kc::div_rn(sum, static_cast<float>(a.P)) + endc, 2.0f);
      total = accum(total, a.w_path, c);
    }
  }
  if (a.O > 0 && a.w_obs > 0.0)
    total = accum(total, a.w_obs,
                  obstacle_cost_from(a, __longlong_as_double(static_cast<long long>(
                                            *const_cast<volatile unsigned long long *>(obest)))));
  if (a.have_vel && !a.defer_vel) total = add_velocity_costs(a, n, total, lane);
  else if (!a.have_vel && a.frz_smooth) total = add_frozen_costs(a, n, total);
  // constant-velocity samples: both terms are exactly 0 and `total += w*0`
  // leaves total unchanged, so nothing to do when !have_vel.
  return total;
}

// ---- the per-sample part of 64 samples at once (sample_cost_batched_kernel) --------------------------------
// One LDS buffer of the batched kernel: what wave_sample_total<kBatched> left for the samples of a group.
struct BatchBuf {
  unsigned long long *obest, *cand;  // [64]
  float *xe, *ye;                    // [64]
  uint32_t *be;                      // [64]
  int *n;                            // [64] sample ids
  float *mind;                       // [64][Pp], Pp odd: a lane a row, conflict-free columns
};
__host__ __device__ inline size_t batch_buf_bytes(int P) { return 2048 + 256 * static_cast<size_t>(P | 1); }
__device__ __forceinline__ BatchBuf batch_buf_at(unsigned char *p) {
  BatchBuf B;
  B.obest = reinterpret_cast<unsigned long long *>(p);
  B.cand = B.obest + 64;
  B.xe = reinterpret_cast<float *>(B.cand + 64);
  B.ye = B.xe + 64;
  B.be = reinterpret_cast<uint32_t *>(B.ye + 64);
  B.n = reinterpret_cast<int *>(B.be + 64);
  B.mind = reinterpret_cast<float *>(B.n + 64);
  return B;
}
// A lane a sample: the ordered sum of its row (the additions of pathCostFunc in the reference's order,
// cost_evaluator.cpp:111-141 -- P of them for 64 samples where the wavefront-per-sample form spends 64 per
// sample), the end point's nearest segment index (the lowest index among the minima of the chunks its search
// scanned, goalCostFunc :157-176), the weighted total in the reference's accumulation order (:59-100), the
// cost and the key.  Samples with caller-provided velocity profiles need their sums precomputed (vsum_*).
template <class Seg>
__device__ __forceinline__ long long batch_totals(const CostArgs &a, const Seg &seg, float sz_end, const BatchBuf &B,
                                                  int size, int lane) {
  if (lane >= size) return KEY_NONE;
  const int Pp = a.P | 1;
  const int n = B.n[lane];
  float sum = 0.0f, goal = 0.0f, endc = 0.0f;
  if (a.use_seg) {
    const float *row = B.mind + lane * Pp;
    {  // (eight loads in flight: the chain is the additions, not the LDS round trips)
      int p = 0;
      for (; p + 8 <= a.P; p += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = row[p + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) sum = sum + v[u];
      }
      for (; p < a.P; ++p) sum = sum + row[p];
    }
    const float xe = B.xe[lane], ye = B.ye[lane];
    const uint32_t be = B.be[lane];
    unsigned long long ce = B.cand[lane];
    const bool flat = a.seg_flat != 0;
    uint32_t arg = 0xFFFFFFFFu;
    if (be < 0x7F7FFFFFu) {
      const int hp = a.seg_chunk >> 1;
      while (ce && arg == 0xFFFFFFFFu) {
        const int c = __ffsll(static_cast<long long>(ce)) - 1;
        ce &= ce - 1ull;
        const int k0 = (c * a.seg_chunk) >> 1;
        for (int k = k0; k < k0 + hp && 2 * k < a.S && arg == 0xFFFFFFFFu; ++k) {
          const f32x2 d = pair_d2(seg, k, xe, ye, flat);
          if (__float_as_uint(d.x) == be) arg = static_cast<uint32_t>(2 * k);
          else if (__float_as_uint(d.y) == be) arg = static_cast<uint32_t>(2 * k + 1);
        }
      }
    }
    const int argi = arg == 0xFFFFFFFFu ? 0 : static_cast<int>(arg);
    const float mind_e = row[a.P - 1];
    const float arc = kc::div_rn(a.ref_len - seg.acc(argi), a.ref_len);
    goal = arc + kc::div_rn(mind_e, a.ref_len);
    // end-point term of pathCostFunc, cost_evaluator.cpp:131-136
    const float2 qe = seg.xy1(a.S - 1);
    const float dx = xe - qe.x, dy = ye - qe.y, dz = 0.0f - sz_end;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    endc = kc::div_rn(kc::sqrt_rn(xx + (yy + zz)), a.seg_len);
  }
  float total = 0.0f;
  if (a.ref_len > 0.0f) {
    if (a.w_goal > 0.0) total = accum(total, a.w_goal, goal);
    if (a.w_path > 0.0) {
      const float c = kc::div_rn(kc::div_rn(sum, static_cast<float>(a.P)) + endc, 2.0f);
      total = accum(total, a.w_path, c);
    }
  }
  if (a.O > 0 && a.w_obs > 0.0)
    total = accum(total, a.w_obs, obstacle_cost_from(a, __longlong_as_double(static_cast<long long>(B.obest[lane]))));
  if (a.have_vel && !a.defer_vel) {
    const float div = static_cast<float>(3L * (a.P - 1));
    if (a.w_smooth > 0.0) total = accum(total, a.w_smooth, kc::div_rn(a.vsum_smooth[n], div));
    if (a.w_jerk > 0.0) total = accum(total, a.w_jerk, kc::div_rn(a.vsum_jerk[n], div));
  } else if (!a.have_vel && a.frz_smooth) {
    total = add_frozen_costs(a, n, total);
  }
  a.costs[n] = total;
  if (total < FLT_MAX && !a.defer_vel) return key_pack(total, static_cast<uint32_t>(a.first + n));
  return KEY_NONE;
}

template <typename T>
__device__ __forceinline__ void st_agent(T *p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ T ld_agent(const T *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the record of a cycle: minimum of the per-workgroup keys, the reference's compacted index of the
// winner (admissible samples in front of it), the record for the host (pinned memory, polled: no D2H copy,
// no stream wait) and the re-arming of the working slots.  One workgroup: publish_kernel behind the cost
// kernels, or the workgroup of sample_cost_kernel that arrives last (PubArgs::fold).
struct PubArgs {
  const long long *block_keys;
  int nblocks;
  const uint8_t *flags;
  int n, first;
  long long *result;
  long long *host_pub;
  long long seq;
  int identity_n;  // > 0: the admissible count of a batch that had no list (CostArgs::identity_n)
  int fold;        // sample_cost_kernel: 1 = its last workgroup publishes (no publish_kernel behind it)
};
template <int kPubBlock>
__device__ __forceinline__ void publish_body(const PubArgs &a) {
  __shared__ long long wkey[kPubBlock / 64];
  __shared__ int wsum[kPubBlock / 64];
  __shared__ long long s_fkey;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long k = KEY_NONE;
  for (int b = threadIdx.x; b < a.nblocks; b += kPubBlock) {
    const long long v = ld_agent(a.block_keys + b);  // (another workgroup of this kernel may have written it)
    k = v < k ? v : k;
  }
  const long long na = a.identity_n > 0 ? a.identity_n : a.result[W_LIST];
  const long long err = a.result[W_NADM];  // device error word (roll-out gave up waiting)
  for (int off = 32; off > 0; off >>= 1) {
    const long long o = __shfl_xor(k, off, 64);
    k = o < k ? o : k;
  }
  if (lane == 0) wkey[wave] = k;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long m = wkey[0];
    for (int w = 1; w < kPubBlock / 64; ++w) m = wkey[w] < m ? wkey[w] : m;
    s_fkey = m;
  }
  __syncthreads();
  const long long fkey = s_fkey;
  // the reference's index counts the admissible samples in front of the winner
  int cnt = 0;
  if (fkey != KEY_NONE) {
    long long lim =
        static_cast<long long>(static_cast<uint32_t>(fkey & 0xFFFFFFFFll)) - a.first;
    if (lim > a.n) lim = a.n;
    // the flags are 0 / 1 bytes: sixteen per load, eight loads in flight per thread (a 65536-sample
    // lattice is 8 loads per thread; byte by byte this loop was 21 us of a 0.2 ms cycle)
    const int full = static_cast<int>(lim >> 4);
    const uint4 *f16 = reinterpret_cast<const uint4 *>(a.flags);
#pragma unroll 8
    for (int i = threadIdx.x; i < full; i += kPubBlock) {
      const uint4 v = f16[i];
      cnt += __popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) +
             __popc(v.w & 0x01010101u);
    }
    for (long long i = (static_cast<long long>(full) << 4) + threadIdx.x; i < lim; i += kPubBlock)
      cnt += a.flags[i] & 1;
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if (lane == 0) wsum[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < kPubBlock / 64; ++w) s += wsum[w];
    if (fkey == KEY_NONE) s = -1;
    const long long na_pub = err ? -1 : na;
    if (a.host_pub) {
      // zero-copy hand-off: the host polls the record.  No fence between the
      // words: the fourth is a mixing checksum over the others (record_check,
      // kc_internal.h), so a half-arrived record is never accepted.
      const long long w1 = (na_pub << 32) | static_cast<long long>(static_cast<uint32_t>(s));
      store_host_record(a.host_pub, fkey, w1, a.seq, 0);  // (no winner row in this record)
    }
    a.result[R_KEY] = fkey;
    a.result[R_NADM] = na_pub;
    a.result[R_COMPACT] = s;
    a.result[W_KEY] = KEY_NONE;
    a.result[W_NADM] = 0;
    a.result[W_TICKET] = 0;
    a.result[W_LIST] = 0;  // admissible-list counter of the next cycle
  }
}

// kLds: the tracked segment (+ chunk spheres), the bucket cell table and the
// skip table are copied into LDS once per workgroup; kObsLds: the obstacle
// coordinates too.  Otherwise they are read in place.
// kFold: the workgroup that arrives last publishes the cycle (PubArgs, publish_body).  The instance without it
// is the one that runs BESIDE velocity_sums_kernel (caller-provided velocity profiles, run_evaluate): it is held
// to 96 VGPRs (five wavefronts per SIMD's worth) so that its four wavefronts per SIMD leave 128 registers --
// three wavefronts of the sums.  With the publish code inside, the kernel took 104 and left room for two: the
// sums ran behind the cost kernel instead of beside it (CostEvaluator_5k_Trajs 0.198 -> 0.273 ms, round 3).
template <bool kLds, bool kObsLds, bool kFold>
__global__ __launch_bounds__(kCostBlock)
__attribute__((amdgpu_waves_per_eu(kFold ? 4 : 5, kFold ? 4 : 5))) void sample_cost_kernel(CostArgs a_, DcArgs t_, PubArgs pub_) {
  const KernargTriple<CostArgs, DcArgs, PubArgs> *ka_ = kernargs_touched<KernargTriple<CostArgs, DcArgs, PubArgs>>();
  const CostArgs &a = ka_->a;  // (the arguments as read behind the touch of every kernarg line)
  const DcArgs &t = ka_->b;
  const PubArgs &pub = ka_->c;
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ long long s_key;
  __shared__ unsigned long long s_obest[kCostWaves];  // per sample: min squared obstacle distance (double bits)
  __shared__ int s_next;  // next sample slot of this workgroup (the wavefronts pull: samples differ in cost)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform
  KC_STAMP(0);
  const int na = a.identity_n > 0 ? a.identity_n : static_cast<int>(*a.adm_count);
  KC_STAMP(1);
  const BucketDev &b = a.b;
  const int ncell = b.W * b.H;
  const int npp = seg_pairs_padded(a.nch, a.seg_chunk);
  const int seg_words = a.use_seg ? 8 * npp + 8 * a.nch + 12 * a.nsup : 0;
  // LDS layout: segment pair records + chunk capsules + super-chunk spheres | cell table | skip table
  // (padded to words) | obstacle coordinates.  The pointers are chosen at compile time
  // so that the LDS variants issue ds_read, not flat loads.
  float *const l_seg = reinterpret_cast<float *>(smem);
  int *const l_cells = reinterpret_cast<int *>(l_seg + seg_words);
  uint8_t *const l_skip = reinterpret_cast<uint8_t *>(l_cells + (a.use_obs ? ncell + 1 : 0));
  float *const l_obs = reinterpret_cast<float *>(l_skip + (a.use_obs ? ((ncell + 3) & ~3) : 0));
  const int *const cells = kLds ? l_cells : b.cell_start;
  const uint8_t *const skip = kLds ? l_skip : b.skip;
  const bool scan = t.onear != nullptr;  // (the obstacle term reads the scan block, not the buckets' arrays)
  const float *const obx = kObsLds ? l_obs : (scan ? t.osx : b.bx);
  const float *const oby = kObsLds ? l_obs + (scan ? t.on : b.nobs) : (scan ? t.osy : b.by);
  // In LDS the segment points are pair records (struct SegPairs): one 16-byte read per TWO points
  // instead of three reads from three rows per point; the capsules and spheres follow
  float4 *const l_xy = reinterpret_cast<float4 *>(l_seg);
  float4 *const l_za = l_xy + npp;
  const float *const cap = kLds ? l_seg + 8 * npp : a.sx + seg_cap_offset(a.S);  // [nch] capsule records
  const float *const sup = cap + 8 * a.nch;                          // [4][nsup]
  const float sz_end = (a.use_seg && a.S > 0) ? a.sz[a.S - 1] : 0.0f;  // z of the last segment point (end term)
  if (threadIdx.x == 0) {
    s_key = KEY_NONE;
    s_next = 0;
  }
  if (na > 0 && kLds) {
    if (a.use_seg) {
      for (int k = threadIdx.x; k < npp; k += kCostBlock)
        seg_pair_from_rows(a.sx, a.sy, a.szz, a.acc_seg, a.S, k, l_xy[k], l_za[k]);
      float *const wc = l_seg + 8 * npp;
      const float *const gc = a.sx + seg_cap_offset(a.S);
      for (int j = threadIdx.x; j < 8 * a.nch + 12 * a.nsup; j += kCostBlock) wc[j] = gc[j];
    }
    if (a.use_obs) {
#pragma unroll 8
      for (int j = threadIdx.x; j <= ncell; j += kCostBlock) l_cells[j] = b.cell_start[j];
      // the skip table is padded to a multiple of 4 bytes on the host
      const uint32_t *gs = reinterpret_cast<const uint32_t *>(b.skip);
      uint32_t *ls = reinterpret_cast<uint32_t *>(l_skip);
      for (int j = threadIdx.x; j < (ncell + 3) / 4; j += kCostBlock) ls[j] = gs[j];
      if (kObsLds) {
        const float *const src = scan ? t.osx : b.bx;  // bx | by contiguous; osx | osy | chunk boxes
        const int cnt = scan ? scan_block_floats(t.on, t.oscs) : 2 * b.nobs;
#pragma unroll 8
        for (int j = threadIdx.x; j < cnt; j += kCostBlock) l_obs[j] = src[j];
      }
    }
  }
  __syncthreads();
  KC_STAMP(6);

#ifdef KC_PHASE_STAMPS
  bool stamped = false;
#endif
  long long wkey = KEY_NONE;
  // sample i of the list belongs to workgroup (i % grid): a short list spreads
  // over all CUs; inside the workgroup the wavefronts pull the next one
  for (;;) {
    int slot = 0;
    if (lane == 0) slot = atomicAdd(&s_next, 1);
    slot = __builtin_amdgcn_readfirstlane(slot);
    const int i = slot * static_cast<int>(gridDim.x) + static_cast<int>(blockIdx.x);
    if (i >= na) break;
    const int n = a.identity_n > 0 ? i : a.adm_list[i];
    const RowPts pts{a.px + (size_t)n * a.P, a.py + (size_t)n * a.P};
#ifdef KC_PHASE_STAMPS
    const bool stamp = a.dbg && wave == 0 && !stamped;  // first sample of wavefront 0
    stamped = true;
#else
    constexpr bool stamp = false;
#endif
    float total;
    if (kLds)
      total = wave_sample_total(a, t, SegPairs{l_xy, l_za}, cap, sup, sz_end, cells, skip, obx, oby, pts,
                                n, lane, &s_obest[wave], stamp);
    else
      total = wave_sample_total(a, t, SegRows{a.sx, a.sy, a.szz, a.acc_seg, a.S}, cap, sup, sz_end,
                                cells, skip, obx, oby, pts, n, lane, &s_obest[wave], stamp);
    if (lane == 0) a.costs[n] = total;
    if (stamp) KC_STAMP(3);
    if (total < FLT_MAX && !a.defer_vel) {  // `total_cost < minCost`, minCost starts at FLT_MAX
      const long long k = key_pack(total, static_cast<uint32_t>(a.first + n));
      wkey = k < wkey ? k : wkey;
    }
  }
  KC_STAMP(2);
  // ---- the workgroup's best key, for publish_kernel ----------------------------
  if (lane == 0 && wkey != KEY_NONE) atomicMin(&s_key, wkey);
  __syncthreads();
  if (!kFold) {
    if (threadIdx.x == 0) a.block_keys[blockIdx.x] = s_key;
    KC_STAMP(4);
    return;
  }
  // Arrival ticket: the workgroup that arrives last publishes the cycle (the workgroups finish microseconds
  // apart -- the atomics do not meet -- and the record is out one dispatch + one kernel earlier).
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    st_agent(a.block_keys + blockIdx.x, s_key);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tk = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a.result + W_TICKET), 1ull,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (tk == static_cast<unsigned long long>(gridDim.x) - 1ull) ? 1 : 0;
  }
  __syncthreads();
  KC_STAMP(4);
  if (s_last) publish_body<kCostBlock>(pub);
}

// sample_cost_kernel with the per-sample part batched (the long lists of the DWA cycle: no caller-provided
// velocity profiles, or their sums precomputed).  A wavefront still takes one sample at a time for the
// per-point work (segment search, obstacle term), but leaves the results in a slot of an LDS buffer of 64
// samples; the wavefront that completes a buffer runs batch_totals on it -- a lane a sample -- while the others
// go on with the next buffer (two buffers; a wavefront that would write into a buffer whose previous round has
// not been consumed waits for it).  The per-sample tail of the wavefront-per-sample form -- 64 dependent adds,
// the end point's index search, four correctly rounded divisions, the weighted total: ~310 of ~900 wave
// instructions per sample -- becomes ~7.  No barrier between the groups.
template <bool kObsLds>
__global__ __launch_bounds__(kCostBlock) void sample_cost_batched_kernel(CostArgs a_, DcArgs t_, PubArgs pub_) {
  const KernargTriple<CostArgs, DcArgs, PubArgs> *ka_ = kernargs_touched<KernargTriple<CostArgs, DcArgs, PubArgs>>();
  const CostArgs &a = ka_->a;  // (the arguments as read behind the touch of every kernarg line)
  const DcArgs &t = ka_->b;
  const PubArgs &pub = ka_->c;
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ long long s_key;
  __shared__ int s_next;      // next sample slot of this workgroup
  __shared__ int s_done[2];   // finished slots of the group that owns the buffer
  __shared__ int s_freed[2];  // rounds of the buffer whose totals are out
  const int lane = threadIdx.x & 63;
  KC_STAMP(0);
  const int na = a.identity_n > 0 ? a.identity_n : static_cast<int>(*a.adm_count);
  KC_STAMP(1);
  const BucketDev &b = a.b;
  const int ncell = b.W * b.H;
  const int npp = seg_pairs_padded(a.nch, a.seg_chunk);
  const int seg_words = a.use_seg ? 8 * npp + 8 * a.nch + 12 * a.nsup : 0;
  // LDS: two sample buffers | segment pair records + capsules + spheres | cell table | skip table | obstacles
  const size_t bufb = batch_buf_bytes(a.P);
  float *const l_seg = reinterpret_cast<float *>(smem + 2 * bufb);
  int *const l_cells = reinterpret_cast<int *>(l_seg + seg_words);
  uint8_t *const l_skip = reinterpret_cast<uint8_t *>(l_cells + (a.use_obs ? ncell + 1 : 0));
  float *const l_obs = reinterpret_cast<float *>(l_skip + (a.use_obs ? ((ncell + 3) & ~3) : 0));
  const int *const cells = l_cells;
  const uint8_t *const skip = l_skip;
  const bool scan = t.onear != nullptr;  // (the obstacle term reads the scan block, not the buckets' arrays)
  const float *const obx = kObsLds ? l_obs : (scan ? t.osx : b.bx);
  const float *const oby = kObsLds ? l_obs + (scan ? t.on : b.nobs) : (scan ? t.osy : b.by);
  float4 *const l_xy = reinterpret_cast<float4 *>(l_seg);
  float4 *const l_za = l_xy + npp;
  const float *const cap = l_seg + 8 * npp;
  const float *const sup = cap + 8 * a.nch;
  const float sz_end = (a.use_seg && a.S > 0) ? a.sz[a.S - 1] : 0.0f;
  if (threadIdx.x == 0) {
    s_key = KEY_NONE;
    s_next = 0;
    s_done[0] = s_done[1] = 0;
    s_freed[0] = s_freed[1] = 0;
  }
  if (na > 0) {
    if (a.use_seg) {
      for (int k = threadIdx.x; k < npp; k += kCostBlock)
        seg_pair_from_rows(a.sx, a.sy, a.szz, a.acc_seg, a.S, k, l_xy[k], l_za[k]);
      float *const wc = l_seg + 8 * npp;
      const float *const gc = a.sx + seg_cap_offset(a.S);
      for (int j = threadIdx.x; j < 8 * a.nch + 12 * a.nsup; j += kCostBlock) wc[j] = gc[j];
    }
    if (a.use_obs) {
#pragma unroll 8
      for (int j = threadIdx.x; j <= ncell; j += kCostBlock) l_cells[j] = b.cell_start[j];
      const uint32_t *gs = reinterpret_cast<const uint32_t *>(b.skip);
      uint32_t *ls = reinterpret_cast<uint32_t *>(l_skip);
      for (int j = threadIdx.x; j < (ncell + 3) / 4; j += kCostBlock) ls[j] = gs[j];
      if (kObsLds) {
        const float *const src = scan ? t.osx : b.bx;  // bx | by contiguous; osx | osy | chunk boxes
        const int cnt = scan ? scan_block_floats(t.on, t.oscs) : 2 * b.nobs;
#pragma unroll 8
        for (int j = threadIdx.x; j < cnt; j += kCostBlock) l_obs[j] = src[j];
      }
    }
  }
  __syncthreads();
  KC_STAMP(6);
  // sample i of the list belongs to workgroup (i % grid); this workgroup has M of them, in groups of 64 slots
  const int G = static_cast<int>(gridDim.x);
  const int M = static_cast<int>(blockIdx.x) < na ? (na - static_cast<int>(blockIdx.x) + G - 1) / G : 0;
  const SegPairs seg{l_xy, l_za};
  long long wkey = KEY_NONE;
  for (;;) {
    int slot = 0;
    if (lane == 0) slot = atomicAdd(&s_next, 1);
    slot = __builtin_amdgcn_readfirstlane(slot);
    if (slot >= M) break;
    const int g = slot >> 6, q = slot & 63, sel = g & 1, round = g >> 1;
    const int i = slot * G + static_cast<int>(blockIdx.x);
    const int n = a.identity_n > 0 ? i : a.adm_list[i];
    const RowPts pts{a.px + (size_t)n * a.P, a.py + (size_t)n * a.P};
    // (the totals of the group that had this buffer two groups ago are out)
    while (__hip_atomic_load(&s_freed[sel], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < round)
      __builtin_amdgcn_s_sleep(1);
    const BatchBuf B = batch_buf_at(smem + sel * bufb);
    if (lane == 0) B.n[q] = n;
    const BatchSlot bs{B.mind + q * (a.P | 1), B.xe + q, B.ye + q, B.be + q, B.cand + q};
    wave_sample_total<SegPairs, RowPts, true>(a, t, seg, cap, sup, sz_end, cells, skip, obx, oby, pts, n, lane,
                                              B.obest + q, false, bs);
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&s_done[sel], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    const int size = min(64, M - 64 * g);
    if (old + 1 == size) {  // this wavefront completed the group: its totals
      long long k = batch_totals(a, seg, sz_end, B, size, lane);
      for (int off = 32; off > 0; off >>= 1) {
        const long long o = __shfl_xor(k, off, 64);
        k = o < k ? o : k;
      }
      wkey = k < wkey ? k : wkey;
      if (lane == 0) {
        __hip_atomic_store(&s_done[sel], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&s_freed[sel], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  KC_STAMP(2);
  if (lane == 0 && wkey != KEY_NONE) atomicMin(&s_key, wkey);
  __syncthreads();
  if (!pub.fold) {
    if (threadIdx.x == 0) a.block_keys[blockIdx.x] = s_key;
    KC_STAMP(4);
    return;
  }
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    st_agent(a.block_keys + blockIdx.x, s_key);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tk = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a.result + W_TICKET), 1ull,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (tk == static_cast<unsigned long long>(gridDim.x) - 1ull) ? 1 : 0;
  }
  __syncthreads();
  KC_STAMP(4);
  if (s_last) publish_body<kCostBlock>(pub);
}

// One workgroup, queued behind sample_cost_kernel: minimum of the per-block
// keys, the reference's compacted index of the winner (admissible samples in
// front of it), the record for the host (pinned memory, polled: no D2H copy, no
// stream wait) and the re-arming of the working slots.  A kernel boundary
// instead of a device-wide "last block" ticket: hundreds of same-address
// atomics cost more than the dispatch of this kernel.
#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(256) void velocity_finish_kernel(VelFinishArgs a) {
  __shared__ long long s_key;
  if (threadIdx.x == 0) s_key = KEY_NONE;
  __syncthreads();
  const int na = a.identity_n > 0 ? a.identity_n : static_cast<int>(*a.adm_count);
  long long wkey = KEY_NONE;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < na; i += gridDim.x * 256) {
    const int n = a.identity_n > 0 ? i : a.adm_list[i];
    float total = a.costs[n];
    if (a.vsum_smooth) total = accum(total, a.w_smooth, kc::div_rn(a.vsum_smooth[n], a.div));
    if (a.vsum_jerk) total = accum(total, a.w_jerk, kc::div_rn(a.vsum_jerk[n], a.div));
    a.costs[n] = total;
    if (total < FLT_MAX) {  // `total_cost < minCost`, minCost starts at FLT_MAX
      const long long k = key_pack(total, static_cast<uint32_t>(a.first + n));
      wkey = k < wkey ? k : wkey;
    }
  }
  if (wkey != KEY_NONE) atomicMin(&s_key, wkey);
  __syncthreads();
  if (threadIdx.x == 0) a.block_keys[blockIdx.x] = s_key;
}
#endif  // KC_TU_CYCLE

constexpr int kPubBlock = 512;
#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(kPubBlock) void publish_kernel(PubArgs a) { publish_body<kPubBlock>(a); }
#endif  // KC_TU_CYCLE

// the (externally reduced) device record again into the pinned mirror
#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ void republish_kernel(const long long *result, long long *host_pub, long long seq) {
  const long long key = result[R_KEY];
  const long long w1 = (result[R_NADM] << 32) |
                       static_cast<long long>(static_cast<uint32_t>(result[R_COMPACT]));
  store_host_record(host_pub, key, w1, seq, 0);
}
#endif  // KC_TU_CYCLE

// ordered compaction of the admissible flags (one workgroup): adm_list[i] =
// i-th admissible local sample id, *adm_count = how many.  Used by the split
// roll-out path and by kc_cost_evaluate (the fused kernel appends to the list
// itself).  Every thread owns a contiguous chunk (all its flags are requested
// up front: one memory latency), a block-wide scan gives the offsets.
constexpr int kCompactMaxPer = 64;  // 1024 threads x 64 = 65536 samples

#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(1024) void compact_kernel(
    const uint8_t *__restrict__ flags, int n, int *__restrict__ adm_list,
    long long *__restrict__ adm_count) {
  __shared__ int wave_tot[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = (n + 1023) / 1024;  // <= kCompactMaxPer (checked on the host)
  const int i0 = threadIdx.x * per;
  unsigned long long bits = 0;  // per <= 64 flags of this thread
  for (int k = 0; k < per; ++k) {
    const int i = i0 + k;
    if (i < n && flags[i] != 0) bits |= 1ull << k;
  }
  const int mine = __popcll(bits);
  int incl = mine;
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int woff = 0, tot = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) woff += wave_tot[w];
    tot += wave_tot[w];
  }
  int dst = woff + incl - mine;
  while (bits) {
    const int k = __ffsll(static_cast<long long>(bits)) - 1;
    bits &= bits - 1;
    adm_list[dst++] = i0 + k;
  }
  if (threadIdx.x == 0) *adm_count = tot;
}
#endif  // KC_TU_CYCLE

// admissible samples in front of a raw index (multi-GPU: rebuilds the
// reference's compacted index across shards).  One workgroup.
#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(1024) void count_before_kernel(
    const uint8_t *__restrict__ flags, int n, int first, long long target_raw,
    long long *result, int slot) {
  long long lim = target_raw - first;  // local bound
  if (lim > n) lim = n;
  int c = 0;
  for (long long i = threadIdx.x; i < lim; i += 1024) c += flags[i];
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  __shared__ int wsum[16];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < 16; ++w) s += wsum[w];
    result[slot] = s;
  }
}
#endif  // KC_TU_CYCLE

// arms the result record (context creation, and the empty-batch case)
#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ void init_result_kernel(long long *result) {
  result[R_KEY] = KEY_NONE;
  result[R_NADM] = 0;
  result[R_COMPACT] = -1;
  result[R_SPARE] = 0;
  result[W_KEY] = KEY_NONE;
  result[W_NADM] = 0;
  result[W_TICKET] = 0;
  result[W_LIST] = 0;
  result[R_SCRATCH] = 0;
  result[R_TRIGSEQ] = 0;
}
#endif  // KC_TU_CYCLE

// Bounding box of caller-provided sample points (kc_cost_upload): min / max of the finite x and y as
// order-preserving unsigned keys (atomicMin / atomicMax), out = {min x, min y, max x, max y}, armed by
// the host with {~0, ~0, 0, 0}; a non-finite coordinate sets out[4] (the box is not used then).
__device__ __forceinline__ unsigned int float_order_key(float v) {
  const unsigned int b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(256) void bbox_kernel(const float *px, const float *py, size_t count, unsigned int *out) {
  unsigned int lo_x = 0xFFFFFFFFu, lo_y = 0xFFFFFFFFu, hi_x = 0u, hi_y = 0u, bad = 0u;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const float x = px[i], y = py[i];
    if (!(fabsf(x) <= FLT_MAX) || !(fabsf(y) <= FLT_MAX)) {
      bad = 1u;
      continue;
    }
    const unsigned int kx = float_order_key(x), ky = float_order_key(y);
    lo_x = min(lo_x, kx);
    hi_x = max(hi_x, kx);
    lo_y = min(lo_y, ky);
    hi_y = max(hi_y, ky);
  }
  for (int off = 32; off > 0; off >>= 1) {
    lo_x = min(lo_x, static_cast<unsigned int>(__shfl_xor(static_cast<int>(lo_x), off, 64)));
    lo_y = min(lo_y, static_cast<unsigned int>(__shfl_xor(static_cast<int>(lo_y), off, 64)));
    hi_x = max(hi_x, static_cast<unsigned int>(__shfl_xor(static_cast<int>(hi_x), off, 64)));
    hi_y = max(hi_y, static_cast<unsigned int>(__shfl_xor(static_cast<int>(hi_y), off, 64)));
    bad |= static_cast<unsigned int>(__shfl_xor(static_cast<int>(bad), off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(out + 0, lo_x);
    atomicMin(out + 1, lo_y);
    atomicMax(out + 2, hi_x);
    atomicMax(out + 3, hi_y);
    if (bad) atomicOr(out + 4, 1u);
  }
}
#endif  // KC_TU_CYCLE

#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ void fill_u8_kernel(uint8_t *p, int n, uint8_t v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
#endif  // KC_TU_CYCLE

}  // namespace kc
