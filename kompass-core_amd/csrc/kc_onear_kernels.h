// Near table of a laser scan's obstacles (DcArgs::onear; consumer: wave_sample_total in kc_cost_kernels.h).
// The obstacle term of the cost (obstaclesDistCostFunc, cost_evaluator.cpp:179-184, over
// TrajectoryPath::minDist2D, trajectory.h:218-235) needs the minimum distance between P trajectory points and
// O obstacles.  In a room every trajectory point is metres away from hundreds of scan points: the bucket ring
// search walks large blocks for every sample (72 us of a 107 us cycle at 1440 beams).  Consecutive beams are a
// polyline, so `cs` consecutive obstacles (at most 64 chunks) have a tight bounding box, and for a grid of
// W x H cells of edge g over the box a roll-out can reach this kernel finds, per cell with centre c:
//   m      the distance from c to the nearest obstacle (float; only bounds come out of it) and an obstacle
//          that attains it (the seed),
//   mask   the chunks whose box comes within m + 2 h of c (h = half a cell diagonal + the slack of the
//          consumer's float cell arithmetic): for a point p of the cell the seed is at most m + h away and
//          every obstacle of a chunk outside the mask is farther than (m + 2 h) - h -- the mask holds p's
//          nearest obstacle and every tie,
//   floor  max(m - h, 0) * 0.9999: no point of the cell is closer to any obstacle.
// Cells whose floor reaches max_obstacles_dist cost nothing: empty mask.  Eight lanes per cell (chunk k
// belongs to lane k mod 8); the obstacle coordinates are staged in LDS when they fit.  Part of kc_dwa.hip.
#pragma once

namespace kc {

struct ObsNearArgs {
  const float *osx, *osy;  // [n] obstacle coordinates in scan order
  const float *aabb;       // [4][64] xmin | xmax | ymin | ymax per chunk (+inf / -inf for the unused ones)
  int n, cs, nch;
  float x0, y0, g;         // origin, cell edge
  float slack;             // added to half a cell diagonal
  float cap;               // max_obstacles_dist
  int W, H;
  uint4 *out;
};
constexpr int kObsNearBlock = 512;  // 64 cells per workgroup
constexpr int kObsNearLanes = 8;
constexpr size_t kObsNearLdsMax = 48 * 1024;

template <bool kLds, int kBlock>
__device__ __forceinline__ void obs_near_body(const ObsNearArgs &a, int block, unsigned char *smem) {
  __shared__ float l_box[256];
  float *lx = reinterpret_cast<float *>(smem), *ly = lx + a.n;
  for (int j = threadIdx.x; j < 256; j += kBlock) l_box[j] = a.aabb[j];
  if (kLds)
    for (int j = threadIdx.x; j < a.n; j += kBlock) {
      lx[j] = a.osx[j];
      ly[j] = a.osy[j];
    }
  __syncthreads();
  const float *ox = kLds ? lx : a.osx, *oy = kLds ? ly : a.osy;
  constexpr int kL = kObsNearLanes, kPer = 64 / kL;
  const int sub = threadIdx.x & (kL - 1);
  const int cell = block * (kBlock / kL) + threadIdx.x / kL;
  const int ncell = a.W * a.H;
  const int cc = min(cell, ncell - 1);  // whole groups stay in step (DPP reductions)
  const int ix = cc % a.W, iy = cc / a.W;
  const float x = a.x0 + (static_cast<float>(ix) + 0.5f) * a.g;
  const float y = a.y0 + (static_cast<float>(iy) + 0.5f) * a.g;
  const float mag = fabsf(x) + fabsf(y);
  auto d2_to = [&](int j) {
    const float dx = ox[j] - x, dy = oy[j] - y;
    return dx * dx + dy * dy;
  };
  // (1) this lane's chunks: lower bound from the box, upper bound from the chunk's first obstacle
  float lb[kPer];
  uint32_t ub = 0x7F7FFFFFu;
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int k = sub + kL * u;
    lb[u] = __builtin_inff();
    if (k < a.nch) {
      const float gx = fmaxf(fmaxf(l_box[k] - x, x - l_box[64 + k]), 0.0f);
      const float gy = fmaxf(fmaxf(l_box[128 + k] - y, y - l_box[192 + k]), 0.0f);
      float v = __builtin_sqrtf(gx * gx + gy * gy) * 0.9999f - 4e-7f * (mag + fabsf(l_box[k]) + fabsf(l_box[128 + k]));
      if (!(v == v)) v = -__builtin_inff();  // NaN: always a candidate
      lb[u] = v;
      ub = min(ub, __float_as_uint(d2_to(k * a.cs)));  // (NaN / inf bits never win)
    }
  }
  ub = group_min_u32<kL>(ub);
  const float h = a.g * 0.70710679f * 1.0001f + a.slack;
  {
    // Every chunk's box beyond max_obstacles_dist + h of the centre: the cell costs nothing, whatever its nearest
    // obstacle is -- no scan.  (Round 4: a room wider than the cap around the robot is nothing but such cells, and the
    // scans of step (2) -- every chunk of a round wall is as near as every other -- made the sensor launch that carries
    // this table 31 us instead of 12.)
    float lmin = __builtin_inff();
#pragma unroll
    for (int u = 0; u < kPer; ++u) lmin = fminf(lmin, fmaxf(lb[u], 0.0f));  // (a NaN box: lb = -inf -> 0: never early)
    const float lall = __uint_as_float(group_min_u32<kL>(__float_as_uint(lmin)));
    const float fl0 = (lall - h) * 0.9999f - 4e-7f * mag;
    if (fl0 >= a.cap) {
      if (sub == 0 && cell < ncell) a.out[cell] = make_uint4(0u, 0u, 0xFFFFFFFFu, __float_as_uint(fl0));
      return;
    }
  }
  const float uthr = __builtin_sqrtf(__uint_as_float(ub)) * 1.0001f;
  // (2) the chunks that may hold something as close as that are scanned by the whole group: m and a seed
  uint32_t qlo = 0u, qhi = 0u;
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int k = sub + kL * u;
    if (k < a.nch && lb[u] <= uthr) {
      if (k < 32) qlo |= 1u << k;
      else qhi |= 1u << (k - 32);
    }
  }
  qlo = group_or_u32<kL>(qlo);
  qhi = group_or_u32<kL>(qhi);
  uint32_t mb = ub, jb = 0xFFFFFFFFu;
  for (unsigned long long q = (static_cast<unsigned long long>(qhi) << 32) | qlo; q;) {  // uniform in the group
    const int k = __ffsll(static_cast<long long>(q)) - 1;
    q &= q - 1ull;
    const int j1 = min((k + 1) * a.cs, a.n);
    for (int j = k * a.cs + sub; j < j1; j += kL) {
      const uint32_t b = __float_as_uint(d2_to(j));
      if (b < mb || (b == mb && static_cast<uint32_t>(j) < jb)) {
        mb = b;
        jb = static_cast<uint32_t>(j);
      }
    }
  }
  const uint32_t mg = group_min_u32<kL>(mb);
  const uint32_t jg = group_min_u32<kL>(mb == mg ? jb : 0xFFFFFFFFu);
  // (3) the chunks within m + 2 h, the floor of the cell
  const float m = __builtin_sqrtf(__uint_as_float(mg));  // (+inf-ish when nothing is finite)
  const float R = m * 1.0001f + 2.0f * h;
  uint32_t lo = 0u, hi = 0u;
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int k = sub + kL * u;
    if (k < a.nch && lb[u] <= R) {
      if (k < 32) lo |= 1u << k;
      else hi |= 1u << (k - 32);
    }
  }
  lo = group_or_u32<kL>(lo);
  hi = group_or_u32<kL>(hi);
  if (sub == 0 && cell < ncell) {
    float fl = (m - h) * 0.9999f - 4e-7f * mag;
    fl = fl > 0.0f ? fl : 0.0f;
    uint32_t seed = jg;
    if (!(m == m) || mg >= 0x7F7FFFFFu) {  // nothing finite was found: trust nothing -- every chunk, no seed
      lo = hi = 0xFFFFFFFFu;
      seed = 0xFFFFFFFFu;
      fl = 0.0f;
    } else if (fl >= a.cap) {  // every point of the cell is beyond max_obstacles_dist: costs nothing
      lo = hi = 0u;
      seed = 0xFFFFFFFFu;
    }
    a.out[cell] = make_uint4(lo, hi, seed, __float_as_uint(fl));
  }
}

template <bool kLds>
__global__ __launch_bounds__(kObsNearBlock) void obs_near_kernel(ObsNearArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  obs_near_body<kLds, kObsNearBlock>(a, static_cast<int>(blockIdx.x), smem);
}

// ---------------------------------------------------------------------------
// The sensor update of a controller cycle in ONE launch with NO inter-workgroup dependency (round 4).
// collision_check.h:91-136 (octree rebuild) + cost_evaluator.h:174-223 (setPointScan) produce, here: the
// voxel bitmap, its two dilations, the obstacle buckets.  Rounds 1-3 built them with one workgroup (slow
// beyond ~4 k points), or with two launches + a byte map + a histogram matrix in global memory, and left
// the dilation to every workgroup of the cycle kernel (3.5 us of its phase A).  Every hand-over between
// workgroups costs a launch boundary or a device-wide barrier (3-7 us on this chip); reading the point list
// again costs next to nothing (105 KB at cfg2, L2-resident: the host has just written it).  So nobody hands
// anything over -- every workgroup reads ALL points and keeps what falls into ITS part of the output:
//   band workgroups    rows [y0, y1) of the bitmap: the points whose voxel row lies within R rows of the
//                      band go into an LDS bitmap (LDS atomics), the band's rows leave as plain stores
//                      together with their two dilations, formed from the LDS rows (R = dilation radius);
//   bucket workgroups  each histograms ALL points over the <= 64 x 64 bucket grid and scans it (redundant:
//                      2 us, instead of a hand-over), then writes ITS slice of the cell starts / skip table and
//                      puts the points of ITS cell range into cell order (rank from an LDS counter: the owner
//                      of a cell is the only one that ranks its points);
//   riders             the near table of a scan polyline (obs_near_body) and the cycle's trig table (TrigJob).
// No global atomics, no scratch buffers, no second launch, nothing for the cycle kernel to dilate.
// Same per-point arithmetic as the kernels above (add_voxel / Rigid3f::apply / cell index).
// ---------------------------------------------------------------------------
struct SensorFusedArgs {
  SensorArgs a;
  ObsNearArgs o;         // the rider (o_blocks > 0)
  int nb, kb, o_blocks;  // band workgroups | bucket workgroups | near-table workgroups (| a.trig.nblk trig workgroups)
  int band_rows;         // rows per band
  float gx0f, gy0f, inv_gf, id_eps;  // sensor_obstacle_fast (id_eps >= 0.5: always the double expression)
  float band_y0, band_dy;            // band b can only hold points with y in band_y0 + b band_dy + [-band_pad, band_dy + band_pad)
  float band_pad;
  unsigned long long *dbg;  // KC_PHASE_STAMPS builds: [workgroup][16] s_memrealtime stamps, or null
  int R;                 // dilation radius in rows; < 0: no masks (spheres without a gap bound, huge robots)
  uint32_t *ginner, *gouter;
  signed char win[kMaxDil + 1], wout[kMaxDil + 1];
  uint8_t *gz;           // spheres: [gH][gwpr * 32] smallest layer code of a voxel column (0: none), else null
};

#ifdef KC_PHASE_STAMPS
#define KC_FSTAMP(slot)                                                                                   \
  do {                                                                                                    \
    if (s.dbg && threadIdx.x == 0) s.dbg[blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime();    \
  } while (0)
#else
#define KC_FSTAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ bool sensor_voxel_cell(const SensorArgs &a, float x, float y, float z, int &cx, int &cy,
                                                  int *code = nullptr) {
  // add_voxel: keys, octree range, z interval of the robot (cylinder / box) or the layer table of a sphere
  const double fx = floor(a.inv_res * static_cast<double>(x));
  const double fy = floor(a.inv_res * static_cast<double>(y));
  const double fz = floor(a.inv_res * static_cast<double>(z));
  if (!(fabs(fx) < 32768.0 && fabs(fy) < 32768.0 && fabs(fz) < 32768.0)) return false;
  const int kz = static_cast<int>(fz);
  if (a.sphere) {
    const int q = kz - a.kz0;
    if (q < 0 || q >= a.nkz) return false;
    const int cd = a.zcode[q];
    if (cd == 0) return false;
    if (code) *code = cd;
  } else {
    const double zlo = static_cast<double>(kz) * a.res;
    const double zhi = static_cast<double>(kz + 1) * a.res;
    if (!(zlo <= a.zc + a.half_height && zhi >= a.zc - a.half_height)) return false;
  }
  cx = static_cast<int>(fx) - a.gkx0;
  cy = static_cast<int>(fy) - a.gky0;
  return cx >= 0 && cy >= 0 && cy < a.gH && (cx >> 5) < a.gwpr;
}

// A thread takes FOUR consecutive points at a time: 48 bytes = three aligned 16-byte loads (the list is [n][3]
// floats; 12-byte loads a point apiece kept the CU's address unit busy for ~2 us per pass of a 8.8 k-point list --
// every workgroup reads every point --, and a second trip doubled it).  Three such groups in flight per thread:
// lists up to 12 k points are ONE memory round trip per pass.  pre() runs once, behind the first trip's loads
// and in front of their first use: what a pass has to set up (LDS zeroing + barrier) hides under the round trip.
// The list is padded to a multiple of four points (host: d_raw), the pad is never handed to f.
constexpr int kSensorGroups = 3;                                      // groups of four points per thread and trip
constexpr int kSensorOneTrip = 4 * kSensorGroups * kSensorBlock;     // points of a list that is one trip
template <class Pre, class F>
__device__ __forceinline__ void sensor_for_points(const SensorArgs &a, Pre &&pre, F &&f) {
  constexpr int kGroups = kSensorGroups;
  const int tid = threadIdx.x;
  const float4 *v = reinterpret_cast<const float4 *>(a.xyz);
  const int ngroups = (a.n + 3) >> 2;
  for (int g0 = 0; g0 < ngroups; g0 += kGroups * kSensorBlock) {
    float4 q[kGroups][3];
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
      const int g = g0 + u * kSensorBlock + tid;
      const int gg = g < ngroups ? g : 0;  // idle slots shadow group 0, used for nothing
      q[u][0] = v[3 * gg];
      q[u][1] = v[3 * gg + 1];
      q[u][2] = v[3 * gg + 2];
    }
    if (g0 == 0) pre();
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
      const int g = g0 + u * kSensorBlock + tid;
      if (g >= ngroups) continue;
      const int i = 4 * g;  // (slot 4 u + p: a compile-time constant in the first trip, for per-thread records)
      f(4 * u, i, q[u][0].x, q[u][0].y, q[u][0].z);
      if (i + 1 < a.n) f(4 * u + 1, i + 1, q[u][0].w, q[u][1].x, q[u][1].y);
      if (i + 2 < a.n) f(4 * u + 2, i + 2, q[u][1].z, q[u][1].w, q[u][2].x);
      if (i + 3 < a.n) f(4 * u + 3, i + 3, q[u][2].y, q[u][2].z, q[u][2].w);
    }
  }
  if (a.n <= 0) pre();
}

// The obstacle of a point and its bucket, as sensor_obstacle -- the cell index from a FLOAT estimate where that is
// safe: the estimate is within id_eps cells of the double expression (host: float rounding of the origin, of the
// difference and of 1 / g), so away from a cell edge by more than that both truncate to the same cell; the few
// points nearer to an edge take the double expression itself.  (f64 conversions run at a quarter of the f32
// rate: they were half of the counting pass of a bucket workgroup, which sees EVERY point.)
__device__ __forceinline__ bool sensor_obstacle_fast(const SensorFusedArgs &s, float x, float y, float z, float &ox,
                                                     float &oy, int &id) {
  const SensorArgs &a = s.a;
  ox = a.t[0] + (a.R[0][0] * x + (a.R[0][1] * y + a.R[0][2] * z));
  oy = a.t[1] + (a.R[1][0] * x + (a.R[1][1] * y + a.R[1][2] * z));
  if (!isfinite(ox) || !isfinite(oy)) return false;
  const float ex = (ox - s.gx0f) * s.inv_gf, ey = (oy - s.gy0f) * s.inv_gf;
  int cx = static_cast<int>(ex), cy = static_cast<int>(ey);
  const float dx = ex - static_cast<float>(cx), dy = ey - static_cast<float>(cy);  // (|.| < 1: truncation)
  const float e = s.id_eps, e1 = 1.0f - s.id_eps;
  if (!(fabsf(dx) > e && fabsf(dx) < e1 && fabsf(dy) > e && fabsf(dy) < e1) || !(fabsf(ex) < 1.0e6f && fabsf(ey) < 1.0e6f)) {
    cx = static_cast<int>((static_cast<double>(ox) - a.gx0) * a.inv_g);
    cy = static_cast<int>((static_cast<double>(oy) - a.gy0) * a.inv_g);
  }
  cx = min(max(cx, 0), a.W - 1);
  cy = min(max(cy, 0), a.H - 1);
  id = cy * a.W + cx;
  return true;
}

__device__ __forceinline__ void sensor_band_body(const SensorFusedArgs &s, int band, unsigned char *smem) {
  const SensorArgs &a = s.a;
  const int tid = threadIdx.x;
  const int R = s.R > 0 ? s.R : 0;
  const int y0 = band * s.band_rows, y1 = min(y0 + s.band_rows, a.gH);
  if (y0 >= y1) return;
  const int lo = y0 - R, nrows = (y1 - y0) + 2 * R;
  uint32_t *lbits = reinterpret_cast<uint32_t *>(smem);  // [nrows][gwpr], row 0 = bitmap row `lo`
  KC_FSTAMP(0);
  // (a band and its halo hold a few per cent of the points: two float compares -- the band's y interval, padded by a
  // voxel for the rounding of the key -- send the others away before any f64 work)
  const float ylo = s.band_y0 + static_cast<float>(band) * s.band_dy - s.band_pad;
  const float yhi = s.band_y0 + static_cast<float>(band + 1) * s.band_dy + s.band_pad;
  // (spheres: the layer codes seen in every column of the band's OWN rows, a word of code bits a column, behind the
  // bitmap rows and the two dilation accumulators)
  const int nown = (y1 - y0) * a.gwpr;
  const int gW = a.gwpr * 32;
  uint32_t *lcode = lbits + nrows * a.gwpr + 2 * nown;
  const bool zcodes = s.gz != nullptr;
  sensor_for_points(a, [&] {
    for (int i = tid; i < nrows * a.gwpr; i += kSensorBlock) lbits[i] = 0u;
    if (zcodes)
      for (int i = tid; i < (y1 - y0) * gW; i += kSensorBlock) lcode[i] = 0u;
    __syncthreads();
    KC_FSTAMP(1);
  }, [&](int, int, float x, float y, float z) {
    if (!(y >= ylo && y <= yhi)) return;
    int cx, cy, code = 0;
    if (sensor_voxel_cell(a, x, y, z, cx, cy, &code)) {
      const int r = cy - lo;
      if (r >= 0 && r < nrows) atomicOr(&lbits[r * a.gwpr + (cx >> 5)], 1u << (cx & 31));
      if (zcodes && r >= R && r < R + (y1 - y0)) atomicOr(&lcode[(r - R) * gW + cx], 1u << (code - 1));
    }
  });
  __syncthreads();
  KC_FSTAMP(2);
  // The band's rows leave as they are; their two dilations are formed from the LDS rows around them (rows outside
  // the bitmap hold no bit: sensor_voxel_cell admits none there).  A task per (row offset j, output word): the
  // (2 R + 1) x words tasks of a band go round ALL lanes (a thread per output word left 7/8 of the workgroup idle
  // behind a loop of (2 R + 1) x two run widths), the contributions meet in two LDS accumulators.
  const int nout = (y1 - y0) * a.gwpr;
  uint32_t *lin = lbits + nrows * a.gwpr, *lout = lin + nout;
  if (s.R >= 0) {
    for (int t = tid; t < 2 * nout; t += kSensorBlock) lin[t] = 0u;
    __syncthreads();
    const int total = nout * (2 * R + 1);
    for (int t = tid; t < total; t += kSensorBlock) {
      const int jj = t / nout, o = t - jj * nout;  // (the row offset is the slow index: a wavefront reads one LDS row)
      const int yr = o / a.gwpr, w = o - yr * a.gwpr;
      const uint32_t *row = lbits + (yr + jj) * a.gwpr;  // row offset j = jj - R of output row yr (LDS row yr + R)
      const uint32_t mid = row[w];
      const uint32_t left = w > 0 ? row[w - 1] : 0u;
      const uint32_t right = w + 1 < a.gwpr ? row[w + 1] : 0u;
      if ((mid | left | right) == 0u) continue;
      const int aj = jj < R ? R - jj : jj - R;
      if (s.win[aj] >= 0) atomicOr(&lin[o], hdilate(left, mid, right, s.win[aj]));
      if (s.wout[aj] >= 0) atomicOr(&lout[o], hdilate(left, mid, right, s.wout[aj]));
    }
    __syncthreads();
  }
  KC_FSTAMP(3);
  for (int t = tid; t < nout; t += kSensorBlock) {
    const int yr = t / a.gwpr, w = t - yr * a.gwpr;
    const size_t g = static_cast<size_t>(y0 + yr) * a.gwpr + w;
    a.gbits[g] = lbits[(yr + R) * a.gwpr + w];
    if (s.R >= 0) {
      s.ginner[g] = lin[t];
      s.gouter[g] = lout[t];
    }
  }
  if (zcodes)  // the smallest code of a column = its smallest z gap (the LUT ascends)
    for (int t = tid; t < (y1 - y0) * gW; t += kSensorBlock) {
      const uint32_t m = lcode[t];
      s.gz[static_cast<size_t>(y0) * gW + t] = m ? static_cast<uint8_t>(__ffs(static_cast<int>(m))) : static_cast<uint8_t>(0);
    }
  KC_FSTAMP(4);
}

__device__ __forceinline__ void sensor_bucket_body(const SensorFusedArgs &s, int me, unsigned char *smem) {
  const SensorArgs &a = s.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncell = a.W * a.H;
  int *lstart = reinterpret_cast<int *>(smem);                 // [ncell + 1] (+ pad): slot k + 1 = count, then start, of cell k
  unsigned long long *lmask = reinterpret_cast<unsigned long long *>(lstart + ((ncell + 1 + 3) & ~3));  // [64]
  // A list of one trip (<= 12 k points): every thread keeps {cell | rank << 12, obstacle x, y} of its <= 12 points
  // in registers from the counting pass -- this workgroup counts ALL points, so the rank it hands out is the rank
  // in the whole list -- and the placing pass is a table read and two stores for the points of its cell range: no
  // second read of the list, no second atomic.  Longer lists: [ncell] next free position of a cell of my range,
  // and the placing pass reads and transforms every point again.
  const bool one_trip = a.n <= kSensorOneTrip;
  int *lpos = reinterpret_cast<int *>(lmask + 64);
  int rec[4 * kSensorGroups];
  float rox[4 * kSensorGroups], roy[4 * kSensorGroups];
#pragma unroll
  for (int k = 0; k < 4 * kSensorGroups; ++k) rec[k] = -1;
  __shared__ int wave_tot[kSensorBlock / 64];
  KC_FSTAMP(0);
  // ---- 1: counts of ALL points ----------------------------------------------------------------------
  sensor_for_points(a, [&] {
    for (int i = tid; i <= ncell; i += kSensorBlock) lstart[i] = 0;
    __syncthreads();
    KC_FSTAMP(1);
  }, [&](int slot, int i, float x, float y, float z) {
    float ox, oy;
    int id;
    if (!sensor_obstacle_fast(s, x, y, a.obs_z_zero ? 0.0f : z, ox, oy, id)) return;
    const int rank = atomicAdd(&lstart[id + 1], 1);
    if (i < kSensorOneTrip) {  // (first trip: `slot` is a constant after unrolling -- the arrays stay in registers)
      rec[slot] = id | (rank << 12);  // id < 4096 cells, rank < 32 k
      rox[slot] = ox;
      roy[slot] = oy;
    }
  });
  __syncthreads();
  KC_FSTAMP(2);
  // ---- 2: row masks of the non-empty cells (from the COUNTS: slot k + 1 = points of cell k; read-only, like
  // the first half of the scan below -- no barrier of their own), then the starts: in-place inclusive scan of
  // the ncell + 1 slots (consecutive slots per thread, <= 8: the host keeps the grid at 64 x 64; wave scan of
  // the thread totals)
  for (int y = wave; y < a.H; y += kSensorBlock / 64) {
    const bool ne = lane < a.W && lstart[y * a.W + lane + 1] > 0;
    const unsigned long long m = __ballot(ne);
    if (lane == 0) lmask[y] = m;
  }
  {
    const int N = ncell + 1;
    const int per = (N + kSensorBlock - 1) / kSensorBlock;
    const int k0 = tid * per;
    int v[8];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = k0 + k;
      if (k < per && idx < N) sum += lstart[idx];
      v[k] = sum;
    }
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if (lane >= off) incl += u;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    const int offset = base + incl - sum;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = k0 + k;
      if (k < per && idx < N) lstart[idx] = offset + v[k];
    }
  }
  __syncthreads();
  KC_FSTAMP(3);
  // ---- 3: my slice of what the cost kernels read: cell starts, skip table (Chebyshev distance to the nearest
  // non-empty cell, from one 64-bit mask per grid row) -- 64 cells per wavefront, the wavefronts of
  // the bucket workgroups interleaved
  const int c0 = static_cast<int>(static_cast<long long>(ncell) * me / s.kb);
  const int c1 = static_cast<int>(static_cast<long long>(ncell) * (me + 1) / s.kb);
  if (!one_trip) {
    for (int k = c0 + tid; k < c1; k += kSensorBlock) lpos[k] = lstart[k];
    __syncthreads();
  }
  KC_FSTAMP(4);
  if (me == 0 && tid < 4) a.skip[ncell + tid] = 255;  // word padding the cost kernels copy
  for (int k = (wave * s.kb + me) * 64 + lane; k <= ncell; k += s.kb * kSensorBlock) {
    a.cell_start[k] = lstart[k];
    if (k == ncell) break;
    const int y = k / a.W, x = k - y * a.W;
    unsigned long long acc = lmask[y];
    int r = 0;
    const int rmax = max(a.W, a.H);
    for (;;) {
      const int x0 = max(x - r, 0), x1 = min(x + r, a.W - 1);
      const unsigned long long win = (x1 - x0 == 63) ? ~0ull : (((1ull << (x1 - x0 + 1)) - 1ull) << x0);
      if (acc & win) break;
      ++r;
      if (r > rmax || r >= 255) {
        r = 255;
        break;
      }
      if (y - r >= 0) acc |= lmask[y - r];
      if (y + r < a.H) acc |= lmask[y + r];
    }
    a.skip[k] = static_cast<uint8_t>(r);
  }
  KC_FSTAMP(5);
  // ---- 4: the points of my cell range into cell order ------------------------------------------------
  if (one_trip) {
#pragma unroll
    for (int k = 0; k < 4 * kSensorGroups; ++k) {
      const int id = rec[k] & 4095;
      if (rec[k] >= 0 && id >= c0 && id < c1) {
        const int pos = lstart[id] + (rec[k] >> 12);
        a.bx[pos] = rox[k];
        a.by[pos] = roy[k];
      }
    }
  } else {
    sensor_for_points(a, [] {}, [&](int, int, float x, float y, float z) {
      float ox, oy;
      int id;
      if (sensor_obstacle_fast(s, x, y, a.obs_z_zero ? 0.0f : z, ox, oy, id) && id >= c0 && id < c1) {
        const int pos = atomicAdd(&lpos[id], 1);
        a.bx[pos] = ox;
        a.by[pos] = oy;
      }
    });
  }
  KC_FSTAMP(6);
}

template <bool kLds>
__global__ __launch_bounds__(kSensorBlock) void sensor_fused_kernel(SensorFusedArgs s_) {
  const SensorFusedArgs &s = *kernargs_touched<SensorFusedArgs>();  // (every kernarg line asked for at once)
  extern __shared__ __align__(16) unsigned char smem[];
  const int b = static_cast<int>(blockIdx.x);
  if (b < s.nb) sensor_band_body(s, b, smem);
  else if (b < s.nb + s.kb) sensor_bucket_body(s, b - s.nb, smem);
  else if (b < s.nb + s.kb + s.o_blocks) obs_near_body<kLds, kSensorBlock>(s.o, b - s.nb - s.kb, smem);
  else trig_job_block<kSensorBlock>(s.a.trig, b - s.nb - s.kb - s.o_blocks);
}

}  // namespace kc
