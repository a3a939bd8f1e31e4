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
  const float h = a.g * 0.70710679f * 1.0001f + a.slack;
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

// A scan update that knows where the next cycle starts builds the table in the launch of the sensor tables:
// workgroup 0 is sensor_build_kernel, the others take kSensorBlock / 8 cells each (nothing of theirs depends
// on workgroup 0: the obstacles in beam order come from the host).  The cycle then finds the table in place.
template <bool kLds>
__global__ __launch_bounds__(kSensorBlock) void sensor_build_scan_kernel(SensorArgs a, ObsNearArgs o) {
  extern __shared__ __align__(16) unsigned char smem[];
  if (blockIdx.x >= gridDim.x - a.trig.nblk) {
    trig_job_block<kSensorBlock>(a.trig, static_cast<int>(blockIdx.x - (gridDim.x - a.trig.nblk)));
    return;
  }
  if (blockIdx.x == 0) sensor_build_body(a, smem);
  else obs_near_body<kLds, kSensorBlock>(o, static_cast<int>(blockIdx.x) - 1, smem);
}

}  // namespace kc
