// laserscan -> occupancy grid on gfx950 with the reference CPU mapper's
// semantics (mapping/local_mapper.{h,cpp}, mapping/line_drawing.h:55-124).
//
// Cell value = max over every write of the sequential algorithm (UNEXPLORED -1
// < EMPTY 0 < OCCUPIED 100; local_mapper.cpp:147-155 only ever raises a cell),
// so the result does not depend on beam order and the rays can be rasterised
// in parallel: pass 1 clears to -1, pass 2 stamps EMPTY on every super-cover
// cell (plain stores, every writer stores the same value), pass 3 stamps
// OCCUPIED on the end cells.  Stream order between the passes gives exactly
// the max.  One wavefront per beam; the Bresenham error term has a closed form
// per step, so the 64 lanes each rasterise a contiguous chunk of the line.
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kc_internal.h"

namespace kc {

struct MapGeom {
  int H, W;      // rows (i), cols (j); cell (i,j) at i + j*H (Eigen MatrixXi)
  int c0, c1;    // m_centralPoint
  int s0, s1;    // m_startPoint
  float res;
  float pos0, pos1;
};

// end cell of a beam: LocalMapper::updateGrid_ (local_mapper.cpp:127-134)
// + localToGrid (local_mapper.h:210-222).  cos/sin come from the host libm
// table (the reference calls ::cos(double) on the float sum orient + angle).
// Both passes recompute it (a handful of operations) instead of a third kernel
// and an array in between.
__device__ __forceinline__ int2 beam_endpoint(const MapGeom &g, float range, double2 cs) {
  const double r = static_cast<double>(range);
  const float x = static_cast<float>(static_cast<double>(g.pos0) + r * cs.x);
  const float y = static_cast<float>(static_cast<double>(g.pos1) + r * cs.y);
  int2 t;
  t.x = g.c0 + static_cast<int>(kc::div_rn(x, g.res));  // trunc toward zero
  t.y = g.c1 + static_cast<int>(kc::div_rn(y, g.res));
  return t;
}

__device__ __forceinline__ void stamp_empty(int *grid, const MapGeom &g, int i,
                                            int j) {
  if (i >= 0 && i < g.H && j >= 0 && j < g.W)
    grid[(size_t)i + (size_t)j * (size_t)g.H] = KC_EMPTY;
}

// pass 2: bresenhamEnhanced (line_drawing.h:55-124), all cells as EMPTY.
// Major-axis step i (1..n): e_i = d + i*dd_minor, k_i = floor((e_i - 1)/dd_major)
// minor increments so far, error_i = e_i - k_i*dd_major in [1, dd_major].
constexpr int kBeamsPerBlock = 4;

__global__ __launch_bounds__(64 * kBeamsPerBlock) void rays_kernel(
    MapGeom g, const float *__restrict__ ranges, const double2 *__restrict__ trig, int n,
    int *__restrict__ grid, int step_limit) {
  // Workgroups go to the eight XCDs round-robin, each with an L2 of its own: XCD x takes the x-th EIGHTH of the
  // beams (a sector of an ordered scan), so the dirty partial lines of a grid region collect in one L2 instead of
  // all eight
  const int per_xcd = static_cast<int>(gridDim.x >> 3);  // (the host launches a multiple of eight workgroups)
  const int block = static_cast<int>(blockIdx.x & 7u) * per_xcd + static_cast<int>(blockIdx.x >> 3);
  const int beam = block * kBeamsPerBlock + (threadIdx.x >> 6);
  if (beam >= n) return;
  const int lane = threadIdx.x & 63;
  const int2 t = beam_endpoint(g, ranges[beam], trig[beam]);
  int dx = t.x - g.s0, dy = t.y - g.s1;
  const int xstep = dx >= 0 ? 1 : -1, ystep = dy >= 0 ? 1 : -1;
  dx = abs(dx);
  dy = abs(dy);
  if (lane == 0) stamp_empty(grid, g, g.s0, g.s1);  // first emitted point
  const bool xmajor = 2 * dx >= 2 * dy;
  // tiled scan: only the first step_limit steps (the tiles own the rest)
  const int nsteps = min(xmajor ? dx : dy, step_limit);
  if (nsteps == 0) return;
  const long long dmaj = xmajor ? dx : dy, dmin = xmajor ? dy : dx;
  const long long ddmaj = 2 * dmaj, ddmin = 2 * dmin;
  const int astep = xmajor ? xstep : ystep, bstep = xmajor ? ystep : xstep;
  const int a0 = xmajor ? g.s0 : g.s1, b0 = xmajor ? g.s1 : g.s0;
  // The 64 lanes take 64 CONSECUTIVE steps per trip (closed-form state before step i, as above): the stores of
  // an x-major line then fall into a few 64-byte lines of the column-major grid instead of 64 -- the L2's write
  // transactions, not the arithmetic, bound this pass (a contiguous chunk of steps per lane: 18.6 us at
  // 4096 beams x ~500 steps)
  for (int i = lane + 1; i <= nsteps; i += 64) {
    const long long eprev = dmaj + (long long)(i - 1) * ddmin;
    const long long k = static_cast<long long>(
        floor(static_cast<double>(eprev - 1) / static_cast<double>(ddmaj)));
    const long long errorprev = eprev - k * ddmaj;
    long long error = errorprev + ddmin;
    const int a = a0 + astep * i;
    int bq = b0 + bstep * (int)k;
    if (error > ddmaj) {
      bq += bstep;
      error -= ddmaj;
      const bool lo = error + errorprev < ddmaj, hi = error + errorprev > ddmaj;
      // x-major: lo -> (x, y - ystep), hi -> (x - xstep, y); y-major mirrored
      if (!hi) {  // lo or both
        if (xmajor) stamp_empty(grid, g, a, bq - bstep);
        else stamp_empty(grid, g, bq - bstep, a);
      }
      if (!lo) {  // hi or both
        if (xmajor) stamp_empty(grid, g, a - astep, bq);
        else stamp_empty(grid, g, bq, a - astep);
      }
    }
    if (xmajor) stamp_empty(grid, g, a, bq);
    else stamp_empty(grid, g, bq, a);
  }
}

// ---- M3: which beam decides a cell -------------------------------------------
// The sequential reference overwrites gridDataProb(pt) beam after beam
// (local_mapper.cpp:199), so the probability of a cell is the one computed with
// the range of the LAST beam whose super-cover line crosses it: tag = beam + 1,
// kept as a maximum.  Global atomics run at the memory side, one request per
// 64-byte line a wave instruction touches, and that request rate is what bounds
// this pass: the tags live in 4x4-cell blocks (one line each) and the 64 lanes
// of a wave take 64 CONSECUTIVE steps of the line, so a wave instruction
// touches ~20 lines instead of 64.
__device__ __forceinline__ size_t tag_index(int i, int j, int hb) {
  return ((static_cast<size_t>(j >> 2) * hb + (i >> 2)) << 4) | ((j & 3) << 2) | (i & 3);
}

__device__ __forceinline__ void stamp_tag(int *grid, unsigned int *last, unsigned int tag, int hb,
                                          const MapGeom &g, int i, int j) {
  if (i >= 0 && i < g.H && j >= 0 && j < g.W) {
    grid[(size_t)i + (size_t)j * (size_t)g.H] = KC_EMPTY;
    atomicMax(&last[tag_index(i, j, hb)], tag);
  }
}

// same line as rays_kernel, one step per lane per pass: the state before step i
// has the closed form used there for the first step of a chunk
__global__ __launch_bounds__(64 * kBeamsPerBlock) void rays_bayes_kernel(
    MapGeom g, const float *__restrict__ ranges, const double2 *__restrict__ trig, int n,
    int *__restrict__ grid, unsigned int *__restrict__ last, int hb, int step_first,
    int step_limit) {
  const int beam = blockIdx.x * kBeamsPerBlock + (threadIdx.x >> 6);
  if (beam >= n) return;
  const unsigned int tag = static_cast<unsigned int>(beam) + 1u;
  const int lane = threadIdx.x & 63;
  const int2 t = beam_endpoint(g, ranges[beam], trig[beam]);
  int dx = t.x - g.s0, dy = t.y - g.s1;
  const int xstep = dx >= 0 ? 1 : -1, ystep = dy >= 0 ? 1 : -1;
  dx = abs(dx);
  dy = abs(dy);
  if (lane == 0 && step_first <= 1) stamp_tag(grid, last, tag, hb, g, g.s0, g.s1);  // first emitted point
  const bool xmajor = 2 * dx >= 2 * dy;
  const int nsteps = xmajor ? dx : dy;
  const long long dmaj = xmajor ? dx : dy, dmin = xmajor ? dy : dx;
  const long long ddmaj = 2 * dmaj, ddmin = 2 * dmin;
  const int astep = xmajor ? xstep : ystep, bstep = xmajor ? ystep : xstep;
  const int a0 = xmajor ? g.s0 : g.s1, b0 = xmajor ? g.s1 : g.s0;
  const int nwalk = min(nsteps, step_limit);
  for (int i = lane + step_first; i <= nwalk; i += 64) {
    const long long eprev = dmaj + (long long)(i - 1) * ddmin;
    const long long k = static_cast<long long>(
        floor(static_cast<double>(eprev - 1) / static_cast<double>(ddmaj)));
    const long long errorprev = eprev - k * ddmaj;
    long long error = errorprev + ddmin;
    const int a = a0 + astep * i;
    int bq = b0 + bstep * (int)k;
    if (error > ddmaj) {
      bq += bstep;
      error -= ddmaj;
      const bool lo = error + errorprev < ddmaj, hi = error + errorprev > ddmaj;
      // x-major: lo -> (x, y - ystep), hi -> (x - xstep, y); y-major mirrored
      if (!hi) {  // lo or both
        if (xmajor) stamp_tag(grid, last, tag, hb, g, a, bq - bstep);
        else stamp_tag(grid, last, tag, hb, g, bq - bstep, a);
      }
      if (!lo) {  // hi or both
        if (xmajor) stamp_tag(grid, last, tag, hb, g, a - astep, bq);
        else stamp_tag(grid, last, tag, hb, g, bq, a - astep);
      }
    }
    if (xmajor) stamp_tag(grid, last, tag, hb, g, a, bq);
    else stamp_tag(grid, last, tag, hb, g, bq, a);
  }
}

// ---- tiled scan -----------------------------------------------------------------
// Cell ownership instead of beam ownership: a workgroup owns a 64 x 16 tile of
// the grid in LDS, finds the beams whose line crosses it (separating-axis band
// test, then the exact step range along the major axis and the minor range of
// those steps) and walks only the steps inside.  Every cell is written once,
// coalesced, UNEXPLORED included -- no clear pass, no global atomics, and steps
// outside the grid are never walked.  Around the sensor every beam crosses the
// same few tiles, so the first kNearSteps steps of each beam stay with the
// beam-parallel kernels (rays_kernel / rays_bayes_kernel with a step limit),
// which run behind this one and only ever raise a cell.
constexpr int kTileI = 64, kTileJ = 16, kNearSteps = 64;
constexpr int kTileThreads = 256;
constexpr int kChunkSteps = 16;  // steps one lane walks: the closed-form start costs a division
constexpr int kMaxChunks = (kTileI + 2 + kChunkSteps - 1) / kChunkSteps + 1;
constexpr int kRoundBeams = 2 * kTileThreads;  // beams tested between two barriers
constexpr int kNearSlices = 64;                // beam slices that share a near-field tile

// direction sector (of 64) of the cell offset (dx, dy) from the start cell: a
// cheap first cut of the beam x tile tests -- a line can only reach a tile whose
// angular span, seen from the start cell, holds the line's direction
__device__ __forceinline__ int sector_of_angle(float a) {
  const int s = static_cast<int>((a + 3.14159265358979f) * (64.0f / 6.28318530717959f));
  return min(max(s, 0), 63);
}

// ends[b] = end cell of beam b; behind the n end cells, one byte per beam: its sector
__global__ void beam_ends_kernel(MapGeom g, const float *__restrict__ ranges,
                                 const double2 *__restrict__ trig, int n, int2 *__restrict__ ends) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < n) {
    const int2 t = beam_endpoint(g, ranges[b], trig[b]);
    ends[b] = t;
    reinterpret_cast<uint8_t *>(ends + n)[b] = static_cast<uint8_t>(
        sector_of_angle(atan2f(static_cast<float>(t.y - g.s1), static_cast<float>(t.x - g.s0))));
  }
}

struct TileTask {
  int beam, ifirst, ilast;
};

struct TileLds {
  unsigned int cell[kTileI * kTileJ];
  TileTask task[kRoundBeams * kMaxChunks];
  int ntask;
};

// line of one beam in major/minor form
struct BeamLine {
  int a0, b0, astep, bstep, nsteps;
  long long dmaj, ddmaj, ddmin;
  bool xmajor;
};

__device__ __forceinline__ BeamLine beam_line(const MapGeom &g, int2 t) {
  BeamLine l;
  const int dx = t.x - g.s0, dy = t.y - g.s1;
  const int adx = abs(dx), ady = abs(dy);
  l.xmajor = adx >= ady;  // 2 dx >= 2 dy
  l.nsteps = l.xmajor ? adx : ady;
  l.astep = l.xmajor ? (dx >= 0 ? 1 : -1) : (dy >= 0 ? 1 : -1);
  l.bstep = l.xmajor ? (dy >= 0 ? 1 : -1) : (dx >= 0 ? 1 : -1);
  l.a0 = l.xmajor ? g.s0 : g.s1;
  l.b0 = l.xmajor ? g.s1 : g.s0;
  l.dmaj = l.nsteps;
  l.ddmaj = 2 * l.dmaj;
  l.ddmin = 2 * static_cast<long long>(l.xmajor ? ady : adx);
  return l;
}

// minor increments made by the steps 1..i (k_i of rays_kernel)
__device__ __forceinline__ long long minor_after(const BeamLine &l, int i) {
  return static_cast<long long>(floor(
      static_cast<double>(l.dmaj + static_cast<long long>(i) * l.ddmin - 1) / static_cast<double>(l.ddmaj)));
}

// Stamps of the beams [b_begin, b_end), steps [step_min, step_max], that fall
// into the tile [I0..I1] x [J0..J1], into L.cell (tag = beam + 1, maximum; any
// non-zero value without kBayes).  Rounds of kTileThreads beams: every thread
// tests one beam and queues the crossing part of its line in chunks of
// kChunkSteps steps, then the threads share the chunks.
template <bool kBayes>
__device__ void tile_accumulate(const MapGeom &g, const int2 *__restrict__ ends, int n_all,
                                int b_begin, int b_end, int step_min, int step_max, int I0, int I1,
                                int J0, int J1, TileLds &L) {
  // tile centre relative to the start cell, half extents + 3 cells: a stamped
  // cell lies within two cells of the ideal line
  const double ci = 0.5 * (I0 + I1) - g.s0, cj = 0.5 * (J0 + J1) - g.s1;
  const double hi = 0.5 * (I1 - I0) + 3.0, hj = 0.5 * (J1 - J0) + 3.0;
  // sectors the tile (grown by the same three cells) spans as seen from the
  // start cell, one more on either side; all of them when the start lies inside
  unsigned long long smask = ~0ull;
  if (fabs(ci) > hi || fabs(cj) > hj) {
    const float ax[4] = {static_cast<float>(ci - hi), static_cast<float>(ci + hi), static_cast<float>(ci - hi),
                         static_cast<float>(ci + hi)};
    const float ay[4] = {static_cast<float>(cj - hj), static_cast<float>(cj - hj), static_cast<float>(cj + hj),
                         static_cast<float>(cj + hj)};
    const float a0 = atan2f(ay[0], ax[0]);
    float lo = 0.0f, up = 0.0f;
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      float dd = atan2f(ay[q], ax[q]) - a0;
      if (dd > 3.14159265358979f) dd -= 6.28318530717959f;
      if (dd <= -3.14159265358979f) dd += 6.28318530717959f;
      lo = fminf(lo, dd);
      up = fmaxf(up, dd);
    }
    auto wrap = [](float a) {
      if (a > 3.14159265358979f) a -= 6.28318530717959f;
      if (a < -3.14159265358979f) a += 6.28318530717959f;
      return a;
    };
    const int s_lo = sector_of_angle(wrap(a0 + lo)), s_up = sector_of_angle(wrap(a0 + up));
    const int cnt = ((s_up - s_lo) & 63) + 3;  // the span plus one sector on either side
    smask = 0ull;
    if (cnt >= 64) smask = ~0ull;
    else
      for (int q = 0; q < cnt; ++q) smask |= 1ull << ((s_lo - 1 + q) & 63);
  }
  const uint8_t *const sect = reinterpret_cast<const uint8_t *>(ends + n_all);
  // the workgroups of a launch start at different rounds: all of them reading the
  // same two kilobytes of end cells at the same time is an L2 hot spot
  const int nrounds = (b_end - b_begin + kRoundBeams - 1) / kRoundBeams;
  const int rot = static_cast<int>(blockIdx.x + blockIdx.y * gridDim.x);
  for (int r = 0; r < nrounds; ++r) {
    const int base = b_begin + ((r + rot) % nrounds) * kRoundBeams;
    if (threadIdx.x == 0) L.ntask = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kRoundBeams / kTileThreads; ++u) {
      const int b = base + u * kTileThreads + static_cast<int>(threadIdx.x);
      if (b >= b_end) continue;
      if (!((smask >> sect[b]) & 1ull)) continue;
      const int2 t = ends[b];
      const int dx = t.x - g.s0, dy = t.y - g.s1;
      const double cross = static_cast<double>(dx) * cj - static_cast<double>(dy) * ci;
      if (fabs(cross) <= fabs(static_cast<double>(dy)) * hi + fabs(static_cast<double>(dx)) * hj) {
        const BeamLine l = beam_line(g, t);
        const int A0 = l.xmajor ? I0 : J0, A1 = l.xmajor ? I1 : J1;
        const int B0 = l.xmajor ? J0 : I0, B1 = l.xmajor ? J1 : I1;
        // a stamp of step i sits at major coordinate a_i or a_i - astep
        int ilo = l.astep > 0 ? A0 - l.a0 : l.a0 - A1;
        int ihi = l.astep > 0 ? A1 + 1 - l.a0 : l.a0 - A0 + 1;
        ilo = max(ilo, max(step_min, 1));
        ihi = min(ihi, min(step_max, l.nsteps));
        if (ilo <= ihi) {
          // the minor coordinate only moves one way: the stamps of steps
          // ilo..ihi lie between the minor cell before step ilo and the one
          // after step ihi
          const int m0 = l.b0 + l.bstep * static_cast<int>(minor_after(l, ilo - 1));
          const int m1 = l.b0 + l.bstep * static_cast<int>(minor_after(l, ihi));
          if (max(m0, m1) >= B0 && min(m0, m1) <= B1) {
            const int nch = (ihi - ilo) / kChunkSteps + 1;
            const int slot = atomicAdd(&L.ntask, nch);
            for (int c = 0; c < nch; ++c)
              L.task[slot + c] = TileTask{b, ilo + c * kChunkSteps, min(ilo + (c + 1) * kChunkSteps - 1, ihi)};
          }
        }
      }
    }
    __syncthreads();
    const int nt = L.ntask;
    for (int q = threadIdx.x; q < nt; q += kTileThreads) {
      const TileTask task = L.task[q];
      const BeamLine l = beam_line(g, ends[task.beam]);
      const unsigned int tag = kBayes ? static_cast<unsigned int>(task.beam) + 1u : 1u;
      // tile-local (major, minor) coordinates and LDS strides
      const int A0 = l.xmajor ? I0 : J0, B0 = l.xmajor ? J0 : I0;
      const unsigned int EA = static_cast<unsigned int>((l.xmajor ? I1 : J1) - A0 + 1);
      const unsigned int EB = static_cast<unsigned int>((l.xmajor ? J1 : I1) - B0 + 1);
      const int SA = l.xmajor ? 1 : kTileI, SB = l.xmajor ? kTileI : 1;
      auto stamp = [&](int am, int bm) {
        if (static_cast<unsigned int>(am) < EA && static_cast<unsigned int>(bm) < EB) {
          const int idx = am * SA + bm * SB;
          if (kBayes) atomicMax(&L.cell[idx], tag);
          else L.cell[idx] = 1u;
        }
      };
      const long long k = minor_after(l, task.ifirst - 1);
      const long long e0 = l.dmaj + static_cast<long long>(task.ifirst - 1) * l.ddmin - k * l.ddmaj;
      int a = l.a0 + l.astep * (task.ifirst - 1) - A0;
      int bq = l.b0 + l.bstep * static_cast<int>(k) - B0;
      if (l.nsteps < (1 << 28)) {
        // the error term stays in [1, ddmaj] and the sums below 2 ddmaj: 32 bits
        const int ddmaj = static_cast<int>(l.ddmaj), ddmin = static_cast<int>(l.ddmin);
        int error = static_cast<int>(e0);
        for (int i = task.ifirst; i <= task.ilast; ++i) {
          const int errorprev = error;
          a += l.astep;
          error += ddmin;
          if (error > ddmaj) {
            bq += l.bstep;
            error -= ddmaj;
            const int sum = error + errorprev;
            if (sum <= ddmaj) stamp(a, bq - l.bstep);  // below the line, or both
            if (sum >= ddmaj) stamp(a - l.astep, bq);  // above the line, or both
          }
          stamp(a, bq);
        }
      } else {
        long long error = e0;
        for (int i = task.ifirst; i <= task.ilast; ++i) {
          const long long errorprev = error;
          a += l.astep;
          error += l.ddmin;
          if (error > l.ddmaj) {
            bq += l.bstep;
            error -= l.ddmaj;
            const long long sum = error + errorprev;
            if (sum <= l.ddmaj) stamp(a, bq - l.bstep);
            if (sum >= l.ddmaj) stamp(a - l.astep, bq);
          }
          stamp(a, bq);
        }
      }
    }
    __syncthreads();
  }
}

// far field: one workgroup per tile, all beams, steps beyond kNearSteps; writes
// every cell of the tile (UNEXPLORED / EMPTY, and the tag grid with kBayes)
template <bool kBayes>
__global__ __launch_bounds__(kTileThreads) void scan_tiles_kernel(
    MapGeom g, const int2 *__restrict__ ends, int n, int *__restrict__ grid,
    unsigned int *__restrict__ last, int hb) {
  __shared__ TileLds L;
  const int I0 = blockIdx.x * kTileI, J0 = blockIdx.y * kTileJ;
  const int I1 = min(I0 + kTileI, g.H) - 1, J1 = min(J0 + kTileJ, g.W) - 1;
  for (int t = threadIdx.x; t < kTileI * kTileJ; t += kTileThreads) L.cell[t] = 0u;
  tile_accumulate<kBayes>(g, ends, n, 0, n, kNearSteps + 1, INT_MAX, I0, I1, J0, J1, L);
  for (int t = threadIdx.x; t < kTileI * kTileJ; t += kTileThreads) {
    const int i = I0 + (t & (kTileI - 1)), j = J0 + t / kTileI;
    if (i <= I1 && j <= J1) {
      const unsigned int v = L.cell[t];
      grid[static_cast<size_t>(i) + static_cast<size_t>(j) * static_cast<size_t>(g.H)] =
          v ? KC_EMPTY : KC_UNEXPLORED;
      if (kBayes) last[tag_index(i, j, hb)] = v;
    }
  }
}

// near field with tags (Bayesian scan): the tiles around the sensor, each
// shared by `gridDim.z` workgroups that take one slice of the beams each --
// contiguous slices, so that a slice of an ordered scan is a sector and touches
// few cells -- and raise what the far-field kernel wrote: EMPTY by plain
// stores, tags by one atomic per touched cell and slice.
__global__ __launch_bounds__(kTileThreads) void near_tiles_kernel(
    MapGeom g, const int2 *__restrict__ ends, int n, int *__restrict__ grid,
    unsigned int *__restrict__ last, int hb, int ti0, int tj0) {
  __shared__ TileLds L;
  const int I0 = (ti0 + static_cast<int>(blockIdx.x)) * kTileI;
  const int J0 = (tj0 + static_cast<int>(blockIdx.y)) * kTileJ;
  const int I1 = min(I0 + kTileI, g.H) - 1, J1 = min(J0 + kTileJ, g.W) - 1;
  const int per = (n + static_cast<int>(gridDim.z) - 1) / static_cast<int>(gridDim.z);
  const int b_begin = static_cast<int>(blockIdx.z) * per, b_end = min(b_begin + per, n);
  if (b_begin >= b_end) return;
  for (int t = threadIdx.x; t < kTileI * kTileJ; t += kTileThreads) L.cell[t] = 0u;
  tile_accumulate<true>(g, ends, n, b_begin, b_end, 1, kNearSteps, I0, I1, J0, J1, L);
  // the first emitted point of every beam is the start cell
  if (threadIdx.x == 0 && g.s0 >= I0 && g.s0 <= I1 && g.s1 >= J0 && g.s1 <= J1)
    atomicMax(&L.cell[(g.s0 - I0) + (g.s1 - J0) * kTileI], static_cast<unsigned int>(b_end));
  __syncthreads();
  for (int t = threadIdx.x; t < kTileI * kTileJ; t += kTileThreads) {
    const unsigned int v = L.cell[t];
    if (v != 0u) {
      const int i = I0 + (t & (kTileI - 1)), j = J0 + t / kTileI;
      grid[static_cast<size_t>(i) + static_cast<size_t>(j) * static_cast<size_t>(g.H)] = KC_EMPTY;
      atomicMax(&last[tag_index(i, j, hb)], v);
    }
  }
}

// pass 3: the end cell of every beam (fillGridAroundPoint with padding 0,
// local_mapper.cpp:148-151)
// The last workgroup to finish tells the host (sequence number into pinned
// memory, polled by kc_mapper_sync instead of a stream wait); every workgroup
// releases its stores (device scope) before it takes its ticket.
// Plain scans alternate between two grids: the workgroups behind the beam blocks fill the OTHER grid
// with UNEXPLORED for the next scan (16-byte stores; the runtime's memset kernel in front of every scan
// was 8 us of a 36 us scan).
__global__ void endpoints_kernel(MapGeom g, const float *__restrict__ ranges,
                                 const double2 *__restrict__ trig, int n, int *__restrict__ grid,
                                 unsigned int *ticket, long long *host_seq, long long seq,
                                 int4 *__restrict__ clear, size_t clear_vec, int beam_blocks) {
  if (static_cast<int>(blockIdx.x) < beam_blocks) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n) {
      const int2 t = beam_endpoint(g, ranges[b], trig[b]);
      if (t.x >= 0 && t.x < g.H && t.y >= 0 && t.y < g.W)
        grid[(size_t)t.x + (size_t)t.y * (size_t)g.H] = KC_OCCUPIED;
    }
  } else {
    const size_t stride = static_cast<size_t>(gridDim.x - beam_blocks) * blockDim.x;
    const int4 v = make_int4(KC_UNEXPLORED, KC_UNEXPLORED, KC_UNEXPLORED, KC_UNEXPLORED);
    for (size_t i = static_cast<size_t>(blockIdx.x - beam_blocks) * blockDim.x + threadIdx.x; i < clear_vec; i += stride)
      clear[i] = v;
    // (no fence, no ticket: the other grid is next read by the kernels of the next scan, in stream
    // order behind this kernel -- a device-scope release per clearing workgroup writes the L2 back a
    // thousand times: 75 us)
    return;
  }
  if (host_seq == nullptr) return;  // Bayesian scan: the cell pass reports
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(ticket, 1u) == static_cast<unsigned int>(beam_blocks) - 1u) {
      *ticket = 0u;
      *reinterpret_cast<volatile long long *>(host_seq) = seq;
    }
  }
}

// ---- M3: Bayesian cell update + previous-grid warp ---------------------------
struct BayesParams {
  float p_prior, p_occupied, p_empty, range_sure, range_max, wall_size;
};

// LocalMapper::updateGridCellProbability (local_mapper.cpp:106-125), expression
// by expression: the literal 1.0 makes the sensor odds and the product around
// them double; the result narrows to float.
__device__ __forceinline__ float bayes_cell(const BayesParams &bp, float res, float distance,
                                            float current_range, float previous_prob) {
  distance = distance * res;
  current_range = current_range - bp.wall_size;
  const float pF = (distance < current_range) ? bp.p_empty : bp.p_occupied;
  const float delta = (distance < bp.range_sure) ? 0.0f : 1.0f;
  const float p_sensor =
      pF + (delta * kc::div_rn(distance - bp.range_sure, bp.range_max) * (bp.p_prior - pF));
  const float prev_odds = kc::div_rn(previous_prob, 1 - previous_prob);
  const double sensor_odds =
      static_cast<double>(p_sensor) / (1.0 - static_cast<double>(p_sensor));
  const float prior_odds = kc::div_rn(1 - bp.p_prior, bp.p_prior);
  const double p_curr =
      1 - (1 / (1 + ((static_cast<double>(prev_odds) * sensor_odds) *
                     static_cast<double>(prior_odds))));
  return static_cast<float>(p_curr);
}

// one thread per cell: the beam that decided the cell (tag) gives the range,
// the cell's own position gives the distance -- (pt - m_startPoint).norm() on
// Vector2i is Eigen's integer norm: the double sqrt truncated back to int
// (local_mapper.cpp:180).  Cells no beam crossed keep the prior
// (gridDataProb.fill, :227).  The tag grid is cleared for the next scan on the
// way.  A workgroup covers 64 x 4 cells: each wave reads four whole tag lines
// and four 64-byte runs of the column-major probability grids.
__global__ __launch_bounds__(256) void bayes_cells_kernel(
    MapGeom g, BayesParams bp, const float *__restrict__ ranges, unsigned int *__restrict__ last,
    int hb, const float *__restrict__ prev, float *__restrict__ prob) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + wave * 16 + (lane & 15);
  const int j = blockIdx.y * 4 + (lane >> 4);
  if (i >= g.H || j >= g.W) return;
  const size_t k = static_cast<size_t>(i) + static_cast<size_t>(j) * static_cast<size_t>(g.H);
  const size_t tk = tag_index(i, j, hb);
  const unsigned int tag = last[tk];
  float v = bp.p_prior;
  if (tag != 0u) {
    const int di = i - g.s0, dj = j - g.s1;
    const float distance =
        static_cast<float>(static_cast<int>(kc::dsqrt_rn(static_cast<double>(di * di + dj * dj))));
    v = bayes_cell(bp, g.res, distance, ranges[tag - 1u], prev[k]);
    last[tk] = 0u;
  }
  prob[k] = v;
}

// one thread after the last pass of a Bayesian scan: the kernel boundary in
// front of it has made the scan visible, it reports the sequence number
__global__ void scan_done_kernel(long long *host_seq, long long seq) {
  *reinterpret_cast<volatile long long *>(host_seq) = seq;
}

// LocalMapper::getPreviousGridInCurrentPose (local_mapper.cpp:17-78): the
// reference inverts the same Matrix3f for every cell; the host does it once
// (Eigen's closed 3x3 form) and hands over the two rows that matter.  The lazy
// 3-term products reduce as a0 + (a1 + a2).  Cell (row y, col x) at y + x*H.
struct WarpArgs {
  float r0[3], r1[3];
  int H, W;
  float prior;
};

__global__ __launch_bounds__(256) void warp_kernel(WarpArgs a, const float *__restrict__ prev,
                                                   float *__restrict__ out, unsigned int cells) {
  const unsigned int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= cells) return;
  const int y = static_cast<int>(k % static_cast<unsigned int>(a.H));
  const int x = static_cast<int>(k / static_cast<unsigned int>(a.H));
  const float fx = static_cast<float>(x), fy = static_cast<float>(y);
  const double srcX = static_cast<double>(a.r0[0] * fx + (a.r0[1] * fy + a.r0[2] * 1.0f));
  const double srcY = static_cast<double>(a.r1[0] * fx + (a.r1[1] * fy + a.r1[2] * 1.0f));
  float value = a.prior;
  if (srcX >= 0 && srcX < a.W - 1 && srcY >= 0 && srcY < a.H - 1) {
    const int x0 = static_cast<int>(floor(srcX)), y0 = static_cast<int>(floor(srcY));
    const int x1 = x0 + 1, y1 = y0 + 1;
    const float w0 = static_cast<float>(srcX - x0), w1 = 1.0f - w0;
    const float h0 = static_cast<float>(srcY - y0), h1 = 1.0f - h0;
    const size_t H = static_cast<size_t>(a.H);
    const float p00 = prev[y0 + x0 * H], p01 = prev[y0 + x1 * H];
    const float p10 = prev[y1 + x0 * H], p11 = prev[y1 + x1 * H];
    value = h1 * (w1 * p00 + w0 * p01) + h0 * (w1 * p10 + w0 * p11);
  }
  out[k] = value;
}

}  // namespace kc

using namespace kc;

struct kc_mapper {
  MapGeom g{};
  float orient = 0.f;
  int device = 0;
  size_t cap = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  Timing timing;
  DevBuf<int> d_grid;      // the grid of the last scan (plain scans alternate between the two)
  DevBuf<int> d_grid_alt;  // ... the other one
  bool alt_cleared = false;  // d_grid_alt holds UNEXPLORED everywhere (left so by the last plain scan)
  DevBuf<float> d_ranges;
  DevBuf<double2> d_trig;
  DevBuf<unsigned int> d_ticket;
  DevBuf<int2> d_ends;       // tiled scan: end cell of every beam
  // KC_MAPPER_TILES: 1 = Bayesian scans tiled (default: the tag maximum needs no
  // global atomic beyond the tiles around the sensor), 3 = Bayesian scans with
  // only the near field tiled and the far field beam-parallel (faster for short
  // ranges, slower for long ones), 2 = plain scans tiled too (slower than the
  // three beam-parallel passes: test hook), 0 = never
  int tile_mode = 1;
  bool tiles = false;        // ... for the scan being queued
  PinBuf<long long> h_seq;   // written by the last endpoints workgroup
  long long seq = 0;         // scans launched
  bool direct = false;       // host stores reach device memory (large BAR)
  PinBuf<float> h_ranges;
  PinBuf<double2> h_trig;
  PinBuf<int> h_grid;
  // M3 (kc_mapper_enable_bayes): probability grids + the deciding-beam tags
  bool bayes = false;
  BayesParams bp{};
  DevBuf<float> d_prob, d_prev, d_prev_tmp;
  DevBuf<unsigned int> d_last;
  PinBuf<float> h_prob;
  // the angle table of a lidar does not change between scans: the trig table
  // is rebuilt only when the angles differ from the previous call
  std::vector<double> last_angles;
  bool trig_valid = false;
};

namespace {

int float_bits(float f) {
  int v;
  std::memcpy(&v, &f, sizeof(v));
  return v;
}

// true when the last scan launched is known to have finished (its sequence
// number arrived), after polling for at most `us` microseconds
bool scan_done(kc_mapper *m, int us) {
  if (m->seq == 0) return true;
  volatile long long *p = m->h_seq.p;
  if (*p == m->seq) return true;
  const auto t0 = std::chrono::steady_clock::now();
  for (long spins = 0;; ++spins) {
    if (*p == m->seq) return true;
    if ((spins & 255) == 255 &&
        std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(us))
      return false;
  }
}

int run_scan(kc_mapper *m, const double *angles, const double *ranges,
             size_t n, bool bayes = false) {
  KC_HIP(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  // the staging / device range buffers are free once the previous scan is done
  if (m->timing.enabled || !scan_done(m, 0)) KC_HIP(hipStreamSynchronize(s));
  m->timing.begin_cycle();
  const size_t cells = static_cast<size_t>(m->g.H) * m->g.W;
  m->tiles = m->tile_mode == 2 || (m->tile_mode == 1 && bayes);
  const bool plain = !bayes && !m->tiles && n != 0;
  if (plain && m->alt_cleared && m->d_grid_alt.p) {
    // the previous plain scan left the other grid UNEXPLORED: this scan takes it (the old one stays
    // readable until this scan's endpoint kernel clears it for the next)
    std::swap(m->d_grid, m->d_grid_alt);
    m->alt_cleared = false;
  } else if (!m->tiles || n == 0) {  // the tiled scan writes every cell itself
    KC_TRY(m->timing.start("grid_clear", s));
    KC_HIP(hipMemsetAsync(m->d_grid.p, 0xFF, cells * sizeof(int), s));  // -1
    KC_TRY(m->timing.stop(s));
  }
  if (n == 0) {
    m->seq = 0;  // nothing will signal: kc_mapper_sync waits on the stream
    if (bayes)   // gridDataProb.fill(m_pPrior), local_mapper.cpp:227
      KC_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(m->d_prob.p),
                               float_bits(m->bp.p_prior), cells, s));
    return KC_OK;
  }
  KC_TRY(m->d_ranges.reserve(n));
  KC_TRY(m->d_trig.reserve(n));
  KC_TRY(m->h_ranges.reserve(n));
  KC_TRY(m->h_trig.reserve(n));
  const bool same = m->trig_valid && m->last_angles.size() == n &&
                    std::memcmp(m->last_angles.data(), angles,
                                n * sizeof(double)) == 0;
  if (!same) {
    for (size_t i = 0; i < n; ++i) {
      // updateGrid_(const float angle, ...): cos(m_laserscanOrientation + angle)
      const float a = static_cast<float>(angles[i]);
      const double th = static_cast<double>(m->orient + a);
      m->h_trig.p[i] = make_double2(std::cos(th), std::sin(th));
    }
    m->last_angles.assign(angles, angles + n);
    m->trig_valid = true;
    KC_HIP(hipMemcpyAsync(m->d_trig.p, m->h_trig.p, n * sizeof(double2),
                          hipMemcpyHostToDevice, s));
  }
  if (m->direct) {
    // ranges straight into device memory (write-combined stores over the BAR)
    float *dst = m->d_ranges.p;
    for (size_t i = 0; i < n; ++i) dst[i] = static_cast<float>(ranges[i]);
#if defined(__x86_64__)
    __builtin_ia32_sfence();
#endif
  } else {
    for (size_t i = 0; i < n; ++i)
      m->h_ranges.p[i] = static_cast<float>(ranges[i]);
    KC_HIP(hipMemcpyAsync(m->d_ranges.p, m->h_ranges.p, n * sizeof(float),
                          hipMemcpyHostToDevice, s));
  }
  const int ni = static_cast<int>(n);
  const dim3 rgrid((ni + kBeamsPerBlock - 1) / kBeamsPerBlock), rblock(64 * kBeamsPerBlock);
  const int hb = (m->g.H + 3) / 4;
  int step_limit = INT_MAX;
  if (m->tiles) {
    KC_TRY(m->d_ends.reserve(n + n / 8 + 2));  // end cells + one sector byte per beam
    KC_TRY(m->timing.start("beam_ends_kernel", s));
    hipLaunchKernelGGL(beam_ends_kernel, dim3((ni + 255) / 256), dim3(256), 0, s, m->g,
                       m->d_ranges.p, m->d_trig.p, ni, m->d_ends.p);
    KC_TRY(m->timing.stop(s));
    const dim3 tgrid((m->g.H + kTileI - 1) / kTileI, (m->g.W + kTileJ - 1) / kTileJ);
    KC_TRY(m->timing.start("scan_tiles_kernel", s));
    if (bayes)
      hipLaunchKernelGGL(scan_tiles_kernel<true>, tgrid, dim3(256), 0, s, m->g, m->d_ends.p, ni,
                         m->d_grid.p, m->d_last.p, hb);
    else
      hipLaunchKernelGGL(scan_tiles_kernel<false>, tgrid, dim3(256), 0, s, m->g, m->d_ends.p, ni,
                         m->d_grid.p, static_cast<unsigned int *>(nullptr), hb);
    KC_TRY(m->timing.stop(s));
    step_limit = kNearSteps;
  }
  const bool hybrid = bayes && m->tile_mode == 3;
  if (hybrid) {
    // far field: beam-parallel, a global atomic per stamp (few beams share a far cell)
    KC_TRY(m->d_ends.reserve(n + n / 8 + 2));  // end cells + one sector byte per beam
    hipLaunchKernelGGL(beam_ends_kernel, dim3((ni + 255) / 256), dim3(256), 0, s, m->g,
                       m->d_ranges.p, m->d_trig.p, ni, m->d_ends.p);
    KC_TRY(m->timing.start("rays_kernel", s));
    hipLaunchKernelGGL(rays_bayes_kernel, rgrid, rblock, 0, s, m->g, m->d_ranges.p, m->d_trig.p,
                       ni, m->d_grid.p, m->d_last.p, hb, kNearSteps + 1, INT_MAX);
    KC_TRY(m->timing.stop(s));
  }
  KC_TRY(m->timing.start(bayes && (m->tiles || hybrid) ? "near_tiles_kernel" : "rays_kernel", s));
  if (bayes && (m->tiles || hybrid)) {
    // tiles that hold cells within kNearSteps + 1 (Chebyshev) of the start cell
    const int ia = std::max(m->g.s0 - kNearSteps - 1, 0), ib = std::min(m->g.s0 + kNearSteps + 1, m->g.H - 1);
    const int ja = std::max(m->g.s1 - kNearSteps - 1, 0), jb = std::min(m->g.s1 + kNearSteps + 1, m->g.W - 1);
    if (ia <= ib && ja <= jb) {
      const int ti0 = ia / kTileI, tj0 = ja / kTileJ;
      const dim3 ngrid(ib / kTileI - ti0 + 1, jb / kTileJ - tj0 + 1, std::min(kNearSlices, std::max(1, ni / 32)));
      hipLaunchKernelGGL(near_tiles_kernel, ngrid, dim3(kTileThreads), 0, s, m->g, m->d_ends.p, ni,
                         m->d_grid.p, m->d_last.p, hb, ti0, tj0);
    }
  } else if (bayes)
    hipLaunchKernelGGL(rays_bayes_kernel, rgrid, rblock, 0, s, m->g, m->d_ranges.p, m->d_trig.p,
                       ni, m->d_grid.p, m->d_last.p, hb, 1, step_limit);
  else  // (a multiple of eight workgroups: the kernel deals the beams to the XCDs by eighths)
    hipLaunchKernelGGL(rays_kernel, dim3((rgrid.x + 7u) & ~7u), rblock, 0, s, m->g, m->d_ranges.p, m->d_trig.p, ni,
                       m->d_grid.p, step_limit);
  KC_TRY(m->timing.stop(s));
  ++m->seq;
  KC_TRY(m->timing.start("endpoints_kernel", s));
  {
    const int beam_blocks = (ni + 255) / 256;
    int clear_blocks = 0;
    int4 *clear = nullptr;
    if (plain && m->d_grid_alt.p && cells % 4 == 0) {
      clear = reinterpret_cast<int4 *>(m->d_grid_alt.p);
      clear_blocks = static_cast<int>(std::min<size_t>(1024, (cells / 4 + 255) / 256));
    }
    hipLaunchKernelGGL(endpoints_kernel, dim3(beam_blocks + clear_blocks), dim3(256), 0, s,
                       m->g, m->d_ranges.p, m->d_trig.p, ni, m->d_grid.p, m->d_ticket.p,
                       bayes ? static_cast<long long *>(nullptr) : m->h_seq.p, m->seq, clear, cells / 4, beam_blocks);
    if (clear) m->alt_cleared = true;
  }
  KC_TRY(m->timing.stop(s));
  if (bayes) {
    KC_TRY(m->timing.start("bayes_cells_kernel", s));
    hipLaunchKernelGGL(bayes_cells_kernel, dim3((m->g.H + 63) / 64, (m->g.W + 3) / 4), dim3(256),
                       0, s, m->g, m->bp, m->d_ranges.p, m->d_last.p, hb, m->d_prev.p,
                       m->d_prob.p);
    KC_TRY(m->timing.stop(s));
    hipLaunchKernelGGL(scan_done_kernel, dim3(1), dim3(1), 0, s, m->h_seq.p, m->seq);
  }
  KC_HIP(hipGetLastError());
  return KC_OK;
}

}  // namespace

namespace kc {
int mapper_view(kc_mapper *m, MapperView *out) {
  if (!m || !out) KC_FAIL(KC_ERR_INVALID, "null argument");
  *out = MapperView{m->d_grid.p, m->g.H, m->g.W, m->g.c0, m->g.c1, m->g.res, m->stream, m->device};
  return KC_OK;
}
}  // namespace kc

extern "C" {

int kc_mapper_create(int H, int W, float res, const float pos[3], float orient,
                     size_t max_scan, int device, kc_mapper **out) {
  if (!out || !pos) KC_FAIL(KC_ERR_INVALID, "null argument");
  *out = nullptr;
  if (H <= 0 || W <= 0 || !(res > 0.0f))
    KC_FAIL(KC_ERR_INVALID, "grid dimensions and resolution must be positive");
  if (static_cast<size_t>(H) * W > 0x3FFFFFFFul)
    KC_FAIL(KC_ERR_RANGE, "grid too large");
  int ndev = 0;
  KC_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev)
    KC_FAIL(KC_ERR_HIP, "HIP device %d not available (%d visible)", device,
            ndev);
  auto *m = new kc_mapper();
  m->device = device;
  m->orient = orient;
  m->g.H = H;
  m->g.W = W;
  m->g.res = res;
  m->g.pos0 = pos[0];
  m->g.pos1 = pos[1];
  // local_mapper.h:26-31: round(gridHeight / 2) - 1 with integer division
  m->g.c0 = static_cast<int>(std::round(static_cast<double>(H / 2))) - 1;
  m->g.c1 = static_cast<int>(std::round(static_cast<double>(W / 2))) - 1;
  m->g.s0 = m->g.c0 + static_cast<int>(pos[0] / res);
  m->g.s1 = m->g.c1 + static_cast<int>(pos[1] / res);
  auto fail = [&](int rc) {
    kc_mapper_destroy(m);
    return rc;
  };
  if (hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) !=
          hipSuccess) {
    set_error("HIP stream creation failed on device %d", device);
    return fail(KC_ERR_HIP);
  }
  m->stream = m->own_stream;
  const size_t cells = static_cast<size_t>(H) * W;
  int rc;
  if ((rc = m->d_grid.reserve(cells)) || (rc = m->d_grid_alt.reserve(cells)) || (rc = m->h_grid.reserve(cells)) ||
      (rc = m->d_ranges.reserve(std::max<size_t>(max_scan, 16))) ||
      (rc = m->d_ticket.reserve(1)) || (rc = m->h_seq.reserve(1)))
    return fail(rc);
  m->h_seq.p[0] = 0;
  if (hipMemset(m->d_ticket.p, 0, sizeof(unsigned int)) != hipSuccess) {
    set_error("ticket initialisation failed");
    return fail(KC_ERR_HIP);
  }
  int large_bar = 0;
  if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess) {
    (void)hipGetLastError();
    large_bar = 0;
  }
  m->direct = large_bar != 0;
  if (const char *e = std::getenv("KC_MAPPER_STAGED"))
    if (e[0] == '1') m->direct = false;  // test hook: the copies of a device without a large BAR
  if (const char *e = std::getenv("KC_MAPPER_TILES"))
    if (e[0] >= '0' && e[0] <= '3') m->tile_mode = e[0] - '0';
  *out = m;
  return KC_OK;
}

void kc_mapper_destroy(kc_mapper *m) {
  if (!m) return;
  hipError_t e = hipSetDevice(m->device);
  if (m->own_stream) {
    e = hipStreamSynchronize(m->own_stream);
    e = hipStreamDestroy(m->own_stream);
  }
  (void)e;
  m->timing.release();
  m->d_grid.release();
  m->d_grid_alt.release();
  m->d_prob.release();
  m->d_prev.release();
  m->d_prev_tmp.release();
  m->d_last.release();
  m->h_prob.release();
  m->d_ranges.release();
  m->d_trig.release();
  m->d_ticket.release();
  m->d_ends.release();
  m->h_seq.release();
  m->h_ranges.release();
  m->h_trig.release();
  m->h_grid.release();
  delete m;
}

int kc_mapper_set_stream(kc_mapper *m, void *hip_stream) {
  if (!m) KC_FAIL(KC_ERR_INVALID, "null context");
  KC_HIP(hipSetDevice(m->device));
  KC_HIP(hipStreamSynchronize(m->stream));
  m->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : m->own_stream;
  return KC_OK;
}

int kc_mapper_scan_to_grid_device(kc_mapper *m, const double *angles,
                                  const double *ranges, size_t n) {
  if (!m || (n && (!angles || !ranges)))
    KC_FAIL(KC_ERR_INVALID, "null argument");
  return run_scan(m, angles, ranges, n);
}

int kc_mapper_scan_to_grid(kc_mapper *m, const double *angles,
                           const double *ranges, size_t n, int32_t *grid_out) {
  if (!grid_out) KC_FAIL(KC_ERR_INVALID, "null argument");
  KC_TRY(kc_mapper_scan_to_grid_device(m, angles, ranges, n));
  const size_t cells = static_cast<size_t>(m->g.H) * m->g.W;
  KC_HIP(hipMemcpyAsync(m->h_grid.p, m->d_grid.p, cells * sizeof(int),
                        hipMemcpyDeviceToHost, m->stream));
  KC_HIP(hipStreamSynchronize(m->stream));
  std::memcpy(grid_out, m->h_grid.p, cells * sizeof(int));
  return KC_OK;
}

// ---- M3 ------------------------------------------------------------------------
int kc_mapper_enable_bayes(kc_mapper *m, const kc_bayes_params *p) {
  if (!m || !p) KC_FAIL(KC_ERR_INVALID, "null argument");
  KC_HIP(hipSetDevice(m->device));
  KC_HIP(hipStreamSynchronize(m->stream));
  const size_t cells = static_cast<size_t>(m->g.H) * m->g.W;
  KC_TRY(m->d_prob.reserve(cells));
  KC_TRY(m->d_prev.reserve(cells));
  KC_TRY(m->d_prev_tmp.reserve(cells));
  // tags in 4x4-cell blocks, one 64-byte line each
  const size_t tag_words = static_cast<size_t>((m->g.H + 3) / 4) * ((m->g.W + 3) / 4) * 16;
  KC_TRY(m->d_last.reserve(tag_words));
  KC_TRY(m->h_prob.reserve(cells));
  m->bp = BayesParams{p->p_prior, p->p_occupied, p->p_empty, p->range_sure, p->range_max,
                      p->wall_size};
  // previousGridDataProb.fill(m_pPrior), local_mapper.h:81-83
  KC_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(m->d_prev.p),
                           float_bits(p->p_prior), cells, m->stream));
  KC_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(m->d_prob.p),
                           float_bits(p->p_prior), cells, m->stream));
  KC_HIP(hipMemsetAsync(m->d_last.p, 0, tag_words * sizeof(unsigned int), m->stream));
  KC_HIP(hipStreamSynchronize(m->stream));
  m->bayes = true;
  return KC_OK;
}

int kc_mapper_scan_to_grid_bayes_device(kc_mapper *m, const double *angles, const double *ranges,
                                        size_t n) {
  if (!m || (n && (!angles || !ranges))) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (!m->bayes) KC_FAIL(KC_ERR_INVALID, "kc_mapper_enable_bayes has not been called");
  return run_scan(m, angles, ranges, n, true);
}

int kc_mapper_scan_to_grid_bayes(kc_mapper *m, const double *angles, const double *ranges,
                                 size_t n, int32_t *grid_out, float *prob_out) {
  if (!grid_out || !prob_out) KC_FAIL(KC_ERR_INVALID, "null argument");
  KC_TRY(kc_mapper_scan_to_grid_bayes_device(m, angles, ranges, n));
  const size_t cells = static_cast<size_t>(m->g.H) * m->g.W;
  KC_HIP(hipMemcpyAsync(m->h_grid.p, m->d_grid.p, cells * sizeof(int), hipMemcpyDeviceToHost,
                        m->stream));
  KC_HIP(hipMemcpyAsync(m->h_prob.p, m->d_prob.p, cells * sizeof(float), hipMemcpyDeviceToHost,
                        m->stream));
  KC_HIP(hipStreamSynchronize(m->stream));
  std::memcpy(grid_out, m->h_grid.p, cells * sizeof(int));
  std::memcpy(prob_out, m->h_prob.p, cells * sizeof(float));
  return KC_OK;
}

int kc_mapper_prob_device(kc_mapper *m, void **dev_prob, void **dev_prev) {
  if (!m || !dev_prob) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (!m->bayes) KC_FAIL(KC_ERR_INVALID, "kc_mapper_enable_bayes has not been called");
  *dev_prob = m->d_prob.p;
  if (dev_prev) *dev_prev = m->d_prev.p;
  return KC_OK;
}

int kc_mapper_warp_previous(kc_mapper *m, const float pos[2], double orient) {
  if (!m || !pos) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (!m->bayes) KC_FAIL(KC_ERR_INVALID, "kc_mapper_enable_bayes has not been called");
  KC_HIP(hipSetDevice(m->device));
  const MapGeom &g = m->g;
  // localToGrid of the new centre, then the Matrix3f of local_mapper.cpp:27-37
  const int cc0 = g.c0 + static_cast<int>(pos[0] / g.res);
  const int cc1 = g.c1 + static_cast<int>(pos[1] / g.res);
  const double ang = -1 * orient;
  const double c = std::cos(ang), sn = std::sin(ang);
  float t[3][3];
  t[0][0] = static_cast<float>(c);
  t[0][1] = static_cast<float>(-sn);
  t[0][2] = static_cast<float>(0.5 * g.H - cc1 + (cc0 * sn - cc1 * c));
  t[1][0] = static_cast<float>(sn);
  t[1][1] = static_cast<float>(c);
  t[1][2] = static_cast<float>(0.5 * g.W - cc0 - (cc0 * c + cc1 * sn));
  t[2][0] = 0.0f;
  t[2][1] = 0.0f;
  t[2][2] = 1.0f;
  // Eigen's 3x3 inverse: cofactors, det over column 0 as a0 + (a1 + a2)
  auto cof = [&](int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return t[i1][j1] * t[i2][j2] - t[i1][j2] * t[i2][j1];
  };
  const float k0 = cof(0, 0), k1 = cof(1, 0), k2 = cof(2, 0);
  const float det = k0 * t[0][0] + (k1 * t[1][0] + k2 * t[2][0]);
  const float invdet = 1.0f / det;
  WarpArgs a{};
  a.r0[0] = k0 * invdet;
  a.r0[1] = k1 * invdet;
  a.r0[2] = k2 * invdet;
  for (int q = 0; q < 3; ++q) a.r1[q] = cof(q, 1) * invdet;
  a.H = g.H;
  a.W = g.W;
  a.prior = m->bp.p_prior;
  const unsigned int cells = static_cast<unsigned int>(static_cast<size_t>(g.H) * g.W);
  m->timing.begin_cycle();
  KC_TRY(m->timing.start("warp_kernel", m->stream));
  hipLaunchKernelGGL(warp_kernel, dim3((cells + 255u) / 256u), dim3(256), 0, m->stream, a,
                     m->d_prev.p, m->d_prev_tmp.p, cells);
  KC_TRY(m->timing.stop(m->stream));
  KC_HIP(hipGetLastError());
  std::swap(m->d_prev, m->d_prev_tmp);  // stream order covers the next reader
  return KC_OK;
}

int kc_mapper_get_previous_prob(kc_mapper *m, float *out) {
  if (!m || !out) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (!m->bayes) KC_FAIL(KC_ERR_INVALID, "kc_mapper_enable_bayes has not been called");
  KC_HIP(hipSetDevice(m->device));
  const size_t cells = static_cast<size_t>(m->g.H) * m->g.W;
  KC_HIP(hipMemcpyAsync(m->h_prob.p, m->d_prev.p, cells * sizeof(float), hipMemcpyDeviceToHost,
                        m->stream));
  KC_HIP(hipStreamSynchronize(m->stream));
  std::memcpy(out, m->h_prob.p, cells * sizeof(float));
  return KC_OK;
}

int kc_mapper_set_previous_prob(kc_mapper *m, const float *in) {
  if (!m) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (!m->bayes) KC_FAIL(KC_ERR_INVALID, "kc_mapper_enable_bayes has not been called");
  KC_HIP(hipSetDevice(m->device));
  const size_t cells = static_cast<size_t>(m->g.H) * m->g.W;
  if (in) {
    KC_HIP(hipStreamSynchronize(m->stream));  // h_prob may still be in flight
    std::memcpy(m->h_prob.p, in, cells * sizeof(float));
    KC_HIP(hipMemcpyAsync(m->d_prev.p, m->h_prob.p, cells * sizeof(float), hipMemcpyHostToDevice,
                          m->stream));
    KC_HIP(hipStreamSynchronize(m->stream));
  } else {
    // feed the last probability grid back as the next scan's prior
    KC_HIP(hipMemcpyAsync(m->d_prev.p, m->d_prob.p, cells * sizeof(float),
                          hipMemcpyDeviceToDevice, m->stream));
  }
  return KC_OK;
}

int kc_mapper_grid_device(kc_mapper *m, void **dev) {
  if (!m || !dev) KC_FAIL(KC_ERR_INVALID, "null argument");
  *dev = m->d_grid.p;
  return KC_OK;
}

int kc_mapper_sync(kc_mapper *m) {
  if (!m) KC_FAIL(KC_ERR_INVALID, "null context");
  KC_HIP(hipSetDevice(m->device));
  // the last endpoints workgroup reports the scan into pinned memory: poll it
  // (a stream wait costs ~10 us), fall back to the stream after 2 ms
  if (m->seq != 0 && scan_done(m, 2000)) return KC_OK;
  KC_HIP(hipStreamSynchronize(m->stream));
  return KC_OK;
}

int kc_mapper_timing_enable(kc_mapper *m, int enable) {
  if (!m) KC_FAIL(KC_ERR_INVALID, "null context");
  m->timing.enabled = enable != 0;
  return KC_OK;
}

int kc_mapper_timing_get(kc_mapper *m, const char **names, float *ms,
                         size_t cap, size_t *count) {
  if (!m) KC_FAIL(KC_ERR_INVALID, "null context");
  KC_HIP(hipSetDevice(m->device));
  return m->timing.get(names, ms, cap, count);
}

}  // extern "C"
