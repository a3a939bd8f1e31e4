// Sensor update on the device (A4 + A8 inputs): world-frame points -> occupancy
// bitmap of the accepted voxel columns + obstacle buckets (cell table, skip
// table, coordinates in cell order), everything the roll-out and cost kernels
// read.  One workgroup: at costmap sizes (<= 16k points, bitmap <= 64 KB) the
// whole structure fits its LDS, so every scatter is an LDS atomic and the
// results leave as plain coalesced stores.  Same arithmetic as the host path
// (add_voxel / Rigid3f::apply / cell index); the order of the points inside a
// bucket is arbitrary on both paths (only minima are taken over them).
// Part of kc_dwa.hip.
#pragma once

namespace kc {

struct SensorArgs {
  const float *xyz;  // [n][3]
  int n;
  // voxel acceptance (add_voxel)
  double inv_res, res, zc, half_height;
  int gkx0, gky0, gH, gwpr;  // bitmap extent (keys), rows, words per row
  uint32_t *gbits;           // [gH][gwpr]
  // obstacle transform (sensor_tf_body * body, float isometry) and bucket grid
  float R[3][3], t[3];
  double gx0, gy0, inv_g;
  int W, H;
  int *cell_start;           // [W*H + 1]
  uint8_t *skip;             // [W*H] (+ padding written by the host)
  float *bx, *by;            // cell-ordered coordinates
  float4 *tmp;               // [n] (ox, oy, cell id, rank inside the cell) between the two passes
  int obs_z_zero;            // laserscan: the obstacle of a point is taken at z = 0
};

constexpr int kSensorBlock = 1024;

__device__ __forceinline__ bool sensor_obstacle(const SensorArgs &a, float x, float y, float z,
                                                float &ox, float &oy, int &id) {
  // Rigid3f::apply: t + (R0*x + (R1*y + R2*z))
  ox = a.t[0] + (a.R[0][0] * x + (a.R[0][1] * y + a.R[0][2] * z));
  oy = a.t[1] + (a.R[1][0] * x + (a.R[1][1] * y + a.R[1][2] * z));
  if (!isfinite(ox) || !isfinite(oy)) return false;
  int cx = static_cast<int>((static_cast<double>(ox) - a.gx0) * a.inv_g);
  int cy = static_cast<int>((static_cast<double>(oy) - a.gy0) * a.inv_g);
  cx = min(max(cx, 0), a.W - 1);
  cy = min(max(cy, 0), a.H - 1);
  id = cy * a.W + cx;
  return true;
}

__global__ __launch_bounds__(kSensorBlock) void sensor_build_kernel(SensorArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int nw = a.gH * a.gwpr, ncell = a.W * a.H;
  uint32_t *lbits = reinterpret_cast<uint32_t *>(smem);   // [nw]
  int *lstart = reinterpret_cast<int *>(lbits + nw);      // [ncell + 1]: counts, then starts
  unsigned long long *lmask = reinterpret_cast<unsigned long long *>(
      smem + ((static_cast<size_t>(nw) * 4 + static_cast<size_t>(ncell + 1) * 4 + 7) & ~size_t(7)));  // [H]
  __shared__ int wave_tot[kSensorBlock / 64];
  const int tid = threadIdx.x;
  for (int i = tid; i < nw; i += kSensorBlock) lbits[i] = 0u;
  for (int i = tid; i <= ncell; i += kSensorBlock) lstart[i] = 0;
  __syncthreads();
  // ---- 1: voxel bits + bucket counts ----------------------------------------
  for (int i = tid; i < a.n; i += kSensorBlock) {
    const float x = a.xyz[3 * i], y = a.xyz[3 * i + 1], z = a.xyz[3 * i + 2];
    // add_voxel: keys, octree range, z interval of the robot (cylinder / box)
    const double fx = floor(a.inv_res * static_cast<double>(x));
    const double fy = floor(a.inv_res * static_cast<double>(y));
    const double fz = floor(a.inv_res * static_cast<double>(z));
    if (fabs(fx) < 32768.0 && fabs(fy) < 32768.0 && fabs(fz) < 32768.0) {
      const int kz = static_cast<int>(fz);
      const double zlo = static_cast<double>(kz) * a.res;
      const double zhi = static_cast<double>(kz + 1) * a.res;
      if (zlo <= a.zc + a.half_height && zhi >= a.zc - a.half_height) {
        const int cx = static_cast<int>(fx) - a.gkx0, cy = static_cast<int>(fy) - a.gky0;
        if (cx >= 0 && cy >= 0 && cy < a.gH && (cx >> 5) < a.gwpr)
          atomicOr(&lbits[cy * a.gwpr + (cx >> 5)], 1u << (cx & 31));
      }
    }
    float ox, oy;
    int id;
    float4 rec = make_float4(0.f, 0.f, __int_as_float(-1), 0.f);
    if (sensor_obstacle(a, x, y, a.obs_z_zero ? 0.0f : z, ox, oy, id))
      rec = make_float4(ox, oy, __int_as_float(id), __int_as_float(atomicAdd(&lstart[id + 1], 1)));
    a.tmp[i] = rec;
  }
  __syncthreads();
  // ---- 2: cell starts.  The count of cell k sits in slot k + 1 (slot 0 is 0), so
  // an in-place inclusive scan of the ncell + 1 slots is the exclusive scan of
  // the counts, total in the last slot.  Consecutive slots per thread (<= 8:
  // the host keeps the grid at 64 x 64), wave scan of the thread totals.
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
    const int lane = tid & 63, wave = tid >> 6;
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
      if (k < per && idx < N) {
        const int s = offset + v[k];
        lstart[idx] = s;
      }
    }
  }
  __syncthreads();
  for (int k = tid; k <= ncell; k += kSensorBlock) a.cell_start[k] = lstart[k];
  // ---- 3: skip table = Chebyshev distance to the nearest non-empty cell.  A row
  // of the grid (<= 64 cells) is one 64-bit mask of its non-empty cells; the
  // distance of cell (x, y) is the smallest r for which the rows y-r..y+r hold
  // a set bit in columns x-r..x+r.
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int y = wave; y < a.H; y += kSensorBlock / 64) {
      const bool ne = lane < a.W && lstart[y * a.W + lane + 1] > lstart[y * a.W + lane];
      const unsigned long long m = __ballot(ne);
      if (lane == 0) lmask[y] = m;
    }
  }
  __syncthreads();
  for (int k = tid; k < ncell; k += kSensorBlock) {
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
  if (tid < 4) a.skip[ncell + tid] = 255;  // word padding the cost kernels copy
  // ---- 4: scatter the coordinates into their cells (start + rank) --------------
  for (int i = tid; i < a.n; i += kSensorBlock) {
    const float4 rec = a.tmp[i];
    const int id = __float_as_int(rec.z);
    if (id >= 0) {
      const int pos = lstart[id] + __float_as_int(rec.w);
      a.bx[pos] = rec.x;
      a.by[pos] = rec.y;
    }
  }
  // ---- 5: the bitmap --------------------------------------------------------------
  for (int i = tid; i < nw; i += kSensorBlock) a.gbits[i] = lbits[i];
}

// ---- occupancy grid -> point list on the device ------------
// One thread per cell of a column-major int32 grid (LocalMapper layout, cell
// (i,j) at i + j*H): OCCUPIED cells become points ((i - c0) res, (j - c1) res, 0)
// -- the inverse of LocalMapper::localToGrid (local_mapper.h:210-222) -- in a
// list whose order is arbitrary (every consumer takes sets or minima).  One
// counter add per wavefront that holds a hit; the index bounds of the hits
// let the host size the voxel bitmap and the bucket grid without seeing a point.
struct GridPtsArgs {
  const int *grid;
  int H, W, c0, c1;
  float res;
  float *xyz;           // [H*W][3] capacity
  unsigned int *cnt;    // count, then (as int) imin, imax, jmin, jmax
};

__global__ __launch_bounds__(256) void grid_points_kernel(GridPtsArgs a) {
  const unsigned int k = blockIdx.x * 256u + threadIdx.x;
  const unsigned int cells = static_cast<unsigned int>(a.H) * static_cast<unsigned int>(a.W);
  const bool hit = k < cells && a.grid[k] == KC_OCCUPIED;
  const unsigned long long m = __ballot(hit);
  if (m == 0ull) return;
  const int lane = threadIdx.x & 63;
  unsigned int base = 0;
  if (lane == __ffsll(static_cast<long long>(m)) - 1) base = atomicAdd(&a.cnt[0], __popcll(m));
  base = __shfl(base, __ffsll(static_cast<long long>(m)) - 1, 64);
  if (hit) {
    const int i = static_cast<int>(k % static_cast<unsigned int>(a.H));
    const int j = static_cast<int>(k / static_cast<unsigned int>(a.H));
    const unsigned int slot = base + __popcll(m & ((1ull << lane) - 1ull));
    a.xyz[3 * static_cast<size_t>(slot)] = static_cast<float>(i - a.c0) * a.res;
    a.xyz[3 * static_cast<size_t>(slot) + 1] = static_cast<float>(j - a.c1) * a.res;
    a.xyz[3 * static_cast<size_t>(slot) + 2] = 0.0f;
    int *b = reinterpret_cast<int *>(a.cnt);
    atomicMin(&b[1], i);
    atomicMax(&b[2], i);
    atomicMin(&b[3], j);
    atomicMax(&b[4], j);
  }
}

// one thread behind it (the kernel boundary has made the counters visible):
// count + bounds into pinned host memory, counters re-armed
__global__ void grid_points_publish_kernel(unsigned int *cnt, long long *host, long long seq) {
  int *b = reinterpret_cast<int *>(cnt);
  host[1] = cnt[0];
  host[2] = b[1];
  host[3] = b[2];
  host[4] = b[3];
  host[5] = b[4];
  cnt[0] = 0u;
  b[1] = INT_MAX;
  b[2] = INT_MIN;
  b[3] = INT_MAX;
  b[4] = INT_MIN;
  __threadfence_system();
  *reinterpret_cast<volatile long long *>(host) = seq;
}

}  // namespace kc
