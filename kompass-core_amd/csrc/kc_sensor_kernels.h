// Sensor update on the device (A4 + A8 inputs): world-frame points -> occupancy
// bitmap of the accepted voxel columns + obstacle buckets (cell table, skip
// table, coordinates in cell order), everything the roll-out and cost kernels
// read.  Up to 32 k points: ONE launch without any hand-over between workgroups
// (sensor_fused_kernel, kc_onear_kernels.h); beyond: the two launches below.  Same
// arithmetic as the host path (add_voxel / Rigid3f::apply / cell index); the order
// of the points inside a bucket is arbitrary on every path (only minima are taken
// over them).  Part of kc_dwa.hip.
#pragma once

namespace kc {

struct SensorArgs {
  TrigJob trig;      // the cycle's cos / sin table, formed by the last trig.nblk workgroups of the launch
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
  int obs_z_zero;            // laserscan: the obstacle of a point is taken at z = 0
  // spheres (sensor_fused_kernel only): a voxel is accepted by its LAYER -- zcode[kz - kz0] != 0, the host evaluates
  // add_voxel's gap rule once per layer -- and the code is the rank of the layer's z gap among the layers the cloud's
  // z range can hold (d_zlut: ascending gaps); a column keeps its smallest code (d_gz, one byte a column)
  int sphere, kz0, nkz;
  unsigned char zcode[36];
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

// ---------------------------------------------------------------------------
// The same structure for LARGE clouds (more than 32 k points, or bitmap bands beyond
// the LDS budget of the one-launch build), two launches and
// NO global atomics: device-scope atomics execute at the memory side on this
// chip (55 k of them took 13 us, and 18 us with sixteen counters per cache line
// queueing behind each other).
//   sensor_points_kernel  one workgroup per 1024 * ppt points: voxel -> ONE BYTE of a
//                         byte map (plain idempotent stores; the L2s merge bytes), bucket
//                         rank from a histogram in LDS, the histogram written as one row
//                         of a [workgroups][4096] matrix
//   sensor_place_kernel   the same workgroups: each sums the matrix columns (its own row's
//                         prefix on the way) and scans them (redundant, 2 us, instead of a
//                         launch in between), puts its points into cell order and writes its
//                         share of cell starts / skip table; further workgroups of the same
//                         launch pack the byte map into the bitmap, clearing it behind (so
//                         neither needs a memset)
// Same per-point arithmetic as above; the order of the points inside a bucket is
// as arbitrary as there.  Reference step: collision_check.h:91-136 (octree
// rebuild) + cost_evaluator.h:174-223 (setPointScan).
// ---------------------------------------------------------------------------
constexpr int kHistRow = 64 * 64;  // ints per row of the histogram matrix
constexpr int kHistRowsMax = 16;   // the host picks points per thread so that the rows fit
struct SensorBigArgs {
  SensorArgs a;
  int ppt;          // points per thread (1 .. 16): a workgroup takes 1024 * ppt consecutive points
  int rows;         // workgroups of the points kernel = rows of hist
  int *hist;        // [rows][kHistRow] points per bucket of each workgroup
  uint8_t *bytes;   // [gH][gwpr * 32] zero on entry, zero again when sensor_place_kernel is done
  float *tox, *toy; // [n] transformed coordinates (scratch)
  int *tcell;       // [n] cell id | rank inside its workgroup (< 16 k) << 12, -1: not an obstacle
  unsigned long long *dbg;  // KC_PHASE_STAMPS builds: [16 workgroups][16] s_memrealtime stamps, or null
};
#ifdef KC_PHASE_STAMPS
#define KC_SSTAMP(row, slot)                                                                         \
  do {                                                                                               \
    if (b.dbg && threadIdx.x == 0 && (row) < 16) b.dbg[(row) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define KC_SSTAMP(row, slot) do { } while (0)
#endif

#ifdef KC_TU_SENSOR  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(kSensorBlock) void sensor_points_kernel(SensorBigArgs b) {
  const SensorArgs &a = b.a;
  __shared__ __align__(16) int lhist[kHistRow];
  const int tid = threadIdx.x;
  if (static_cast<int>(blockIdx.x) >= b.rows) {  // (the workgroups behind the rows: the trig job)
    trig_job_block<kSensorBlock>(a.trig, static_cast<int>(blockIdx.x) - b.rows);
    return;
  }
  KC_SSTAMP(blockIdx.x, 0);
  reinterpret_cast<int4 *>(lhist)[tid] = make_int4(0, 0, 0, 0);
  __syncthreads();
  KC_SSTAMP(blockIdx.x, 1);
  auto point = [&](int i, float x, float y, float z) {
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
          b.bytes[(static_cast<size_t>(cy) * a.gwpr << 5) + cx] = 1;
      }
    }
    float ox, oy;
    int id;
    int rec = -1;
    if (sensor_obstacle(a, x, y, a.obs_z_zero ? 0.0f : z, ox, oy, id)) {
      rec = id | (atomicAdd(&lhist[id], 1) << 12);  // id < 4096 cells, rank < 16 k
      b.tox[i] = ox;
      b.toy[i] = oy;
    }
    b.tcell[i] = rec;
  };
  // four points per trip, their loads issued together
  const int i0 = blockIdx.x * kSensorBlock * b.ppt;
  for (int q = 0; q < b.ppt; q += 4) {
    float px[4], py[4], pz[4];
    int pi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + (q + u) * kSensorBlock + tid;
      pi[u] = (q + u < b.ppt && i < a.n) ? i : -1;
      const int j = pi[u] >= 0 ? i : 0;  // idle slots shadow point 0, used for nothing
      px[u] = a.xyz[3 * j];
      py[u] = a.xyz[3 * j + 1];
      pz[u] = a.xyz[3 * j + 2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (pi[u] >= 0) point(pi[u], px[u], py[u], pz[u]);
  }
  __syncthreads();
  KC_SSTAMP(blockIdx.x, 2);
  reinterpret_cast<int4 *>(b.hist + static_cast<size_t>(blockIdx.x) * kHistRow)[tid] = reinterpret_cast<int4 *>(lhist)[tid];
  KC_SSTAMP(blockIdx.x, 3);
}
#endif  // KC_TU_SENSOR

#ifdef KC_TU_SENSOR  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(kSensorBlock) void sensor_place_kernel(SensorBigArgs b) {
  const SensorArgs &a = b.a;
  const int ncell = a.W * a.H;
  __shared__ __align__(16) int lstart[kHistRow + 4];  // slot k + 1: count, then start, of cell k
  __shared__ __align__(16) int lbase[kHistRow];       // points of the cell in the workgroups before this one
  __shared__ unsigned long long lmask[64];
  __shared__ int wave_tot[kSensorBlock / 64];
  const int tid = threadIdx.x;
  const int me = blockIdx.x;
  // ---- byte map -> bitmap, bytes cleared behind: the workgroups beyond the rows do only this -------------
  if (me >= b.rows) {
    KC_SSTAMP(me - b.rows, 10);
    const size_t nwords = static_cast<size_t>(a.gH) * a.gwpr;
    const size_t stride = static_cast<size_t>(gridDim.x - b.rows) * kSensorBlock;
    for (size_t w0 = static_cast<size_t>(me - b.rows) * kSensorBlock + tid; w0 < nwords; w0 += 4 * stride) {
      // bytes are 0 / 1: (v * 0x0102040810204080) >> 56 gathers byte i into bit i (all partial products
      // fall on distinct bit positions: no carries).  Four words per trip, their loads issued together.
      constexpr unsigned long long kGather = 0x0102040810204080ull;
      ulonglong2 lo[4], hi[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t w = w0 + u * stride;
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(b.bytes + ((w < nwords ? w : w0) << 5));
        lo[u] = src[0];
        hi[u] = src[1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t w = w0 + u * stride;
        if (w >= nwords) break;
        const uint32_t bits = static_cast<uint32_t>((lo[u].x * kGather) >> 56) | static_cast<uint32_t>((lo[u].y * kGather) >> 56) << 8 |
                              static_cast<uint32_t>((hi[u].x * kGather) >> 56) << 16 | static_cast<uint32_t>((hi[u].y * kGather) >> 56) << 24;
        a.gbits[w] = bits;
        if (bits) {
          ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(b.bytes + (w << 5));
          dst[0] = make_ulonglong2(0ull, 0ull);
          dst[1] = make_ulonglong2(0ull, 0ull);
        }
      }
    }
    KC_SSTAMP(me - b.rows, 11);
    return;
  }
  KC_SSTAMP(me, 4);
  // this workgroup's first point records, asked for now and used at the very end
  const int i_first = me * kSensorBlock * b.ppt + tid;
  const bool have_first = i_first < a.n;
  const int rec_first = have_first ? b.tcell[i_first] : -1;
  const float ox_first = have_first ? b.tox[i_first] : 0.0f, oy_first = have_first ? b.toy[i_first] : 0.0f;
  // ---- column sums of the histogram matrix (four cells per thread) -----------------------------------------
  {
    int4 acc = make_int4(0, 0, 0, 0), mine = make_int4(0, 0, 0, 0);
    for (int w0 = 0; w0 < b.rows; w0 += 16) {  // sixteen loads in flight (all of them: the host keeps rows <= 16)
      int4 h[16];
#pragma unroll
      for (int j = 0; j < 16; ++j)
        h[j] = w0 + j < b.rows ? reinterpret_cast<const int4 *>(b.hist + static_cast<size_t>(w0 + j) * kHistRow)[tid]
                               : make_int4(0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (w0 + j == me) mine = acc;
        acc.x += h[j].x;
        acc.y += h[j].y;
        acc.z += h[j].z;
        acc.w += h[j].w;
      }
    }
    if (tid == 0) {
      lstart[0] = 0;
    }
    lstart[4 * tid + 1] = acc.x;
    lstart[4 * tid + 2] = acc.y;
    lstart[4 * tid + 3] = acc.z;
    lstart[4 * tid + 4] = acc.w;
    reinterpret_cast<int4 *>(lbase)[tid] = mine;
  }
  __syncthreads();
  KC_SSTAMP(me, 5);
  // ---- starts: in-place inclusive scan of the ncell + 1 slots (consecutive slots per thread, <= 8: the host
  // keeps the grid at 64 x 64; wave scan of the thread totals)
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
      if (k < per && idx < N) lstart[idx] = offset + v[k];
    }
  }
  __syncthreads();
  KC_SSTAMP(me, 6);
  // ---- what the cost kernels read: cell starts, skip table, dc_enable -- 64 cells per wavefront, the wavefronts
  // of all workgroups interleaved
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int y = wave; y < a.H; y += kSensorBlock / 64) {
      const bool ne = lane < a.W && lstart[y * a.W + lane + 1] > lstart[y * a.W + lane];
      const unsigned long long m = __ballot(ne);
      if (lane == 0) {
        lmask[y] = m;
      }
    }
    __syncthreads();
    KC_SSTAMP(me, 7);
    if (me == 0 && tid < 4) a.skip[ncell + tid] = 255;  // word padding the cost kernels copy
    const int nb = b.rows;
    for (int k = (wave * nb + me) * 64 + lane; k <= ncell; k += nb * kSensorBlock) {
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
  }
  KC_SSTAMP(me, 8);
  // ---- this workgroup's points into cell order
  if (rec_first >= 0) {
    const int id = rec_first & 4095;
    const int pos = lstart[id] + lbase[id] + (rec_first >> 12);
    a.bx[pos] = ox_first;
    a.by[pos] = oy_first;
  }
  for (int q = 1; q < b.ppt; ++q) {
    const int i = i_first + q * kSensorBlock;
    if (i >= a.n) break;
    const int rec = b.tcell[i];
    if (rec < 0) continue;
    const int id = rec & 4095;
    const int pos = lstart[id] + lbase[id] + (rec >> 12);
    a.bx[pos] = b.tox[i];
    a.by[pos] = b.toy[i];
  }
  KC_SSTAMP(me, 9);
}
#endif  // KC_TU_SENSOR

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
  unsigned int *cnt;    // count, then (as int) imin, imax, jmin, jmax -- one 64-byte line each
                        // (word k at cnt[16 k]): atomics on one line serialise
};
constexpr int kGridCntStride = 16;

#ifdef KC_TU_SENSOR  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(256) void grid_points_kernel(GridPtsArgs a) {
  // hits are counted and bounded per workgroup in LDS first: one slot range and
  // four bound updates per workgroup that holds a hit (same-address global
  // atomics serialise at ~90 per microsecond)
  __shared__ unsigned int s_hits, s_base;
  __shared__ int s_lo_i, s_hi_i, s_lo_j, s_hi_j;
  if (threadIdx.x == 0) {
    s_hits = 0u;
    s_lo_i = INT_MAX;
    s_hi_i = INT_MIN;
    s_lo_j = INT_MAX;
    s_hi_j = INT_MIN;
  }
  __syncthreads();
  const unsigned int k = blockIdx.x * 256u + threadIdx.x;
  const unsigned int cells = static_cast<unsigned int>(a.H) * static_cast<unsigned int>(a.W);
  const bool hit = k < cells && a.grid[k] == KC_OCCUPIED;
  const unsigned long long m = __ballot(hit);
  const int lane = threadIdx.x & 63;
  const int i = static_cast<int>(k % static_cast<unsigned int>(a.H));
  const int j = static_cast<int>(k / static_cast<unsigned int>(a.H));
  unsigned int rank = 0;
  if (m != 0ull) {
    const int leader = __ffsll(static_cast<long long>(m)) - 1;
    unsigned int wbase = 0;
    if (lane == leader) wbase = atomicAdd(&s_hits, static_cast<unsigned int>(__popcll(m)));
    wbase = __shfl(wbase, leader, 64);
    rank = wbase + __popcll(m & ((1ull << lane) - 1ull));
    if (hit) {
      atomicMin(&s_lo_i, i);
      atomicMax(&s_hi_i, i);
      atomicMin(&s_lo_j, j);
      atomicMax(&s_hi_j, j);
    }
  }
  __syncthreads();
  if (s_hits == 0u) return;
  if (threadIdx.x == 0) {
    s_base = atomicAdd(&a.cnt[0], s_hits);
    int *b = reinterpret_cast<int *>(a.cnt);
    atomicMin(&b[1 * kGridCntStride], s_lo_i);
    atomicMax(&b[2 * kGridCntStride], s_hi_i);
    atomicMin(&b[3 * kGridCntStride], s_lo_j);
    atomicMax(&b[4 * kGridCntStride], s_hi_j);
  }
  __syncthreads();
  if (hit) {
    const size_t slot = static_cast<size_t>(s_base) + rank;
    a.xyz[3 * slot] = static_cast<float>(i - a.c0) * a.res;
    a.xyz[3 * slot + 1] = static_cast<float>(j - a.c1) * a.res;
    a.xyz[3 * slot + 2] = 0.0f;
  }
}
#endif  // KC_TU_SENSOR

// one thread behind it (the kernel boundary has made the counters visible):
// count + bounds into pinned host memory, counters re-armed
#ifdef KC_TU_SENSOR  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ void grid_points_publish_kernel(unsigned int *cnt, long long *host, long long seq) {
  int *b = reinterpret_cast<int *>(cnt);
  host[1] = cnt[0];
  host[2] = b[1 * kGridCntStride];
  host[3] = b[2 * kGridCntStride];
  host[4] = b[3 * kGridCntStride];
  host[5] = b[4 * kGridCntStride];
  cnt[0] = 0u;
  b[1 * kGridCntStride] = INT_MAX;
  b[2 * kGridCntStride] = INT_MIN;
  b[3 * kGridCntStride] = INT_MAX;
  b[4 * kGridCntStride] = INT_MIN;
  __threadfence_system();
  *reinterpret_cast<volatile long long *>(host) = seq;
}
#endif  // KC_TU_SENSOR

}  // namespace kc
