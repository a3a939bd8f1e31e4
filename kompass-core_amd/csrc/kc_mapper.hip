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
    int *__restrict__ grid) {
  const int beam = blockIdx.x * kBeamsPerBlock + (threadIdx.x >> 6);
  if (beam >= n) return;
  const int lane = threadIdx.x & 63;
  const int2 t = beam_endpoint(g, ranges[beam], trig[beam]);
  int dx = t.x - g.s0, dy = t.y - g.s1;
  const int xstep = dx >= 0 ? 1 : -1, ystep = dy >= 0 ? 1 : -1;
  dx = abs(dx);
  dy = abs(dy);
  if (lane == 0) stamp_empty(grid, g, g.s0, g.s1);  // first emitted point
  const bool xmajor = 2 * dx >= 2 * dy;
  const int nsteps = xmajor ? dx : dy;
  if (nsteps == 0) return;
  const long long dmaj = xmajor ? dx : dy, dmin = xmajor ? dy : dx;
  const long long ddmaj = 2 * dmaj, ddmin = 2 * dmin;
  const int chunk = (nsteps + 63) / 64;
  const int i0 = lane * chunk + 1;  // first step of this lane (1-based)
  const int i1 = min(nsteps, i0 + chunk - 1);
  if (i0 > i1) return;
  // state after step i0-1
  const long long eprev = dmaj + (long long)(i0 - 1) * ddmin;
  long long k = static_cast<long long>(
      floor(static_cast<double>(eprev - 1) / static_cast<double>(ddmaj)));
  if (eprev - 1 < 0) k = 0;  // dmaj >= 1 here, kept for clarity
  long long error = eprev - k * ddmaj;
  int a = (xmajor ? g.s0 : g.s1) + (xmajor ? xstep : ystep) * (i0 - 1);
  int bq = (xmajor ? g.s1 : g.s0) + (xmajor ? ystep : xstep) * (int)k;
  const int astep = xmajor ? xstep : ystep, bstep = xmajor ? ystep : xstep;
  for (int i = i0; i <= i1; ++i) {
    const long long errorprev = error;
    a += astep;
    error += ddmin;
    if (error > ddmaj) {
      bq += bstep;
      error -= ddmaj;
      // (a, b) major/minor coordinates -> (x, y)
      if (error + errorprev < ddmaj) {
        // x-major: (x, y - ystep); y-major: (x - xstep, y)
        if (xmajor) stamp_empty(grid, g, a, bq - bstep);
        else stamp_empty(grid, g, bq - bstep, a);
      } else if (error + errorprev > ddmaj) {
        // x-major: (x - xstep, y); y-major: (x, y - ystep)
        if (xmajor) stamp_empty(grid, g, a - astep, bq);
        else stamp_empty(grid, g, bq, a - astep);
      } else {
        if (xmajor) {
          stamp_empty(grid, g, a - astep, bq);
          stamp_empty(grid, g, a, bq - bstep);
        } else {
          stamp_empty(grid, g, bq - bstep, a);
          stamp_empty(grid, g, bq, a - astep);
        }
      }
    }
    if (xmajor) stamp_empty(grid, g, a, bq);
    else stamp_empty(grid, g, bq, a);
  }
}

// pass 3: the end cell of every beam (fillGridAroundPoint with padding 0,
// local_mapper.cpp:148-151)
// The last workgroup to finish tells the host (sequence number into pinned
// memory, polled by kc_mapper_sync instead of a stream wait); every workgroup
// releases its stores (device scope) before it takes its ticket.
__global__ void endpoints_kernel(MapGeom g, const float *__restrict__ ranges,
                                 const double2 *__restrict__ trig, int n, int *__restrict__ grid,
                                 unsigned int *ticket, long long *host_seq, long long seq) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < n) {
    const int2 t = beam_endpoint(g, ranges[b], trig[b]);
    if (t.x >= 0 && t.x < g.H && t.y >= 0 && t.y < g.W)
      grid[(size_t)t.x + (size_t)t.y * (size_t)g.H] = KC_OCCUPIED;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      *ticket = 0u;
      *reinterpret_cast<volatile long long *>(host_seq) = seq;
    }
  }
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
  DevBuf<int> d_grid;
  DevBuf<float> d_ranges;
  DevBuf<double2> d_trig;
  DevBuf<unsigned int> d_ticket;
  PinBuf<long long> h_seq;   // written by the last endpoints workgroup
  long long seq = 0;         // scans launched
  bool direct = false;       // host stores reach device memory (large BAR)
  PinBuf<float> h_ranges;
  PinBuf<double2> h_trig;
  PinBuf<int> h_grid;
  // the angle table of a lidar does not change between scans: the trig table
  // is rebuilt only when the angles differ from the previous call
  std::vector<double> last_angles;
  bool trig_valid = false;
};

namespace {

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
             size_t n) {
  KC_HIP(hipSetDevice(m->device));
  hipStream_t s = m->stream;
  // the staging / device range buffers are free once the previous scan is done
  if (m->timing.enabled || !scan_done(m, 0)) KC_HIP(hipStreamSynchronize(s));
  m->timing.begin_cycle();
  const size_t cells = static_cast<size_t>(m->g.H) * m->g.W;
  KC_TRY(m->timing.start("grid_clear", s));
  KC_HIP(hipMemsetAsync(m->d_grid.p, 0xFF, cells * sizeof(int), s));  // -1
  KC_TRY(m->timing.stop(s));
  if (n == 0) {
    m->seq = 0;  // nothing will signal: kc_mapper_sync waits on the stream
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
  KC_TRY(m->timing.start("rays_kernel", s));
  hipLaunchKernelGGL(rays_kernel,
                     dim3((ni + kBeamsPerBlock - 1) / kBeamsPerBlock),
                     dim3(64 * kBeamsPerBlock), 0, s, m->g, m->d_ranges.p, m->d_trig.p, ni,
                     m->d_grid.p);
  KC_TRY(m->timing.stop(s));
  ++m->seq;
  KC_TRY(m->timing.start("endpoints_kernel", s));
  hipLaunchKernelGGL(endpoints_kernel, dim3((ni + 255) / 256), dim3(256), 0, s,
                     m->g, m->d_ranges.p, m->d_trig.p, ni, m->d_grid.p, m->d_ticket.p,
                     m->h_seq.p, m->seq);
  KC_TRY(m->timing.stop(s));
  KC_HIP(hipGetLastError());
  return KC_OK;
}

}  // namespace

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
  if ((rc = m->d_grid.reserve(cells)) || (rc = m->h_grid.reserve(cells)) ||
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
  if (const char *e = std::getenv("KC_TRIG_COPY"))
    if (e[0] == '1') m->direct = false;  // test hook: staged copies
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
  m->d_ranges.release();
  m->d_trig.release();
  m->d_ticket.release();
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
