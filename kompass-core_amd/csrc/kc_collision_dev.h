// Device-side argument blocks of the roll-out / collision kernels and the
// exact shape-vs-voxel tests (restated A4 contract).  Part of kc_dwa.hip.
#pragma once

namespace kc {

// ===========================================================================
// device-side parameter blocks (passed by value)
// ===========================================================================
struct CollDev {
  int shape;        // KC_CYLINDER / KC_BOX / KC_SPHERE
  int enabled;      // 0 => no occupied cell inside the reachable window
  int lds;          // stage the occupancy bits in LDS
  int kx0, ky0;     // window origin (voxel keys, octree frame)
  int W, H, wpr;    // window size in cells, 32-bit words per row
  uint32_t wpr_magic;  // ceil(2^32 / wpr) (wpr > 1): word index / wpr == __umulhi(index, wpr_magic) for every index < 2^24
  double r00, r01, r10, r11;  // octree-frame rotation (float values widened)
  double tx, ty;              // octree-frame origin in the world
  double res, inv;            // voxel edge, 1/res (octomap resolution_factor)
  double radius, rr;          // cylinder / sphere radius, radius^2
  double a, b;                // box half extents
  const uint32_t *bits;       // [H][wpr] occupancy bits (global)
  const double *ddz;          // sphere only: per-cell z gap [H][W] (split path: window built on the host)
  // sphere on the fused path: per cell of the sensor bitmap a code (0 none, k + 1 = k-th smallest
  // z gap of this sensor update) and the gaps themselves -- a handful of voxel layers lie within
  // a sphere's height, so a byte per cell replaces 8; read from global memory by the exact tests only
  const uint8_t *gz;          // [gH][gwpr * 32] or null
  const double *zlut;         // [<= 255]
  double zconst;              // ... or, when every accepted voxel lies in one layer (a planar scan), its gap
  int zmode;                  // 0: ddz (window table), 1: gz + zlut, 2: zconst
  // occupancy bits of ALL accepted voxel columns (built once per sensor
  // update); the fused kernel copies its window out of it, word aligned
  const uint32_t *gbits;      // [gH][gwpr]
  int gkx0, gky0, gH, gwpr;   // origin (keys), rows, words per row
  // the same bitmap dilated twice (built by dilate_kernel once per sensor
  // update): `ginner` marks the cells from which an occupied cell is SURELY
  // within the robot's inscribed radius, `gouter` the cells from which one is
  // POSSIBLY within its circumscribed radius.  A pose in a cell outside
  // gouter cannot collide, one inside ginner does; only the thin shell in
  // between needs the exact test.
  const uint32_t *ginner, *gouter;
  int dil;                    // masks present
  // Long boxes (one half extent at least twice the other): the masks are dilated for `cover` circles laid along the
  // long axis instead of one around the centre -- the outer one with the radius hypot(A / cover, B) of a circle that
  // covers its share of the box, the inner one with B as before -- and a pose looks all of them up: a 1.5 x 0.2 m
  // box has a shell of 0.66 m between its inscribed and its circumscribed circle, in which every pose took the exact
  // test (cycle kernel 44 us against 31 for a cylinder, tools/geometry_sweep.py); seven circles leave 5 cm.
  // Low byte: the number of circles (0 / 1: the single look-up), bit 8: the long axis is the box's y axis.
  int cover;
};

struct RollArgs {
  int n;            // samples in this launch (shard)
  int first;        // offset of the shard in the sample arrays
  int P;            // points per trajectory
  int A;            // trig-table row count
  int stage;        // LDS transposition of the outputs
  double x0, y0, dt;
  // the sample list: value tables of the axes + per sample (index into vxt) | (index into vyt) << 16
  // (hm::VelocityLattice: a new window rewrites the small tables only)
  const double *vxt, *vyt;
  int nvx, nvy;     // their lengths
  const uint32_t *vidx;
  const int32_t *row;
  const int32_t *perm;  // fused kernel: local sample ids ordered by omega row, so that the
                        // samples of a workgroup share as few trig rows as possible
  const int32_t *prow;      // trig rows in that order
  const uint32_t *pvi;      // value indices in that order
  const double2 *trig;  // [P][A] (cos, sin) of yaw_k per omega row
  float *px, *py;       // [n][P] sample-major
  double2 *pos;         // [P][n] step-major double poses (collision pass input)
  uint8_t *flags;       // [n] admissible
  int *adm_list;        // admissible local sample ids, appended (any order)
  long long *adm_count; // device counter (re-armed by the cost kernel)
  long long *dev_err;   // device error word of the cycle record (no kernel sets it at present; the epilogues carry it)
  // Device trig (round 3, kc_trig_exact.h): no host table at all -- the fused kernel forms yaw_k of its own
  // omega rows by repeated addition from yaw0 (path.h:30) and evaluates glibc's sincos algorithm itself; the
  // split path's kernels read a table trig_table_kernel has filled the same way.
  int trig_dev;             // 1: fused kernel computes its trig rows; 0: it reads `trig`
  double yaw0;
  const double *sincostab;  // [440] the table sincos reads, in the context's device memory
  double2 *trig_out;        // box footprints: the rows also go here (the exact tests read yaw_k of a pose back)
  unsigned long long *dbg;  // diagnostic build only (KC_DEBUG_STAMPS)
  CollDev c;
  // drop_samples_ == false (trajectory_sampler.cpp:157-168): a sample whose first collision comes at loop
  // step i with last_free_index = i - 1 > num_ctrl stays admissible -- path points i + 1 .. P - 1 repeat point
  // i - 1, velocities i .. P - 2 are zero.  Needs the FIRST colliding pose of a sample, not just any.
  int freeze;                     // 1: that mode
  int num_ctrl;                   // numCtrlPoints_ (:88)
  int *freeze_step;               // [n] by shard-local id: 0, or the first zero-velocity step i of a frozen sample
  float *frz_smooth, *frz_jerk;   // [n] the smoothness / jerk sums of the frozen profile (0 when not frozen)
  const double *omega_values;     // [A] omega of every trig row (the frozen profile's velocity step)
  float acc0, acc1, acc2;         // cost_evaluator.cpp:18-20
  int *first_hit;                 // split path only: [n] first colliding pose index (INT_MAX: none)
};

// smoothness / jerk sums of a profile that is (fvx, fvy, fom) up to step f - 1 and zero from step f
// (cost_evaluator.cpp:187-233: float += pow(delta, 2) / accLimit per axis, in index order; every other term of
// the two loops is +0.0).  smoothness: delta = 0 - v at index f; jerk: v[f] - 2 v[f-1] + v[f-2] = -v at index
// f and v[f+1] - 2 v[f] + v[f-1] = +v at index f + 1 (when f + 1 <= nv - 1).
__host__ __device__ inline void frozen_velocity_sums(float fvx, float fvy, float fom, int f, int nv, float acc0,
                                                     float acc1, float acc2, float *smooth, float *jerk) {
  auto sq = [](float v, float lim) { return (static_cast<double>(v) * static_cast<double>(v)) / static_cast<double>(lim); };
  const double tx = acc0 > 0 ? sq(fvx, acc0) : 0.0, ty = acc1 > 0 ? sq(fvy, acc1) : 0.0, to = acc2 > 0 ? sq(fom, acc2) : 0.0;
  float s = 0.0f;
  if (f >= 1 && f < nv) {
    s = static_cast<float>(static_cast<double>(s) + tx);
    s = static_cast<float>(static_cast<double>(s) + ty);
    s = static_cast<float>(static_cast<double>(s) + to);
  }
  float j = 0.0f;
  for (int q = 0; q < 2; ++q) {
    const int idx = f + q;
    if (idx >= 2 && idx < nv) {
      j = static_cast<float>(static_cast<double>(j) + tx);
      j = static_cast<float>(static_cast<double>(j) + ty);
      j = static_cast<float>(static_cast<double>(j) + to);
    }
  }
  *smooth = s;
  *jerk = j;
}

__device__ __forceinline__ double sample_vx(const RollArgs &a, int local) { return a.vxt[a.vidx[a.first + local] & 0xFFFFu]; }
__device__ __forceinline__ double sample_vy(const RollArgs &a, int local) { return a.vyt[a.vidx[a.first + local] >> 16]; }

// Kernel arguments of the large kernels (the cycle kernel: 1.3 KB = 21 cache lines) reach a wavefront through
// scalar loads from the kernarg segment, and the compiler reloads them lazily -- one s_load + s_waitcnt per
// struct member, in a row, each the first touch of its line (round 4 phase clocks: 3 us between two barriers of the
// cycle kernel's phase A with nothing in it but 25 such pairs).  kernarg_touch asks for EVERY line of the segment
// at once when the wavefront starts -- one scalar round trip; the lazy reloads then hit the scalar cache.
// Lines beyond the segment are clamped to its last line (the segment may end a page).
// The kernel then reads its arguments THROUGH the pointer the touch hands back (an opaque value to the compiler, so no
// argument load can be placed in front of the touch; constant address space, so the loads stay scalar and may be
// moved and merged like kernarg loads): `const auto &ka = *kernargs_touched<Args...>()`.
template <class A, class B>
struct KernargPair {  // the kernarg segment of a kernel (A, B): members in order, each at its own alignment
  A a;
  B b;
};
template <class A, class B, class C>
struct KernargTriple {
  A a;
  B b;
  C c;
};
template <class KA>
__device__ __forceinline__ const KA *kernargs_touched() {
  constexpr int kLines = (static_cast<int>(sizeof(KA)) + 63) / 64;
  static_assert(kLines >= 1 && kLines <= 24, "kernargs_touched: up to 1536 bytes");
  unsigned long long ka = reinterpret_cast<unsigned long long>(__builtin_amdgcn_kernarg_segment_ptr());
  // (every load names the same destination: the values are never used, and the wait sits in the same block,
  // so the register cannot be handed to anything else while a load is in flight)
  unsigned int t;
#define KC_KA_OFF(i) "i"(((i) < kLines ? (i) : kLines - 1) * 64)
#define KC_KA_L(n) "s_load_dword %0, %1, %" #n "\n\t"
  asm volatile(KC_KA_L(2) KC_KA_L(3) KC_KA_L(4) KC_KA_L(5) KC_KA_L(6) KC_KA_L(7) KC_KA_L(8) KC_KA_L(9) KC_KA_L(10)
               KC_KA_L(11) KC_KA_L(12) KC_KA_L(13) KC_KA_L(14) KC_KA_L(15) KC_KA_L(16) KC_KA_L(17) KC_KA_L(18)
               KC_KA_L(19) KC_KA_L(20) KC_KA_L(21) KC_KA_L(22) KC_KA_L(23) KC_KA_L(24) KC_KA_L(25)
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(t), "+s"(ka)
               : KC_KA_OFF(0), KC_KA_OFF(1), KC_KA_OFF(2), KC_KA_OFF(3), KC_KA_OFF(4), KC_KA_OFF(5), KC_KA_OFF(6),
                 KC_KA_OFF(7), KC_KA_OFF(8), KC_KA_OFF(9), KC_KA_OFF(10), KC_KA_OFF(11), KC_KA_OFF(12), KC_KA_OFF(13),
                 KC_KA_OFF(14), KC_KA_OFF(15), KC_KA_OFF(16), KC_KA_OFF(17), KC_KA_OFF(18), KC_KA_OFF(19), KC_KA_OFF(20),
                 KC_KA_OFF(21), KC_KA_OFF(22), KC_KA_OFF(23)
               : "memory");
#undef KC_KA_L
#undef KC_KA_OFF
  typedef const KA __attribute__((address_space(4))) *ConstPtr;
  return (const KA *)reinterpret_cast<ConstPtr>(ka);
}

// Phase clocks for kernel tuning: compiled in only with -DKC_PHASE_STAMPS (the
// product build carries none of it); KC_DEBUG_STAMPS=1 then dumps them when the
// context is destroyed.
#ifdef KC_PHASE_STAMPS
#define KC_RSTAMP(slot)                                                    \
  do {                                                                     \
    if (a.dbg && threadIdx.x == 0 && blockIdx.x < 512)                                    \
      a.dbg[(size_t)blockIdx.x * 32 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define KC_RSTAMP(slot) do { } while (0)
#endif

// ===========================================================================
// collision: analytic shape-vs-occupied-voxel test (restated A4 contract)
// ===========================================================================
template <typename BitsPtr>
__device__ __forceinline__ bool hit_round(const CollDev &c, BitsPtr bits,
                                          double x, double y, int row0 = 0, int row_step = 1) {
  const double dx = x - c.tx, dy = y - c.ty;
  const double xf = c.r00 * dx + c.r10 * dy;
  const double yf = c.r01 * dx + c.r11 * dy;
  const double r = c.radius;
  int cx0 = static_cast<int>(floor((xf - r) * c.inv)) - 1 - c.kx0;
  int cx1 = static_cast<int>(floor((xf + r) * c.inv)) + 1 - c.kx0;
  int cy0 = static_cast<int>(floor((yf - r) * c.inv)) - 1 - c.ky0;
  int cy1 = static_cast<int>(floor((yf + r) * c.inv)) + 1 - c.ky0;
  cx0 = max(cx0, 0);
  cy0 = max(cy0, 0);
  cx1 = min(cx1, c.W - 1);
  cy1 = min(cy1, c.H - 1);
  // (row0, row_step: the rows of the window can be shared out among several lanes; a hit is a hit
  // whoever finds it)
  for (int cy = cy0 + row0; cy <= cy1; cy += row_step) {
    const int ky = c.ky0 + cy;
    const double ylo = static_cast<double>(ky) * c.res;
    const double yhi = static_cast<double>(ky + 1) * c.res;
    double gy = 0.0;
    if (ylo - yf > gy) gy = ylo - yf;
    if (yf - yhi > gy) gy = yf - yhi;
    for (int wbase = cx0 & ~31; wbase <= cx1; wbase += 32) {
      uint32_t m = bits[cy * c.wpr + (wbase >> 5)];
      if (wbase < cx0) m &= 0xFFFFFFFFu << (cx0 - wbase);
      if (cx1 - wbase < 31) m &= 0xFFFFFFFFu >> (31 - (cx1 - wbase));
      while (m) {
        const int b = __ffs(static_cast<int>(m)) - 1;
        m &= m - 1;
        const int cx = wbase + b;
        const int kx = c.kx0 + cx;
        const double xlo = static_cast<double>(kx) * c.res;
        const double xhi = static_cast<double>(kx + 1) * c.res;
        double gx = 0.0;
        if (xlo - xf > gx) gx = xlo - xf;
        if (xf - xhi > gx) gx = xf - xhi;
        double zz = 0.0;
        if (c.shape == KC_SPHERE) {
          double g;
          if (c.zmode == 2) {
            g = c.zconst;
          } else if (c.zmode == 1) {
            const int code = c.gz[static_cast<size_t>(c.ky0 + cy - c.gky0) * (c.gwpr * 32) + (kx - c.gkx0)];
            g = c.zlut[code - 1];  // (a set bit has a code)
          } else {
            g = c.ddz[cy * c.W + cx];
          }
          zz = g * g;
        }
        const double d2 = gx * gx + gy * gy + zz;
        if (d2 <= c.rr) return true;
      }
    }
  }
  return false;
}

template <typename BitsPtr>
__device__ __forceinline__ bool hit_box(const CollDev &c, BitsPtr bits,
                                        double x, double y, double cw,
                                        double sw, int row0 = 0, int row_step = 1) {
  const double dx = x - c.tx, dy = y - c.ty;
  const double xf = c.r00 * dx + c.r10 * dy;
  const double yf = c.r01 * dx + c.r11 * dy;
  const double ux = c.r00 * cw + c.r10 * sw;
  const double uy = c.r01 * cw + c.r11 * sw;
  const double vx = -uy, vy = ux;
  const double ex = c.a * fabs(ux) + c.b * fabs(vx);
  const double ey = c.a * fabs(uy) + c.b * fabs(vy);
  int cx0 = static_cast<int>(floor((xf - ex) * c.inv)) - 1 - c.kx0;
  int cx1 = static_cast<int>(floor((xf + ex) * c.inv)) + 1 - c.kx0;
  int cy0 = static_cast<int>(floor((yf - ey) * c.inv)) - 1 - c.ky0;
  int cy1 = static_cast<int>(floor((yf + ey) * c.inv)) + 1 - c.ky0;
  cx0 = max(cx0, 0);
  cy0 = max(cy0, 0);
  cx1 = min(cx1, c.W - 1);
  cy1 = min(cy1, c.H - 1);
  const double h = c.res / 2.0;
  const double hu = h * (fabs(ux) + fabs(uy));
  const double hv = h * (fabs(vx) + fabs(vy));
  for (int cy = cy0 + row0; cy <= cy1; cy += row_step) {
    const int ky = c.ky0 + cy;
    const double qy = (static_cast<double>(ky) + 0.5) * c.res - yf;
    for (int wbase = cx0 & ~31; wbase <= cx1; wbase += 32) {
      uint32_t m = bits[cy * c.wpr + (wbase >> 5)];
      if (wbase < cx0) m &= 0xFFFFFFFFu << (cx0 - wbase);
      if (cx1 - wbase < 31) m &= 0xFFFFFFFFu >> (31 - (cx1 - wbase));
      while (m) {
        const int b = __ffs(static_cast<int>(m)) - 1;
        m &= m - 1;
        const int kx = c.kx0 + wbase + b;
        const double qx = (static_cast<double>(kx) + 0.5) * c.res - xf;
        if (fabs(qx) > h + ex) continue;
        if (fabs(qy) > h + ey) continue;
        if (fabs(qx * ux + qy * uy) > c.a + hu) continue;
        if (fabs(qx * vx + qy * vy) > c.b + hv) continue;
        return true;
      }
    }
  }
  return false;
}

}  // namespace kc
