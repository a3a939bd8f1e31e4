// Collision against a TILTED octree (A4 with a non-planar sensor mount, LaserScan input: the octree frame
// F = body_tf * sensor_tf_body of collision_check.h:99-117 is not upright, collision_check.cpp:61-68 accepts
// any quaternion).  The voxels of a laser scan form ONE layer kz = floor(hz / res) of cubes
// [k res, (k + 1) res]^3 in F; the robot shape stands upright in the world.  Exact closed-set tests in f64
// with a fixed operation order (the oracle restates them; FCL does these with GJK -- the reference holds no
// vector for a tilted mount: parity unpinned beyond this build's restatement):
//   sphere    distance from the centre (taken into F) to the cube
//   box       separating axes of two boxes: 3 cube axes, 3 box axes, 9 cross products
//   cylinder  in the robot's frame the cylinder is {|z| <= hh} x disc(r): clip the cube to the slab, project
//             onto xy (the hull of the kept vertices and the edge / plane crossings) and test that convex
//             polygon against the disc -- origin strictly inside, or some chord of the point set within r
// Part of kc_dwa.hip (split roll-out path only: the fused kernels keep their planar window machinery).
#pragma once

namespace kc {

struct TiltDev {
  double R[3][3];  // F: rotation (float values widened) ...
  double t[3];     // ... and origin in the world
  double res, inv, h;  // voxel edge, 1 / res, res / 2
  int kz;              // the scan's voxel layer
  int shape;
  double radius, hh;   // cylinder: radius, half height; sphere: radius
  double a, b, c;      // box half extents
  double rho;          // circumscribed radius of the shape about its centre
  const uint32_t *gbits;  // occupied (kx, ky) columns over their bounding box
  int gkx0, gky0, gH, gwpr;
};

__host__ __device__ inline double tilt_abs(double v) { return v < 0.0 ? -v : v; }

// squared distance from the origin to the segment A-B (2-D)
__host__ __device__ inline double tilt_seg_d2(double ax, double ay, double bx, double by) {
  const double dx = bx - ax, dy = by - ay;
  const double l2 = dx * dx + dy * dy;
  double t = 0.0;
  if (l2 > 0.0) {
    t = -(ax * dx + ay * dy) / l2;
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
  }
  const double qx = ax + t * dx, qy = ay + t * dy;
  return qx * qx + qy * qy;
}

// cube centre m (in F) against the shape at world (x, y, 0) with yaw (cw, sw); cf = the shape's centre in F
__host__ __device__ inline bool tilt_cube_hit(const TiltDev &c, const double m[3], const double cf[3], double x,
                                              double y, double cw, double sw) {
  const double h = c.h;
  if (c.shape == KC_SPHERE) {
    double d2 = 0.0;
    for (int i = 0; i < 3; ++i) {
      double g = tilt_abs(cf[i] - m[i]) - h;
      g = g > 0.0 ? g : 0.0;
      d2 = d2 + g * g;
    }
    return d2 <= c.radius * c.radius;
  }
  if (c.shape == KC_BOX) {
    // box axes in F: u = R^T (cw, sw, 0), v = R^T (-sw, cw, 0), w = R^T (0, 0, 1)
    double A[3][3];
    for (int i = 0; i < 3; ++i) {
      A[0][i] = c.R[0][i] * cw + c.R[1][i] * sw;
      A[1][i] = c.R[1][i] * cw - c.R[0][i] * sw;
      A[2][i] = c.R[2][i];
    }
    const double e[3] = {c.a, c.b, c.c};
    const double T[3] = {m[0] - cf[0], m[1] - cf[1], m[2] - cf[2]};
    // separated along L  <=>  |T . L| > h (|L0| + |L1| + |L2|) + sum_k e_k |A_k . L|
    auto separated = [&](double l0, double l1, double l2) {
      const double lhs = tilt_abs(T[0] * l0 + T[1] * l1 + T[2] * l2);
      double rhs = h * (tilt_abs(l0) + tilt_abs(l1) + tilt_abs(l2));
      for (int k = 0; k < 3; ++k) rhs = rhs + e[k] * tilt_abs(A[k][0] * l0 + A[k][1] * l1 + A[k][2] * l2);
      return lhs > rhs;
    };
    if (separated(1.0, 0.0, 0.0) || separated(0.0, 1.0, 0.0) || separated(0.0, 0.0, 1.0)) return false;
    for (int k = 0; k < 3; ++k)
      if (separated(A[k][0], A[k][1], A[k][2])) return false;
    for (int k = 0; k < 3; ++k) {
      // e_0 x A_k = (0, -A_k2, A_k1); e_1 x A_k = (A_k2, 0, -A_k0); e_2 x A_k = (-A_k1, A_k0, 0)
      if (separated(0.0, -A[k][2], A[k][1])) return false;
      if (separated(A[k][2], 0.0, -A[k][0])) return false;
      if (separated(-A[k][1], A[k][0], 0.0)) return false;
    }
    return true;
  }
  // cylinder: the cube in the robot's frame (world axes, origin at the shape's centre)
  double O[3], E[3][3];  // centre, half edges
  for (int i = 0; i < 3; ++i) {
    O[i] = (c.R[i][0] * m[0] + c.R[i][1] * m[1] + c.R[i][2] * m[2]) + c.t[i];
    for (int k = 0; k < 3; ++k) E[k][i] = c.R[i][k] * h;
  }
  O[0] = O[0] - x;
  O[1] = O[1] - y;
  double V[8][3];
  for (int s = 0; s < 8; ++s)
    for (int i = 0; i < 3; ++i)
      V[s][i] = O[i] + ((s & 1) ? E[0][i] : -E[0][i]) + ((s & 2) ? E[1][i] : -E[1][i]) + ((s & 4) ? E[2][i] : -E[2][i]);
  double px[32], py[32];
  int np = 0;
  const double hh = c.hh;
  for (int s = 0; s < 8; ++s)
    if (tilt_abs(V[s][2]) <= hh) {
      px[np] = V[s][0];
      py[np] = V[s][1];
      ++np;
    }
  for (int s = 0; s < 8; ++s)
    for (int bit = 1; bit < 8; bit <<= 1) {
      if (s & bit) continue;  // every edge once: from the vertex with the bit clear
      const double *P = V[s], *Q = V[s | bit];
      for (int side = 0; side < 2; ++side) {
        const double zp = side ? -hh : hh;
        const double da = P[2] - zp, db = Q[2] - zp;
        if ((da < 0.0 && db > 0.0) || (da > 0.0 && db < 0.0)) {
          const double tt = da / (da - db);
          px[np] = P[0] + tt * (Q[0] - P[0]);
          py[np] = P[1] + tt * (Q[1] - P[1]);
          ++np;
        }
      }
    }
  if (np == 0) return false;
  // origin strictly inside the hull: every point has another one clockwise of it
  bool inside = true;
  for (int i = 0; i < np && inside; ++i) {
    bool cw_found = false;
    for (int j = 0; j < np; ++j)
      if (px[i] * py[j] - py[i] * px[j] < 0.0) {
        cw_found = true;
        break;
      }
    inside = cw_found;
  }
  if (inside) return true;
  const double rr = c.radius * c.radius;
  for (int i = 0; i < np; ++i)
    for (int j = i; j < np; ++j)
      if (tilt_seg_d2(px[i], py[i], px[j], py[j]) <= rr) return true;
  return false;
}

// the shape at world pose (x, y, 0, yaw) against every occupied column within reach
template <typename BitsPtr>
__host__ __device__ inline bool tilt_hit(const TiltDev &c, BitsPtr gbits, double x, double y, double cw, double sw) {
  const double d[3] = {x - c.t[0], y - c.t[1], 0.0 - c.t[2]};
  double cf[3];
  for (int i = 0; i < 3; ++i) cf[i] = c.R[0][i] * d[0] + c.R[1][i] * d[1] + c.R[2][i] * d[2];
  // the layer's slab against the shape's bounding sphere
  const double zlo = static_cast<double>(c.kz) * c.res, zhi = static_cast<double>(c.kz + 1) * c.res;
  if (zlo - cf[2] > c.rho || cf[2] - zhi > c.rho) return false;
  const int kx0 = static_cast<int>(floor((cf[0] - c.rho) * c.inv)) - 1, kx1 = static_cast<int>(floor((cf[0] + c.rho) * c.inv)) + 1;
  const int ky0 = static_cast<int>(floor((cf[1] - c.rho) * c.inv)) - 1, ky1 = static_cast<int>(floor((cf[1] + c.rho) * c.inv)) + 1;
  for (int ky = ky0; ky <= ky1; ++ky) {
    const int gy = ky - c.gky0;
    if (gy < 0 || gy >= c.gH) continue;
    for (int kx = kx0; kx <= kx1; ++kx) {
      const int gx = kx - c.gkx0;
      if (gx < 0 || gx >= c.gwpr * 32) continue;
      if (!((gbits[static_cast<size_t>(gy) * c.gwpr + (gx >> 5)] >> (gx & 31)) & 1u)) continue;
      const double m[3] = {(static_cast<double>(kx) + 0.5) * c.res, (static_cast<double>(ky) + 0.5) * c.res,
                           (static_cast<double>(c.kz) + 0.5) * c.res};
      if (tilt_cube_hit(c, m, cf, x, y, cw, sw)) return true;
    }
  }
  return false;
}

struct TiltArgs {
  TiltDev c;
  const double2 *pos;   // [P][n] step-major double poses of the roll-out
  const double2 *trig;  // [P][A]
  const int32_t *row;
  int n, first, P, A;
  uint8_t *flags;
  int *first_hit;       // drop_samples = false, else null
};

// split path, tilted octree: one lane per pose (trajectory_sampler.cpp:147-152: any colliding pose drops the sample)
#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ __launch_bounds__(256) void collision_tilted_kernel(TiltArgs a) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)a.n * (a.P - 1)) return;
  const int k = static_cast<int>(t / a.n) + 1;
  const int n = static_cast<int>(t - (long)(k - 1) * a.n);
  const double2 p = a.pos[(size_t)k * a.n + n];
  double cw = 1.0, sw = 0.0;
  if (a.c.shape == KC_BOX) {
    const double2 cs = a.trig[(size_t)k * a.A + a.row[a.first + n]];  // yaw_k
    cw = cs.x;
    sw = cs.y;
  }
  if (tilt_hit(a.c, a.c.gbits, p.x, p.y, cw, sw)) {
    a.flags[n] = 0;
    if (a.first_hit) atomicMin(&a.first_hit[n], k);
  }
}
#endif  // KC_TU_CYCLE

// batch pose check (CollisionChecker::checkCollisions) against a tilted octree
#ifdef KC_TU_CYCLE  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
__global__ void pose_check_tilted_kernel(TiltDev c, const double2 *__restrict__ pos, const double2 *__restrict__ cs,
                                         int n, uint8_t *__restrict__ hit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  hit[i] = tilt_hit(c, c.gbits, pos[i].x, pos[i].y, cs[i].x, cs[i].y) ? 1 : 0;
}
#endif  // KC_TU_CYCLE

}  // namespace kc
