/*
 * kompass_oracle.c -- CPU restatement of the kompass_cpp sampling-controller
 * hot path + LocalMapper.  TEST INFRASTRUCTURE ONLY (see kompass_oracle.h).
 *
 * Every function cites the reference file:line it restates; paths are relative
 * to <reference>/src/kompass_cpp/kompass_cpp/.  Float/double mixing follows
 * the C++ source expression by expression (usual arithmetic conversions,
 * `float += double*float` rounded once, unqualified cos()/pow()/sqrt() on a
 * float argument resolving to the double overloads, ...).  Compiled with
 * -ffp-contract=off and no -march, so no FMA is ever formed -- same as an
 * x86-64 build of the reference.
 */
#define _GNU_SOURCE
#include "kompass_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define KO_MIN_VEL 0.01 /* utils/trajectory_sampler.h:13-15 */
#define KO_DEFAULT_MIN_DIST FLT_MAX /* datatypes/trajectory.h:12 */

/* ------------------------------------------------------------------------ */
/* float helpers restating Eigen fixed-size reductions                      */
/* redux_novec_unroller<.,.,0,3>: func(e0, func(e1, e2))                    */
/* ------------------------------------------------------------------------ */
static inline float sum3f(float a, float b, float c) {
  float t = b + c;
  return a + t;
}
/* (p1 - p2).squaredNorm() on Vector3f -- path.h:209-211 */
static inline float dist_sq3f(float ax, float ay, float az, float bx, float by,
                              float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  return sum3f(xx, yy, zz);
}
/* (p1 - p2).norm() -- path.h:204-206 */
static inline float dist3f(float ax, float ay, float az, float bx, float by,
                           float bz) {
  return sqrtf(dist_sq3f(ax, ay, az, bx, by, bz));
}

/* ------------------------------------------------------------------------ */
/* Eigen float isometry restatement (utils/transformation.h:9-41)           */
/* ------------------------------------------------------------------------ */
typedef struct {
  float R[3][3];
  float t[3];
} iso3f;

typedef struct {
  float w, x, y, z;
} quatf;

/* Eigen QuaternionBase::toRotationMatrix */
static void quat_to_rot(quatf q, float R[3][3]) {
  const float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0][0] = 1.0f - (tyy + tzz);
  R[0][1] = txy - twz;
  R[0][2] = txz + twy;
  R[1][0] = txy + twz;
  R[1][1] = 1.0f - (txx + tzz);
  R[1][2] = tyz - twx;
  R[2][0] = txz - twy;
  R[2][1] = tyz + twx;
  R[2][2] = 1.0f - (txx + tyy);
}

/* Eigen quaternionbase_assign_impl<Other,3,3>::run (matrix -> quaternion) */
static quatf rot_to_quat(const float R[3][3]) {
  quatf q;
  float t = sum3f(R[0][0], R[1][1], R[2][2]);
  if (t > 0.0f) {
    t = sqrtf(t + 1.0f);
    q.w = 0.5f * t;
    t = 0.5f / t;
    q.x = (R[2][1] - R[1][2]) * t;
    q.y = (R[0][2] - R[2][0]) * t;
    q.z = (R[1][0] - R[0][1]) * t;
  } else {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    float v[3];
    t = sqrtf(R[i][i] - R[j][j] - R[k][k] + 1.0f);
    v[i] = 0.5f * t;
    t = 0.5f / t;
    q.w = (R[k][j] - R[j][k]) * t;
    v[j] = (R[j][i] + R[i][j]) * t;
    v[k] = (R[k][i] + R[i][k]) * t;
    q.x = v[0];
    q.y = v[1];
    q.z = v[2];
  }
  return q;
}

/* getTransformation(Quaternionf, Vector3f): Identity.translate(t).rotate(q) */
static iso3f iso_from_quat(quatf q, const float t[3]) {
  iso3f T;
  quat_to_rot(q, T.R);
  T.t[0] = t[0];
  T.t[1] = t[1];
  T.t[2] = t[2];
  return T;
}
/* getTransformation(Matrix3f, Vector3f): rotate(Quaternionf(matrix)) */
static iso3f iso_from_rot(const float R[3][3], const float t[3]) {
  return iso_from_quat(rot_to_quat(R), t);
}
/* eulerToRotationMatrix(0, 0, yaw): (rotZ*rotY*rotX).matrix() with the two
 * zero-angle factors being exact identity quaternions */
static void euler_yaw_to_rot(float yaw, float R[3][3]) {
  quatf q;
  float ha = 0.5f * yaw;
  q.w = cosf(ha);
  q.x = 0.0f;
  q.y = 0.0f;
  q.z = sinf(ha);
  quat_to_rot(q, R);
}
/* getTransformation(const Path::State) -- transformation.h:35-41 */
static iso3f iso_from_state(const ko_state *s) {
  float R[3][3];
  euler_yaw_to_rot((float)s->yaw, R);
  float t[3] = {(float)s->x, (float)s->y, 0.0f};
  return iso_from_rot(R, t);
}
/* Transform * Transform (Eigen transform_transform_product_impl) */
static iso3f iso_mul(const iso3f *A, const iso3f *B) {
  iso3f C;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      C.R[i][j] = sum3f(A->R[i][0] * B->R[0][j], A->R[i][1] * B->R[1][j],
                        A->R[i][2] * B->R[2][j]);
    C.t[i] = sum3f(A->R[i][0] * B->t[0], A->R[i][1] * B->t[1],
                   A->R[i][2] * B->t[2]) +
             A->t[i];
  }
  return C;
}
/* Transform * Vector3f */
static void iso_apply(const iso3f *T, float px, float py, float pz,
                      float out[3]) {
  for (int i = 0; i < 3; ++i)
    out[i] = T->t[i] + sum3f(T->R[i][0] * px, T->R[i][1] * py, T->R[i][2] * pz);
}

/* ======================================================================== */
/* Path  (datatypes/path.h, src/datatypes/path.cpp)                         */
/* ======================================================================== */
struct ko_path {
  float *X, *Y, *Z, *K;
  size_t cap;          /* allocated */
  size_t size;         /* current_size_ */
  float total_length;  /* current_total_length_ */
  float *acc;          /* accumulated_path_length_ */
  size_t acc_size;
  size_t *seg;         /* segment_indices_ */
  size_t nseg;
  int interpolated;
};

static void path_alloc(ko_path *p, size_t n) {
  p->X = (float *)calloc(n ? n : 1, sizeof(float));
  p->Y = (float *)calloc(n ? n : 1, sizeof(float));
  p->Z = (float *)calloc(n ? n : 1, sizeof(float));
  p->K = (float *)calloc(n ? n : 1, sizeof(float));
  p->cap = n;
}

ko_path *ko_path_new(const float *x, const float *y, const float *z, size_t n) {
  if (n < 2) return NULL; /* path.cpp:14-17 */
  ko_path *p = (ko_path *)calloc(1, sizeof(ko_path));
  path_alloc(p, n);
  for (size_t i = 0; i < n; ++i) {
    p->X[i] = x[i];
    p->Y[i] = y[i];
    p->Z[i] = z ? z[i] : 0.0f;
    p->K[i] = 0.0f;
  }
  p->size = n;
  return p;
}

ko_path *ko_path_clone(const ko_path *s) {
  ko_path *p = (ko_path *)calloc(1, sizeof(ko_path));
  path_alloc(p, s->cap);
  memcpy(p->X, s->X, s->cap * sizeof(float));
  memcpy(p->Y, s->Y, s->cap * sizeof(float));
  memcpy(p->Z, s->Z, s->cap * sizeof(float));
  memcpy(p->K, s->K, s->cap * sizeof(float));
  p->size = s->size;
  p->total_length = s->total_length;
  p->interpolated = s->interpolated;
  p->acc_size = s->acc_size;
  p->acc = (float *)malloc((s->acc_size ? s->acc_size : 1) * sizeof(float));
  if (s->acc_size) memcpy(p->acc, s->acc, s->acc_size * sizeof(float));
  p->nseg = s->nseg;
  p->seg = (size_t *)malloc((s->nseg ? s->nseg : 1) * sizeof(size_t));
  if (s->nseg) memcpy(p->seg, s->seg, s->nseg * sizeof(size_t));
  return p;
}

void ko_path_free(ko_path *p) {
  if (!p) return;
  free(p->X);
  free(p->Y);
  free(p->Z);
  free(p->K);
  free(p->acc);
  free(p->seg);
  free(p);
}

size_t ko_path_size(const ko_path *p) { return p->size; }
const float *ko_path_x(const ko_path *p) { return p->X; }
const float *ko_path_y(const ko_path *p) { return p->Y; }
const float *ko_path_z(const ko_path *p) { return p->Z; }
const float *ko_path_curvature(const ko_path *p) { return p->K; }
const float *ko_path_acc(const ko_path *p) { return p->acc; }
size_t ko_path_acc_size(const ko_path *p) { return p->acc_size; }
size_t ko_path_num_segments(const ko_path *p) { return p->nseg; }

/* path.cpp:148-165 */
float ko_path_total_length(const ko_path *p) {
  if (p->size < 2) return 0.0f;
  if (p->interpolated) return p->total_length;
  float total = 0.0f;
  for (size_t i = 1; i < p->size; ++i)
    total += dist3f(p->X[i - 1], p->Y[i - 1], p->Z[i - 1], p->X[i], p->Y[i],
                    p->Z[i]);
  return total;
}

/* tk::spline linear mode: spline.h:197-225 (set_points) */
typedef struct {
  double *x, *y, *b;
  int n;
} lin_spline;

static void lin_spline_set(lin_spline *s, const double *x, const double *y,
                           int n) {
  s->n = n;
  s->x = (double *)malloc(sizeof(double) * n);
  s->y = (double *)malloc(sizeof(double) * n);
  s->b = (double *)malloc(sizeof(double) * n);
  memcpy(s->x, x, sizeof(double) * n);
  memcpy(s->y, y, sizeof(double) * n);
  for (int i = 0; i < n - 1; ++i)
    s->b[i] = (s->y[i + 1] - s->y[i]) / (s->x[i + 1] - s->x[i]);
  s->b[n - 1] = s->b[n - 2];
}
static void lin_spline_free(lin_spline *s) {
  free(s->x);
  free(s->y);
  free(s->b);
}
/* spline.h:390-422 (find_closest + operator()) with m_c = m_d = 0 */
static double lin_spline_eval(const lin_spline *s, double x) {
  /* std::upper_bound: first element > x */
  int lo = 0, hi = s->n;
  while (lo < hi) {
    int mid = lo + (hi - lo) / 2;
    if (s->x[mid] > x)
      hi = mid;
    else
      lo = mid + 1;
  }
  int idx = lo - 1;
  if (idx < 0) idx = 0;
  double h = x - s->x[idx];
  double interpol;
  if (x < s->x[0]) {
    /* left extrapolation; m_c0 is never initialised in linear mode and this
     * branch is unreachable for s >= 0 */
    interpol = (0.0 * h + s->b[0]) * h + s->y[0];
  } else if (x > s->x[s->n - 1]) {
    interpol = (0.0 * h + s->b[s->n - 1]) * h + s->y[s->n - 1];
  } else {
    interpol = ((0.0 * h + 0.0) * h + s->b[idx]) * h + s->y[idx];
  }
  return interpol;
}

/* path.cpp:167-288 */
int ko_path_interpolate_linear(ko_path *p, double max_dist) {
  if (p->size < 2) return -1;
  size_t n = p->size;
  double *s_vals = (double *)malloc(sizeof(double) * n);
  double *x_vals = (double *)malloc(sizeof(double) * n);
  double *y_vals = (double *)malloc(sizeof(double) * n);
  s_vals[0] = 0.0;
  x_vals[0] = p->X[0];
  y_vals[0] = p->Y[0];
  p->total_length = 0.0f;
  for (size_t i = 1; i < n; ++i) {
    /* std::hypot(float, float) -> float overload */
    double seg_dist = hypotf(p->X[i] - p->X[i - 1], p->Y[i] - p->Y[i - 1]);
    p->total_length = (float)((double)p->total_length + seg_dist);
    s_vals[i] = p->total_length;
    x_vals[i] = p->X[i];
    y_vals[i] = p->Y[i];
  }
  lin_spline sx, sy;
  lin_spline_set(&sx, s_vals, x_vals, (int)n);
  lin_spline_set(&sy, s_vals, y_vals, (int)n);

  size_t new_size = (size_t)((double)p->total_length / max_dist) + 1;
  free(p->X);
  free(p->Y);
  free(p->Z);
  free(p->K);
  path_alloc(p, new_size); /* zero-filled: Z_, Curvature_ setZero */
  /* std::vector<float>::resize keeps old entries, value-inits new ones */
  float *nacc = (float *)calloc(new_size ? new_size : 1, sizeof(float));
  if (p->acc) {
    size_t keep = p->acc_size < new_size ? p->acc_size : new_size;
    memcpy(nacc, p->acc, keep * sizeof(float));
    free(p->acc);
  }
  p->acc = nacc;
  p->acc_size = new_size;

  size_t idx = 0;
  const double total = (double)p->total_length;
  for (double s = 0.0; s <= total && idx < new_size; s += max_dist) {
    p->acc[idx] = (float)s;
    p->X[idx] = (float)lin_spline_eval(&sx, s);
    p->Y[idx] = (float)lin_spline_eval(&sy, s);
    idx++;
  }
  if (idx < new_size && idx > 0) { /* path.cpp:249-254 (no acc entry: Q5) */
    p->X[idx] = (float)lin_spline_eval(&sx, total);
    p->Y[idx] = (float)lin_spline_eval(&sy, total);
    idx++;
  }
  p->interpolated = 1;
  p->size = idx;

  /* curvature, path.cpp:260-287 */
  if (p->size >= 2) {
    float dx_old = p->X[1] - p->X[0];
    float dy_old = p->Y[1] - p->Y[0];
    for (size_t i = 1; i + 1 < p->size; ++i) {
      float dx = p->X[i + 1] - p->X[i];
      float dy = p->Y[i + 1] - p->Y[i];
      float ddx = dx - dx_old;
      float ddy = dy - dy_old;
      float val = dx * dx + dy * dy;
      float denominator = val * sqrtf(val);
      if (denominator > 1e-6f)
        p->K[i] = (dx_old * ddy - ddx * dy_old) / denominator;
      else
        p->K[i] = 0.0f;
      dx_old = dx;
      dy_old = dy;
    }
  }
  lin_spline_free(&sx);
  lin_spline_free(&sy);
  free(s_vals);
  free(x_vals);
  free(y_vals);
  return 0;
}

/* path.cpp:290-330 */
void ko_path_segment(ko_path *p, double seg_len, size_t max_pts) {
  if (p->size < 2) return;
  free(p->seg);
  p->seg = (size_t *)malloc(sizeof(size_t) * p->size);
  p->nseg = 0;
  p->seg[p->nseg++] = 0;
  if (!p->interpolated) { /* Q6: per-edge lengths, not prefix sums */
    free(p->acc);
    p->acc_size = p->size - 1;
    p->acc = (float *)calloc(p->acc_size ? p->acc_size : 1, sizeof(float));
    for (size_t i = 0; i + 1 < p->size; ++i)
      p->acc[i] = dist3f(p->X[i], p->Y[i], p->Z[i], p->X[i + 1], p->Y[i + 1],
                         p->Z[i + 1]);
  }
  size_t start_idx = 0;
  float start_len = p->acc[0];
  for (size_t i = 1; i < p->size; ++i) {
    const size_t pts = i - start_idx + 1;
    /* reading past acc_size is UB in the reference for the non-interpolated
     * case; clamp (parity domain excludes it, Q6) */
    const float acc_i = (i < p->acc_size) ? p->acc[i] : 0.0f;
    const float len = acc_i - start_len;
    const int length_exceeded = (seg_len > 0.0 && (double)len >= seg_len);
    const int points_exceeded = (max_pts > 0 && pts > max_pts);
    if (length_exceeded || points_exceeded) {
      p->seg[p->nseg++] = i;
      start_idx = i;
      start_len = acc_i;
    }
  }
}

size_t ko_path_segment_start(const ko_path *p, size_t s) { return p->seg[s]; }
size_t ko_path_segment_end(const ko_path *p, size_t s) {
  if (s + 1 < p->nseg) return p->seg[s + 1] - 1;
  return p->size - 1;
}

/* ======================================================================== */
/* A1: dynamic window + lattice                                             */
/* ======================================================================== */
static int make_odd(int n) { return (n % 2 == 0) ? n + 1 : n; }
static int imax(int a, int b) { return a > b ? a : b; }

/* trajectory.h:19-29 */
void ko_linear_sample_split(int ctr_type, int max_lin, int *vx_n, int *vy_n) {
  if (ctr_type == KO_OMNI) {
    *vx_n = make_odd(imax(3, max_lin * 3 / 4));
    *vy_n = make_odd(imax(3, max_lin * 1 / 4));
  } else {
    *vx_n = make_odd(imax(3, max_lin));
    *vy_n = 1;
  }
}
/* trajectory.h:32-45 (called with the already-bumped angular count,
 * trajectory_sampler.cpp:48,56-57) */
size_t ko_num_trajectories(int ctr_type, int max_lin, int max_ang) {
  const int ang_slots = max_ang + 1 - (max_ang % 2);
  int vx_n, vy_n;
  ko_linear_sample_split(ctr_type, max_lin, &vx_n, &vy_n);
  if (ctr_type == KO_OMNI)
    return (size_t)vx_n * (size_t)ang_slots + (size_t)vx_n * (size_t)vy_n;
  return (size_t)vx_n * (size_t)ang_slots;
}
/* trajectory.h:48-51 */
size_t ko_num_points_per_trajectory(double time_step, double horizon) {
  return (size_t)(horizon / time_step);
}

static inline double dmin(double a, double b) { return b < a ? b : a; }
static inline double dmax(double a, double b) { return a < b ? b : a; }

typedef struct {
  double *vx, *vy, *om;
  size_t cap, n;
  int overflow;
} vel_sink;

/* trajectory_sampler.cpp:122-125: all-zero sample is never rolled out */
static void sink_push(vel_sink *s, double vx, double vy, double om) {
  if (fabs(vx) < KO_MIN_VEL && fabs(vy) < KO_MIN_VEL && fabs(om) < KO_MIN_VEL)
    return;
  if (s->n >= s->cap) {
    s->overflow = 1;
    return;
  }
  s->vx[s->n] = vx;
  s->vy[s->n] = vy;
  s->om[s->n] = om;
  s->n++;
}

long ko_sample_velocities(int ctr_type, const ko_limits *L, double cvx,
                          double cvy, double com, double dt, int max_lin,
                          int max_ang, double *vx, double *vy, double *om,
                          size_t cap) {
  int lin_x, lin_y;
  ko_linear_sample_split(ctr_type, max_lin, &lin_x, &lin_y);
  const int ang_n = max_ang + 1 - (max_ang % 2); /* trajectory_sampler.cpp:48 */
  /* trajectory_sampler.cpp:51-54 */
  double vy_max = L->vy_max, vy_acc = L->vy_acc, vy_dec = L->vy_dec;
  if (ctr_type != KO_OMNI) vy_max = vy_acc = vy_dec = 0.0;

  /* trajectory_sampler.cpp:328-372 */
  double max_vx = dmin(L->vx_max, cvx + L->vx_acc * dt);
  double min_vx = dmax(-L->vx_max, cvx - L->vx_dec * dt);
  double max_vy, min_vy;
  if (ctr_type == KO_OMNI) {
    max_vy = dmin(vy_max, cvy + vy_acc * dt);
    min_vy = dmax(-vy_max, cvy - vy_dec * dt);
  } else {
    max_vy = 0.0;
    min_vy = 0.0;
  }
  double res_x = dmax((max_vx - min_vx) / (lin_x - 1), 0.001);
  double res_y =
      (lin_y > 1) ? dmax((max_vy - min_vy) / (lin_y - 1), 0.001) : 0.001;
  double max_om = dmin(L->omega_max, com + L->omega_acc * dt);
  double min_om = dmax(-L->omega_max, com - L->omega_dec * dt);
  double res_om = dmax((max_om - min_om) / (ang_n - 1), 0.001);

  vel_sink s = {vx, vy, om, cap, 0, 0};
  if (ctr_type == KO_OMNI) {
    /* trajectory_sampler.cpp:256-272 */
    for (double v = min_vx; v <= max_vx; v += res_x) {
      for (double w = min_vy; w <= max_vy; w += res_y) sink_push(&s, v, w, 0.0);
      if (fabs(v) >= KO_MIN_VEL)
        for (double o = min_om; o <= max_om; o += res_om)
          sink_push(&s, v, 0.0, o);
    }
  } else {
    /* trajectory_sampler.cpp:207-217 */
    for (double v = min_vx; v <= max_vx; v += res_x)
      if (fabs(v) >= KO_MIN_VEL)
        for (double o = min_om; o <= max_om; o += res_om)
          sink_push(&s, v, 0.0, o);
  }
  if (s.overflow) return -1;
  return (long)s.n;
}

/* ======================================================================== */
/* A4: collision checker -- analytic restatement of FCL octree-vs-shape      */
/* ======================================================================== */
/*
 * Semantics ("Restatement contract for A4"):
 *  - occupied voxel key per axis = floor(coord * (1/res))  (octomap
 *    coordToKey), voxel = closed cube [k res, (k+1) res]^3 in the octree
 *    frame F = sensor_tf_world_ captured at updateSensorData time
 *    (collision_check.h:101,121-125); free-space ray cells never collide;
 *  - robot shape (Cylinder(r,h) / Box(x,y,z) / Sphere(r), centred at z = 0,
 *    collision_check.cpp:38-58) placed at (x, y, 0, yaw);
 *  - collision <=> some occupied cube intersects the shape, closed sets
 *    (touching counts: tests/collisions_test.cpp:43-61).
 * All tests below are evaluated in double with a fixed operation order; the
 * HIP kernel repeats the same operations so both agree bit for bit.
 */
typedef struct {
  int64_t key; /* (kx << 32) | (uint32)ky ; INT64_MIN = empty */
  double ddz;  /* sphere: min z-gap to the sphere centre plane; else 0 */
} cell_slot;

struct ko_coll {
  int shape;
  float dims[3];
  double radius; /* robotRadius_ */
  double height; /* robotHeight_ */
  double res;
  iso3f sensor_tf_body;
  iso3f body_tf;        /* at the state last set */
  iso3f F;              /* sensor_tf_world_ captured at update time */
  double sx, sy, syaw;  /* state last set (double, for ko_coll_check) */
  /* a LaserScan seen through a mount that is not a rotation about z: the octree frame F is tilted
   * against the upright robot shape (collision_check.cpp:61-68 takes any quaternion); the scan's voxels
   * are one layer kz of F, kept without a z gate and tested in 3-D (tilted_cube_hit) */
  int tilted;
  int32_t tilt_kz;
  /* hash set of occupied (kx,ky) columns that can touch the robot in z */
  cell_slot *tab;
  size_t tab_cap; /* power of two */
  size_t n_cells;
};

#define KO_EMPTY_KEY INT64_MIN

static void coll_clear(ko_coll *c) {
  for (size_t i = 0; i < c->tab_cap; ++i) c->tab[i].key = KO_EMPTY_KEY;
  c->n_cells = 0;
}
static void coll_reserve(ko_coll *c, size_t n) {
  size_t cap = 64;
  while (cap < 2 * n + 8) cap <<= 1;
  if (cap != c->tab_cap) {
    free(c->tab);
    c->tab = (cell_slot *)malloc(cap * sizeof(cell_slot));
    c->tab_cap = cap;
  }
  coll_clear(c);
}
static inline uint64_t hash64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
static inline int64_t pack_key(int32_t kx, int32_t ky) {
  return (int64_t)(((uint64_t)(uint32_t)kx << 32) | (uint64_t)(uint32_t)ky);
}
static void coll_insert(ko_coll *c, int32_t kx, int32_t ky, double ddz) {
  int64_t key = pack_key(kx, ky);
  size_t m = c->tab_cap - 1;
  size_t i = (size_t)hash64((uint64_t)key) & m;
  while (c->tab[i].key != KO_EMPTY_KEY) {
    if (c->tab[i].key == key) {
      if (ddz < c->tab[i].ddz) c->tab[i].ddz = ddz;
      return;
    }
    i = (i + 1) & m;
  }
  c->tab[i].key = key;
  c->tab[i].ddz = ddz;
  c->n_cells++;
}
static inline const cell_slot *coll_find(const ko_coll *c, int32_t kx,
                                         int32_t ky) {
  int64_t key = pack_key(kx, ky);
  size_t m = c->tab_cap - 1;
  size_t i = (size_t)hash64((uint64_t)key) & m;
  while (c->tab[i].key != KO_EMPTY_KEY) {
    if (c->tab[i].key == key) return &c->tab[i];
    i = (i + 1) & m;
  }
  return NULL;
}

ko_coll *ko_coll_new(int shape, const float *dims, int ndims,
                     const float spos[3], const float srot[4], double res) {
  ko_coll *c = (ko_coll *)calloc(1, sizeof(ko_coll));
  c->shape = shape;
  for (int i = 0; i < 3; ++i) c->dims[i] = (i < ndims) ? dims[i] : 0.0f;
  /* collision_check.cpp:38-58 */
  if (shape == KO_CYLINDER) {
    c->height = c->dims[1];
    c->radius = c->dims[0];
  } else if (shape == KO_BOX) {
    c->height = c->dims[2];
    c->radius = sqrt(pow(c->dims[0], 2) + pow(c->dims[1], 2)) / 2;
  } else if (shape == KO_SPHERE) {
    c->radius = c->dims[0];
    c->height = 2 * c->dims[0];
  } else {
    free(c);
    return NULL;
  }
  c->res = res;
  quatf q = {srot[3], srot[0], srot[1], srot[2]};
  c->sensor_tf_body = iso_from_quat(q, spos);
  c->F = c->sensor_tf_body; /* collision_check.cpp:67 */
  ko_state s0 = {0, 0, 0, 0};
  c->body_tf = iso_from_state(&s0);
  /* Body::tf starts as Identity (collision_check.h:33) */
  memset(&c->body_tf, 0, sizeof(iso3f));
  c->body_tf.R[0][0] = c->body_tf.R[1][1] = c->body_tf.R[2][2] = 1.0f;
  coll_reserve(c, 16);
  return c;
}
void ko_coll_free(ko_coll *c) {
  if (!c) return;
  free(c->tab);
  free(c);
}
void ko_coll_set_resolution(ko_coll *c, double res) {
  if (res != c->res) c->res = res;
}
float ko_coll_radius(const ko_coll *c) { return (float)c->radius; }
size_t ko_coll_num_voxels(const ko_coll *c) { return c->n_cells; }

/* collision_check.cpp:125-147 */
void ko_coll_update_state(ko_coll *c, double x, double y, double yaw) {
  float R[3][3];
  euler_yaw_to_rot((float)yaw, R); /* eulerToRotationMatrix(0.0, 0.0, yaw) */
  float t[3] = {(float)x, (float)y, 0.0f};
  c->body_tf = iso_from_rot(R, t);
  c->sx = x;
  c->sy = y;
  c->syaw = yaw;
}

static int frame_is_planar(const iso3f *F) {
  const float eps = 1e-6f;
  return fabsf(F->R[0][2]) < eps && fabsf(F->R[1][2]) < eps &&
         fabsf(F->R[2][0]) < eps && fabsf(F->R[2][1]) < eps && F->R[2][2] > 0.f;
}

/* insert the voxel of one octree-frame point */
static void coll_add_point(ko_coll *c, float px, float py, float pz) {
  const double inv = 1.0 / c->res; /* octomap resolution_factor */
  const double fx = floor(inv * (double)px);
  const double fy = floor(inv * (double)py);
  const double fz = floor(inv * (double)pz);
  if (!(fabs(fx) < 32768.0 && fabs(fy) < 32768.0 && fabs(fz) < 32768.0))
    return; /* outside the 16-level octree: coordToKeyChecked fails */
  const int32_t kx = (int32_t)fx, ky = (int32_t)fy, kz = (int32_t)fz;
  if (c->tilted) {
    c->tilt_kz = kz;
    coll_insert(c, kx, ky, 0.0);
    return;
  }
  /* z extent of the voxel in F and of the robot (centre z_w = 0) in F */
  const double zlo = (double)kz * c->res, zhi = (double)(kz + 1) * c->res;
  const double zc = -(double)c->F.t[2];
  if (c->shape == KO_SPHERE) {
    double ddz = 0.0;
    if (zlo - zc > ddz) ddz = zlo - zc;
    if (zc - zhi > ddz) ddz = zc - zhi;
    if (ddz > c->radius) return;
    coll_insert(c, kx, ky, ddz);
  } else {
    const double hz = c->height / 2.0;
    if (zlo <= zc + hz && zhi >= zc - hz) coll_insert(c, kx, ky, 0.0);
  }
}

/* collision_check.h:99-117,134 */
int ko_coll_update_scan(ko_coll *c, const double *ranges, const double *angles,
                        size_t n) {
  c->F = iso_mul(&c->body_tf, &c->sensor_tf_body);
  c->tilted = !frame_is_planar(&c->F);
  coll_reserve(c, n);
  /* float height_in_sensor = -sensor_tf_body_.translation().z() / 2.0; */
  float height_in_sensor = (float)(-(double)c->sensor_tf_body.t[2] / 2.0);
  for (size_t i = 0; i < n; ++i) {
    double angle = angles[i], r = ranges[i];
    if (isfinite(r)) {
      float x = (float)(r * cos(angle));
      float y = (float)(r * sin(angle));
      coll_add_point(c, x, y, height_in_sensor);
    }
  }
  return 0;
}
/* collision_check.h:119-131,134 */
int ko_coll_update_points(ko_coll *c, const float *xyz, size_t n,
                          int global_frame) {
  c->tilted = 0;
  if (global_frame) {
    memset(&c->F, 0, sizeof(iso3f));
    c->F.R[0][0] = c->F.R[1][1] = c->F.R[2][2] = 1.0f;
  } else {
    c->F = iso_mul(&c->body_tf, &c->sensor_tf_body);
  }
  if (!frame_is_planar(&c->F)) return -2; /* (several voxel layers in a tilted frame: not restated) */
  coll_reserve(c, n);
  for (size_t i = 0; i < n; ++i)
    coll_add_point(c, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  return 0;
}

/* ---- tilted octree frame: the shape (upright in the world, centre (x, y, 0)) against ONE voxel cube of F.
 * Closed sets, f64, fixed operation order (kc_tilt_dev.h of the build repeats it).
 *   sphere:   distance from the centre, taken into F, to the cube
 *   box:      separating axes of two boxes (3 + 3 + 9)
 *   cylinder: in the robot's frame it is {|z| <= hh} x disc(r): the cube is clipped to the slab and projected
 *             onto xy -- the hull of the kept vertices and of the edge / plane crossings -- and that convex
 *             polygon meets the disc when the origin lies strictly inside it or some chord of the point set
 *             comes within r of the origin */
static double seg_d2_origin(double ax, double ay, double bx, double by) {
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
static int sat_separated(const double T[3], double A[3][3], const double e[3], double h, double l0, double l1,
                         double l2) {
  const double lhs = fabs(T[0] * l0 + T[1] * l1 + T[2] * l2);
  double rhs = h * (fabs(l0) + fabs(l1) + fabs(l2));
  for (int k = 0; k < 3; ++k) rhs = rhs + e[k] * fabs(A[k][0] * l0 + A[k][1] * l1 + A[k][2] * l2);
  return lhs > rhs;
}
static int tilted_cube_hit(const ko_coll *c, const double m[3], const double cf[3], double x, double y,
                           double cw, double sw) {
  const double h = c->res / 2.0;
  double R[3][3], t[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R[i][j] = (double)c->F.R[i][j];
    t[i] = (double)c->F.t[i];
  }
  if (c->shape == KO_SPHERE) {
    double d2 = 0.0;
    for (int i = 0; i < 3; ++i) {
      double g = fabs(cf[i] - m[i]) - h;
      g = g > 0.0 ? g : 0.0;
      d2 = d2 + g * g;
    }
    return d2 <= c->radius * c->radius;
  }
  if (c->shape == KO_BOX) {
    double A[3][3];
    for (int i = 0; i < 3; ++i) {
      A[0][i] = R[0][i] * cw + R[1][i] * sw;
      A[1][i] = R[1][i] * cw - R[0][i] * sw;
      A[2][i] = R[2][i];
    }
    const double e[3] = {(double)c->dims[0] / 2.0, (double)c->dims[1] / 2.0, (double)c->dims[2] / 2.0};
    const double T[3] = {m[0] - cf[0], m[1] - cf[1], m[2] - cf[2]};
    if (sat_separated(T, A, e, h, 1.0, 0.0, 0.0) || sat_separated(T, A, e, h, 0.0, 1.0, 0.0) ||
        sat_separated(T, A, e, h, 0.0, 0.0, 1.0))
      return 0;
    for (int k = 0; k < 3; ++k)
      if (sat_separated(T, A, e, h, A[k][0], A[k][1], A[k][2])) return 0;
    for (int k = 0; k < 3; ++k) {
      if (sat_separated(T, A, e, h, 0.0, -A[k][2], A[k][1])) return 0;
      if (sat_separated(T, A, e, h, A[k][2], 0.0, -A[k][0])) return 0;
      if (sat_separated(T, A, e, h, -A[k][1], A[k][0], 0.0)) return 0;
    }
    return 1;
  }
  /* cylinder */
  double O[3], E[3][3];
  for (int i = 0; i < 3; ++i) {
    O[i] = (R[i][0] * m[0] + R[i][1] * m[1] + R[i][2] * m[2]) + t[i];
    for (int k = 0; k < 3; ++k) E[k][i] = R[i][k] * h;
  }
  O[0] = O[0] - x;
  O[1] = O[1] - y;
  double V[8][3];
  for (int s = 0; s < 8; ++s)
    for (int i = 0; i < 3; ++i)
      V[s][i] = O[i] + ((s & 1) ? E[0][i] : -E[0][i]) + ((s & 2) ? E[1][i] : -E[1][i]) +
                ((s & 4) ? E[2][i] : -E[2][i]);
  double px[32], py[32];
  int np = 0;
  const double hh = c->height / 2.0;
  for (int s = 0; s < 8; ++s)
    if (fabs(V[s][2]) <= hh) {
      px[np] = V[s][0];
      py[np] = V[s][1];
      ++np;
    }
  for (int s = 0; s < 8; ++s)
    for (int bit = 1; bit < 8; bit <<= 1) {
      if (s & bit) continue;
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
  if (np == 0) return 0;
  int inside = 1;
  for (int i = 0; i < np && inside; ++i) {
    int found = 0;
    for (int j = 0; j < np; ++j)
      if (px[i] * py[j] - py[i] * px[j] < 0.0) {
        found = 1;
        break;
      }
    inside = found;
  }
  if (inside) return 1;
  const double rr = c->radius * c->radius;
  for (int i = 0; i < np; ++i)
    for (int j = i; j < np; ++j)
      if (seg_d2_origin(px[i], py[i], px[j], py[j]) <= rr) return 1;
  return 0;
}
static int tilted_check_at(ko_coll *c, double x, double y, double yaw) {
  const double res = c->res, inv = 1.0 / c->res;
  const double d[3] = {x - (double)c->F.t[0], y - (double)c->F.t[1], 0.0 - (double)c->F.t[2]};
  double cf[3];
  for (int i = 0; i < 3; ++i)
    cf[i] = (double)c->F.R[0][i] * d[0] + (double)c->F.R[1][i] * d[1] + (double)c->F.R[2][i] * d[2];
  double rho;
  if (c->shape == KO_SPHERE) rho = c->radius;
  else if (c->shape == KO_BOX) {
    const double a = (double)c->dims[0] / 2.0, b = (double)c->dims[1] / 2.0, cc = (double)c->dims[2] / 2.0;
    rho = sqrt(a * a + b * b + cc * cc);
  } else {
    const double hh = c->height / 2.0;
    rho = sqrt(c->radius * c->radius + hh * hh);
  }
  const double zlo = (double)c->tilt_kz * res, zhi = (double)(c->tilt_kz + 1) * res;
  if (zlo - cf[2] > rho || cf[2] - zhi > rho) return 0;
  const int32_t kx0 = (int32_t)floor((cf[0] - rho) * inv) - 1, kx1 = (int32_t)floor((cf[0] + rho) * inv) + 1;
  const int32_t ky0 = (int32_t)floor((cf[1] - rho) * inv) - 1, ky1 = (int32_t)floor((cf[1] + rho) * inv) + 1;
  const double cw = cos(yaw), sw = sin(yaw);
  for (int32_t ky = ky0; ky <= ky1; ++ky)
    for (int32_t kx = kx0; kx <= kx1; ++kx) {
      if (!coll_find(c, kx, ky)) continue;
      const double m[3] = {((double)kx + 0.5) * res, ((double)ky + 0.5) * res, ((double)c->tilt_kz + 0.5) * res};
      if (tilted_cube_hit(c, m, cf, x, y, cw, sw)) return 1;
    }
  return 0;
}

/* shape-vs-occupied-columns test at world pose (x, y, yaw) */
int ko_coll_check_at(ko_coll *c, double x, double y, double yaw) {
  if (c->n_cells == 0) return 0;
  if (c->tilted) return tilted_check_at(c, x, y, yaw);
  const double res = c->res, inv = 1.0 / c->res;
  /* pose in F: p_F = R_F^T (p_w - t_F) */
  const double r00 = c->F.R[0][0], r01 = c->F.R[0][1];
  const double r10 = c->F.R[1][0], r11 = c->F.R[1][1];
  const double dx = x - (double)c->F.t[0];
  const double dy = y - (double)c->F.t[1];
  const double xf = r00 * dx + r10 * dy;
  const double yf = r01 * dx + r11 * dy;

  if (c->shape == KO_CYLINDER || c->shape == KO_SPHERE) {
    const double r = c->radius;
    const int32_t kx0 = (int32_t)floor((xf - r) * inv) - 1;
    const int32_t kx1 = (int32_t)floor((xf + r) * inv) + 1;
    const int32_t ky0 = (int32_t)floor((yf - r) * inv) - 1;
    const int32_t ky1 = (int32_t)floor((yf + r) * inv) + 1;
    const double rr = r * r;
    for (int32_t ky = ky0; ky <= ky1; ++ky)
      for (int32_t kx = kx0; kx <= kx1; ++kx) {
        const cell_slot *s = coll_find(c, kx, ky);
        if (!s) continue;
        double gx = 0.0, gy = 0.0;
        const double xlo = (double)kx * res, xhi = (double)(kx + 1) * res;
        const double ylo = (double)ky * res, yhi = (double)(ky + 1) * res;
        if (xlo - xf > gx) gx = xlo - xf;
        if (xf - xhi > gx) gx = xf - xhi;
        if (ylo - yf > gy) gy = ylo - yf;
        if (yf - yhi > gy) gy = yf - yhi;
        const double d2 = gx * gx + gy * gy + s->ddz * s->ddz;
        if (d2 <= rr) return 1;
      }
    return 0;
  }
  /* BOX: oriented rectangle (half extents a, b) vs axis-aligned squares */
  const double a = (double)c->dims[0] / 2.0, b = (double)c->dims[1] / 2.0;
  const double cw = cos(yaw), sw = sin(yaw);
  const double ux = r00 * cw + r10 * sw; /* box x-axis in F */
  const double uy = r01 * cw + r11 * sw;
  const double vx = -uy, vy = ux; /* box y-axis in F */
  const double ex = a * fabs(ux) + b * fabs(vx); /* AABB half extents */
  const double ey = a * fabs(uy) + b * fabs(vy);
  const int32_t kx0 = (int32_t)floor((xf - ex) * inv) - 1;
  const int32_t kx1 = (int32_t)floor((xf + ex) * inv) + 1;
  const int32_t ky0 = (int32_t)floor((yf - ey) * inv) - 1;
  const int32_t ky1 = (int32_t)floor((yf + ey) * inv) + 1;
  const double h = res / 2.0;
  const double hu = h * (fabs(ux) + fabs(uy));
  const double hv = h * (fabs(vx) + fabs(vy));
  for (int32_t ky = ky0; ky <= ky1; ++ky)
    for (int32_t kx = kx0; kx <= kx1; ++kx) {
      if (!coll_find(c, kx, ky)) continue;
      const double cx = ((double)kx + 0.5) * res - xf;
      const double cy = ((double)ky + 0.5) * res - yf;
      if (fabs(cx) > h + ex) continue;
      if (fabs(cy) > h + ey) continue;
      if (fabs(cx * ux + cy * uy) > a + hu) continue;
      if (fabs(cx * vx + cy * vy) > b + hv) continue;
      return 1;
    }
  return 0;
}
int ko_coll_check(ko_coll *c) {
  return ko_coll_check_at(c, c->sx, c->sy, c->syaw);
}

/* ======================================================================== */
/* A2/A3: roll-out (trajectory_sampler.cpp:118-179, path.h:24-30)           */
/* ======================================================================== */
static int rollout_one(ko_coll *coll, const ko_state *start, double dt_d,
                       size_t P, double vx, double vy, double om, float *px,
                       float *py) {
  /* State::update(const Velocity2D&, const float timeStep): dt narrowed */
  const double dt = (double)(float)dt_d;
  double x = start->x, y = start->y, yaw = start->yaw;
  px[0] = (float)x;
  py[0] = (float)y;
  for (size_t i = 0; i + 1 < P; ++i) {
    /* path.h:24-30 writes cos(yaw), sin(yaw); the reference's gcc release build folds the pair into ONE
       sincos call (as gcc -O2 does here) -- called explicitly so that the oracle does not depend on its own
       optimisation level: glibc's sin() / cos() take FMA variants on an FMA CPU and differ from sincos in
       the last bit of 0.14 % of arguments */
    double c, s;
    sincos(yaw, &s, &c);
    x += (vx * c - vy * s) * dt;
    y += (vx * s + vy * c) * dt;
    yaw += om * dt;
    if (coll && ko_coll_check_at(coll, x, y, yaw)) return 0;
    px[i + 1] = (float)x;
    py[i + 1] = (float)y;
  }
  return 1;
}

/* TrajectorySampler::getAdmissibleTrajsFromVel, trajectory_sampler.cpp:118-179, with both values of
 * drop_samples_: the loop breaks at the first colliding step i and remembers last_free_index = i - 1
 * (when i > 0; else it stays P - 1, :130,147-152); with drop_samples_ == false a sample whose
 * last_free_index lies beyond numCtrlPoints_ (= control_horizon / time_step as size_t, :88) is kept:
 * velocities j = last_free_index + 1 .. P - 2 become zero and path points j + 1 repeat the point at
 * last_free_index (:157-168) -- i.e. points 0 .. i are the rolled-out ones, points i + 1 .. P - 1 repeat
 * point i - 1.  fvx / fvy / fom: the sample's velocity profile [P - 1] (TrajectoryVelocities2D::add:
 * float = double, trajectory.h:96-103).  Returns 1 when the sample is admissible. */
static int rollout_one_mode(ko_coll *coll, const ko_state *start, double dt_d, size_t P, double vx,
                            double vy, double om, int drop, size_t num_ctrl, float *px, float *py,
                            float *fvx, float *fvy, float *fom) {
  const double dt = (double)(float)dt_d;
  double x = start->x, y = start->y, yaw = start->yaw;
  px[0] = (float)x;
  py[0] = (float)y;
  int is_collision = 0;
  size_t last_free = P - 1;
  for (size_t i = 0; i + 1 < P; ++i) {
    /* path.h:24-30 writes cos(yaw), sin(yaw); the reference's gcc release build folds the pair into ONE
       sincos call (as gcc -O2 does here) -- called explicitly so that the oracle does not depend on its own
       optimisation level: glibc's sin() / cos() take FMA variants on an FMA CPU and differ from sincos in
       the last bit of 0.14 % of arguments */
    double c, s;
    sincos(yaw, &s, &c);
    x += (vx * c - vy * s) * dt;
    y += (vx * s + vy * c) * dt;
    yaw += om * dt;
    if (coll && ko_coll_check_at(coll, x, y, yaw)) {
      is_collision = 1;
      if (i > 0) last_free = i - 1;
      break;
    }
    fvx[i] = (float)vx;
    fvy[i] = (float)vy;
    fom[i] = (float)om;
    px[i + 1] = (float)x;
    py[i + 1] = (float)y;
  }
  if (!drop && is_collision && last_free > num_ctrl && last_free < P - 1) {
    const float lx = px[last_free], ly = py[last_free];
    for (size_t j = last_free + 1; j < P - 1; ++j) {
      fvx[j] = 0.0f;
      fvy[j] = 0.0f;
      fom[j] = 0.0f;
      px[j + 1] = lx;
      py[j + 1] = ly;
    }
    is_collision = 0;
  }
  return !is_collision;
}

long ko_rollout_mode(ko_coll *coll, const ko_state *start, double time_step, size_t P,
                     const double *vx, const double *vy, const double *omega, size_t n,
                     int drop_samples, size_t num_ctrl_points, float *paths_x, float *paths_y,
                     float *vel_vx, float *vel_vy, float *vel_omega, int32_t *raw_index) {
  long na = 0;
  const size_t nv = P - 1;
  float *tx = (float *)malloc(sizeof(float) * (P ? P : 1) * 5);
  float *ty = tx + P, *tvx = ty + P, *tvy = tvx + P, *tom = tvy + P;
  for (size_t k = 0; k < n; ++k) {
    if (!rollout_one_mode(coll, start, time_step, P, vx[k], vy[k], omega[k], drop_samples,
                          num_ctrl_points, tx, ty, tvx, tvy, tom))
      continue;
    memcpy(paths_x + (size_t)na * P, tx, sizeof(float) * P);
    memcpy(paths_y + (size_t)na * P, ty, sizeof(float) * P);
    if (vel_vx) {
      memcpy(vel_vx + (size_t)na * nv, tvx, sizeof(float) * nv);
      memcpy(vel_vy + (size_t)na * nv, tvy, sizeof(float) * nv);
      memcpy(vel_omega + (size_t)na * nv, tom, sizeof(float) * nv);
    }
    if (raw_index) raw_index[na] = (int32_t)k;
    na++;
  }
  free(tx);
  return na;
}

long ko_rollout(ko_coll *coll, const ko_state *start, double time_step,
                size_t P, const double *vx, const double *vy,
                const double *omega, size_t n, float *paths_x, float *paths_y,
                float *vel_vx, float *vel_vy, float *vel_omega,
                int32_t *raw_index) {
  long na = 0;
  float *tx = (float *)malloc(sizeof(float) * (P ? P : 1));
  float *ty = (float *)malloc(sizeof(float) * (P ? P : 1));
  for (size_t k = 0; k < n; ++k) {
    if (!rollout_one(coll, start, time_step, P, vx[k], vy[k], omega[k], tx, ty))
      continue;
    memcpy(paths_x + (size_t)na * P, tx, sizeof(float) * P);
    memcpy(paths_y + (size_t)na * P, ty, sizeof(float) * P);
    if (vel_vx)
      for (size_t i = 0; i + 1 < P; ++i) {
        vel_vx[(size_t)na * (P - 1) + i] = (float)vx[k];
        vel_vy[(size_t)na * (P - 1) + i] = (float)vy[k];
        vel_omega[(size_t)na * (P - 1) + i] = (float)omega[k];
      }
    if (raw_index) raw_index[na] = (int32_t)k;
    na++;
  }
  free(tx);
  free(ty);
  return na;
}

/* ======================================================================== */
/* A5-A10: cost evaluator (src/utils/cost_evaluator.cpp)                     */
/* ======================================================================== */
/* path.h:85-91 */
float ko_segment_length(const float *x, const float *y, const float *z,
                        size_t n) {
  float length = 0.0f;
  for (size_t i = 0; i + 1 < n; ++i)
    length += dist3f(x[i], y[i], z[i], x[i + 1], y[i + 1], z[i + 1]);
  return length;
}

/* cost_evaluator.cpp:111-141 */
float ko_path_cost(const ko_cost_ctx *cx, const float *px, const float *py,
                   size_t P) {
  const float seg_len =
      ko_segment_length(cx->seg_x, cx->seg_y, cx->seg_z, cx->seg_size);
  float total_cost = 0.0f;
  for (size_t i = 0; i < P; ++i) {
    float min_dist = KO_DEFAULT_MIN_DIST;
    for (size_t j = 0; j < cx->seg_size; ++j) {
      float d = dist3f(cx->seg_x[j], cx->seg_y[j], cx->seg_z[j], px[i], py[i],
                       0.0f);
      if (d < min_dist) min_dist = d;
    }
    total_cost += min_dist;
  }
  const size_t e = cx->seg_size - 1;
  float end_dist_error = dist3f(px[P - 1], py[P - 1], 0.0f, cx->seg_x[e],
                                cx->seg_y[e], cx->seg_z[e]) /
                         seg_len;
  return (total_cost / (float)(long)P + end_dist_error) / 2;
}

/* cost_evaluator.cpp:150-177 */
float ko_goal_cost(const ko_cost_ctx *cx, const float *px, const float *py,
                   size_t P) {
  const float ex = px[P - 1], ey = py[P - 1];
  float min_dist_sq = KO_DEFAULT_MIN_DIST;
  size_t closest = 0;
  for (size_t i = 0; i < cx->seg_size; ++i) {
    const float d_sq =
        dist_sq3f(ex, ey, 0.0f, cx->seg_x[i], cx->seg_y[i], cx->seg_z[i]);
    if (d_sq < min_dist_sq) {
      min_dist_sq = d_sq;
      closest = i;
    }
  }
  const size_t abs_idx = closest + cx->seg_start_idx;
  /* Path::getDistanceAtIndex, path.h:190-194 */
  const float at = (abs_idx >= cx->path_acc_size) ? 0.0f : cx->path_acc[abs_idx];
  const float L = cx->ref_path_length;
  const float arc_remaining_normalized = (L - at) / L;
  return arc_remaining_normalized + (sqrtf(min_dist_sq) / L);
}

/* cost_evaluator.cpp:179-184 + trajectory.h:218-235 */
float ko_obstacle_cost(const ko_cost_ctx *cx, const float *px, const float *py,
                       size_t P) {
  float dist_min;
  if (cx->n_obs == 0) {
    dist_min = 0.0f;
  } else {
    float minDist = KO_DEFAULT_MIN_DIST;
    for (size_t i = 0; i < cx->n_obs; ++i)
      for (size_t j = 0; j < P; ++j) {
        /* pow(float, int) -> double; float = double + double */
        float dist = (float)(pow((double)(cx->obs_x[i] - px[j]), 2) +
                             pow((double)(cx->obs_y[i] - py[j]), 2));
        if (dist < minDist) minDist = dist;
      }
    dist_min = (float)sqrt((double)minDist);
  }
  const float D = cx->max_obstacles_dist;
  float v = D - dist_min;
  if (v < 0.0f) v = 0.0f; /* std::max(D - dist, 0.0f) */
  return v / D;
}

/* cost_evaluator.cpp:187-206 */
float ko_smoothness_cost(const ko_cost_ctx *cx, const float *vx,
                         const float *vy, const float *om, size_t nv) {
  float cost = 0.0f;
  for (size_t i = 1; i < nv; ++i) {
    if (cx->acc_limits[0] > 0) {
      float d = vx[i] - vx[i - 1];
      cost = (float)((double)cost + pow((double)d, 2) / cx->acc_limits[0]);
    }
    if (cx->acc_limits[1] > 0) {
      float d = vy[i] - vy[i - 1];
      cost = (float)((double)cost + pow((double)d, 2) / cx->acc_limits[1]);
    }
    if (cx->acc_limits[2] > 0) {
      float d = om[i] - om[i - 1];
      cost = (float)((double)cost + pow((double)d, 2) / cx->acc_limits[2]);
    }
  }
  return cost / (float)(3 * (long)nv);
}

/* cost_evaluator.cpp:209-233 */
float ko_jerk_cost(const ko_cost_ctx *cx, const float *vx, const float *vy,
                   const float *om, size_t nv) {
  float cost = 0.0f;
  for (size_t i = 2; i < nv; ++i) {
    if (cx->acc_limits[0] > 0) {
      float j = vx[i] - 2 * vx[i - 1] + vx[i - 2];
      cost = (float)((double)cost + pow((double)j, 2) / cx->acc_limits[0]);
    }
    if (cx->acc_limits[1] > 0) {
      float j = vy[i] - 2 * vy[i - 1] + vy[i - 2];
      cost = (float)((double)cost + pow((double)j, 2) / cx->acc_limits[1]);
    }
    if (cx->acc_limits[2] > 0) {
      float j = om[i] - 2 * om[i - 1] + om[i - 2];
      cost = (float)((double)cost + pow((double)j, 2) / cx->acc_limits[2]);
    }
  }
  return cost / (float)(3 * (long)nv);
}

static float total_cost_one(const ko_cost_ctx *cx, const float *px,
                            const float *py, const float *vx, const float *vy,
                            const float *om, size_t P) {
  /* cost_evaluator.cpp:59-100: total_cost is float, every += is a double
   * multiply-add rounded once to float */
  float total = 0.0f;
  double w;
  if (cx->ref_path_length > 0.0f) {
    if ((w = cx->w.goal_distance_weight) > 0.0) {
      float c = ko_goal_cost(cx, px, py, P);
      total = (float)((double)total + w * (double)c);
    }
    if ((w = cx->w.reference_path_distance_weight) > 0.0) {
      float c = ko_path_cost(cx, px, py, P);
      total = (float)((double)total + w * (double)c);
    }
  }
  if (cx->n_obs > 0 && (w = cx->w.obstacles_distance_weight) > 0.0) {
    float c = ko_obstacle_cost(cx, px, py, P);
    total = (float)((double)total + w * (double)c);
  }
  if ((w = cx->w.smoothness_weight) > 0.0) {
    float c = vx ? ko_smoothness_cost(cx, vx, vy, om, P - 1) : 0.0f;
    total = (float)((double)total + w * (double)c);
  }
  if ((w = cx->w.jerk_weight) > 0.0) {
    float c = vx ? ko_jerk_cost(cx, vx, vy, om, P - 1) : 0.0f;
    total = (float)((double)total + w * (double)c);
  }
  return total;
}

/* cost_evaluator.cpp:49-109 */
long ko_min_trajectory_cost(const ko_cost_ctx *cx, const float *paths_x,
                            const float *paths_y, const float *vel_vx,
                            const float *vel_vy, const float *vel_omega,
                            size_t N, size_t P, size_t sp, size_t sv,
                            float *costs_out, float *min_cost_out) {
  float min_cost = KO_DEFAULT_MIN_DIST;
  long best = -1;
  for (size_t n = 0; n < N; ++n) {
    const float *vx = vel_vx ? vel_vx + n * sv : NULL;
    const float *vy = vel_vx ? vel_vy + n * sv : NULL;
    const float *om = vel_vx ? vel_omega + n * sv : NULL;
    float total =
        total_cost_one(cx, paths_x + n * sp, paths_y + n * sp, vx, vy, om, P);
    if (costs_out) costs_out[n] = total;
    if (total < min_cost) { /* strict: lowest index wins ties */
      min_cost = total;
      best = (long)n;
    }
  }
  if (min_cost_out) *min_cost_out = min_cost;
  return best;
}

/* cost_evaluator.h:174-193 */
void ko_obstacles_from_scan(const float spos[3], const float srot[4],
                            const ko_state *state, const double *ranges,
                            const double *angles, size_t n, float *ox,
                            float *oy) {
  quatf q = {srot[3], srot[0], srot[1], srot[2]};
  iso3f sensor_tf_body = iso_from_quat(q, spos);
  iso3f body_tf_world = iso_from_state(state);
  iso3f T = iso_mul(&sensor_tf_body, &body_tf_world);
  for (size_t i = 0; i < n; ++i) {
    double point_x = ranges[i] * cos(angles[i]);
    double point_y = ranges[i] * sin(angles[i]);
    float out[3];
    iso_apply(&T, (float)point_x, (float)point_y, 0.0f, out);
    ox[i] = out[0];
    oy[i] = out[1];
  }
}
/* cost_evaluator.h:207-223 */
void ko_obstacles_from_points(const float spos[3], const float srot[4],
                              const ko_state *state, const float *xyz,
                              size_t n, float *ox, float *oy) {
  quatf q = {srot[3], srot[0], srot[1], srot[2]};
  iso3f sensor_tf_body = iso_from_quat(q, spos);
  iso3f body_tf_world = iso_from_state(state);
  iso3f T = iso_mul(&sensor_tf_body, &body_tf_world);
  for (size_t i = 0; i < n; ++i) {
    float out[3];
    iso_apply(&T, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], out);
    ox[i] = out[0];
    oy[i] = out[1];
  }
}

/* ======================================================================== */
/* A11: controller glue (controllers/{controller,follower,dwa})              */
/* ======================================================================== */
typedef struct {
  size_t index, segment_index;
  double segment_length, parallel_distance, normal_distance;
  ko_state state;
} path_position; /* path.h:301-308 */

struct ko_dwa {
  ko_dwa_config cfg;
  /* Controller base members DWA never populates (Q1): defaults of
   * control.h:192-217 and ControlType() == ACKERMANN */
  ko_limits ctr_limits_base;
  int rotate_in_place;
  /* FollowerParameters defaults, follower.h:18-64 (Q2) */
  double goal_dist_tolerance, goal_orientation_tolerance, loosing_goal_distance;
  double curvature_horizon_tolerance, path_segment_length;
  double max_point_interpolation_distance, lookahead_distance;
  size_t max_segment_size;
  ko_path *path;
  path_position closest;
  int path_processing;
  size_t current_segment_index, max_segment_index;
  double goal_distance;
  int reached_goal;
  double heading_error;
  ko_state state;
  /* sampler */
  ko_coll *coll;
  double base_max_time, max_time;
  size_t num_trajectories, P;
  double max_forward_distance;
  float max_local_range;
  /* buffers */
  size_t buf_cap_n, buf_cap_p;
  double *svx, *svy, *som;
  float *px, *py, *costs;
  int32_t *raw;
  float *best_x, *best_y, *best_v[3];
  float *ox, *oy;
  size_t ocap;
  long n_adm;
};

static void dwa_alloc(ko_dwa *d) {
  size_t n = d->num_trajectories + 8, p = d->P + 2;
  d->buf_cap_n = n;
  d->buf_cap_p = p;
  d->svx = (double *)malloc(sizeof(double) * n);
  d->svy = (double *)malloc(sizeof(double) * n);
  d->som = (double *)malloc(sizeof(double) * n);
  d->px = (float *)malloc(sizeof(float) * n * p);
  d->py = (float *)malloc(sizeof(float) * n * p);
  d->costs = (float *)malloc(sizeof(float) * n);
  d->raw = (int32_t *)malloc(sizeof(int32_t) * n);
  d->best_x = (float *)calloc(p, sizeof(float));
  d->best_y = (float *)calloc(p, sizeof(float));
  for (int i = 0; i < 3; ++i) d->best_v[i] = (float *)calloc(p, sizeof(float));
}

ko_dwa *ko_dwa_new(const ko_dwa_config *cfg) {
  ko_dwa *d = (ko_dwa *)calloc(1, sizeof(ko_dwa));
  d->cfg = *cfg;
  /* control.h:192-217 defaults */
  d->ctr_limits_base.vx_max = 1.0;
  d->ctr_limits_base.vx_acc = 10.0;
  d->ctr_limits_base.vx_dec = 10.0;
  d->ctr_limits_base.vy_max = 1.0;
  d->ctr_limits_base.vy_acc = 10.0;
  d->ctr_limits_base.vy_dec = 10.0;
  d->ctr_limits_base.omega_max_angle = M_PI;
  d->ctr_limits_base.omega_max = 1.0;
  d->ctr_limits_base.omega_acc = 10.0;
  d->ctr_limits_base.omega_dec = 10.0;
  d->rotate_in_place = 0; /* follower.cpp:42-46 with ctrType == ACKERMANN */
  d->goal_dist_tolerance = 0.1;
  d->goal_orientation_tolerance = 0.1;
  d->loosing_goal_distance = 0.5;
  d->curvature_horizon_tolerance = 1.5;
  d->path_segment_length = 1.0;
  d->max_point_interpolation_distance = 0.01;
  d->lookahead_distance = 1.0;
  /* follower.cpp:54-59 */
  d->max_segment_size =
      (size_t)(d->path_segment_length / d->max_point_interpolation_distance + 1);
  d->closest.segment_length = -1.0;
  d->goal_distance = DBL_MAX;
  d->max_local_range = 10.0f; /* dwa.h:236 */
  /* sampler ctor, trajectory_sampler.cpp:23-60 */
  d->coll = ko_coll_new(cfg->shape, cfg->dims, cfg->ndims, cfg->sensor_pos,
                        cfg->sensor_rot_xyzw, cfg->octree_res);
  d->base_max_time = d->max_time = cfg->prediction_horizon;
  const int ang = cfg->max_angular_samples + 1 - (cfg->max_angular_samples % 2);
  d->P = ko_num_points_per_trajectory(cfg->time_step, cfg->prediction_horizon);
  d->num_trajectories =
      ko_num_trajectories(cfg->ctr_type, cfg->max_linear_samples, ang);
  /* dwa.cpp:31-37 */
  if (cfg->ctr_type == KO_OMNI)
    d->max_forward_distance =
        dmax(cfg->limits.vx_max, cfg->limits.vy_max) * cfg->prediction_horizon;
  else
    d->max_forward_distance = cfg->limits.vx_max * cfg->prediction_horizon;
  dwa_alloc(d);
  return d;
}

void ko_dwa_free(ko_dwa *d) {
  if (!d) return;
  ko_coll_free(d->coll);
  ko_path_free(d->path);
  free(d->svx);
  free(d->svy);
  free(d->som);
  free(d->px);
  free(d->py);
  free(d->costs);
  free(d->raw);
  free(d->best_x);
  free(d->best_y);
  for (int i = 0; i < 3; ++i) free(d->best_v[i]);
  free(d->ox);
  free(d->oy);
  free(d);
}

/* follower.cpp:80-105 */
int ko_dwa_set_path(ko_dwa *d, const float *x, const float *y, const float *z,
                    size_t n) {
  ko_path *p = ko_path_new(x, y, z, n);
  if (!p) return -1;
  ko_path_free(d->path);
  d->path = p;
  ko_path_interpolate_linear(p, d->max_point_interpolation_distance);
  ko_path_segment(p, d->path_segment_length, d->max_segment_size);
  d->max_segment_index = ko_path_num_segments(p) - 1;
  d->path_processing = 1;
  d->current_segment_index = 0;
  d->goal_distance = DBL_MAX;
  d->reached_goal = 0;
  return 0;
}

void ko_dwa_set_state(ko_dwa *d, double x, double y, double yaw, double speed) {
  d->state.x = x;
  d->state.y = y;
  d->state.yaw = yaw;
  d->state.speed = speed;
  /* DWA::setCurrentState also updates the sampler's checker, dwa.cpp:152-155;
   * the Python binding calls Controller::setCurrentState (controller.cpp:46-52)
   * which does not -- immaterial: generateTrajectories updates it again
   * (trajectory_sampler.cpp:299) */
  ko_coll_update_state(d->coll, x, y, yaw);
}
void ko_dwa_set_max_range(ko_dwa *d, float r) { d->max_local_range = r; }

/* follower.cpp:109-142 */
int ko_dwa_is_goal_reached(ko_dwa *d) {
  if (!d->path_processing) return 1;
  const ko_path *p = d->path;
  const float gx = p->X[p->size - 1], gy = p->Y[p->size - 1];
  int loosing_goal = 0;
  const double dist = hypot(d->state.x - (double)gx, d->state.y - (double)gy);
  int end_reached = dist <= d->goal_dist_tolerance;
  if ((d->current_segment_index + 1) >= d->max_segment_index) {
    if (dist < d->goal_distance) {
      d->goal_distance = dist;
      loosing_goal = 0;
    } else if (fabs(dist - d->goal_distance) > d->loosing_goal_distance) {
      loosing_goal = 1;
    }
  }
  if (end_reached || loosing_goal) {
    d->path_processing = 0;
    d->reached_goal = 1;
  }
  return d->reached_goal;
}

/* Path::distanceSquared(const State&, const Point&), path.h:214-217 */
static float state_point_dist_sq(const ko_state *s, float x, float y, float z) {
  return dist_sq3f((float)s->x, (float)s->y, 0.0f, x, y, z);
}

/* follower.cpp:155-183 */
static size_t find_closest_segment(ko_dwa *d, size_t left, size_t right) {
  if (left == right) return left;
  const ko_path *p = d->path;
  size_t mid = (left + right) / 2;
  size_t li = ko_path_segment_start(p, left), ri = ko_path_segment_start(p, right);
  float ld = state_point_dist_sq(&d->state, p->X[li], p->Y[li], p->Z[li]);
  float rd = state_point_dist_sq(&d->state, p->X[ri], p->Y[ri], p->Z[ri]);
  if (mid == right || mid == left) return (ld <= rd) ? left : right;
  if (ld <= rd) return find_closest_segment(d, left, mid);
  return find_closest_segment(d, mid, right);
}

/* follower.cpp:199-264 */
static path_position find_closest_on_segment(ko_dwa *d, size_t seg) {
  const ko_path *p = d->path;
  const size_t s0 = ko_path_segment_start(p, seg), s1 = ko_path_segment_end(p, seg);
  const size_t seg_size = s1 - s0 + 1;
  double min_d2 = (double)FLT_MAX;
  ko_state closest = {0, 0, 0, 0};
  double seg_pos = 0.0;
  size_t closest_idx = 0;
  /* std::atan2(float, float) -> float overload */
  double heading = (double)atan2f(p->Y[s1] - p->Y[s0], p->X[s1] - p->X[s0]);
  for (size_t k = 0; k < seg_size; ++k) {
    const size_t i = s0 + k;
    double d2 = (double)state_point_dist_sq(&d->state, p->X[i], p->Y[i], p->Z[i]);
    if (d2 <= min_d2) {
      min_d2 = d2;
      closest.x = p->X[i];
      closest.y = p->Y[i];
      closest.yaw = heading;
      closest_idx = k;
      if (seg_size > 1)
        seg_pos = (double)k / (double)(seg_size - 1);
      else
        seg_pos = 1.0;
    }
  }
  path_position pp;
  memset(&pp, 0, sizeof(pp));
  pp.index = closest_idx + s0;
  pp.segment_index = seg;
  pp.segment_length = seg_pos;
  pp.state = closest;
  pp.normal_distance = sqrt(min_d2);
  double vx = d->state.x - closest.x, vy = d->state.y - closest.y;
  double cross = cos(closest.yaw) * vy - sin(closest.yaw) * vx;
  pp.parallel_distance = cross > 0 ? pp.normal_distance : -pp.normal_distance;
  return pp;
}

/* utils/angles.h:21-29 */
static double normalize_mpi_pi(double a) {
  a = fmod(a + M_PI, 2 * M_PI);
  if (a < 0) a += 2 * M_PI;
  a -= M_PI;
  return a;
}

/* follower.cpp:266-304 */
static void determine_target(ko_dwa *d) {
  if ((d->closest.segment_length <= 0.0) ||
      (d->closest.index >=
       ko_path_segment_end(d->path, d->current_segment_index)) ||
      (d->closest.segment_length >= 0.9)) {
    d->current_segment_index = find_closest_segment(d, 0, d->max_segment_index);
    d->closest = find_closest_on_segment(d, d->current_segment_index);
  } else {
    d->closest = find_closest_on_segment(d, d->closest.segment_index);
  }
  d->heading_error = normalize_mpi_pi(d->closest.state.yaw - d->state.yaw);
}

/* trajectory_sampler.cpp:316-326 */
static void set_prediction_horizon(ko_dwa *d, double horizon) {
  const double min_h = 2.0 * d->cfg.time_step;
  if (horizon < min_h) horizon = min_h;
  if (horizon > d->base_max_time) horizon = d->base_max_time;
  d->max_time = horizon;
  d->P = ko_num_points_per_trajectory(d->cfg.time_step, d->max_time);
}

/* dwa.cpp:157-206 */
static void adapt_horizon(ko_dwa *d) {
  const double base = d->base_max_time;
  const double v_max = d->ctr_limits_base.vx_max; /* Q1: always 1.0 */
  if (!d->path || v_max < 1e-3 || d->max_point_interpolation_distance <= 0.0) {
    set_prediction_horizon(d, base);
    d->max_forward_distance = base * v_max;
    return;
  }
  const size_t psz = d->path->size;
  const size_t start = d->closest.index < psz - 1 ? d->closest.index : psz - 1;
  const size_t peek =
      (size_t)ceil(base * v_max / d->max_point_interpolation_distance);
  const size_t end = (start + peek < psz - 1) ? start + peek : psz - 1;
  float kappa_max = 0.0f;
  for (size_t i = start; i <= end; ++i) {
    /* Path::getCurvature returns double of a float; static_cast<float> */
    float k = fabsf((float)(double)d->path->K[i]);
    if (k > kappa_max) kappa_max = k;
  }
  double adaptive = base;
  if ((double)kappa_max > d->curvature_horizon_tolerance) {
    const double cap =
        sqrt(8.0 * d->curvature_horizon_tolerance / (double)kappa_max) / v_max;
    adaptive = dmin(base, cap);
  }
  set_prediction_horizon(d, adaptive);
  d->max_forward_distance = adaptive * v_max;
}

/* dwa.cpp:208-233 */
static void tracked_segment(ko_dwa *d, size_t *start, size_t *size) {
  const size_t psz = d->path->size;
  size_t gs = d->closest.index;
  if (gs >= psz) gs = psz - 1;
  size_t look = d->max_segment_size;
  if (d->max_point_interpolation_distance > 0.0) {
    size_t dyn = (size_t)ceil(d->max_forward_distance /
                              d->max_point_interpolation_distance) +
                 1;
    look = d->max_segment_size > dyn ? d->max_segment_size : dyn;
  }
  size_t ge = (gs + look < psz - 1) ? gs + look : psz - 1;
  *start = gs;
  *size = ge - gs + 1;
}

static int dwa_compute(ko_dwa *d, double vx, double vy, double om,
                       const double *ranges, const double *angles,
                       const float *xyz, size_t n, ko_dwa_result *res) {
  memset(res, 0, sizeof(*res));
  res->index = res->raw_index = -1;
  if (!d->path) return -1; /* dwa.h:187-191 std::invalid_argument */
  determine_target(d);
  /* rotate-in-place shortcut (dwa.h:195-205) is dead: rotate_in_place is
   * always false (Q1) */
  adapt_horizon(d);
  if (d->P > d->buf_cap_p - 2) return -3;

  /* generateTrajectories, trajectory_sampler.cpp:295-314 */
  ko_coll_update_state(d->coll, d->state.x, d->state.y, d->state.yaw);
  int rc = ranges ? ko_coll_update_scan(d->coll, ranges, angles, n)
                  : ko_coll_update_points(d->coll, xyz, n, 1);
  if (rc) return rc;
  long ng = ko_sample_velocities(
      d->cfg.ctr_type, &d->cfg.limits, vx, vy, om, d->cfg.time_step,
      d->cfg.max_linear_samples, d->cfg.max_angular_samples, d->svx, d->svy,
      d->som, d->buf_cap_n);
  if (ng < 0) return -4;
  res->n_generated = ng;
  res->P = d->P;
  long na = ko_rollout(d->coll, &d->state, d->cfg.time_step, d->P, d->svx,
                       d->svy, d->som, (size_t)ng, d->px, d->py, NULL, NULL,
                       NULL, d->raw);
  d->n_adm = na;
  res->n_admissible = na;
  if (na == 0) return 0; /* dwa.h:219-221 */

  /* setPointScan, dwa.h:223 */
  if (n > d->ocap) {
    free(d->ox);
    free(d->oy);
    d->ox = (float *)malloc(sizeof(float) * n);
    d->oy = (float *)malloc(sizeof(float) * n);
    d->ocap = n;
  }
  if (ranges)
    ko_obstacles_from_scan(d->cfg.sensor_pos, d->cfg.sensor_rot_xyzw, &d->state,
                           ranges, angles, n, d->ox, d->oy);
  else
    ko_obstacles_from_points(d->cfg.sensor_pos, d->cfg.sensor_rot_xyzw,
                             &d->state, xyz, n, d->ox, d->oy);

  size_t ss, sz;
  tracked_segment(d, &ss, &sz);
  res->seg_start = ss;
  res->seg_size = sz;

  ko_cost_ctx cx;
  memset(&cx, 0, sizeof(cx));
  cx.seg_x = d->path->X + ss;
  cx.seg_y = d->path->Y + ss;
  cx.seg_z = d->path->Z + ss;
  cx.seg_size = sz;
  cx.seg_start_idx = ss;
  cx.path_acc = d->path->acc;
  cx.path_acc_size = d->path->acc_size;
  cx.ref_path_length = ko_path_total_length(d->path);
  cx.obs_x = d->ox;
  cx.obs_y = d->oy;
  cx.n_obs = n;
  cx.max_obstacles_dist = d->max_local_range / 3.0f; /* cost_evaluator.h:179 */
  cx.acc_limits[0] = (float)d->cfg.limits.vx_acc;
  cx.acc_limits[1] = (float)d->cfg.limits.vy_acc;
  cx.acc_limits[2] = (float)d->cfg.limits.omega_acc;
  cx.w = d->cfg.weights;

  float min_cost;
  long best = ko_min_trajectory_cost(&cx, d->px, d->py, NULL, NULL, NULL,
                                     (size_t)na, d->P, d->P, 0, d->costs,
                                     &min_cost);
  if (best >= 0) {
    res->found = 1;
    res->cost = min_cost;
    res->index = best;
    res->raw_index = d->raw[best];
    memcpy(d->best_x, d->px + (size_t)best * d->P, sizeof(float) * d->P);
    memcpy(d->best_y, d->py + (size_t)best * d->P, sizeof(float) * d->P);
    const long r = d->raw[best];
    for (size_t i = 0; i + 1 < d->P; ++i) {
      d->best_v[0][i] = (float)d->svx[r];
      d->best_v[1][i] = (float)d->svy[r];
      d->best_v[2][i] = (float)d->som[r];
    }
  }
  return 0;
}

int ko_dwa_compute_scan(ko_dwa *d, double vx, double vy, double om,
                        const double *ranges, const double *angles, size_t n,
                        ko_dwa_result *res) {
  static const double dummy = 0.0;
  if (n == 0) {
    ranges = &dummy;
    angles = &dummy;
  }
  return dwa_compute(d, vx, vy, om, ranges, angles, NULL, n, res);
}
int ko_dwa_compute_points(ko_dwa *d, double vx, double vy, double om,
                          const float *xyz, size_t n, ko_dwa_result *res) {
  static const float dummy[3] = {0, 0, 0};
  if (n == 0) xyz = dummy;
  return dwa_compute(d, vx, vy, om, NULL, NULL, xyz, n, res);
}
const float *ko_dwa_best_path_x(const ko_dwa *d) { return d->best_x; }
const float *ko_dwa_best_path_y(const ko_dwa *d) { return d->best_y; }
const float *ko_dwa_best_vel(const ko_dwa *d, int c) { return d->best_v[c]; }
const float *ko_dwa_samples_x(const ko_dwa *d) { return d->px; }
const float *ko_dwa_samples_y(const ko_dwa *d) { return d->py; }
const float *ko_dwa_costs(const ko_dwa *d) { return d->costs; }
const int32_t *ko_dwa_raw_index(const ko_dwa *d) { return d->raw; }
const ko_path *ko_dwa_path(const ko_dwa *d) { return d->path; }
size_t ko_dwa_max_segment_size(const ko_dwa *d) { return d->max_segment_size; }
size_t ko_dwa_closest_index(const ko_dwa *d) { return d->closest.index; }

/* ======================================================================== */
/* M1/M2: LocalMapper CPU semantics                                          */
/* ======================================================================== */
typedef struct {
  int32_t *g;
  int H, W;
  int to0, to1;
} grid_sink;

static inline void grid_emit(grid_sink *s, int i, int j) {
  /* local_mapper.cpp:140-157 */
  if (i >= 0 && i < s->H && j >= 0 && j < s->W) {
    int32_t *cell = &s->g[(size_t)i + (size_t)j * (size_t)s->H];
    if (i == s->to0 && j == s->to1)
      *cell = KO_OCCUPIED;
    else if (*cell < KO_EMPTY)
      *cell = KO_EMPTY;
  }
}

/* line_drawing.h:55-124 */
static void bresenham_enhanced(int x0, int y0, int x1, int y1, grid_sink *s) {
  int x = x0, y = y0;
  int dx = x1 - x0, dy = y1 - y0;
  grid_emit(s, x, y);
  int xstep = (dx >= 0) ? 1 : -1;
  int ystep = (dy >= 0) ? 1 : -1;
  dx = abs(dx);
  dy = abs(dy);
  int ddy = 2 * dy, ddx = 2 * dx;
  if (ddx >= ddy) {
    int errorprev = dx, error = dx;
    for (int i = 0; i < dx; i++) {
      x += xstep;
      error += ddy;
      if (error > ddx) {
        y += ystep;
        error -= ddx;
        if (error + errorprev < ddx) {
          grid_emit(s, x, y - ystep);
        } else if (error + errorprev > ddx) {
          grid_emit(s, x - xstep, y);
        } else {
          grid_emit(s, x - xstep, y);
          grid_emit(s, x, y - ystep);
        }
      }
      grid_emit(s, x, y);
      errorprev = error;
    }
  } else {
    int errorprev = dy, error = dy;
    for (int i = 0; i < dy; i++) {
      y += ystep;
      error += ddx;
      if (error > ddy) {
        x += xstep;
        error -= ddy;
        if (error + errorprev < ddy) {
          grid_emit(s, x - xstep, y);
        } else if (error + errorprev > ddy) {
          grid_emit(s, x, y - ystep);
        } else {
          grid_emit(s, x - xstep, y);
          grid_emit(s, x, y - ystep);
        }
      }
      grid_emit(s, x, y);
      errorprev = error;
    }
  }
}

int ko_mapper_scan_to_grid(int H, int W, float res, const float pos[3],
                           float orient, const double *angles,
                           const double *ranges, size_t n, int32_t *grid) {
  /* local_mapper.h:26-31: round(H / 2) - 1 with integer division */
  const int c0 = (int)round((double)(H / 2)) - 1;
  const int c1 = (int)round((double)(W / 2)) - 1;
  /* localToGrid, local_mapper.h:210-222 */
  const int s0 = c0 + (int)(pos[0] / res);
  const int s1 = c1 + (int)(pos[1] / res);
  for (size_t k = 0; k < (size_t)H * (size_t)W; ++k) grid[k] = KO_UNEXPLORED;
  grid_sink s = {grid, H, W, 0, 0};
  for (size_t b = 0; b < n; ++b) {
    /* updateGrid_(const float angle, const float range), :127-134 */
    const float angle = (float)angles[b], range = (float)ranges[b];
    const float x =
        (float)((double)pos[0] + ((double)range * cos((double)(orient + angle))));
    const float y =
        (float)((double)pos[1] + ((double)range * sin((double)(orient + angle))));
    s.to0 = c0 + (int)(x / res);
    s.to1 = c1 + (int)(y / res);
    bresenham_enhanced(s0, s1, s.to0, s.to1, &s);
  }
  return 0;
}

/* ======================================================================== */
/* M3: Bayesian update + previous-grid warp (parity unpinned: the reference    */
/* tests only print these grids, mapper_test.cpp:136-220)                      */
/* ======================================================================== */
struct ko_bmap {
  int H, W;
  int c0, c1, s0, s1;
  float res, pos0, pos1, orient;
  float p_prior, p_occupied, p_empty, range_sure, range_max, wall_size;
  float *prev; /* previousGridDataProb, column-major [H x W] */
};

ko_bmap *ko_bmap_create(int H, int W, float res, const float pos[3], float orient,
                        float p_prior, float p_occupied, float p_empty,
                        float range_sure, float range_max, float wall_size) {
  if (H <= 0 || W <= 0 || !(res > 0.0f) || !pos) return NULL;
  ko_bmap *b = (ko_bmap *)calloc(1, sizeof(*b));
  if (!b) return NULL;
  b->H = H;
  b->W = W;
  b->res = res;
  b->pos0 = pos[0];
  b->pos1 = pos[1];
  b->orient = orient;
  b->c0 = (int)round((double)(H / 2)) - 1; /* local_mapper.h:72-73 */
  b->c1 = (int)round((double)(W / 2)) - 1;
  b->s0 = b->c0 + (int)(pos[0] / res);
  b->s1 = b->c1 + (int)(pos[1] / res);
  b->p_prior = p_prior;
  b->p_occupied = p_occupied;
  b->p_empty = p_empty;
  b->range_sure = range_sure;
  b->range_max = range_max;
  b->wall_size = wall_size;
  b->prev = (float *)malloc(sizeof(float) * (size_t)H * (size_t)W);
  if (!b->prev) {
    free(b);
    return NULL;
  }
  /* local_mapper.h:81-83 */
  for (size_t k = 0; k < (size_t)H * (size_t)W; ++k) b->prev[k] = p_prior;
  return b;
}

void ko_bmap_destroy(ko_bmap *b) {
  if (!b) return;
  free(b->prev);
  free(b);
}

const float *ko_bmap_previous(const ko_bmap *b) { return b->prev; }

void ko_bmap_set_previous(ko_bmap *b, const float *prob) {
  memcpy(b->prev, prob, sizeof(float) * (size_t)b->H * (size_t)b->W);
}

/* LocalMapper::updateGridCellProbability, local_mapper.cpp:106-125.  The
 * literal 1.0 makes the sensor odds, and everything multiplied with them, a
 * double expression; the result narrows to float on return. */
static float bayes_cell(const ko_bmap *b, float distance, float current_range,
                        float previous_prob) {
  distance = distance * b->res;
  current_range = current_range - b->wall_size;
  const float pF = (distance < current_range) ? b->p_empty : b->p_occupied;
  const float delta = (distance < b->range_sure) ? 0.0f : 1.0f;
  const float p_sensor =
      pF + (delta * ((distance - b->range_sure) / b->range_max) * (b->p_prior - pF));
  const float prev_odds = previous_prob / (1 - previous_prob);
  const double sensor_odds = (double)p_sensor / (1.0 - (double)p_sensor);
  const float prior_odds = (1 - b->p_prior) / b->p_prior;
  const double p_curr =
      1 - (1 / (1 + (((double)prev_odds * sensor_odds) * (double)prior_odds)));
  return (float)p_curr;
}

typedef struct {
  grid_sink g;
  const ko_bmap *b;
  float *prob;
  float range;
} bayes_sink;

/* the per-point body of updateGridBaysian_, local_mapper.cpp:177-201 */
static inline void bayes_emit(bayes_sink *s, int i, int j) {
  const ko_bmap *b = s->b;
  if (i >= 0 && i < b->H && j >= 0 && j < b->W) {
    /* (pt - m_startPoint).norm() on Vector2i: Eigen's integer norm, the double
     * sqrt truncated back to int */
    const int di = i - b->s0, dj = j - b->s1;
    const float distance = (float)(int)sqrt((double)(di * di + dj * dj));
    const size_t k = (size_t)i + (size_t)j * (size_t)b->H;
    s->prob[k] = bayes_cell(b, distance, s->range, b->prev[k]);
    grid_emit(&s->g, i, j);
  }
}

/* bresenhamEnhanced (line_drawing.h:55-124) feeding bayes_emit */
static void bresenham_bayes(int x0, int y0, int x1, int y1, bayes_sink *s) {
  int x = x0, y = y0;
  int dx = x1 - x0, dy = y1 - y0;
  bayes_emit(s, x, y);
  const int xstep = (dx >= 0) ? 1 : -1, ystep = (dy >= 0) ? 1 : -1;
  dx = abs(dx);
  dy = abs(dy);
  const int ddy = 2 * dy, ddx = 2 * dx;
  const int xmajor = ddx >= ddy;
  const int n = xmajor ? dx : dy, ddmaj = xmajor ? ddx : ddy, ddmin = xmajor ? ddy : ddx;
  int errorprev = n, error = n;
  for (int i = 0; i < n; i++) {
    if (xmajor) x += xstep; else y += ystep;
    error += ddmin;
    if (error > ddmaj) {
      if (xmajor) y += ystep; else x += xstep;
      error -= ddmaj;
      /* x-major emits (x, y-ystep) below the line and (x-xstep, y) above; the
       * y-major branch mirrors that */
      const int lo = xmajor ? (error + errorprev < ddmaj) : (error + errorprev > ddmaj);
      const int hi = xmajor ? (error + errorprev > ddmaj) : (error + errorprev < ddmaj);
      if (lo) {
        bayes_emit(s, x, y - ystep);
      } else if (hi) {
        bayes_emit(s, x - xstep, y);
      } else {
        bayes_emit(s, x - xstep, y);
        bayes_emit(s, x, y - ystep);
      }
    }
    bayes_emit(s, x, y);
    errorprev = error;
  }
}

/* LocalMapper::scanToGridBaysian, local_mapper.cpp:222-241 (single thread:
 * the last beam that crosses a cell decides its probability) */
int ko_bmap_scan(ko_bmap *b, const double *angles, const double *ranges, size_t n,
                 int32_t *grid, float *prob) {
  const size_t cells = (size_t)b->H * (size_t)b->W;
  for (size_t k = 0; k < cells; ++k) grid[k] = KO_UNEXPLORED;
  for (size_t k = 0; k < cells; ++k) prob[k] = b->p_prior;
  bayes_sink s = {{grid, b->H, b->W, 0, 0}, b, prob, 0.0f};
  for (size_t q = 0; q < n; ++q) {
    const float angle = (float)angles[q], range = (float)ranges[q];
    const float x =
        (float)((double)b->pos0 + ((double)range * cos((double)(b->orient + angle))));
    const float y =
        (float)((double)b->pos1 + ((double)range * sin((double)(b->orient + angle))));
    s.g.to0 = b->c0 + (int)(x / b->res);
    s.g.to1 = b->c1 + (int)(y / b->res);
    s.range = range;
    bresenham_bayes(b->s0, b->s1, s.g.to0, s.g.to1, &s);
  }
  return 0;
}

/* LocalMapper::getPreviousGridInCurrentPose, local_mapper.cpp:17-78.  The
 * reference inverts the same Matrix3f for every cell; the inverse is Eigen's
 * closed 3x3 form (cofactors, determinant over column 0 as a0 + (a1 + a2)) and
 * the product rows are lazy 3-term reductions a0 + (a1 + a2). */
static inline float cof3(const float m[3][3], int i, int j) {
  const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
}

void ko_bmap_warp_matrix(const ko_bmap *b, const float pos[2], double orient, float inv[3][3]) {
  const int cc0 = b->c0 + (int)(pos[0] / b->res); /* localToGrid */
  const int cc1 = b->c1 + (int)(pos[1] / b->res);
  const double ang = -1 * orient;
  const double c = cos(ang), s = sin(ang);
  float m[3][3];
  m[0][0] = (float)c;
  m[0][1] = (float)(-s);
  m[0][2] = (float)(0.5 * b->H - cc1 + (cc0 * s - cc1 * c));
  m[1][0] = (float)s;
  m[1][1] = (float)c;
  m[1][2] = (float)(0.5 * b->W - cc0 - (cc0 * c + cc1 * s));
  m[2][0] = 0.0f;
  m[2][1] = 0.0f;
  m[2][2] = 1.0f;
  const float k0 = cof3(m, 0, 0), k1 = cof3(m, 1, 0), k2 = cof3(m, 2, 0);
  const float det = k0 * m[0][0] + (k1 * m[1][0] + k2 * m[2][0]);
  const float invdet = 1.0f / det;
  inv[0][0] = k0 * invdet;
  inv[0][1] = k1 * invdet;
  inv[0][2] = k2 * invdet;
  for (int r = 1; r < 3; ++r)
    for (int q = 0; q < 3; ++q) inv[r][q] = cof3(m, q, r) * invdet;
}

int ko_bmap_warp(ko_bmap *b, const float pos[2], double orient) {
  float inv[3][3];
  ko_bmap_warp_matrix(b, pos, orient, inv);
  const size_t cells = (size_t)b->H * (size_t)b->W;
  float *out = (float *)malloc(sizeof(float) * cells);
  if (!out) return -1;
  for (size_t k = 0; k < cells; ++k) out[k] = b->p_prior;
  const int rows = b->H, cols = b->W;
  for (int y = 0; y < b->H; ++y) {
    for (int x = 0; x < b->W; ++x) {
      const float fx = (float)x, fy = (float)y;
      const double srcX = (double)(inv[0][0] * fx + (inv[0][1] * fy + inv[0][2] * 1.0f));
      const double srcY = (double)(inv[1][0] * fx + (inv[1][1] * fy + inv[1][2] * 1.0f));
      if (srcX >= 0 && srcX < cols - 1 && srcY >= 0 && srcY < rows - 1) {
        const int x0 = (int)floor(srcX), y0 = (int)floor(srcY);
        const int x1 = x0 + 1, y1 = y0 + 1;
        const float w0 = (float)(srcX - x0), w1 = 1.0f - w0;
        const float h0 = (float)(srcY - y0), h1 = 1.0f - h0;
#define KO_P(r, c) b->prev[(size_t)(r) + (size_t)(c) * (size_t)rows]
        const float value = h1 * (w1 * KO_P(y0, x0) + w0 * KO_P(y0, x1)) +
                            h0 * (w1 * KO_P(y1, x0) + w0 * KO_P(y1, x1));
#undef KO_P
        out[(size_t)y + (size_t)x * (size_t)rows] = value;
      }
    }
  }
  free(b->prev);
  b->prev = out;
  return 0;
}

/* ======================================================================== */
/* bounded multi-thread baseline (bench.py cpu_baseline only)                */
/* ======================================================================== */
typedef struct {
  ko_coll *coll;
  const ko_cost_ctx *cx;
  const ko_state *start;
  double dt;
  size_t P;
  const double *vx, *vy, *om;
  size_t lo, hi, stride;
  float best_cost;
  long best_idx, n_adm;
} bl_job;

static void *bl_worker(void *arg) {
  bl_job *j = (bl_job *)arg;
  float *px = (float *)malloc(sizeof(float) * j->P);
  float *py = (float *)malloc(sizeof(float) * j->P);
  j->best_cost = KO_DEFAULT_MIN_DIST;
  j->best_idx = -1;
  j->n_adm = 0;
  for (size_t k = j->lo; k < j->hi; k += j->stride) {
    if (!rollout_one(j->coll, j->start, j->dt, j->P, j->vx[k], j->vy[k],
                     j->om[k], px, py))
      continue;
    j->n_adm++;
    float c = total_cost_one(j->cx, px, py, NULL, NULL, NULL, j->P);
    if (c < j->best_cost) {
      j->best_cost = c;
      j->best_idx = (long)k;
    }
  }
  free(px);
  free(py);
  return NULL;
}

long ko_baseline_cycle(ko_coll *coll, const ko_cost_ctx *cx,
                       const ko_state *start, double dt, size_t P,
                       const double *vx, const double *vy, const double *om,
                       size_t n, int threads, float *min_cost_out,
                       long *n_adm_out) {
  if (threads < 1) threads = 1;
  if ((size_t)threads > n) threads = (int)(n ? n : 1);
  bl_job *jobs = (bl_job *)calloc((size_t)threads, sizeof(bl_job));
  pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; ++t) {
    jobs[t].coll = coll;
    jobs[t].cx = cx;
    jobs[t].start = start;
    jobs[t].dt = dt;
    jobs[t].P = P;
    jobs[t].vx = vx;
    jobs[t].vy = vy;
    jobs[t].om = om;
    /* interleaved assignment (dynamic-like balance of the per-sample tasks of
     * trajectory_sampler.cpp:192-205) */
    jobs[t].lo = (size_t)t;
    jobs[t].hi = n;
    jobs[t].stride = (size_t)threads;
    if (threads == 1)
      bl_worker(&jobs[t]);
    else
      pthread_create(&tid[t], NULL, bl_worker, &jobs[t]);
  }
  float best = KO_DEFAULT_MIN_DIST;
  long bi = -1, na = 0;
  for (int t = 0; t < threads; ++t) {
    if (threads > 1) pthread_join(tid[t], NULL);
    na += jobs[t].n_adm;
    if (jobs[t].best_idx >= 0 &&
        (jobs[t].best_cost < best ||
         (jobs[t].best_cost == best && jobs[t].best_idx < bi))) {
      best = jobs[t].best_cost;
      bi = jobs[t].best_idx;
    }
  }
  free(jobs);
  free(tid);
  if (min_cost_out) *min_cost_out = best;
  if (n_adm_out) *n_adm_out = na;
  return bi;
}


/* ---- full-size parity helper (tests only): the same per-sample work as
 * ko_rollout + ko_min_trajectory_cost (trajectory_sampler.cpp:118-179,
 * cost_evaluator.cpp:49-109), every sample evaluated independently by
 * `threads` workers that pull blocks of 16 samples from a shared counter.  All
 * per-sample outputs are kept (raw numbering): px/py [n][P], adm [n] (1 =
 * admissible), costs [n] (only where adm).  The caller compacts and takes the
 * first strict minimum, exactly as the serial loop does. */
typedef struct {
  ko_coll *coll;
  const ko_cost_ctx *cx;
  const ko_state *start;
  double dt;
  size_t P, n;
  const double *vx, *vy, *om;
  float *px, *py, *costs;
  uint8_t *adm;
  size_t *next;
  pthread_mutex_t *mu;
  long n_adm;
  int mode, drop;          /* mode: ko_full_cycle_mode (velocity profiles kept, both drop_samples values) */
  size_t num_ctrl;
  float *fvx, *fvy, *fom;  /* [n][P - 1] */
} fc_job;

static void *fc_worker(void *arg) {
  fc_job *j = (fc_job *)arg;
  j->n_adm = 0;
  for (;;) {
    pthread_mutex_lock(j->mu);
    const size_t k0 = *j->next;
    *j->next = k0 + 16;
    pthread_mutex_unlock(j->mu);
    if (k0 >= j->n) break;
    const size_t k1 = k0 + 16 < j->n ? k0 + 16 : j->n;
    for (size_t k = k0; k < k1; ++k) {
      float *px = j->px + k * j->P, *py = j->py + k * j->P;
      if (j->mode) {
        const size_t nv = j->P - 1;
        float *fx = j->fvx + k * nv, *fy = j->fvy + k * nv, *fo = j->fom + k * nv;
        const int ok = rollout_one_mode(j->coll, j->start, j->dt, j->P, j->vx[k], j->vy[k], j->om[k], j->drop,
                                        j->num_ctrl, px, py, fx, fy, fo);
        j->adm[k] = ok ? 1 : 0;
        if (!ok) continue;
        j->n_adm++;
        if (j->cx && j->costs) j->costs[k] = total_cost_one(j->cx, px, py, fx, fy, fo, j->P);
        continue;
      }
      const int ok = rollout_one(j->coll, j->start, j->dt, j->P, j->vx[k],
                                 j->vy[k], j->om[k], px, py);
      j->adm[k] = ok ? 1 : 0;
      if (!ok) continue;
      j->n_adm++;
      if (j->cx && j->costs)
        j->costs[k] = total_cost_one(j->cx, px, py, NULL, NULL, NULL, j->P);
    }
  }
  return NULL;
}

long ko_full_cycle_mode(ko_coll *coll, const ko_cost_ctx *cx, const ko_state *start, double dt, size_t P,
                        const double *vx, const double *vy, const double *om, size_t n, int threads,
                        int drop_samples, size_t num_ctrl_points, float *px, float *py, float *fvx,
                        float *fvy, float *fom, uint8_t *adm, float *costs);

long ko_full_cycle(ko_coll *coll, const ko_cost_ctx *cx, const ko_state *start,
                   double dt, size_t P, const double *vx, const double *vy,
                   const double *om, size_t n, int threads, float *px,
                   float *py, uint8_t *adm, float *costs) {
  return ko_full_cycle_mode(coll, cx, start, dt, P, vx, vy, om, n, threads, 1, 0, px, py, NULL, NULL, NULL, adm, costs);
}

long ko_full_cycle_mode(ko_coll *coll, const ko_cost_ctx *cx, const ko_state *start, double dt, size_t P,
                        const double *vx, const double *vy, const double *om, size_t n, int threads,
                        int drop_samples, size_t num_ctrl_points, float *px, float *py, float *fvx,
                        float *fvy, float *fom, uint8_t *adm, float *costs) {
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  fc_job jobs[64];
  pthread_t tid[64];
  size_t next = 0;
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  for (int t = 0; t < threads; ++t) {
    fc_job *j = &jobs[t];
    j->coll = coll;
    j->cx = cx;
    j->start = start;
    j->dt = dt;
    j->P = P;
    j->n = n;
    j->vx = vx;
    j->vy = vy;
    j->om = om;
    j->px = px;
    j->py = py;
    j->costs = costs;
    j->adm = adm;
    j->next = &next;
    j->mu = &mu;
    j->n_adm = 0;
    j->mode = fvx != NULL;
    j->drop = drop_samples;
    j->num_ctrl = num_ctrl_points;
    j->fvx = fvx;
    j->fvy = fvy;
    j->fom = fom;
    if (threads == 1)
      fc_worker(j);
    else
      pthread_create(&tid[t], NULL, fc_worker, j);
  }
  long na = 0;
  for (int t = 0; t < threads; ++t) {
    if (threads > 1) pthread_join(tid[t], NULL);
    na += jobs[t].n_adm;
  }
  return na;
}

/* the same for caller-provided trajectories (cost_evaluator.cpp:49-109 body
 * per sample; velocities optional): costs [n] */
typedef struct {
  const ko_cost_ctx *cx;
  const float *px, *py, *vx, *vy, *om;
  size_t n, P, lo, hi;
  float *costs;
} ce_job;

static void *ce_worker(void *arg) {
  ce_job *j = (ce_job *)arg;
  const size_t nv = j->P - 1;
  for (size_t k = j->lo; k < j->hi; ++k)
    j->costs[k] = total_cost_one(j->cx, j->px + k * j->P, j->py + k * j->P,
                                 j->vx ? j->vx + k * nv : NULL,
                                 j->vy ? j->vy + k * nv : NULL,
                                 j->om ? j->om + k * nv : NULL, j->P);
  return NULL;
}

void ko_costs_mt(const ko_cost_ctx *cx, const float *px, const float *py,
                 const float *vx, const float *vy, const float *om, size_t n,
                 size_t P, int threads, float *costs) {
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  ce_job jobs[64];
  pthread_t tid[64];
  for (int t = 0; t < threads; ++t) {
    ce_job *j = &jobs[t];
    j->cx = cx;
    j->px = px;
    j->py = py;
    j->vx = vx;
    j->vy = vy;
    j->om = om;
    j->n = n;
    j->P = P;
    j->lo = n * (size_t)t / (size_t)threads;
    j->hi = n * (size_t)(t + 1) / (size_t)threads;
    j->costs = costs;
    if (threads == 1)
      ce_worker(j);
    else
      pthread_create(&tid[t], NULL, ce_worker, j);
  }
  if (threads > 1)
    for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
}


/* ===========================================================================
 * M5: pointCloudToLaserScanFromRaw, utils/pointcloud.h:116-177 and :205-259
 * =========================================================================== */
/* load_and_cast_val, utils/pointcloud.h:49-87: a field of any PointCloud2 datatype (ids 1-8, :37-46) as
 * float, read byte by byte (no alignment assumed); unknown ids give 0 */
static size_t field_size(int t) {
  switch (t) {
    case 1: case 2: return 1;
    case 3: case 4: return 2;
    case 5: case 6: case 7: return 4;
    case 8: return 8;
    default: return 4;
  }
}
static float load_and_cast(const int8_t *addr, int t) {
  switch (t) {
    case 1: return (float)*addr;
    case 2: return (float)*(const uint8_t *)addr;
    case 3: { int16_t v; memcpy(&v, addr, 2); return (float)v; }
    case 4: { uint16_t v; memcpy(&v, addr, 2); return (float)v; }
    case 5: { int32_t v; memcpy(&v, addr, 4); return (float)v; }
    case 6: { uint32_t v; memcpy(&v, addr, 4); return (float)v; }
    case 7: { float v; memcpy(&v, addr, 4); return v; }
    case 8: { double v; memcpy(&v, addr, 8); return (float)v; }
    default: return 0.0f;
  }
}

long ko_pointcloud_to_laserscan_typed(const int8_t *data, size_t nbytes, int point_step,
                                      int row_step, int height, int width, int x_offset,
                                      int y_offset, int z_offset, double max_range,
                                      double min_z, double max_z, double angle_step,
                                      int num_bins, int field_type, double *ranges_out,
                                      double *angles_out, size_t cap);

long ko_pointcloud_to_laserscan(const int8_t *data, size_t nbytes, int point_step,
                                int row_step, int height, int width, int x_offset,
                                int y_offset, int z_offset, double max_range,
                                double min_z, double max_z, double angle_step,
                                int num_bins, double *ranges_out,
                                double *angles_out, size_t cap) {
  return ko_pointcloud_to_laserscan_typed(data, nbytes, point_step, row_step, height, width, x_offset,
                                          y_offset, z_offset, max_range, min_z, max_z, angle_step, num_bins,
                                          7, ranges_out, angles_out, cap);
}

/* The CPU loop of pointcloud.h:116-177 / :205-259 with the fields decoded by load_and_cast_val (what the
 * reference's device paths do for non-FLOAT32 clouds, local_mapper_gpu.cpp:117-140,
 * critical_zone_check_gpu.cpp; bounds check with the field's own size).  FLOAT32 (7) is the CPU function. */
long ko_pointcloud_to_laserscan_typed(const int8_t *data, size_t nbytes, int point_step,
                                      int row_step, int height, int width, int x_offset,
                                      int y_offset, int z_offset, double max_range,
                                      double min_z, double max_z, double angle_step,
                                      int num_bins, int field_type, double *ranges_out,
                                      double *angles_out, size_t cap) {
  (void)width; /* pointcloud.h:137-138: the loops use row_step / point_step only */
  const double two_pi = 2.0 * M_PI;
  const int by_step = angle_step > 0.0;
  if (point_step <= 0 || row_step < 0 || height < 0 || x_offset < 0 ||
      y_offset < 0 || z_offset < 0 || !ranges_out)
    return -1;
  if (by_step) num_bins = (int)ceil(two_pi / angle_step); /* :124 */
  if (num_bins <= 0 || (size_t)num_bins > cap) return -1;
  for (int i = 0; i < num_bins; ++i) { /* :127-132 / :214 */
    if (by_step && angles_out) angles_out[i] = i * angle_step;
    ranges_out[i] = max_range;
  }
  int max_off = x_offset > y_offset ? x_offset : y_offset;
  if (z_offset > max_off) max_off = z_offset;
  for (int row = 0; row < height; ++row) {
    for (int col = 0; col < row_step; col += point_step) {
      const size_t point_start = (size_t)row * (size_t)row_step + (size_t)col;
      if (point_start + (size_t)max_off + field_size(field_type) > nbytes) continue; /* :139-146 */
      const float x = load_and_cast(data + point_start + x_offset, field_type);
      const float y = load_and_cast(data + point_start + y_offset, field_type);
      const float z = load_and_cast(data + point_start + z_offset, field_type);
      const float xx = x * x, yy = y * y;
      const float range_sq = xx + yy; /* :153 */
      if ((double)range_sq < 1e-6) continue;
      if ((double)z < min_z || (max_z >= 0.0 && (double)z > max_z)) continue; /* :159 */
      if (!isfinite(x) || !isfinite(y)) continue; /* reference: int(NaN) index, UB */
      double angle = (double)atan2f(y, x); /* std::atan2(float, float) */
      if (angle < 0.0) angle += two_pi;
      int bin = by_step ? (int)(angle / angle_step)            /* :168 */
                        : (int)((angle / two_pi) * num_bins);  /* :250 */
      if (bin > num_bins - 1) bin = num_bins - 1;
      const double distance = (double)sqrtf(range_sq); /* std::sqrt(float) */
      if (distance < ranges_out[bin]) ranges_out[bin] = distance;
    }
  }
  return num_bins;
}


/* ===========================================================================
 * CriticalZoneChecker, utils/critical_zone_check.cpp
 * =========================================================================== */
struct ko_czc {
  int field_type;          /* PointFieldType of raw clouds (0 / 7: FLOAT32) */
  double robot_radius;     /* robotRadius_ (double member) */
  float min_height, max_height, range_max;
  float critical_angle;    /* half cone, normalised, stored as float */
  float critical_distance, slowdown_distance;
  size_t n;
  float *sin_a, *cos_a;
  size_t *fwd, *bwd;
  size_t n_fwd, n_bwd;
  iso3f tf;                /* sensor_tf_body_ */
};

ko_czc *ko_czc_create(int shape, const float *dims, const float sensor_pos[3],
                      const float sensor_rot_xyzw[4], float critical_angle_deg,
                      float critical_distance, float slowdown_distance,
                      const double *angles, size_t n, float min_height,
                      float max_height, float range_max) {
  if (!(slowdown_distance > critical_distance)) return NULL; /* :52-56 */
  ko_czc *z = (ko_czc *)calloc(1, sizeof(*z));
  if (!z) return NULL;
  z->min_height = min_height;
  z->max_height = max_height;
  z->range_max = range_max;
  if (shape == 0) { /* :26-28 */
    z->robot_radius = dims[0];
  } else if (shape == 1) { /* :29-33: std::sqrt(pow(float,2) + pow(float,2)) / 2 in double */
    z->robot_radius = sqrt(pow((double)dims[0], 2) + pow((double)dims[1], 2)) / 2;
  } else if (shape == 2) {
    z->robot_radius = dims[0];
  } else {
    free(z);
    return NULL;
  }
  /* Eigen::Quaternionf(Vector4f) takes the coefficients in (x, y, z, w) order */
  quatf q = {sensor_rot_xyzw[3], sensor_rot_xyzw[0], sensor_rot_xyzw[1], sensor_rot_xyzw[2]};
  z->tf = iso_from_quat(q, sensor_pos);
  /* :46-48: float angle_rad = critical_angle * M_PI / 180.0 (double product, float store) */
  const float angle_rad = (float)((double)critical_angle_deg * M_PI / 180.0);
  z->critical_angle = (float)normalize_mpi_pi((double)(angle_rad / 2));
  z->critical_distance = critical_distance;
  z->slowdown_distance = slowdown_distance;
  z->n = n;
  z->sin_a = (float *)malloc(sizeof(float) * (n ? n : 1));
  z->cos_a = (float *)malloc(sizeof(float) * (n ? n : 1));
  z->fwd = (size_t *)malloc(sizeof(size_t) * (n ? n : 1));
  z->bwd = (size_t *)malloc(sizeof(size_t) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) { /* preset, :60-83 */
    z->cos_a[i] = (float)cos(angles[i]);
    z->sin_a[i] = (float)sin(angles[i]);
    float p[3];
    iso_apply(&z->tf, z->cos_a[i], z->sin_a[i], 0.0f, p);
    const float abs_theta = fabsf(atan2f(p[1], p[0]));
    if (abs_theta <= z->critical_angle) z->fwd[z->n_fwd++] = i;
    /* float >= double(M_PI - float) */
    if ((double)abs_theta >= M_PI - (double)z->critical_angle) z->bwd[z->n_bwd++] = i;
  }
  return z;
}

void ko_czc_destroy(ko_czc *z) {
  if (!z) return;
  free(z->sin_a);
  free(z->cos_a);
  free(z->fwd);
  free(z->bwd);
  free(z);
}

size_t ko_czc_indices(const ko_czc *z, int forward, size_t *out, size_t cap) {
  const size_t n = forward ? z->n_fwd : z->n_bwd;
  const size_t *src = forward ? z->fwd : z->bwd;
  for (size_t i = 0; i < n && i < cap; ++i) out[i] = src[i];
  return n;
}

float ko_czc_check(const ko_czc *z, const double *ranges, int forward) { /* :85-117 */
  const size_t *idx = forward ? z->fwd : z->bwd;
  const size_t n = forward ? z->n_fwd : z->n_bwd;
  float slowdown_factor = 1.0f;
  for (size_t k = 0; k < n; ++k) {
    const size_t i = idx[k];
    const float x = (float)(ranges[i] * (double)z->cos_a[i]);
    const float y = (float)(ranges[i] * (double)z->sin_a[i]);
    float p[3];
    iso_apply(&z->tf, x, y, 0.0f, p);
    /* std::sqrt(std::pow(float, 2) + std::pow(float, 2)): double, stored as float */
    const float converted_range = (float)sqrt(pow((double)p[1], 2) + pow((double)p[0], 2));
    const float distance = (float)((double)converted_range - z->robot_radius);
    if (distance <= z->critical_distance) return 0.0f;
    if (distance <= z->slowdown_distance) {
      const float f = (distance - z->critical_distance) /
                      (z->slowdown_distance - z->critical_distance);
      if (f < slowdown_factor) slowdown_factor = f; /* std::min(a, b): b < a ? b : a */
    }
  }
  return slowdown_factor;
}

float ko_czc_check_cloud(const ko_czc *z, const int8_t *data, size_t nbytes,
                         int point_step, int row_step, int height, int width,
                         int x_offset, int y_offset, int z_offset, int forward) { /* :119-131 */
  if (z->n == 0) return 1.0f;
  double *ranges = (double *)malloc(sizeof(double) * z->n);
  const long nb = ko_pointcloud_to_laserscan_typed(data, nbytes, point_step, row_step, height, width,
                                                   x_offset, y_offset, z_offset, (double)z->range_max,
                                                   (double)z->min_height, (double)z->max_height, 0.0,
                                                   (int)z->n, z->field_type ? z->field_type : 7, ranges, NULL, z->n);
  float r = 1.0f;
  if (nb == (long)z->n) r = ko_czc_check(z, ranges, forward);
  free(ranges);
  return r;
}

void ko_czc_set_field_type(ko_czc *z, int field_type) { z->field_type = field_type; }
