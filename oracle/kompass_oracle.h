/*
 * kompass_oracle.h -- CPU restatement (TEST INFRASTRUCTURE ONLY) of the
 * kompass_cpp sampling-controller hot path and LocalMapper.
 *
 * This is the parity oracle: a plain-C restatement of the reference's CPU
 * algorithm, function by function, each citing the reference file:line it
 * follows (paths relative to the reference checkout,
 * src/kompass_cpp/kompass_cpp/...).  It is NOT part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product path (kompass-core_amd/) never links, imports or calls it.
 *
 * Pinning: checked against the reference's own known-answer tests
 * (tests/cost_evaluator_test.cpp:217-461, 12 closed-form cost cases;
 * tests/collisions_test.cpp:25-77, 3 booleans) in tests/test_oracle_kat.py.
 * The reference itself cannot be compiled here (needs Eigen/FCL/octomap/OMPL,
 * all absent) -- see DESIGN.md.  Parity UNPINNED parts: FCL/GJK boundary
 * behaviour at float-ulp level (collision semantics are restated
 * analytically), Eigen evaluation order inside 3-term float reductions
 * (restated as a0 + (a1 + a2)), mapper cell-level output (reference holds
 * invariants only).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -march => no FMA).
 */
#ifndef KOMPASS_ORACLE_H
#define KOMPASS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums (values follow the reference) -------------------------------- */
/* datatypes/control.h:12 */
enum { KO_ACKERMANN = 0, KO_DIFFERENTIAL_DRIVE = 1, KO_OMNI = 2 };
/* utils/collision_check.h:25 */
enum { KO_CYLINDER = 0, KO_BOX = 1, KO_SPHERE = 2 };
/* mapping/local_mapper.h:9 */
enum { KO_UNEXPLORED = -1, KO_EMPTY = 0, KO_OCCUPIED = 100 };

/* ---- plain structs ------------------------------------------------------ */
typedef struct {
  double x, y, yaw, speed; /* datatypes/path.h:14-22 */
} ko_state;

typedef struct {
  /* datatypes/control.h:191-235 */
  double vx_max, vx_acc, vx_dec;
  double vy_max, vy_acc, vy_dec;
  double omega_max_angle, omega_max, omega_acc, omega_dec;
} ko_limits;

typedef struct {
  /* utils/cost_evaluator.h:22-50 (parameter names verbatim, order =
   * accumulation order of cost_evaluator.cpp:61-93 is goal, path, obstacles,
   * smoothness, jerk) */
  double reference_path_distance_weight;
  double goal_distance_weight;
  double obstacles_distance_weight;
  double smoothness_weight;
  double jerk_weight;
} ko_weights;

/* ---- Path (datatypes/path.h:37-299, src/datatypes/path.cpp) -------------- */
typedef struct ko_path ko_path;
ko_path *ko_path_new(const float *x, const float *y, const float *z, size_t n);
ko_path *ko_path_clone(const ko_path *p);
void ko_path_free(ko_path *p);
/* path.cpp:167-288, LINEAR type only (tk::spline linear, spline.h:197-225,
 * 390-422). Returns 0 on success. */
int ko_path_interpolate_linear(ko_path *p, double max_interpolation_point_dist);
/* path.cpp:290-330 */
void ko_path_segment(ko_path *p, double path_segment_length,
                     size_t max_points_per_segment);
size_t ko_path_size(const ko_path *p);
const float *ko_path_x(const ko_path *p);
const float *ko_path_y(const ko_path *p);
const float *ko_path_z(const ko_path *p);
const float *ko_path_curvature(const ko_path *p);
const float *ko_path_acc(const ko_path *p);
size_t ko_path_acc_size(const ko_path *p);
float ko_path_total_length(const ko_path *p);        /* path.cpp:148-165 */
size_t ko_path_num_segments(const ko_path *p);
size_t ko_path_segment_start(const ko_path *p, size_t seg); /* path.cpp:373-381 */
size_t ko_path_segment_end(const ko_path *p, size_t seg);   /* path.cpp:383-398 */

/* ---- A1: dynamic window + velocity lattice ------------------------------ */
/* trajectory.h:19-51 */
void ko_linear_sample_split(int ctr_type, int max_linear_samples, int *vx_n,
                            int *vy_n);
size_t ko_num_trajectories(int ctr_type, int max_linear_samples,
                           int max_angular_samples_bumped);
size_t ko_num_points_per_trajectory(double time_step, double horizon);
/* trajectory_sampler.cpp:328-372 (window) + :181-220 / :256-272 (lattice,
 * single-thread ordering).  The all-zero sample filter of :122-125 is applied
 * here so the returned list is exactly the list of roll-outs attempted.
 * Writes up to cap triples, returns the count (or -1 if cap too small).  */
long ko_sample_velocities(int ctr_type, const ko_limits *lim /* raw limits */,
                          double cur_vx, double cur_vy, double cur_omega,
                          double time_step, int max_linear_samples,
                          int max_angular_samples /* un-bumped */, double *vx,
                          double *vy, double *omega, size_t cap);

/* ---- A4: collision checker (restated, analytic) -------------------------- */
typedef struct ko_coll ko_coll;
/* collision_check.cpp:18-68.  sensor_rot is Eigen coeff order (x,y,z,w). */
ko_coll *ko_coll_new(int shape, const float *dims, int ndims,
                     const float sensor_pos[3], const float sensor_rot_xyzw[4],
                     double octree_res);
void ko_coll_free(ko_coll *c);
void ko_coll_set_resolution(ko_coll *c, double res); /* :70-75 */
void ko_coll_update_state(ko_coll *c, double x, double y, double yaw); /* :125-147 */
/* collision_check.h:91-136; returns 0 ok, <0 unsupported (non-planar sensor) */
int ko_coll_update_scan(ko_coll *c, const double *ranges, const double *angles,
                        size_t n);
int ko_coll_update_points(ko_coll *c, const float *xyz, size_t n,
                          int global_frame);
int ko_coll_check(ko_coll *c); /* :149-162 at the state last set */
int ko_coll_check_at(ko_coll *c, double x, double y, double yaw); /* :225-246 */
float ko_coll_radius(const ko_coll *c);
size_t ko_coll_num_voxels(const ko_coll *c);

/* ---- A2/A3: roll-out ------------------------------------------------------ */
/* trajectory_sampler.cpp:118-179 for every (vx,vy,omega) in order,
 * drop_samples = true.  Outputs are the compacted sample-major matrices of
 * trajectory.h:326-503: paths_x/y [cap x P], vel_* [cap x (P-1)] (may be
 * NULL), raw_index[cap] = index into the input list of each admissible row.
 * Returns the number of admissible samples.  coll may be NULL (no obstacles).*/
long ko_rollout(ko_coll *coll, const ko_state *start, double time_step,
                size_t P, const double *vx, const double *vy,
                const double *omega, size_t n, float *paths_x, float *paths_y,
                float *vel_vx, float *vel_vy, float *vel_omega,
                int32_t *raw_index);
/* the same with both values of drop_samples_ (trajectory_sampler.cpp:157-168: a colliding sample whose
 * last free index lies beyond num_ctrl_points is frozen at that point with zero velocities and stays
 * admissible); vel_* [Na x (P - 1)] receive the velocity profiles */
long ko_rollout_mode(ko_coll *coll, const ko_state *start, double time_step, size_t P,
                     const double *vx, const double *vy, const double *omega, size_t n,
                     int drop_samples, size_t num_ctrl_points, float *paths_x, float *paths_y,
                     float *vel_vx, float *vel_vy, float *vel_omega, int32_t *raw_index);


/* ---- A5-A10: cost evaluator ---------------------------------------------- */
typedef struct {
  const float *seg_x, *seg_y, *seg_z; /* tracked segment View (path.h:39-91) */
  size_t seg_size;
  size_t seg_start_idx;     /* View::getStartIndex */
  const float *path_acc;    /* parent accumulated_path_length_ */
  size_t path_acc_size;
  float ref_path_length;    /* reference_path->totalPathLength() */
  const float *obs_x, *obs_y; /* world-frame obstacle points (setPointScan) */
  size_t n_obs;
  float max_obstacles_dist; /* cost_evaluator.h:179 */
  float acc_limits[3];      /* cost_evaluator.cpp:18-20 */
  ko_weights w;
} ko_cost_ctx;

/* cost_evaluator.cpp:49-109.  costs_out[N] (may be NULL) receives every
 * sample's total; returns argmin index or -1 when nothing beats FLT_MAX.
 * vel_* may be NULL => treated as constant velocities (smooth = jerk = 0). */
long ko_min_trajectory_cost(const ko_cost_ctx *cx, const float *paths_x,
                            const float *paths_y, const float *vel_vx,
                            const float *vel_vy, const float *vel_omega,
                            size_t N, size_t P, size_t row_stride_path,
                            size_t row_stride_vel, float *costs_out,
                            float *min_cost_out);
/* individual terms, cost_evaluator.cpp:111-233 */
float ko_path_cost(const ko_cost_ctx *cx, const float *px, const float *py,
                   size_t P);
float ko_goal_cost(const ko_cost_ctx *cx, const float *px, const float *py,
                   size_t P);
float ko_obstacle_cost(const ko_cost_ctx *cx, const float *px, const float *py,
                       size_t P);
float ko_smoothness_cost(const ko_cost_ctx *cx, const float *vx,
                         const float *vy, const float *om, size_t nv);
float ko_jerk_cost(const ko_cost_ctx *cx, const float *vx, const float *vy,
                   const float *om, size_t nv);
float ko_segment_length(const float *x, const float *y, const float *z,
                        size_t n); /* path.h:85-91 */

/* cost_evaluator.h:174-223: obstacle points -> "world" via
 * sensor_tf_body * body_tf_world (float Eigen isometries restated). */
void ko_obstacles_from_scan(const float sensor_pos[3],
                            const float sensor_rot_xyzw[4],
                            const ko_state *state, const double *ranges,
                            const double *angles, size_t n, float *ox,
                            float *oy);
void ko_obstacles_from_points(const float sensor_pos[3],
                              const float sensor_rot_xyzw[4],
                              const ko_state *state, const float *xyz,
                              size_t n, float *ox, float *oy);

/* ---- A11: DWA controller (controllers/{controller,follower,dwa}) --------- */
typedef struct {
  ko_limits limits;
  int ctr_type;
  double time_step, prediction_horizon, control_horizon;
  int max_linear_samples, max_angular_samples;
  int shape;
  float dims[3];
  int ndims;
  float sensor_pos[3];
  float sensor_rot_xyzw[4];
  double octree_res;
  ko_weights weights;
} ko_dwa_config;

typedef struct {
  int found;
  float cost;
  long index;          /* compacted (admissible-only) index, -1 if none */
  long raw_index;      /* index into the generated velocity list */
  long n_generated;
  long n_admissible;
  size_t P;            /* numPointsPerTrajectory used this cycle */
  size_t seg_start, seg_size; /* tracked segment used */
} ko_dwa_result;

typedef struct ko_dwa ko_dwa;
ko_dwa *ko_dwa_new(const ko_dwa_config *cfg); /* dwa.cpp:14-41,93-116 */
void ko_dwa_free(ko_dwa *d);
/* follower.cpp:80-105 (interpolate=true, LINEAR) */
int ko_dwa_set_path(ko_dwa *d, const float *x, const float *y, const float *z,
                    size_t n);
void ko_dwa_set_state(ko_dwa *d, double x, double y, double yaw, double speed);
int ko_dwa_is_goal_reached(ko_dwa *d); /* follower.cpp:109-142 */
void ko_dwa_set_max_range(ko_dwa *d, float r); /* dwa.cpp:143-145 */
/* dwa.h:183-230.  sensor: either scan (ranges/angles) or points (xyz). */
int ko_dwa_compute_scan(ko_dwa *d, double vx, double vy, double omega,
                        const double *ranges, const double *angles, size_t n,
                        ko_dwa_result *res);
int ko_dwa_compute_points(ko_dwa *d, double vx, double vy, double omega,
                          const float *xyz, size_t n, ko_dwa_result *res);
/* winner row of the last compute (trajectory.h:556-562) */
const float *ko_dwa_best_path_x(const ko_dwa *d);
const float *ko_dwa_best_path_y(const ko_dwa *d);
const float *ko_dwa_best_vel(const ko_dwa *d, int comp /*0 vx,1 vy,2 omega*/);
/* all admissible samples + costs of the last compute (debugging/parity) */
const float *ko_dwa_samples_x(const ko_dwa *d);
const float *ko_dwa_samples_y(const ko_dwa *d);
const float *ko_dwa_costs(const ko_dwa *d);
const int32_t *ko_dwa_raw_index(const ko_dwa *d);
const ko_path *ko_dwa_path(const ko_dwa *d);
size_t ko_dwa_max_segment_size(const ko_dwa *d);
size_t ko_dwa_closest_index(const ko_dwa *d);

/* ---- M1/M2: LocalMapper (CPU semantics) ---------------------------------- */
/* local_mapper.h:14-56,198-222; local_mapper.cpp:127-159,204-220;
 * line_drawing.h:55-124.  grid_out is column-major int32 [H x W]
 * (Eigen::MatrixXi), i.e. cell (i,j) at i + j*H. */
int ko_mapper_scan_to_grid(int grid_height, int grid_width, float resolution,
                           const float laserscan_position[3],
                           float laserscan_orientation, const double *angles,
                           const double *ranges, size_t n, int32_t *grid_out);

/* ---- M3: Bayesian update + previous-grid warp (CPU semantics) ------------- */
/* local_mapper.h:58-103 (ctor), local_mapper.cpp:17-78 (warp), :106-125 (cell
 * probability), :161-202,222-241 (scan).  PARITY UNPINNED: the reference's
 * tests only print these grids (mapper_test.cpp:136-220); the restatement
 * follows the source expression by expression, including Eigen's integer
 * Vector2i::norm() (double sqrt truncated to int) and its closed-form 3x3
 * inverse.  Grids are column-major float/int32 [H x W]. */
typedef struct ko_bmap ko_bmap;
ko_bmap *ko_bmap_create(int grid_height, int grid_width, float resolution,
                        const float laserscan_position[3], float laserscan_orientation,
                        float p_prior, float p_occupied, float p_empty, float range_sure,
                        float range_max, float wall_size);
void ko_bmap_destroy(ko_bmap *b);
int ko_bmap_scan(ko_bmap *b, const double *angles, const double *ranges, size_t n,
                 int32_t *grid_out, float *prob_out);
int ko_bmap_warp(ko_bmap *b, const float current_position_in_previous_pose[2],
                 double current_orientation_in_previous_pose);
void ko_bmap_warp_matrix(const ko_bmap *b, const float pos[2], double orient, float inv[3][3]);
const float *ko_bmap_previous(const ko_bmap *b);
/* not in the reference (previousGridDataProb is only ever warped): lets a test
 * start from a non-constant previous grid */
void ko_bmap_set_previous(ko_bmap *b, const float *prob);

/* ---- M5: raw point cloud -> laserscan (CPU semantics) ----------------------- */
/* utils/pointcloud.h:116-177 (angle_step overload: angle_step > 0, *num_bins is
 * an output = ceil(2 pi / angle_step), angles_out[i] = i * angle_step) and
 * :205-259 (num_bins overload: angle_step <= 0, *num_bins is the input,
 * angles_out may be NULL).  Points are float32 triples at byte offsets inside
 * `point_step`-byte records; the column loop runs over row_step in steps of
 * point_step (`width` is not used by the reference either).  Returns the number
 * of bins, or -1 when cap is too small / arguments are invalid.  Non-finite x
 * or y make the reference index out of bounds (int(NaN)); here they are
 * skipped. */
long ko_pointcloud_to_laserscan(const int8_t *data, size_t nbytes, int point_step,
                                int row_step, int height, int width, int x_offset,
                                int y_offset, int z_offset, double max_range,
                                double min_z, double max_z, double angle_step,
                                int num_bins, double *ranges_out,
                                double *angles_out, size_t cap);
/* ... with the x / y / z fields of any PointCloud2 datatype (utils/pointcloud.h:37-87 ids 1-8) */
long ko_pointcloud_to_laserscan_typed(const int8_t *data, size_t nbytes, int point_step,
                                      int row_step, int height, int width, int x_offset,
                                      int y_offset, int z_offset, double max_range,
                                      double min_z, double max_z, double angle_step,
                                      int num_bins, int field_type, double *ranges_out,
                                      double *angles_out, size_t cap);


/* ---- CriticalZoneChecker (CPU semantics), SURVEY 8f rank 2 ------------------- */
/* utils/critical_zone_check.{h,cpp}: ctor :13-58 (shape -> radius, sensor
 * transform from the (x, y, z, w) rotation 4-vector, half cone angle), preset
 * :60-83 (forward / backward index sets), check(ranges) :85-117, check(cloud)
 * :119-131 (num_bins overload of pointCloudToLaserScanFromRaw with range_max /
 * min / max height, then check(ranges)).  shape: 0 cylinder, 1 box, 2 sphere. */
typedef struct ko_czc ko_czc;
ko_czc *ko_czc_create(int shape, const float *dims, const float sensor_pos[3],
                      const float sensor_rot_xyzw[4], float critical_angle_deg,
                      float critical_distance, float slowdown_distance,
                      const double *angles, size_t n, float min_height,
                      float max_height, float range_max);
void ko_czc_destroy(ko_czc *z);
float ko_czc_check(const ko_czc *z, const double *ranges, int forward);
float ko_czc_check_cloud(const ko_czc *z, const int8_t *data, size_t nbytes,
                         int point_step, int row_step, int height, int width,
                         int x_offset, int y_offset, int z_offset, int forward);
void ko_czc_set_field_type(ko_czc *z, int field_type);

/* introspection for the tests: index sets and trig tables */
size_t ko_czc_indices(const ko_czc *z, int forward, size_t *out, size_t cap);

/* bounded multi-thread CPU baseline helper: roll-out + costs with `threads`
 * workers over contiguous sample blocks (mirrors the reference ThreadPool
 * scheme, trajectory_sampler.cpp:192-205) -- used only by bench.py. */
long ko_baseline_cycle(ko_coll *coll, const ko_cost_ctx *cx,
                       const ko_state *start, double time_step, size_t P,
                       const double *vx, const double *vy, const double *omega,
                       size_t n, int threads, float *min_cost_out,
                       long *n_admissible_out);

/* full-size parity helpers (tests only): per-sample outputs in raw numbering,
 * samples evaluated independently by `threads` workers; same per-sample
 * arithmetic as ko_rollout + ko_min_trajectory_cost.  Returns the admissible
 * count. */
long ko_full_cycle(ko_coll *coll, const ko_cost_ctx *cx, const ko_state *start,
                   double time_step, size_t P, const double *vx,
                   const double *vy, const double *omega, size_t n, int threads,
                   float *px, float *py, uint8_t *admissible, float *costs);
/* ... with velocity profiles kept and both values of drop_samples_ (see ko_rollout_mode) */
long ko_full_cycle_mode(ko_coll *coll, const ko_cost_ctx *cx, const ko_state *start, double dt, size_t P,
                        const double *vx, const double *vy, const double *om, size_t n, int threads,
                        int drop_samples, size_t num_ctrl_points, float *px, float *py, float *fvx,
                        float *fvy, float *fom, uint8_t *adm, float *costs);

void ko_costs_mt(const ko_cost_ctx *cx, const float *px, const float *py,
                 const float *vx, const float *vy, const float *om, size_t n,
                 size_t P, int threads, float *costs);

#ifdef __cplusplus
}
#endif
#endif
