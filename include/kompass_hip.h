/*
 * kompass_hip.h -- C ABI of libkompass_hip.so: the MI355X (gfx950) hot path of
 * kompass_cpp's sampling controller (DWA sampler -> roll-out -> collision ->
 * weighted cost -> argmin) and of the laserscan -> occupancy LocalMapper.
 *
 * The reference (automatika-robotics/kompass-core v0.8.1) has no C ABI: its
 * boundary is a C++ class surface + a nanobind module.
 * This header is the thin shim the host C++ classes
 * (kompass-core_amd/csrc/host/) and any other binding call into.  Every entry
 * point names the reference interface it replaces; paths are relative to
 * <reference>/src/kompass_cpp/kompass_cpp/.
 *
 * Conventions
 *  - plain C types only: pointers + sizes, caller-owned host buffers, opaque
 *    contexts; no C++/torch types.
 *  - every call returns KC_OK (0) or a negative kc_status; the message of the
 *    last failure on the calling thread is kc_last_error().  A HIP failure
 *    never crosses the boundary as UB.  There is NO CPU fallback: without a
 *    usable HIP device every compute call returns KC_ERR_HIP.
 *  - one HIP stream per context; calls on one context are serial, distinct
 *    contexts may be driven from distinct threads (reference objects are not
 *    thread-safe either: trajectory_sampler.cpp:18, local_mapper.cpp:15).
 *  - results are bit-exact on integers/indices and on float costs w.r.t. the
 *    reference CPU path with maxNumThreads = 1 ordering.
 */
#ifndef KOMPASS_HIP_H
#define KOMPASS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KC_ABI_VERSION 1

typedef enum {
  KC_OK = 0,
  KC_ERR_INVALID = -1,     /* bad argument (reference: std::invalid_argument) */
  KC_ERR_RANGE = -2,       /* capacity / range (reference: std::out_of_range) */
  KC_ERR_HIP = -3,         /* HIP runtime failure or no device */
  KC_ERR_UNSUPPORTED = -4, /* outside the restated domain (e.g. non-planar
                              sensor rotation for the collision checker) */
  KC_ERR_STATE = -5        /* call order (e.g. evaluate before roll-out) */
} kc_status;

/* datatypes/control.h:12 */
enum { KC_ACKERMANN = 0, KC_DIFFERENTIAL_DRIVE = 1, KC_OMNI = 2 };
/* utils/collision_check.h:25 */
enum { KC_CYLINDER = 0, KC_BOX = 1, KC_SPHERE = 2 };
/* mapping/local_mapper.h:9 */
enum { KC_UNEXPLORED = -1, KC_EMPTY = 0, KC_OCCUPIED = 100 };

const char *kc_last_error(void);
int kc_abi_version(void);
/* number of visible HIP devices (0 when none; never fails) */
int kc_device_count(void);

/* ------------------------------------------------------------------------ */
/* plain structs                                                            */
/* ------------------------------------------------------------------------ */
typedef struct { /* Path::State, datatypes/path.h:14-22 */
  double x, y, yaw, speed;
} kc_state;

typedef struct { /* ControlLimitsParams, datatypes/control.h:191-235 */
  double vx_max, vx_acc, vx_dec;
  double vy_max, vy_acc, vy_dec;
  double omega_max_angle, omega_max, omega_acc, omega_dec;
} kc_limits;

typedef struct { /* TrajectoryCostsWeights, utils/cost_evaluator.h:22-50 */
  double reference_path_distance_weight;
  double goal_distance_weight;
  double obstacles_distance_weight;
  double smoothness_weight;
  double jerk_weight;
} kc_weights;

typedef struct {
  /* CollisionChecker ctor, utils/collision_check.h:46-50 */
  int shape;                /* KC_CYLINDER / KC_BOX / KC_SPHERE */
  float dims[3];            /* cylinder (r,h) / box (x,y,z) / sphere (r) */
  int ndims;
  float sensor_pos[3];      /* sensor position in the body frame */
  float sensor_rot_xyzw[4]; /* Eigen coefficient order (x,y,z,w) */
  double octree_res;
  /* TrajectorySampler / CostEvaluator sizing, trajectory_sampler.h:75-83,
   * cost_evaluator.h:69-71 (device buffers sized once, like
   * cost_evaluator_gpu.cpp:54-120; segment / obstacle buffers grow) */
  double time_step;
  size_t max_samples;       /* capacity N (numTrajectories) */
  size_t max_points;        /* capacity P (numPointsPerTrajectory) */
  size_t max_segment;       /* initial tracked-segment capacity */
  size_t max_obstacles;     /* initial obstacle capacity */
  float acc_limits[3];      /* cost_evaluator.cpp:18-20 */
  int device;               /* HIP device ordinal */
} kc_dwa_params;

typedef struct {
  /* TrajSearchResult (trajectory.h:611-618) + LowestCost (:621-644) */
  int found;           /* isTrajFound */
  float cost;          /* trajCost (minCost) */
  int64_t index;       /* index into the admissible-only list (reference
                          numbering); -1 if not found */
  int64_t raw_index;   /* index into the generated sample list (global, i.e.
                          including the shard offset); -1 if not found */
  int64_t n_admissible;/* samples->size() on this context's shard (kc_dwa_cycle_sharded:
                          over all ranks) */
  int64_t n_samples;   /* samples rolled out on this context's shard
                          (kc_dwa_cycle_sharded: over all ranks) */
} kc_result;

/* ------------------------------------------------------------------------ */
/* sampling controller                                                      */
/* ------------------------------------------------------------------------ */
typedef struct kc_dwa kc_dwa;

/* TrajectorySampler::TrajectorySampler (trajectory_sampler.cpp:23-60) +
 * CostEvaluator::CostEvaluator (cost_evaluator.cpp:23-37) */
int kc_dwa_create(const kc_dwa_params *params, kc_dwa **out);
void kc_dwa_destroy(kc_dwa *ctx);

/* adopt an external hipStream_t for all work of this context; NULL restores the
 * context's own (non-blocking) stream.  NB: the handle of the legacy default
 * stream IS NULL -- to share a stream with a framework create an explicit one
 * there (torch.cuda.Stream()) and pass its handle. */
int kc_dwa_set_stream(kc_dwa *ctx, void *hip_stream);
/* CollisionChecker::resetOctreeResolution, collision_check.cpp:70-75 */
int kc_dwa_set_resolution(kc_dwa *ctx, double octree_res);
/* CostEvaluator::updateCostWeights, cost_evaluator.cpp:39-41 */
int kc_dwa_set_weights(kc_dwa *ctx, const kc_weights *w);

/* Per-context switches of the device path (no counterpart in the reference, whose
 * SYCL path has none; every setting gives bit-identical results -- tests/
 * test_gpu_parity.py runs the cycle under each).  value: 0 / 1 unless stated.
 *   "fused_cycle"    (1) kc_dwa_cycle runs the whole cycle as ONE launch when the
 *                        cost tables fit in LDS beside the roll-out tile, the shard is one
 *                        resident wave of workgroups (<= 32 samples x CUs) and either fills
 *                        half the CUs or left few survivors last cycle; 2: whenever the
 *                        tables fit; 0: roll-out, cost and publish kernels
 *   "cycle_samples"  (0) samples per workgroup of the single-launch cycle: 0 = 32, or 16 when
 *                        32 would give at most half of the CUs a workgroup (shards <= 4096
 *                        samples on an MI355X); 16 / 32: fixed
 *   "velocity_group" (0) kc_cost_evaluate with velocity profiles: samples per wavefront of the
 *                        ordered smoothness / jerk sums -- 1: inside the cost kernel, 4 / 16: a
 *                        pass of its own (velocity_sums_kernel); 0: by batch size (1 below
 *                        ~5 profiles per SIMD, 16 from ~96)
 *   "velocity_beside" (1) that pass on a second stream beside the wavefront-per-sample cost kernel
 *                        (its serial chains leave most issue slots idle); a short kernel behind both
 *                        adds the two terms and forms the keys.  0: one after the other
 *   "near_table"   (128) cells per side (16..512) of the near table of the tracked segment
 *                        (per cell of a grid over the reachable box: the chunk range that can
 *                        hold a point's nearest segment point + a seed), built when the
 *                        wavefront-per-sample cost search is expected to run; 0: off (the
 *                        chunk hierarchy alone).  Same minimum, same index, same bits
 *   "host_reduce"    (1) single-GPU single-launch cycles end without a device-side reduction:
 *                        every workgroup posts a 32-byte slot to pinned memory, the host
 *                        reduces them in kc_dwa_fetch_result; 0: arrival ticket + last
 *                        workgroup (what kc_dwa_cycle_sharded always uses)
 *   "write_paths"    (0) the single-launch cycle also stores every float row
 *                        (otherwise rows are produced on demand by kc_dwa_get_samples)
 *   "cost_kernel"    (0) stand-alone cost stage: 0 = 2 (round 4: the wavefront-per-sample kernel is ahead at
 *                        every list length), 1 workgroup per sample, 2 wavefront per sample
 *   "drop_samples"   (1) TrajectorySampler::setSampleDroppingMode (trajectory_sampler.cpp:103-105).  0: a
 *                        sample that collides at loop step i with last_free_index = i - 1 beyond
 *                        "num_ctrl_points" is KEPT (:157-168): path points i + 1 .. P - 1 repeat point
 *                        i - 1, velocities i .. P - 2 are zero, and smoothness / jerk see that step
 *   "num_ctrl_points" (0) numCtrlPoints_ = control_horizon / time_step as size_t (:88: the config-object
 *                        constructor's definition; the explicit-argument constructor leaves it
 *                        uninitialised)
 *   "obs_near"       (1) LaserScan input (beams without a return -- inf / NaN -- stay out of the boxes): consecutive beams are a polyline -- the
 *                        obstacle term of long admissible lists goes through a near table of the scan
 *                        (per cell of the reachable box: the beam chunks that can hold the nearest
 *                        obstacle, a seed, a floor; obs_near_kernel) instead of the bucket ring search
 *                        From the second cycle on kc_dwa_set_scan builds the table for the pose it
 *                        is given inside the launch of the sensor tables; a cycle that starts
 *                        elsewhere (or reaches further) builds its own
 *   "obs_union"     (512) obstacle term of long admissible lists, point-cloud / costmap input: where a
 *                        trajectory runs through occupied bucket cells, the obstacles of the ONE
 *                        rectangle of cells that can hold the trajectory's nearest obstacle are
 *                        broadcast to all of its points (obstacle_union_scan) instead of a ring walk
 *                        per point; rectangles with more obstacles than this fall back.  0: off
 *   "box_cover"      (1) BOX robots at least twice as long as wide whose inscribed and circumscribed circles are
 *                        4.5 voxels or more apart: the pose gate of the fused kernels looks up several circles
 *                        laid along the long axis in the dilated sensor masks (2 .. 8 of them) instead of one
 *                        around the centre; 0: the single look-up.  Takes effect with the next sensor update
 *   "cost_batch"     (1) the long-list cost kernel leaves the per-sample part (ordered sum, the end
 *                        point's index, weighted total, key) to a pass over 64 samples at once, a lane a
 *                        sample (sample_cost_batched_kernel), when the list fills several buffers per
 *                        workgroup (>= 10240 expected); 2: for every list length; 0: off
 *   "device_trig"    (1) cos / sin(yaw_k) formed by the roll-out kernels themselves (below:
 *                        kc_trig_selfcheck); 0: the FALLBACK -- the host's libm table, complete before the
 *                        launch (no kernel ever waits for the host)
 *   "sensor_on_host" (0) voxel bitmap / obstacle buckets built on the host (spheres beyond 32 k points or 32 z layers are)
 *   "sensor_two_launch" (0) the sensor build of clouds beyond 32 k points (two launches) for every size
 *   "force_split"    (0) roll-out, collision and compaction as separate kernels
 *   "team_max"       (4) workgroups of the single-launch cycle with up to this many survivors cost them by teams
 *                        (halves / quarters of the workgroup, eight / four lanes a trajectory point); more: a
 *                        wavefront a sample.  0: always a wavefront a sample
 * kc_dwa_get_option also reads "last_cycle_single_launch", "last_cycle_samples",
 * "host_threads", "trig_rows" (rows of the host's cos / sin table: distinct omegas of
 * this context's share), "shard_samples", and the counters "obs_near_rides" /
 * "obs_near_builds" (near tables built inside a sensor launch / by a launch of their own),
 * "pattern_hits" / "pattern_builds" (index patterns of a window lattice found on the device /
 * walking orders built for a new one).
 * Waits for the context's stream.  Nothing that selects a path is read from the environment (round 4); the
 * variables that remain are diagnostics (Switches). */
int kc_dwa_set_option(kc_dwa *ctx, const char *name, double value);
int kc_dwa_get_option(kc_dwa *ctx, const char *name, double *value);
/* threads of the process-wide host pool that evaluates the libm trig table of a
 * roll-out (path.h:24-30 calls cos/sin per step; here once per distinct omega and
 * step): 1..64, the calling thread included.  Default: from the CPUs the process
 * may use (cgroup quota / ranks on the node) or KC_HOST_THREADS. */
int kc_set_host_threads(int n);
/* Device trig (option "device_trig", default 1; csrc/kc_trig_exact.h): the roll-out kernels form
 * yaw_k of their omega rows by repeated addition and evaluate glibc's `sincos` algorithm themselves
 * (State::update, datatypes/path.h:24-30, calls cos / sin of the host libm per step; gcc folds the
 * pair into `sincos`) -- every operation a correctly rounded IEEE double add / multiply in the
 * library's order over its 440-entry table, so the bits are the host's.
 * kc_trig_selfcheck compares the restatement with the INSTALLED sincos on a fixed argument set
 * (run once when the first context asks; a difference switches device trig off for the process and
 * the host table of rounds 1-3 is used); kc_trig_table fills cos_sin_out[k * n_rows + r] =
 * {cos, sin}(yaw_k of row r), yaw_0 = yaw0, yaw_{k+1} = yaw_k + omega[r] * dt, on the current
 * device (tests: against the host's sincos). */
int kc_trig_selfcheck(int64_t *compared_out);
int kc_trig_table(double yaw0, const double *omega, size_t n_rows, size_t n_steps, double dt,
                  double *cos_sin_out);

/* A1 on the host: TrajectorySampler::UpdateReachableVelocityRange
 * (trajectory_sampler.cpp:328-372) + the lattice loops (:181-220 / :256-272,
 * maxNumThreads = 1 ordering) with the all-zero filter of :122-125.
 * max_angular_samples is the un-bumped constructor value.  The list becomes
 * the context's sample set (uploaded); returns the count in *n_out.  Optional
 * vx/vy/omega receive the list (capacity cap). */
int kc_dwa_sample_window(kc_dwa *ctx, int ctr_type, const kc_limits *limits,
                         double cur_vx, double cur_vy, double cur_omega,
                         int max_linear_samples, int max_angular_samples,
                         size_t *n_out, double *vx, double *vy, double *omega,
                         size_t cap);
/* explicit sample list in generation order (what getAdmissibleTrajsFromVel is
 * called with, trajectory_sampler.cpp:213,260,268) */
int kc_dwa_set_samples(kc_dwa *ctx, size_t n, const double *vx,
                       const double *vy, const double *omega);
/* multi-GPU sharding: this context rolls out only samples [first, first+count)
 * of the list; raw indices and the packed key carry the global index.  Clears a
 * rule set with kc_dwa_set_shard_rule. */
int kc_dwa_set_shard(kc_dwa *ctx, size_t first, size_t count);
/* Sharding by RULE (what kc_dwa_cycle_sharded needs with more than one rank: every
 * rank must know every rank's share to turn the exchanged bitmaps into the
 * reference's admissible-only index).  Every rank passes the FULL list to
 * kc_dwa_sample_window / kc_dwa_set_samples (the reference generates the whole
 * lattice on the host too, trajectory_sampler.cpp:181-275); the context keeps its
 * share.  Applies to the current list and to every later one.
 *   KC_SHARD_BLOCKS  rank r owns the contiguous block [n r / W, n (r + 1) / W)
 *   KC_SHARD_ROWS    samples are dealt by trig row (distinct omega): row k (in order
 *                    of first appearance) -> rank k mod W, a row with more than twice
 *                    the mean number of samples (omni: the omega = 0 row) sample by
 *                    sample.  A rank then evaluates 1 / W of the host's cos / sin
 *                    table (path.h:24-30: one libm pair per step and omega) instead
 *                    of all of it, and uploads only its own samples.
 * Raw indices in results, keys and kc_dwa_get_samples stay GLOBAL (positions in the
 * full list = the reference's generation order) under either rule.
 * mode < 0 clears the rule (the context owns the whole list again). */
enum { KC_SHARD_BLOCKS = 0, KC_SHARD_ROWS = 1 };
int kc_dwa_set_shard_rule(kc_dwa *ctx, int rank, int world, int mode);
/* the deal itself, as a pure host function (no device): rows[i] = trig row of
 * sample i (any labelling of the distinct omegas), owner_out[i] = rank that owns it */
int kc_shard_plan(const int32_t *rows, size_t n, int world, int mode, int32_t *owner_out);
/* the reduced exchange record of a sharded cycle -> result (pure host function;
 * layout in kc_shard.h / DESIGN.md section 8: [key, error, world x words_per_rank
 * bitmap words]).  owner: as from kc_shard_plan (NULL with KC_SHARD_BLOCKS).
 * Returns KC_ERR_HIP when the record carries a rank's error word. */
int kc_shard_merge(const int64_t *record, size_t words_per_rank, int world, int mode,
                   const int32_t *owner, size_t n_total, kc_result *out);
/* 1 when sample global_raw_index of the current list belongs to this context's
 * share (its row can be fetched here), else 0 */
int kc_dwa_owns_sample(kc_dwa *ctx, int64_t global_raw_index, int *owned_out);

/* sensor data for BOTH consumers, once per cycle:
 *  CollisionChecker::updateState + updateSensorData<T> (collision_check.cpp:
 *  125-135, collision_check.h:91-136) and CostEvaluator::setPointScan
 *  (cost_evaluator.h:174-223; maxObstaclesDist = max_sensor_range / 3). */
int kc_dwa_set_scan(kc_dwa *ctx, const kc_state *state, const double *ranges,
                    const double *angles, size_t n, float max_sensor_range);
int kc_dwa_set_points(kc_dwa *ctx, const kc_state *state, const float *xyz,
                      size_t n, float max_sensor_range);
/* the same with the list in the SENSOR frame -- updateSensorData(cloud, global_frame = false),
 * collision_check.h:119-131: the octree frame is body->tf * sensor_tf_body (as for a laser scan), the obstacle
 * list of the cost term is the same (setPointScan(cloud), cost_evaluator.h:205-223).  Mounts that are not a
 * rotation about z: KC_ERR_UNSUPPORTED for point lists (laser scans take any mount). */
int kc_dwa_set_points_sensor_frame(kc_dwa *ctx, const kc_state *state, const float *xyz, size_t n,
                                   float max_sensor_range);

/* SURVEY 8f rank 4 -- occupancy grid -> obstacle set without the host round
 * trip (the reference goes grid -> host -> point list -> DWA, control/dwa.py:
 * 298-299 with local_map).  dev_grid: int32, column-major [H x W] (the
 * LocalMapper layout, kc_mapper_grid_device); every OCCUPIED (100) cell (i,j)
 * becomes the point ((i - c0) res, (j - c1) res, 0), c0/c1 = the mapper's
 * central cell (local_mapper.h:26-27), i.e. the inverse of localToGrid
 * (:210-222).  State afterwards == kc_dwa_set_points with that list.  The grid
 * must be complete in stream order of the controller's stream, or finished. */
int kc_dwa_set_grid_device(kc_dwa *ctx, const kc_state *state, const int32_t *dev_grid,
                           int grid_height, int grid_width, float resolution, int central_i,
                           int central_j, float max_sensor_range);
/* same, from a mapper context of this library: geometry and stream ordering
 * (event wait, no host synchronisation) are taken from it */
struct kc_mapper;
int kc_dwa_set_grid_from_mapper(kc_dwa *ctx, const kc_state *state, struct kc_mapper *mapper,
                                float max_sensor_range);

/* the (reference_path, tracked_segment) arguments of getMinTrajectoryCost
 * (cost_evaluator.h:139-142): segment points (Path::View X/Y/Z, path.h:39-76),
 * acc_at_seg[j] = reference_path->getDistanceAtIndex(seg_start + j)
 * (path.h:190-194), ref_path_length = reference_path->totalPathLength(). */
int kc_dwa_set_tracked_segment(kc_dwa *ctx, const float *x, const float *y,
                               const float *z, const float *acc_at_seg,
                               size_t seg_size, float ref_path_length);
/* the same segment from ONE array of points, xyz[j] = {x, y, z} of segment point j (a std::vector<Path::Point>
 * / an (S, 3) float32 array where it lies, datatypes/path.h:14-21): the library de-interleaves while it fills
 * its rows, the caller makes no column copies (they were 6 us of a 12 us call through the Python binding). */
int kc_dwa_set_tracked_segment_xyz(kc_dwa *ctx, const float *xyz, const float *acc_at_seg,
                                   size_t seg_size, float ref_path_length);

/* SURVEY 8f rank 4, second half -- the reference path resident on the device.
 * kc_dwa_set_path: the whole (interpolated) path once per path: points, the
 * accumulated length at every point (Path::getDistanceAtIndex, path.h:74-76)
 * and the total length.  kc_dwa_set_tracked_window(start, size): the tracked
 * segment = points [start, start + size) of that path; same state as
 * kc_dwa_set_tracked_segment(x + start, y + start, z + start, acc + start,
 * size, total_length), with the search tables built by a kernel, stream-ordered
 * between two cycles (no host synchronisation). */
int kc_dwa_set_path(kc_dwa *ctx, const float *x, const float *y, const float *z,
                    const float *acc_length, size_t n, float total_length);
int kc_dwa_set_tracked_window(kc_dwa *ctx, size_t start, size_t size);

/* A2-A4: TrajectorySampler::generateTrajectories (trajectory_sampler.cpp:
 * 295-314 -> :118-179) for the context's samples; option "drop_samples" selects the mode.
 * num_points = numPointsPerTrajectory of this cycle (<= max_points). */
int kc_dwa_rollout(kc_dwa *ctx, const kc_state *start, size_t num_points);
/* CollisionChecker::checkCollisions (collision_check.cpp:149-162, 225-246) /
 * TrajectorySampler::checkStatesFeasibility (trajectory_sampler.cpp:378-407)
 * for a batch of poses against the sensor data of the last kc_dwa_set_scan /
 * kc_dwa_set_points: hit_out[i] = 1 when the robot shape at pose i touches an
 * occupied voxel. */
int kc_dwa_check_poses(kc_dwa *ctx, const double *x, const double *y,
                       const double *yaw, size_t n, uint8_t *hit_out);

/* A5-A10: CostEvaluator::getMinTrajectoryCost (cost_evaluator.cpp:49-109) on
 * the rolled-out samples; result stays on the device until fetched */
int kc_dwa_evaluate(kc_dwa *ctx);
/* blocks until the cycle finished and copies the 32-byte result record */
int kc_dwa_fetch_result(kc_dwa *ctx, kc_result *out);
/* DWA::findBestPath body after the host glue (dwa.h:215-229):
 * roll-out + evaluate + fetch */
int kc_dwa_cycle(kc_dwa *ctx, const kc_state *start, size_t num_points,
                 kc_result *out);

/* ONE reference controller cycle in one call -- DWA::findBestPath (controllers/dwa.h:183-230): a new dynamic
 * window and lattice (trajectory_sampler.cpp:328-372), this cycle's sensor data (collision_check.h:91-136,
 * cost_evaluator.h:174-223), the tracked segment (dwa.cpp:208-233), roll-out + costs + argmin (dwa.h:215-229).
 * Exactly kc_dwa_sample_window + kc_dwa_set_points | kc_dwa_set_scan + kc_dwa_set_tracked_segment[_xyz] +
 * kc_dwa_cycle in that order, same state, same results; what it saves is the caller's per-call cost (a binding
 * crossing per entry: 4 x ~1.5 us through ctypes) on the host chain in front of the cycle kernel's launch.
 * Any part with a null / zero input is skipped and keeps what the context holds (limits NULL: the current sample
 * list; no points and no scan: the current sensor data; seg_size 0: the current segment). */
typedef struct kc_step_inputs {
  /* window: kc_dwa_sample_window(ctr_type, limits, cur_*, max_*); limits NULL: skip */
  int ctr_type;
  const kc_limits *limits;
  double cur_vx, cur_vy, cur_omega;
  int max_linear_samples, max_angular_samples;
  /* sensor data: points_xyz [n_points][3] (kc_dwa_set_points), else scan_ranges / scan_angles [n_beams]
   * (kc_dwa_set_scan); both NULL: skip */
  const float *points_xyz;
  size_t n_points;
  const double *scan_ranges, *scan_angles;
  size_t n_beams;
  float max_sensor_range;
  /* tracked segment: seg_xyz [seg_size][3] (kc_dwa_set_tracked_segment_xyz), else seg_x / seg_y / seg_z
   * (kc_dwa_set_tracked_segment); seg_size 0: skip */
  const float *seg_xyz, *seg_x, *seg_y, *seg_z, *acc_at_seg;
  size_t seg_size;
  float ref_path_length;
  /* kc_dwa_cycle(state, num_points) */
  size_t num_points;
} kc_step_inputs;
int kc_dwa_find_best_path(kc_dwa *ctx, const kc_state *state, const kc_step_inputs *in, kc_result *out);

/* winner row: TrajectorySamples2D::getIndex (trajectory.h:556-562).  path_* are
 * num_points floats, vel_* are num_points-1 floats; any pointer may be NULL */
int kc_dwa_get_best(kc_dwa *ctx, float *path_x, float *path_y, float *vel_vx,
                    float *vel_vy, float *vel_omega);
/* velocity triple of sample raw_index (global numbering) of the current list */
int kc_dwa_get_sample_velocity(kc_dwa *ctx, int64_t raw_index, double *vx,
                               double *vy, double *omega);
/* compacted samples in reference order (TrajectorySamples2D, trajectory.h:
 * 506-603): paths_* [n_admissible x num_points] row-major, raw_index and costs
 * [n_admissible]; any pointer may be NULL.  cap_rows bounds the rows copied. */
int kc_dwa_get_samples(kc_dwa *ctx, float *paths_x, float *paths_y,
                       int32_t *raw_index, float *costs, size_t cap_rows,
                       size_t *n_rows_out);

/* drop_samples = 0 only: per row of kc_dwa_get_samples the first zero-velocity step of a frozen
 * sample (its velocity profile is the sample's velocity before that step and zero from it on), 0 for
 * a sample that never collided.  All zero with drop_samples = 1. */
int kc_dwa_get_freeze_steps(kc_dwa *ctx, int32_t *steps_out, size_t cap_rows, size_t *n_rows_out);

/* CostEvaluator::getMinTrajectoryCost on caller-provided trajectories
 * (sample-major host matrices exactly as TrajectorySamples2D stores them;
 * vel_* may all be NULL => constant-velocity samples).  Uses the context's
 * tracked segment, obstacles and weights.  costs_out[N] may be NULL. */
int kc_cost_evaluate(kc_dwa *ctx, const float *paths_x, const float *paths_y,
                     const float *vel_vx, const float *vel_vy,
                     const float *vel_omega, size_t n, size_t num_points,
                     float *costs_out, kc_result *out);

/* the same in two steps, for callers that score one set of trajectories again and
 * again (the reference's own benchmark does: benchmarks/benchmark_runner.cpp:
 * 152-185 times getMinTrajectoryCost on pre-generated samples): the samples stay
 * resident in HBM, kc_cost_evaluate_resident runs cost kernels + argmin only */
int kc_cost_upload(kc_dwa *ctx, const float *paths_x, const float *paths_y,
                   const float *vel_vx, const float *vel_vy, const float *vel_omega,
                   size_t n, size_t num_points);
int kc_cost_evaluate_resident(kc_dwa *ctx, float *costs_out, kc_result *out);

/* multi-GPU exchange: after kc_dwa_evaluate the context
 * holds, on the device, int64 key = (sortable(cost) << 32) | global raw index
 * (signed-comparable; INT64_MAX = nothing found) and int64 n_admissible.
 * kc_dwa_result_device() returns the device address of that int64[2] record so
 * the caller can all-reduce(min) / all-gather it on the same stream. */
int kc_dwa_result_device(kc_dwa *ctx, void **dev_int64x2);
/* after the caller has reduced that record in place (all-reduce on the
 * context's stream): queue a one-wavefront kernel that hands the record to the
 * host through pinned memory, so that kc_dwa_fetch_result returns the REDUCED
 * key without a D2H copy or a stream wait (index / n_admissible of the result
 * are the local shard's and only meaningful on the owning rank) */
int kc_dwa_publish_result(kc_dwa *ctx);
/* number of admissible samples with raw index < raw_index on this shard (used
 * to rebuild the reference's compacted index across shards) */
int kc_dwa_count_admissible_before(kc_dwa *ctx, int64_t global_raw_index,
                                   int64_t *count_out);
/* The exchange itself, inside this library (no framework in the product path): one
 * process per GPU, RCCL over xGMI.  kc_comm_unique_id on one rank, the 128 bytes
 * to every rank by any transport the caller has, kc_comm_create on all of them
 * (collective: ncclCommInitRank).  librccl is opened on first use (dlopen). */
#define KC_COMM_ID_BYTES 128
typedef struct kc_comm kc_comm;
int kc_comm_unique_id(uint8_t id_out[KC_COMM_ID_BYTES]);
int kc_comm_create(int rank, int world, const uint8_t id_in[KC_COMM_ID_BYTES], int device, kc_comm **out);
/* Rehearsal / test transport for hosts with fewer GPUs than ranks (RCCL refuses two
 * ranks on one device): the ranks are processes of ONE host that meet in a POSIX
 * shared-memory segment called `name` (unique per communicator; rank 0 creates it),
 * the exchange record is reduced by the host (D2H, min, H2D in stream order).  Same
 * kc_dwa_cycle_sharded code on both sides of the one all-reduce call; not the
 * multi-GPU product path.  KC_SHM_TIMEOUT_MS bounds every wait for a peer (20 s). */
int kc_comm_create_shm(int rank, int world, const char *name, int device, kc_comm **out);
void kc_comm_destroy(kc_comm *comm);
int kc_comm_rank(const kc_comm *comm);
int kc_comm_world(const kc_comm *comm);
enum { KC_COMM_RCCL = 0, KC_COMM_SHM = 1 };
int kc_comm_transport(const kc_comm *comm);
/* What the transport ITSELF reports (any pointer may be null): RCCL -- ncclCommCount, ncclCommUserRank,
 * ncclCommCuDevice of the communicator (kc_comm_create fails when they disagree with its arguments); shared
 * memory -- the number of ranks attached to the segment.  kc_comm_world / kc_comm_rank return the constructor's
 * arguments.  The reference has no collective (cost_evaluator_gpu.cpp:55): nothing to cite. */
int kc_comm_query(kc_comm *comm, int *n_ranks, int *user_rank, int *device);
/* after kc_dwa_evaluate: ONE ncclAllReduce(1 x int64, ncclMin) of the packed key in
 * the device record, on the context's stream, and the hand-off of the reduced
 * record to the host (kc_dwa_fetch_result then returns the GLOBAL winner: found,
 * cost, raw_index; index / n_admissible stay the shard's).  Building block for
 * callers with a protocol of their own; kc_dwa_cycle_sharded does not use it. */
int kc_dwa_allreduce_best(kc_dwa *ctx, kc_comm *comm);
/* DWA::findBestPath body of a sharded controller: this context's share rolled out
 * and scored (single launch when it fits), ONE all-reduce(int64 x (2 + world x
 * words), min) of the exchange record -- best key, error word, every rank's
 * admissible bitmap -- and the result, the same on every rank: found, cost,
 * raw_index, index (the reference's admissible-only numbering over ALL ranks) and
 * n_admissible / n_samples over all ranks.  Collective: every rank of the
 * communicator calls it with the same start / num_points, after
 * kc_dwa_set_shard_rule(rank, world, ...) (a world of one may use kc_dwa_set_shard).
 * Errors are collective too: when any rank fails (a call failed before the exchange,
 * or its device error word is set) EVERY rank returns an error for this cycle and all
 * of them have taken part in exactly one all-reduce, so the next cycle pairs up again. */
int kc_dwa_cycle_sharded(kc_dwa *ctx, kc_comm *comm, const kc_state *start, size_t num_points,
                         kc_result *out);
/* The same exchange for a cycle whose LAST cost terms the host has added -- custom cost callbacks of a sharded
 * DWA ("evaluated on host over gathered paths"; cost_evaluator.cpp:96-100: customTrajCostsPtrs_
 * after the built-in terms, total += weight * cost in float-from-double).  Every rank runs kc_dwa_cycle on its
 * share, reads its admissible rows (kc_dwa_get_samples: GLOBAL raw indices, device totals), adds the callbacks
 * in registration order and hands its own best {found, cost, raw_index} in; the record that is all-reduced is
 * kc_dwa_cycle_sharded's (this key, the error word, the admissible bitmap of the cycle just run), the result the
 * same on every rank.  status != 0: this rank failed before -- it still takes part and every rank fails the
 * cycle.  Collective. */
int kc_dwa_exchange_best(kc_dwa *ctx, kc_comm *comm, int status, int found, float cost, int64_t raw_index,
                         kc_result *out);
/* the winner's index in the reference's admissible-only numbering: admissible
 * samples in front of it on every shard, one ncclAllReduce(1 x int64, ncclSum).
 * Collective.  kc_dwa_cycle_sharded returns that index already; this remains for
 * callers of kc_dwa_allreduce_best. */
int kc_dwa_global_index(kc_dwa *ctx, kc_comm *comm, int64_t global_raw_index, int64_t *index_out);

/* decode helpers for the packed key (pure host functions) */
float kc_key_cost(int64_t key);
int64_t kc_key_index(int64_t key);
int64_t kc_key_pack(float cost, int64_t index);

/* HIP-event timing of the kernels launched by the last cycle, on the stream
 * they were launched on (bench.py roofline leg).  enable=1 records events
 * around each kernel; names/ms arrays of capacity cap, returns count. */
int kc_dwa_timing_enable(kc_dwa *ctx, int enable);
int kc_dwa_timing_get(kc_dwa *ctx, const char **names, float *ms, size_t cap,
                      size_t *count_out);

/* ------------------------------------------------------------------------ */
/* LocalMapper                                                              */
/* ------------------------------------------------------------------------ */
typedef struct kc_mapper kc_mapper;

/* LocalMapper / LocalMapperGPU ctor (mapping/local_mapper.h:14-56,
 * local_mapper_gpu.h:15-66) -- the scan -> grid subset */
int kc_mapper_create(int grid_height, int grid_width, float resolution,
                     const float laserscan_position[3],
                     float laserscan_orientation, size_t max_scan_size,
                     int device, kc_mapper **out);
void kc_mapper_destroy(kc_mapper *ctx);
int kc_mapper_set_stream(kc_mapper *ctx, void *hip_stream);
/* LocalMapper::scanToGrid (local_mapper.cpp:204-220) with the CPU semantics
 * (super-cover Bresenham, line_drawing.h:55-124); grid_out is the
 * Eigen::MatrixXi layout: int32, column-major, cell (i,j) at i + j*height.
 * Same cell values as the reference CPU mapper, bit for bit. */
int kc_mapper_scan_to_grid(kc_mapper *ctx, const double *angles,
                           const double *ranges, size_t n, int32_t *grid_out);
/* same, but the grid stays on the device (no D2H); the address of the int32
 * column-major device grid is returned by kc_mapper_grid_device.  Plain scans
 * alternate between two device grids (the scan that follows clears the other
 * one while it runs): ask for the address after every scan; the grid of a scan
 * stays intact until the NEXT scan's kernels have run. */
int kc_mapper_scan_to_grid_device(kc_mapper *ctx, const double *angles,
                                  const double *ranges, size_t n);
int kc_mapper_grid_device(kc_mapper *ctx, void **dev_grid_int32);

/* LocalMapper's second ctor (local_mapper.h:58-103): the inverse sensor model of
 * the Bayesian update.  kc_mapper_enable_bayes allocates the probability grids
 * (float, column-major like the occupancy grid) and fills the previous grid
 * with p_prior (local_mapper.h:81-83). */
typedef struct kc_bayes_params {
  float p_prior, p_occupied, p_empty, range_sure, range_max, wall_size;
} kc_bayes_params;
int kc_mapper_enable_bayes(kc_mapper *ctx, const kc_bayes_params *params);
/* LocalMapper::scanToGridBaysian (local_mapper.cpp:161-202,222-241) in the
 * reference's single-thread order: the probability of a cell is
 * updateGridCellProbability (:106-125) of the LAST beam that crosses it and of
 * the previous grid; untouched cells hold p_prior.  The occupancy grid is the
 * one scanToGrid gives.  Parity is against this build's restatement only (the
 * reference tests print these grids without asserting on them). */
int kc_mapper_scan_to_grid_bayes(kc_mapper *ctx, const double *angles, const double *ranges,
                                 size_t n, int32_t *grid_out, float *prob_out);
int kc_mapper_scan_to_grid_bayes_device(kc_mapper *ctx, const double *angles,
                                        const double *ranges, size_t n);
/* device addresses of gridDataProb and (optional) previousGridDataProb; the
 * previous grid lives in two buffers that swap on every kc_mapper_warp_previous:
 * ask again after a warp */
int kc_mapper_prob_device(kc_mapper *ctx, void **dev_prob_f32, void **dev_prev_f32);
/* LocalMapper::getPreviousGridInCurrentPose (local_mapper.cpp:17-78): bilinear
 * warp of the previous probability grid, in place (stream-ordered) */
int kc_mapper_warp_previous(kc_mapper *ctx, const float current_position_in_previous_pose[2],
                            double current_orientation_in_previous_pose);
int kc_mapper_get_previous_prob(kc_mapper *ctx, float *prob_out);
/* not in the reference, whose previous grid is only ever warped: upload a
 * previous grid (in != NULL) or feed the last scan's probabilities back on the
 * device (in == NULL) */
int kc_mapper_set_previous_prob(kc_mapper *ctx, const float *in);
int kc_mapper_sync(kc_mapper *ctx);
int kc_mapper_timing_enable(kc_mapper *ctx, int enable);
int kc_mapper_timing_get(kc_mapper *ctx, const char **names, float *ms,
                         size_t cap, size_t *count_out);

/* ------------------------------------------------------------------------ */
/* Raw point cloud -> laserscan                           */
/* ------------------------------------------------------------------------ */
typedef struct kc_cloud kc_cloud;

/* scratch for clouds of up to max_bytes bytes / max_bins angular bins (both
 * grow on demand) */
int kc_cloud_create(size_t max_bytes, size_t max_bins, int device, kc_cloud **out);
void kc_cloud_destroy(kc_cloud *ctx);
/* pointCloudToLaserScanFromRaw (utils/pointcloud.h:116-177: angle_step > 0,
 * bins = ceil(2 pi / angle_step), angles_out[i] = i * angle_step; and
 * :205-259: angle_step <= 0, num_bins bins, angles_out may be NULL).  data is
 * a PointCloud2-style byte buffer (float32 x, y, z at the given byte offsets of
 * every point_step-byte record; rows row_step bytes apart); data_on_device != 0
 * means `data` is a device address on ctx's device.  ranges_out[bin] = the
 * smallest planar distance of the points of that bin that pass the origin and z
 * filters, max_range where there is none: the same doubles as the reference's
 * CPU loop (bins from the host libm's atan2f: the device bins every point, the
 * few within 1e-6 rad of a bin edge are re-binned on the host).  Points with a
 * non-finite x or y are skipped (the reference indexes out of bounds). */
int kc_cloud_to_laserscan(kc_cloud *ctx, const int8_t *data, size_t nbytes,
                          int data_on_device, int point_step, int row_step,
                          int height, int width, int x_offset, int y_offset,
                          int z_offset, double max_range, double min_z,
                          double max_z, double angle_step, int num_bins,
                          double *ranges_out, double *angles_out, size_t cap,
                          size_t *bins_out);
/* the same for x / y / z fields of any PointCloud2 datatype (PointFieldType, utils/pointcloud.h:37-46:
 * ids 1-8), decoded as load_and_cast_val does (:49-87: byte by byte, value cast to float) -- what the
 * reference's device paths accept (local_mapper_gpu.cpp:117-140, critical_zone_check_gpu.h:36-53); the
 * bounds check uses the field's own size.  KC_FIELD_FLOAT32 is kc_cloud_to_laserscan.  The reference
 * holds no vector for the other types: parity is against this build's restatement. */
enum { KC_FIELD_INT8 = 1, KC_FIELD_UINT8 = 2, KC_FIELD_INT16 = 3, KC_FIELD_UINT16 = 4, KC_FIELD_INT32 = 5,
       KC_FIELD_UINT32 = 6, KC_FIELD_FLOAT32 = 7, KC_FIELD_FLOAT64 = 8 };
int kc_cloud_to_laserscan_typed(kc_cloud *ctx, const int8_t *data, size_t nbytes,
                                int data_on_device, int point_step, int row_step,
                                int height, int width, int x_offset, int y_offset,
                                int z_offset, int field_type, double max_range, double min_z,
                                double max_z, double angle_step, int num_bins,
                                double *ranges_out, double *angles_out, size_t cap,
                                size_t *bins_out);
/* points the last call sent back to the host for exact binning */
int kc_cloud_last_rebinned(kc_cloud *ctx, size_t *count_out);
int kc_cloud_timing_enable(kc_cloud *ctx, int enable);
int kc_cloud_timing_get(kc_cloud *ctx, const char **names, float *ms, size_t cap,
                        size_t *count_out);

/* ------------------------------------------------------------------------ */
/* CriticalZoneChecker                                    */
/* ------------------------------------------------------------------------ */
typedef struct kc_zone kc_zone;

/* CriticalZoneChecker ctor + preset (utils/critical_zone_check.cpp:13-83;
 * critical_zone_check_gpu.h ctor for the device variant).  shape / dims as in
 * kc_dwa_params; sensor_rot_xyzw is the Eigen::Vector4f of the reference
 * ((x, y, z, w), not normalised by the reference either); critical_angle in
 * degrees; angles = the scan angles the index sets are preset for.
 * KC_ERR_INVALID when slowdown_distance <= critical_distance (:52-56). */
int kc_zone_create(int shape, const float *dims, int ndims,
                   const float sensor_pos[3], const float sensor_rot_xyzw[4],
                   float critical_angle, float critical_distance,
                   float slowdown_distance, const double *angles, size_t n,
                   float min_height, float max_height, float range_max,
                   int device, kc_zone **out);
void kc_zone_destroy(kc_zone *ctx);
/* check(ranges, forward) (:85-117): 0.0 stop, (0, 1) slow-down factor, 1.0 clear;
 * ranges has one entry per preset angle */
int kc_zone_check(kc_zone *ctx, const double *ranges, size_t n, int forward,
                  float *factor_out);
/* check(raw cloud, forward) (:119-131): cloud -> ranges over the preset number
 * of bins (kc_cloud_to_laserscan semantics) -> check */
int kc_zone_check_cloud(kc_zone *ctx, const int8_t *data, size_t nbytes,
                        int point_step, int row_step, int height, int width,
                        int x_offset, int y_offset, int z_offset, int forward,
                        float *factor_out);
/* ... with the cloud_field_type of CriticalZoneCheckerGPU's constructor (critical_zone_check_gpu.h:36-53) */
int kc_zone_check_cloud_typed(kc_zone *ctx, const int8_t *data, size_t nbytes,
                              int point_step, int row_step, int height, int width,
                              int x_offset, int y_offset, int z_offset, int field_type, int forward,
                              float *factor_out);
/* the preset index sets (tests / debugging) */
int kc_zone_indices(kc_zone *ctx, int forward, int64_t *out, size_t cap,
                    size_t *count_out);

#ifdef __cplusplus
}
#endif
#endif /* KOMPASS_HIP_H */
