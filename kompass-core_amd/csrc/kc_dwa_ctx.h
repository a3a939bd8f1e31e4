#pragma once
// Sampling-controller hot path on gfx950: the host context shared by its four translation units (kc_dwa.hip,
// kc_dwa_sensor.hip, kc_dwa_cycle.hip, kc_dwa_shard.hip).  C ABI in include/kompass_hip.h; reference citations
// are relative to <reference>/src/kompass_cpp/kompass_cpp/.
//
// Non-template kernels of the kernel headers are compiled into ONE unit each: the unit that launches them
// defines its KC_TU_* macro in front of this header (templates are instantiated where they are used).
//
// Numerics contract (DESIGN.md "Exactness"): every device expression repeats
// the reference CPU expression with the same types and the same operation
// order; the file is compiled with -ffp-contract=off so no mul+add pair is
// fused, divisions and square roots use the correctly rounded forms, and the
// only transcendental inputs (cos/sin of the rolled-out yaw) are produced on
// the host by the same libm the reference calls (path.h:24-30) and handed to
// the kernel as a table -- the device never evaluates a trig function.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <unordered_map>

#include "kc_hostmath.h"
#include "kc_internal.h"
#include "kc_pool.h"
#include "kc_seg_tables.h"
#include "kc_scan_tables.h"

#include <type_traits>
#if defined(__SSE2__)
#include <emmintrin.h>
#include <immintrin.h>
#endif

#include "kc_collision_dev.h"
#include "kc_cost_kernels.h"
#include "kc_shard.h"
#include "kc_cycle_dev.h"
#include "kc_rollout_kernels.h"
#include "kc_sensor_kernels.h"
#include "kc_segment_kernels.h"
#include "kc_onear_kernels.h"
#include "kc_tilt_dev.h"

// ===========================================================================
// host context
// ===========================================================================
using namespace kc;

struct kc_dwa {
  kc_dwa_params prm{};
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  kc_weights w{1, 1, 1, 1, 1};
  Timing timing;

  // collision checker state (host)
  hm::Rigid3f sensor_tf_body;
  hm::Rigid3f frame;          // sensor_tf_world_ captured at set_scan/points
  double radius = 0, height = 0, res = 0.1;
  std::vector<int32_t> vox_kx, vox_ky;  // z-accepted occupied columns (host lists; built lazily
                                        // when the sensor update ran on the device)
  bool host_lists_valid = true;
  hm::Rigid3f obs_tf{};                 // sensor_tf_body * body of the last point update
  std::vector<float> raw_xyz;           // input of the last device-side sensor update
  bool raw_is_scan = false;             // ... laserscan points: obstacles are taken at z = 0
  std::vector<double> scan_angles;      // angles of the last laserscan + their cos/sin (a lidar's
  std::vector<double> scan_cos, scan_sin;  // angle table does not change between scans)
  std::vector<float> scan_xyz;          // sensor-frame points of the last laserscan
  DevBuf<float> d_raw;
  DevBuf<uint32_t> d_sensor_tmp;        // scratch of the multi-workgroup sensor build
  DevBuf<uint8_t> d_sensor_bytes;       // its voxel byte map (zero between updates)
  // grid hand-off (kc_dwa_set_grid_device): the point list is produced on the
  // device; the host copy is fetched only if something walks the lists
  bool raw_on_device = false;
  size_t raw_n = 0;
  DevBuf<unsigned int> d_gridcnt;
  PinBuf<long long> h_gridrec;  // {seq, count, imin, imax, jmin, jmax}
  long long grid_seq = 0;
  hipEvent_t grid_ready = nullptr;  // mapper stream -> this stream
  bool device_sensor = true;            // KC_SENSOR_HOST=1 turns the device-side update off
  long sensor_stamp_calls = 0;
  bool sensor_fused_ok = false;         // sensor_fused_kernel may take kSensorFusedLds
  bool sensor_two_launch = false;       // option: the two-launch build (clouds beyond kSensorFusedMax) for every size
  std::vector<double> vox_ddz;          // sphere: z gap per accepted voxel
  // occupancy bits of all accepted voxel columns over their bounding box
  PinBuf<uint32_t> h_gbits;
  PinBuf<uint8_t> h_gz;      // sphere: z-gap code per cell of the sensor bitmap (CollDev::gz)
  DevBuf<uint8_t> d_gz;
  PinBuf<double> h_zlut;
  DevBuf<double> d_zlut;
  bool gz_valid = false;
  size_t sphere_layers = 0;
  double sphere_ddz_max = -1.0;  // largest z gap among the accepted voxels of this sensor update (< 0: unknown)
  DevBuf<uint32_t> d_gbits;
  int gkx0 = 0, gky0 = 0, gH = 0, gwpr = 0;
  bool have_gbits = false;
  size_t lds_limit = 64 * 1024;         // dynamic LDS the fused kernel may use
  DevBuf<long long> d_block_keys;       // per-workgroup best keys of the cost kernel
  DevBuf<uint32_t> d_ginner, d_gouter;  // dilated sensor bitmaps
  bool have_dil = false;
  DevBuf<unsigned long long> d_dbg2;    // roll-out kernel stamps (diagnostic)
  DevBuf<int32_t> d_prow;               // trig rows in d_perm order (velocities are read through d_perm)
  DevBuf<int32_t> d_cprow, d_cperm;     // the same in the dealt order of the single-launch cycle
  DevBuf<uint32_t> d_pvi, d_cpvi;       // value indices in those two orders
  DevBuf<int32_t> d_perm;               // shard-local sample ids ordered by trig row
  std::vector<int32_t> h_perm;
  std::vector<int32_t> perm_scratch;    // (counting sort of build_perm)
  std::vector<int32_t> perm_prow;       // (its staging rows)
  std::vector<uint32_t> perm_pvi;
  bool perm_plain_dev = false, perm_dealt_dev = false;  // which of the two orders the device holds (perm_valid: the host does)
  std::vector<int32_t> uploaded_rows;   // trig-row pattern the orders on the device were built for
  size_t perm_first = 0, perm_count = 0;  // ... and the shard
  double inv_res = 0.0;      // 1.0 / res (octomap resolution_factor)
  bool perm_valid = false;
  bool bar_dirty = false;    // BAR stores not yet fenced
  bool update_busy = false;  // an update call queued device work since the last idle point
  bool seg_busy = false;     // ... a kernel that writes the tracked-segment table (resident-path window)
  std::vector<int> cell_id, cell_cursor;  // bucketing scratch (reused)
  std::vector<uint8_t> skip_pad;
  // ... and a cycle that follows a sensor update finds its table made already: the sensor build launch carries
  // a few workgroups that form it for the update's yaw, the current lattice and the last horizon (TrigJob)
  bool trig_plan = false;               // kc_dwa_set_points / set_scan: a job may ride in this update's launch
  double trig_plan_yaw = 0.0;
  bool trig_ahead_valid = false;        // d_trig holds the table of (trig_ahead_yaw, trig_ahead_P, trig_ahead_lat)
  double trig_ahead_yaw = 0.0;
  size_t trig_ahead_P = 0;
  unsigned long long trig_ahead_lat = 0, lat_version = 0;  // lattice uploads (upload_samples)
  long long trig_rides = 0;             // get_option "trig_rides"
  bool device_trig = true;              // option "device_trig" / KC_DEVICE_TRIG: cos / sin(yaw_k) formed by the kernels
                                        // (kc_trig_exact.h); off: the host's libm table over the BAR (rounds 1-3)
  int seg_chunk = kSegChunkMin, seg_nch = 0, seg_nsup = 0;  // chunking of the tracked segment (cost kernel)
  long long last_nadm = -1;             // admissible count of the previous cycle (kernel choice)
  int cost_kernel_force = 0;            // 1: workgroup-per-sample, 2: wavefront-per-sample (KC_COST_KERNEL)
  bool trig_direct = false;             // host writes the trig table into device memory (large BAR)
  bool cost_batch_ok = false;           // sample_cost_batched_kernel may take kCostLdsBudget
  bool cost_batch_forced = false;       // ... value 2: for every list length (tests)
  int dil_cover = 0;                    // CollDev::cover the masks of the last sensor update were dilated for
  bool box_cover_on = true;             // option "box_cover": long boxes look several circles up in the dilated masks (CollDev::cover)
  bool cost_batch = true;               // option "cost_batch": the long-list cost kernel batches its per-sample part
  bool fold_publish = true;             // test hook KC_FOLD_PUBLISH=0: publish_kernel behind every cost kernel
  bool cost_obs_lds = true;             // tuning hook KC_COST_OBS_LDS=0: obstacle coordinates stay in global memory
  bool cost_lds_ok = false;             // sample_cost_kernel<true> may take kCostLdsBudget
  int fused_samples = 32, fused_block = 1024;
  int cycle_samples = 32;  // samples per workgroup of the single-launch cycle: 16 when 32 would leave half the CUs idle
  int perm_cs = 0;         // ... the dealt order on the device was built for
  int cycle_samples_opt = 0;  // option "cycle_samples": 0 auto, 16, 32
  int velocity_group = 0;     // option "velocity_group": samples per wavefront of the velocity sums (0 auto, 1, 4, 16)
  bool velocity_beside = true;  // option "velocity_beside": velocity_sums_kernel on a second stream beside the cost kernel
  hipStream_t aux_stream = nullptr;
  hipEvent_t aux_fork = nullptr, aux_join = nullptr;
  DevBuf<float> d_vsum;       // [2][n] smoothness / jerk sums of velocity_sums_kernel
  bool fused_shape_fixed = false;  // KC_FUSED_CFG given: no per-lattice choice of the roll-out tile
  bool have_sensor = false;

  // samples
  hm::VelocityLattice lat;    // host copy (vx, vy, row, omega values)
  size_t shard_first = 0, shard_count = 0;
  double vmax_lin = 0.0;      // max hypot(vx, vy) over the list
  DevBuf<double> d_vxt, d_vyt;   // value tables of the axes (rewritten by every new window)
  DevBuf<uint32_t> d_vidx;       // [n] (index into d_vxt) | (index into d_vyt) << 16: rewritten when the pattern changes
  // The index tables and the two walking orders of a window lattice belong to its PATTERN (which sample takes which
  // axis value; lattice signature).  A robot whose velocity moves changes pattern in every second cycle -- an axis
  // gains or loses a value, a value crosses |v| = kMinVel -- between a handful of patterns: the tables of the last
  // few stay on the device and a change back is a swap of pointers (tools/window_sweep.py: 380 us per change
  // before, the sort + six pageable copies of build_perm).
  struct PatternTables {
    uint64_t sig = 0;
    size_t n = 0;
    uint64_t stamp = 0;  // last use
    DevBuf<uint32_t> vidx, pvi, cpvi;
    DevBuf<int32_t> row, perm, prow, cperm, cprow;
    std::vector<int32_t> h_perm, h_dealt, rows;
    bool perm_valid = false, perm_plain_dev = false, perm_dealt_dev = false;
    size_t perm_first = 0, perm_count = 0;
    int perm_cs = 0;
  };
  std::vector<PatternTables> patterns;  // (inactive ones; the active pattern lives in the members above / below)
  uint64_t pattern_clock = 0;
  long pattern_hits = 0, pattern_builds = 0;
  DevBuf<int32_t> d_row;
  uint64_t up_sig = 0;           // signature / size of the pattern on the device
  size_t up_n = 0;
  std::vector<uint16_t> up_ix, up_iy;  // ... and the pattern itself for lists without a signature

  // per cycle
  size_t P = 0;               // points of the last roll-out
  size_t n_roll = 0;          // samples of the last roll-out (shard size)
  bool rolled = false, evaluated = false, external = false;
  PinBuf<double2> h_trig;
  DevBuf<double2> d_trig;
  PinBuf<uint32_t> h_bits;
  DevBuf<uint32_t> d_bits;
  PinBuf<double> h_ddz;
  DevBuf<double> d_ddz;
  DevBuf<float> d_px, d_py, d_costs;
  DevBuf<int> d_adm;  // admissible local sample ids (count lives in d_result[W_LIST])
  DevBuf<double2> d_pos;
  DevBuf<uint8_t> d_flags;
  DevBuf<float> d_vvx, d_vvy, d_vom;  // kc_cost_evaluate velocities
  bool have_vel = false;
  bool need_compact = false;  // flags exist but the admissible list does not
  bool list_dirty = false;    // a fused roll-out appended, no cost kernel re-armed
  DevBuf<unsigned long long> d_dbg;  // KC_DEBUG_STAMPS diagnostic only
  bool debug_stamps = false;

  // tracked segment + obstacles
  size_t S = 0, O = 0;
  float seg_len = 0.f, ref_len = 0.f, max_obs_dist = 0.f;
  bool seg_flat = false;      // every z of the tracked segment is +0.0f
  bool path_flat = false;     // ... of the resident path
  PinBuf<float> h_seg;  // sx | sy | sz | szz | acc
  std::vector<float> seg_stage;  // ... built here (cached memory), copied out once
  DevBuf<float> d_seg;
  // near table of the tracked segment (segment_near_kernel): rebuilt when the segment or the
  // reachable box changes, and only for cycles whose cost stage is expected to run the
  // wavefront-per-sample search (option "near_table": cells per side, 0 off)
  DevBuf<uint32_t> d_near;
  int near_side = 128;
  unsigned long long seg_version = 0, near_version = ~0ull;  // segment the table was built from
  float near_x0 = 0.f, near_y0 = 0.f, near_g = 0.f;
  bool near_ok = false;       // the table covers the running cycle
  // KC_DEBUG_HOST=1: where the host side of a cycle goes (steady_clock marks, printed at destroy)
  struct HostProf {
    bool on = false;
    std::chrono::steady_clock::time_point t[10];
    double sum[10] = {0};
    long n = 0;
    void mark(int i) { if (on) t[i] = std::chrono::steady_clock::now(); }
    long seen = 0;
    void close() {
      if (!on || ++seen <= 200) return;  // (the first cycles build orders and tables once)
      for (int i = 1; i < 8; ++i) sum[i] += std::chrono::duration<double, std::micro>(t[i] - t[i - 1]).count();
      sum[8] += std::chrono::duration<double, std::micro>(t[8] - t[0]).count();
      sum[9] += std::chrono::duration<double, std::micro>(t[9] - t[8]).count();
      ++n;
    }
  } hprof;
  bool ext_box_valid = false;  // bounding box of the caller-provided samples (kc_cost_upload / kc_cost_evaluate)
  double ext_box[4] = {0, 0, 0, 0};
  DevBuf<unsigned int> d_bbox;
  bool near_wanted = false;   // the last cycle asked for the table: the next segment update builds it ahead
  // resident reference path (kc_dwa_set_path): rows x | y | z | acc on the
  // device, edge lengths on the host (the window length is an ordered float sum)
  DevBuf<float> d_path;
  std::vector<float> path_edge;
  size_t path_n = 0;
  float path_len = 0.f;
  PinBuf<float> h_obs;  // ox | oy (sensor order, as setPointScan stores them)
  // obstacle buckets for the exact nearest-obstacle search (K3)
  BucketDev bucket{};
  PinBuf<int> h_cells;
  DevBuf<int> d_cells;
  PinBuf<float> h_bobs;  // bx | by in cell order
  DevBuf<float> d_bobs;
  PinBuf<uint8_t> h_skip;  // Chebyshev distance to the nearest non-empty cell
  DevBuf<uint8_t> d_skip;
  size_t n_bucketed = 0;

  DevBuf<long long> d_result;  // key, n_adm, compact index, scratch
  PinBuf<long long> h_result;
  PinBuf<long long> h_pub;     // {key, n_adm, compact, seq} written by the GPU
  long long seq = 0;           // last cycle sequence handed to finalize
  bool pub_pending = false;
  int team_max = 4;        // option "team_max" (0..4): workgroups of the cycle kernel with that many survivors cost them by teams
  bool perm_busy = false;  // a queued roll-out reads the walking orders (d_perm ...): cleared with `drained`
  bool drained = false;  // the host saw the last cost kernel's record: every earlier
                         // command of the stream has finished with the staging buffers
  PinBuf<float> h_row;         // winner row staging
  kc_result last{};
  bool have_last = false;

  // single-launch cycle (CycleTail form of rollout_collide_kernel)
  size_t lds_limit_hw = 64 * 1024;  // what the device grants (options toggle lds_limit / cost_lds_ok)
  bool cost_lds_hw = false, large_bar = false;
  bool write_paths = false;    // option "write_paths": the single-launch cycle stores the float rows too
  bool cycle_fused = true;     // option "fused_cycle": kc_dwa_cycle may take the single launch
  bool cycle_forced = false;   // ... value 2: also when the shard needs more than one workgroup per CU
  int num_cus = 256;
  bool cycle_launched = false; // the last roll-out call was a whole cycle
  bool paths_valid = true;     // d_px / d_py hold the rows of the last roll-out (a fused cycle
                               // materialises them only on demand)
  bool in_materialise = false;
  kc_state last_start{};       // start pose of the last roll-out (re-materialisation)
  DevBuf<uint32_t> d_adm_bits;         // admissible local ids of the running cycle (bitmap)
  PinBuf<uint32_t> h_wrow;             // best row of every workgroup of a single-launch cycle
  size_t wrow_off = 0;                 // words in front of the winner's row
  bool host_reduce = true;     // option "host_reduce": single-GPU cycles leave the reduction over the workgroups
                               // to the host (32-byte slots in pinned memory; no device-side epilogue)
  bool slots_pending = false;  // the last launch was such a cycle: fetch reduces the slots
  unsigned slots_G = 0;
  bool device_record_valid = true;  // d_result holds the last cycle's record (not after a host-reduced cycle)
  PinBuf<long long> h_slots;   // [grid][4]
  std::vector<int32_t> h_dealt;     // host copy of the dealt order (compacted index of the winner)
  std::vector<uint64_t> slot_pending;   // scratch of fetch_slots
  long long *xchg_send = nullptr;  // sharded call: the send record, this rank, words per rank (set by kc_dwa_cycle_sharded)
  int xchg_rank = 0, xchg_rw = 0;
  bool xchg_packed = false;        // ... and the cycle kernel of this call has written the rank's words itself
  bool sharded_call = false;   // kc_dwa_cycle_sharded: the cycle kernel leaves the host record to the
                               // hand-off behind the all-reduce
  long long rec_w4 = 0;        // row word of the record fetched last
  bool row_valid = false;      // h_wrow holds the winner row of `last`

  // sharding by rule + the exchange record of a sharded cycle (kc_shard.h)
  ShardLayout layout;
  hm::VelocityLattice full;    // KC_SHARD_ROWS: the full list (`lat` is this rank's share of it)
  std::vector<int32_t> gid;    // KC_SHARD_ROWS: id in `lat` -> global id (position in `full`)
  bool rows_active = false;    // `lat` is this rank's KC_SHARD_ROWS share of `full` (the share -- and gid -- may be EMPTY:
                               // more ranks than dealt rows; the state is this flag, never gid.empty())
  DevBuf<int32_t> d_gid;
  DevBuf<long long> d_xs, d_xr;      // send / reduced record
  PinBuf<long long> h_xvec, h_xrec;  // the reduced record and its 5-word hand-off record, written by the GPU
  int x_world = 0, x_rank = -1;      // what d_xs is armed for (the other ranks' words hold INT64_MAX)
  size_t x_rw = 0;
  long long xseq = 0;
  int64_t last_lat = -1;       // id in `lat` of the last winner when it lives on this context, else -1

  // a laser scan as a polyline: obstacle coordinates in beam order + the boxes of its <= 64 chunks, and the
  // near table of the obstacles over the reachable box (kc_onear_kernels.h; option "obs_near")
  std::vector<float> h_oscan;            // x[n] | y[n] | boxes [4][64]
  DevBuf<float> d_oscan;
  bool oscan_valid = false;
  size_t oscan_n = 0;
  int oscan_cs = 0, oscan_nch = 0;
  int oscan_scs = 0;                     // obstacles per quarter of a chunk (long scans: quarter boxes behind the chunk boxes), else 0
  unsigned long long sensor_version = 0, onear_version = ~0ull;
  DevBuf<uint4> d_onear;
  float onear_x0 = 0.f, onear_y0 = 0.f, onear_g = 0.f;
  bool onear_ok = false;                 // the table covers the running cycle
  int obs_union = 512;                   // option "obs_union": obstacle_union_scan up to this many obstacles (0: off).  (96 until a sweep over
                                         // dense 3-D clouds, tools/big_cloud_sweep.py: clusters of a hundred points put more than that into the
                                         // rectangle of every trajectory that passes them, and the ring walks behind the limit evaluate every
                                         // one of them exactly per point -- 10 k points 97 -> 55 us of cycle kernel, 30 k points 129 -> 78)
  bool obs_near_ahead = true;            // test hook KC_OBS_NEAR_AHEAD=0: the cycle builds the table itself
  long long onear_rides = 0, onear_builds = 0;  // tables built in the sensor launch / by a launch of their own
  bool onear_ahead = false;              // kc_dwa_set_scan planned a table (onear_args) for the sensor build launch
  ObsNearArgs onear_args{};
  bool obs_near_opt = true;
  int onear_side = 128;                  // cells per side of that table (option "obs_near": 0 off, 16..512)

  // non-planar sensor mount with LaserScan input: the octree frame is tilted (kc_tilt_dev.h)
  bool tilted = false;
  int tilt_kz = 0;             // the scan's voxel layer in the octree frame
  // A tilted scan whose voxel columns span more than 8192 cells (fine octrees, long ranges) keeps the columns
  // within kTiltCrop cells of the robot's own column: nothing farther can be reached by a roll-out (checked
  // per cycle against the horizon: rollout_impl), so dropping it changes no collision result.
  bool tilt_cropped = false;
  int tilt_cx = 0, tilt_cy = 0;  // the robot's column at the update (octree keys)
  double tilt_body_x = 0, tilt_body_y = 0;  // the pose of that update

  // drop_samples_ == false (trajectory_sampler.cpp:157-168; option "drop_samples" = 0)
  bool drop_samples = true;
  size_t num_ctrl_points = 0;  // numCtrlPoints_ = control_horizon / time_step (:88; option "num_ctrl_points")
  DevBuf<int> d_freeze, d_first_hit;   // [n] first zero-velocity step of a frozen sample (0: not frozen) / split path scratch
  DevBuf<float> d_frz;                 // [2][n] smoothness | jerk sums of the frozen profiles
  DevBuf<double> d_omega;              // [A] omega of every trig row
  DevBuf<double> d_sincostab;          // the 440 table values of kc_trig_exact.h beside the context's other tables
  bool freeze_valid = false;           // d_freeze describes the last roll-out
};

// largest point list the device-side sensor update takes (bucket grid of at most 64 x 64 cells: about one obstacle per
// cell up to 4 k points, hundreds per cell here); beyond: the host path, finer grid.  (262144 until a raw depth image --
// 307 200 points -- was priced: 6 ms of host build at 500 k points against 0.16 ms here, tools/big_cloud_sweep.py.)
constexpr size_t kSensorDeviceMax = 1048576;
constexpr int kTiltCrop = 4000;                // half side of the kept window of a cropped tilted scan, in voxel columns
constexpr size_t kSensorFusedMax = 32768;       // points up to which the one-launch sensor build CAN be used (spheres: it is their only device build)
constexpr size_t kSensorFusedPays = 18432;      // ... and up to which it is ahead: every workgroup reads every point (tools/big_cloud_sweep.py,
                                                // set_points + cycle with the one launch / the two: 10 k points 80.6 / 86.4 us, 16 k 161 / 160,
                                                // 20 k 84.1 / 79.6, 24 k 91.7 / 84.4, 30 k 100.3 / 93.6)
constexpr size_t kSensorFusedLds = 100 * 1024;  // dynamic LDS of sensor_fused_kernel (band rows; bucket tables + point ids)

inline int use_device(const kc_dwa *c) {
  KC_HIP(hipSetDevice(c->prm.device));
  return KC_OK;
}

inline unsigned blocks_for(size_t n, unsigned per) {
  return static_cast<unsigned>((n + per - 1) / per);
}

// accept one octree-frame point into the voxel column list
// host -> device for the per-update tables: plain stores through the BAR when
// the host can address device memory (the stream must not hold readers of
// `dst`, see quiesce_for_update), else a copy command
inline int upload_table(kc_dwa *c, void *dst, const void *src, size_t bytes) {
  if (bytes == 0) return KC_OK;
  if (c->trig_direct) {
    std::memcpy(dst, src, bytes);
    c->bar_dirty = true;
    return KC_OK;
  }
  KC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  return KC_OK;
}
// write-combined stores out of the core before anything is launched behind them
inline void bar_flush(kc_dwa *c) {
  if (c->bar_dirty) {
#if defined(__x86_64__)
    __builtin_ia32_sfence();
#endif
    c->bar_dirty = false;
  }
}
// Before the host overwrites per-update tables: nothing queued may still read
// them.  A cycle whose record the host has seen proves that everything queued
// before it has finished; work queued since then (dilate_kernel, copies) is
// tracked by `update_busy`.
inline int quiesce_for_update(kc_dwa *c, bool sensor_tables = true) {
  // (the tracked-segment table is only read by cost kernels, i.e. by cycles:
  // work queued by a sensor update since the last cycle does not touch it)
  if (!c->drained || (sensor_tables && c->update_busy) || (!sensor_tables && c->seg_busy)) {
    KC_HIP(hipStreamSynchronize(c->stream));
    c->update_busy = false;
    c->seg_busy = false;
    c->drained = true;
    c->perm_busy = false;
  }
  return KC_OK;
}

// global id (position in the caller's full list) of sample `lat_id` of this context's list
inline int64_t global_of(const kc_dwa *c, int64_t lat_id) {
  if (!c->rows_active || lat_id < 0) return lat_id;
  return static_cast<size_t>(lat_id) < c->gid.size() ? static_cast<int64_t>(c->gid[static_cast<size_t>(lat_id)]) : -1;
}

// the caller's full list (sample_window output, velocity look-ups by global id)
inline const hm::VelocityLattice &full_list(const kc_dwa *c) { return c->rows_active ? c->full : c->lat; }


// is there any occupied voxel column?  (after a device-side update the count is
// not known on the host: any point may be one)
inline bool any_voxel(const kc_dwa *c) {
  return c->host_lists_valid ? !c->vox_kx.empty() : c->O > 0;
}

// ---- host functions shared between the translation units (kc_dwa.hip: context, options, lattice and shares;
// kc_dwa_sensor.hip: sensor data, tracked segment, near tables; kc_dwa_cycle.hip: roll-out, costs, results;
// kc_dwa_shard.hip: the exchange of a sharded cycle)
// context / lattice
bool trig_selfcheck_ok();
int ensure_sincostab(kc_dwa *c);   // the 440 table values of kc_trig_exact.h in the context's device memory
int upload_omega(kc_dwa *c);
int upload_samples(kc_dwa *c);
int apply_shard_rule(kc_dwa *c);
int build_perm(kc_dwa *c, bool want_dealt);
// sensor data, tracked segment, near tables
void build_host_lists(kc_dwa *c, const float *xyz, size_t n);
int ensure_host_lists(kc_dwa *c);
int launch_dilate(kc_dwa *c);
void sensor_kernel_limits(kc_dwa *c);  // dynamic-LDS limit of sensor_fused_kernel -> sensor_fused_ok
int ensure_near_table_box(kc_dwa *c, double lo_x, double lo_y, double hi_x, double hi_y, double margin);
int ensure_near_table(kc_dwa *c, double x, double y, double margin = 0.0);
int near_table_ahead(kc_dwa *c);
int ensure_onear(kc_dwa *c, double x, double y, bool build = true);
// cycle
double cycle_reach(const kc_dwa *c);
int ensure_cycle_buffers(kc_dwa *c, size_t n, size_t P);
int window_geometry(kc_dwa *c, double wx, double wy, double reach, CollDev &cd);
int window_bits_host(kc_dwa *c, CollDev &cd);
int build_window_at(kc_dwa *c, double wx, double wy, double reach, CollDev &cd);
int rollout_impl(kc_dwa *c, const kc_state *start, size_t P, bool want_cycle, bool trig_ready = false);
int materialise_paths(kc_dwa *c);
void cycle_kernel_limits(kc_dwa *c);   // dynamic-LDS limits of the roll-out / cost kernels -> lds_limit, cost_lds_ok, cost_batch_ok
int launch_init_result(kc_dwa *c);                       // init_result_kernel on the context's stream
int launch_trig_table(const TrigJob &tj, hipStream_t s);  // trig_table_kernel
// shard
long long local_bound(const kc_dwa *c, int64_t raw);
