// Sampling-controller hot path on gfx950: roll-out + collision, cost terms,
// argmin.  C ABI in include/kompass_hip.h; reference citations are relative to
// <reference>/src/kompass_cpp/kompass_cpp/.
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
  std::vector<double2> scan_cs;         // angle table does not change between scans)
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
  unsigned long long sensor_version = 0, onear_version = ~0ull;
  DevBuf<uint4> d_onear;
  float onear_x0 = 0.f, onear_y0 = 0.f, onear_g = 0.f;
  bool onear_ok = false;                 // the table covers the running cycle
  int obs_union = 96;                    // option "obs_union": obstacle_union_scan up to this many obstacles (0: off)
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

namespace {

// largest point list the device-side sensor update takes (bucket grid of at most 64 x 64 cells:
// about one obstacle per cell up to 4 k points, 64 per cell here); beyond: the host path, finer grid
constexpr size_t kSensorDeviceMax = 262144;
constexpr int kTiltCrop = 4000;                // half side of the kept window of a cropped tilted scan, in voxel columns
constexpr size_t kSensorFusedMax = 32768;       // points up to which the one-launch sensor build is used
constexpr size_t kSensorFusedLds = 100 * 1024;  // dynamic LDS of sensor_fused_kernel (band rows; bucket tables + point ids)

int use_device(const kc_dwa *c) {
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
int upload_table(kc_dwa *c, void *dst, const void *src, size_t bytes) {
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
int quiesce_for_update(kc_dwa *c, bool sensor_tables = true) {
  // (the tracked-segment table is only read by cost kernels, i.e. by cycles:
  // work queued by a sensor update since the last cycle does not touch it)
  if (!c->drained || (sensor_tables && c->update_busy) || (!sensor_tables && c->seg_busy)) {
    KC_HIP(hipStreamSynchronize(c->stream));
    c->update_busy = false;
    c->seg_busy = false;
    c->drained = true;
  }
  return KC_OK;
}

inline void add_voxel(kc_dwa *c, float px, float py, float pz) {
  const double inv = c->inv_res;
  const double fx = std::floor(inv * static_cast<double>(px));
  const double fy = std::floor(inv * static_cast<double>(py));
  const double fz = std::floor(inv * static_cast<double>(pz));
  if (!(std::fabs(fx) < 32768.0 && std::fabs(fy) < 32768.0 &&
        std::fabs(fz) < 32768.0))
    return;  // outside the 16-level octree: octomap drops the point
  const int32_t kz = static_cast<int32_t>(fz);
  if (c->tilted) {  // tilted octree frame: no z interval to gate with, the exact 3-D test decides
    c->tilt_kz = kz;
    c->vox_kx.push_back(static_cast<int32_t>(fx));
    c->vox_ky.push_back(static_cast<int32_t>(fy));
    return;
  }
  const double zlo = static_cast<double>(kz) * c->res;
  const double zhi = static_cast<double>(kz + 1) * c->res;
  const double zc = -static_cast<double>(c->frame.t[2]);
  if (c->prm.shape == KC_SPHERE) {
    double ddz = 0.0;
    if (zlo - zc > ddz) ddz = zlo - zc;
    if (zc - zhi > ddz) ddz = zc - zhi;
    if (ddz > c->radius) return;
    c->vox_ddz.push_back(ddz);
  } else {
    const double hz = c->height / 2.0;
    if (!(zlo <= zc + hz && zhi >= zc - hz)) return;
  }
  c->vox_kx.push_back(static_cast<int32_t>(fx));
  c->vox_ky.push_back(static_cast<int32_t>(fy));
}

// dilation radii in cells (see dilate_kernel)
struct DilGeom {
  double rho_in, rho_out;
  int R;
};
DilGeom dil_geom(const kc_dwa *c) {
  DilGeom g;
  g.rho_in = (c->prm.shape == KC_BOX ? std::min(static_cast<double>(c->prm.dims[0]),
                                                static_cast<double>(c->prm.dims[1])) / 2.0
                                     : c->radius) / c->res;
  if (c->prm.shape == KC_SPHERE) {
    // a voxel column with z gap g collides within the horizontal radius sqrt(R^2 - g^2): every column
    // of this update does so at least within the radius of the largest gap (certain hits), and at most
    // within R (possible hits)
    const double gmax = c->sphere_ddz_max;
    const double r2 = c->radius * c->radius - gmax * gmax;
    g.rho_in = (gmax >= 0.0 && r2 > 0.0) ? std::sqrt(r2) * (1.0 - 1e-9) / c->res : -1.0;
  }
  g.rho_out = (c->prm.shape == KC_BOX
                   ? std::sqrt(std::pow(static_cast<double>(c->prm.dims[0]) / 2.0, 2) +
                               std::pow(static_cast<double>(c->prm.dims[1]) / 2.0, 2))
                   : c->radius) / c->res;
  g.R = static_cast<int>(std::floor(g.rho_out + 1e-6)) + 1;
  return g;
}

// extent of the sensor bitmap from the key bounding box (padded so that the
// dilated masks fit); *fits = false when it is too sparse / far for the fused
// path.  Reserves the three device bitmaps.
int bitmap_extent(kc_dwa *c, int lox, int loy, int hix, int hiy, bool *fits) {
  const DilGeom dg = dil_geom(c);
  c->have_dil = (c->prm.shape != KC_SPHERE || c->sphere_ddz_max >= 0.0) && std::isfinite(dg.rho_out) &&
                dg.R <= 30 && (dg.rho_in >= 0.0 || c->prm.shape == KC_SPHERE);
  if (c->have_dil) {
    const int pad = dg.R + 1;
    lox -= pad;
    loy -= pad;
    hix += pad;
    hiy += pad;
  }
  const long W = static_cast<long>(hix) - lox + 1, H = static_cast<long>(hiy) - loy + 1;
  *fits = !(W > 8192 || H > 8192);
  if (!*fits) return KC_OK;
  c->gkx0 = lox;
  c->gky0 = loy;
  c->gH = static_cast<int>(H);
  c->gwpr = static_cast<int>((W + 31) / 32);
  const size_t nwords = static_cast<size_t>(c->gH) * c->gwpr;
  KC_TRY(c->d_gbits.reserve(nwords));
  if (c->have_dil) {
    KC_TRY(c->d_ginner.reserve(nwords));
    KC_TRY(c->d_gouter.reserve(nwords));
  }
  return KC_OK;
}

// run half-widths of the two discs per row offset (see DilArgs)
void dil_tables(const DilGeom &dg, signed char win[kMaxDil + 1], signed char wout[kMaxDil + 1]);

// the two dilated masks from the bitmap in d_gbits
int launch_dilate(kc_dwa *c) {
  if (!c->have_dil) return KC_OK;
  const DilGeom dg = dil_geom(c);
  const size_t nwords = static_cast<size_t>(c->gH) * c->gwpr;
  DilArgs da{};
  da.g = c->d_gbits.p;
  da.inner = c->d_ginner.p;
  da.outer = c->d_gouter.p;
  da.H = c->gH;
  da.wpr = c->gwpr;
  da.R = dg.R;
  dil_tables(dg, da.win, da.wout);
  const unsigned nb = static_cast<unsigned>((nwords + 255) / 256);
  KC_TRY(c->timing.start("dilate_kernel", c->stream));
  hipLaunchKernelGGL(dilate_kernel, dim3(nb), dim3(256), 0, c->stream, da);
  KC_TRY(c->timing.stop(c->stream));
  KC_HIP(hipGetLastError());
  c->update_busy = true;
  return KC_OK;
}

void dil_tables(const DilGeom &dg, signed char win[kMaxDil + 1], signed char wout[kMaxDil + 1]) {
  for (int j = 0; j <= kMaxDil; ++j) {
    win[j] = wout[j] = -1;
    if (j > dg.R) continue;
    // inner: largest i with hypot(i, j) <= rho_in - 1e-6
    const double ri = dg.rho_in - 1e-6;
    if (ri >= 0.0 && static_cast<double>(j) <= ri) {
      int i = static_cast<int>(std::floor(std::sqrt(ri * ri - static_cast<double>(j) * j)));
      while (i >= 0 && std::hypot(static_cast<double>(i), static_cast<double>(j)) > ri) --i;
      win[j] = static_cast<signed char>(std::min(i, 31));
    }
    // outer: largest i with hypot((i-1)+, (j-1)+) <= rho_out + 1e-6
    const double ro = dg.rho_out + 1e-6;
    const double jj = std::max(j - 1, 0);
    if (jj <= ro) {
      int i = static_cast<int>(std::floor(std::sqrt(ro * ro - jj * jj))) + 2;
      while (i > 0 && std::hypot(static_cast<double>(std::max(i - 1, 0)), jj) > ro) --i;
      wout[j] = static_cast<signed char>(std::min(i, 31));
    }
  }
}

// occupancy bits of the accepted voxel columns over their bounding box ->
// device, once per sensor update (the fused roll-out kernel copies its
// reachable window out of it)
int upload_voxels(kc_dwa *c) {
  c->have_gbits = false;
  size_t nv = c->vox_kx.size();
  c->tilt_cropped = false;
  if (nv == 0) return KC_OK;
  int lox = INT32_MAX, loy = INT32_MAX, hix = INT32_MIN, hiy = INT32_MIN;
  auto bounds = [&] {
    lox = loy = INT32_MAX;
    hix = hiy = INT32_MIN;
    for (size_t i = 0; i < nv; ++i) {
      lox = std::min(lox, c->vox_kx[i]);
      hix = std::max(hix, c->vox_kx[i]);
      loy = std::min(loy, c->vox_ky[i]);
      hiy = std::max(hiy, c->vox_ky[i]);
    }
  };
  bounds();
  if (c->tilted && (static_cast<long>(hix) - lox + 1 > 8192 || static_cast<long>(hiy) - loy + 1 > 8192)) {
    // the robot's own column: the body origin in octree coordinates, F^-1 (x, y, 0) = R^T ((x, y, 0) - t)
    const hm::Rigid3f &F = c->frame;
    const double d[3] = {c->tilt_body_x - static_cast<double>(F.t[0]), c->tilt_body_y - static_cast<double>(F.t[1]),
                         0.0 - static_cast<double>(F.t[2])};
    const double ox = F.R[0][0] * d[0] + F.R[1][0] * d[1] + F.R[2][0] * d[2];
    const double oy = F.R[0][1] * d[0] + F.R[1][1] * d[1] + F.R[2][1] * d[2];
    c->tilt_cx = static_cast<int>(std::floor(ox * c->inv_res));
    c->tilt_cy = static_cast<int>(std::floor(oy * c->inv_res));
    size_t w = 0;
    for (size_t i = 0; i < nv; ++i)
      if (std::abs(c->vox_kx[i] - c->tilt_cx) <= kTiltCrop && std::abs(c->vox_ky[i] - c->tilt_cy) <= kTiltCrop) {
        c->vox_kx[w] = c->vox_kx[i];
        c->vox_ky[w] = c->vox_ky[i];
        if (c->vox_ddz.size() == nv) c->vox_ddz[w] = c->vox_ddz[i];
        ++w;
      }
    c->vox_kx.resize(w);
    c->vox_ky.resize(w);
    if (c->vox_ddz.size() == nv) c->vox_ddz.resize(w);
    nv = w;
    c->tilt_cropped = true;
    if (nv == 0) return KC_OK;
    bounds();
  }
  c->sphere_ddz_max = -1.0;
  if (c->prm.shape == KC_SPHERE && c->vox_ddz.size() == nv)
    c->sphere_ddz_max = *std::max_element(c->vox_ddz.begin(), c->vox_ddz.end());
  bool fits = false;
  KC_TRY(bitmap_extent(c, lox, loy, hix, hiy, &fits));
  if (!fits) return KC_OK;  // too sparse/far: split path only
  const size_t nwords = static_cast<size_t>(c->gH) * c->gwpr;
  KC_TRY(c->h_gbits.reserve(nwords));
  std::memset(c->h_gbits.p, 0, nwords * sizeof(uint32_t));
  for (size_t i = 0; i < nv; ++i) {
    const int cx = c->vox_kx[i] - c->gkx0, cy = c->vox_ky[i] - c->gky0;
    c->h_gbits.p[static_cast<size_t>(cy) * c->gwpr + (cx >> 5)] |= 1u << (cx & 31);
  }
  KC_TRY(upload_table(c, c->d_gbits.p, c->h_gbits.p, nwords * sizeof(uint32_t)));
  c->gz_valid = false;
  if (c->prm.shape == KC_SPHERE) {
    // z gaps of the accepted voxels: one value per voxel layer within the sphere's height
    std::vector<double> lut(c->vox_ddz.begin(), c->vox_ddz.end());
    std::sort(lut.begin(), lut.end());
    lut.erase(std::unique(lut.begin(), lut.end()), lut.end());
    if (lut.size() <= 255) {
      const size_t gW = static_cast<size_t>(c->gwpr) * 32, ncell = gW * c->gH;
      KC_TRY(c->h_gz.reserve(ncell));
      KC_TRY(c->d_gz.reserve(ncell));
      KC_TRY(c->h_zlut.reserve(256));
      KC_TRY(c->d_zlut.reserve(256));
      std::memset(c->h_gz.p, 0, ncell);
      for (size_t i = 0; i < nv; ++i) {
        const size_t cell = static_cast<size_t>(c->vox_ky[i] - c->gky0) * gW + (c->vox_kx[i] - c->gkx0);
        const uint8_t code =
            static_cast<uint8_t>(std::lower_bound(lut.begin(), lut.end(), c->vox_ddz[i]) - lut.begin() + 1);
        uint8_t &g = c->h_gz.p[cell];
        if (g == 0 || code < g) g = code;  // the smallest gap of the column decides
      }
      for (size_t k = 0; k < lut.size(); ++k) c->h_zlut.p[k] = lut[k];
      c->sphere_layers = lut.size();
      KC_TRY(upload_table(c, c->d_gz.p, c->h_gz.p, ncell));
      KC_TRY(upload_table(c, c->d_zlut.p, c->h_zlut.p, lut.size() * sizeof(double)));
      c->gz_valid = true;
    }
  }
  if (!c->trig_direct) c->update_busy = true;
  bar_flush(c);  // the kernels behind it read the bitmap
  KC_TRY(launch_dilate(c));
  c->have_gbits = true;
  return KC_OK;
}

// Bucket the world-frame obstacle points (h_obs) on a uniform grid and upload
// them in cell order.  Non-finite points can never win `dist < minDist`
// (trajectory.h:229) and are left out.
int upload_obstacles(kc_dwa *c, size_t n) {
  c->O = n;
  c->n_bucketed = 0;
  if (n == 0) return KC_OK;
  const float *ox = c->h_obs.p, *oy = c->h_obs.p + n;
  double lox = DBL_MAX, loy = DBL_MAX, hix = -DBL_MAX, hiy = -DBL_MAX;
  size_t nf = 0;
  for (size_t i = 0; i < n; ++i) {
    if (!std::isfinite(ox[i]) || !std::isfinite(oy[i])) continue;
    lox = std::min(lox, static_cast<double>(ox[i]));
    loy = std::min(loy, static_cast<double>(oy[i]));
    hix = std::max(hix, static_cast<double>(ox[i]));
    hiy = std::max(hiy, static_cast<double>(oy[i]));
    ++nf;
  }
  BucketDev &b = c->bucket;
  std::memset(&b, 0, sizeof(b));
  b.cap = static_cast<double>(c->max_obs_dist) * 1.001;
  if (nf == 0) {  // nothing can ever be closer than FLT_MAX
    b.W = b.H = 1;
    b.g = 1.0;
    b.inv_g = 1.0;
    KC_TRY(c->h_cells.reserve(2));
    KC_TRY(c->d_cells.reserve(2));
    c->h_cells.p[0] = c->h_cells.p[1] = 0;
    KC_HIP(hipMemcpyAsync(c->d_cells.p, c->h_cells.p, 2 * sizeof(int),
                          hipMemcpyHostToDevice, c->stream));
    KC_TRY(c->d_bobs.reserve(2));
    KC_TRY(c->h_skip.reserve(4));
    KC_TRY(c->d_skip.reserve(4));
    std::memset(c->h_skip.p, 255, 4);
    KC_HIP(hipMemcpyAsync(c->d_skip.p, c->h_skip.p, 4, hipMemcpyHostToDevice, c->stream));
    b.skip = c->d_skip.p;
    b.cell_start = c->d_cells.p;
    b.bx = c->d_bobs.p;
    b.by = c->d_bobs.p + 1;
    return KC_OK;
  }
  // about one obstacle per cell, at most 64 x 64 cells so that the cell and
  // skip tables sit in LDS (sample_cost_kernel); a finer grid in global memory
  // for very long lists
  const int kMaxSide =
      nf <= 65536 ? std::min(64, std::max(8, static_cast<int>(std::ceil(std::sqrt(
                                                 static_cast<double>(nf))))))
                  : 256;
  const double ext = std::max(hix - lox, hiy - loy);
  b.g = std::max(0.125, ext / (kMaxSide - 1));
  b.inv_g = 1.0 / b.g;
  b.gx0 = lox;
  b.gy0 = loy;
  // (the quotients are >= 0 and far below 2^31: truncation is floor)
  b.W = std::min(kMaxSide, static_cast<int>((hix - lox) * b.inv_g) + 1);
  b.H = std::min(kMaxSide, static_cast<int>((hiy - loy) * b.inv_g) + 1);
  const size_t ncell = static_cast<size_t>(b.W) * b.H;
  KC_TRY(c->h_cells.reserve(ncell + 1));
  KC_TRY(c->d_cells.reserve(ncell + 1));
  KC_TRY(c->h_bobs.reserve(2 * nf));
  KC_TRY(c->d_bobs.reserve(2 * nf));
  int *cs = c->h_cells.p;
  std::fill(cs, cs + ncell + 1, 0);
  // one pass for the cell of every point (-1: not finite), one for the scatter
  c->cell_id.resize(n);
  int *cid = c->cell_id.data();
  for (size_t i = 0; i < n; ++i) {
    if (!std::isfinite(ox[i]) || !std::isfinite(oy[i])) {
      cid[i] = -1;
      continue;
    }
    int cx = static_cast<int>((static_cast<double>(ox[i]) - b.gx0) * b.inv_g);
    int cy = static_cast<int>((static_cast<double>(oy[i]) - b.gy0) * b.inv_g);
    cx = std::min(std::max(cx, 0), b.W - 1);
    cy = std::min(std::max(cy, 0), b.H - 1);
    const int id = cy * b.W + cx;
    cid[i] = id;
    cs[id + 1]++;
  }
  for (size_t k = 0; k < ncell; ++k) cs[k + 1] += cs[k];
  c->cell_cursor.assign(cs, cs + ncell);
  int *cursor = c->cell_cursor.data();
  float *bx = c->h_bobs.p, *by = c->h_bobs.p + nf;
  for (size_t i = 0; i < n; ++i) {
    if (cid[i] < 0) continue;
    const int dst = cursor[cid[i]]++;
    bx[dst] = ox[i];
    by[dst] = oy[i];
  }
  // Chebyshev distance transform of the non-empty cells (two chamfer passes
  // with the 8-neighbourhood are exact for the Chebyshev metric); a border of
  // 255 around the table keeps the inner loops free of range tests
  KC_TRY(c->h_skip.reserve(ncell + 4));
  KC_TRY(c->d_skip.reserve(ncell + 4));
  {
    const int W = b.W, H = b.H, Wp = W + 2;
    c->skip_pad.assign(static_cast<size_t>(Wp) * (H + 2), 255);
    uint8_t *pad = c->skip_pad.data();
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        const size_t k = static_cast<size_t>(y) * W + x;
        if (cs[k + 1] > cs[k]) pad[(y + 1) * Wp + x + 1] = 0;
      }
    for (int y = 1; y <= H; ++y) {
      uint8_t *r = pad + y * Wp, *u = r - Wp;
      for (int x = 1; x <= W; ++x) {
        const int m = std::min(std::min<int>(r[x - 1], u[x]), std::min<int>(u[x - 1], u[x + 1]));
        if (m + 1 < r[x]) r[x] = static_cast<uint8_t>(m + 1);
      }
    }
    for (int y = H; y >= 1; --y) {
      uint8_t *r = pad + y * Wp, *l = r + Wp;
      for (int x = W; x >= 1; --x) {
        const int m = std::min(std::min<int>(r[x + 1], l[x]), std::min<int>(l[x + 1], l[x - 1]));
        if (m + 1 < r[x]) r[x] = static_cast<uint8_t>(m + 1);
      }
    }
    uint8_t *sk = c->h_skip.p;
    for (int y = 0; y < H; ++y) std::memcpy(sk + static_cast<size_t>(y) * W, pad + (y + 1) * Wp + 1, W);
  }
  for (size_t k = ncell; k < ncell + 4; ++k) c->h_skip.p[k] = 255;  // word padding
  KC_TRY(upload_table(c, c->d_skip.p, c->h_skip.p, ncell + 4));
  KC_TRY(upload_table(c, c->d_cells.p, cs, (ncell + 1) * sizeof(int)));
  KC_TRY(upload_table(c, c->d_bobs.p, c->h_bobs.p, 2 * nf * sizeof(float)));
  if (!c->trig_direct) c->update_busy = true;
  bar_flush(c);
  b.skip = c->d_skip.p;
  b.cell_start = c->d_cells.p;
  b.bx = c->d_bobs.p;
  b.by = c->d_bobs.p + nf;
  b.nobs = static_cast<int>(nf);
  c->n_bucketed = nf;
  return KC_OK;
}

// kc_trig_exact.h against the installed libm, once per process: a fixed argument set over every branch of the
// algorithm (tiny, Taylor, table, pi/2 - x, Cody-Waite with every quadrant) and yaw chains as the roll-out forms
// them.  Any difference (another libm: a build with FMA contraction, a different algorithm) switches the device
// trig off for the process -- the host table path is exact by construction.
static const double kc_sincostab_host[440] = {KC_SINCOSTAB_VALUES};
int trig_selfcheck_run(long *compared) {
  unsigned long long st = 0x9E3779B97F4A7C15ull;
  auto next = [&st]() {  // splitmix64
    unsigned long long z = (st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  auto unit = [&next]() { return static_cast<double>(next() >> 11) * 0x1p-53; };
  long n = 0, bad = 0;
  auto chk = [&](double x) {
    double s, c, rs, rc;
    if (!trig::sincos_exact(x, &s, &c, kc_sincostab_host)) return;
    ::sincos(x, &rs, &rc);
    ++n;
    if (std::memcmp(&s, &rs, 8) != 0 || std::memcmp(&c, &rc, 8) != 0) ++bad;
  };
  const double ranges[][2] = {{0.0, 1e-7}, {0.0, 0.13}, {0.12, 0.86}, {0.85, 2.43}, {2.42, 7.0}, {0.0, 100.0}, {100.0, 1.0e8}};
  for (const auto &r : ranges)
    for (int i = 0; i < 4000; ++i) {
      const double x = r[0] + (r[1] - r[0]) * unit();
      chk(x);
      chk(-x);
    }
  for (int i = 0; i < 100; ++i) {
    double yaw = -3.2 + 6.4 * unit();
    const double w = (-3.0 + 6.0 * unit()) * 0.05;
    for (int k = 0; k < 100; ++k) {
      chk(yaw);
      yaw += w;
    }
  }
  for (int e = -1074; e < 27; e += 3) chk(std::ldexp(1.0 + unit(), e));
  chk(0.0);
  chk(-0.0);
  if (compared) *compared = n;
  return static_cast<int>(bad);
}
bool trig_selfcheck_ok() {
  static const bool ok = trig_selfcheck_run(nullptr) == 0;
  return ok;
}

// omega of every trig row on the device (drop_samples = false: the velocity step of a frozen profile)
int upload_omega(kc_dwa *c) {
  const size_t A = c->lat.omega_values.size();
  if (A == 0) return KC_OK;
  KC_TRY(c->d_omega.reserve(A));
  KC_HIP(hipMemcpyAsync(c->d_omega.p, c->lat.omega_values.data(), A * sizeof(double), hipMemcpyHostToDevice, c->stream));
  KC_HIP(hipStreamSynchronize(c->stream));  // pageable source
  return KC_OK;
}

int upload_samples(kc_dwa *c) {
  const hm::VelocityLattice &lat = c->lat;
  const size_t n = lat.size();
  if (n > c->prm.max_samples)
    KC_FAIL(KC_ERR_RANGE, "sample count %zu exceeds max_samples %zu", n,
            c->prm.max_samples);
  // (the reach radius this feeds is a bound with 1e-4 of slack: the largest |vx| and |vy| of the axes)
  double ax = 0.0, ay = 0.0;
  for (double v : lat.vx_values) ax = std::max(ax, std::fabs(v));
  for (double v : lat.vy_values) ay = std::max(ay, std::fabs(v));
  c->vmax_lin = std::sqrt(ax * ax + ay * ay) * (1.0 + 1e-12);
  // A controller draws a new window every cycle: the velocities change, the pattern -- which sample
  // takes which axis value, which samples share an omega -- rarely does.  The index arrays on the device
  // and the orders the kernels walk the list in depend on that pattern only.
  const bool same = n == c->up_n && ((lat.signature != 0 && lat.signature == c->up_sig) ||
                                     (lat.signature == 0 && c->up_sig == 0 && c->uploaded_rows == lat.row &&
                                      c->up_ix == lat.ix && c->up_iy == lat.iy));
  ++c->lat_version;
  c->shard_first = 0;
  c->shard_count = n;
  if (!same) c->perm_valid = false;  // (the orders also belong to one shard: perm_first / perm_count)
  if (n == 0) return KC_OK;
  const size_t nx = lat.vx_values.size(), ny = lat.vy_values.size();
  KC_TRY(c->d_vxt.reserve(nx));
  KC_TRY(c->d_vyt.reserve(ny));
  KC_TRY(c->d_vidx.reserve(n));
  KC_TRY(c->d_row.reserve(n));
  KC_TRY(c->d_omega.reserve(std::max<size_t>(lat.omega_values.size(), 1)));  // (the kernels' trig rows: omega of a row)
  std::vector<uint32_t> packed;
  if (!same) {
    packed.resize(n);
    for (size_t i = 0; i < n; ++i) packed[i] = static_cast<uint32_t>(lat.ix[i]) | (static_cast<uint32_t>(lat.iy[i]) << 16);
  }
  if (c->trig_direct) {
    // straight into device memory over the BAR (no copy command, no stream wait);
    // nothing queued may still read the old list
    if (!c->drained) {
      KC_HIP(hipStreamSynchronize(c->stream));
      c->drained = true;
      c->update_busy = false;
    }
    std::memcpy(c->d_vxt.p, lat.vx_values.data(), nx * sizeof(double));
    std::memcpy(c->d_vyt.p, lat.vy_values.data(), ny * sizeof(double));
    std::memcpy(c->d_omega.p, lat.omega_values.data(), lat.omega_values.size() * sizeof(double));
    if (!same) {
      std::memcpy(c->d_vidx.p, packed.data(), n * sizeof(uint32_t));
      std::memcpy(c->d_row.p, lat.row.data(), n * sizeof(int32_t));
    }
    c->bar_dirty = true;
    bar_flush(c);
  } else {
    KC_HIP(hipMemcpyAsync(c->d_vxt.p, lat.vx_values.data(), nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    KC_HIP(hipMemcpyAsync(c->d_vyt.p, lat.vy_values.data(), ny * sizeof(double), hipMemcpyHostToDevice, c->stream));
    KC_HIP(hipMemcpyAsync(c->d_omega.p, lat.omega_values.data(), lat.omega_values.size() * sizeof(double),
                          hipMemcpyHostToDevice, c->stream));
    if (!same) {
      KC_HIP(hipMemcpyAsync(c->d_vidx.p, packed.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
      KC_HIP(hipMemcpyAsync(c->d_row.p, lat.row.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    }
    KC_HIP(hipStreamSynchronize(c->stream));  // pageable sources
  }
  if (!same) {
    c->uploaded_rows = lat.row;
    c->up_sig = lat.signature;
    c->up_n = n;
    if (lat.signature == 0) {
      c->up_ix = lat.ix;
      c->up_iy = lat.iy;
    } else {
      c->up_ix.clear();
      c->up_iy.clear();
    }
  }
  return KC_OK;
}

// global id (position in the caller's full list) of sample `lat_id` of this context's list
inline int64_t global_of(const kc_dwa *c, int64_t lat_id) {
  if (!c->rows_active || lat_id < 0) return lat_id;
  return static_cast<size_t>(lat_id) < c->gid.size() ? static_cast<int64_t>(c->gid[static_cast<size_t>(lat_id)]) : -1;
}

// c->lat holds the caller's FULL list: keep this rank's share under the shard rule and upload
int apply_shard_rule(kc_dwa *c) {
  ShardLayout &L = c->layout;
  c->gid.clear();
  c->rows_active = false;
  c->full.clear();
  if (L.mode < 0) return upload_samples(c);
  const size_t n = c->lat.size();
  L.n_total = n;
  const size_t me = static_cast<size_t>(L.rank);
  if (L.mode == KC_SHARD_BLOCKS) {
    shard_blocks(n, L.world, L);
    KC_TRY(upload_samples(c));
    c->shard_first = L.first[me];
    c->shard_count = L.count[me];
    return KC_OK;
  }
  // KC_SHARD_ROWS: the deal depends on the pattern of trig rows only (a controller draws a new
  // window every cycle: the velocities change, the pattern rarely does)
  const bool same = L.rows_seen == c->lat.row && L.gids.size() == static_cast<size_t>(L.world);
  if (!same) shard_rows(c->lat.row, L.world, L);
  c->full = std::move(c->lat);
  c->lat.clear();
  const std::vector<int32_t> &mine = L.gids[me];
  c->gid = mine;
  c->rows_active = true;
  // this rank's rows, relabelled in ascending order of the full list's labels
  std::vector<int32_t> relabel(c->full.omega_values.size(), -1);
  for (int32_t g : mine) relabel[static_cast<size_t>(c->full.row[static_cast<size_t>(g)])] = 0;
  for (size_t a = 0; a < relabel.size(); ++a)
    if (relabel[a] == 0) {
      relabel[a] = static_cast<int32_t>(c->lat.omega_values.size());
      c->lat.omega_values.push_back(c->full.omega_values[a]);
    }
  // (the axis tables stay whole, the share keeps its samples' indices into them)
  c->lat.vx_values = c->full.vx_values;
  c->lat.vy_values = c->full.vy_values;
  c->lat.ix.reserve(mine.size());
  c->lat.iy.reserve(mine.size());
  c->lat.row.reserve(mine.size());
  for (int32_t g : mine)
    c->lat.push(c->full.ix[static_cast<size_t>(g)], c->full.iy[static_cast<size_t>(g)],
                relabel[static_cast<size_t>(c->full.row[static_cast<size_t>(g)])]);
  if (c->full.signature)  // the share of a window lattice has a pattern of its own
    c->lat.signature = hm::lattice_mix(hm::lattice_mix(c->full.signature, 0x726f7773ull + static_cast<uint64_t>(L.rank)),
                                       static_cast<uint64_t>(L.world)) | 1ull;
  KC_TRY(upload_samples(c));  // (shard = the whole of `lat`)
  if (!same || c->d_gid.cap < mine.size()) {
    KC_TRY(c->d_gid.reserve(std::max<size_t>(mine.size(), 1)));
    if (!mine.empty()) KC_HIP(hipMemcpy(c->d_gid.p, mine.data(), mine.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  return KC_OK;
}

// the caller's full list (sample_window output, velocity look-ups by global id)
inline const hm::VelocityLattice &full_list(const kc_dwa *c) { return c->rows_active ? c->full : c->lat; }

// host lists of a global-frame point update (add_voxel per point, obstacle
// coordinates through obs_tf): the sensor path of the host, and the lazy
// fallback of the device path for code that walks the lists (split roll-out,
// pose batches)
void build_host_lists(kc_dwa *c, const float *xyz, size_t n) {
  c->vox_kx.clear();
  c->vox_ky.clear();
  c->vox_ddz.clear();
  c->vox_kx.reserve(n);
  c->vox_ky.reserve(n);
  if (c->h_obs.reserve(2 * std::max<size_t>(n, 1)) != KC_OK) return;
  for (size_t i = 0; i < n; ++i) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    add_voxel(c, x, y, z);
    float o[3];
    c->obs_tf.apply(x, y, c->raw_is_scan ? 0.0f : z, o);
    c->h_obs.p[i] = o[0];
    c->h_obs.p[n + i] = o[1];
  }
  c->host_lists_valid = true;
}
inline int ensure_host_lists(kc_dwa *c) {
  if (c->host_lists_valid) return KC_OK;
  if (c->raw_on_device) {
    // the list of a grid hand-off never left the device: fetch it now
    c->raw_xyz.resize(3 * c->raw_n);
    KC_HIP(hipMemcpyAsync(c->raw_xyz.data(), c->d_raw.p, 3 * c->raw_n * sizeof(float),
                          hipMemcpyDeviceToHost, c->stream));
    KC_HIP(hipStreamSynchronize(c->stream));
    c->raw_on_device = false;
  }
  build_host_lists(c, c->raw_xyz.data(), c->raw_xyz.size() / 3);
  return KC_OK;
}
// is there any occupied voxel column?  (after a device-side update the count is
// not known on the host: any point may be one)
inline bool any_voxel(const kc_dwa *c) {
  return c->host_lists_valid ? !c->vox_kx.empty() : c->O > 0;
}

// Sensor update on the device (kc_sensor_kernels.h): the host only bounds the
// cloud (one min/max pass), derives the bitmap extent and the bucket grid from
// the bounds, stores the raw points through the BAR and queues two kernels.
// *done = false: conditions not met, the caller takes the host path.
int sensor_update_device_bounded(kc_dwa *c, const float *xyz, size_t n, const float lo[3],
                                 const float hi[3], bool *done, bool raw_copied = false);

#if defined(__x86_64__)
inline bool cpu_has_avx512f() {
  static const bool v = __builtin_cpu_supports("avx512f");
  return v;
}
// The head of the bounds + copy pass of sensor_update_device with 64-byte vectors: floats [0, 48 k) of src are
// stored to dst (non-temporal: dst is device memory behind the BAR) and folded into min / max accumulators laid
// out like the 16-byte loop's (acc[0|1][m]: the SSE vector m = 0..2 of the 12-float period); *ok = false when a
// value is not finite.  Returns the number of floats done (a multiple of 48: the loop that follows continues in
// phase).
__attribute__((target("avx512f"))) size_t bounds_copy_avx512(const float *src, float *dst, size_t total, float acc[2][3][4],
                                                             bool *ok) {
  const __m512 big = _mm512_set1_ps(FLT_MAX);
  __m512 mn[3] = {big, big, big}, mx[3] = {_mm512_sub_ps(_mm512_setzero_ps(), big), _mm512_sub_ps(_mm512_setzero_ps(), big),
                                           _mm512_sub_ps(_mm512_setzero_ps(), big)};
  __mmask16 bad = 0;
  size_t i = 0;
  for (; i + 48 <= total; i += 48) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const __m512 v = _mm512_loadu_ps(src + i + 16 * q);
      _mm512_stream_ps(dst + i + 16 * q, v);
      mn[q] = _mm512_min_ps(mn[q], v);
      mx[q] = _mm512_max_ps(mx[q], v);
      const __m512 d = _mm512_sub_ps(v, v);
      bad |= _mm512_cmp_ps_mask(d, d, _CMP_UNORD_Q);
    }
  }
  // 64-byte vector q, 16-byte lane l = SSE vector (4 q + l) of the stream: period 3
  for (int m = 0; m < 3; ++m)
    for (int k = 0; k < 4; ++k) {
      acc[0][m][k] = FLT_MAX;
      acc[1][m][k] = -FLT_MAX;
    }
  alignas(64) float lo[16], hi[16];
  for (int q = 0; q < 3; ++q) {
    _mm512_store_ps(lo, mn[q]);
    _mm512_store_ps(hi, mx[q]);
    for (int l = 0; l < 4; ++l) {
      const int m = (4 * q + l) % 3;
      for (int k = 0; k < 4; ++k) {
        acc[0][m][k] = std::min(acc[0][m][k], lo[4 * l + k]);
        acc[1][m][k] = std::max(acc[1][m][k], hi[4 * l + k]);
      }
    }
  }
  *ok = bad == 0;
  return i;
}
#endif

int sensor_update_device(kc_dwa *c, const float *xyz, size_t n, bool *done) {
  *done = false;
  c->raw_on_device = false;
  if (!c->device_sensor || !c->trig_direct || c->prm.shape == KC_SPHERE ||
      n == 0 || n > kSensorDeviceMax)
    return KC_OK;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  size_t nfin = 0;
  bool bounded = false, raw_copied = false;
#if defined(__x86_64__)
  // This pass sits on the critical path of a sensor update (nothing is launched
  // before the bounds are known): four points per step with SSE min / max;
  // any non-finite coordinate (v - v != 0) sends the whole list to the loop below.
  // The same pass stores the points to their device buffer through the BAR
  // (write-combining stores): one trip over the list instead of two.
  KC_TRY(c->d_raw.reserve(3 * n + 16));
  {
    float *dst = c->d_raw.p;
    typedef float v4 __attribute__((vector_size(16)));
    typedef int v4i __attribute__((vector_size(16)));
    const v4 big = {FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX}, zero = {0.f, 0.f, 0.f, 0.f};
    v4 mn[3] = {big, big, big}, mx[3] = {-big, -big, -big};
    v4i ok = {-1, -1, -1, -1};
    const size_t total = 3 * n;
    size_t i = 0;
    if (total >= 96 && cpu_has_avx512f()) {
      // 48 floats (16 points) per step as three 64-byte vectors: a write-combining store per cache line
      float acc[2][3][4];
      bool ok512 = true;
      i = bounds_copy_avx512(xyz, dst, total, acc, &ok512);
      for (int q = 0; q < 3; ++q) {
        std::memcpy(&mn[q], acc[0][q], sizeof(v4));
        std::memcpy(&mx[q], acc[1][q], sizeof(v4));
      }
      if (!ok512) ok = v4i{0, 0, 0, 0};
    }
    for (; i + 12 <= total; i += 12) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        v4 v;
        std::memcpy(&v, xyz + i + 4 * q, sizeof(v));
        __builtin_nontemporal_store(v, reinterpret_cast<v4 *>(dst + i + 4 * q));
        mn[q] = __builtin_ia32_minps(mn[q], v);
        mx[q] = __builtin_ia32_maxps(mx[q], v);
        const v4 dv = v - v;
        ok &= (dv == zero);
      }
    }
    if ((ok[0] & ok[1] & ok[2] & ok[3]) != 0) {
      // lanes: v0 = x0 y0 z0 x1 | v1 = y1 z1 x2 y2 | v2 = z2 x3 y3 z3
      static const int vec_of[3][4] = {{0, 0, 1, 2}, {0, 1, 1, 2}, {0, 1, 2, 2}};
      static const int lane_of[3][4] = {{0, 3, 2, 1}, {1, 0, 3, 2}, {2, 1, 0, 3}};
      for (int a = 0; a < 3; ++a)
        for (int q = 0; q < 4; ++q) {
          lo[a] = std::min(lo[a], mn[vec_of[a][q]][lane_of[a][q]]);
          hi[a] = std::max(hi[a], mx[vec_of[a][q]][lane_of[a][q]]);
        }
      bool tail_ok = true;
      for (; i < total; ++i) {  // fewer than four points
        const float v = xyz[i];
        dst[i] = v;
        tail_ok = tail_ok && std::isfinite(v);
        lo[i % 3] = std::min(lo[i % 3], v);
        hi[i % 3] = std::max(hi[i % 3], v);
      }
      raw_copied = true;
      if (tail_ok) {
        bounded = true;
        nfin = n;
      } else {
        for (int a = 0; a < 3; ++a) {
          lo[a] = FLT_MAX;
          hi[a] = -FLT_MAX;
        }
      }
    }
  }
#endif
  for (size_t i = 0; i < n && !bounded; ++i) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    if (!(std::isfinite(x) && std::isfinite(y) && std::isfinite(z))) continue;
    lo[0] = std::min(lo[0], x);
    hi[0] = std::max(hi[0], x);
    lo[1] = std::min(lo[1], y);
    hi[1] = std::max(hi[1], y);
    lo[2] = std::min(lo[2], z);
    hi[2] = std::max(hi[2], z);
    ++nfin;
  }
  if (nfin == 0) return KC_OK;
  return sensor_update_device_bounded(c, xyz, n, lo, hi, done, raw_copied);
}

// the part behind the bounds; xyz == nullptr: the points are in d_raw already
// (grid hand-off)
// The trig job of a sensor update (SensorArgs::trig): only for a context that has run a cycle (the horizon), whose
// lattice is on the device and whose yaw chain stays inside the range of kc_trig_exact.h.
int plan_trig_job(kc_dwa *c, TrigJob &j) {
  j = TrigJob{};
  c->trig_ahead_valid = false;  // (whatever follows overwrites or outdates the table)
  const size_t A = c->lat.omega_values.size(), P = c->P;
  if (!c->trig_plan || !c->device_trig || !trig_selfcheck_ok() || A == 0 || P < 2 || !c->d_omega.p || c->d_omega.cap < A ||
      !std::isfinite(c->trig_plan_yaw))
    return KC_OK;
  const double dt = static_cast<double>(static_cast<float>(c->prm.time_step));
  double om_max = 0.0;
  for (double v : c->lat.omega_values) om_max = std::max(om_max, std::fabs(v));
  const double reach = std::fabs(c->trig_plan_yaw) + om_max * dt * static_cast<double>(P);
  if (!(reach < 1.0e8)) return KC_OK;
  KC_TRY(c->d_trig.reserve(A * P));
  if (!c->d_sincostab.p) {
    KC_TRY(c->d_sincostab.reserve(440));
    KC_HIP(hipMemcpyAsync(c->d_sincostab.p, kc_sincostab_host, sizeof(kc_sincostab_host), hipMemcpyHostToDevice, c->stream));
    KC_HIP(hipStreamSynchronize(c->stream));
  }
  j.yaw0 = c->trig_plan_yaw;
  j.dt = dt;
  j.omega = c->d_omega.p;
  j.tab = c->d_sincostab.p;
  j.out = c->d_trig.p;
  j.A = static_cast<int>(A);
  j.P = static_cast<int>(P);
  j.nblk = static_cast<int>(std::min<size_t>(32, (A * P + kSensorBlock - 1) / kSensorBlock));
  c->trig_ahead_yaw = c->trig_plan_yaw;
  c->trig_ahead_P = P;
  c->trig_ahead_lat = c->lat_version;
  return KC_OK;
}

int sensor_update_device_bounded(kc_dwa *c, const float *xyz, size_t n, const float lo[3],
                                 const float hi[3], bool *done, bool raw_copied) {
  *done = false;
  if (!c->device_sensor || !c->trig_direct || c->prm.shape == KC_SPHERE ||
      n == 0 || n > kSensorDeviceMax)
    return KC_OK;
  // bitmap: keys of the bounds (points beyond the 16-level octree are dropped
  // by add_voxel anyway)
  auto key = [&](float v) {
    const double f = std::floor(c->inv_res * static_cast<double>(v));
    return static_cast<int>(std::min(std::max(f, -32768.0), 32767.0));
  };
  bool fits = false;
  KC_TRY(bitmap_extent(c, key(lo[0]), key(lo[1]), key(hi[0]), key(hi[1]), &fits));
  const size_t nwords = fits ? static_cast<size_t>(c->gH) * c->gwpr : 0;
  if (!fits) {
    c->have_gbits = false;
      return KC_OK;
  }
  // one workgroup with everything in LDS, or (large clouds / bitmaps) the points
  // over many workgroups with device atomics
  const bool big_only = c->sensor_two_launch;  // option "sensor_two_launch": the build for clouds beyond kSensorFusedMax, for any size (tests)
  // bucket grid: covers the image of the bounding box (an affine map takes the
  // box into the hull of its eight transformed corners)
  double blo[2] = {DBL_MAX, DBL_MAX}, bhi[2] = {-DBL_MAX, -DBL_MAX};
  for (int k = 0; k < 8; ++k) {
    float o[3];
    const float zc = c->raw_is_scan ? 0.0f : ((k & 4) ? hi[2] : lo[2]);
    c->obs_tf.apply((k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], zc, o);
    if (!std::isfinite(o[0]) || !std::isfinite(o[1])) return KC_OK;
    blo[0] = std::min(blo[0], static_cast<double>(o[0]));
    bhi[0] = std::max(bhi[0], static_cast<double>(o[0]));
    blo[1] = std::min(blo[1], static_cast<double>(o[1]));
    bhi[1] = std::max(bhi[1], static_cast<double>(o[1]));
  }
  const double ext0 = std::max(bhi[0] - blo[0], bhi[1] - blo[1]);
  const double margin = 1e-4 * ext0 + 1e-4;  // float rounding of the transformed points
  blo[0] -= margin;
  blo[1] -= margin;
  bhi[0] += margin;
  bhi[1] += margin;
  BucketDev &b = c->bucket;
  std::memset(&b, 0, sizeof(b));
  b.cap = static_cast<double>(c->max_obs_dist) * 1.001;
  const int side = std::min(64, std::max(8, static_cast<int>(std::ceil(std::sqrt(
                                                static_cast<double>(n))))));
  const double ext = std::max(bhi[0] - blo[0], bhi[1] - blo[1]);
  b.g = std::max(0.125, ext / (side - 1));
  b.inv_g = 1.0 / b.g;
  b.gx0 = blo[0];
  b.gy0 = blo[1];
  b.W = std::min(side, static_cast<int>((bhi[0] - blo[0]) * b.inv_g) + 1);
  b.H = std::min(side, static_cast<int>((bhi[1] - blo[1]) * b.inv_g) + 1);
  const size_t ncell = static_cast<size_t>(b.W) * b.H;
  KC_TRY(c->d_cells.reserve(ncell + 1));
  KC_TRY(c->d_skip.reserve(ncell + 4));
  KC_TRY(c->d_bobs.reserve(2 * n));
  KC_TRY(c->d_raw.reserve(3 * n + 16));
  // the raw points: host copy for the lazy lists, device copy through the BAR
  c->host_lists_valid = false;
  if (xyz) {
    // (no host copy: the lists that the split path and the debug getters need are rebuilt from the
    // device copy on demand, ensure_host_lists)
    const auto tb0 = std::chrono::steady_clock::now();
    if (!raw_copied) std::memcpy(c->d_raw.p, xyz, 3 * n * sizeof(float));
    c->bar_dirty = true;
    bar_flush(c);
    if (c->hprof.on)
      std::fprintf(stderr, "[kc host] raw points over the BAR: %zu bytes in %.1f us\n", 3 * n * sizeof(float),
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tb0).count());
  }
  c->raw_xyz.clear();
  c->raw_on_device = true;
  c->raw_n = n;
  SensorArgs a{};
  a.xyz = c->d_raw.p;
  a.n = static_cast<int>(n);
  a.inv_res = c->inv_res;
  a.res = c->res;
  a.zc = -static_cast<double>(c->frame.t[2]);
  a.half_height = c->height / 2.0;
  a.gkx0 = c->gkx0;
  a.gky0 = c->gky0;
  a.gH = c->gH;
  a.gwpr = c->gwpr;
  a.gbits = c->d_gbits.p;
  for (int r = 0; r < 3; ++r) {
    for (int q = 0; q < 3; ++q) a.R[r][q] = c->obs_tf.R[r][q];
    a.t[r] = c->obs_tf.t[r];
  }
  a.gx0 = b.gx0;
  a.gy0 = b.gy0;
  a.inv_g = b.inv_g;
  a.W = b.W;
  a.H = b.H;
  a.cell_start = c->d_cells.p;
  a.skip = c->d_skip.p;
  a.bx = c->d_bobs.p;
  a.by = c->d_bobs.p + n;
  a.obs_z_zero = c->raw_is_scan ? 1 : 0;
  KC_TRY(plan_trig_job(c, a.trig));
  const unsigned tj = static_cast<unsigned>(a.trig.nblk);
  if (tj) {
    c->trig_ahead_valid = true;
    ++c->trig_rides;
  }
  // One launch, no hand-over between workgroups (sensor_fused_kernel): every workgroup reads all points and keeps
  // its part -- bands of the bitmap with their dilations, slices of the bucket tables.  Beyond 32 k points (every
  // workgroup reading every point stops being free) or with bands that do not fit LDS: the two-launch build.
  const DilGeom dg = dil_geom(c);
  const int dilR = c->have_dil ? dg.R : -1;
  int nb = std::min(64, c->gH), band_rows = (c->gH + nb - 1) / nb;
  // (LDS of a band: its rows + R rows of halo either side, and the two dilation accumulators of its own rows)
  auto band_bytes = [&] { return (3 * static_cast<size_t>(band_rows) + 2 * static_cast<size_t>(std::max(dilR, 0))) * c->gwpr * 4; };
  while (band_bytes() > kSensorFusedLds && band_rows > 1) {
    band_rows = (band_rows + 1) / 2;
  }
  nb = (c->gH + band_rows - 1) / band_rows;
  const bool fused = !big_only && c->sensor_fused_ok && n <= kSensorFusedMax && band_bytes() <= kSensorFusedLds && nb <= 1024;
  bool masks_built = false;
  if (fused) {
    SensorFusedArgs f{};
    f.a = a;
    f.nb = nb;
    f.kb = 8;
    f.band_rows = band_rows;
    f.R = dilR;
    f.ginner = c->d_ginner.p;
    f.gouter = c->d_gouter.p;
    if (dilR >= 0) dil_tables(dg, f.win, f.wout);
    // bucket workgroup: cell slots + row masks + (lists of more than one trip) a position per cell
    const size_t bucket_lds = ((ncell + 4) & ~size_t(3)) * 4 + 64 * 8 + ((ncell + 3) & ~size_t(3)) * 4;
    // float estimate of the cell index (sensor_obstacle_fast): its distance from the double expression
    {
      const double span = std::max(std::fabs(b.gx0), std::fabs(b.gy0)) + 64.0 * b.g;  // largest |coordinate| inside the grid
      const double ulp = span * 1.2e-7;                                                  // float spacing there
      const double err = (2.0 * ulp) * b.inv_g + 66.0 * 2.4e-7;                          // origin + difference, scaled; product rounding
      f.gx0f = static_cast<float>(b.gx0);
      f.gy0f = static_cast<float>(b.gy0);
      f.inv_gf = static_cast<float>(b.inv_g);
      f.id_eps = static_cast<float>(std::min(0.5, 8.0 * err));
    }
    // a band's y interval (sensor_band_body's first filter): keys gky0 + rows, padded by a voxel and the float rounding of y
    f.band_y0 = static_cast<float>(static_cast<double>(c->gky0) * c->res);
    f.band_dy = static_cast<float>(static_cast<double>(band_rows) * c->res);
    f.band_pad = static_cast<float>((static_cast<double>(std::max(dilR, 0)) + 2.0) * c->res +
                                    1e-5 * (std::fabs(static_cast<double>(c->gky0)) + c->gH) * c->res);
    size_t lds = std::max(band_bytes(), bucket_lds) + 16;
    const size_t olds = 2 * static_cast<size_t>(c->onear_args.n) * sizeof(float);
    const bool ride = c->onear_ahead && olds <= kObsNearLdsMax;
    if (ride) {
      const int cells = c->onear_args.W * c->onear_args.H, per = kSensorBlock / kObsNearLanes;
      f.o = c->onear_args;
      f.o_blocks = (cells + per - 1) / per;
      lds = std::max(lds, olds);
      c->onear_version = c->sensor_version;
      ++c->onear_rides;
    }
#ifdef KC_PHASE_STAMPS
    if (c->debug_stamps) {
      KC_TRY(c->d_dbg.reserve(512 * 16));
      KC_HIP(hipMemsetAsync(c->d_dbg.p, 0, 512 * 16 * 8, c->stream));
      f.dbg = c->d_dbg.p;
    }
#endif
    KC_TRY(c->timing.start("sensor_fused_kernel", c->stream));
    hipLaunchKernelGGL(sensor_fused_kernel<true>, dim3(f.nb + f.kb + f.o_blocks + tj), dim3(kSensorBlock), lds, c->stream, f);
    KC_TRY(c->timing.stop(c->stream));
#ifdef KC_PHASE_STAMPS
    if (f.dbg && (++c->sensor_stamp_calls % 100) == 50) {
      const int G = std::min(512, f.nb + f.kb);
      std::vector<unsigned long long> h(static_cast<size_t>(G) * 16);
      KC_HIP(hipStreamSynchronize(c->stream));
      KC_HIP(hipMemcpy(h.data(), c->d_dbg.p, h.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull;
      for (int r = 0; r < G; ++r) if (h[r * 16]) t0 = std::min(t0, h[r * 16]);
      auto dump = [&](const char *what, int r0, int r1, const char *const *nm, int cnt) {
        std::fprintf(stderr, "[kc stamps] sensor_fused_kernel %s, us since the first workgroup (avg / max):\n", what);
        for (int k = 0; k < cnt; ++k) {
          double sm = 0, mx = 0; int m = 0;
          for (int r = r0; r < r1; ++r) {
            if (!h[r * 16 + k]) continue;
            const double us = (h[r * 16 + k] - t0) / 100.0;
            sm += us; mx = std::max(mx, us); ++m;
          }
          if (m) std::fprintf(stderr, "  %-18s %6.2f / %6.2f\n", nm[k], sm / m, mx);
        }
      };
      static const char *bn[5] = {"start", "lds zero", "points", "dilated", "rows out"};
      static const char *kn[7] = {"start", "lds zero", "counted", "scanned", "masks + pos", "slice out", "placed"};
      dump("bands", 0, std::min(G, f.nb), bn, 5);
      dump("buckets", f.nb, G, kn, 7);
    }
#endif
    masks_built = true;
  } else {
    {  // byte map of the voxels: zero between updates (sensor_place_kernel clears what it packs)
      const uint8_t *was = c->d_sensor_bytes.p;
      KC_TRY(c->d_sensor_bytes.reserve(nwords * 32));
      if (c->d_sensor_bytes.p != was)
        KC_HIP(hipMemsetAsync(c->d_sensor_bytes.p, 0, c->d_sensor_bytes.cap, c->stream));
    }
    // scratch: [cell records n | ox n | oy n | histogram rows]
    SensorBigArgs sb{};
    sb.a = a;
    sb.ppt = static_cast<int>((n + static_cast<size_t>(kHistRowsMax) * kSensorBlock - 1) / (static_cast<size_t>(kHistRowsMax) * kSensorBlock));
    sb.rows = static_cast<int>(blocks_for(n, static_cast<size_t>(kSensorBlock) * sb.ppt));
    KC_TRY(c->d_sensor_tmp.reserve(3 * n + static_cast<size_t>(sb.rows) * kHistRow + 4));
    sb.tcell = reinterpret_cast<int *>(c->d_sensor_tmp.p);
    sb.tox = reinterpret_cast<float *>(sb.tcell + n);
    sb.toy = sb.tox + n;
    sb.hist = reinterpret_cast<int *>((reinterpret_cast<uintptr_t>(sb.toy + n) + 15) & ~uintptr_t(15));
    sb.bytes = c->d_sensor_bytes.p;
#ifdef KC_PHASE_STAMPS
    if (c->debug_stamps) {
      KC_TRY(c->d_dbg.reserve(512 * 16));
      KC_HIP(hipMemsetAsync(c->d_dbg.p, 0, 16 * 16 * 8, c->stream));
      sb.dbg = c->d_dbg.p;
    }
#endif
    KC_TRY(c->timing.start("sensor_points_kernel", c->stream));
    hipLaunchKernelGGL(sensor_points_kernel, dim3(sb.rows + tj), dim3(kSensorBlock), 0, c->stream, sb);
    KC_TRY(c->timing.stop(c->stream));
    KC_TRY(c->timing.start("sensor_place_kernel", c->stream));
    const unsigned pack_blocks = std::min(240u, blocks_for(nwords, kSensorBlock));  // pack-only workgroups behind the rows
    hipLaunchKernelGGL(sensor_place_kernel, dim3(sb.rows + pack_blocks), dim3(kSensorBlock), 0, c->stream, sb);
    KC_TRY(c->timing.stop(c->stream));
#ifdef KC_PHASE_STAMPS
    if (sb.dbg) {
      std::vector<unsigned long long> h(16 * 16);
      KC_HIP(hipStreamSynchronize(c->stream));
      KC_HIP(hipMemcpy(h.data(), c->d_dbg.p, h.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull;
      for (int r = 0; r < 16; ++r) if (h[r * 16]) t0 = std::min(t0, h[r * 16]);
      static const char *nm[12] = {"points: start", "lds zero", "points done", "row out", "place: start", "sums", "scan", "masks",
                                   "cells", "placed", "pack: start", "pack: end"};
      std::fprintf(stderr, "[kc stamps] sensor build, us since the first points workgroup (avg / max over workgroups):\n");
      for (int k = 0; k < 12; ++k) {
        double sm = 0, mx = 0; int cnt = 0;
        for (int r = 0; r < 16; ++r) {
          if (!h[r * 16 + k]) continue;
          const double us = (h[r * 16 + k] - t0) / 100.0;
          sm += us; mx = std::max(mx, us); ++cnt;
        }
        if (cnt) std::fprintf(stderr, "  %-14s %6.2f / %6.2f\n", nm[k], sm / cnt, mx);
      }
    }
#endif
  }
  KC_HIP(hipGetLastError());
  c->update_busy = true;
  if (!masks_built) KC_TRY(launch_dilate(c));  // (sensor_fused_kernel writes both dilations beside the bitmap)
  c->have_gbits = true;
  b.skip = c->d_skip.p;
  b.cell_start = c->d_cells.p;
  b.bx = c->d_bobs.p;
  b.by = c->d_bobs.p + n;
  b.nobs = static_cast<int>(n);  // upper bound: the tail of each half is never indexed
  c->O = n;
  c->n_bucketed = n;
  *done = true;
  return KC_OK;
}

// shard-local sample ids ordered by trig row (stable): consecutive samples of a
// fused workgroup then share one or two rows of the table.  The single-launch
// cycle gets a second order, dealt from the first: survivors of the collision
// gate cluster (a few adjacent omega rows, the low speeds of each), and a
// workgroup costs its own survivors, so a cluster of any shape has to land on
// many workgroups instead of a few (see below).
int build_perm(kc_dwa *c) {
  const size_t n = c->shard_count, first = c->shard_first;
  c->perm_valid = true;
  c->perm_first = first;
  c->perm_count = n;
  if (n == 0) return KC_OK;
  c->h_perm.resize(n);
  for (size_t i = 0; i < n; ++i) c->h_perm[i] = static_cast<int32_t>(i);
  const int32_t *row = c->lat.row.data() + first;
  std::stable_sort(c->h_perm.begin(), c->h_perm.end(),
                   [row](int32_t x, int32_t y) { return row[x] < row[y]; });
  std::vector<int32_t> &dealt = c->h_dealt;
  dealt.clear();
  dealt.reserve(n);
  {
    // Rectangular lattice (R trig rows of L samples each -- the non-holonomic
    // window is one): a workgroup takes 8 rows, R/8 apart, and 4 samples of each,
    // L/4 apart.  Survivors cluster in adjacent rows and adjacent speeds, so at
    // most a couple land in one workgroup, and a workgroup reads 8 rows of the
    // trig table instead of 32.  Anything else (omni windows, ragged shards): the
    // skewed stride, one sample per row.
    size_t R = 0, L = 0;
    bool rect = true;
    for (size_t i = 0; i < n && rect;) {
      size_t j = i;
      while (j < n && row[c->h_perm[j]] == row[c->h_perm[i]]) ++j;
      if (R == 0) L = j - i;
      rect = (j - i) == L;
      ++R;
      i = j;
    }
    const size_t cs = static_cast<size_t>(c->cycle_samples);  // 32: 8 rows x 4 samples, 16: 4 x 4
    const size_t rows_per = cs / 4;
    rect = rect && R * L == n && R % rows_per == 0 && L % 4 == 0;
    if (rect) {
      const size_t A = R / rows_per, B = L / 4;
      for (size_t a = 0; a < A; ++a)
        for (size_t b = 0; b < B; ++b)
          for (size_t i = 0; i < rows_per; ++i)
            for (size_t k = 0; k < 4; ++k) dealt.push_back(c->h_perm[(a + A * i) * L + b + B * k]);
    } else {
      // Ragged rows (omni windows, shares dealt by row): quads again -- every row of the sorted order is cut
      // into groups of (up to) four samples a quarter of the row apart, and the quads are dealt with the skewed
      // stride (a workgroup takes its quads from cs / 4 regions of the sorted order: adjacent rows and adjacent
      // speeds go to different workgroups, and the samples of a workgroup share cs / 4 trig rows or a few more
      // instead of cs -- the rows its lanes have to form).  A workgroup is whatever cs consecutive
      // entries of the flat list are: partial quads only shift the boundaries.
      std::vector<int32_t> qstart, qstep, qcount;  // quad = h_perm[qstart + k * qstep], k < qcount
      for (size_t i = 0; i < n;) {
        size_t j = i;
        while (j < n && row[c->h_perm[j]] == row[c->h_perm[i]]) ++j;
        const size_t len = j - i, nq = (len + 3) / 4;
        for (size_t q = 0; q < nq; ++q) {
          qstart.push_back(static_cast<int32_t>(i + q));
          qstep.push_back(static_cast<int32_t>(nq));
          qcount.push_back(static_cast<int32_t>((len - q + nq - 1) / nq));  // elements q, q + nq, ... below len
        }
        i = j;
      }
      const size_t Q = qstart.size(), qper = cs / 4;
      const size_t G = (Q + qper - 1) / qper;
      for (size_t g = 0; g < G; ++g)
        for (size_t j = 0; j < qper; ++j) {
          const size_t e = j * G + (g + 37 * j) % G;
          if (e >= Q) continue;
          for (int32_t k = 0; k < qcount[e]; ++k) dealt.push_back(c->h_perm[static_cast<size_t>(qstart[e] + k * qstep[e])]);
        }
    }
  }
  c->perm_cs = c->cycle_samples;
  std::vector<int32_t> prow(n);
  std::vector<uint32_t> pvi(n);
  for (int pass = 0; pass < 2; ++pass) {
    const std::vector<int32_t> &order = pass == 0 ? c->h_perm : dealt;
    DevBuf<int32_t> &dperm = pass == 0 ? c->d_perm : c->d_cperm;
    DevBuf<int32_t> &drow = pass == 0 ? c->d_prow : c->d_cprow;
    DevBuf<uint32_t> &dvi = pass == 0 ? c->d_pvi : c->d_cpvi;
    KC_TRY(dperm.reserve(n));
    KC_TRY(drow.reserve(n));
    KC_TRY(dvi.reserve(n));
    for (size_t i = 0; i < n; ++i) {
      const size_t g = first + static_cast<size_t>(order[i]);
      prow[i] = c->lat.row[g];
      pvi[i] = static_cast<uint32_t>(c->lat.ix[g]) | (static_cast<uint32_t>(c->lat.iy[g]) << 16);
    }
    KC_HIP(hipMemcpyAsync(dperm.p, order.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    KC_HIP(hipMemcpyAsync(drow.p, prow.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    KC_HIP(hipMemcpyAsync(dvi.p, pvi.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    KC_HIP(hipStreamSynchronize(c->stream));  // pageable sources
  }
  return KC_OK;
}

int build_window_at(kc_dwa *c, double wx, double wy, double reach, CollDev &cd);
int window_geometry(kc_dwa *c, double wx, double wy, double reach, CollDev &cd);
int window_bits_host(kc_dwa *c, CollDev &cd);

// every pose of every sample stays within this distance of the start
double cycle_reach(const kc_dwa *c) {
  const double dt = static_cast<double>(static_cast<float>(c->prm.time_step));
  return c->vmax_lin * dt * static_cast<double>(c->P) * 1.0001;
}

// frame + extent of the window of voxels within reach (+ robot bound) of
// (wx, wy); enabled = there is sensor data at all
int window_geometry(kc_dwa *c, double wx, double wy, double reach, CollDev &cd) {
  std::memset(&cd, 0, sizeof(cd));
  cd.shape = c->prm.shape;
  const hm::Rigid3f &F = c->frame;
  cd.r00 = F.R[0][0];
  cd.r01 = F.R[0][1];
  cd.r10 = F.R[1][0];
  cd.r11 = F.R[1][1];
  cd.tx = F.t[0];
  cd.ty = F.t[1];
  cd.res = c->res;
  cd.inv = 1.0 / c->res;
  cd.radius = c->radius;
  cd.rr = c->radius * c->radius;
  cd.a = static_cast<double>(c->prm.dims[0]) / 2.0;
  cd.b = static_cast<double>(c->prm.dims[1]) / 2.0;
  if (!c->have_sensor || !any_voxel(c)) return KC_OK;  // enabled = 0
  const double bound = (c->prm.shape == KC_BOX)
                           ? std::sqrt(cd.a * cd.a + cd.b * cd.b)
                           : c->radius;
  const double dx = wx - cd.tx, dy = wy - cd.ty;
  const double xf = cd.r00 * dx + cd.r10 * dy;
  const double yf = cd.r01 * dx + cd.r11 * dy;
  const long half = static_cast<long>(std::ceil((reach + bound) * cd.inv)) + 3;
  if (half > 8190)
    KC_FAIL(KC_ERR_RANGE,
            "reachable collision window of %ld cells per side is too large "
            "(octree resolution %g m, reach %g m)",
            2 * half + 1, c->res, reach + bound);
  cd.kx0 = static_cast<int>(std::floor(xf * cd.inv)) - static_cast<int>(half);
  cd.ky0 = static_cast<int>(std::floor(yf * cd.inv)) - static_cast<int>(half);
  cd.W = cd.H = static_cast<int>(2 * half + 1);
  if (c->have_gbits) {
    // shift the origin left to a word boundary of the sensor bitmap
    const long rel = static_cast<long>(cd.kx0) - c->gkx0;
    const long aligned = (rel >= 0 ? rel / 32 : -((-rel + 31) / 32)) * 32;
    cd.W += static_cast<int>(rel - aligned);
    cd.kx0 = static_cast<int>(c->gkx0 + aligned);
    cd.gbits = c->d_gbits.p;
    if (c->prm.shape == KC_SPHERE && c->gz_valid) {
      cd.gz = c->d_gz.p;
      cd.zlut = c->d_zlut.p;
      cd.zmode = 1;
      if (c->sphere_layers == 1) {
        cd.zmode = 2;
        cd.zconst = c->h_zlut.p[0];
      }
    }
    cd.ginner = c->d_ginner.p;
    cd.gouter = c->d_gouter.p;
    cd.dil = c->have_dil ? 1 : 0;
    cd.gkx0 = c->gkx0;
    cd.gky0 = c->gky0;
    cd.gH = c->gH;
    cd.gwpr = c->gwpr;
  }
  cd.wpr = (cd.W + 31) / 32;
  cd.enabled = 1;
  return KC_OK;
}

// host-built occupancy bits (+ sphere z gaps) of the window, uploaded to global
// memory: the path for windows that do not fit LDS, spheres and pose batches
int window_bits_host(kc_dwa *c, CollDev &cd) {
  if (!cd.enabled) return KC_OK;
  KC_TRY(ensure_host_lists(c));
  cd.enabled = 0;
  const size_t nwords = static_cast<size_t>(cd.H) * cd.wpr;
  KC_TRY(c->h_bits.reserve(nwords));
  std::memset(c->h_bits.p, 0, nwords * sizeof(uint32_t));
  const bool sphere = c->prm.shape == KC_SPHERE;
  if (sphere) {
    KC_TRY(c->h_ddz.reserve(static_cast<size_t>(cd.W) * cd.H));
    std::fill(c->h_ddz.p, c->h_ddz.p + static_cast<size_t>(cd.W) * cd.H,
              DBL_MAX);
  }
  size_t hits = 0;
  for (size_t i = 0; i < c->vox_kx.size(); ++i) {
    const long cx = static_cast<long>(c->vox_kx[i]) - cd.kx0;
    const long cy = static_cast<long>(c->vox_ky[i]) - cd.ky0;
    if (cx < 0 || cy < 0 || cx >= cd.W || cy >= cd.H) continue;
    c->h_bits.p[cy * cd.wpr + (cx >> 5)] |= 1u << (cx & 31);
    if (sphere) {
      double &g = c->h_ddz.p[cy * cd.W + cx];
      g = std::min(g, c->vox_ddz[i]);
    }
    ++hits;
  }
  if (hits == 0) return KC_OK;
  cd.enabled = 1;
  KC_TRY(c->d_bits.reserve(nwords));
  KC_HIP(hipMemcpyAsync(c->d_bits.p, c->h_bits.p, nwords * sizeof(uint32_t),
                        hipMemcpyHostToDevice, c->stream));
  cd.bits = c->d_bits.p;
  if (sphere) {
    const size_t nc = static_cast<size_t>(cd.W) * cd.H;
    KC_TRY(c->d_ddz.reserve(nc));
    KC_HIP(hipMemcpyAsync(c->d_ddz.p, c->h_ddz.p, nc * sizeof(double),
                          hipMemcpyHostToDevice, c->stream));
    cd.ddz = c->d_ddz.p;
  }
  cd.lds = (nwords * 4 <= 48 * 1024) ? 1 : 0;
  return KC_OK;
}

int build_window_at(kc_dwa *c, double wx, double wy, double reach, CollDev &cd) {
  KC_TRY(window_geometry(c, wx, wy, reach, cd));
  return window_bits_host(c, cd);
}

int ensure_cycle_buffers(kc_dwa *c, size_t n, size_t P) {
  KC_TRY(c->d_px.reserve(n * P));
  KC_TRY(c->d_py.reserve(n * P));
  KC_TRY(c->d_flags.reserve(n));
  KC_TRY(c->d_costs.reserve(n));
  KC_TRY(c->d_adm.reserve(n + 1));
  return KC_OK;
}

// Near table for the cycle that starts at (x, y): kept when the segment is the one it was built
// from and the reachable box still lies inside it.
// Near table over the box [lo, hi] (every query point of the coming cost stage lies inside): kept when
// the segment is the one it was built from and the box still lies inside it.
int ensure_near_table_box(kc_dwa *c, double lo_x, double lo_y, double hi_x, double hi_y, double margin) {
  c->near_ok = false;
  const bool use_seg = c->ref_len > 0.0f && (c->w.reference_path_distance_weight > 0.0 ||
                                             c->w.goal_distance_weight > 0.0);
  if (c->near_side == 0 || !use_seg || c->S == 0 || c->S >= 65536) return KC_OK;
  if (!std::isfinite(lo_x) || !std::isfinite(lo_y) || !std::isfinite(hi_x) || !std::isfinite(hi_y) ||
      !(hi_x >= lo_x) || !(hi_y >= lo_y))
    return KC_OK;
  const int N = c->near_side;
  if (c->near_version == c->seg_version && c->near_g > 0.f) {
    const double t_lo_x = c->near_x0, t_lo_y = c->near_y0, side = static_cast<double>(c->near_g) * N;
    if (lo_x >= t_lo_x && lo_y >= t_lo_y && hi_x <= t_lo_x + side && hi_y <= t_lo_y + side) {
      c->near_ok = true;
      return KC_OK;
    }
  }
  const double ext = std::max(hi_x - lo_x, hi_y - lo_y);
  const double pad = 0.01 * ext + 1e-3 + margin;
  c->near_x0 = static_cast<float>(lo_x - pad);
  c->near_y0 = static_cast<float>(lo_y - pad);
  // the float origins may have been rounded up: the edge covers that too
  const double side = std::max(hi_x + pad - c->near_x0, hi_y + pad - c->near_y0) * 1.0001;
  c->near_g = static_cast<float>(side / N);
  if (!(c->near_g > 0.f) || !std::isfinite(c->near_g) || !std::isfinite(1.0f / c->near_g)) return KC_OK;
  KC_TRY(c->d_near.reserve(static_cast<size_t>(N) * N));
  SegNearArgs na{};
  na.seg = c->d_seg.p;
  na.S = static_cast<int>(c->S);
  na.chunk = c->seg_chunk;
  na.nch = c->seg_nch;
  na.flat = c->seg_flat ? 1 : 0;
  na.x0 = c->near_x0;
  na.y0 = c->near_y0;
  na.g = c->near_g;
  // the kernels take a point's cell from (x - x0) * (1 / g) in float
  na.slack = static_cast<float>(1e-5 * side + 1e-6 * (std::fabs(c->near_x0) + std::fabs(c->near_y0) + side));
  na.W = na.H = N;
  na.out = c->d_near.p;
  KC_TRY(c->timing.start("segment_near_kernel", c->stream));
  {
    const dim3 grid((N * N + kSegNearBlock / kSegNearLanes - 1) / (kSegNearBlock / kSegNearLanes));
    const size_t lds = 32 * static_cast<size_t>(seg_pairs_padded(na.nch, na.chunk));
    if (lds <= kSegNearLdsMax && lds <= c->lds_limit_hw)
      hipLaunchKernelGGL(segment_near_kernel<true>, grid, dim3(kSegNearBlock), lds, c->stream, na);
    else
      hipLaunchKernelGGL(segment_near_kernel<false>, grid, dim3(kSegNearBlock), 0, c->stream, na);
  }
  KC_TRY(c->timing.stop(c->stream));
  c->near_version = c->seg_version;
  c->near_ok = true;
  return KC_OK;
}
// ... for the cycle that starts at (x, y): everything a roll-out can reach
int ensure_near_table(kc_dwa *c, double x, double y, double margin = 0.0) {
  c->near_ok = false;
  const double reach = cycle_reach(c);
  if (!(reach > 0.0) || !std::isfinite(reach) || !std::isfinite(x) || !std::isfinite(y)) return KC_OK;
  return ensure_near_table_box(c, x - reach, y - reach, x + reach, y + reach, margin);
}

// A new tracked segment while the cycles use the table: build the next one now, around the last start
// pose with room for the robot to have moved, so that the kernel runs under the host's preparation of
// the next cycle and under that cycle's launch latency instead of in front of its kernel.  The cycle
// keeps it when its reachable box lies inside (ensure_near_table), else builds its own.
int near_table_ahead(kc_dwa *c) {
  if (!c->near_wanted || c->P < 2) return KC_OK;
  const double reach = cycle_reach(c);
  KC_TRY(ensure_near_table(c, c->last_start.x, c->last_start.y, std::max(0.1 * reach, 0.25)));
  if (c->near_ok) c->seg_busy = true;  // a queued kernel reads the segment table: the next host write waits
  c->near_ok = false;                  // (the cycle decides)
  return KC_OK;
}

// The near table of the scan's obstacles over everything the cycle that starts at (x, y) can reach: kept while
// the sensor data stays and the box lies inside the table, else built (one launch, stream-ordered in front of
// the cost stage that reads it).
// geometry + argument block of a table over the box (x, y) +- reach; *ok = false: no table (degenerate box)
int onear_plan(kc_dwa *c, double x, double y, double reach, ObsNearArgs &oa, bool *ok) {
  *ok = false;
  const double lo_x = x - reach, lo_y = y - reach, hi_x = x + reach, hi_y = y + reach;
  const int N = c->onear_side;
  const double ext = 2.0 * reach;
  const double pad = 0.02 * ext + 1e-3;
  c->onear_x0 = static_cast<float>(lo_x - pad);
  c->onear_y0 = static_cast<float>(lo_y - pad);
  const double side = std::max(hi_x + pad - c->onear_x0, hi_y + pad - c->onear_y0) * 1.0001;
  c->onear_g = static_cast<float>(side / N);
  c->onear_version = ~0ull;
  if (!(c->onear_g > 0.f) || !std::isfinite(c->onear_g) || !std::isfinite(1.0f / c->onear_g)) return KC_OK;
  KC_TRY(c->d_onear.reserve(static_cast<size_t>(N) * N));
  oa = ObsNearArgs{};
  const size_t n = c->oscan_n;
  oa.osx = c->d_oscan.p;
  oa.osy = c->d_oscan.p + n;
  oa.aabb = c->d_oscan.p + 2 * n;
  oa.n = static_cast<int>(n);
  oa.cs = c->oscan_cs;
  oa.nch = c->oscan_nch;
  oa.x0 = c->onear_x0;
  oa.y0 = c->onear_y0;
  oa.g = c->onear_g;
  // the cost kernels take a point's cell from (x - x0) * (1 / g) in float
  oa.slack = static_cast<float>(1e-5 * side + 1e-6 * (std::fabs(c->onear_x0) + std::fabs(c->onear_y0) + side));
  oa.cap = c->max_obs_dist;
  oa.W = oa.H = N;
  oa.out = c->d_onear.p;
  *ok = true;
  return KC_OK;
}

bool onear_wanted(const kc_dwa *c) {
  return c->oscan_valid && c->w.obstacles_distance_weight > 0.0 && !c->external;
}

// kc_dwa_set_scan knows the pose the next cycle starts from: the table over what the LAST cycle's lattice and
// horizon reach from there (+ 15 %: the velocity window moves with the robot's speed) rides in the launch of
// the sensor tables (sensor_fused_kernel).  A cycle the guess does not cover builds its own.
int onear_plan_ahead(kc_dwa *c, double x, double y) {
  c->onear_ahead = false;
  if (!onear_wanted(c) || c->P < 2) return KC_OK;
  const double reach = cycle_reach(c) * 1.15;
  if (!(reach > 0.0) || !std::isfinite(reach) || !std::isfinite(x) || !std::isfinite(y)) return KC_OK;
  bool ok = false;
  KC_TRY(onear_plan(c, x, y, reach, c->onear_args, &ok));
  c->onear_ahead = ok;
  return KC_OK;
}

int ensure_onear(kc_dwa *c, double x, double y) {
  c->onear_ok = false;
  if (!onear_wanted(c)) return KC_OK;
  const double reach = cycle_reach(c);
  if (!(reach > 0.0) || !std::isfinite(reach) || !std::isfinite(x) || !std::isfinite(y)) return KC_OK;
  const double lo_x = x - reach, lo_y = y - reach, hi_x = x + reach, hi_y = y + reach;
  const int N = c->onear_side;
  if (c->onear_version == c->sensor_version && c->onear_g > 0.f) {
    const double t_lo_x = c->onear_x0, t_lo_y = c->onear_y0, side = static_cast<double>(c->onear_g) * N;
    if (lo_x >= t_lo_x && lo_y >= t_lo_y && hi_x <= t_lo_x + side && hi_y <= t_lo_y + side) {
      c->onear_ok = true;
      return KC_OK;
    }
  }
  ObsNearArgs oa{};
  bool ok = false;
  KC_TRY(onear_plan(c, x, y, reach, oa, &ok));
  if (!ok) return KC_OK;
  KC_TRY(c->timing.start("obs_near_kernel", c->stream));
  {
    const dim3 grid((N * N + kObsNearBlock / kObsNearLanes - 1) / (kObsNearBlock / kObsNearLanes));
    const size_t lds = 2 * static_cast<size_t>(oa.n) * sizeof(float);
    if (lds <= kObsNearLdsMax && lds <= c->lds_limit_hw)
      hipLaunchKernelGGL(obs_near_kernel<true>, grid, dim3(kObsNearBlock), lds, c->stream, oa);
    else
      hipLaunchKernelGGL(obs_near_kernel<false>, grid, dim3(kObsNearBlock), 0, c->stream, oa);
  }
  KC_TRY(c->timing.stop(c->stream));
  ++c->onear_builds;
  c->onear_version = c->sensor_version;
  c->onear_ok = true;
  c->update_busy = true;  // a queued kernel reads the scan tables: the next sensor update waits for it
  return KC_OK;
}

// argument blocks of the cost stage (stand-alone kernels and the cycle tail)
int build_cost_args(kc_dwa *c, size_t n, size_t first, CostArgs &ca, DcArgs &dt) {
  const size_t P = c->P;
  const bool use_path = c->ref_len > 0.0f &&
                        c->w.reference_path_distance_weight > 0.0;
  const bool use_goal = c->ref_len > 0.0f && c->w.goal_distance_weight > 0.0;
  if ((use_path || use_goal) && c->S == 0)
    KC_FAIL(KC_ERR_STATE, "tracked segment not set");
  const bool use_obs = c->O > 0 && c->w.obstacles_distance_weight > 0.0;
  const float *seg = c->d_seg.p;
  const size_t S = c->S;
  ca = CostArgs{};
  ca.n = static_cast<int>(n);
  ca.first = static_cast<int>(first);
  ca.P = static_cast<int>(P);
  ca.S = static_cast<int>(S);
  ca.O = static_cast<int>(c->O);
  ca.use_seg = (use_path || use_goal) ? 1 : 0;
  ca.use_obs = use_obs ? 1 : 0;
  ca.have_vel = c->have_vel ? 1 : 0;
  ca.px = c->d_px.p;
  ca.py = c->d_py.p;
  ca.flags = c->d_flags.p;
  ca.adm_list = c->d_adm.p;
  ca.adm_count = c->d_result.p + W_LIST;
  ca.sx = seg;
  ca.sy = seg + S;
  ca.sz = seg + 2 * S;
  ca.szz = seg + 3 * S;
  ca.acc_seg = seg + 4 * S;
  ca.seg_chunk = c->seg_chunk;
  ca.nch = c->seg_nch;
  ca.nsup = c->seg_nsup;
  ca.seg_flat = c->seg_flat ? 1 : 0;
  dt = DcArgs{};
  if (c->near_ok && ca.use_seg) {
    dt.near = c->d_near.p;
    dt.nx0 = c->near_x0;
    dt.ny0 = c->near_y0;
    dt.ninv = 1.0f / c->near_g;
    dt.nW = dt.nH = c->near_side;
  }
  if (c->onear_ok && ca.use_obs && c->oscan_valid) {
    const size_t on = c->oscan_n;
    dt.onear = c->d_onear.p;
    dt.ox0 = c->onear_x0;
    dt.oy0 = c->onear_y0;
    dt.oinv = 1.0f / c->onear_g;
    dt.oW = dt.oH = c->onear_side;
    dt.osx = c->d_oscan.p;
    dt.osy = c->d_oscan.p + on;
    dt.oaabb = c->d_oscan.p + 2 * on;
    dt.on = static_cast<int>(on);
    dt.ocs = c->oscan_cs;
    dt.onch = c->oscan_nch;
    dt.ocap = static_cast<double>(c->max_obs_dist);
  }
  dt.ounion = (c->bucket.W <= 64 && c->bucket.H <= 64) ? c->obs_union : 0;
  ca.seg_len = c->seg_len;
  ca.ref_len = c->ref_len;
  ca.b = c->bucket;
  ca.vvx = c->d_vvx.p;
  ca.vvy = c->d_vvy.p;
  ca.vom = c->d_vom.p;
  ca.max_obs_dist = c->max_obs_dist;
  ca.acc0 = c->prm.acc_limits[0];
  ca.acc1 = c->prm.acc_limits[1];
  ca.acc2 = c->prm.acc_limits[2];
  ca.w_path = c->w.reference_path_distance_weight;
  ca.w_goal = c->w.goal_distance_weight;
  ca.w_obs = c->w.obstacles_distance_weight;
  ca.w_smooth = c->w.smoothness_weight;
  ca.w_jerk = c->w.jerk_weight;
  ca.costs = c->d_costs.p;
  ca.result = c->d_result.p;
  if (!c->drop_samples && !c->external && c->d_frz.p) {
    ca.frz_smooth = c->d_frz.p;
    ca.frz_jerk = c->d_frz.p + c->n_roll;
  }
  return KC_OK;
}

int run_evaluate(kc_dwa *c, size_t n, size_t first) {
  const size_t P = c->P;
  hipStream_t s = c->stream;
  c->row_valid = false;
  c->slots_pending = false;
  c->device_record_valid = true;
  if (n == 0) {  // empty batch: publish "nothing found"
    hipLaunchKernelGGL(init_result_kernel, dim3(1), dim3(1), 0, s,
                       c->d_result.p);
    c->pub_pending = false;
    return KC_OK;
  }
  if (n > 1024u * kCompactMaxPer)
    KC_FAIL(KC_ERR_RANGE, "more than %d samples per context", 1024 * kCompactMaxPer);
  // Short admissible lists (the count of the previous cycle is the predictor)
  // go to the workgroup-per-sample kernel, long ones to the wavefront-per-
  // sample kernel; both are correct for any list.
  if (c->h_pub.p && c->seq > 0) {
    // callers that never fetch (multi-GPU: the key is all-reduced on the
    // device) still leave the previous cycle's record in the pinned mirror
    volatile long long *hp = c->h_pub.p;
    const long long w0 = hp[0], w1 = hp[1], w2 = hp[2], w3 = hp[3], w4 = hp[4];
    if (w2 == c->seq && w3 == record_check(w0, w1, w2, w4)) c->last_nadm = w1 >> 32;
  }
  bool use_block = c->last_nadm >= 0 && c->last_nadm <= kBlockKernelMaxAdm;
  if (c->cost_kernel_force == 1) use_block = true;
  if (c->cost_kernel_force == 2) use_block = false;
  // the wavefront-per-sample search of a roll-out's samples goes through the near table
  c->near_ok = false;
  c->near_wanted = !use_block && !c->external;
  c->onear_ok = false;
  if (c->near_wanted) {
    KC_TRY(ensure_near_table(c, c->last_start.x, c->last_start.y));
    KC_TRY(ensure_onear(c, c->last_start.x, c->last_start.y));
  }
  // caller-provided samples: the box found when they were uploaded
  if (!use_block && c->external && c->ext_box_valid)
    KC_TRY(ensure_near_table_box(c, c->ext_box[0], c->ext_box[1], c->ext_box[2], c->ext_box[3], 0.0));
  CostArgs ca{};
  DcArgs dt{};
  KC_TRY(build_cost_args(c, n, first, ca, dt));
  const size_t S = c->S;
  bool vel_beside = false;
  VelFinishArgs vf{};
  std::function<int()> vel_launch;
  if (ca.have_vel && (ca.w_smooth > 0.0 || ca.w_jerk > 0.0) && n == c->n_roll && first == 0) {
    // ordered sums of the velocity profiles.  One sample per wavefront inside the cost kernel while the
    // batch leaves a SIMD fewer than ~5 of these serial chains (latency bound either way); beyond, 4 samples
    // per wavefront in a pass of their own (a quarter of the chain instructions), 16 for batches that still
    // give every SIMD several chains then (tools/cost5k_terms.py)
    const int kinds = (ca.w_smooth > 0.0 ? 1 : 0) + (ca.w_jerk > 0.0 ? 1 : 0);
    const size_t simds = 4 * static_cast<size_t>(c->num_cus);
    int group = c->velocity_group;
    if (group == 0) group = kinds * n < 5 * simds ? 1 : (kinds * n < 96 * simds ? 4 : 16);
    if (group > 1) {
      KC_TRY(c->d_vsum.reserve(2 * n));
      VelSumArgs va{};
      va.vx = c->d_vvx.p;
      va.vy = c->d_vvy.p;
      va.om = c->d_vom.p;
      va.n = static_cast<int>(n);
      va.nv = static_cast<int>(P - 1);
      va.acc0 = ca.acc0;
      va.acc1 = ca.acc1;
      va.acc2 = ca.acc2;
      va.out[0] = c->d_vsum.p;
      va.out[1] = c->d_vsum.p + n;
      va.first_kind = ca.w_smooth > 0.0 ? 0 : 1;
      const dim3 grid(blocks_for(n, (kVelBlock / 64) * static_cast<size_t>(group)), kinds);
      // Beside the wavefront-per-sample cost kernel on a second stream: these chains leave most issue slots
      // of their SIMDs idle, the segment searches fill them (not while kernels are being timed one by one)
      vel_beside = !use_block && !c->timing.enabled && c->velocity_beside;
      hipStream_t vs = s;
      if (vel_beside) {
        if (!c->aux_stream) {
          KC_HIP(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
          KC_HIP(hipEventCreateWithFlags(&c->aux_fork, hipEventDisableTiming));
          KC_HIP(hipEventCreateWithFlags(&c->aux_join, hipEventDisableTiming));
        }
        vs = c->aux_stream;
        KC_HIP(hipEventRecord(c->aux_fork, s));  // behind everything queued so far (the last reader of d_vsum too)
        KC_HIP(hipStreamWaitEvent(vs, c->aux_fork, 0));
      }
      vel_launch = [=]() -> int {
        KC_TRY(c->timing.start("velocity_sums_kernel", vs));
        if (group == 4)
          hipLaunchKernelGGL(velocity_sums_kernel<16>, grid, dim3(kVelBlock), 0, vs, va);
        else
          hipLaunchKernelGGL(velocity_sums_kernel<4>, grid, dim3(kVelBlock), 0, vs, va);
        KC_TRY(c->timing.stop(vs));
        return KC_OK;
      };
      if (!vel_beside) KC_TRY(vel_launch());  // in front of the cost kernel, same stream
      if (vel_beside) {
        ca.defer_vel = 1;
        vf.adm_list = ca.adm_list;
        vf.adm_count = ca.adm_count;
        vf.costs = ca.costs;
        vf.vsum_smooth = ca.w_smooth > 0.0 ? va.out[0] : nullptr;
        vf.vsum_jerk = ca.w_jerk > 0.0 ? va.out[1] : nullptr;
        vf.w_smooth = ca.w_smooth;
        vf.w_jerk = ca.w_jerk;
        vf.div = static_cast<float>(3L * static_cast<long>(P - 1));
        vf.first = ca.first;
      } else {
        if (ca.w_smooth > 0.0) ca.vsum_smooth = va.out[0];
        if (ca.w_jerk > 0.0) ca.vsum_jerk = va.out[1];
      }
    }
  }
  // caller-provided batches: every sample is admissible (kc_cost_upload), the list is the identity
  if (c->external && n == c->n_roll && first == 0) {
    ca.identity_n = static_cast<int>(n);
    vf.identity_n = ca.identity_n;
  }
  if (c->need_compact && ca.identity_n == 0) {  // split roll-out path
    KC_TRY(c->timing.start("compact_kernel", s));
    hipLaunchKernelGGL(compact_kernel, dim3(1), dim3(1024), 0, s, c->d_flags.p,
                       static_cast<int>(n), c->d_adm.p, c->d_result.p + W_LIST);
    KC_TRY(c->timing.stop(s));
  }
  KC_TRY(c->d_block_keys.reserve(512));
  ca.block_keys = c->d_block_keys.p;
#ifdef KC_PHASE_STAMPS
  if (c->debug_stamps) {
    KC_TRY(c->d_dbg.reserve(512 * 16));
    KC_HIP(hipMemsetAsync(c->d_dbg.p, 0, 512 * 16 * 8, s));
    ca.dbg = c->d_dbg.p;
  }
#endif
  unsigned cost_blocks;
  size_t lds_tab = 0, lds_obs = 0;
  if (ca.use_obs) {
    const size_t ncell = static_cast<size_t>(ca.b.W) * ca.b.H;
    lds_tab += (ncell + 1) * sizeof(int) + ((ncell + 3) & ~size_t(3));
    lds_obs = 2 * static_cast<size_t>(ca.b.nobs) * sizeof(float);
  }
  PubArgs pa{};
  pa.block_keys = c->d_block_keys.p;
  pa.flags = c->d_flags.p;
  pa.n = static_cast<int>(n);
  pa.first = static_cast<int>(first);
  pa.result = c->d_result.p;
  pa.host_pub = c->h_pub.p;
  pa.seq = ++c->seq;
  pa.identity_n = ca.identity_n;
  // the long-list kernel publishes by itself (its last workgroup) unless the velocity sums finish behind it
  pa.fold = (!use_block && !vel_beside && c->fold_publish) ? 1 : 0;
  if (use_block) {
    KC_TRY(c->timing.start("sample_cost_block_kernel", s));
    cost_blocks = static_cast<unsigned>(std::min<size_t>(n, 512));
    size_t lds = (P * 3 * sizeof(float) + 15) & ~size_t(15);
    if (ca.use_seg) lds_tab += 5 * S * sizeof(float);
    const bool tab_lds = c->cost_lds_ok && lds + lds_tab + 64 <= kBlkLdsBudget;
    const bool obs_lds = tab_lds && ca.use_obs && lds + lds_tab + lds_obs + 64 <= kBlkLdsBudget;
    if (obs_lds)
      hipLaunchKernelGGL((sample_cost_block_kernel<true, true>), dim3(cost_blocks),
                         dim3(kBlkCostBlock), lds + lds_tab + lds_obs, s, ca);
    else if (tab_lds)
      hipLaunchKernelGGL((sample_cost_block_kernel<true, false>), dim3(cost_blocks),
                         dim3(kBlkCostBlock), lds + lds_tab, s, ca);
    else
      hipLaunchKernelGGL((sample_cost_block_kernel<false, false>), dim3(cost_blocks),
                         dim3(kBlkCostBlock), lds, s, ca);
  } else {
    // one workgroup per CU, sixteen samples (wavefronts) in flight in each
    cost_blocks = static_cast<unsigned>(std::min<size_t>(n, kCostGrid));
    pa.nblocks = static_cast<int>(cost_blocks);
    if (ca.use_seg)
      lds_tab += (8 * static_cast<size_t>(seg_pairs_padded(ca.nch, ca.seg_chunk)) + 8 * static_cast<size_t>(ca.nch) +
                  12 * static_cast<size_t>(ca.nsup)) * sizeof(float);  // pair records, capsules, spheres
    // batched per-sample part (two buffers of 64 samples in front of the tables): the DWA cycle's lists, and
    // caller-provided batches whose velocity sums are precomputed or not asked for
    const size_t lds_batch = 2 * batch_buf_bytes(static_cast<int>(P));
    const bool wave_sums = ca.have_vel && !ca.defer_vel &&
                           ((ca.w_smooth > 0.0 && !ca.vsum_smooth) || (ca.w_jerk > 0.0 && !ca.vsum_jerk));
    // ... and lists that fill more than one buffer per workgroup now and then (the last cycle's count is the
    // predictor; measured: 141 samples per workgroup -14 % kernel time, 50: -3 %, 18: +4 %, 10: +6 %)
    const long long expect = c->external ? static_cast<long long>(n) : (c->last_nadm >= 0 ? c->last_nadm : static_cast<long long>(n));
    const bool batched = c->cost_batch && c->cost_batch_ok && c->cost_lds_ok && !wave_sums &&
                         (c->cost_batch_forced || expect >= 40ll * kCostGrid) && lds_tab + lds_batch + 64 <= kCostLdsBudget;
    if (batched) lds_tab += lds_batch;
    const bool tab_lds = c->cost_lds_ok && lds_tab + 64 <= kCostLdsBudget;
    const bool obs_lds = tab_lds && ca.use_obs && lds_tab + lds_obs + 64 <= kCostLdsBudget && c->cost_obs_lds;
    if (c->debug_stamps && c->seq <= 2)
      std::fprintf(stderr, "[kc] cost kernel: tables=%zu obstacles=%zu nobs=%d grid=%dx%d S=%zu chunk=%d tab_lds=%d obs_lds=%d\n",
                   lds_tab, lds_obs, ca.b.nobs, ca.b.W, ca.b.H, S, ca.seg_chunk, int(tab_lds), int(obs_lds));
    KC_TRY(c->timing.start(batched ? "sample_cost_batched_kernel" : "sample_cost_kernel", s));
    if (batched && obs_lds)
      hipLaunchKernelGGL((sample_cost_batched_kernel<true>), dim3(cost_blocks), dim3(kCostBlock),
                         lds_tab + lds_obs, s, ca, dt, pa);
    else if (batched)
      hipLaunchKernelGGL((sample_cost_batched_kernel<false>), dim3(cost_blocks), dim3(kCostBlock),
                         lds_tab, s, ca, dt, pa);
    else {
      auto launch = [&](auto kernel, size_t lds) {
        hipLaunchKernelGGL(kernel, dim3(cost_blocks), dim3(kCostBlock), lds, s, ca, dt, pa);
      };
      if (pa.fold) {
        if (obs_lds) launch(sample_cost_kernel<true, true, true>, lds_tab + lds_obs);
        else if (tab_lds) launch(sample_cost_kernel<true, false, true>, lds_tab);
        else launch(sample_cost_kernel<false, false, true>, 0);
      } else {
        if (obs_lds) launch(sample_cost_kernel<true, true, false>, lds_tab + lds_obs);
        else if (tab_lds) launch(sample_cost_kernel<true, false, false>, lds_tab);
        else launch(sample_cost_kernel<false, false, false>, 0);
      }
    }
  }
  KC_TRY(c->timing.stop(s));
  if (vel_beside) {
    // queued BEHIND the cost kernel: its one-per-CU workgroups take their registers first, the chains' small
    // workgroups fill what is left (the other way round the cost kernel waits for CUs the chains have filled)
    KC_TRY(vel_launch());
    KC_HIP(hipEventRecord(c->aux_join, c->aux_stream));
    KC_HIP(hipStreamWaitEvent(s, c->aux_join, 0));
    cost_blocks = std::min(512u, blocks_for(n, 256));
    vf.block_keys = c->d_block_keys.p;
    hipLaunchKernelGGL(velocity_finish_kernel, dim3(cost_blocks), dim3(256), 0, s, vf);
  }
  c->pub_pending = true;
  if (!pa.fold) {
    pa.nblocks = static_cast<int>(cost_blocks);
    KC_TRY(c->timing.start("publish_kernel", s));
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(kPubBlock), 0, s, pa);
    KC_TRY(c->timing.stop(s));
  }
  // the kernel re-armed the list counter: a second evaluate of the same
  // roll-out has to rebuild the list from the flags
  c->list_dirty = false;
  c->need_compact = true;
  KC_HIP(hipGetLastError());
  return KC_OK;
}

// Single-GPU cycle without a device-side epilogue: wait for the slot of every
// workgroup (sequence number + checksum: the 32 bytes of a slot are two unfenced
// stores), then the reduction the last workgroup would have done -- minimum key,
// admissible count, the winner's index in the admissible-only numbering from
// the survivor masks and the dealt order this host built (build_perm).
int fetch_slots(kc_dwa *c, kc_result *out, size_t n) {
  const unsigned G = c->slots_G;
  volatile long long *hs = c->h_slots.p;
  const long long seq_mask = (1ll << 61) - 1;
  const auto t0 = std::chrono::steady_clock::now();
  bool synced = false;
  // Slots are taken in whatever order they arrive (a pending set, swept until it is empty) and
  // folded into the reduction at once: when the slowest workgroup reports, nothing else is left to do
  // but the index of the winner.
  std::vector<uint64_t> &pend = c->slot_pending;
  pend.assign((G + 63) / 64, ~0ull);
  if (G & 63) pend.back() = (1ull << (G & 63)) - 1ull;
  unsigned remaining = G;
  long long fkey = KEY_NONE;
  unsigned bw = 0;
  long long na = 0;
  for (long sweeps = 0; remaining; ++sweeps) {
    for (size_t w = 0; w < pend.size(); ++w) {
      for (uint64_t m = pend[w]; m;) {
        const unsigned g = static_cast<unsigned>(w * 64 + __builtin_ctzll(m));
        m &= m - 1;
        const long long w0 = hs[4 * g], w1 = hs[4 * g + 1], w2 = hs[4 * g + 2], w3 = hs[4 * g + 3];
        if ((w2 & seq_mask) != c->seq || w3 != record_check(w0, w1, w2, static_cast<long long>(g))) continue;
        pend[w] &= ~(1ull << (g & 63));
        --remaining;
        if (w0 < fkey || (w0 == fkey && g < bw)) {
          fkey = w0;
          bw = g;
        }
        na += __builtin_popcountll(static_cast<unsigned long long>(w1) & 0xFFFFFFFFull);
      }
    }
    if (remaining && (sweeps & 255) == 255 &&
        std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
      if (synced) KC_FAIL(KC_ERR_HIP, "%u workgroups of the cycle kernel never reported", remaining);
      KC_HIP(hipStreamSynchronize(c->stream));  // a kernel fault surfaces here
      synced = true;
    }
  }
  c->hprof.mark(6);
  c->slots_pending = false;
  c->drained = true;        // every workgroup is past its last table read
  c->update_busy = false;
  c->seg_busy = false;
  c->timing.mark("host:wait_result");
  kc_result r{};
  r.n_admissible = na;
  c->last_nadm = na;
  r.n_samples = static_cast<int64_t>(n);
  c->row_valid = false;
  if (fkey == KEY_NONE) {
    r.found = 0;
    r.cost = 0.0f;
    r.index = -1;
    r.raw_index = -1;
  } else {
    r.found = 1;
    r.cost = kc_key_cost(fkey);
    r.raw_index = kc_key_index(fkey);
    // admissible samples in front of the winner (generation order = local id order)
    const int lim = static_cast<int>(r.raw_index - static_cast<int64_t>(c->shard_first));
    const int32_t *ids = c->h_dealt.data();
    const size_t nd = c->h_dealt.size();
    long long cnt = 0;
    for (unsigned g = 0; g < G; ++g) {
      const uint32_t m = static_cast<uint32_t>(static_cast<unsigned long long>(c->h_slots.p[4 * g + 1]) & 0xFFFFFFFFull);
      if (!m) continue;
      const size_t cs = static_cast<size_t>(c->perm_cs);
      const size_t base = static_cast<size_t>(g) * cs;
      uint32_t below = 0u;
#if defined(__SSE2__)
      if (base + cs <= nd) {
        const __m128i vl = _mm_set1_epi32(lim);
        for (int q = 0; q < static_cast<int>(cs / 4); ++q) {
          const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(ids + base + 4 * q));
          below |= static_cast<uint32_t>(_mm_movemask_ps(_mm_castsi128_ps(_mm_cmplt_epi32(v, vl)))) << (4 * q);
        }
      } else
#endif
      {
        for (size_t s = 0; s < cs && base + s < nd; ++s)
          if (ids[base + s] < lim) below |= 1u << s;
      }
      cnt += __builtin_popcount(m & below);
    }
    r.index = cnt;
    // the winner's row: slot bw of the pinned row buffer, checked against the word of its slot
    const unsigned long long w1 = static_cast<unsigned long long>(c->h_slots.p[4 * bw + 1]);
    const bool has_row = (c->h_slots.p[4 * bw + 2] >> 61) & 1;
    const size_t nw = 2 * c->P;
    if (has_row && c->h_wrow.p && (static_cast<size_t>(bw) + 1) * nw <= c->h_wrow.cap) {
      const uint32_t want = static_cast<uint32_t>(w1 >> 32);
      const auto t1 = std::chrono::steady_clock::now();
      for (long spins = 0;; ++spins) {
        volatile uint32_t *row = c->h_wrow.p + bw * nw;
        uint32_t x = 0u;
        for (size_t q = 0; q < nw; ++q) x ^= row[q] * (2u * static_cast<uint32_t>(q) + 1u);
        if (x == want) {
          c->row_valid = true;
          c->wrow_off = bw * nw;
          break;
        }
        if ((spins & 63) == 63 && std::chrono::steady_clock::now() - t1 > std::chrono::milliseconds(20)) break;
      }
    }
  }
  c->last_lat = r.found ? r.raw_index : -1;
  if (r.found && !c->external) r.raw_index = global_of(c, r.raw_index);
  c->last = r;
  c->have_last = true;
  if (out) *out = r;
  return KC_OK;
}

int fetch(kc_dwa *c, kc_result *out, size_t n) {
  if (c->slots_pending) return fetch_slots(c, out, n);
  bool got = false;
  if (c->pub_pending) {
    // spin on the sequence word the last finalize block writes into pinned
    // host memory (bounded: fall back to a stream sync + D2H)
    volatile long long *hp = c->h_pub.p;
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 0;; ++spins) {
      const long long w0 = hp[0], w1 = hp[1], w2 = hp[2], w3 = hp[3], w4 = hp[4];
      if (w2 == c->seq && w3 == record_check(w0, w1, w2, w4)) {
        c->rec_w4 = w4;
        c->h_result.p[0] = w0;
        c->h_result.p[1] = w1 >> 32;  // n_admissible (-1: device error)
        c->h_result.p[2] = static_cast<long long>(static_cast<int32_t>(w1 & 0xFFFFFFFFll));
        got = true;
        c->drained = true;
        c->update_busy = false;  // queued in front of the cycle whose record just arrived
        c->seg_busy = false;
        break;
      }
      if ((spins & 1023) == 1023 &&
          std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200))
        break;
    }
    c->pub_pending = false;
  }
  if (!got) {
    KC_HIP(hipMemcpyAsync(c->h_result.p, c->d_result.p, 4 * sizeof(long long),
                          hipMemcpyDeviceToHost, c->stream));
    KC_HIP(hipStreamSynchronize(c->stream));
  }
  c->timing.mark("host:wait_result");
  kc_result r{};
  const long long key = c->h_result.p[0];
  if (c->h_result.p[1] < 0)
    KC_FAIL(KC_ERR_HIP, "the cycle's device error word is set");
  r.n_admissible = c->h_result.p[1];
  c->last_nadm = r.n_admissible;
  r.n_samples = static_cast<int64_t>(n);
  if (key == KEY_NONE) {
    r.found = 0;
    r.cost = 0.0f;
    r.index = -1;
    r.raw_index = -1;
  } else {
    r.found = 1;
    r.cost = kc_key_cost(key);
    r.raw_index = kc_key_index(key);
    r.index = c->h_result.p[2];
  }
  // winner row of a single-launch cycle: arrives in pinned memory beside the
  // record; its check word is part of the record (stores are not fenced: poll
  // until the words add up, bounded)
  c->row_valid = false;
  if (got && r.found && (c->rec_w4 & 1) && c->h_wrow.p) {
    const unsigned long long w4 = static_cast<unsigned long long>(c->rec_w4);
    const uint32_t want = static_cast<uint32_t>(w4 >> 32);
    const size_t bw = static_cast<size_t>((w4 & 0xFFFFFFFFull) >> 1);
    const size_t nw = 2 * c->P;
    if ((bw + 1) * nw <= c->h_wrow.cap) {
      const auto t0 = std::chrono::steady_clock::now();
      for (long spins = 0;; ++spins) {
        volatile uint32_t *row = c->h_wrow.p + bw * nw;
        uint32_t x = 0u;
        for (size_t q = 0; q < nw; ++q) x ^= row[q] * (2u * static_cast<uint32_t>(q) + 1u);
        if (x == want) {
          c->row_valid = true;
          c->wrow_off = bw * nw;
          break;
        }
        if ((spins & 63) == 63 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
          break;  // get_best falls back to the device copy
      }
    }
  }
  // (a key that came back from kc_dwa_allreduce_best may name another rank's sample)
  c->last_lat = r.found ? r.raw_index : -1;
  if (r.found && !c->external) {
    if (c->last_lat < static_cast<int64_t>(c->shard_first) ||
        c->last_lat >= static_cast<int64_t>(c->shard_first + c->n_roll))
      c->last_lat = -1;
    else
      r.raw_index = global_of(c, r.raw_index);
  }
  c->last = r;
  c->have_last = true;
  if (out) *out = r;
  return KC_OK;
}

}  // namespace

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

int kc_dwa_create(const kc_dwa_params *p, kc_dwa **out) {
  if (!p || !out) KC_FAIL(KC_ERR_INVALID, "null argument");
  *out = nullptr;
  if (p->shape != KC_CYLINDER && p->shape != KC_BOX && p->shape != KC_SPHERE)
    KC_FAIL(KC_ERR_INVALID, "Invalid robot geometry type");
  if (!(p->octree_res > 0.0) || !(p->time_step > 0.0))
    KC_FAIL(KC_ERR_INVALID, "octree_res and time_step must be positive");
  if (p->max_samples == 0 || p->max_points < 2)
    KC_FAIL(KC_ERR_INVALID, "max_samples >= 1 and max_points >= 2 required");
  if (p->max_samples > 0x7FFFFFFFu / std::max<size_t>(p->max_points, 1))
    KC_FAIL(KC_ERR_RANGE, "max_samples * max_points exceeds 2^31");
  int ndev = 0;
  KC_HIP(hipGetDeviceCount(&ndev));
  if (p->device < 0 || p->device >= ndev)
    KC_FAIL(KC_ERR_HIP, "HIP device %d not available (%d visible)", p->device,
            ndev);
  auto *c = new kc_dwa();
  c->prm = *p;
  for (int i = p->ndims; i < 3; ++i) c->prm.dims[i] = 0.0f;
  // collision_check.cpp:38-58
  if (p->shape == KC_CYLINDER) {
    c->radius = c->prm.dims[0];
    c->height = c->prm.dims[1];
  } else if (p->shape == KC_BOX) {
    c->height = c->prm.dims[2];
    c->radius = std::sqrt(std::pow(c->prm.dims[0], 2) +
                          std::pow(c->prm.dims[1], 2)) /
                2;
  } else {
    c->radius = c->prm.dims[0];
    c->height = 2 * c->prm.dims[0];
  }
  c->res = p->octree_res;
  c->inv_res = 1.0 / c->res;
  hm::Quat q{p->sensor_rot_xyzw[3], p->sensor_rot_xyzw[0],
             p->sensor_rot_xyzw[1], p->sensor_rot_xyzw[2]};
  c->sensor_tf_body = hm::Rigid3f::from_quat(q, p->sensor_pos);
  c->frame = c->sensor_tf_body;
  auto fail = [&](int rc) {
    kc_dwa_destroy(c);
    return rc;
  };
  if (hipSetDevice(p->device) != hipSuccess) {
    set_error("hipSetDevice(%d) failed", p->device);
    return fail(KC_ERR_HIP);
  }
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) !=
      hipSuccess) {
    set_error("hipStreamCreate failed");
    return fail(KC_ERR_HIP);
  }
  c->stream = c->own_stream;
  (void)WorkerPool::instance();  // start the host workers now, not inside the first cycle
  int rc;
  if ((rc = c->h_pub.reserve(8)) ||
      (rc = c->h_wrow.reserve(((p->max_samples + 31) / 32) * 2 * p->max_points)))
    return fail(rc);
  for (int i = 0; i < 8; ++i) c->h_pub.p[i] = 0;
  if ((rc = c->d_result.reserve(R_SLOTS)) ||
      (rc = c->h_result.reserve(R_SLOTS)) ||
      (rc = ensure_cycle_buffers(c, p->max_samples, p->max_points)) ||
      (rc = c->d_seg.reserve(5 * std::max<size_t>(p->max_segment, 16) + 4 + 8 * 64 + 12 * 8)) ||
      (rc = c->h_seg.reserve(5 * std::max<size_t>(p->max_segment, 16) + 4 + 8 * 64 + 12 * 8)) ||
      (rc = c->h_obs.reserve(2 * std::max<size_t>(p->max_obstacles, 16))) ||
      (rc = c->d_bobs.reserve(2 * std::max<size_t>(p->max_obstacles, 16))) ||
      (rc = c->h_bobs.reserve(2 * std::max<size_t>(p->max_obstacles, 16))))
    return fail(rc);
  // opt in to more than 64 KB of dynamic LDS for the fused kernel (gfx950: 160 KB)
  {
    bool ok = true;
    auto optin = [&](const void *f) {
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
        (void)hipGetLastError();
        ok = false;
      }
    };
    optin(reinterpret_cast<const void *>(rollout_collide_kernel<32, 512>));
    optin(reinterpret_cast<const void *>(rollout_collide_kernel<16, 256>));
    optin(reinterpret_cast<const void *>(rollout_collide_kernel<16, 512>));
    optin(reinterpret_cast<const void *>(rollout_collide_kernel<32, 1024>));
    optin(reinterpret_cast<const void *>(rollout_collide_kernel<64, 1024>));
    optin(reinterpret_cast<const void *>(rollout_collide_kernel<32, 1024, CycleTail>));
    optin(reinterpret_cast<const void *>(rollout_collide_kernel<16, 1024, CycleTail>));
    if (ok) c->lds_limit = 150 * 1024;
    c->lds_limit_hw = c->lds_limit;
    c->cost_lds_ok =
        hipFuncSetAttribute(reinterpret_cast<const void *>(sample_cost_kernel<true, true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(kCostLdsBudget)) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(sample_cost_kernel<true, false, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(kCostLdsBudget)) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(sample_cost_kernel<true, true, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(kCostLdsBudget)) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(sample_cost_kernel<true, false, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(kCostLdsBudget)) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(sample_cost_block_kernel<true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(kBlkLdsBudget)) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(sample_cost_block_kernel<true, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(kBlkLdsBudget)) == hipSuccess;
    if (!c->cost_lds_ok) (void)hipGetLastError();
    c->cost_lds_hw = c->cost_lds_ok;
    c->cost_batch_ok =
        c->cost_lds_ok &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(sample_cost_batched_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kCostLdsBudget)) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(sample_cost_batched_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kCostLdsBudget)) == hipSuccess;
    if (!c->cost_batch_ok) (void)hipGetLastError();
  }
  {
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, p->device) != hipSuccess) {
      (void)hipGetLastError();
      large_bar = 0;
    }
    c->trig_direct = large_bar != 0;
    c->large_bar = large_bar != 0;
    {
      int cus = 0;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device) == hipSuccess && cus > 0)
        c->num_cus = cus;
      else
        (void)hipGetLastError();
    }
    // Process-wide defaults from the environment: diagnostics only (everything that selects a path is a
    // per-context option, kc_dwa_set_option).
    if (const char *e = std::getenv("KC_DEBUG_HOST")) c->hprof.on = e[0] == '1';
    c->sensor_fused_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(sensor_fused_kernel<true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(kSensorFusedLds)) == hipSuccess;
    if (!c->sensor_fused_ok) (void)hipGetLastError();
  }
  if (const char *e = std::getenv("KC_DEBUG_STAMPS")) c->debug_stamps = e[0] == '1';
  hipLaunchKernelGGL(init_result_kernel, dim3(1), dim3(1), 0, c->stream,
                     c->d_result.p);
  if (hipStreamSynchronize(c->stream) != hipSuccess) {
    set_error("result record initialisation failed: %s",
              hipGetErrorString(hipGetLastError()));
    return fail(KC_ERR_HIP);
  }
  *out = c;
  return KC_OK;
}

void kc_dwa_destroy(kc_dwa *c) {
  if (!c) return;
  if (c->hprof.on && c->hprof.n) {
    const char *nm[8] = {"", "entry -> launch call", "launch call", "wait for the trig pool", "flag store", "back in kc_dwa_cycle", "wait for the slots", "fetch (slots + reduce + row)"};
    std::fprintf(stderr, "[kc host] %ld single-launch cycles, us per cycle:\n", c->hprof.n);
    for (int i = 1; i < 8; ++i) std::fprintf(stderr, "  %-28s %6.2f\n", nm[i], c->hprof.sum[i] / c->hprof.n);
    std::fprintf(stderr, "  (of the first: entry -> pool start %.2f, starting the pool %.2f)\n", c->hprof.sum[8] / c->hprof.n,
                 c->hprof.sum[9] / c->hprof.n);
  }
  hipError_t e = hipSetDevice(c->prm.device);
  if (c->debug_stamps && c->d_dbg2.p) {
    std::vector<unsigned long long> h(512 * 32);
    e = hipDeviceSynchronize();
    e = hipMemcpy(h.data(), c->d_dbg2.p, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < 512; ++b) if (h[b * 32]) t0 = std::min(t0, h[b * 32]);
    const char *nm[20] = {"start", "phase A done", "trig entries in LDS", "increments in LDS", "recurrence done", "poses checked", "flags out", "increments done", "costs done", "epilogue done", "pass 0 searched", "pass 0 barrier", "pass 0 total", "poses classified / ticket", "last: reduced", "", "A: loads issued", "A: window stored", "A: sincos / omega stored", "A: cost tables stored"};
    std::fprintf(stderr, "[kc stamps] roll-out kernel, us since first block start (avg / max):\n");
    for (int k = 0; k < 20; ++k) {
      if (k == 15) continue;
      double sm = 0, mx = 0; int nb = 0;
      for (int b = 0; b < 512; ++b) {
        if (!h[b * 32] || !h[b * 32 + k]) continue;
        const double us = (h[b * 32 + k] - t0) / 100.0;
        sm += us; mx = std::max(mx, us); ++nb;
      }
      std::fprintf(stderr, "  %-18s %7.2f / %7.2f  (%d blocks)\n", nm[k], nb ? sm / nb : 0.0, mx, nb);
    }
    {
      double sm = 0, mx = 0; int nb = 0;
      for (int b = 0; b < 512; ++b) {
        if (!h[b * 32]) continue;
        sm += static_cast<double>(h[b * 32 + 15]); mx = std::max(mx, static_cast<double>(h[b * 32 + 15])); ++nb;
      }
      std::fprintf(stderr, "  undecided poses per workgroup (exact shell tests): %.1f / %.0f\n", nb ? sm / nb : 0.0, mx);
    }
  }
  if (c->debug_stamps && c->d_dbg.p) {  // diagnostic dump of the last cycle
    std::vector<unsigned long long> h(512 * 16);
    e = hipDeviceSynchronize();
    e = hipMemcpy(h.data(), c->d_dbg.p, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < 512; ++b) if (h[b * 16]) t0 = std::min(t0, h[b * 16]);
    double mx[16] = {0}, sm[16] = {0};
    int nb = 0;
    for (int b = 0; b < 512; ++b) {
      if (!h[b * 16] || !h[b * 16 + 4]) continue;
      ++nb;
      for (int k = 0; k < 16; ++k) {
        if (!h[b * 16 + k]) continue;
        const double us = (h[b * 16 + k] - t0) / 100.0;
        mx[k] = std::max(mx[k], us);
        sm[k] += us;
      }
    }
    std::fprintf(stderr, "[kc stamps] %d working blocks; us since first block start (avg / max):\n", nb);
    const char *nm[16] = {"start", "count loaded", "all samples done", "first sample done", "key written", "-", "lds filled", "s0 points loaded", "s0 seg pass 1", "s0 seg pass 2", "s0 seg pass 3", "s0 seg done", "s0 obstacles done", "", "", ""};
    {
      double mhz = 0; int cnt = 0;
      for (int b = 0; b < 512; ++b) {
        if (!h[b * 16] || !h[b * 16 + 2] || !h[b * 16 + 14]) continue;
        mhz += double(h[b * 16 + 14] - h[b * 16 + 13]) / (double(h[b * 16 + 2] - h[b * 16]) / 100.0);
        ++cnt;
      }
      std::fprintf(stderr, "  s_memtime ticks per us (start -> points done): %.1f\n", cnt ? mhz / cnt : 0.0);
    }
    const int order[13] = {0, 1, 6, 7, 8, 9, 10, 11, 12, 3, 2, 4, 5};
    for (int q = 0; q < 13; ++q) {
      const int k = order[q];
      std::fprintf(stderr, "  %-14s %7.2f / %7.2f\n", nm[k], nb ? sm[k] / nb : 0.0, mx[k]);
    }
  }
  if (c->own_stream) {
    e = hipStreamSynchronize(c->own_stream);
    e = hipStreamDestroy(c->own_stream);
  }
  (void)e;
  c->timing.release();
  c->d_vxt.release();
  c->d_vyt.release();
  c->d_vidx.release();
  c->d_pvi.release();
  c->d_cpvi.release();
  c->d_row.release();
  c->h_trig.release();
  c->d_trig.release();
  c->h_bits.release();
  c->d_bits.release();
  c->h_ddz.release();
  c->d_ddz.release();
  c->d_px.release();
  c->d_py.release();
  c->d_costs.release();
  c->d_flags.release();
  c->d_dbg.release();
  c->d_dbg2.release();
  c->d_raw.release();
  c->d_sensor_tmp.release();
  c->d_sensor_bytes.release();
  c->d_vsum.release();
  c->d_gridcnt.release();
  c->h_gridrec.release();
  if (c->aux_stream) {
    hipError_t ae = hipStreamSynchronize(c->aux_stream);
    ae = hipStreamDestroy(c->aux_stream);
    ae = hipEventDestroy(c->aux_fork);
    ae = hipEventDestroy(c->aux_join);
    (void)ae;
  }
  if (c->grid_ready) {
    hipError_t ge = hipEventDestroy(c->grid_ready);
    (void)ge;
  }
  c->d_perm.release();
  c->d_prow.release();
  c->d_vvx.release();
  c->d_vvy.release();
  c->d_vom.release();
  c->h_seg.release();
  c->d_seg.release();
  c->d_near.release();
  c->d_bbox.release();
  c->d_path.release();
  c->h_obs.release();
  c->h_cells.release();
  c->d_cells.release();
  c->h_bobs.release();
  c->d_bobs.release();
  c->h_skip.release();
  c->d_skip.release();
  c->h_gbits.release();
  c->h_gz.release();
  c->d_gz.release();
  c->h_zlut.release();
  c->d_zlut.release();
  c->d_block_keys.release();
  c->d_gbits.release();
  c->d_ginner.release();
  c->d_gouter.release();
  c->d_adm.release();
  c->d_pos.release();
  c->d_result.release();
  c->h_result.release();
  c->h_pub.release();
  c->h_row.release();
  c->d_adm_bits.release();
  c->d_cperm.release();
  c->d_cprow.release();
  c->h_wrow.release();
  c->h_slots.release();
  c->d_oscan.release();
  c->d_onear.release();
  c->d_freeze.release();
  c->d_first_hit.release();
  c->d_frz.release();
  c->d_omega.release();
  c->d_sincostab.release();
  c->d_gid.release();
  c->d_xs.release();
  c->d_xr.release();
  c->h_xvec.release();
  c->h_xrec.release();
  delete c;
}

int kc_dwa_set_stream(kc_dwa *c, void *hip_stream) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  KC_TRY(use_device(c));
  KC_HIP(hipStreamSynchronize(c->stream));
  c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  return KC_OK;
}

int kc_dwa_set_resolution(kc_dwa *c, double res) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  if (!(res > 0.0)) KC_FAIL(KC_ERR_RANGE, "octree resolution must be > 0");
  c->res = res;
  return KC_OK;
}

// per-context switches (header: kc_dwa_set_option)
int kc_dwa_set_option(kc_dwa *c, const char *name, double v) {
  if (!c || !name) KC_FAIL(KC_ERR_INVALID, "null argument");
  KC_TRY(use_device(c));
  // whatever is queued was built under the old settings
  KC_HIP(hipStreamSynchronize(c->stream));
  c->drained = true;
  c->update_busy = false;
  const std::string n(name);
  const bool on = v != 0.0;
  if (n == "fused_cycle") {
    if (!(v == 0.0 || v == 1.0 || v == 2.0)) KC_FAIL(KC_ERR_RANGE, "fused_cycle: 0 off, 1 when it pays, 2 whenever it fits");
    c->cycle_fused = on;
    c->cycle_forced = v == 2.0;
  }
  else if (n == "write_paths") c->write_paths = on;
  else if (n == "host_reduce") c->host_reduce = on;
  else if (n == "cost_kernel") {
    if (!(v == 0.0 || v == 1.0 || v == 2.0)) KC_FAIL(KC_ERR_RANGE, "cost_kernel: 0 auto, 1 workgroup per sample, 2 wavefront per sample");
    c->cost_kernel_force = static_cast<int>(v);
  } else if (n == "cycle_samples") {
    if (!(v == 0.0 || v == 16.0 || v == 32.0)) KC_FAIL(KC_ERR_RANGE, "cycle_samples: 0 (by shard size), 16 or 32");
    c->cycle_samples_opt = static_cast<int>(v);
  } else if (n == "velocity_beside") c->velocity_beside = on;
  else if (n == "velocity_group") {
    if (!(v == 0.0 || v == 1.0 || v == 4.0 || v == 16.0)) KC_FAIL(KC_ERR_RANGE, "velocity_group: 0 (by batch size), 1, 4 or 16");
    c->velocity_group = static_cast<int>(v);
  } else if (n == "near_table") {
    if (v != 0.0 && !(v >= 16.0 && v <= 512.0)) KC_FAIL(KC_ERR_RANGE, "near_table: 0 (off) or 16..512 cells per side");
    c->near_side = static_cast<int>(v);
    c->near_version = ~0ull;
    c->near_ok = false;
  } else if (n == "drop_samples") {
    c->drop_samples = on;
    c->freeze_valid = false;
    if (!on) KC_TRY(upload_omega(c));
  } else if (n == "num_ctrl_points") {
    if (!(v >= 0.0 && v <= 1e9)) KC_FAIL(KC_ERR_RANGE, "num_ctrl_points: a count >= 0");
    c->num_ctrl_points = static_cast<size_t>(v);
  } else if (n == "cost_batch") {
    c->cost_batch = on;
    c->cost_batch_forced = v == 2.0;
  } else if (n == "obs_union") {
    if (!(v >= 0.0 && v <= 4096.0)) KC_FAIL(KC_ERR_RANGE, "obs_union %g outside [0, 4096]", v);
    c->obs_union = static_cast<int>(v);
  } else if (n == "obs_near") {
    if (v != 0.0 && v != 1.0 && !(v >= 16.0 && v <= 512.0)) KC_FAIL(KC_ERR_RANGE, "obs_near: 0 (off), 1 (on) or 16..512 cells per side");
    c->obs_near_opt = on;
    if (v >= 16.0) c->onear_side = static_cast<int>(v);
    c->onear_version = ~0ull;
    c->onear_ok = false;
    if (!on) c->oscan_valid = false;
  } else if (n == "sensor_two_launch") c->sensor_two_launch = on;
  else if (n == "device_trig") c->device_trig = on;
  else if (n == "sensor_on_host") c->device_sensor = !on;
  else if (n == "force_split") {
    c->lds_limit = on ? 0 : c->lds_limit_hw;
    c->cost_lds_ok = on ? false : c->cost_lds_hw;
  } else
    KC_FAIL(KC_ERR_INVALID, "unknown option '%s'", name);
  return KC_OK;
}

int kc_dwa_get_option(kc_dwa *c, const char *name, double *v) {
  if (!c || !name || !v) KC_FAIL(KC_ERR_INVALID, "null argument");
  const std::string n(name);
  if (n == "fused_cycle") *v = c->cycle_fused ? (c->cycle_forced ? 2.0 : 1.0) : 0.0;
  else if (n == "write_paths") *v = c->write_paths;
  else if (n == "host_reduce") *v = c->host_reduce;
  else if (n == "cost_kernel") *v = c->cost_kernel_force;
  else if (n == "sensor_two_launch") *v = c->sensor_two_launch;
  else if (n == "near_table") *v = c->near_side;
  else if (n == "cycle_samples") *v = c->cycle_samples_opt;
  else if (n == "velocity_group") *v = c->velocity_group;
  else if (n == "velocity_beside") *v = c->velocity_beside;
  else if (n == "last_cycle_samples") *v = c->cycle_samples;  // read-only
  else if (n == "obs_near") *v = c->obs_near_opt ? c->onear_side : 0;
  else if (n == "cost_batch") *v = c->cost_batch ? (c->cost_batch_forced ? 2.0 : 1.0) : 0.0;
  else if (n == "obs_union") *v = c->obs_union;
  else if (n == "obs_near_rides") *v = static_cast<double>(c->onear_rides);    // read-only
  else if (n == "obs_near_builds") *v = static_cast<double>(c->onear_builds);  // read-only
  else if (n == "device_trig") *v = c->device_trig && trig_selfcheck_ok();
  else if (n == "trig_rides") *v = static_cast<double>(c->trig_rides);  // read-only
  else if (n == "sensor_on_host") *v = !c->device_sensor;
  else if (n == "force_split") *v = c->lds_limit == 0;
  else if (n == "last_cycle_single_launch") *v = c->cycle_launched;  // read-only
  else if (n == "host_threads") *v = WorkerPool::instance().workers() + 1;  // read-only here: kc_set_host_threads
  else if (n == "drop_samples") *v = c->drop_samples;
  else if (n == "num_ctrl_points") *v = static_cast<double>(c->num_ctrl_points);
  else if (n == "trig_rows") *v = static_cast<double>(c->lat.omega_values.size());  // read-only: rows of the host's cos / sin table
  else if (n == "shard_samples") *v = static_cast<double>(c->shard_count);          // read-only: samples this context rolls out
  else
    KC_FAIL(KC_ERR_INVALID, "unknown option '%s'", name);
  return KC_OK;
}

int kc_set_host_threads(int n) {
  if (n < 1 || n > 64) KC_FAIL(KC_ERR_RANGE, "host threads must be 1..64");
  WorkerPool::instance().resize(n);
  return KC_OK;
}

int kc_trig_selfcheck(int64_t *compared_out) {
  long n = 0;
  const int bad = trig_selfcheck_run(&n);
  if (compared_out) *compared_out = n;
  if (bad) KC_FAIL(KC_ERR_STATE, "%d of %ld arguments: the restated sincos differs from the installed libm's", bad, n);
  return KC_OK;
}

int kc_trig_table(double yaw0, const double *omega, size_t n_rows, size_t n_steps, double dt, double *cos_sin_out) {
  if (!omega || !cos_sin_out) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (n_rows == 0 || n_steps == 0) return KC_OK;
  if (n_steps > 4096 || n_rows > (1u << 20)) KC_FAIL(KC_ERR_RANGE, "table of %zu x %zu entries", n_rows, n_steps);
  double om_max = 0.0;
  for (size_t i = 0; i < n_rows; ++i) om_max = std::max(om_max, std::fabs(omega[i]));
  const double reach = std::fabs(yaw0) + om_max * std::fabs(dt) * static_cast<double>(n_steps);
  if (!(reach < 1.0e8)) KC_FAIL(KC_ERR_RANGE, "yaw reaches %g: outside the table + Cody-Waite range of sincos", reach);
  DevBuf<double> d_om, d_tab;
  DevBuf<double2> d_out;
  KC_TRY(d_om.reserve(n_rows));
  KC_TRY(d_tab.reserve(440));
  KC_TRY(d_out.reserve(n_rows * n_steps));
  KC_HIP(hipMemcpy(d_om.p, omega, n_rows * sizeof(double), hipMemcpyHostToDevice));
  KC_HIP(hipMemcpy(d_tab.p, kc_sincostab_host, sizeof(kc_sincostab_host), hipMemcpyHostToDevice));
  TrigJob tj{};
  tj.yaw0 = yaw0;
  tj.dt = dt;
  tj.omega = d_om.p;
  tj.tab = d_tab.p;
  tj.out = d_out.p;
  tj.A = static_cast<int>(n_rows);
  tj.P = static_cast<int>(n_steps);
  tj.nblk = static_cast<int>(std::min<size_t>(1024, blocks_for(n_rows * n_steps, kTrigBlock)));
  hipLaunchKernelGGL(trig_table_kernel, dim3(tj.nblk), dim3(kTrigBlock), 0, nullptr, tj);
  KC_HIP(hipGetLastError());
  KC_HIP(hipMemcpy(cos_sin_out, d_out.p, n_rows * n_steps * sizeof(double2), hipMemcpyDeviceToHost));
  return KC_OK;
}

int kc_dwa_set_weights(kc_dwa *c, const kc_weights *w) {
  if (!c || !w) KC_FAIL(KC_ERR_INVALID, "null argument");
  c->w = *w;
  return KC_OK;
}

int kc_dwa_sample_window(kc_dwa *c, int ctr_type, const kc_limits *limits,
                         double cvx, double cvy, double com, int max_lin,
                         int max_ang, size_t *n_out, double *vx, double *vy,
                         double *omega, size_t cap) {
  if (!c || !limits) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (ctr_type < KC_ACKERMANN || ctr_type > KC_OMNI)
    KC_FAIL(KC_ERR_INVALID, "Invalid control type");
  if (max_lin < 1 || max_ang < 1)
    KC_FAIL(KC_ERR_RANGE, "sample counts must be >= 1");
  KC_TRY(use_device(c));
  if (c->rows_active) c->lat = std::move(c->full);  // the previous FULL window: its index pattern may carry over
  hm::build_window_lattice(ctr_type, *limits, cvx, cvy, com, c->prm.time_step,
                           max_lin, max_ang, c->lat);
  KC_TRY(apply_shard_rule(c));
  const hm::VelocityLattice &fl = full_list(c);
  const size_t n = fl.size();
  if (n_out) *n_out = n;
  if (vx || vy || omega) {
    if (cap < n) KC_FAIL(KC_ERR_RANGE, "output capacity %zu < %zu", cap, n);
    for (size_t i = 0; i < n; ++i) {
      if (vx) vx[i] = fl.vx(i);
      if (vy) vy[i] = fl.vy(i);
      if (omega) omega[i] = fl.omega(i);
    }
  }
  return KC_OK;
}

int kc_dwa_set_samples(kc_dwa *c, size_t n, const double *vx, const double *vy,
                       const double *omega) {
  if (!c || (n && (!vx || !vy || !omega)))
    KC_FAIL(KC_ERR_INVALID, "null argument");
  KC_TRY(use_device(c));
  if (n > 65536) KC_FAIL(KC_ERR_RANGE, "more than 65536 samples per list");
  c->lat.clear();
  // an explicit list: the distinct values of each axis become its tables (bit patterns: -0.0 and +0.0 of a
  // linear velocity stay two entries -- the product vx * cos keeps the sign; omegas share a row, yaw += 0)
  std::unordered_map<uint64_t, int32_t> rows, xs, ys;
  rows.reserve(1024);
  xs.reserve(1024);
  ys.reserve(1024);
  auto slot = [](std::unordered_map<uint64_t, int32_t> &m, std::vector<double> &values, double key, double value) {
    uint64_t bits;
    std::memcpy(&bits, &key, 8);
    auto it = m.find(bits);
    if (it != m.end()) return it->second;
    const int32_t r = static_cast<int32_t>(values.size());
    values.push_back(value);
    m.emplace(bits, r);
    return r;
  };
  for (size_t i = 0; i < n; ++i) {
    const int32_t r = slot(rows, c->lat.omega_values, omega[i] + 0.0, omega[i]);  // -0.0 and +0.0 share a row
    const int32_t a = slot(xs, c->lat.vx_values, vx[i], vx[i]);
    const int32_t b = slot(ys, c->lat.vy_values, vy[i], vy[i]);
    c->lat.push(static_cast<uint16_t>(a), static_cast<uint16_t>(b), r);
  }
  return apply_shard_rule(c);
}

int kc_dwa_set_shard(kc_dwa *c, size_t first, size_t count) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  KC_TRY(use_device(c));
  if (c->rows_active) {  // the full list again
    c->lat = std::move(c->full);
    c->layout.mode = -1;
    KC_TRY(apply_shard_rule(c));
  }
  c->layout.mode = -1;
  if (first + count > c->lat.size())
    KC_FAIL(KC_ERR_RANGE, "shard [%zu, %zu) outside the %zu samples", first,
            first + count, c->lat.size());
  c->shard_first = first;
  c->shard_count = count;
  return KC_OK;
}

int kc_dwa_set_shard_rule(kc_dwa *c, int rank, int world, int mode) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  if (mode >= 0 && mode != KC_SHARD_BLOCKS && mode != KC_SHARD_ROWS) KC_FAIL(KC_ERR_INVALID, "unknown shard mode %d", mode);
  if (mode >= 0 && (world < 1 || rank < 0 || rank >= world)) KC_FAIL(KC_ERR_RANGE, "rank %d outside world %d", rank, world);
  KC_TRY(use_device(c));
  if (c->rows_active) c->lat = std::move(c->full);  // the full list back in front of the rule
  c->layout = ShardLayout{};
  c->layout.mode = mode < 0 ? -1 : mode;
  c->layout.rank = mode < 0 ? 0 : rank;
  c->layout.world = mode < 0 ? 1 : world;
  return apply_shard_rule(c);
}

int kc_shard_plan(const int32_t *rows, size_t n, int world, int mode, int32_t *owner_out) {
  if ((n && (!rows || !owner_out))) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (world < 1) KC_FAIL(KC_ERR_RANGE, "world %d", world);
  if (mode == KC_SHARD_ROWS) {
    for (size_t i = 0; i < n; ++i)
      if (rows[i] < 0) KC_FAIL(KC_ERR_RANGE, "negative row label at %zu", i);
    shard_rows_owner(rows, n, world, owner_out);
  } else if (mode == KC_SHARD_BLOCKS) {
    ShardLayout L;
    shard_blocks(n, world, L);
    for (int r = 0; r < world; ++r)
      for (size_t i = 0; i < L.count[static_cast<size_t>(r)]; ++i) owner_out[L.first[static_cast<size_t>(r)] + i] = r;
  } else {
    KC_FAIL(KC_ERR_INVALID, "unknown shard mode %d", mode);
  }
  return KC_OK;
}

int kc_shard_merge(const int64_t *record, size_t words_per_rank, int world, int mode, const int32_t *owner,
                   size_t n_total, kc_result *out) {
  if (!record || !out) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (world < 1) KC_FAIL(KC_ERR_RANGE, "world %d", world);
  ShardLayout L;
  L.mode = mode;
  L.world = world;
  L.n_total = n_total;
  if (mode == KC_SHARD_BLOCKS) {
    shard_blocks(n_total, world, L);
  } else if (mode == KC_SHARD_ROWS) {
    if (n_total && !owner) KC_FAIL(KC_ERR_INVALID, "KC_SHARD_ROWS needs the owner table");
    L.count.assign(static_cast<size_t>(world), 0);
    L.first.assign(static_cast<size_t>(world), 0);
    L.gids.assign(static_cast<size_t>(world), {});
    for (size_t g = 0; g < n_total; ++g) {
      if (owner[g] < 0 || owner[g] >= world) KC_FAIL(KC_ERR_RANGE, "owner[%zu] = %d outside world %d", g, owner[g], world);
      L.gids[static_cast<size_t>(owner[g])].push_back(static_cast<int32_t>(g));
    }
    for (int r = 0; r < world; ++r) L.count[static_cast<size_t>(r)] = L.gids[static_cast<size_t>(r)].size();
  } else {
    KC_FAIL(KC_ERR_INVALID, "unknown shard mode %d", mode);
  }
  if (64 * words_per_rank < L.max_count())
    KC_FAIL(KC_ERR_RANGE, "%zu words per rank cannot hold a share of %zu samples", words_per_rank, L.max_count());
  bool failed = false;
  static_assert(sizeof(long long) == sizeof(int64_t), "record words");
  merge_exchange(L, reinterpret_cast<const long long *>(record), words_per_rank, out, &failed);
  if (failed) KC_FAIL(KC_ERR_HIP, "the exchange record carries a rank's error word (%lld)", static_cast<long long>(record[X_ERR]));
  return KC_OK;
}

int kc_dwa_owns_sample(kc_dwa *c, int64_t raw, int *owned) {
  if (!c || !owned) KC_FAIL(KC_ERR_INVALID, "null argument");
  int64_t lat_id = raw;
  if (c->rows_active) {
    const auto it = std::lower_bound(c->gid.begin(), c->gid.end(), static_cast<int32_t>(std::min<int64_t>(std::max<int64_t>(raw, -1), INT32_MAX)));
    lat_id = (raw >= 0 && it != c->gid.end() && *it == raw) ? static_cast<int64_t>(it - c->gid.begin()) : -1;
  }
  *owned = (lat_id >= static_cast<int64_t>(c->shard_first) && lat_id < static_cast<int64_t>(c->shard_first + c->shard_count)) ? 1 : 0;
  return KC_OK;
}

int kc_dwa_set_scan(kc_dwa *c, const kc_state *st, const double *ranges,
                    const double *angles, size_t n, float max_range) {
  if (!c || !st || (n && (!ranges || !angles)))
    KC_FAIL(KC_ERR_INVALID, "null argument");
  KC_TRY(use_device(c));
  KC_TRY(quiesce_for_update(c));  // staging buffers and device tables are reused
  c->host_lists_valid = true;
  // CollisionChecker::updateState + updateSensorData<LaserScan>
  const hm::Rigid3f body = hm::Rigid3f::from_pose2d(st->x, st->y, st->yaw);
  c->frame = body * c->sensor_tf_body;
  // a mount that is not a rotation about z tilts the octree against the upright robot shape: exact
  // 3-D tests on the split roll-out path (kc_tilt_dev.h), host-built voxel columns, no dilated masks
  c->tilted = !c->frame.planar();
  c->tilt_body_x = st->x;
  c->tilt_body_y = st->y;
  const float hz = static_cast<float>(
      -static_cast<double>(c->sensor_tf_body.t[2]) / 2.0);
  // CostEvaluator::setPointScan(LaserScan): sensor_tf_body * body_tf_world
  c->obs_tf = c->sensor_tf_body * body;
  c->raw_is_scan = true;
  // cos/sin of the beam angles (host libm, like the reference), kept while the
  // angle table stays the same
  if (c->scan_angles.size() != n ||
      (n && std::memcmp(c->scan_angles.data(), angles, n * sizeof(double)) != 0)) {
    c->scan_angles.assign(angles, angles + n);
    c->scan_cs.resize(n);
    for (size_t i = 0; i < n; ++i) c->scan_cs[i] = make_double2(std::cos(angles[i]), std::sin(angles[i]));
  }
  // sensor-frame points: voxels at z = hz (collision_check.h:110-115; a
  // non-finite range gives non-finite coordinates, which add_voxel drops),
  // obstacles from the same x, y at z = 0 (cost path: no filter)
  c->scan_xyz.resize(3 * n);
  for (size_t i = 0; i < n; ++i) {
    const double r = ranges[i];
    c->scan_xyz[3 * i] = static_cast<float>(r * c->scan_cs[i].x);
    c->scan_xyz[3 * i + 1] = static_cast<float>(r * c->scan_cs[i].y);
    c->scan_xyz[3 * i + 2] = hz;
  }
  c->have_sensor = true;
  c->max_obs_dist = max_range / 3.0f;  // cost_evaluator.h:179
  ++c->sensor_version;
  c->oscan_valid = false;
  c->onear_ok = false;
  if (c->obs_near_opt && n >= 64 && n <= 65536) {
    // the obstacles in beam order (CostEvaluator::setPointScan, cost_evaluator.h:174-193: sensor_tf_body *
    // body_tf_world applied to (r cos a, r sin a, 0)) and the boxes of their chunks, for the near table of the
    // scan; a non-finite range leaves the scan to the bucket search
    bool finite = true;
    for (size_t i = 0; i < n && finite; ++i) finite = std::isfinite(ranges[i]);
    if (finite) {
      const int cs = static_cast<int>((n + 63) / 64);
      const int nch = static_cast<int>((n + cs - 1) / cs);
      c->h_oscan.resize(2 * n + 256);
      float *hx = c->h_oscan.data(), *hy = hx + n, *box = hy + n;
      for (size_t i = 0; i < n; ++i) {
        float o[3];
        c->obs_tf.apply(c->scan_xyz[3 * i], c->scan_xyz[3 * i + 1], 0.0f, o);
        hx[i] = o[0];
        hy[i] = o[1];
      }
      const float inf = std::numeric_limits<float>::infinity();
      for (int k = 0; k < 64; ++k) {
        float x0 = inf, x1 = -inf, y0 = inf, y1 = -inf;
        if (k < nch)
          for (size_t j = static_cast<size_t>(k) * cs; j < std::min(n, static_cast<size_t>(k + 1) * cs); ++j) {
            x0 = std::min(x0, hx[j]);
            x1 = std::max(x1, hx[j]);
            y0 = std::min(y0, hy[j]);
            y1 = std::max(y1, hy[j]);
          }
        box[k] = x0;
        box[64 + k] = x1;
        box[128 + k] = y0;
        box[192 + k] = y1;
      }
      KC_TRY(c->d_oscan.reserve(2 * n + 256));
      KC_TRY(upload_table(c, c->d_oscan.p, hx, (2 * n + 256) * sizeof(float)));
      if (!c->trig_direct) {
        KC_HIP(hipStreamSynchronize(c->stream));  // (pageable source)
      } else {
        bar_flush(c);
      }
      c->oscan_valid = true;
      c->oscan_n = n;
      c->oscan_cs = cs;
      c->oscan_nch = nch;
    }
  }
  bool done = false;
  c->onear_ahead = false;
  if (!c->tilted) {
    if (c->obs_near_ahead) KC_TRY(onear_plan_ahead(c, st->x, st->y));
    c->trig_plan = true;  // (the launch of this update may carry the trig table of the cycle that follows)
    c->trig_plan_yaw = st->yaw;
    const int rc = sensor_update_device(c, c->scan_xyz.data(), n, &done);
    c->trig_plan = false;
    c->onear_ahead = false;
    KC_TRY(rc);
  }
  if (done) return KC_OK;
  build_host_lists(c, c->scan_xyz.data(), n);
  KC_TRY(upload_voxels(c));
  if (c->tilted) {
    c->have_dil = false;
    if (!c->vox_kx.empty() && !c->have_gbits)  // (cannot happen: a wider span was cropped to the reachable window above)
      KC_FAIL(KC_ERR_UNSUPPORTED, "tilted sensor frame: the scan's voxel columns span more than 8192 cells");
  }
  return upload_obstacles(c, n);
}

}  // extern "C"
namespace {
// updateSensorData<std::vector<Path::Point>>(cloud, global_frame), collision_check.h:119-131: the octree of a
// world-frame list lies in the world frame (identity); that of a SENSOR-frame list in body->tf * sensor_tf_body,
// like a laser scan's (the voxel keys are taken from the points as they are; the poses go into that frame).
int set_points_impl(kc_dwa *c, const kc_state *st, const float *xyz, size_t n, float max_range, bool global_frame) {
  if (!c || !st || (n && !xyz)) KC_FAIL(KC_ERR_INVALID, "null argument");
  KC_TRY(use_device(c));
  const auto dbg_t0 = std::chrono::steady_clock::now();
  KC_TRY(quiesce_for_update(c));
  const auto dbg_t1 = std::chrono::steady_clock::now();
  const hm::Rigid3f body = hm::Rigid3f::from_pose2d(st->x, st->y, st->yaw);
  c->frame = global_frame ? hm::Rigid3f::identity() : body * c->sensor_tf_body;
  c->tilted = false;
  if (!c->frame.planar())
    KC_FAIL(KC_ERR_UNSUPPORTED, "a sensor-frame point list under a sensor mount that is not a rotation about z (several "
                                "voxel layers in a tilted octree frame) is not restated; laser scans are");
  ++c->sensor_version;
  c->oscan_valid = false;
  c->onear_ok = false;
  c->obs_tf = c->sensor_tf_body * body;  // setPointScan(cloud): the same whatever frame the octree takes
  c->raw_is_scan = false;
  c->have_sensor = true;
  c->max_obs_dist = max_range / 3.0f;
  c->host_lists_valid = true;
  bool done = false;
  c->trig_plan = true;  // (the launch of this update may carry the trig table of the cycle that follows)
  c->trig_plan_yaw = st->yaw;
  const int rc_dev = sensor_update_device(c, xyz, n, &done);
  c->trig_plan = false;
  KC_TRY(rc_dev);
  if (done) {
    if (c->debug_stamps)
      std::fprintf(stderr, "[kc] set_points (device build): sync %.1f | host part %.1f us\n",
                   std::chrono::duration<double, std::micro>(dbg_t1 - dbg_t0).count(),
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_t1).count());
    return KC_OK;
  }
  build_host_lists(c, xyz, n);
  const auto dbg_t2 = std::chrono::steady_clock::now();
  KC_TRY(upload_voxels(c));
  const auto dbg_t3 = std::chrono::steady_clock::now();
  const int rc = upload_obstacles(c, n);
  const auto dbg_t4 = std::chrono::steady_clock::now();
  if (c->debug_stamps) {
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    std::fprintf(stderr, "[kc] set_points: sync %.1f | voxelise+transform %.1f | upload_voxels %.1f | upload_obstacles %.1f us\n",
                 us(dbg_t0, dbg_t1), us(dbg_t1, dbg_t2), us(dbg_t2, dbg_t3), us(dbg_t3, dbg_t4));
  }
  return rc;
}
}  // namespace
extern "C" {

int kc_dwa_set_points(kc_dwa *c, const kc_state *st, const float *xyz, size_t n, float max_range) {
  return set_points_impl(c, st, xyz, n, max_range, true);
}

int kc_dwa_set_points_sensor_frame(kc_dwa *c, const kc_state *st, const float *xyz, size_t n, float max_range) {
  return set_points_impl(c, st, xyz, n, max_range, false);
}

// the mapper's grid feeds the controller without leaving the
// device.  Same state as kc_dwa_set_points with the list of the OCCUPIED cells.
int kc_dwa_set_grid_device(kc_dwa *c, const kc_state *st, const int32_t *dev_grid, int H, int W,
                           float res, int c0, int c1, float max_range) {
  if (!c || !st || !dev_grid) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (H <= 0 || W <= 0 || !(res > 0.0f) || static_cast<size_t>(H) * W > 0x3FFFFFFFul)
    KC_FAIL(KC_ERR_INVALID, "grid dimensions and resolution must be positive");
  KC_TRY(use_device(c));
  KC_TRY(quiesce_for_update(c));
  c->frame = hm::Rigid3f::identity();
  c->tilted = false;
  ++c->sensor_version;
  c->oscan_valid = false;
  c->onear_ok = false;
  const hm::Rigid3f body = hm::Rigid3f::from_pose2d(st->x, st->y, st->yaw);
  c->obs_tf = c->sensor_tf_body * body;
  c->raw_is_scan = false;
  c->raw_on_device = false;
  c->have_sensor = true;
  c->max_obs_dist = max_range / 3.0f;
  c->host_lists_valid = true;
  const size_t cells = static_cast<size_t>(H) * W;
  KC_TRY(c->d_raw.reserve(3 * cells + 16));
  KC_TRY(c->h_gridrec.reserve(8));
  if (!c->d_gridcnt.p) {
    KC_TRY(c->d_gridcnt.reserve(5 * kGridCntStride));
    int init[5 * kGridCntStride] = {0};
    init[1 * kGridCntStride] = INT_MAX;
    init[2 * kGridCntStride] = INT_MIN;
    init[3 * kGridCntStride] = INT_MAX;
    init[4 * kGridCntStride] = INT_MIN;
    KC_HIP(hipMemcpyAsync(c->d_gridcnt.p, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    KC_HIP(hipStreamSynchronize(c->stream));
    c->h_gridrec.p[0] = 0;
  }
  GridPtsArgs ga{};
  ga.grid = dev_grid;
  ga.H = H;
  ga.W = W;
  ga.c0 = c0;
  ga.c1 = c1;
  ga.res = res;
  ga.xyz = c->d_raw.p;
  ga.cnt = c->d_gridcnt.p;
  const long long seq = ++c->grid_seq;
  KC_TRY(c->timing.start("grid_points_kernel", c->stream));
  hipLaunchKernelGGL(grid_points_kernel, dim3(blocks_for(cells, 256)), dim3(256), 0, c->stream, ga);
  KC_TRY(c->timing.stop(c->stream));
  hipLaunchKernelGGL(grid_points_publish_kernel, dim3(1), dim3(1), 0, c->stream, c->d_gridcnt.p,
                     c->h_gridrec.p, seq);
  KC_HIP(hipGetLastError());
  c->update_busy = true;
  {
    volatile long long *p = c->h_gridrec.p;
    const auto t0 = std::chrono::steady_clock::now();
    for (long spins = 0; *p != seq; ++spins) {
      if ((spins & 255) == 255 &&
          std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
        KC_HIP(hipStreamSynchronize(c->stream));
        break;
      }
    }
    if (*p != seq) KC_FAIL(KC_ERR_HIP, "the grid hand-off kernels did not report");
  }
  const size_t n = static_cast<size_t>(c->h_gridrec.p[1]);
  if (n == 0) {
    build_host_lists(c, nullptr, 0);
    KC_TRY(upload_voxels(c));
    return upload_obstacles(c, 0);
  }
  const float lo[3] = {static_cast<float>(static_cast<int>(c->h_gridrec.p[2]) - c0) * res,
                       static_cast<float>(static_cast<int>(c->h_gridrec.p[4]) - c1) * res, 0.0f};
  const float hi[3] = {static_cast<float>(static_cast<int>(c->h_gridrec.p[3]) - c0) * res,
                       static_cast<float>(static_cast<int>(c->h_gridrec.p[5]) - c1) * res, 0.0f};
  bool done = false;
  KC_TRY(sensor_update_device_bounded(c, nullptr, n, lo, hi, &done));
  if (done) return KC_OK;
  // large maps / spheres: the host path, on the (small) list instead of the grid
  c->raw_xyz.resize(3 * n);
  KC_HIP(hipMemcpyAsync(c->raw_xyz.data(), c->d_raw.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost,
                        c->stream));
  KC_HIP(hipStreamSynchronize(c->stream));
  c->update_busy = false;
  c->raw_on_device = false;
  build_host_lists(c, c->raw_xyz.data(), n);
  KC_TRY(upload_voxels(c));
  return upload_obstacles(c, n);
}

int kc_dwa_set_grid_from_mapper(kc_dwa *c, const kc_state *st, kc_mapper *m, float max_range) {
  if (!c || !st || !m) KC_FAIL(KC_ERR_INVALID, "null argument");
  kc::MapperView v{};
  KC_TRY(kc::mapper_view(m, &v));
  if (v.device != c->prm.device)
    KC_FAIL(KC_ERR_INVALID, "mapper on device %d, controller on device %d", v.device, c->prm.device);
  KC_TRY(use_device(c));
  if (v.stream != c->stream) {
    // the controller's stream waits for the scan; the host does not
    if (!c->grid_ready) KC_HIP(hipEventCreateWithFlags(&c->grid_ready, hipEventDisableTiming));
    KC_HIP(hipEventRecord(c->grid_ready, v.stream));
    KC_HIP(hipStreamWaitEvent(c->stream, c->grid_ready, 0));
  }
  return kc_dwa_set_grid_device(c, st, v.grid, v.H, v.W, v.res, v.c0, v.c1, max_range);
}

}  // extern "C"
namespace {
// x / y / z rows, or xyz = [S][3] interleaved points (Path::Point order) de-interleaved on the way into the rows
int set_tracked_segment_impl(kc_dwa *c, const float *x, const float *y, const float *z, const float *xyz,
                             const float *acc, size_t S, float ref_len) {
  static double dbg_sum[6] = {0};
  static long dbg_n = 0;
  const auto dbg0 = std::chrono::steady_clock::now();
  auto dbg_mark = [&](int i, std::chrono::steady_clock::time_point &last) {
    if (!c->hprof.on) return;
    const auto now = std::chrono::steady_clock::now();
    dbg_sum[i] += std::chrono::duration<double, std::micro>(now - last).count();
    last = now;
  };
  auto dbg_t = dbg0;
  KC_TRY(use_device(c));
  KC_TRY(quiesce_for_update(c, /*sensor_tables=*/false));
  dbg_mark(0, dbg_t);
  c->S = S;
  c->ref_len = ref_len;
  if (S == 0) return KC_OK;
  // rows [5][S], then capsules of the chunks [8][nch] and bounding spheres of
  // the super-chunks (8 chunks) [4][nsup] (sample_cost_kernel, steps 2 and 4)
  const size_t chunk = (std::max<size_t>(kSegChunkMin, (S + 63) / 64) + 1) & ~size_t(1);  // even: whole pair records
  const size_t nch = (S + chunk - 1) / chunk;
  const size_t nsup = (nch + 7) / 8;
  c->seg_chunk = static_cast<int>(chunk);
  c->seg_nch = static_cast<int>(nch);
  c->seg_nsup = static_cast<int>(nsup);
  const size_t seg_words = static_cast<size_t>(seg_cap_offset(static_cast<int>(S))) + 8 * nch + 12 * nsup;
  KC_TRY(c->h_seg.reserve(seg_words));
  KC_TRY(c->d_seg.reserve(seg_words));
  // built in ordinary (cached) host memory -- the table passes read every point several times -- and stored to
  // the device (BAR) or the pinned staging buffer in one copy at the end
  if (c->seg_stage.size() < seg_words) c->seg_stage.resize(seg_words + seg_words / 4 + 16);
  float *h = c->seg_stage.data();
  // rows: whole-row copies (this call is on the host's critical path in front of every cycle launch)
  if (xyz) {
    float *hx = h, *hy = h + S, *hz0 = h + 2 * S;
    for (size_t j = 0; j < S; ++j) {
      hx[j] = xyz[3 * j];
      hy[j] = xyz[3 * j + 1];
      hz0[j] = xyz[3 * j + 2];
    }
  } else {
    std::memcpy(h, x, S * sizeof(float));
    std::memcpy(h + S, y, S * sizeof(float));
    if (z) std::memcpy(h + 2 * S, z, S * sizeof(float));
    else std::memset(h + 2 * S, 0, S * sizeof(float));
  }
  std::memcpy(h + 4 * S, acc, S * sizeof(float));
  uint32_t zbits = 0u;
  {
    const float *hz = h + 2 * S;
    float *hzz = h + 3 * S;
    for (size_t j = 0; j < S; ++j) {
      uint32_t zb;
      std::memcpy(&zb, &hz[j], 4);
      zbits |= zb;
      hzz[j] = hz[j] * hz[j];  // (seg.z - 0)^2 of Path::distance
    }
  }
  const bool flat = zbits == 0u;  // every z is +0.0f exactly (z^2 of -0.0f is +0 as well, but keep the test plain)
  c->seg_flat = flat;
  ++c->seg_version;
  dbg_mark(1, dbg_t);
  const float kInf = std::numeric_limits<float>::infinity();
  auto up = [](double v) {  // to float, rounded up
    return std::nextafter(static_cast<float>(v), std::numeric_limits<float>::infinity());
  };
  auto pt = [&](size_t j, double p[3]) {
    p[0] = h[j];
    p[1] = h[S + j];
    p[2] = h[2 * S + j];
  };
  const segtab::Span span{h, h + S, h + 2 * S};
  {
    float *cap = h + seg_cap_offset(static_cast<int>(S));
    // capsule of the points [j0, j1): chord A -> B of the first and last point as the kernels see it
    // (float A, float AB, float 1/|AB|^2) + the largest deviation of the points from it, rounded up
    auto capsule = [&](size_t j0, size_t j1, float *out, size_t k) {  // record k of `out` (struct Capsule)
      const bool finite = segtab::finite_span(span, j0, j1);
      double A[3], B[3];
      pt(j0, A);
      pt(j1 - 1, B);
      const float ab[3] = {static_cast<float>(B[0] - A[0]), static_cast<float>(B[1] - A[1]),
                           static_cast<float>(B[2] - A[2])};
      const double l2 = static_cast<double>(ab[0]) * ab[0] + static_cast<double>(ab[1]) * ab[1] +
                        static_cast<double>(ab[2]) * ab[2];
      const float inv = (finite && l2 > 0.0 && std::isfinite(1.0 / l2)) ? static_cast<float>(1.0 / l2) : 0.0f;
      double eps = 0.0, mag = 0.0;
      if (finite) segtab::capsule_span(span, j0, j1, A, ab, inv, eps, mag);  // (kc_seg_tables.h: four points at a time)
      eps = std::sqrt(eps);  // sqrt is monotonic and correctly rounded: max of the roots
      float *rec = out + 8 * k;
      rec[0] = static_cast<float>(A[0]);
      rec[1] = static_cast<float>(A[1]);
      rec[2] = finite ? ab[0] : 0.0f;
      rec[3] = finite ? ab[1] : 0.0f;
      rec[4] = inv;
      // deviation of the points from the chord, plus slack for the float chord
      // parameter and coordinate rounding
      rec[5] = finite ? up(eps * (1.0 + 1e-6) + 2e-6 * std::sqrt(l2) + 1e-6 * mag + 1e-30) : kInf;
      rec[6] = static_cast<float>(A[2]);
      rec[7] = finite ? ab[2] : 0.0f;
    };
    float *supc = cap + 8 * nch + 4 * nsup;  // [nsup] records behind the spheres
    float *sup = cap + 8 * nch;
    float seg_len_out = 0.0f;
    // Task ids: [0, nch) chunk capsules | [nch, nch + nsup) super-chunk capsules | [.., + nsup) spheres | last: length.
    auto sphere = [&](size_t s) {
      const size_t j0 = s * 8 * chunk, j1 = std::min(j0 + 8 * chunk, S);
      double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
      const bool finite = segtab::finite_span(span, j0, j1);
      if (finite) segtab::box_span(span, j0, j1, lo, hi);
      if (!finite) {  // never skipped
        sup[s] = sup[nsup + s] = sup[2 * nsup + s] = 0.0f;
        sup[3 * nsup + s] = kInf;
        return;
      }
      // centre stored as float; the radius is taken around the STORED centre
      // and rounded up with slack for the float evaluation on the device
      const float fc[3] = {static_cast<float>(0.5 * (lo[0] + hi[0])),
                           static_cast<float>(0.5 * (lo[1] + hi[1])),
                           static_cast<float>(0.5 * (lo[2] + hi[2]))};
      double r = segtab::radius2_span(span, j0, j1, fc);
      r = std::sqrt(r);
      const double mag = std::fabs(fc[0]) + std::fabs(fc[1]) + std::fabs(fc[2]) + r;
      sup[s] = fc[0];
      sup[nsup + s] = fc[1];
      sup[2 * nsup + s] = fc[2];
      sup[3 * nsup + s] = up(r * (1.0 + 1e-6) + 1e-6 * mag + 1e-30);
    };
    auto length = [&]() { seg_len_out = segtab::length(span, S); };  // View::totalSegmentLength, path.h:85-91
    const size_t ntasks = nch + 2 * nsup + 1;
    auto run_task = [&](size_t t) {
      if (t < nch) capsule(t * chunk, std::min(t * chunk + chunk, S), cap, t);
      else if (t < nch + nsup) capsule((t - nch) * 8 * chunk, std::min((t - nch) * 8 * chunk + 8 * chunk, S), supc, t - nch);
      else if (t < nch + 2 * nsup) sphere(t - nch - nsup);
      else length();
    };
    // (measured: handing these ~40 small tasks to the host pool costs more than it saves -- 7.0 us for the
    // fork / join of 12 threads against 2 us on the calling thread; the rows above are the larger part)
    for (size_t t = 0; t < ntasks; ++t) run_task(t);
    c->seg_len = seg_len_out;
  }
  dbg_mark(2, dbg_t);
  if (!c->trig_direct) std::memcpy(c->h_seg.p, h, seg_words * sizeof(float));  // (the copy command reads pinned memory)
  KC_TRY(upload_table(c, c->d_seg.p, c->trig_direct ? h : c->h_seg.p, seg_words * sizeof(float)));
  if (!c->trig_direct) c->update_busy = true;
  bar_flush(c);
  dbg_mark(3, dbg_t);
  KC_TRY(near_table_ahead(c));
  dbg_mark(4, dbg_t);
  if (c->hprof.on && ++dbg_n % 500 == 0)
    std::fprintf(stderr, "[kc host] set_tracked_segment us: quiesce %.2f rows %.2f tables %.2f upload %.2f near %.2f\n",
                 dbg_sum[0] / dbg_n, dbg_sum[1] / dbg_n, dbg_sum[2] / dbg_n, dbg_sum[3] / dbg_n, dbg_sum[4] / dbg_n);
  return KC_OK;
}
}  // namespace
extern "C" {

int kc_dwa_set_tracked_segment(kc_dwa *c, const float *x, const float *y, const float *z, const float *acc, size_t S,
                               float ref_len) {
  if (!c || (S && (!x || !y || !acc))) KC_FAIL(KC_ERR_INVALID, "null argument");
  return set_tracked_segment_impl(c, x, y, z, nullptr, acc, S, ref_len);
}

int kc_dwa_set_tracked_segment_xyz(kc_dwa *c, const float *xyz, const float *acc, size_t S, float ref_len) {
  if (!c || (S && (!xyz || !acc))) KC_FAIL(KC_ERR_INVALID, "null argument");
  return set_tracked_segment_impl(c, nullptr, nullptr, nullptr, xyz, acc, S, ref_len);
}

// SURVEY 8f rank 4, second half: the interpolated reference path stays on the
// device; a cycle moves the tracked window and a kernel builds the tables.
int kc_dwa_set_path(kc_dwa *c, const float *x, const float *y, const float *z, const float *acc,
                    size_t n, float total_length) {
  if (!c || (n && (!x || !y || !acc))) KC_FAIL(KC_ERR_INVALID, "null argument");
  KC_TRY(use_device(c));
  // the old rows may still be read by a queued window kernel
  KC_HIP(hipStreamSynchronize(c->stream));
  c->update_busy = false;
  c->drained = true;
  c->path_n = n;
  c->path_len = total_length;
  c->path_edge.assign(n > 1 ? n - 1 : 0, 0.0f);
  if (n == 0) return KC_OK;
  KC_TRY(c->d_path.reserve(4 * n));
  std::vector<float> rows(4 * n);
  bool flat = true;
  for (size_t j = 0; j < n; ++j) {
    rows[j] = x[j];
    rows[n + j] = y[j];
    rows[2 * n + j] = z ? z[j] : 0.0f;
    rows[3 * n + j] = acc[j];
    uint32_t zb;
    std::memcpy(&zb, &rows[2 * n + j], 4);
    flat = flat && zb == 0u;
  }
  c->path_flat = flat;
  for (size_t j = 0; j + 1 < n; ++j) {  // the terms of View::totalSegmentLength, path.h:85-91
    const float dx = rows[j] - rows[j + 1], dy = rows[n + j] - rows[n + j + 1],
                dz = rows[2 * n + j] - rows[2 * n + j + 1];
    c->path_edge[j] = std::sqrt(hm::add3(dx * dx, dy * dy, dz * dz));
  }
  KC_HIP(hipMemcpyAsync(c->d_path.p, rows.data(), 4 * n * sizeof(float), hipMemcpyHostToDevice,
                        c->stream));
  KC_HIP(hipStreamSynchronize(c->stream));  // pageable source
  return KC_OK;
}

int kc_dwa_set_tracked_window(kc_dwa *c, size_t start, size_t S) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  if (start > c->path_n || S > c->path_n - start)
    KC_FAIL(KC_ERR_RANGE, "window [%zu, %zu) outside the resident path of %zu points", start,
            start + S, c->path_n);
  KC_TRY(use_device(c));
  c->S = S;
  c->ref_len = c->path_len;
  c->seg_flat = c->path_flat;
  ++c->seg_version;
  if (S == 0) return KC_OK;
  const size_t chunk = (std::max<size_t>(kSegChunkMin, (S + 63) / 64) + 1) & ~size_t(1);  // even: whole pair records
  const size_t nch = (S + chunk - 1) / chunk;
  const size_t nsup = (nch + 7) / 8;
  c->seg_chunk = static_cast<int>(chunk);
  c->seg_nch = static_cast<int>(nch);
  c->seg_nsup = static_cast<int>(nsup);
  const size_t seg_words = static_cast<size_t>(seg_cap_offset(static_cast<int>(S))) + 8 * nch + 12 * nsup;
  if (seg_words > c->d_seg.cap) {  // growing frees the old table: nothing may still read or write it
    KC_HIP(hipStreamSynchronize(c->stream));
    c->update_busy = false;
    c->drained = true;
    KC_TRY(c->d_seg.reserve(seg_words));
    KC_TRY(c->h_seg.reserve(seg_words));
  }
  // View::totalSegmentLength: float sum in index order
  float len = 0.0f;
  for (size_t j = start; j + 1 < start + S; ++j) len += c->path_edge[j];
  c->seg_len = len;
  const size_t n = c->path_n;
  SegWindowArgs a{};
  a.px = c->d_path.p + start;
  a.py = c->d_path.p + n + start;
  a.pz = c->d_path.p + 2 * n + start;
  a.pacc = c->d_path.p + 3 * n + start;
  a.S = static_cast<int>(S);
  a.chunk = static_cast<int>(chunk);
  a.nch = static_cast<int>(nch);
  a.nsup = static_cast<int>(nsup);
  a.seg = c->d_seg.p;
  // stream order: behind the cost kernel of the last cycle, in front of the
  // next (a side stream + event was measured as well: the cross-stream wait costs
  // as much as the kernel it hides)
  KC_TRY(c->timing.start("segment_window_kernel", c->stream));
  hipLaunchKernelGGL(segment_window_kernel, dim3(1), dim3(kSegWinBlock), 0, c->stream, a);
  KC_TRY(c->timing.stop(c->stream));
  KC_HIP(hipGetLastError());
  KC_TRY(near_table_ahead(c));  // (in stream order behind the kernel that writes the table)
  c->seg_busy = true;  // a queued kernel writes d_seg: host stores into the table wait for the stream
  return KC_OK;
}

}  // extern "C"

namespace {
// LDS bytes of the cost tables of the cycle tail (cycle_tabs, kc_cycle_dev.h)
size_t cycle_table_bytes(const CostArgs &ca) {
  size_t b = 0;
  if (ca.use_seg)
    b += 32 * static_cast<size_t>(seg_pairs_padded(ca.nch, ca.seg_chunk)) +
         4 * (8 * static_cast<size_t>(ca.nch) + 12 * static_cast<size_t>(ca.nsup));
  if (ca.use_obs) {
    const size_t ncell = static_cast<size_t>(ca.b.W) * ca.b.H;
    b += 4 * (ncell + 1) + ((ncell + 3) & ~size_t(3));
  }
  return b + 4 * static_cast<size_t>(ca.P) * 4;
}

// parameters of the tilted-octree tests (kc_tilt_dev.h) from the frame captured by kc_dwa_set_scan
int tilt_params(kc_dwa *c, TiltDev &t) {
  if (!c->have_gbits) KC_FAIL(KC_ERR_STATE, "tilted sensor frame without a voxel bitmap");
  std::memset(&t, 0, sizeof(t));
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) t.R[i][j] = c->frame.R[i][j];
    t.t[i] = c->frame.t[i];
  }
  t.res = c->res;
  t.inv = 1.0 / c->res;
  t.h = c->res / 2.0;
  t.kz = c->tilt_kz;
  t.shape = c->prm.shape;
  t.radius = c->radius;
  t.hh = c->height / 2.0;
  t.a = static_cast<double>(c->prm.dims[0]) / 2.0;
  t.b = static_cast<double>(c->prm.dims[1]) / 2.0;
  t.c = static_cast<double>(c->prm.dims[2]) / 2.0;
  if (c->prm.shape == KC_SPHERE) t.rho = c->radius;
  else if (c->prm.shape == KC_BOX) t.rho = std::sqrt(t.a * t.a + t.b * t.b + t.c * t.c);
  else t.rho = std::sqrt(c->radius * c->radius + t.hh * t.hh);
  t.gbits = c->d_gbits.p;
  t.gkx0 = c->gkx0;
  t.gky0 = c->gky0;
  t.gH = c->gH;
  t.gwpr = c->gwpr;
  return KC_OK;
}

// kc_dwa_rollout, or -- want_cycle -- the whole cycle in one launch when the
// cost tables fit beside the roll-out tile (c->cycle_launched tells)
int rollout_impl(kc_dwa *c, const kc_state *start, size_t P, bool want_cycle, bool trig_ready = false) {
  if (!c || !start) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (P < 2 || P > c->prm.max_points)
    KC_FAIL(KC_ERR_RANGE, "num_points %zu outside [2, %zu]", P,
            c->prm.max_points);
  KC_TRY(use_device(c));
  hipStream_t s = c->stream;
  // The staging buffers of the last cycle must be free.  When the host has
  // already seen the record the last cost kernel publishes at its very end,
  // everything in front of it has completed and the (slow) stream wait is
  // skipped; commands queued since then only read buffers this call leaves alone.
  if (!c->drained || c->timing.enabled) KC_HIP(hipStreamSynchronize(s));
  c->drained = false;
  if (!c->in_materialise) c->timing.begin_cycle();
  c->P = P;
  c->rolled = false;
  c->evaluated = false;
  c->external = false;
  c->have_vel = false;
  c->cycle_launched = false;
  c->slots_pending = false;
  c->paths_valid = true;
  c->last_start = *start;
  const size_t n = c->shard_count;
  c->n_roll = n;
  if (n == 0) {
    c->rolled = true;
    return KC_OK;
  }
  // trig table: cos/sin of yaw_k for every omega row, from the host libm the
  // reference calls (path.h:24-30); yaw_k by repeated addition of omega * dt
  const size_t A = c->lat.omega_values.size();
  KC_TRY(c->h_trig.reserve(A * P));
  KC_TRY(c->d_trig.reserve(A * P));
  const double dt = static_cast<double>(static_cast<float>(c->prm.time_step));
  // Where the table is written: straight into device memory when the host
  // can address it (large BAR: write-combined stores, no copy command and no
  // copy engine latency on the critical path), else into pinned memory
  // followed by an H2D copy.
  const double yaw0 = start->yaw;
  const double *om_v = c->lat.omega_values.data();
  double2 *tab = c->trig_direct ? c->d_trig.p : c->h_trig.p;
  auto trig_rows = [=](size_t r0, size_t r1) {
    // a worker's rows are computed into a small local tile and written out
    // as one contiguous run per step (the table is step-major: the kernels
    // read consecutive omega rows with consecutive lanes)
    constexpr size_t kTileRows = 16;
    double2 tile[kTileRows];
    double yaw[kTileRows];
    for (size_t rb = r0; rb < r1; rb += kTileRows) {
      const size_t nr = std::min(kTileRows, r1 - rb);
      for (size_t i = 0; i < nr; ++i) yaw[i] = yaw0;
      for (size_t k = 0; k < P; ++k) {
        for (size_t i = 0; i < nr; ++i) {
          double sn, cs;
          ::sincos(yaw[i], &sn, &cs);  // bit-identical to sin()/cos() (tested)
          tile[i] = make_double2(cs, sn);
          yaw[i] += om_v[rb + i] * dt;
        }
        std::memcpy(tab + k * A + rb, tile, nr * sizeof(double2));
      }
    }
#if defined(__x86_64__)
    __builtin_ia32_sfence();  // write-combined stores leave the core before "done"
#endif
  };
  // Device trig (kc_trig_exact.h): the kernels form cos / sin(yaw_k) themselves -- no host table at all.  Only
  // while every yaw_k stays inside the range the restated algorithm covers (|yaw| < 105414350; a bound on
  // |yaw0| + P |omega| dt decides), and only when the restatement agreed with the installed libm when the
  // library was loaded.  Otherwise -- the FALLBACK -- the host fills the table with its libm (the worker pool of
  // kc_set_host_threads shares the rows), in front of the launch: no kernel ever waits for the host.
  bool dev_trig = c->device_trig && trig_selfcheck_ok() && std::isfinite(yaw0);
  if (dev_trig) {
    double om_max = 0.0;
    for (size_t i = 0; i < A; ++i) om_max = std::max(om_max, std::fabs(om_v[i]));
    const double reach = std::fabs(yaw0) + om_max * dt * static_cast<double>(P);
    dev_trig = std::isfinite(reach) && reach < 1.0e8;
  }
  // ... or the table is there already: formed inside the launch of the sensor update this cycle follows
  // (plan_trig_job), for this yaw, this lattice and this horizon
  bool table_ahead = false;
  if (dev_trig && c->trig_ahead_valid) {
    table_ahead = P == c->trig_ahead_P && c->lat_version == c->trig_ahead_lat && c->d_trig.cap >= A * P &&
                  std::memcmp(&yaw0, &c->trig_ahead_yaw, sizeof(double)) == 0;
  }
  if (!table_ahead) c->trig_ahead_valid = false;  // (d_trig is about to be rewritten, or belongs to another pose)
  if (!dev_trig && !trig_ready) {
    WorkerPool::instance().parallel_for(A, 2, trig_rows);
    c->timing.mark("host:trig_table");
    if (!c->trig_direct)
      KC_HIP(hipMemcpyAsync(c->d_trig.p, c->h_trig.p, A * P * sizeof(double2), hipMemcpyHostToDevice, s));
  }
  trig_ready = true;
  c->hprof.mark(9);
  RollArgs a{};
  KC_TRY(ensure_cycle_buffers(c, n, P));
  a.n = static_cast<int>(n);
  a.first = static_cast<int>(c->shard_first);
  a.P = static_cast<int>(P);
  a.A = static_cast<int>(A);
  a.x0 = start->x;
  a.y0 = start->y;
  a.dt = dt;
  a.vxt = c->d_vxt.p;
  a.vyt = c->d_vyt.p;
  a.vidx = c->d_vidx.p;
  a.row = c->d_row.p;
  a.trig = c->d_trig.p;
  a.trig_dev = (dev_trig && !table_ahead) ? 1 : 0;
  if (dev_trig && !c->d_sincostab.p) {
    KC_TRY(c->d_sincostab.reserve(440));
    KC_HIP(hipMemcpyAsync(c->d_sincostab.p, kc_sincostab_host, sizeof(kc_sincostab_host), hipMemcpyHostToDevice, s));
    KC_HIP(hipStreamSynchronize(s));
  }
  a.sincostab = c->d_sincostab.p;
  a.yaw0 = yaw0;
  a.trig_out = c->d_trig.p;
  a.omega_values = c->d_omega.p;
  a.px = c->d_px.p;
  a.py = c->d_py.p;
  a.flags = c->d_flags.p;
  a.adm_list = c->d_adm.p;
  a.adm_count = c->d_result.p + W_LIST;
  c->freeze_valid = false;
  if (!c->drop_samples) {
    KC_TRY(c->d_freeze.reserve(n));
    KC_TRY(c->d_frz.reserve(2 * n));
    if (!c->d_omega.p || c->d_omega.cap < A) KC_TRY(upload_omega(c));
    a.freeze = 1;
    a.num_ctrl = static_cast<int>(std::min<size_t>(c->num_ctrl_points, 0x3FFFFFFF));
    a.freeze_step = c->d_freeze.p;
    a.frz_smooth = c->d_frz.p;
    a.frz_jerk = c->d_frz.p + n;
    a.omega_values = c->d_omega.p;
    a.acc0 = c->prm.acc_limits[0];
    a.acc1 = c->prm.acc_limits[1];
    a.acc2 = c->prm.acc_limits[2];
    c->freeze_valid = true;
  }
  const bool may_collide = c->have_sensor && any_voxel(c);
  KC_TRY(window_geometry(c, start->x, start->y, cycle_reach(c), a.c));
  // single-launch cycle: cost arguments up front (their checks must not fail
  // behind a launched kernel)
  CycleTail tail{};
  // One launch pays while every workgroup of the shard is resident at once (32 samples per
  // workgroup, one workgroup per CU: 8192 samples on an MI355X -- the per-GPU share of every
  // BASELINE config on 8 GPUs).  Beyond, the cycle kernel's LDS footprint (one workgroup per CU)
  // loses to the three-kernel cycle, whose roll-out kernel fits two per CU (cfg5 on ONE GPU,
  // 65536 samples: 0.214 against 0.129 ms).
  // And a small shard with many survivors (cfg1: 128 samples in 4 workgroups, 104 admissible) is
  // better served by the stand-alone cost kernels, which spread the survivors over all CUs; the
  // admissible count of the previous cycle is the predictor (as for the choice of cost kernel).
  // 32 samples per workgroup; 16 when that would leave half of the CUs without one (a 4096-sample
  // shard -- cfg3 split over 8 GPUs -- or any mid-size lattice): twice the workgroups, half the poses
  // and survivors in each.  (Option "cycle_samples": 0 = this rule, 16 / 32 = fixed.)
  int cs = c->cycle_samples_opt;
  if (cs == 0) cs = 2 * blocks_for(n, 32) <= static_cast<unsigned>(c->num_cus) ? 16 : 32;
  // the last arriver of the ticket epilogue holds two workgroup keys per lane (kc_cycle_dev.h): at most
  // 2048 workgroups, whatever the option says (65536 samples in 16-sample workgroups would be 4096)
  if (blocks_for(n, static_cast<unsigned>(cs)) > 2048u) cs = 32;
  c->cycle_samples = cs;
  const unsigned cyc_G = blocks_for(n, static_cast<unsigned>(cs));
  const bool cyc_wave = cyc_G <= static_cast<unsigned>(c->num_cus);
  const bool cyc_few = c->last_nadm < 0 || c->last_nadm <= 4ll * cyc_G;
  const bool cyc_full = 2 * cyc_G >= static_cast<unsigned>(c->num_cus);
  const bool sphere_ok = c->prm.shape != KC_SPHERE || (c->have_gbits && c->gz_valid);  // (fused path)
  bool cycle = want_cycle && c->cycle_fused && sphere_ok && n <= 1024u * kCompactMaxPer &&
               (c->cycle_forced || (cyc_wave && (cyc_few || cyc_full)));
  if (cycle) {
    // workgroups with more than a handful of survivors search wavefront-per-sample: through the
    // near table when the last cycle had that many
    c->near_ok = false;
    c->near_wanted = c->last_nadm < 0 || c->last_nadm > 2ll * cyc_G;
    c->onear_ok = false;
    if (c->near_wanted) {
      KC_TRY(ensure_near_table(c, start->x, start->y));
      KC_TRY(ensure_onear(c, start->x, start->y));
    }
    KC_TRY(build_cost_args(c, n, c->shard_first, tail.c, tail.t));
  }
  // fused path: trig rows + poses (64 x P double2) and the window bits in LDS
  // Roll-out tile of the three-kernel cycle: 32 samples per workgroup; 1024 threads, or 512 for a large
  // lattice of short trajectories (cfg5, 65536 x 50: more workgroups resident per CU hide the serial
  // recurrence of each other, 80 -> 45 us; P = 100 or one resident round: 1024 is better, tools/fused_cfg_sweep.sh)
  int plain_fb = c->fused_block;
  if (!c->fused_shape_fixed && P <= 64 && blocks_for(n, 32) > 4u * static_cast<unsigned>(c->num_cus)) plain_fb = 512;
  const int fs = cycle ? cs : c->fused_samples, fb = cycle ? 1024 : plain_fb;
  const size_t pos_bytes = static_cast<size_t>(fs) * (P | 1) * sizeof(double2);
  size_t bits_bytes =
      (a.c.enabled ? static_cast<size_t>(a.c.H) * a.c.wpr * 4 * (a.c.dil ? 3 : 1) : 0) +
      static_cast<size_t>(fs) * P * sizeof(int);  // + queue of undecided poses
  const bool fused = sphere_ok && !c->tilted && (!a.c.enabled || c->have_gbits) &&
                     pos_bytes + bits_bytes + 512 <= c->lds_limit;
  const size_t tab_off = (pos_bytes + bits_bytes + 15) & ~size_t(15);
  cycle = cycle && fused && tab_off + cycle_table_bytes(tail.c) + 2048 <= c->lds_limit;
  if (want_cycle && !cycle && fused && (fs != c->fused_samples || fb != plain_fb))
  {
    // sized for the cycle shape: start over for the plain one (a host-built table stays valid: same pose, same rows)
    return rollout_impl(c, start, P, false, true);
  }
  c->need_compact = !fused || cycle;
  if (dev_trig && !fused && !table_ahead) {  // the split path's kernels read a table: filled on the device, in stream order
    KC_TRY(c->timing.start("trig_table_kernel", s));
    TrigJob tj{};
    tj.yaw0 = yaw0;
    tj.dt = dt;
    tj.omega = c->d_omega.p;
    tj.tab = c->d_sincostab.p;
    tj.out = c->d_trig.p;
    tj.A = static_cast<int>(A);
    tj.P = static_cast<int>(P);
    tj.nblk = static_cast<int>(std::min<size_t>(1024, blocks_for(A * P, kTrigBlock)));
    hipLaunchKernelGGL(trig_table_kernel, dim3(tj.nblk), dim3(kTrigBlock), 0, s, tj);
    KC_TRY(c->timing.stop(s));
  }
  if (fused) {
    if (!c->perm_valid || c->perm_first != c->shard_first || c->perm_count != c->shard_count ||
        (cycle && c->perm_cs != cs))
      KC_TRY(build_perm(c));
    a.perm = cycle ? c->d_cperm.p : c->d_perm.p;
    a.prow = cycle ? c->d_cprow.p : c->d_prow.p;
    a.pvi = cycle ? c->d_cpvi.p : c->d_pvi.p;
#ifdef KC_PHASE_STAMPS
    if (c->debug_stamps) {
      KC_TRY(c->d_dbg2.reserve(512 * 32));
      KC_HIP(hipMemsetAsync(c->d_dbg2.p, 0, 512 * 32 * 8, s));
      a.dbg = c->d_dbg2.p;
    }
#endif
    if (c->list_dirty)  // previous roll-out was never evaluated: re-arm the list (and the error word a
                        // failed cycle may have left)
      KC_HIP(hipMemsetAsync(c->d_result.p + W_NADM, 0, 3 * sizeof(long long), s));
    c->list_dirty = !cycle;
    a.c.lds = 1;
    if (cycle) {
      const unsigned G = blocks_for(n, fs);
      KC_TRY(c->d_block_keys.reserve(std::max<size_t>(512, 2 * static_cast<size_t>(G))));
      {
        const size_t words = n / 32 + 2;
        const uint32_t *before = c->d_adm_bits.p;
        KC_TRY(c->d_adm_bits.reserve(words));
        if (c->d_adm_bits.p != before)  // a fresh bitmap starts clear; the last workgroup keeps it so
          KC_HIP(hipMemsetAsync(c->d_adm_bits.p, 0, c->d_adm_bits.cap * sizeof(uint32_t), s));
      }
      KC_TRY(c->h_wrow.reserve(static_cast<size_t>(G) * 2 * P));
      tail.tab_off = static_cast<unsigned>(tab_off);
      tail.write_paths = c->write_paths ? 1 : 0;
      tail.block_keys = c->d_block_keys.p;
      tail.adm_bits = c->d_adm_bits.p;
      tail.result = c->d_result.p;
      tail.host_pub = c->sharded_call ? nullptr : c->h_pub.p;
      tail.host_rows = c->sharded_call ? nullptr : c->h_wrow.p;
      tail.host_slots = nullptr;
      if (!c->sharded_call && c->host_reduce) {
        KC_TRY(c->h_slots.reserve(4 * static_cast<size_t>(G)));
        tail.host_slots = c->h_slots.p;
        tail.host_pub = nullptr;
      }
      tail.seq = ++c->seq;
      tail.c.block_keys = c->d_block_keys.p;
      // sharded call: the last workgroup also writes this rank's words of the exchange record (no pack launch)
      // (cycle_epilogue holds kMaxWords x kBlock = 2048 32-bit words of the bitmap in registers: a wider region --
      // a share beyond 65536 samples -- is packed by xchg_pack_kernel behind the cycle instead)
      tail.xs = (c->sharded_call && 2 * static_cast<size_t>(c->xchg_rw) <= 2048) ? c->xchg_send : nullptr;
      tail.xgid = c->rows_active ? c->d_gid.p : nullptr;
      tail.xrank = c->xchg_rank;
      tail.xrw = c->xchg_rw;
      c->xchg_packed = tail.xs != nullptr;
      a.dev_err = c->d_result.p + W_NADM;
    }
    c->hprof.mark(1);
    KC_TRY(c->timing.start(cycle ? "cycle_kernel" : "rollout_collide_kernel", s));
    const dim3 grid(blocks_for(n, fs)), block(fb);
    const size_t smem = pos_bytes + bits_bytes;
    const NoTail nt{};
    if (cycle && cs == 16)
      hipLaunchKernelGGL((rollout_collide_kernel<16, 1024, CycleTail>), grid, block,
                         tab_off + cycle_table_bytes(tail.c), s, a, tail);
    else if (cycle)
      hipLaunchKernelGGL((rollout_collide_kernel<32, 1024, CycleTail>), grid, block,
                         tab_off + cycle_table_bytes(tail.c), s, a, tail);
    else if (fs == 16 && fb == 256) hipLaunchKernelGGL((rollout_collide_kernel<16, 256>), grid, block, smem, s, a, nt);
    else if (fs == 16 && fb == 512) hipLaunchKernelGGL((rollout_collide_kernel<16, 512>), grid, block, smem, s, a, nt);
    else if (fs == 32 && fb == 1024) hipLaunchKernelGGL((rollout_collide_kernel<32, 1024>), grid, block, smem, s, a, nt);
    else if (fs == 64 && fb == 1024) hipLaunchKernelGGL((rollout_collide_kernel<64, 1024>), grid, block, smem, s, a, nt);
    else hipLaunchKernelGGL((rollout_collide_kernel<32, 512>), grid, block, smem, s, a, nt);
    KC_TRY(c->timing.stop(s));
    if (cycle) {
      c->cycle_launched = true;
      c->paths_valid = c->write_paths;
      c->slots_pending = tail.host_slots != nullptr;
      c->slots_G = grid.x;
      c->pub_pending = !c->slots_pending;
      c->device_record_valid = !c->slots_pending;
      c->row_valid = false;
    }
    c->timing.mark("host:launch_rollout");
    c->hprof.mark(2);
  } else {
    // split path (sphere, very long horizons, windows beyond LDS): roll-out
    // first, window bits built on the host while it runs, then the pose-
    // parallel collision pass
    CollDev geom = a.c;
    if (may_collide) {
      KC_TRY(c->d_pos.reserve(n * P));
      a.pos = c->d_pos.p;
    }
    a.c.enabled = may_collide ? 1 : 0;  // roll-out: "store the double poses"
    if (a.freeze) {
      KC_TRY(c->d_first_hit.reserve(n));
      a.first_hit = c->d_first_hit.p;
    }
    const size_t tile_bytes = 2 * static_cast<size_t>(kRollBlock) * (P | 1) * 4;
    a.stage = (tile_bytes <= 64 * 1024) ? 1 : 0;
    KC_TRY(c->timing.start("rollout_kernel", s));
    hipLaunchKernelGGL(rollout_kernel, dim3(blocks_for(n, kRollBlock)),
                       dim3(kRollBlock), a.stage ? tile_bytes : 0, s, a);
    KC_TRY(c->timing.stop(s));
    c->timing.mark("host:launch_rollout");
    if (may_collide && c->tilted) {
      // tilted octree frame: every pose against the voxel columns within its reach, exact 3-D tests
      TiltArgs ta{};
      KC_TRY(tilt_params(c, ta.c));
      if (c->tilt_cropped) {
        // the cropped window (upload_voxels) must hold every column a pose of this roll-out can touch: the start's
        // distance from the update's pose + the horizon's reach + the robot's bounding radius, in columns
        const double far = std::hypot(start->x - c->tilt_body_x, start->y - c->tilt_body_y) + cycle_reach(c) + ta.c.rho;
        if (!(far * c->inv_res + 4.0 < static_cast<double>(kTiltCrop)))
          KC_FAIL(KC_ERR_UNSUPPORTED, "tilted sensor frame: the roll-out reaches %.0f voxel columns from the pose of the scan, "
                                      "beyond the %d kept of a scan that spans more than 8192", far * c->inv_res, kTiltCrop);
      }
      ta.pos = c->d_pos.p;
      ta.trig = c->d_trig.p;
      ta.row = c->d_row.p;
      ta.n = static_cast<int>(n);
      ta.first = static_cast<int>(c->shard_first);
      ta.P = static_cast<int>(P);
      ta.A = static_cast<int>(A);
      ta.flags = c->d_flags.p;
      ta.first_hit = a.first_hit;
      KC_TRY(c->timing.start("collision_tilted_kernel", s));
      hipLaunchKernelGGL(collision_tilted_kernel, dim3(blocks_for(n * (P - 1), 256)), dim3(256), 0, s, ta);
      KC_TRY(c->timing.stop(s));
    } else if (may_collide) {
      a.c = geom;
      KC_TRY(window_bits_host(c, a.c));
      c->timing.mark("host:window_bits");
      if (a.c.enabled) {
        const size_t bb = static_cast<size_t>(a.c.H) * a.c.wpr * 4;
        KC_TRY(c->timing.start("collision_kernel", s));
        hipLaunchKernelGGL(collision_kernel,
                           dim3(blocks_for(n * (P - 1), kCollBlock)),
                           dim3(kCollBlock), a.c.lds ? bb : 0, s, a);
        KC_TRY(c->timing.stop(s));
      }
    }
    if (a.freeze)  // (no collision pass: first_hit stays INT_MAX everywhere, nothing is frozen)
      hipLaunchKernelGGL(freeze_fixup_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, a);
  }
  KC_HIP(hipGetLastError());
  c->timing.mark("host:launch_collision");
  c->rolled = true;
  if (c->cycle_launched) c->evaluated = true;
  return KC_OK;
}

// the float rows of the last roll-out, when a single-launch cycle left them out:
// the same roll-out again through the materialising kernel (same inputs, same
// bits); costs and result of the cycle stay
int materialise_paths(kc_dwa *c) {
  if (c->paths_valid) return KC_OK;
  const bool evaluated = c->evaluated, have_last = c->have_last, pub = c->pub_pending, row = c->row_valid,
             was_cycle = c->cycle_launched;
  const kc_result last = c->last;
  const kc_state st = c->last_start;
  if (c->pub_pending) KC_HIP(hipStreamSynchronize(c->stream));  // the cycle itself must be through
  c->in_materialise = true;
  const int rc = rollout_impl(c, &st, c->P, false);
  c->in_materialise = false;
  KC_TRY(rc);
  c->evaluated = evaluated;
  c->have_last = have_last;
  c->last = last;
  c->pub_pending = pub;
  c->row_valid = row;
  c->cycle_launched = was_cycle;
  return KC_OK;
}
}  // namespace

extern "C" {

int kc_dwa_rollout(kc_dwa *c, const kc_state *start, size_t P) {
  return rollout_impl(c, start, P, false);
}

int kc_dwa_check_poses(kc_dwa *c, const double *x, const double *y,
                       const double *yaw, size_t n, uint8_t *hit_out) {
  if (!c || (n && (!x || !y || !yaw || !hit_out)))
    KC_FAIL(KC_ERR_INVALID, "null argument");
  if (n == 0) return KC_OK;
  if (n > 0x7FFFFFFFul) KC_FAIL(KC_ERR_RANGE, "too many poses");
  KC_TRY(use_device(c));
  hipStream_t s = c->stream;
  KC_HIP(hipStreamSynchronize(s));
  double reach = 0.0;
  for (size_t i = 1; i < n; ++i)
    reach = std::max(reach, std::hypot(x[i] - x[0], y[i] - y[0]));
  if (c->tilted) {
    if (!c->have_sensor || c->vox_kx.empty()) {
      std::memset(hit_out, 0, n);
      return KC_OK;
    }
    TiltDev td;
    KC_TRY(tilt_params(c, td));
    if (c->tilt_cropped) {  // (see rollout_impl: every pose inside the kept window of the cropped scan)
      double far = 0.0;
      for (size_t i = 0; i < n; ++i) far = std::max(far, std::hypot(x[i] - c->tilt_body_x, y[i] - c->tilt_body_y));
      if (!((far + td.rho) * c->inv_res + 4.0 < static_cast<double>(kTiltCrop)))
        KC_FAIL(KC_ERR_UNSUPPORTED, "tilted sensor frame: a pose lies %.0f voxel columns from the pose of the scan, beyond the "
                                    "%d kept of a scan that spans more than 8192", far * c->inv_res, kTiltCrop);
    }
    KC_TRY(c->h_trig.reserve(2 * n));
    KC_TRY(c->d_trig.reserve(2 * n));
    for (size_t i = 0; i < n; ++i) {
      c->h_trig.p[i] = make_double2(x[i], y[i]);
      c->h_trig.p[n + i] = make_double2(std::cos(yaw[i]), std::sin(yaw[i]));
    }
    KC_HIP(hipMemcpyAsync(c->d_trig.p, c->h_trig.p, 2 * n * sizeof(double2), hipMemcpyHostToDevice, s));
    KC_TRY(c->d_flags.reserve(n));
    hipLaunchKernelGGL(pose_check_tilted_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, td, c->d_trig.p,
                       c->d_trig.p + n, static_cast<int>(n), c->d_flags.p);
    KC_HIP(hipGetLastError());
    KC_HIP(hipMemcpyAsync(hit_out, c->d_flags.p, n, hipMemcpyDeviceToHost, s));
    KC_HIP(hipStreamSynchronize(s));
    c->rolled = false;
    c->evaluated = false;
    return KC_OK;
  }
  CollDev cd;
  KC_TRY(build_window_at(c, x[0], y[0], reach * 1.0001 + 1e-9, cd));
  if (!cd.enabled) {
    std::memset(hit_out, 0, n);
    return KC_OK;
  }
  cd.lds = 0;
  KC_TRY(c->h_trig.reserve(2 * n));
  KC_TRY(c->d_trig.reserve(2 * n));
  for (size_t i = 0; i < n; ++i) {
    c->h_trig.p[i] = make_double2(x[i], y[i]);
    c->h_trig.p[n + i] = make_double2(std::cos(yaw[i]), std::sin(yaw[i]));
  }
  KC_HIP(hipMemcpyAsync(c->d_trig.p, c->h_trig.p, 2 * n * sizeof(double2),
                        hipMemcpyHostToDevice, s));
  KC_TRY(c->d_flags.reserve(n));
  hipLaunchKernelGGL(pose_check_kernel, dim3(blocks_for(n, 256)), dim3(256), 0,
                     s, cd, c->d_trig.p, c->d_trig.p + n, static_cast<int>(n),
                     c->d_flags.p);
  KC_HIP(hipGetLastError());
  KC_HIP(hipMemcpyAsync(hit_out, c->d_flags.p, n, hipMemcpyDeviceToHost, s));
  KC_HIP(hipStreamSynchronize(s));
  c->rolled = false;  // the flag buffer no longer describes a roll-out
  c->evaluated = false;
  return KC_OK;
}

int kc_dwa_evaluate(kc_dwa *c) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  if (!c->rolled) KC_FAIL(KC_ERR_STATE, "kc_dwa_rollout has not run");
  KC_TRY(use_device(c));
  KC_TRY(materialise_paths(c));
  c->drained = false;  // queued work reads the per-update tables again
  KC_TRY(run_evaluate(c, c->n_roll, c->shard_first));
  c->timing.mark("host:launch_evaluate");
  c->evaluated = true;
  return KC_OK;
}

int kc_dwa_fetch_result(kc_dwa *c, kc_result *out) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  if (!c->evaluated) KC_FAIL(KC_ERR_STATE, "kc_dwa_evaluate has not run");
  KC_TRY(use_device(c));
  return fetch(c, out, c->n_roll);
}

int kc_dwa_cycle(kc_dwa *c, const kc_state *start, size_t P, kc_result *out) {
  if (c) c->hprof.mark(0);
  KC_TRY(rollout_impl(c, start, P, true));
  if (!c->cycle_launched) KC_TRY(kc_dwa_evaluate(c));
  c->hprof.mark(5);
  const int rc = kc_dwa_fetch_result(c, out);
  c->hprof.mark(7);
  c->hprof.close();
  return rc;
}

int kc_dwa_find_best_path(kc_dwa *c, const kc_state *st, const kc_step_inputs *in, kc_result *out) {
  if (!c || !st || !in || !out) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (in->limits)
    KC_TRY(kc_dwa_sample_window(c, in->ctr_type, in->limits, in->cur_vx, in->cur_vy, in->cur_omega, in->max_linear_samples,
                                in->max_angular_samples, nullptr, nullptr, nullptr, nullptr, 0));
  if (in->points_xyz)
    KC_TRY(kc_dwa_set_points(c, st, in->points_xyz, in->n_points, in->max_sensor_range));
  else if (in->scan_ranges && in->scan_angles)
    KC_TRY(kc_dwa_set_scan(c, st, in->scan_ranges, in->scan_angles, in->n_beams, in->max_sensor_range));
  if (in->seg_size) {
    if (in->seg_xyz) KC_TRY(kc_dwa_set_tracked_segment_xyz(c, in->seg_xyz, in->acc_at_seg, in->seg_size, in->ref_path_length));
    else KC_TRY(kc_dwa_set_tracked_segment(c, in->seg_x, in->seg_y, in->seg_z, in->acc_at_seg, in->seg_size, in->ref_path_length));
  }
  return kc_dwa_cycle(c, st, in->num_points, out);
}

int kc_dwa_get_best(kc_dwa *c, float *path_x, float *path_y, float *vvx,
                    float *vvy, float *vom) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  if (!c->have_last || !c->last.found)
    KC_FAIL(KC_ERR_STATE, "no trajectory found in the last cycle");
  KC_TRY(use_device(c));
  const size_t P = c->P;
  if (c->last_lat < 0)
    KC_FAIL(KC_ERR_STATE, "winner %lld is not on this shard",
            static_cast<long long>(c->last.raw_index));
  const size_t local = static_cast<size_t>(c->last_lat) - (c->external ? 0 : c->shard_first);
  if (local >= c->n_roll)
    KC_FAIL(KC_ERR_STATE, "winner %lld is not on this shard",
            static_cast<long long>(c->last.raw_index));
  if (c->row_valid) {  // single-launch cycle: the row came with the record, no copy, no stream wait
    if (path_x) std::memcpy(path_x, c->h_wrow.p + c->wrow_off, P * sizeof(float));
    if (path_y) std::memcpy(path_y, c->h_wrow.p + c->wrow_off + P, P * sizeof(float));
  } else {
    KC_TRY(materialise_paths(c));
    KC_TRY(c->h_row.reserve(2 * P));
    KC_HIP(hipMemcpyAsync(c->h_row.p, c->d_px.p + local * P, P * sizeof(float),
                          hipMemcpyDeviceToHost, c->stream));
    KC_HIP(hipMemcpyAsync(c->h_row.p + P, c->d_py.p + local * P,
                          P * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    KC_HIP(hipStreamSynchronize(c->stream));
    if (path_x) std::memcpy(path_x, c->h_row.p, P * sizeof(float));
    if (path_y) std::memcpy(path_y, c->h_row.p + P, P * sizeof(float));
  }
  if (vvx || vvy || vom) {
    if (c->external)
      KC_FAIL(KC_ERR_STATE, "velocities belong to the caller in evaluate mode");
    const size_t g = static_cast<size_t>(c->last_lat);
    // TrajectoryVelocities2D::add: float = double (trajectory.h:96-103)
    const float fx = static_cast<float>(c->lat.vx(g));
    const float fy = static_cast<float>(c->lat.vy(g));
    const float fo = static_cast<float>(c->lat.omega(g));
    // drop_samples = false: a frozen winner's profile is zero from its freeze step on (trajectory_sampler.cpp:160-163)
    size_t fstep = P;
    if (!c->drop_samples && c->freeze_valid) {
      int fs = 0;
      KC_HIP(hipMemcpyAsync(&fs, c->d_freeze.p + local, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      KC_HIP(hipStreamSynchronize(c->stream));
      if (fs > 0) fstep = static_cast<size_t>(fs);
    }
    for (size_t i = 0; i + 1 < P; ++i) {
      const bool z = i >= fstep;
      if (vvx) vvx[i] = z ? 0.0f : fx;
      if (vvy) vvy[i] = z ? 0.0f : fy;
      if (vom) vom[i] = z ? 0.0f : fo;
    }
  }
  return KC_OK;
}

int kc_dwa_get_sample_velocity(kc_dwa *c, int64_t raw, double *vx, double *vy,
                               double *omega) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  const hm::VelocityLattice &fl = full_list(c);
  if (raw < 0 || static_cast<size_t>(raw) >= fl.size())
    KC_FAIL(KC_ERR_RANGE, "sample %lld outside the %zu samples",
            static_cast<long long>(raw), fl.size());
  const size_t g = static_cast<size_t>(raw);
  if (vx) *vx = fl.vx(g);
  if (vy) *vy = fl.vy(g);
  if (omega) *omega = fl.omega(g);
  return KC_OK;
}

int kc_dwa_get_samples(kc_dwa *c, float *paths_x, float *paths_y,
                       int32_t *raw_index, float *costs, size_t cap_rows,
                       size_t *n_rows_out) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  if (!c->rolled) KC_FAIL(KC_ERR_STATE, "kc_dwa_rollout has not run");
  KC_TRY(use_device(c));
  if (paths_x || paths_y) KC_TRY(materialise_paths(c));
  const size_t n = c->n_roll, P = c->P;
  std::vector<uint8_t> flags(n);
  std::vector<float> hx, hy, hc;
  KC_HIP(hipStreamSynchronize(c->stream));
  if (n) {
    KC_HIP(hipMemcpy(flags.data(), c->d_flags.p, n, hipMemcpyDeviceToHost));
    if (paths_x) {
      hx.resize(n * P);
      KC_HIP(hipMemcpy(hx.data(), c->d_px.p, n * P * 4, hipMemcpyDeviceToHost));
    }
    if (paths_y) {
      hy.resize(n * P);
      KC_HIP(hipMemcpy(hy.data(), c->d_py.p, n * P * 4, hipMemcpyDeviceToHost));
    }
    if (costs) {
      if (!c->evaluated) KC_FAIL(KC_ERR_STATE, "costs need kc_dwa_evaluate");
      hc.resize(n);
      KC_HIP(hipMemcpy(hc.data(), c->d_costs.p, n * 4, hipMemcpyDeviceToHost));
    }
  }
  size_t row = 0;
  for (size_t i = 0; i < n; ++i) {
    if (!flags[i]) continue;
    if (row < cap_rows) {
      if (paths_x) std::memcpy(paths_x + row * P, hx.data() + i * P, P * 4);
      if (paths_y) std::memcpy(paths_y + row * P, hy.data() + i * P, P * 4);
      if (raw_index)
        raw_index[row] = static_cast<int32_t>(
            c->external ? static_cast<int64_t>(i) : global_of(c, static_cast<int64_t>(i + c->shard_first)));
      if (costs) costs[row] = hc[i];
    }
    ++row;
  }
  if (n_rows_out) *n_rows_out = row;
  return KC_OK;
}

int kc_dwa_get_freeze_steps(kc_dwa *c, int32_t *steps, size_t cap_rows, size_t *n_rows_out) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  if (!c->rolled || c->external) KC_FAIL(KC_ERR_STATE, "kc_dwa_rollout has not run");
  KC_TRY(use_device(c));
  const size_t n = c->n_roll;
  std::vector<uint8_t> flags(n);
  std::vector<int> fz(n, 0);
  KC_HIP(hipStreamSynchronize(c->stream));
  if (n) {
    KC_HIP(hipMemcpy(flags.data(), c->d_flags.p, n, hipMemcpyDeviceToHost));
    if (!c->drop_samples && c->freeze_valid)
      KC_HIP(hipMemcpy(fz.data(), c->d_freeze.p, n * sizeof(int), hipMemcpyDeviceToHost));
  }
  size_t row = 0;
  for (size_t i = 0; i < n; ++i) {
    if (!flags[i]) continue;
    if (steps && row < cap_rows) steps[row] = fz[i];
    ++row;
  }
  if (n_rows_out) *n_rows_out = row;
  return KC_OK;
}

// caller-provided trajectories -> device (kc_cost_evaluate = upload + evaluate)
int kc_cost_upload(kc_dwa *c, const float *paths_x, const float *paths_y, const float *vvx,
                   const float *vvy, const float *vom, size_t n, size_t P) {
  if (!c || (n && (!paths_x || !paths_y)))
    KC_FAIL(KC_ERR_INVALID, "null argument");
  if (P < 2) KC_FAIL(KC_ERR_RANGE, "num_points must be >= 2");
  if (n * P > 0x7FFFFFFFul) KC_FAIL(KC_ERR_RANGE, "n * num_points >= 2^31");
  const bool vel = vvx && vvy && vom;
  KC_TRY(use_device(c));
  hipStream_t s = c->stream;
  KC_HIP(hipStreamSynchronize(s));
  c->drained = true;
  c->update_busy = false;
  c->P = P;
  c->n_roll = n;
  c->external = true;
  c->need_compact = true;
  c->have_vel = vel;
  c->rolled = true;
  c->evaluated = false;
  c->cycle_launched = false;
  c->paths_valid = true;
  c->row_valid = false;
  c->ext_box_valid = false;
  KC_TRY(ensure_cycle_buffers(c, std::max<size_t>(n, 1), P));
  if (n) {
    KC_HIP(hipMemcpyAsync(c->d_px.p, paths_x, n * P * 4, hipMemcpyHostToDevice, s));
    KC_HIP(hipMemcpyAsync(c->d_py.p, paths_y, n * P * 4, hipMemcpyHostToDevice, s));
    if (vel) {
      const size_t nv = n * (P - 1);
      KC_TRY(c->d_vvx.reserve(nv));
      KC_TRY(c->d_vvy.reserve(nv));
      KC_TRY(c->d_vom.reserve(nv));
      KC_HIP(hipMemcpyAsync(c->d_vvx.p, vvx, nv * 4, hipMemcpyHostToDevice, s));
      KC_HIP(hipMemcpyAsync(c->d_vvy.p, vvy, nv * 4, hipMemcpyHostToDevice, s));
      KC_HIP(hipMemcpyAsync(c->d_vom.p, vom, nv * 4, hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(fill_u8_kernel, dim3(blocks_for(n, 256)), dim3(256), 0,
                       s, c->d_flags.p, static_cast<int>(n), uint8_t(1));
    // bounding box of the points: the wavefront-per-sample search lays its near table over it
    c->ext_box_valid = false;
    unsigned int hb[5] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    if (c->near_side != 0) {
      KC_TRY(c->d_bbox.reserve(8));
      KC_HIP(hipMemcpyAsync(c->d_bbox.p, hb, sizeof(hb), hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(bbox_kernel, dim3(512), dim3(256), 0, s, c->d_px.p, c->d_py.p, n * P, c->d_bbox.p);
      KC_HIP(hipMemcpyAsync(hb, c->d_bbox.p, sizeof(hb), hipMemcpyDeviceToHost, s));
    }
    KC_HIP(hipStreamSynchronize(s));  // pageable sources
    if (c->near_side != 0 && hb[4] == 0u && hb[0] <= hb[2] && hb[1] <= hb[3]) {
      auto unkey = [](unsigned int k) {
        const unsigned int b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
        float f;
        std::memcpy(&f, &b, 4);
        return static_cast<double>(f);
      };
      c->ext_box[0] = unkey(hb[0]);
      c->ext_box[1] = unkey(hb[1]);
      c->ext_box[2] = unkey(hb[2]);
      c->ext_box[3] = unkey(hb[3]);
      c->ext_box_valid = true;
    }
  }
  return KC_OK;
}

int kc_cost_evaluate_resident(kc_dwa *c, float *costs_out, kc_result *out) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  if (!c->external || !c->rolled) KC_FAIL(KC_ERR_STATE, "kc_cost_upload has not run");
  KC_TRY(use_device(c));
  const size_t n = c->n_roll;
  if (!c->drained || c->timing.enabled) KC_HIP(hipStreamSynchronize(c->stream));
  c->drained = false;
  c->timing.begin_cycle();
  c->need_compact = true;
  KC_TRY(run_evaluate(c, n, 0));
  c->evaluated = true;
  KC_TRY(fetch(c, out, n));
  if (costs_out && n)
    KC_HIP(hipMemcpy(costs_out, c->d_costs.p, n * 4, hipMemcpyDeviceToHost));
  return KC_OK;
}

int kc_cost_evaluate(kc_dwa *c, const float *paths_x, const float *paths_y,
                     const float *vvx, const float *vvy, const float *vom,
                     size_t n, size_t P, float *costs_out, kc_result *out) {
  KC_TRY(kc_cost_upload(c, paths_x, paths_y, vvx, vvy, vom, n, P));
  return kc_cost_evaluate_resident(c, costs_out, out);
}

namespace {
// shard-local ids in front of global sample `raw` on this context
long long local_bound(const kc_dwa *c, int64_t raw) {
  if (raw <= 0) return 0;
  long long lat_lim;
  if (!c->rows_active || c->external)
    lat_lim = raw;
  else
    lat_lim = std::lower_bound(c->gid.begin(), c->gid.end(), static_cast<int32_t>(std::min<int64_t>(raw, INT32_MAX))) -
              c->gid.begin();
  const long long first = c->external ? 0 : static_cast<long long>(c->shard_first);
  return std::min<long long>(std::max<long long>(lat_lim - first, 0), static_cast<long long>(c->n_roll));
}

// d_xs / d_xr / pinned mirrors of the exchange record for (world, rank, words per rank); the
// words of the OTHER ranks in the send record hold INT64_MAX for good (the minimum passes the
// owner's words through), this rank's are rewritten every cycle
int ensure_xchg(kc_dwa *c, int world, int rank, size_t rw) {
  const size_t len = X_REGIONS + static_cast<size_t>(world) * rw;
  if (c->x_world == world && c->x_rank == rank && c->x_rw == rw && c->d_xs.p) return KC_OK;
  KC_HIP(hipStreamSynchronize(c->stream));
  KC_TRY(c->d_xs.reserve(len));
  KC_TRY(c->d_xr.reserve(len));
  KC_TRY(c->h_xvec.reserve(len));
  KC_TRY(c->h_xrec.reserve(8));
  std::vector<long long> init(len, INT64_MAX);
  init[X_KEY] = KEY_NONE;
  init[X_ERR] = 0;
  for (size_t j = 0; j < rw; ++j) init[X_REGIONS + static_cast<size_t>(rank) * rw + j] = 0;
  KC_HIP(hipMemcpy(c->d_xs.p, init.data(), len * sizeof(long long), hipMemcpyHostToDevice));
  std::memset(c->h_xrec.p, 0, 8 * sizeof(long long));
  c->x_world = world;
  c->x_rank = rank;
  c->x_rw = rw;
  return KC_OK;
}

// the reduced record of a sharded cycle -> result (the same on every rank)
int fetch_xchg(kc_dwa *c, const ShardLayout &L, size_t rw, kc_result *out) {
  const size_t len = X_REGIONS + static_cast<size_t>(L.world) * rw;
  volatile long long *hr = c->h_xrec.p;
  const long long *xv = c->h_xvec.p;
  const auto t0 = std::chrono::steady_clock::now();
  bool synced = false;
  for (long spins = 0;; ++spins) {
    const long long w0 = hr[0], w1 = hr[1], w2 = hr[2], w3 = hr[3], w4 = hr[4];
    if (w2 == c->xseq && w3 == record_check(w0, w1, w2, w4) && w1 == static_cast<long long>(len)) {
      unsigned long long sum = 0ull;
      for (size_t i = 0; i < len; ++i)
        sum += xchg_word_mix(const_cast<const volatile long long *>(xv)[i], static_cast<unsigned>(i));
      if (static_cast<long long>(sum) == w0) break;
    }
    if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
      // a peer may be late: wait for the stream, which ends behind
      // the all-reduce and the hand-off kernel; a record that still does not add up then is an error
      if (synced) KC_FAIL(KC_ERR_HIP, "the reduced exchange record never arrived intact");
      KC_HIP(hipStreamSynchronize(c->stream));
      synced = true;
    }
  }
  c->pub_pending = false;
  c->drained = true;
  c->update_busy = false;
  c->seg_busy = false;
  c->timing.mark("host:wait_result");
  kc_result r{};
  bool failed = false;
  merge_exchange(L, xv, rw, &r, &failed);
  c->last_nadm = popcount_prefix(xv + X_REGIONS + static_cast<size_t>(L.rank) * rw, L.count[static_cast<size_t>(L.rank)]);
  c->row_valid = false;
  c->last_lat = -1;
  if (failed) {
    c->have_last = false;
    if (xv[X_ERR] == -1)
      KC_FAIL(KC_ERR_HIP, "sharded cycle: a rank's device error word is set (every rank fails this cycle)");
    KC_FAIL(KC_ERR_HIP, "sharded cycle: a rank failed before the exchange (every rank fails this cycle)");
  }
  if (r.found) {
    const int64_t loc = L.local_of(L.rank, r.raw_index);
    if (loc >= 0) c->last_lat = static_cast<int64_t>(c->shard_first) + loc;
  }
  c->last = r;
  c->have_last = true;
  if (out) *out = r;
  return KC_OK;
}
}  // namespace

int kc_dwa_result_device(kc_dwa *c, void **dev) {
  if (!c || !dev) KC_FAIL(KC_ERR_INVALID, "null argument");
  *dev = c->d_result.p;
  return KC_OK;
}

int kc_dwa_publish_result(kc_dwa *c) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  if (!c->evaluated) KC_FAIL(KC_ERR_STATE, "nothing evaluated yet");
  if (!c->device_record_valid)
    KC_FAIL(KC_ERR_STATE, "the last cycle was reduced on the host (kc_dwa_cycle): no device-resident record");
  KC_TRY(use_device(c));
  hipLaunchKernelGGL(republish_kernel, dim3(1), dim3(1), 0, c->stream, c->d_result.p,
                     c->h_pub.p, ++c->seq);
  KC_HIP(hipGetLastError());
  c->drained = false;  // (set again by the fetch that sees this record)
  c->row_valid = false;
  c->pub_pending = true;
  return KC_OK;
}

int kc_dwa_allreduce_best(kc_dwa *c, kc_comm *m) {
  if (!c || !m) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (!c->evaluated) KC_FAIL(KC_ERR_STATE, "nothing evaluated yet");
  if (!c->device_record_valid)
    KC_FAIL(KC_ERR_STATE, "the last cycle was reduced on the host (kc_dwa_cycle): use kc_dwa_cycle_sharded, or "
                          "kc_dwa_rollout + kc_dwa_evaluate, for a device-resident record");
  if (kc::comm_device(m) != c->prm.device)
    KC_FAIL(KC_ERR_INVALID, "communicator on device %d, controller on device %d", kc::comm_device(m), c->prm.device);
  if (c->rows_active)
    KC_FAIL(KC_ERR_STATE, "KC_SHARD_ROWS: the device record carries this rank's own numbering; use kc_dwa_cycle_sharded");
  KC_TRY(use_device(c));
  KC_TRY(kc::comm_allreduce_i64(m, c->d_result.p + R_KEY, c->d_result.p + R_KEY, 1, /*sum=*/false, c->stream));
  return kc_dwa_publish_result(c);
}

}  // extern "C"
namespace {
// the part of a sharded cycle behind this rank's words of the send record: the ONE all-reduce, the hand-off of the
// reduced record to the host, the merge.  rc / why: this rank's own failure so far (it has taken part all the same).
int finish_exchange(kc_dwa *c, kc_comm *m, const ShardLayout &L, size_t rw, int rc, const std::string &why, kc_result *out) {
  const size_t len = X_REGIONS + static_cast<size_t>(kc::comm_world(m)) * rw;
  hipStream_t s = c->stream;
  int trc = c->timing.start("all_reduce", s);
  const int rc_x = kc::comm_allreduce_i64(m, c->d_xs.p, c->d_xr.p, len, /*sum=*/false, s);
  if (trc == KC_OK) trc = c->timing.stop(s);
  if (rc_x != KC_OK) {
    if (rc != KC_OK) set_error("%s", why.c_str());
    return rc != KC_OK ? rc : rc_x;
  }
  hipLaunchKernelGGL(xchg_publish_kernel, dim3(1), dim3(256), 0, s, c->d_xr.p, static_cast<int>(len), c->h_xvec.p,
                     c->h_xrec.p, ++c->xseq);
  c->drained = false;
  kc_result r{};
  const int rc_f = fetch_xchg(c, L, rw, &r);
  if (rc != KC_OK) {  // this rank's own failure is the more specific message
    set_error("%s", why.c_str());
    return rc;
  }
  KC_TRY(rc_f);
  if (out) *out = r;
  return KC_OK;
}
}  // namespace
extern "C" {

int kc_dwa_cycle_sharded(kc_dwa *c, kc_comm *m, const kc_state *start, size_t P, kc_result *out) {
  if (!c || !m) KC_FAIL(KC_ERR_INVALID, "null argument");
  const int world = kc::comm_world(m), rank = kc::comm_rank(m);
  // ---- everything that can fail without the peers noticing comes first: a rank that returns
  // here has not entered the collective, and must not be the only one (argument errors are
  // the same on every rank, or a caller bug)
  if (kc::comm_device(m) != c->prm.device)
    KC_FAIL(KC_ERR_INVALID, "communicator on device %d, controller on device %d", kc::comm_device(m), c->prm.device);
  ShardLayout implicit;
  const ShardLayout *L = &c->layout;
  if (c->layout.mode < 0) {
    if (world > 1)
      KC_FAIL(KC_ERR_STATE, "a sharded cycle over %d ranks needs kc_dwa_set_shard_rule (every rank must know every "
                            "rank's share)", world);
    implicit.mode = KC_SHARD_BLOCKS;
    implicit.first = {c->shard_first};
    implicit.count = {c->shard_count};
    implicit.n_total = c->shard_count;
    L = &implicit;
  } else if (c->layout.world != world || c->layout.rank != rank) {
    KC_FAIL(KC_ERR_INVALID, "shard rule is for rank %d of %d, the communicator is rank %d of %d", c->layout.rank,
            c->layout.world, rank, world);
  }
  KC_TRY(use_device(c));
  const size_t rw = std::max<size_t>((L->max_count() + 63) / 64, 1);
  KC_TRY(ensure_xchg(c, world, rank, rw));
  hipStream_t s = c->stream;
  // ---- this rank's cycle.  From here on the rank takes part in the exchange whatever happens:
  // a failure travels in the record's error word and fails the cycle on EVERY rank.
  c->sharded_call = true;
  c->xchg_send = c->d_xs.p;
  c->xchg_rank = rank;
  c->xchg_rw = static_cast<int>(rw);
  c->xchg_packed = false;
  int rc = rollout_impl(c, start, P, true);
  c->sharded_call = false;
  if (rc == KC_OK && !c->cycle_launched) rc = kc_dwa_evaluate(c);
  std::string why;
  if (rc != KC_OK) why = kc_last_error();
  c->pub_pending = false;  // (a sharded cycle hands its record over through the exchange, not h_pub)
  if (rc == KC_OK && c->cycle_launched && c->xchg_packed) {
    // (the single-launch cycle's last workgroup has written this rank's words: cycle_epilogue)
  } else if (rc == KC_OK) {
    PackArgs pa{};
    pa.result = c->d_result.p;
    pa.flags = c->d_flags.p;
    pa.n = static_cast<int>(c->n_roll);
    pa.gid = c->rows_active ? c->d_gid.p : nullptr;
    pa.xs = c->d_xs.p;
    pa.rank = rank;
    pa.rw = static_cast<int>(rw);
    int trc = c->timing.start("xchg_pack_kernel", s);
    hipLaunchKernelGGL(xchg_pack_kernel, dim3(1), dim3(1024), 0, s, pa);
    if (trc == KC_OK) trc = c->timing.stop(s);
  } else {
    (void)hipGetLastError();
    hipLaunchKernelGGL(xchg_fail_kernel, dim3(1), dim3(256), 0, s, c->d_xs.p, rank, static_cast<int>(rw));
  }
  return finish_exchange(c, m, *L, rw, rc, why, out);
}

// The exchange of a cycle whose LAST cost terms were added on the host (custom cost callbacks of a sharded DWA:
// cost_evaluator.cpp:96-100 -- every rank adds the callbacks to the device totals of its own admissible rows, in
// the reference's order, and knows its own best): the same record as kc_dwa_cycle_sharded -- this rank's key
// {cost, GLOBAL raw index} as handed in, the error word, its admissible bitmap from the flags of the cycle it has
// just run (kc_dwa_cycle on its share) -- through the same single all-reduce and the same merge.  status != 0:
// this rank failed somewhere before; it still takes part, and the cycle fails on every rank.
int kc_dwa_exchange_best(kc_dwa *c, kc_comm *m, int status, int found, float cost, int64_t raw_index, kc_result *out) {
  if (!c || !m) KC_FAIL(KC_ERR_INVALID, "null argument");
  const int world = kc::comm_world(m), rank = kc::comm_rank(m);
  if (kc::comm_device(m) != c->prm.device)
    KC_FAIL(KC_ERR_INVALID, "communicator on device %d, controller on device %d", kc::comm_device(m), c->prm.device);
  ShardLayout implicit;
  const ShardLayout *L = &c->layout;
  if (c->layout.mode < 0) {
    if (world > 1) KC_FAIL(KC_ERR_STATE, "an exchange over %d ranks needs kc_dwa_set_shard_rule", world);
    implicit.mode = KC_SHARD_BLOCKS;
    implicit.first = {c->shard_first};
    implicit.count = {c->shard_count};
    implicit.n_total = c->shard_count;
    L = &implicit;
  } else if (c->layout.world != world || c->layout.rank != rank) {
    KC_FAIL(KC_ERR_INVALID, "shard rule is for rank %d of %d, the communicator is rank %d of %d", c->layout.rank,
            c->layout.world, rank, world);
  }
  KC_TRY(use_device(c));
  const size_t rw = std::max<size_t>((L->max_count() + 63) / 64, 1);
  KC_TRY(ensure_xchg(c, world, rank, rw));
  hipStream_t s = c->stream;
  int rc = KC_OK;
  std::string why;
  if (status != 0 || !c->rolled) {
    rc = KC_ERR_STATE;
    why = status != 0 ? "this rank failed before the exchange" : "no cycle has run on this rank's share";
    hipLaunchKernelGGL(xchg_fail_kernel, dim3(1), dim3(256), 0, s, c->d_xs.p, rank, static_cast<int>(rw));
  } else {
    PackArgs pa{};
    pa.result = c->d_result.p;
    pa.flags = c->d_flags.p;
    pa.n = static_cast<int>(c->n_roll);
    pa.gid = nullptr;
    pa.xs = c->d_xs.p;
    pa.rank = rank;
    pa.rw = static_cast<int>(rw);
    pa.host_key = 1;
    pa.key = found ? key_pack(cost, static_cast<uint32_t>(raw_index)) : KEY_NONE;
    hipLaunchKernelGGL(xchg_pack_kernel, dim3(1), dim3(1024), 0, s, pa);
  }
  c->pub_pending = false;
  return finish_exchange(c, m, *L, rw, rc, why, out);
}

int kc_dwa_global_index(kc_dwa *c, kc_comm *m, int64_t raw, int64_t *index_out) {
  if (!c || !m || !index_out) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (!c->rolled) KC_FAIL(KC_ERR_STATE, "kc_dwa_rollout has not run");
  KC_TRY(use_device(c));
  if (c->n_roll == 0 || raw < 0) {
    KC_HIP(hipMemsetAsync(c->d_result.p + R_SCRATCH, 0, sizeof(long long), c->stream));
  } else {
    hipLaunchKernelGGL(count_before_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_flags.p,
                       static_cast<int>(c->n_roll), 0, local_bound(c, raw), c->d_result.p, R_SCRATCH);
  }
  KC_TRY(kc::comm_allreduce_i64(m, c->d_result.p + R_SCRATCH, c->d_result.p + R_SCRATCH, 1, /*sum=*/true, c->stream));
  KC_HIP(hipMemcpyAsync(c->h_result.p + R_SCRATCH, c->d_result.p + R_SCRATCH, sizeof(long long),
                        hipMemcpyDeviceToHost, c->stream));
  KC_HIP(hipStreamSynchronize(c->stream));
  *index_out = raw < 0 ? -1 : c->h_result.p[R_SCRATCH];
  return KC_OK;
}

int kc_dwa_count_admissible_before(kc_dwa *c, int64_t raw, int64_t *count) {
  if (!c || !count) KC_FAIL(KC_ERR_INVALID, "null argument");
  if (!c->rolled) KC_FAIL(KC_ERR_STATE, "kc_dwa_rollout has not run");
  KC_TRY(use_device(c));
  if (c->n_roll == 0 || raw < 0) {
    *count = 0;
    return KC_OK;
  }
  hipLaunchKernelGGL(count_before_kernel, dim3(1), dim3(1024), 0, c->stream,
                     c->d_flags.p, static_cast<int>(c->n_roll), 0, local_bound(c, raw), c->d_result.p, R_SCRATCH);
  KC_HIP(hipMemcpyAsync(c->h_result.p + R_SCRATCH, c->d_result.p + R_SCRATCH,
                        sizeof(long long), hipMemcpyDeviceToHost, c->stream));
  KC_HIP(hipStreamSynchronize(c->stream));
  *count = c->h_result.p[R_SCRATCH];
  return KC_OK;
}

int kc_dwa_timing_enable(kc_dwa *c, int enable) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  c->timing.enabled = enable != 0;
  return KC_OK;
}

int kc_dwa_timing_get(kc_dwa *c, const char **names, float *ms, size_t cap,
                      size_t *count) {
  if (!c) KC_FAIL(KC_ERR_INVALID, "null context");
  KC_TRY(use_device(c));
  return c->timing.get(names, ms, cap, count);
}

}  // extern "C"
