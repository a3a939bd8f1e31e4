// Sampling-controller hot path on gfx950, translation unit 3 of 4: the CYCLE -- roll-out + collision gate, cost
// stage, result record (single launch or three kernels), caller-provided batches, results and rows.
// Kernels: kc_rollout_kernels.h, kc_cost_kernels.h, kc_cycle_dev.h, kc_tilt_dev.h.
#define KC_TU_CYCLE
#include "kc_dwa_ctx.h"


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
    cd.cover = c->have_dil ? c->dil_cover : 0;
    cd.gkx0 = c->gkx0;
    cd.gky0 = c->gky0;
    cd.gH = c->gH;
    cd.gwpr = c->gwpr;
  }
  cd.wpr = (cd.W + 31) / 32;
  cd.wpr_magic = cd.wpr > 1 ? 0xFFFFFFFFu / static_cast<uint32_t>(cd.wpr) + 1u : 0u;
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
    dt.oscs = c->oscan_scs;
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
    // (with a scan's near table the wavefront kernels keep the scan block there instead: x | y | chunk boxes)
    if (dt.onear) lds_obs = std::max(lds_obs, static_cast<size_t>(scan_block_floats(dt.on, dt.oscs)) * sizeof(float));
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
  c->perm_busy = false;
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
        c->perm_busy = false;
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

// LDS bytes of the cost tables of the cycle tail (cycle_tabs, kc_cycle_dev.h)
size_t cycle_table_bytes(const CostArgs &ca) {
  size_t b = 0;
  if (ca.use_seg)
    b += 32 * static_cast<size_t>(seg_pairs_padded(ca.nch, ca.seg_chunk)) +
         4 * ((8 * static_cast<size_t>(ca.nch) + 12 * static_cast<size_t>(ca.nsup) + 3) & ~size_t(3));
  if (ca.use_obs) {
    const size_t ncell = static_cast<size_t>(ca.b.W) * ca.b.H;
    b += 4 * ((ncell + 1 + 3) & ~size_t(3)) + ((ncell + 15) & ~size_t(15));  // (16-byte rows: copied as vectors)
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
int rollout_impl(kc_dwa *c, const kc_state *start, size_t P, bool want_cycle, bool trig_ready) {
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
  a.nvx = static_cast<int>(c->lat.vx_values.size());
  a.nvy = static_cast<int>(c->lat.vy_values.size());
  a.vidx = c->d_vidx.p;
  a.row = c->d_row.p;
  a.trig = c->d_trig.p;
  a.trig_dev = (dev_trig && !table_ahead) ? 1 : 0;
  if (dev_trig) KC_TRY(ensure_sincostab(c));
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
  // (Rounds 2-3 sent small lattices with many survivors to the stand-alone cost kernels -- "they spread the survivors
  // over all CUs".  Round 4's lattice sweep, 110 .. 2025 samples, half or all of them admissible: the single launch is
  // 3 us ahead everywhere -- the second launch costs more than the spreading gains.  One resident round of workgroups is
  // the only condition left.)
  const bool cyc_wave = cyc_G <= static_cast<unsigned>(c->num_cus);
  const bool sphere_ok = c->prm.shape != KC_SPHERE || (c->have_gbits && c->gz_valid);  // (fused path)
  bool cycle = want_cycle && c->cycle_fused && sphere_ok && n <= 1024u * kCompactMaxPer &&
               (c->cycle_forced || cyc_wave);
  if (cycle) {
    // workgroups with more than a handful of survivors search wavefront-per-sample: through the
    // near table when the last cycle had that many
    c->near_ok = false;
    c->near_wanted = c->last_nadm < 0 || c->last_nadm > 2ll * cyc_G;
    c->onear_ok = false;
    if (c->near_wanted) {
      KC_TRY(ensure_near_table(c, start->x, start->y));
      KC_TRY(ensure_onear(c, start->x, start->y));
    } else {
      // few survivors: the scan's near table only if it is there already (the teams' last wavefronts use it: a tiny room
      // puts the whole scan into the union rectangle of a sample, 35 us for 44 survivors without the table)
      KC_TRY(ensure_onear(c, start->x, start->y, false));
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
    if (!c->perm_valid || c->perm_first != c->shard_first || c->perm_count != c->shard_count || c->perm_cs != cs ||
        !(cycle ? c->perm_dealt_dev : c->perm_plain_dev))
      KC_TRY(build_perm(c, cycle));
    a.perm = cycle ? c->d_cperm.p : c->d_perm.p;
    a.prow = cycle ? c->d_cprow.p : c->d_prow.p;
    a.pvi = cycle ? c->d_cpvi.p : c->d_pvi.p;
    c->perm_busy = true;  // (until the host has seen this cycle's record)
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
      tail.team_max = c->team_max;
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
  c->perm_busy = false;
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

int launch_init_result(kc_dwa *c) {
  hipLaunchKernelGGL(init_result_kernel, dim3(1), dim3(1), 0, c->stream, c->d_result.p);
  return KC_OK;
}

// (kc_trig_table: the kernels' cos / sin rows for a list of omega values -- tests/test_device_trig.py)
int launch_trig_table(const TrigJob &tj, hipStream_t s) {
  hipLaunchKernelGGL(trig_table_kernel, dim3(tj.nblk), dim3(kTrigBlock), 0, s, tj);
  return KC_OK;
}


// opt in to more than 64 KB of dynamic LDS for the fused roll-out / cycle kernels and the cost kernels (gfx950: 160 KB)
void cycle_kernel_limits(kc_dwa *c) {
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
}
