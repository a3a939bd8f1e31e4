// Sampling-controller hot path on gfx950, translation unit 4 of 4: the exchange of a SHARDED cycle (one all-reduce
// of the exchange record, kc_shard.h) and its building blocks.
#define KC_TU_SHARD
#include "kc_dwa_ctx.h"

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
  c->perm_busy = false;
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

