// Sampling-controller hot path on gfx950, translation unit 1 of 4: the context (create / destroy, options), the
// sample lattice (window, explicit lists, shares by rule), the trig self-check.  C ABI: include/kompass_hip.h.
#define KC_TU_CONTEXT
#include "kc_dwa_ctx.h"

// kc_trig_exact.h against the installed libm, once per process: a fixed argument set over every branch of the
// algorithm (tiny, Taylor, table, pi/2 - x, Cody-Waite with every quadrant) and yaw chains as the roll-out forms
// them.  Any difference (another libm: a build with FMA contraction, a different algorithm) switches the device
// trig off for the process -- the host table path is exact by construction.
static const double kc_sincostab_host[440] = {KC_SINCOSTAB_VALUES};
int ensure_sincostab(kc_dwa *c) {
  if (c->d_sincostab.p) return KC_OK;
  KC_TRY(c->d_sincostab.reserve(440));
  KC_HIP(hipMemcpyAsync(c->d_sincostab.p, kc_sincostab_host, sizeof(kc_sincostab_host), hipMemcpyHostToDevice, c->stream));
  KC_HIP(hipStreamSynchronize(c->stream));
  return KC_OK;
}
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

// The tables of the active pattern change places with those of a kept one (signature + size), or -- a pattern not
// seen lately -- with the least recently used slot, whose buffers the caller then overwrites.  true: the device
// holds the tables of (sig, n) now.  Pointers only: nothing is written, nothing queued reads these tables between
// cycles (the sensor kernels do not).
static bool swap_in_pattern(kc_dwa *c, uint64_t sig, size_t n) {
  // (40 bytes of device memory a sample and pattern: 128 MB of them at most, 16 to 256 patterns)
  const size_t kKept = std::min<size_t>(256, std::max<size_t>(16, (size_t(128) << 20) / (40 * std::max<size_t>(n, 1))));
  auto exchange = [&](kc_dwa::PatternTables &t) {
    std::swap(t.vidx, c->d_vidx);
    std::swap(t.row, c->d_row);
    std::swap(t.perm, c->d_perm);
    std::swap(t.prow, c->d_prow);
    std::swap(t.pvi, c->d_pvi);
    std::swap(t.cperm, c->d_cperm);
    std::swap(t.cprow, c->d_cprow);
    std::swap(t.cpvi, c->d_cpvi);
    std::swap(t.h_perm, c->h_perm);
    std::swap(t.h_dealt, c->h_dealt);
    std::swap(t.rows, c->uploaded_rows);
    std::swap(t.perm_valid, c->perm_valid);
    std::swap(t.perm_plain_dev, c->perm_plain_dev);
    std::swap(t.perm_dealt_dev, c->perm_dealt_dev);
    std::swap(t.perm_first, c->perm_first);
    std::swap(t.perm_count, c->perm_count);
    std::swap(t.perm_cs, c->perm_cs);
    std::swap(t.sig, c->up_sig);
    std::swap(t.n, c->up_n);
  };
  ++c->pattern_clock;
  const bool active_keeps = c->up_sig != 0 && c->up_n > 0;  // (lists without a signature are not kept)
  for (auto &t : c->patterns)
    if (t.sig == sig && t.n == n) {
      if (!active_keeps) {  // nothing worth keeping comes back: the slot's tables move in, the slot empties
        exchange(t);
        t.sig = 0;
        t.n = 0;
        t.perm_valid = t.perm_plain_dev = t.perm_dealt_dev = false;
      } else {
        exchange(t);
      }
      t.stamp = c->pattern_clock;
      c->up_ix.clear();
      c->up_iy.clear();
      ++c->pattern_hits;
      return true;
    }
  if (!active_keeps) return false;
  // keep the active tables: an empty slot, a new slot, or the one unused for longest
  kc_dwa::PatternTables *slot = nullptr;
  for (auto &t : c->patterns)
    if (t.sig == 0) slot = &t;
  if (!slot && c->patterns.size() < kKept) {
    c->patterns.emplace_back();
    slot = &c->patterns.back();
  }
  if (!slot) {
    slot = &c->patterns[0];
    for (auto &t : c->patterns)
      if (t.stamp < slot->stamp) slot = &t;
  }
  exchange(*slot);
  slot->stamp = c->pattern_clock;
  // (what came back is a stale or empty set of buffers: the caller rebuilds into it)
  c->up_sig = 0;
  c->up_n = 0;
  c->perm_valid = c->perm_plain_dev = c->perm_dealt_dev = false;
  return false;
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
  const bool same_active = n == c->up_n && ((lat.signature != 0 && lat.signature == c->up_sig) ||
                                            (lat.signature == 0 && c->up_sig == 0 && c->uploaded_rows == lat.row &&
                                             c->up_ix == lat.ix && c->up_iy == lat.iy));
  ++c->lat_version;
  c->shard_first = 0;
  c->shard_count = n;
  bool same = same_active;
  if (!same && n > 0 && lat.signature != 0) same = swap_in_pattern(c, lat.signature, n);
  if (!same) c->perm_valid = c->perm_plain_dev = c->perm_dealt_dev = false;  // (the orders also belong to one shard)
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
      c->perm_busy = false;
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



// ===========================================================================
// C ABI
// ===========================================================================

// shard-local sample ids ordered by trig row (stable): consecutive samples of a
// fused workgroup then share one or two rows of the table.  The single-launch
// cycle gets a second order, dealt from the first: survivors of the collision
// gate cluster (a few adjacent omega rows, the low speeds of each), and a
// workgroup costs its own survivors, so a cluster of any shape has to land on
// many workgroups instead of a few (see below).
int build_perm(kc_dwa *c, bool want_dealt) {
  const size_t n = c->shard_count, first = c->shard_first;
  const bool host_ok = c->perm_valid && c->perm_first == first && c->perm_count == n && c->perm_cs == c->cycle_samples;
  if (!host_ok) c->perm_plain_dev = c->perm_dealt_dev = false;
  c->perm_valid = true;
  c->perm_first = first;
  c->perm_count = n;
  if (n == 0) return KC_OK;
  std::vector<int32_t> &dealt = c->h_dealt;
  if (!host_ok) {
    c->h_perm.resize(n);
    const int32_t *row = c->lat.row.data() + first;
    {
      // the shard's samples ordered by trig row, ties in list order: a counting sort (rows are small integers; this
      // runs whenever the window's pattern is new, which a robot crossing v = 0 makes a frequent event)
      int32_t rmax = 0;
      for (size_t i = 0; i < n; ++i) rmax = std::max(rmax, row[i]);
      std::vector<int32_t> &start = c->perm_scratch;
      start.assign(static_cast<size_t>(rmax) + 2, 0);
      for (size_t i = 0; i < n; ++i) ++start[static_cast<size_t>(row[i]) + 1];
      for (size_t r = 1; r < start.size(); ++r) start[r] += start[r - 1];
      for (size_t i = 0; i < n; ++i) c->h_perm[static_cast<size_t>(start[static_cast<size_t>(row[i])]++)] = static_cast<int32_t>(i);
    }
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
  }  // (host orders)
  c->perm_cs = c->cycle_samples;
  if (want_dealt ? c->perm_dealt_dev : c->perm_plain_dev) return KC_OK;
  std::vector<int32_t> &prow = c->perm_prow;
  std::vector<uint32_t> &pvi = c->perm_pvi;
  prow.resize(n);
  pvi.resize(n);
  if (c->trig_direct && c->perm_busy) {  // (stores through the BAR below: no queued roll-out may still read the old orders;
    KC_HIP(hipStreamSynchronize(c->stream));  // a sensor update in flight does not touch them and is not waited for)
    c->drained = true;
    c->perm_busy = false;
    c->update_busy = false;
  }
  // only the order the launch at hand walks goes to the device (the single-launch cycle: the dealt one): 12 bytes
  // a sample over the BAR, on the host's critical path whenever the window's pattern is new
  for (int pass = want_dealt ? 1 : 0; pass < (want_dealt ? 2 : 1); ++pass) {
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
    if (c->trig_direct) {
      // straight into device memory over the BAR (upload_samples does the same with the index tables): the orders are
      // read by roll-out kernels only, and the cycle that asks for them has seen the previous cycle's record
      KC_TRY(upload_table(c, dperm.p, order.data(), n * sizeof(int32_t)));
      KC_TRY(upload_table(c, drow.p, prow.data(), n * sizeof(int32_t)));
      KC_TRY(upload_table(c, dvi.p, pvi.data(), n * sizeof(uint32_t)));
    } else {
      KC_HIP(hipMemcpyAsync(dperm.p, order.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
      KC_HIP(hipMemcpyAsync(drow.p, prow.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
      KC_HIP(hipMemcpyAsync(dvi.p, pvi.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
      KC_HIP(hipStreamSynchronize(c->stream));  // pageable sources
    }
  }
  if (c->trig_direct) bar_flush(c);
  (want_dealt ? c->perm_dealt_dev : c->perm_plain_dev) = true;
  if (!host_ok) ++c->pattern_builds;
  return KC_OK;
}

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
  cycle_kernel_limits(c);   // (dynamic LDS beyond 64 KB: the unit that owns the kernels asks for it)
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
    sensor_kernel_limits(c);
  }
  if (const char *e = std::getenv("KC_DEBUG_STAMPS")) c->debug_stamps = e[0] == '1';
  (void)launch_init_result(c);
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
    const char *nm[22] = {"start", "phase A done", "trig entries in LDS", "increments in LDS", "recurrence done", "poses checked", "flags out", "increments done", "costs done", "epilogue done", "pass 0 searched", "pass 0 barrier", "pass 0 total", "poses classified / ticket", "last: reduced", "", "A: small tables in LDS", "A: bulk loads issued", "A: trig entries formed", "A: window stored", "A: table loads issued", ""};
    std::fprintf(stderr, "[kc stamps] roll-out kernel, us since first block start (avg / max):\n");
    for (int k = 0; k < 22; ++k) {
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
    {
      // when the workgroups end (slot 9) and how long their cost phase takes (slot 8 - slot 6), in 1 us bins
      int end_hist[64] = {0}, cost_hist[64] = {0};
      for (int b = 0; b < 512; ++b) {
        if (!h[b * 32] || !h[b * 32 + 9] || !h[b * 32 + 8] || !h[b * 32 + 6]) continue;
        const int e = static_cast<int>((h[b * 32 + 9] - t0) / 100), d = static_cast<int>((h[b * 32 + 8] - h[b * 32 + 6]) / 100);
        ++end_hist[std::min(std::max(e, 0), 63)];
        ++cost_hist[std::min(std::max(d, 0), 63)];
      }
      {  // the cost phase by the number of survivors of the workgroup
        double sum[34] = {0}, mx[34] = {0};
        int cnt[34] = {0};
        for (int b = 0; b < 512; ++b) {
          if (!h[b * 32] || !h[b * 32 + 8] || !h[b * 32 + 6] || !h[b * 32 + 21]) continue;
          const int R = std::min<int>(static_cast<int>(h[b * 32 + 21]) - 1, 33);
          const double d = (h[b * 32 + 8] - h[b * 32 + 6]) / 100.0;
          sum[R] += d; mx[R] = std::max(mx[R], d); ++cnt[R];
        }
        std::fprintf(stderr, "  cost phase by survivors of the workgroup, survivors:workgroups avg/max us:");
        for (int R = 0; R < 34; ++R) if (cnt[R]) std::fprintf(stderr, " %d:%d %.1f/%.1f", R, cnt[R], sum[R] / cnt[R], mx[R]);
        std::fprintf(stderr, "\n");
      }
      std::fprintf(stderr, "  workgroups by end of epilogue (us):");
      for (int i = 0; i < 64; ++i) if (end_hist[i]) std::fprintf(stderr, " %d:%d", i, end_hist[i]);
      std::fprintf(stderr, "\n  workgroups by duration of the cost phase (us):");
      for (int i = 0; i < 64; ++i) if (cost_hist[i]) std::fprintf(stderr, " %d:%d", i, cost_hist[i]);
      std::fprintf(stderr, "\n");
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
  for (auto &t : c->patterns) {
    t.vidx.release();
    t.pvi.release();
    t.cpvi.release();
    t.row.release();
    t.perm.release();
    t.prow.release();
    t.cperm.release();
    t.cprow.release();
  }
  c->patterns.clear();
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
  c->perm_busy = false;
  c->update_busy = false;
  const std::string n(name);
  const bool on = v != 0.0;
  if (n == "fused_cycle") {
    if (!(v == 0.0 || v == 1.0 || v == 2.0)) KC_FAIL(KC_ERR_RANGE, "fused_cycle: 0 off, 1 when it pays, 2 whenever it fits");
    c->cycle_fused = on;
    c->cycle_forced = v == 2.0;
  }
  else if (n == "write_paths") c->write_paths = on;
  else if (n == "team_max") c->team_max = std::min(4, std::max(0, static_cast<int>(v)));
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
  } else if (n == "box_cover") {
    c->box_cover_on = on;  // (from the next sensor update on: the masks are dilated there)
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
  else if (n == "team_max") *v = c->team_max;
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
  else if (n == "box_cover") *v = c->box_cover_on;
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
  else if (n == "pattern_hits") *v = static_cast<double>(c->pattern_hits);      // read-only: window patterns found on the device
  else if (n == "pattern_builds") *v = static_cast<double>(c->pattern_builds);  // read-only: walking orders built (build_perm)
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
  KC_TRY(launch_trig_table(tj, nullptr));
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


