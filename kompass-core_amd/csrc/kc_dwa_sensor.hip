// Sampling-controller hot path on gfx950, translation unit 2 of 4: per-cycle INPUTS -- sensor data (points, scans,
// mapper grids: voxel bitmap + dilations + obstacle buckets), the tracked segment and its search tables, the near
// tables.  Kernels: kc_sensor_kernels.h, kc_onear_kernels.h, kc_segment_kernels.h, dilate_kernel.
#define KC_TU_SENSOR
#include "kc_dwa_ctx.h"

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

// CollDev::cover of this robot: circles along a box at least twice as long as wide -- as many as keep the inner
// circles (radius = the short half extent) inside the box, eight at most
int box_cover(const kc_dwa *c) {
  if (c->prm.shape != KC_BOX || !c->box_cover_on) return 0;
  const double a = c->prm.dims[0] / 2.0, b = c->prm.dims[1] / 2.0;
  const double A = std::max(a, b), B = std::min(a, b);
  if (!(B > 0.0) || !(A >= 2.0 * B) || !std::isfinite(A / B)) return 0;
  // (the shell between the two circles of the single look-up is A - B thick: below 4.5 voxels the look-ups of every
  // pose cost what the exact tests of the shell's poses did -- tools/geometry_sweep.py mid boxes: 1.2 x 0.4 m at 10 cm
  // voxels 32.9 / 34.4 us on / off in clutter, 17.5 / 15.6 where everything collides)
  if ((A - B) / c->res < 4.5) return 0;
  const int nc = static_cast<int>(std::min(8.0, std::floor(A / B)));
  return nc > 1 ? (nc | (b > a ? 0x100 : 0)) : 0;
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
  // a long box: the outer mask for one of the circles laid along it (CollDev::cover)
  if (const int nc = box_cover(c) & 0xFF; nc > 1) {
    const double A = std::max(c->prm.dims[0], c->prm.dims[1]) / 2.0, B = std::min(c->prm.dims[0], c->prm.dims[1]) / 2.0;
    g.rho_out = std::sqrt((A / nc) * (A / nc) + B * B) * (1.0 + 1e-9) / c->res;
  }
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
  c->dil_cover = c->have_dil ? box_cover(c) : 0;
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
    // (the gaps are a function of the voxel LAYER: a handful of distinct values among thousands of voxels -- collected
    // by a scan against the values seen so far, sorted afterwards; sorting every voxel's gap was most of a sphere's
    // sensor update)
    std::vector<double> lut;
    for (double g : c->vox_ddz) {
      bool seen = false;
      for (double v : lut) seen = seen || v == g;
      if (!seen) {
        lut.push_back(g);
        if (lut.size() > 255) break;
      }
    }
    std::sort(lut.begin(), lut.end());
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
    KC_TRY(c->d_cells.reserve(8));
    c->h_cells.p[0] = c->h_cells.p[1] = 0;
    KC_HIP(hipMemcpyAsync(c->d_cells.p, c->h_cells.p, 2 * sizeof(int),
                          hipMemcpyHostToDevice, c->stream));
    KC_TRY(c->d_bobs.reserve(2));
    KC_TRY(c->h_skip.reserve(4));
    KC_TRY(c->d_skip.reserve(16));
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
  KC_TRY(c->d_cells.reserve(ncell + 4));   // (+ the tail of the cycle kernel's 16-byte copies)
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
  KC_TRY(c->d_skip.reserve(ncell + 16));
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
int ensure_host_lists(kc_dwa *c) {
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
      // (a value that is not finite -- a beam without a return, a NaN of a depth image -- leaves the bounds alone)
      const __m512 d = _mm512_sub_ps(v, v);
      const __mmask16 fin = _mm512_cmp_ps_mask(d, d, _CMP_ORD_Q);
      mn[q] = _mm512_mask_min_ps(mn[q], fin, mn[q], v);
      mx[q] = _mm512_mask_max_ps(mx[q], fin, mx[q], v);
      bad |= static_cast<__mmask16>(~fin);
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
  if (!c->device_sensor || !c->trig_direct || n == 0 || n > kSensorDeviceMax)
    return KC_OK;
  if (c->prm.shape == KC_SPHERE && (c->sensor_two_launch || n > kSensorFusedMax || !c->sensor_fused_ok)) return KC_OK;
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
        const v4 dv = v - v;
        const v4i fin = (dv == zero);  // (not finite: the element leaves the bounds alone)
        mn[q] = __builtin_ia32_minps(mn[q], (v4)(((v4i)v & fin) | ((v4i)big & ~fin)));
        mx[q] = __builtin_ia32_maxps(mx[q], (v4)(((v4i)v & fin) | ((v4i)(-big) & ~fin)));
        ok &= fin;
      }
    }
    // (Round 4: values that are not finite no longer send the list to the scalar loop below -- a LaserScan with beams
    // without a return, a depth image with holes: 22 us at 4096 beams.  The bounds are those of the finite VALUES: a
    // point with one coordinate missing still widens the other two -- a looser box, which only sizes the tables; the
    // kernels drop the point as before.)
    (void)ok;
    {
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
        if (!std::isfinite(v)) continue;
        lo[i % 3] = std::min(lo[i % 3], v);
        hi[i % 3] = std::max(hi[i % 3], v);
      }
      raw_copied = true;
      tail_ok = lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2];  // (a finite value on every axis)
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
  KC_TRY(ensure_sincostab(c));
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
  if (!c->device_sensor || !c->trig_direct || n == 0 || n > kSensorDeviceMax)
    return KC_OK;
  // bitmap: keys of the bounds (points beyond the 16-level octree are dropped
  // by add_voxel anyway)
  auto key = [&](float v) {
    const double f = std::floor(c->inv_res * static_cast<double>(v));
    return static_cast<int>(std::min(std::max(f, -32768.0), 32767.0));
  };
  // Spheres (round 4; the host build took 120 us of a 176 us cycle): a voxel's z gap to the sphere's centre is a function
  // of its LAYER, so add_voxel's rule is evaluated here once per layer the cloud's z range can hold -- accepted layers,
  // their gaps in ascending order (the LUT of the exact tests), the code of every layer -- and the one-launch build keeps
  // the smallest code of every voxel column (sensor_band_body).  The gap bound of the dilated masks is the largest gap of
  // those layers: at least the largest gap present, so "certain hits" stay certain.  More than 32 layers, more than
  // 32 k points, bands beyond LDS: the host build.
  const bool sphere = c->prm.shape == KC_SPHERE;
  int sph_kz0 = 0, sph_nkz = 0;
  unsigned char sph_code[36] = {0};
  std::vector<double> sph_lut;
  if (sphere) {
    if (c->sensor_two_launch || n > kSensorFusedMax || !c->sensor_fused_ok) return KC_OK;
    // (only the layers that can touch the sphere: a 3-D cloud spans two metres of height, forty layers of 5 cm -- the
    // rule below rejects a layer whose gap exceeds the radius, i.e. every layer outside [zc - r, zc + r] and a layer of
    // slack; the kernel rejects whatever lies outside the table)
    const double zc = -static_cast<double>(c->frame.t[2]);
    const int k0 = std::max(key(lo[2]), static_cast<int>(std::max(std::floor((zc - c->radius) * c->inv_res) - 1.0, -32768.0)));
    const int k1 = std::min(key(hi[2]), static_cast<int>(std::min(std::floor((zc + c->radius) * c->inv_res) + 1.0, 32767.0)));
    if (k1 - k0 + 1 > 36) return KC_OK;
    double gap[36];
    double gmax = -1.0;
    for (int kz = k0; kz <= k1; ++kz) {  // add_voxel, the sphere branch
      const double zlo = static_cast<double>(kz) * c->res, zhi = static_cast<double>(kz + 1) * c->res;
      double ddz = 0.0;
      if (zlo - zc > ddz) ddz = zlo - zc;
      if (zc - zhi > ddz) ddz = zc - zhi;
      gap[kz - k0] = ddz > c->radius ? -1.0 : ddz;
      if (gap[kz - k0] >= 0.0) {
        gmax = std::max(gmax, ddz);
        bool seen = false;
        for (double v : sph_lut) seen = seen || v == ddz;
        if (!seen) sph_lut.push_back(ddz);
      }
    }
    std::sort(sph_lut.begin(), sph_lut.end());
    if (sph_lut.size() > 32) return KC_OK;
    for (int kz = k0; kz <= k1; ++kz)
      if (gap[kz - k0] >= 0.0)
        sph_code[kz - k0] = static_cast<unsigned char>(std::lower_bound(sph_lut.begin(), sph_lut.end(), gap[kz - k0]) - sph_lut.begin() + 1);
    sph_kz0 = k0;
    sph_nkz = std::max(k1 - k0 + 1, 0);
    c->sphere_ddz_max = gmax;  // (dil_geom: -1 = no layer of the cloud can touch the sphere: no masks, no voxels)
  }
  bool fits = false;
  KC_TRY(bitmap_extent(c, key(lo[0]), key(lo[1]), key(hi[0]), key(hi[1]), &fits));
  const size_t nwords = fits ? static_cast<size_t>(c->gH) * c->gwpr : 0;
  if (!fits) {
    c->have_gbits = false;
      return KC_OK;
  }
  if (sphere) {
    // (decided before anything is written: a sphere either takes the one-launch build or the host's)
    const DilGeom dg0 = dil_geom(c);
    const int dilR0 = c->have_dil ? dg0.R : -1;
    int nb0 = std::min(64, c->gH), rows0 = (c->gH + nb0 - 1) / nb0;
    auto bytes0 = [&] { return (3 * static_cast<size_t>(rows0) + 2 * static_cast<size_t>(std::max(dilR0, 0)) + 32 * static_cast<size_t>(rows0)) * c->gwpr * 4; };
    while (bytes0() > kSensorFusedLds && rows0 > 1) rows0 = (rows0 + 1) / 2;
    if (bytes0() > kSensorFusedLds || (c->gH + rows0 - 1) / rows0 > 1024) return KC_OK;
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
  KC_TRY(c->d_cells.reserve(ncell + 4));   // (+ the tail of the cycle kernel's 16-byte copies)
  KC_TRY(c->d_skip.reserve(ncell + 16));
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
  a.sphere = sphere ? 1 : 0;
  a.kz0 = sph_kz0;
  a.nkz = sph_nkz;
  std::memcpy(a.zcode, sph_code, sizeof(a.zcode));
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
  auto band_bytes = [&] {
    return (3 * static_cast<size_t>(band_rows) + 2 * static_cast<size_t>(std::max(dilR, 0)) + (sphere ? 32 * static_cast<size_t>(band_rows) : 0)) * c->gwpr * 4;
  };
  while (band_bytes() > kSensorFusedLds && band_rows > 1) {
    band_rows = (band_rows + 1) / 2;
  }
  nb = (c->gH + band_rows - 1) / band_rows;
  const bool fused = !big_only && c->sensor_fused_ok && n <= (sphere ? kSensorFusedMax : kSensorFusedPays) &&
                     band_bytes() <= kSensorFusedLds && nb <= 1024;
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
    c->gz_valid = false;
    if (sphere) {
      const size_t gcells = static_cast<size_t>(c->gwpr) * 32 * c->gH;
      KC_TRY(c->d_gz.reserve(gcells));
      KC_TRY(c->h_zlut.reserve(256));
      KC_TRY(c->d_zlut.reserve(256));
      for (size_t k = 0; k < sph_lut.size(); ++k) c->h_zlut.p[k] = sph_lut[k];
      if (!sph_lut.empty()) KC_TRY(upload_table(c, c->d_zlut.p, c->h_zlut.p, sph_lut.size() * sizeof(double)));
      bar_flush(c);
      c->sphere_layers = sph_lut.size();
      c->gz_valid = !sph_lut.empty();
      f.gz = c->d_gz.p;
    }
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
    if (sphere) KC_FAIL(KC_ERR_STATE, "sphere: the one-launch sensor build was decided above");  // (cannot happen)
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
int ensure_near_table(kc_dwa *c, double x, double y, double margin) {
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

int ensure_onear(kc_dwa *c, double x, double y, bool build) {
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
  if (!build) return KC_OK;  // (a cycle with a handful of survivors takes the table when it is there -- it rode in the sensor
                             // launch -- and does not pay a launch for it)
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
    c->scan_cos.resize(n);
    c->scan_sin.resize(n);
    for (size_t i = 0; i < n; ++i) {
      c->scan_cos[i] = std::cos(angles[i]);
      c->scan_sin[i] = std::sin(angles[i]);
    }
  }
  // sensor-frame points: voxels at z = hz (collision_check.h:110-115; a
  // non-finite range gives non-finite coordinates, which add_voxel drops),
  // obstacles from the same x, y at z = 0 (cost path: no filter) -- one pass over the beams, four at a time
  // (kc_scan_tables.h): points, the obstacles in beam order (CostEvaluator::setPointScan, cost_evaluator.h:174-193:
  // sensor_tf_body * body_tf_world applied to (r cos a, r sin a, 0)), "every range finite"
  c->scan_xyz.resize(3 * n);
  c->h_oscan.resize(2 * n + 256 + 1024);
  float *hx = c->h_oscan.data(), *hy = hx + n, *box = hy + n, *sub = box + 256;
  const scantab::Place place{c->obs_tf.R[0][0], c->obs_tf.R[0][1], c->obs_tf.R[0][2] * 0.0f, c->obs_tf.t[0],
                             c->obs_tf.R[1][0], c->obs_tf.R[1][1], c->obs_tf.R[1][2] * 0.0f, c->obs_tf.t[1]};
  const bool finite = scantab::points(ranges, c->scan_cos.data(), c->scan_sin.data(), n, hz, place, c->scan_xyz.data(), hx, hy);
  c->have_sensor = true;
  c->max_obs_dist = max_range / 3.0f;  // cost_evaluator.h:179
  ++c->sensor_version;
  c->oscan_valid = false;
  c->onear_ok = false;
  if (c->obs_near_opt && n >= 64 && n <= 65536) {
    // the boxes of the chunks of the scan polyline, for the near table of the scan.  Beams without a return (inf / NaN
    // ranges: what a real scanner reports beyond its range) have obstacles that never win a minimum: they stay in the
    // list -- the indices are the scan's -- and out of the boxes.  (Until round 4 one such beam left the whole scan to
    // the bucket search: 65-235 us of cycle kernel in a room against 25-45, tools/room_sweep.py ... obs_near=0.)
    {
      auto box_of = [&](size_t j0, size_t j1) {
        return finite ? scantab::box_of(hx, hy, j0, j1) : scantab::box_of_finite(hx, hy, j0, j1);
      };
      const int cs = static_cast<int>((n + 63) / 64);
      const int nch = static_cast<int>((n + cs - 1) / cs);
      // chunks of 32 obstacles and more also get the boxes of their four quarters (wave_sample_total prunes by them)
      const int scs = cs >= 32 ? (cs + 3) / 4 : 0;
      auto put = [](float *out, int stride, int k, const scantab::Box &b) {
        out[k] = b.x0;
        out[stride + k] = b.x1;
        out[2 * stride + k] = b.y0;
        out[3 * stride + k] = b.y1;
      };
      for (int k = 0; k < 64; ++k) {
        const size_t j0 = std::min(n, static_cast<size_t>(k) * cs), j1 = std::min(n, static_cast<size_t>(k + 1) * cs);
        if (scs > 0) {
          scantab::Box whole = scantab::box_empty();
          for (int q = 0; q < 4; ++q) {
            const size_t q0 = std::min(j1, j0 + static_cast<size_t>(q) * scs), q1 = std::min(j1, q0 + static_cast<size_t>(scs));
            const scantab::Box b = (k < nch) ? box_of(q0, q1) : scantab::box_empty();
            put(sub, 256, 4 * k + q, b);
            whole = scantab::box_join(whole, b);
          }
          put(box, 64, k, whole);
        } else {
          put(box, 64, k, (k < nch) ? box_of(j0, j1) : scantab::box_empty());
        }
      }
      KC_TRY(c->d_oscan.reserve(2 * n + 256 + 1024));
      KC_TRY(upload_table(c, c->d_oscan.p, hx, (2 * n + 256 + (scs > 0 ? 1024 : 0)) * sizeof(float)));
      if (!c->trig_direct) {
        KC_HIP(hipStreamSynchronize(c->stream));  // (pageable source)
      } else {
        bar_flush(c);
      }
      c->oscan_scs = scs;
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
  // large maps: the host path, on the (small) list instead of the grid
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
  c->perm_busy = false;
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
    c->perm_busy = false;
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



void sensor_kernel_limits(kc_dwa *c) {
  c->sensor_fused_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(sensor_fused_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(kSensorFusedLds)) == hipSuccess;
  if (!c->sensor_fused_ok) (void)hipGetLastError();
}
