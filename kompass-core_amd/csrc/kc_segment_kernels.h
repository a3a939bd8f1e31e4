// Tracked-segment tables on the device (second half): the
// interpolated reference path stays resident (kc_dwa_set_path), a cycle only
// moves the window (kc_dwa_set_tracked_window) and this kernel writes what
// kc_dwa_set_tracked_segment builds on the host -- the rows [5][S], the chunk
// capsules [8][nch], the super-chunk spheres [4][nsup] and capsules [8][nsup] -- with the same
// double arithmetic and the same slack.  The bounds only prune the searches of
// the cost kernels, so validity is what matters; they come out identical to the
// host's (maxima and minima do not depend on the order).  Part of kc_dwa.hip.
#pragma once

namespace kc {

struct SegWindowArgs {
  const float *px, *py, *pz, *pacc;  // resident path rows, already offset to the window start
  int S, chunk, nch, nsup;
  float *seg;                        // out: d_seg
};

constexpr int kSegWinBlock = 1024;

__device__ __forceinline__ float seg_round_up(double v) {  // nextafter((float)v, +inf), v >= 0
  const float f = static_cast<float>(v);
  if (!(f < __builtin_inff())) return f;
  return __uint_as_float(__float_as_uint(f) + 1u);
}

__device__ __forceinline__ double wave_max_f64(double v) {
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

__global__ __launch_bounds__(kSegWinBlock) void segment_window_kernel(SegWindowArgs a) {
  const int S = a.S, nch = a.nch, nsup = a.nsup;
  float *h = a.seg;
  for (int j = threadIdx.x; j < S; j += kSegWinBlock) {
    const float zz = a.pz[j];
    h[j] = a.px[j];
    h[S + j] = a.py[j];
    h[2 * S + j] = zz;
    h[3 * S + j] = zz * zz;  // (seg.z - 0)^2 of Path::distance
    h[4 * S + j] = a.pacc[j];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int kWaves = kSegWinBlock / 64;
  float *cap = h + seg_cap_offset(S);
  // capsules: one wavefront per chunk (then per super-chunk of eight chunks), lanes over its points
  float *supc = cap + 8 * nch + 4 * nsup;  // [8][nsup] behind the spheres
  for (int kk = wave; kk < nch + nsup; kk += kWaves) {
    const bool super = kk >= nch;
    const int k = super ? kk - nch : kk;
    const int span = super ? 8 * a.chunk : a.chunk;
    float *out = super ? supc : cap;
    const int j0 = k * span, j1 = min(j0 + span, S);
    bool fin = true;
    for (int j = j0 + lane; j < j1; j += 64)
      fin = fin && isfinite(a.px[j]) && isfinite(a.py[j]) && isfinite(a.pz[j]);
    const bool finite = __ballot(!fin) == 0ull;
    const double A[3] = {a.px[j0], a.py[j0], a.pz[j0]};
    const double B[3] = {a.px[j1 - 1], a.py[j1 - 1], a.pz[j1 - 1]};
    // the chord as the cost kernels see it: float A, float AB, float 1/|AB|^2
    const float ab[3] = {static_cast<float>(B[0] - A[0]), static_cast<float>(B[1] - A[1]),
                         static_cast<float>(B[2] - A[2])};
    const double l2 = static_cast<double>(ab[0]) * ab[0] + static_cast<double>(ab[1]) * ab[1] +
                      static_cast<double>(ab[2]) * ab[2];
    const float inv = (finite && l2 > 0.0 && isfinite(1.0 / l2)) ? static_cast<float>(1.0 / l2) : 0.0f;
    double eps = 0.0, mag = 0.0;
    if (finite) {
      for (int j = j0 + lane; j < j1; j += 64) {
        const double P[3] = {a.px[j], a.py[j], a.pz[j]};
        const double q[3] = {P[0] - A[0], P[1] - A[1], P[2] - A[2]};
        double t = (q[0] * ab[0] + q[1] * ab[1] + q[2] * ab[2]) * static_cast<double>(inv);
        t = fmin(fmax(t, 0.0), 1.0);
        const double e[3] = {q[0] - t * ab[0], q[1] - t * ab[1], q[2] - t * ab[2]};
        eps = fmax(eps, sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]));
        mag = fmax(mag, fabs(P[0]) + fabs(P[1]) + fabs(P[2]));
      }
    }
    eps = wave_max_f64(eps);
    mag = wave_max_f64(mag);
    if (lane == 0) {
      float *rec = out + 8 * k;  // struct Capsule (kc_cost_kernels.h)
      rec[0] = static_cast<float>(A[0]);
      rec[1] = static_cast<float>(A[1]);
      rec[2] = finite ? ab[0] : 0.0f;
      rec[3] = finite ? ab[1] : 0.0f;
      rec[4] = inv;
      rec[5] = finite ? seg_round_up(eps * (1.0 + 1e-6) + 2e-6 * sqrt(l2) + 1e-6 * mag + 1e-30) : __builtin_inff();
      rec[6] = static_cast<float>(A[2]);
      rec[7] = finite ? ab[2] : 0.0f;
    }
  }
  // spheres: one wavefront per super-chunk of eight chunks
  float *sup = cap + 8 * nch;
  for (int s = wave; s < nsup; s += kWaves) {
    const int j0 = s * 8 * a.chunk, j1 = min(j0 + 8 * a.chunk, S);
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    bool fin = true;
    for (int j = j0 + lane; j < j1; j += 64) {
      const double P[3] = {a.px[j], a.py[j], a.pz[j]};
      for (int q = 0; q < 3; ++q) {
        fin = fin && isfinite(P[q]);
        lo[q] = fmin(lo[q], P[q]);
        hi[q] = fmax(hi[q], P[q]);
      }
    }
    const bool finite = __ballot(!fin) == 0ull;
    if (!finite) {  // never skipped
      if (lane == 0) {
        sup[s] = sup[nsup + s] = sup[2 * nsup + s] = 0.0f;
        sup[3 * nsup + s] = __builtin_inff();
      }
      continue;
    }
    float fc[3];
    for (int q = 0; q < 3; ++q) fc[q] = static_cast<float>(0.5 * (wave_min_f64(lo[q]) + wave_max_f64(hi[q])));
    // the radius is taken around the STORED centre and rounded up with slack
    double r = 0.0;
    for (int j = j0 + lane; j < j1; j += 64) {
      const double dx = static_cast<double>(a.px[j]) - fc[0], dy = static_cast<double>(a.py[j]) - fc[1],
                   dz = static_cast<double>(a.pz[j]) - fc[2];
      r = fmax(r, sqrt(dx * dx + dy * dy + dz * dz));
    }
    r = wave_max_f64(r);
    if (lane == 0) {
      const double mag = fabs(static_cast<double>(fc[0])) + fabs(static_cast<double>(fc[1])) +
                         fabs(static_cast<double>(fc[2])) + r;
      sup[s] = fc[0];
      sup[nsup + s] = fc[1];
      sup[2 * nsup + s] = fc[2];
      sup[3 * nsup + s] = seg_round_up(r * (1.0 + 1e-6) + 1e-6 * mag + 1e-30);
    }
  }
}

// ---------------------------------------------------------------------------
// Near table of the tracked segment (DcArgs::near): a grid of W x H cells of
// edge g over the box a roll-out can reach; per cell
//   clo | chi << 8 | j* << 16
// j* = a segment point nearest to the cell centre c (distance m), [clo, chi] =
// the hull of the chunks whose capsule comes within m + 2 h of c (h = half a
// cell diagonal + the slack of the kernels' float cell arithmetic).  For a
// point p of the cell the seed is at most m + h away, and every point of a
// chunk outside the range is farther than (m + 2 h) - h from p: the range
// holds p's nearest point and every tie.  Eight lanes per cell (chunk k belongs
// to lane k mod 8), the tables read from LDS.
// ---------------------------------------------------------------------------
struct SegNearArgs {
  const float *seg;     // d_seg: rows [5][S], capsule records from seg_cap_offset(S)
  int S, chunk, nch, flat;
  float x0, y0, g;      // origin, cell edge
  float slack;          // added to half a cell diagonal
  int W, H;
  uint32_t *out;
};
constexpr int kSegNearBlock = 512;  // 64 cells per workgroup

__global__ __launch_bounds__(kSegNearBlock) void segment_near_kernel(SegNearArgs a) {
  __shared__ __align__(16) float l_cap[8 * 64];
  __shared__ float4 l_head[64];
  const int S = a.S, nch = a.nch;
  const float *sx = a.seg, *sy = a.seg + S, *szz = a.seg + 3 * S;
  const float *gcap = a.seg + seg_cap_offset(S);
  for (int j = threadIdx.x; j < 8 * nch; j += kSegNearBlock) l_cap[j] = gcap[j];
  for (int k = threadIdx.x; k < nch; k += kSegNearBlock) {
    const int j = k * a.chunk;
    l_head[k] = make_float4(sx[j], sy[j], szz[j], 0.0f);
  }
  __syncthreads();
  const int sub = threadIdx.x & 7;
  const int cell = blockIdx.x * (kSegNearBlock / 8) + (threadIdx.x >> 3);
  const int ncell = a.W * a.H;
  const int cc = min(cell, ncell - 1);  // whole groups stay in step (DPP reductions)
  const int ix = cc % a.W, iy = cc / a.W;
  const float x = a.x0 + (static_cast<float>(ix) + 0.5f) * a.g;
  const float y = a.y0 + (static_cast<float>(iy) + 0.5f) * a.g;
  const bool flat = a.flat != 0;
  auto d2_of = [&](float qx, float qy, float qzz) {
    const float dx = qx - x, dy = qy - y;
    return dx * dx + (dy * dy + qzz);
  };
  // (1) upper bound of m: the chunk heads
  uint32_t ub = 0x7F7FFFFFu;
  for (int k = sub; k < nch; k += 8) {
    const float4 q = l_head[k];
    ub = min(ub, __float_as_uint(d2_of(q.x, q.y, q.z)));  // NaN / inf bits never win
  }
  ub = group_min_u32<8>(ub);
  const float uthr = __builtin_sqrtf(__uint_as_float(ub)) * 1.0001f;
  // (2) lower bound of every chunk of this lane (the kernels' capsule test, solved for the
  // threshold); chunks that may hold something as close as the bound are scanned: exact m
  float lb[8];
  uint32_t mb = ub;
  uint32_t jb = 0xFFFFFFFFu;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int k = sub + 8 * u;
    lb[u] = __builtin_inff();
    if (k < nch) {
      const Capsule cp = load_capsule(l_cap, k);
      float d2, mag;
      capsule_dist2(cp, x, y, flat, d2, mag);
      // capsule_may_hold(thr): !(d2 > (thr + eps + 4e-7 mag)^2 * 1.0001)
      float v = __builtin_sqrtf(d2) * 0.9999f - cp.eps - 4e-7f * mag;
      if (!(v == v)) v = -__builtin_inff();  // NaN: always a candidate
      lb[u] = v;
      if (v <= uthr) {
        const int j0 = k * a.chunk, j1 = min(j0 + a.chunk, S);
        for (int j = j0; j < j1; ++j) {
          const uint32_t b = __float_as_uint(d2_of(sx[j], sy[j], szz[j]));
          if (b < mb || (b == mb && static_cast<uint32_t>(j) < jb)) {
            mb = b;
            jb = static_cast<uint32_t>(j);
          }
        }
      }
    }
  }
  const uint32_t mg = group_min_u32<8>(mb);
  const uint32_t jg = group_min_u32<8>(mb == mg ? jb : 0xFFFFFFFFu);
  // (3) the chunks within m + 2 h
  const float h = a.g * 0.70710679f * 1.0001f + a.slack;
  const float R = __builtin_sqrtf(__uint_as_float(mg)) * 1.0001f + 2.0f * h;  // +inf when nothing is finite
  uint32_t lo = 0xFFu, hi_inv = 0xFFu;  // hi tracked as 255 - k (a minimum again)
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int k = sub + 8 * u;
    if (k < nch && lb[u] <= R) {
      lo = min(lo, static_cast<uint32_t>(k));
      hi_inv = min(hi_inv, static_cast<uint32_t>(255 - k));
    }
  }
  lo = group_min_u32<8>(lo);
  hi_inv = group_min_u32<8>(hi_inv);
  if (sub == 0 && cell < ncell) {
    uint32_t clo = lo, chi = 255u - hi_inv;
    if (lo == 0xFFu) {  // cannot happen (the chunk of j* is always in); never trust a table that says "nothing"
      clo = 0u;
      chi = static_cast<uint32_t>(nch - 1);
    }
    const uint32_t seed = jg == 0xFFFFFFFFu ? 0u : jg;
    a.out[cell] = clo | (chi << 8) | (seed << 16);
  }
}

}  // namespace kc
