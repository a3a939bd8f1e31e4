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

#ifdef KC_TU_SENSOR  // (a non-template kernel is defined in ONE translation unit: kc_dwa_ctx.h)
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
#endif  // KC_TU_SENSOR

// ---------------------------------------------------------------------------
// Near table of the tracked segment (DcArgs::near): a grid of W x H cells of
// edge g over the box a roll-out can reach; per cell
//   clo | chi << 8 | j* << 16
// j* = a segment point nearest to the cell centre c (distance m), [clo, chi] =
// the hull of the chunks whose capsule comes within m + 2 h of c (h = half a
// cell diagonal + the slack of the kernels' float cell arithmetic).  For a
// point p of the cell the seed is at most m + h away, and every point of a
// chunk outside the range is farther than (m + 2 h) - h from p: the range
// holds p's nearest point and every tie.  Sixteen lanes per cell (chunk k belongs
// to lane k mod 16), the tables read from LDS.
// ---------------------------------------------------------------------------
struct SegNearArgs {
  const float *seg;     // d_seg: rows [5][S], capsule records from seg_cap_offset(S)
  int S, chunk, nch, flat;
  float x0, y0, g;      // origin, cell edge
  float slack;          // added to half a cell diagonal
  int W, H;
  uint32_t *out;
};
constexpr int kSegNearBlock = 512;  // 32 cells per workgroup
constexpr int kSegNearLanes = 8;     // lanes per cell (16 = a DPP row: no faster)
constexpr size_t kSegNearLdsMax = 60 * 1024;  // pair records of the segment in LDS up to here (S < ~3800)

// kLds: the segment's pair records (struct SegPairs) are staged in dynamic LDS -- one global round
// trip per workgroup instead of one per scanned point (a lane's scan of a chunk is a dependent chain:
// 13 us per table from global memory, 3 from LDS)
template <bool kLds>
__global__ __launch_bounds__(kSegNearBlock) void segment_near_kernel(SegNearArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ __align__(16) float l_cap[8 * 64];
  const int S = a.S, nch = a.nch;
  const float *sx = a.seg, *sy = a.seg + S, *szz = a.seg + 3 * S, *sacc = a.seg + 4 * S;
  const float *gcap = a.seg + seg_cap_offset(S);
  for (int j = threadIdx.x; j < 8 * nch; j += kSegNearBlock) l_cap[j] = gcap[j];
  const int npp = seg_pairs_padded(nch, a.chunk);
  float4 *l_xy = reinterpret_cast<float4 *>(smem);
  float4 *l_za = l_xy + npp;
  if (kLds) {
    for (int k = threadIdx.x; k < npp; k += kSegNearBlock) seg_pair_from_rows(sx, sy, szz, sacc, S, k, l_xy[k], l_za[k]);
  }
  __syncthreads();
  constexpr int kL = kSegNearLanes, kPer = 64 / kL;  // chunks per lane (<= 64 chunks)
  const int sub = threadIdx.x & (kL - 1);
  const int cell = blockIdx.x * (kSegNearBlock / kL) + threadIdx.x / kL;
  const int ncell = a.W * a.H;
  const int cc = min(cell, ncell - 1);  // whole groups stay in step (DPP reductions)
  const int ix = cc % a.W, iy = cc / a.W;
  const float x = a.x0 + (static_cast<float>(ix) + 0.5f) * a.g;
  const float y = a.y0 + (static_cast<float>(iy) + 0.5f) * a.g;
  const bool flat = a.flat != 0;
  // squared distances of the cell centre to the two points of pair k (the cost kernels' expression)
  auto pair_of = [&](int k) {
    if (kLds) return pair_d2(SegPairs{l_xy, l_za}, k, x, y, flat);
    return pair_d2(SegRows{sx, sy, szz, sacc, S}, k, x, y, flat);
  };
  // (1) one pass over this lane's chunks: lower bound of the chunk (the kernels' capsule test, solved for
  // the threshold) and the distance to its head (the chord's first end: |q|^2 of the same computation) --
  // the smallest head distance is an upper bound of m
  float lb[kPer];
  uint32_t ub = 0x7F7FFFFFu;
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int k = sub + kL * u;
    lb[u] = __builtin_inff();
    if (k < nch) {
      const Capsule cp = load_capsule(l_cap, k);
      float d2, mag;
      capsule_dist2(cp, x, y, flat, d2, mag);
      // capsule_may_hold(thr): !(d2 > (thr + eps + 4e-7 mag)^2 * 1.0001)
      float v = __builtin_sqrtf(d2) * 0.9999f - cp.eps - 4e-7f * mag;
      if (!(v == v)) v = -__builtin_inff();  // NaN: always a candidate
      lb[u] = v;
      ub = min(ub, __float_as_uint(pair_of((k * a.chunk) >> 1).x));  // NaN / inf bits never win
    }
  }
  ub = group_min_u32<kL>(ub);
  const float uthr = __builtin_sqrtf(__uint_as_float(ub)) * 1.0001f;
  // (2) chunks that may hold something as close as that bound are scanned by the whole group, one
  // pair per lane and step: exact m and a point that attains it
  uint32_t qlo = 0u, qhi = 0u;
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int k = sub + kL * u;
    if (k < nch && lb[u] <= uthr) {
      if (k < 32) qlo |= 1u << k;
      else qhi |= 1u << (k - 32);
    }
  }
  qlo = group_or_u32<kL>(qlo);
  qhi = group_or_u32<kL>(qhi);
  uint32_t mb = ub;
  uint32_t jb = 0xFFFFFFFFu;
  const int hp = a.chunk >> 1;
  for (unsigned long long q = (static_cast<unsigned long long>(qhi) << 32) | qlo; q;) {  // uniform in the group
    const int k = __ffsll(static_cast<long long>(q)) - 1;
    q &= q - 1ull;
    const int k0 = k * hp;
    for (int kk = k0 + sub; kk < k0 + hp && 2 * kk < S; kk += kL) {
      const f32x2 d = pair_of(kk);
      const uint32_t b0 = __float_as_uint(d.x), b1 = __float_as_uint(d.y);
      const uint32_t j0 = static_cast<uint32_t>(2 * kk), j1 = static_cast<uint32_t>(min(2 * kk + 1, S - 1));
      if (b0 < mb || (b0 == mb && j0 < jb)) {
        mb = b0;
        jb = j0;
      }
      if (b1 < mb || (b1 == mb && j1 < jb)) {
        mb = b1;
        jb = j1;
      }
    }
  }
  const uint32_t mg = group_min_u32<kL>(mb);
  const uint32_t jg = group_min_u32<kL>(mb == mg ? jb : 0xFFFFFFFFu);
  // (3) the chunks within m + 2 h
  const float h = a.g * 0.70710679f * 1.0001f + a.slack;
  const float R = __builtin_sqrtf(__uint_as_float(mg)) * 1.0001f + 2.0f * h;  // +inf when nothing is finite
  uint32_t lo = 0xFFu, hi_inv = 0xFFu;  // hi tracked as 255 - k (a minimum again)
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int k = sub + kL * u;
    if (k < nch && lb[u] <= R) {
      lo = min(lo, static_cast<uint32_t>(k));
      hi_inv = min(hi_inv, static_cast<uint32_t>(255 - k));
    }
  }
  lo = group_min_u32<kL>(lo);
  hi_inv = group_min_u32<kL>(hi_inv);
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
