// Host side of kc_dwa_set_scan: the per-beam loops between the caller's ranges and the launch of the sensor
// update -- sensor-frame points (collision_check.h:110-115: x = r cos a, y = r sin a as doubles, rounded to
// float), the obstacle coordinates of the cost path (CostEvaluator::setPointScan, cost_evaluator.h:174-193:
// sensor_tf_body * body_tf_world applied to (x, y, 0)) and the bounding boxes of the chunks of the scan polyline.
// Scalar, they were 17 us of a 24 us call at 4096 beams (5 ns a beam: the conversion loop, an isfinite loop, the
// transform and four min / max chains); here four beams at a time (AVX2, chosen at run time).  Every vector
// operation is the IEEE operation of the scalar form in the same order (-ffp-contract=off: no fused
// multiply-add either way), so the floats are the same bits; the scalar forms are the fallback and the definition.
// Part of kc_dwa_sensor.hip (host code only).
#pragma once

#include <immintrin.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <limits>

#include "kc_seg_tables.h"  // cpu_has_avx2

namespace kc {
namespace scantab {

// rows of the isometry that places an obstacle: out = t + (R0 x + (R1 y + R2 * 0)) per row (hm::Rigid3f::apply with
// pz = 0, Eigen's a + (b + c))
struct Place {
  float r00, r01, z0, t0;  // z0 = R[0][2] * 0.0f
  float r10, r11, z1, t1;
};

// xyz[3 i ..] = {float(r cos), float(r sin), hz}; hx / hy = the placed obstacle; returns false when a range is
// not finite (the outputs are complete either way)
inline bool points_scalar(const double *ranges, const double *cosv, const double *sinv, size_t i0, size_t n, float hz,
                          const Place &p, float *xyz, float *hx, float *hy) {
  bool finite = true;
  for (size_t i = i0; i < n; ++i) {
    const double r = ranges[i];
    finite = finite && std::isfinite(r);
    const float x = static_cast<float>(r * cosv[i]), y = static_cast<float>(r * sinv[i]);
    xyz[3 * i] = x;
    xyz[3 * i + 1] = y;
    xyz[3 * i + 2] = hz;
    hx[i] = p.t0 + (p.r00 * x + (p.r01 * y + p.z0));
    hy[i] = p.t1 + (p.r10 * x + (p.r11 * y + p.z1));
  }
  return finite;
}
__attribute__((target("avx2"))) inline bool points_avx2(const double *ranges, const double *cosv, const double *sinv, size_t n,
                                                        float hz, const Place &p, float *xyz, float *hx, float *hy) {
  const __m128 r00 = _mm_set1_ps(p.r00), r01 = _mm_set1_ps(p.r01), z0 = _mm_set1_ps(p.z0), t0 = _mm_set1_ps(p.t0);
  const __m128 r10 = _mm_set1_ps(p.r10), r11 = _mm_set1_ps(p.r11), z1 = _mm_set1_ps(p.z1), t1 = _mm_set1_ps(p.t1);
  const __m128 vz = _mm_set1_ps(hz);
  __m256d bad = _mm256_setzero_pd();
  size_t i = 0;
  for (; i + 4 <= n; i += 4) {
    const __m256d r = _mm256_loadu_pd(ranges + i);
    const __m256d d = _mm256_sub_pd(r, r);  // 0 for a finite range, NaN otherwise
    bad = _mm256_or_pd(bad, _mm256_cmp_pd(d, d, _CMP_UNORD_Q));
    const __m128 x = _mm256_cvtpd_ps(_mm256_mul_pd(r, _mm256_loadu_pd(cosv + i)));
    const __m128 y = _mm256_cvtpd_ps(_mm256_mul_pd(r, _mm256_loadu_pd(sinv + i)));
    // x0 y0 z x1 | y1 z x2 y2 | z x3 y3 z
    const __m128 xy01 = _mm_unpacklo_ps(x, y);   // x0 y0 x1 y1
    const __m128 xy23 = _mm_unpackhi_ps(x, y);   // x2 y2 x3 y3
    const __m128 v0 = _mm_shuffle_ps(xy01, _mm_shuffle_ps(vz, xy01, _MM_SHUFFLE(2, 2, 0, 0)), _MM_SHUFFLE(2, 0, 1, 0));
    const __m128 v1 = _mm_shuffle_ps(_mm_shuffle_ps(xy01, vz, _MM_SHUFFLE(0, 0, 3, 3)), xy23, _MM_SHUFFLE(1, 0, 2, 0));
    const __m128 v2 = _mm_shuffle_ps(_mm_shuffle_ps(vz, xy23, _MM_SHUFFLE(2, 2, 0, 0)), _mm_shuffle_ps(xy23, vz, _MM_SHUFFLE(0, 0, 3, 3)),
                                     _MM_SHUFFLE(2, 0, 2, 0));
    _mm_storeu_ps(xyz + 3 * i, v0);
    _mm_storeu_ps(xyz + 3 * i + 4, v1);
    _mm_storeu_ps(xyz + 3 * i + 8, v2);
    _mm_storeu_ps(hx + i, _mm_add_ps(t0, _mm_add_ps(_mm_mul_ps(r00, x), _mm_add_ps(_mm_mul_ps(r01, y), z0))));
    _mm_storeu_ps(hy + i, _mm_add_ps(t1, _mm_add_ps(_mm_mul_ps(r10, x), _mm_add_ps(_mm_mul_ps(r11, y), z1))));
  }
  const bool f = _mm256_movemask_pd(bad) == 0;
  return points_scalar(ranges, cosv, sinv, i, n, hz, p, xyz, hx, hy) && f;
}
inline bool points(const double *ranges, const double *cosv, const double *sinv, size_t n, float hz, const Place &p,
                   float *xyz, float *hx, float *hy) {
  return segtab::cpu_has_avx2() ? points_avx2(ranges, cosv, sinv, n, hz, p, xyz, hx, hy)
                                : points_scalar(ranges, cosv, sinv, 0, n, hz, p, xyz, hx, hy);
}

// box of the obstacles [j0, j1) (finite coordinates; an empty range gives the +inf / -inf box)
struct Box {
  float x0, x1, y0, y1;
};
inline Box box_scalar(const float *hx, const float *hy, size_t j0, size_t j1, Box b) {
  for (size_t j = j0; j < j1; ++j) {
    b.x0 = std::min(b.x0, hx[j]);
    b.x1 = std::max(b.x1, hx[j]);
    b.y0 = std::min(b.y0, hy[j]);
    b.y1 = std::max(b.y1, hy[j]);
  }
  return b;
}
inline Box box_empty() {
  const float inf = std::numeric_limits<float>::infinity();
  return Box{inf, -inf, inf, -inf};
}
__attribute__((target("avx2"))) inline Box box_avx2(const float *hx, const float *hy, size_t j0, size_t j1) {
  Box b = box_empty();
  size_t j = j0;
  if (j + 8 <= j1) {
    __m256 x0 = _mm256_loadu_ps(hx + j), x1 = x0, y0 = _mm256_loadu_ps(hy + j), y1 = y0;
    for (j += 8; j + 8 <= j1; j += 8) {
      const __m256 vx = _mm256_loadu_ps(hx + j), vy = _mm256_loadu_ps(hy + j);
      x0 = _mm256_min_ps(x0, vx);
      x1 = _mm256_max_ps(x1, vx);
      y0 = _mm256_min_ps(y0, vy);
      y1 = _mm256_max_ps(y1, vy);
    }
    alignas(32) float a[4][8];
    _mm256_store_ps(a[0], x0);
    _mm256_store_ps(a[1], x1);
    _mm256_store_ps(a[2], y0);
    _mm256_store_ps(a[3], y1);
    for (int k = 0; k < 8; ++k) {
      b.x0 = std::min(b.x0, a[0][k]);
      b.x1 = std::max(b.x1, a[1][k]);
      b.y0 = std::min(b.y0, a[2][k]);
      b.y1 = std::max(b.y1, a[3][k]);
    }
  }
  return box_scalar(hx, hy, j, j1, b);
}
inline Box box_of(const float *hx, const float *hy, size_t j0, size_t j1) {
  return segtab::cpu_has_avx2() ? box_avx2(hx, hy, j0, j1) : box_scalar(hx, hy, j0, j1, box_empty());
}
// ... of the obstacles of [j0, j1) whose coordinates are both finite (a beam without a return: its range is inf or NaN,
// its obstacle never wins `dist < minDist` -- it is not part of any box either)
inline Box box_finite_scalar(const float *hx, const float *hy, size_t j0, size_t j1, Box b) {
  for (size_t j = j0; j < j1; ++j) {
    if (!std::isfinite(hx[j]) || !std::isfinite(hy[j])) continue;
    b.x0 = std::min(b.x0, hx[j]);
    b.x1 = std::max(b.x1, hx[j]);
    b.y0 = std::min(b.y0, hy[j]);
    b.y1 = std::max(b.y1, hy[j]);
  }
  return b;
}
// (eight at a time: a lane whose x or y is not finite -- v - v is NaN then -- takes +inf into the minima and -inf
// into the maxima, the neutral elements; min / max of finite floats do not depend on the order)
__attribute__((target("avx2"))) inline Box box_finite_avx2(const float *hx, const float *hy, size_t j0, size_t j1) {
  const __m256 pinf = _mm256_set1_ps(std::numeric_limits<float>::infinity());
  const __m256 ninf = _mm256_set1_ps(-std::numeric_limits<float>::infinity());
  __m256 x0 = pinf, x1 = ninf, y0 = pinf, y1 = ninf;
  size_t j = j0;
  for (; j + 8 <= j1; j += 8) {
    const __m256 vx = _mm256_loadu_ps(hx + j), vy = _mm256_loadu_ps(hy + j);
    const __m256 dx = _mm256_sub_ps(vx, vx), dy = _mm256_sub_ps(vy, vy);
    const __m256 ok = _mm256_and_ps(_mm256_cmp_ps(dx, dx, _CMP_ORD_Q), _mm256_cmp_ps(dy, dy, _CMP_ORD_Q));
    x0 = _mm256_min_ps(x0, _mm256_blendv_ps(pinf, vx, ok));
    x1 = _mm256_max_ps(x1, _mm256_blendv_ps(ninf, vx, ok));
    y0 = _mm256_min_ps(y0, _mm256_blendv_ps(pinf, vy, ok));
    y1 = _mm256_max_ps(y1, _mm256_blendv_ps(ninf, vy, ok));
  }
  alignas(32) float a[4][8];
  _mm256_store_ps(a[0], x0);
  _mm256_store_ps(a[1], x1);
  _mm256_store_ps(a[2], y0);
  _mm256_store_ps(a[3], y1);
  Box b = box_empty();
  for (int k = 0; k < 8; ++k) {
    b.x0 = std::min(b.x0, a[0][k]);
    b.x1 = std::max(b.x1, a[1][k]);
    b.y0 = std::min(b.y0, a[2][k]);
    b.y1 = std::max(b.y1, a[3][k]);
  }
  return box_finite_scalar(hx, hy, j, j1, b);
}
inline Box box_of_finite(const float *hx, const float *hy, size_t j0, size_t j1) {
  return segtab::cpu_has_avx2() ? box_finite_avx2(hx, hy, j0, j1) : box_finite_scalar(hx, hy, j0, j1, box_empty());
}
inline Box box_join(const Box &a, const Box &b) {
  return Box{std::min(a.x0, b.x0), std::max(a.x1, b.x1), std::min(a.y0, b.y0), std::max(a.y1, b.y1)};
}

}  // namespace scantab
}  // namespace kc
