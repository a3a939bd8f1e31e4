// Host side of kc_dwa_set_tracked_segment: the inner loops of the search tables over the tracked segment
// (chunk / super-chunk capsules, bounding spheres, segment length).  The call sits on the host's critical
// path in front of every cycle launch (DESIGN: host chain of a reference cycle) and was 5 us of scalar double
// arithmetic for a 501-point segment: ~2 000 point visits in four passes.  Here the same expressions four
// points at a time (AVX2, chosen at run time; the scalar forms are the fallback and the definition).
// The tables are BOUNDS (rounded up, with slack: kc_dwa.hip) -- the vector forms evaluate the same double
// expressions per point and reduce with max / min, which do not depend on the order.
// Part of kc_dwa.hip (host code only).
#pragma once

#include <immintrin.h>

#include <cfloat>
#include <cmath>
#include <cstddef>

namespace kc {
namespace segtab {

struct Span {
  const float *x, *y, *z;  // rows of the segment
};

inline bool cpu_has_avx2() {
  static const bool v = __builtin_cpu_supports("avx2");
  return v;
}

// ---- every coordinate of [j0, j1) finite? ---------------------------------------------------------------
inline bool finite_span_scalar(const Span &p, size_t j0, size_t j1) {
  bool f = true;
  for (size_t j = j0; j < j1; ++j) f = f && std::isfinite(p.x[j]) && std::isfinite(p.y[j]) && std::isfinite(p.z[j]);
  return f;
}
__attribute__((target("avx2"))) inline bool finite_span_avx2(const Span &p, size_t j0, size_t j1) {
  // v - v == 0 exactly for finite v, NaN otherwise
  __m256 bad = _mm256_setzero_ps();
  size_t j = j0;
  for (; j + 8 <= j1; j += 8) {
    const __m256 vx = _mm256_loadu_ps(p.x + j), vy = _mm256_loadu_ps(p.y + j), vz = _mm256_loadu_ps(p.z + j);
    const __m256 d = _mm256_add_ps(_mm256_add_ps(_mm256_sub_ps(vx, vx), _mm256_sub_ps(vy, vy)), _mm256_sub_ps(vz, vz));
    bad = _mm256_or_ps(bad, _mm256_cmp_ps(d, d, _CMP_UNORD_Q));
  }
  bool f = _mm256_movemask_ps(bad) == 0;
  return f && finite_span_scalar(p, j, j1);
}
inline bool finite_span(const Span &p, size_t j0, size_t j1) {
  return cpu_has_avx2() ? finite_span_avx2(p, j0, j1) : finite_span_scalar(p, j0, j1);
}

// ---- capsule of [j0, j1) around the chord A + t ab: largest squared deviation of a point from the chord and
// largest |x| + |y| + |z| (all points finite) ---------------------------------------------------------------
inline void capsule_span_scalar(const Span &p, size_t j0, size_t j1, const double A[3], const float ab[3], float inv,
                                double &eps2, double &mag) {
  for (size_t j = j0; j < j1; ++j) {
    const double P[3] = {p.x[j], p.y[j], p.z[j]};
    const double q[3] = {P[0] - A[0], P[1] - A[1], P[2] - A[2]};
    double t = (q[0] * ab[0] + q[1] * ab[1] + q[2] * ab[2]) * static_cast<double>(inv);
    t = std::min(std::max(t, 0.0), 1.0);
    const double e[3] = {q[0] - t * ab[0], q[1] - t * ab[1], q[2] - t * ab[2]};
    eps2 = std::max(eps2, e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    mag = std::max(mag, std::fabs(P[0]) + std::fabs(P[1]) + std::fabs(P[2]));
  }
}
__attribute__((target("avx2"))) inline void capsule_span_avx2(const Span &p, size_t j0, size_t j1, const double A[3],
                                                              const float ab[3], float inv, double &eps2, double &mag) {
  const __m256d ax = _mm256_set1_pd(A[0]), ay = _mm256_set1_pd(A[1]), az = _mm256_set1_pd(A[2]);
  const __m256d bx = _mm256_set1_pd(ab[0]), by = _mm256_set1_pd(ab[1]), bz = _mm256_set1_pd(ab[2]);
  const __m256d vinv = _mm256_set1_pd(static_cast<double>(inv));
  const __m256d zero = _mm256_setzero_pd(), one = _mm256_set1_pd(1.0);
  const __m256d absmask = _mm256_castsi256_pd(_mm256_set1_epi64x(0x7FFFFFFFFFFFFFFFll));
  __m256d veps = _mm256_setzero_pd(), vmag = _mm256_setzero_pd();
  size_t j = j0;
  for (; j + 4 <= j1; j += 4) {
    const __m256d px = _mm256_cvtps_pd(_mm_loadu_ps(p.x + j)), py = _mm256_cvtps_pd(_mm_loadu_ps(p.y + j)),
                  pz = _mm256_cvtps_pd(_mm_loadu_ps(p.z + j));
    const __m256d qx = _mm256_sub_pd(px, ax), qy = _mm256_sub_pd(py, ay), qz = _mm256_sub_pd(pz, az);
    __m256d t = _mm256_mul_pd(_mm256_add_pd(_mm256_add_pd(_mm256_mul_pd(qx, bx), _mm256_mul_pd(qy, by)), _mm256_mul_pd(qz, bz)), vinv);
    t = _mm256_min_pd(_mm256_max_pd(t, zero), one);
    const __m256d ex = _mm256_sub_pd(qx, _mm256_mul_pd(t, bx)), ey = _mm256_sub_pd(qy, _mm256_mul_pd(t, by)),
                  ez = _mm256_sub_pd(qz, _mm256_mul_pd(t, bz));
    veps = _mm256_max_pd(veps, _mm256_add_pd(_mm256_add_pd(_mm256_mul_pd(ex, ex), _mm256_mul_pd(ey, ey)), _mm256_mul_pd(ez, ez)));
    vmag = _mm256_max_pd(vmag, _mm256_add_pd(_mm256_add_pd(_mm256_and_pd(px, absmask), _mm256_and_pd(py, absmask)), _mm256_and_pd(pz, absmask)));
  }
  alignas(32) double e4[4], m4[4];
  _mm256_store_pd(e4, veps);
  _mm256_store_pd(m4, vmag);
  for (int k = 0; k < 4; ++k) {
    eps2 = std::max(eps2, e4[k]);
    mag = std::max(mag, m4[k]);
  }
  capsule_span_scalar(p, j, j1, A, ab, inv, eps2, mag);
}
inline void capsule_span(const Span &p, size_t j0, size_t j1, const double A[3], const float ab[3], float inv, double &eps2,
                         double &mag) {
  if (cpu_has_avx2()) capsule_span_avx2(p, j0, j1, A, ab, inv, eps2, mag);
  else capsule_span_scalar(p, j0, j1, A, ab, inv, eps2, mag);
}

// ---- bounding box of [j0, j1) (all points finite) ---------------------------------------------------------
inline void box_span_scalar(const Span &p, size_t j0, size_t j1, double lo[3], double hi[3]) {
  for (size_t j = j0; j < j1; ++j) {
    const double P[3] = {p.x[j], p.y[j], p.z[j]};
    for (int q = 0; q < 3; ++q) {
      lo[q] = std::min(lo[q], P[q]);
      hi[q] = std::max(hi[q], P[q]);
    }
  }
}
__attribute__((target("avx2"))) inline void box_span_avx2(const Span &p, size_t j0, size_t j1, double lo[3], double hi[3]) {
  const float *rows[3] = {p.x, p.y, p.z};
  for (int q = 0; q < 3; ++q) {
    __m256 vlo = _mm256_set1_ps(FLT_MAX), vhi = _mm256_set1_ps(-FLT_MAX);
    size_t j = j0;
    for (; j + 8 <= j1; j += 8) {
      const __m256 v = _mm256_loadu_ps(rows[q] + j);
      vlo = _mm256_min_ps(vlo, v);
      vhi = _mm256_max_ps(vhi, v);
    }
    alignas(32) float l8[8], h8[8];
    _mm256_store_ps(l8, vlo);
    _mm256_store_ps(h8, vhi);
    for (int k = 0; k < 8; ++k) {  // (float min / max of floats: the same values the double form compares)
      lo[q] = std::min(lo[q], static_cast<double>(l8[k]));
      hi[q] = std::max(hi[q], static_cast<double>(h8[k]));
    }
    for (; j < j1; ++j) {
      lo[q] = std::min(lo[q], static_cast<double>(rows[q][j]));
      hi[q] = std::max(hi[q], static_cast<double>(rows[q][j]));
    }
  }
}
inline void box_span(const Span &p, size_t j0, size_t j1, double lo[3], double hi[3]) {
  if (cpu_has_avx2()) box_span_avx2(p, j0, j1, lo, hi);
  else box_span_scalar(p, j0, j1, lo, hi);
}

// ---- largest squared distance of a point of [j0, j1) from the (float) centre fc ------------------------------
inline double radius2_span_scalar(const Span &p, size_t j0, size_t j1, const float fc[3]) {
  double r = 0.0;
  for (size_t j = j0; j < j1; ++j) {
    const double dx = static_cast<double>(p.x[j]) - fc[0], dy = static_cast<double>(p.y[j]) - fc[1],
                 dz = static_cast<double>(p.z[j]) - fc[2];
    r = std::max(r, dx * dx + dy * dy + dz * dz);
  }
  return r;
}
__attribute__((target("avx2"))) inline double radius2_span_avx2(const Span &p, size_t j0, size_t j1, const float fc[3]) {
  const __m256d cx = _mm256_set1_pd(fc[0]), cy = _mm256_set1_pd(fc[1]), cz = _mm256_set1_pd(fc[2]);
  __m256d vr = _mm256_setzero_pd();
  size_t j = j0;
  for (; j + 4 <= j1; j += 4) {
    const __m256d dx = _mm256_sub_pd(_mm256_cvtps_pd(_mm_loadu_ps(p.x + j)), cx),
                  dy = _mm256_sub_pd(_mm256_cvtps_pd(_mm_loadu_ps(p.y + j)), cy),
                  dz = _mm256_sub_pd(_mm256_cvtps_pd(_mm_loadu_ps(p.z + j)), cz);
    vr = _mm256_max_pd(vr, _mm256_add_pd(_mm256_add_pd(_mm256_mul_pd(dx, dx), _mm256_mul_pd(dy, dy)), _mm256_mul_pd(dz, dz)));
  }
  alignas(32) double r4[4];
  _mm256_store_pd(r4, vr);
  double r = std::max(std::max(r4[0], r4[1]), std::max(r4[2], r4[3]));
  return std::max(r, radius2_span_scalar(p, j, j1, fc));
}
inline double radius2_span(const Span &p, size_t j0, size_t j1, const float fc[3]) {
  return cpu_has_avx2() ? radius2_span_avx2(p, j0, j1, fc) : radius2_span_scalar(p, j0, j1, fc);
}

// ---- View::totalSegmentLength (path.h:85-91): len += sqrt(dx^2 + (dy^2 + dz^2)) in float, point after point.
// The ORDER of the additions is the reference's (a float sum); the roots do not depend on it: they are formed
// eight at a time (IEEE mul / add / sqrt: the same bits as the scalar form), the sum walks them in order.
inline float length_scalar(const Span &p, size_t S) {
  float len = 0.0f;
  for (size_t j = 0; j + 1 < S; ++j) {
    const float dx = p.x[j] - p.x[j + 1], dy = p.y[j] - p.y[j + 1], dz = p.z[j] - p.z[j + 1];
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    len += std::sqrt(xx + (yy + zz));
  }
  return len;
}
__attribute__((target("avx2"))) inline float length_avx2(const Span &p, size_t S) {
  float len = 0.0f;
  size_t j = 0;
  for (; j + 9 <= S; j += 8) {  // steps j .. j + 7 need points j .. j + 8
    const __m256 dx = _mm256_sub_ps(_mm256_loadu_ps(p.x + j), _mm256_loadu_ps(p.x + j + 1));
    const __m256 dy = _mm256_sub_ps(_mm256_loadu_ps(p.y + j), _mm256_loadu_ps(p.y + j + 1));
    const __m256 dz = _mm256_sub_ps(_mm256_loadu_ps(p.z + j), _mm256_loadu_ps(p.z + j + 1));
    const __m256 r = _mm256_sqrt_ps(_mm256_add_ps(_mm256_mul_ps(dx, dx), _mm256_add_ps(_mm256_mul_ps(dy, dy), _mm256_mul_ps(dz, dz))));
    alignas(32) float r8[8];
    _mm256_store_ps(r8, r);
    for (int k = 0; k < 8; ++k) len += r8[k];
  }
  for (; j + 1 < S; ++j) {
    const float dx = p.x[j] - p.x[j + 1], dy = p.y[j] - p.y[j + 1], dz = p.z[j] - p.z[j + 1];
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    len += std::sqrt(xx + (yy + zz));
  }
  return len;
}
inline float length(const Span &p, size_t S) { return cpu_has_avx2() ? length_avx2(p, S) : length_scalar(p, S); }

}  // namespace segtab
}  // namespace kc
