// CPU check of csrc/kc_scan_tables.h: the four-beams-at-a-time forms of kc_dwa_set_scan's host loops give the
// bits of the scalar forms (which restate collision_check.h:110-115 and cost_evaluator.h:174-193), for every
// list length around the vector width, with non-finite ranges, zero ranges and signed zeros.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "kc_scan_tables.h"

using namespace kc;

static bool same_bits(const std::vector<float> &a, const std::vector<float> &b) {
  return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0;
}

int main() {
  if (!segtab::cpu_has_avx2()) {
    std::printf("no AVX2 on this CPU: nothing to compare, 0 bad\n");
    return 0;
  }
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> ur(0.0, 12.0), ua(-3.2, 3.2);
  int bad = 0, cases = 0;
  for (int rep = 0; rep < 400; ++rep) {
    const size_t n = rep < 40 ? static_cast<size_t>(rep) : 1 + g() % 5000;
    std::vector<double> r(n), c(n), s(n);
    for (size_t i = 0; i < n; ++i) {
      const double a = ua(g);
      r[i] = ur(g);
      c[i] = std::cos(a);
      s[i] = std::sin(a);
    }
    if (n > 3 && rep % 3 == 0) r[g() % n] = 0.0;
    if (n > 3 && rep % 5 == 0) r[g() % n] = -0.0;
    const bool poison = n > 0 && rep % 7 == 0;
    if (poison) r[g() % n] = (rep % 14 == 0) ? std::numeric_limits<double>::infinity() : std::nan("");
    scantab::Place p{};
    const float yaw = static_cast<float>(ua(g));
    p.r00 = std::cos(yaw); p.r01 = -std::sin(yaw); p.r10 = std::sin(yaw); p.r11 = std::cos(yaw);
    p.z0 = (rep % 2 ? 0.0f : -0.0f); p.z1 = 0.0f * p.r10;
    p.t0 = static_cast<float>(ua(g)); p.t1 = static_cast<float>(ua(g));
    const float hz = -0.125f;
    std::vector<float> xa(3 * n + 4, 7.f), xb(3 * n + 4, 7.f), hxa(n + 4, 7.f), hxb(n + 4, 7.f), hya(n + 4, 7.f), hyb(n + 4, 7.f);
    const bool fa = scantab::points_avx2(r.data(), c.data(), s.data(), n, hz, p, xa.data(), hxa.data(), hya.data());
    const bool fb = scantab::points_scalar(r.data(), c.data(), s.data(), 0, n, hz, p, xb.data(), hxb.data(), hyb.data());
    ++cases;
    if (fa != fb || fa == poison || !same_bits(xa, xb) || !same_bits(hxa, hxb) || !same_bits(hya, hyb)) {
      ++bad;
      std::printf("points differ at n = %zu (finite %d / %d)\n", n, fa, fb);
    }
    if (n > 0) {  // the boxes over the finite obstacles only (beams without a return), poisoned lists included
      std::vector<float> px(hxb), py(hyb);
      for (size_t i = 0; i < n; ++i)
        if (g() % 9 == 0) (g() % 2 ? px : py)[i] = (g() % 2) ? std::numeric_limits<float>::infinity() : std::nanf("");
      for (int k = 0; k < 20; ++k) {
        const size_t j0 = g() % n, j1 = j0 + g() % (n - j0 + 1);
        const scantab::Box a = scantab::box_finite_avx2(px.data(), py.data(), j0, j1);
        const scantab::Box b = scantab::box_finite_scalar(px.data(), py.data(), j0, j1, scantab::box_empty());
        ++cases;
        if (std::memcmp(&a, &b, sizeof(a)) != 0) {
          ++bad;
          std::printf("finite box differs at [%zu, %zu)\n", j0, j1);
        }
      }
    }
    if (!poison && n > 0) {
      for (int k = 0; k < 20; ++k) {
        const size_t j0 = g() % n, j1 = j0 + g() % (n - j0 + 1);
        const scantab::Box a = scantab::box_avx2(hxb.data(), hyb.data(), j0, j1);
        const scantab::Box b = scantab::box_scalar(hxb.data(), hyb.data(), j0, j1, scantab::box_empty());
        ++cases;
        if (std::memcmp(&a, &b, sizeof(a)) != 0) {
          ++bad;
          std::printf("box differs at [%zu, %zu)\n", j0, j1);
        }
      }
    }
  }
  std::printf("%d cases, %d bad\n", cases, bad);
  return bad ? 1 : 0;
}
