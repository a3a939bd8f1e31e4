// cos / sin of a double, bit for bit what the host's libm `sincos` returns -- computable on the device.
//
// Why: the reference rolls a sample out with cos(yaw) / sin(yaw) of the host libm (datatypes/path.h:24-30;
// gcc folds the pair into one `sincos` call), and the roll-out is compared bit for bit.  Rounds 1-3 therefore
// had the HOST produce the (omega row x step) table every cycle and handed it to the kernel over the BAR.
// glibc's `sincos` (sysdeps/ieee754/dbl-64/s_sincos.c, the IBM Accurate Mathematical Library as simplified
// in glibc 2.28; the x86-64 build of this entry point has no FMA variant) is plain IEEE double arithmetic
// over a 440-entry table of sin / cos (k / 128) in double-double: every operation below is one correctly
// rounded add / mul (the library is compiled with -ffp-contract=off), in the published order, so the device
// produces the same bits.  The table (kc_sincostab.h) is regenerated from first principles by
// tools/gen_sincostab.py (round-to-nearest high part + low part of the exact value) and checked against
// the one in the installed libm; `kc_trig_selfcheck` compares this restatement with the installed `sincos`
// on a fixed argument set when the library is loaded, and the host table path remains the fallback
// when they disagree (another libm) or an argument lies outside the range below.
//
// Range: |x| < 105414350 (table + Cody-Waite reduction); beyond, `kc_sincos_exact` reports failure and
// the caller uses the host (glibc goes on to a 1200-bit reduction there).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define KC_TRIG_HD __host__ __device__ __forceinline__
#else
#define KC_TRIG_HD inline
#endif

namespace kc {
namespace trig {

struct Bits {
  static KC_TRIG_HD uint64_t of(double x) {
    uint64_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = static_cast<uint64_t>(__double_as_longlong(x));
#else
    std::memcpy(&u, &x, 8);
#endif
    return u;
  }
};

KC_TRIG_HD double absd(double x) { return __builtin_fabs(x); }

// The table as do_sin / do_cos index it -- tab[4 i + c], c = 0..3: sn, ssn, cs, ccs of entry i -- held as four
// rows of 110 (an LDS copy whose lanes look up different entries: see rollout_collide_kernel)
struct TabRows {
  const double *p;
  KC_TRIG_HD double operator[](int k) const { return p[(k & 3) * 110 + (k >> 2)]; }
};

// do_sin / do_cos of s_sin.c: x + dx is the argument (|x + dx| < 0.86), tab = sincostab
template <class Tab>
KC_TRIG_HD double do_cos(double x, double dx, Tab tab) {
  constexpr double big = 0x1.8p45;
  constexpr double sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7;
  constexpr double cs2 = 0.5, cs4 = -0x1.5555555555535p-5, cs6 = 0x1.6c16bedd9e239p-10;
  if (x < 0) dx = -dx;
  const double u = big + absd(x);
  x = absd(x) - (u - big) + dx;
  const double xx = x * x;
  const double s = x + x * xx * (sn3 + xx * sn5);
  const double c = xx * (cs2 + xx * (cs4 + xx * cs6));
  const int k = static_cast<int>(static_cast<uint32_t>(Bits::of(u))) * 4;
  const double sn = tab[k], ssn = tab[k + 1], cs = tab[k + 2], ccs = tab[k + 3];
  const double cor = (ccs - s * ssn - cs * c) - sn * s;
  return cs + cor;
}

template <class Tab>
KC_TRIG_HD double do_sin(double x, double dx, Tab tab) {
  constexpr double big = 0x1.8p45;
  constexpr double sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7;
  constexpr double cs2 = 0.5, cs4 = -0x1.5555555555535p-5, cs6 = 0x1.6c16bedd9e239p-10;
  constexpr double s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ecep-7, s3 = -0x1.a01a019db08b8p-13,
                   s4 = 0x1.71de27b9a7ed9p-19, s5 = -0x1.addffc2fcdf59p-26;
  const double xold = x;
  if (absd(x) < 0.126) {
    const double xx = x * x;
    const double p = ((((s5 * xx + s4) * xx + s3) * xx + s2) * xx) + s1;
    const double t = (p * x - 0.5 * dx) * xx + dx;
    return x + t;
  }
  if (x <= 0) dx = -dx;
  const double u = big + absd(x);
  x = absd(x) - (u - big);
  const double xx = x * x;
  const double s = x + (dx + x * xx * (sn3 + xx * sn5));
  const double c = x * dx + xx * (cs2 + xx * (cs4 + xx * cs6));
  const int k = static_cast<int>(static_cast<uint32_t>(Bits::of(u))) * 4;
  const double sn = tab[k], ssn = tab[k + 1], cs = tab[k + 2], ccs = tab[k + 3];
  const double cor = (ssn + s * ccs - sn * c) + cs * s;
  return __builtin_copysign(sn + cor, xold);
}

// s_sincos.c: __sincos.  Returns false outside the table + Cody-Waite range (NaN, inf, |x| >= 105414350).
// The library's three argument ranges differ only in how they reach (a, da) with |a + da| < 0.86 and in which
// of do_sin / do_cos gives which result with which sign; they are restated as ONE evaluation of the pair behind
// selects (a wavefront whose lanes fall into different ranges runs the polynomials once, not three times) --
// in three steps, so that two lanes can share an argument: sincos_reduce, do_sin | do_cos, sincos_finish.
struct Reduced {
  double a, da;  // the reduced argument
  int mode;      // 0: tiny, 1: direct, 2: pi/2 - |x|, 4 + n: Cody-Waite with quadrant n
};
KC_TRIG_HD bool sincos_reduce(double x, Reduced *r) {
  const uint32_t k = static_cast<uint32_t>(Bits::of(x) >> 32) & 0x7fffffffu;
  if (k >= 0x419921FBu) return false;
  const bool direct = k < 0x3feb6000u;   // |x| < 0.855469: do_sin (x, 0), do_cos (x, 0)
  const bool halfpi = k < 0x400368fdu;   // |x| < 2.426265: through pi/2 - |x|
  // pi/2 - |x|
  constexpr double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
  const double y2 = hp0 - absd(x);
  const double a2 = y2 + hp1;
  const double da2 = (y2 - a2) + hp1;
  // reduce_sincos
  constexpr double hpinv = 0x1.45f306dc9c883p-1, toint = 0x1.8p52;
  constexpr double mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde973c000000p-27;
  constexpr double pp3 = -0x1.cb3b398000000p-55, pp4 = -0x1.d747f23e32ed7p-83;
  const double t = x * hpinv + toint;
  const double xn = t - toint;
  const double y = (x - xn * mp1) - xn * mp2;
  const int n = static_cast<int>(static_cast<uint32_t>(Bits::of(t))) & 3;
  double t1 = xn * pp3;
  const double t2 = y - t1;
  double db = (y - t2) - t1;
  t1 = xn * pp4;
  const double b = t2 - t1;
  db += (t2 - b) - t1;
  r->a = direct ? x : halfpi ? a2 : b;
  r->da = direct ? 0.0 : halfpi ? da2 : db;
  r->mode = k < 0x3e400000u ? 0 : direct ? 1 : halfpi ? 2 : 4 + n;  // (|x| < 2^-27: sin = x, cos = 1)
  return true;
}
// sv = do_sin (a, da), cv = do_cos (a, da)
KC_TRIG_HD void sincos_finish(double x, int mode, double sv, double cv, double *sinx, double *cosx) {
  double s, c;
  if (mode == 1) {
    s = sv;
    c = cv;
  } else if (mode == 2) {
    s = __builtin_copysign(cv, x);
    c = sv;
  } else {  // do_sincos (a, da, n) and (a, da, n + 1)
    const int n = mode & 3;
    s = (n & 1) ? cv : sv;
    if (n & 2) s = -s;
    const int m = n + 1;
    c = (m & 1) ? cv : sv;
    if (m & 2) c = -c;
  }
  if (mode == 0) {
    s = x;
    c = 1.0;
  }
  *sinx = s;
  *cosx = c;
}
template <class Tab>
KC_TRIG_HD bool sincos_exact(double x, double *sinx, double *cosx, Tab tab) {
  Reduced r;
  if (!sincos_reduce(x, &r)) return false;
  sincos_finish(x, r.mode, do_sin(r.a, r.da, tab), do_cos(r.a, r.da, tab), sinx, cosx);
  return true;
}

}  // namespace trig
}  // namespace kc
