// CPU check of hm::build_window_lattice (csrc/kc_hostmath.h): a lattice object that is REUSED from window to window
// (value tables rewritten, index tables kept while the pattern stays, resized in place and refilled by blocks when it
// changes) equals one built from scratch, sample by sample, over a random walk of the velocity that changes the
// pattern often -- all three control types.
#include <cstdio>
#include <cstring>
#include <random>

#include "kc_hostmath.h"

using namespace kc;

int main() {
  kc_limits lim{};
  lim.vx_max = 1.0; lim.vx_acc = 2.0; lim.vx_dec = 2.0;
  lim.vy_max = 1.0; lim.vy_acc = 2.0; lim.vy_dec = 2.0;
  lim.omega_max_angle = 2.0; lim.omega_max = 2.0; lim.omega_acc = 3.0; lim.omega_dec = 3.0;
  std::mt19937_64 g(11);
  std::normal_distribution<double> nv(0, 0.03), no(0, 0.08);
  int bad = 0, changes = 0, cases = 0;
  for (int ctr = 0; ctr < 3; ++ctr) {
    hm::VelocityLattice reused;
    double vx = 0.2, vy = 0.0, om = 0.0;
    uint64_t last_sig = 0;
    for (int i = 0; i < 1500; ++i) {
      vx = std::min(1.0, std::max(-0.3, vx + nv(g)));
      vy = ctr == 2 ? std::min(0.5, std::max(-0.5, vy + nv(g))) : 0.0;
      om = std::min(1.5, std::max(-1.5, om + no(g)));
      const int ml = ctr == 2 ? 21 : 31, ma = ctr == 2 ? 9 : 31;
      hm::build_window_lattice(ctr, lim, vx, vy, om, 0.1, ml, ma, reused);
      hm::VelocityLattice fresh;
      hm::build_window_lattice(ctr, lim, vx, vy, om, 0.1, ml, ma, fresh);
      ++cases;
      changes += reused.signature != last_sig;
      last_sig = reused.signature;
      bool same = reused.size() == fresh.size() && reused.signature == fresh.signature && reused.vx_values == fresh.vx_values &&
                  reused.vy_values == fresh.vy_values && reused.omega_values == fresh.omega_values && reused.ix == fresh.ix &&
                  reused.iy == fresh.iy && reused.row == fresh.row;
      if (!same) {
        ++bad;
        if (bad < 5) std::printf("ctr %d step %d: reused lattice differs (%zu vs %zu samples)\n", ctr, i, reused.size(), fresh.size());
      }
    }
  }
  std::printf("%d windows, %d pattern changes, %d bad\n", cases, changes, bad);
  return (bad || changes < 100) ? 1 : 0;
}
