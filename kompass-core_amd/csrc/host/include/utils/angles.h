// Angle normalisation helpers (reference: utils/angles.h).
#pragma once
#include <cmath>

class Angle {
 public:
  static double normalizeTo02Pi(double a) {
    a = std::fmod(a, 2 * M_PI);
    return a < 0 ? a + 2 * M_PI : a;
  }
  static double normalizeToMinusPiPlusPi(double a) {
    a = std::fmod(a + M_PI, 2 * M_PI);
    if (a < 0) a += 2 * M_PI;
    return a - M_PI;
  }
};
