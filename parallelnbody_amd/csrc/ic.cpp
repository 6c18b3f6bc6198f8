// Host-side initial-condition generators of the C-ABI (include/nbody.h).  No device code.
//   nbody_ic_reference_box <- AOctreeSearch::CreateSpacePoints, OctreeSearch.cpp:58-72 (distribution only:
//                             the reference draws from the engine's unseeded global RNG)
//   nbody_ic_plummer       <- build-defined workload (Aarseth, Henon & Wielen 1974 recipe)
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/nbody.h"

namespace {

struct Rng {   // xoshiro256** seeded by splitmix64: identical streams on every platform
  uint64_t s[4];
  explicit Rng(uint64_t seed) {
    uint64_t z = seed;
    for (int i = 0; i < 4; ++i) {
      z += 0x9E3779B97F4A7C15ull;
      uint64_t x = z;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      s[i] = x ^ (x >> 31);
    }
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }          // [0,1)
  double range(double a, double b) { return a + (b - a) * uniform(); }
  void unit_vector(double v[3]) {   // rejection in the unit ball, then normalise (as FMath::VRand does)
    for (;;) {
      v[0] = range(-1, 1); v[1] = range(-1, 1); v[2] = range(-1, 1);
      const double l2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
      if (l2 > 1e-8 && l2 <= 1.0) { const double il = 1.0 / std::sqrt(l2); v[0] *= il; v[1] *= il; v[2] *= il; return; }
    }
  }
};

}  // namespace

extern "C" {

int nbody_ic_reference_box(int32_t n, float size, const float center[3], uint64_t seed, float *posm4, float *vel4) {
  if (n <= 0 || !posm4 || !vel4 || !(size > 0.0f)) return NBODY_ERR_INVALID;
  Rng rng(seed);
  const double c[3] = {center ? center[0] : 0.0, center ? center[1] : 0.0, center ? center[2] : 0.0};
  const double s[3] = {size, size, size / 10.0};                               // OctreeSearch.cpp:61
  for (int i = 0; i < n; ++i) {
    float *p = posm4 + 4 * (size_t)i, *v = vel4 + 4 * (size_t)i;
    for (int k = 0; k < 3; ++k) p[k] = (float)(c[k] + rng.range(-s[k], s[k]));   // .cpp:64
    double d[3];
    rng.unit_vector(d);
    const double speed = 10.0 * rng.range(25.0, 50.0);                         // .cpp:65
    for (int k = 0; k < 3; ++k) v[k] = (float)(speed * d[k]);
    v[3] = 0.0f;
    p[3] = (float)rng.range(1.0, 5000.0);                                      // .cpp:66
  }
  posm4[0] = posm4[1] = posm4[2] = 0.0f; posm4[3] = 5000.0f;                    // .cpp:68-70
  vel4[0] = vel4[1] = vel4[2] = vel4[3] = 0.0f;
  return NBODY_OK;
}

int nbody_ic_plummer(int32_t n, double total_mass, double scale_radius, double G, uint64_t seed, float *posm4,
                     float *vel4) {
  if (n <= 0 || !posm4 || !vel4 || !(total_mass > 0) || !(scale_radius > 0) || !(G > 0)) return NBODY_ERR_INVALID;
  Rng rng(seed);
  std::vector<double> x((size_t)n * 3), v((size_t)n * 3);
  const double a = scale_radius;
  const double rmax = 30.0 * a;
  double cx[3] = {0, 0, 0}, cv[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) {
    double r;
    do {
      double u = rng.uniform();
      if (u < 1e-12) u = 1e-12;
      r = a / std::sqrt(std::pow(u, -2.0 / 3.0) - 1.0);
    } while (!(r < rmax));
    double d[3];
    rng.unit_vector(d);
    for (int k = 0; k < 3; ++k) x[3 * (size_t)i + k] = r * d[k];
    // speed: q = v / v_esc with density g(q) = q^2 (1-q^2)^(7/2), max(g) < 0.1
    double q, y;
    do { q = rng.uniform(); y = 0.1 * rng.uniform(); } while (y > q * q * std::pow(1.0 - q * q, 3.5));
    const double vesc = std::sqrt(2.0 * G * total_mass / a) * std::pow(1.0 + r * r / (a * a), -0.25);
    rng.unit_vector(d);
    for (int k = 0; k < 3; ++k) v[3 * (size_t)i + k] = q * vesc * d[k];
    for (int k = 0; k < 3; ++k) { cx[k] += x[3 * (size_t)i + k]; cv[k] += v[3 * (size_t)i + k]; }
  }
  for (int k = 0; k < 3; ++k) { cx[k] /= n; cv[k] /= n; }                       // centre-of-mass frame (equal masses)
  const float m = (float)(total_mass / n);
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) {
      posm4[4 * (size_t)i + k] = (float)(x[3 * (size_t)i + k] - cx[k]);
      vel4[4 * (size_t)i + k] = (float)(v[3 * (size_t)i + k] - cv[k]);
    }
    posm4[4 * (size_t)i + 3] = m;
    vel4[4 * (size_t)i + 3] = 0.0f;
  }
  return NBODY_OK;
}

}  // extern "C"
