/*
 * cpu_baseline.c — the CPU throughput baseline SURVEY 8(d)(A) / BASELINE.md section 3 ask for.  TEST/BENCH
 * INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/test_oracle.py load it; nothing under parallelnbody_amd/ does).
 *
 * The reference has no direct-sum loop to time (its only force code is the octree walk, OctreeSearch.h:99-108), so the
 * baseline is the build's own plain fp32 restatement of the pair law (OctreeSearch.h:101-104: a_i += G m_j d / |d|^3,
 * d == 0 skipped) as a CPU programmer would write it for speed: OpenMP over i, SIMD over j on structure-of-arrays
 * inputs, 1/sqrtf cubed in single precision.  It is NOT the parity oracle — nbody_oracle.c is, with the reference's
 * double-precision pow() per pair — and differs from it by fp32 rounding only (checked in tests/test_oracle.py).
 * Built with -O3 -march=native -ffast-math (oracle/Makefile: vrsqrt14ps + a Newton step on AVX-512 hosts) by the process that
 * loads it, so always for the host it runs on.
 */
#include <math.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BASE_API __attribute__((visibility("default")))

BASE_API int cpubase_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* acc[i] (rows i0..i1-1 of [n][3]) = sum over all j of the pair law; x, y, z, m are [n] each. */
BASE_API int cpubase_forces_f32(int n, const float *x, const float *y, const float *z, const float *m, float g, float eps2,
                                int i0, int i1, float *acc, int nthreads) {
  if (n < 0 || i0 < 0 || i1 > n || i0 > i1) return -1;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (int i = i0; i < i1; ++i) {
    const float xi = x[i], yi = y[i], zi = z[i];
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
#pragma omp simd reduction(+ : ax, ay, az)
    for (int j = 0; j < n; ++j) {
      const float dx = x[j] - xi, dy = y[j] - yi, dz = z[j] - zi;
      const float r2 = dx * dx + dy * dy + dz * dz + eps2;
      const float rinv = r2 > 0.0f ? 1.0f / sqrtf(r2) : 0.0f;          /* OctreeSearch.h:102: d == 0 contributes nothing */
      const float s = g * m[j] * (rinv * rinv * rinv);
      ax += s * dx; ay += s * dy; az += s * dz;
    }
    acc[3 * i + 0] = ax; acc[3 * i + 1] = ay; acc[3 * i + 2] = az;
  }
  return 0;
}

/* OctreeSearch.cpp:28-31 on SoA: v += dt a; x += dt v. */
BASE_API void cpubase_kick_drift_f32(int n, float *x, float *y, float *z, float *vx, float *vy, float *vz, const float *acc,
                                     float dt, int nthreads) {
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for simd schedule(static) num_threads(nthreads)
  for (int i = 0; i < n; ++i) {
    vx[i] += dt * acc[3 * i + 0]; vy[i] += dt * acc[3 * i + 1]; vz[i] += dt * acc[3 * i + 2];
    x[i] += dt * vx[i]; y[i] += dt * vy[i]; z[i] += dt * vz[i];
  }
}
