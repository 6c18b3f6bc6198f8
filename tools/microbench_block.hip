// microbench_block — forces_block_pk_kernel (kernels_block.hip) on its own: whole one-launch steps of a random scene,
// against an fp64 direct sum on a few bodies.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I parallelnbody_amd/csrc tools/microbench_block.hip -o tools/microbench_block
//   tools/microbench_block N NP [uni=1] [optimistic=1] [steps=2000] [eps=0] [coincident=0]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../parallelnbody_amd/csrc/kernels_block.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 8192, np = argc > 2 ? atoi(argv[2]) : 8;
  const int uni = argc > 3 ? atoi(argv[3]) : 1, det = argc > 4 ? atoi(argv[4]) : 1, steps = argc > 5 ? atoi(argv[5]) : 2000;
  const double eps = argc > 6 ? atof(argv[6]) : 0.0;
  const int dups = argc > 7 ? atoi(argv[7]) : 0;      // bodies put on another body's position
  std::vector<float> h((size_t)n * 4), hv((size_t)n * 4, 0.f);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  for (int i = 0; i < n; ++i) {
    h[4 * i] = (float)(rnd() * 2000 - 1000); h[4 * i + 1] = (float)(rnd() * 2000 - 1000); h[4 * i + 2] = (float)(rnd() * 2000 - 1000);
    h[4 * i + 3] = uni ? 3.0f : (float)(1.0 + rnd() * 4999.0);
  }
  for (int d = 0; d < dups; ++d) {
    const int a = (int)(rnd() * n), b = (int)(rnd() * n);
    if (a != b) for (int c = 0; c < 3; ++c) h[4 * a + c] = h[4 * b + c];
  }
  float4 *p[2], *vel, *acc;
  CK(hipMalloc(&p[0], (size_t)n * 16)); CK(hipMalloc(&p[1], (size_t)n * 16)); CK(hipMalloc(&vel, (size_t)n * 16)); CK(hipMalloc(&acc, (size_t)n * 16));
  CK(hipMemcpy(p[0], h.data(), (size_t)n * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(vel, hv.data(), (size_t)n * 16, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  nbody::BlockLaunch L;
  L.n_total = n; L.i_begin = 0; L.i_count = n; L.np = np; L.G = 1e4; L.eps2 = eps * eps; L.uni = uni;
  int cur = 0;
  L.optimistic = 0;
  auto step = [&](float dt) {
    L.posm = p[cur]; L.posm_out = p[cur ^ 1]; L.vel = vel; L.acc = acc; L.dt = dt;
    hipError_t e = nbody::launch_block(L, st);
    if (dt > 0.f) cur ^= 1;
    return e;
  };
  // accelerations of the initial state against an fp64 direct sum
  CK(step(0.f));
  std::vector<float> ha((size_t)n * 4), hb((size_t)n * 4);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(hb.data(), acc, (size_t)n * 16, hipMemcpyDeviceToHost));     // guarded everywhere
  L.optimistic = det;
  CK(step(0.f));
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(ha.data(), acc, (size_t)n * 16, hipMemcpyDeviceToHost));
  const bool same = memcmp(ha.data(), hb.data(), (size_t)n * 16) == 0;
  double worst = 0;
  for (int q = 0; q < 24; ++q) {
    const int i = q < 4 ? q : (q < 8 ? n - 1 - (q - 4) : (int)(rnd() * n));
    double ax = 0, ay = 0, az = 0;
    for (int j = 0; j < n; ++j) {
      const double dx = (double)h[4 * j] - h[4 * i], dy = (double)h[4 * j + 1] - h[4 * i + 1], dz = (double)h[4 * j + 2] - h[4 * i + 2];
      const double r2 = dx * dx + dy * dy + dz * dz + eps * eps;
      if (r2 == 0) continue;
      const double sc = 1e4 * h[4 * j + 3] / (r2 * sqrt(r2));
      ax += sc * dx; ay += sc * dy; az += sc * dz;
    }
    const double ex = ha[4 * i] - ax, ey = ha[4 * i + 1] - ay, ez = ha[4 * i + 2] - az;
    const double rel = sqrt((ex * ex + ey * ey + ez * ez) / (ax * ax + ay * ay + az * az));
    if (rel > worst) worst = rel;
  }
  for (int w = 0; w < 200; ++w) CK(step(1e-4f));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int w = 0; w < steps; ++w) CK(step(1e-4f));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double us = best * 1e3 / steps, rate = (double)n * n / (us * 1e-6);
  printf("N=%d NP=%d uni=%d optimistic=%d eps=%g: %.2f us/step  %.3e pairs/s  %.1f %% of peak  max rel err (24 bodies, first pass) %.2e  %s\n",
         n, np, uni, det, eps, us, rate, rate * 20 / 157.3e12 * 100, worst, same ? "bits = guarded" : "BITS DIFFER FROM THE GUARDED PASS");
  return 0;
}
