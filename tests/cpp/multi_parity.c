/* nbody_create_multi from a plain C host: one context over every visible GPU (one on the test box: RCCL with a single
 * rank, every collective still runs), stepped next to an ordinary one-device context of the same system.
 *   n_dev == 1: the two trajectories must be equal in every bit (same kernels, same order);
 *   n_dev  > 1: equal to fp32 tolerance (the symmetric algorithm's summation order depends on the partition).
 * Exit code 0 = ok; 2 = no HIP device. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nbody.h"

#define CHECK(expr) do { int rc_ = (expr); if (rc_) { printf("multi parity: %s -> %d (%s)\n", #expr, rc_, nbody_last_error(ctx_for_err)); return 1; } } while (0)

static nbody_ctx *ctx_for_err = NULL;

static int run(int32_t n, int precision, double eps, int steps) {
  int32_t devs[64];
  int n_dev = nbody_device_count();
  /* NBODY_TEST_PARTS=<n> (with NBODY_RCCL_LIB = the suite's stand-in library and NBODY_MULTI_SHARE_DEVICE=1): device 0 listed n
   * times, so that a one-GPU box runs this host's n_dev > 1 branch (tests/test_multi_parts_gpu.py) */
  const char *parts = getenv("NBODY_TEST_PARTS");
  const int shared = parts && atoi(parts) > 1;
  if (shared) n_dev = atoi(parts);
  if (n_dev > 64) n_dev = 64;
  while (n_dev > 1 && n % (4096 * n_dev) != 0) --n_dev;      /* equal slices of whole i-sets */
  for (int k = 0; k < n_dev; ++k) devs[k] = shared ? 0 : k;

  float *posm = malloc(sizeof(float) * 4 * (size_t)n), *vel = malloc(sizeof(float) * 4 * (size_t)n);
  if (nbody_ic_plummer(n, 1000.0, 100.0, 1.0e4, 42u, posm, vel)) return 1;

  nbody_params p;
  nbody_default_params(&p);
  p.n_total = n; p.precision = precision; p.eps = eps;
  nbody_ctx *one = NULL, *many = NULL;
  ctx_for_err = NULL;
  CHECK(nbody_create(&p, &one));
  CHECK(nbody_create_multi(&p, devs, n_dev, &many));
  ctx_for_err = one;  CHECK(nbody_set_state_soa(one, posm, vel, n));
  ctx_for_err = many; CHECK(nbody_set_state_soa(many, posm, vel, n));
  /* the device-plumbing entry points are refused, loudly, on the multi-device context */
  void *ptr = NULL;
  if (nbody_device_ptr(many, NBODY_BUF_POSM, &ptr, NULL) != NBODY_ERR_UNSUPPORTED || nbody_step_begin(many) != NBODY_ERR_UNSUPPORTED) {
    printf("multi parity: plumbing call not refused\n"); return 1;
  }

  float size_one = 0, size_many = 0;
  nbody_particle *rec_one = malloc(sizeof(nbody_particle) * (size_t)n), *rec_many = malloc(sizeof(nbody_particle) * (size_t)n);
  for (int s = 0; s < steps; ++s) {
    ctx_for_err = one;  CHECK(nbody_tick(one, 0.01f, &size_one, rec_one, sizeof(nbody_particle)));
    ctx_for_err = many; CHECK(nbody_tick(many, 0.01f, &size_many, rec_many, sizeof(nbody_particle)));
  }
  double ke1, pe1, ke2, pe2;
  ctx_for_err = one;  CHECK(nbody_energy(one, &ke1, &pe1));
  ctx_for_err = many; CHECK(nbody_energy(many, &ke2, &pe2));
  int64_t done = 0;
  CHECK(nbody_steps_done(many, &done));

  int bad = 0;
  double worst = 0.0;
  if (n_dev == 1) {
    bad = memcmp(rec_one, rec_many, sizeof(nbody_particle) * (size_t)n) != 0 || size_one != size_many;
  } else {
    for (int32_t i = 0; i < n; ++i)
      for (int k = 0; k < 3; ++k) {
        const double d = fabs((double)rec_one[i].Position[k] - (double)rec_many[i].Position[k]);
        if (d > worst) worst = d;
      }
    bad = worst > 1e-3 * size_one * 1e-3 || size_one != size_many;   /* 1e-6 of the scene size */
  }
  bad = bad || done != steps || fabs(ke1 - ke2) > 1e-9 * fabs(ke1) + (n_dev > 1 ? 1e-6 * fabs(ke1) : 0.0) ||
        fabs(pe1 - pe2) > 1e-9 * fabs(pe1) + (n_dev > 1 ? 1e-6 * fabs(pe1) : 0.0);
  printf("multi parity: N %d precision %d eps %.2f on %d device(s), %d steps: %s (kernel %s, Size %.4f / %.4f, worst |dx| %.3g)\n",
         n, precision, eps, n_dev, steps, bad ? "MISMATCH" : (n_dev == 1 ? "bit-identical" : "within tolerance"),
         nbody_force_kernel_name(many), size_one, size_many, worst);
  nbody_destroy(one);
  nbody_destroy(many);
  free(posm); free(vel); free(rec_one); free(rec_many);
  return bad;
}

/* theta = 1.0, the reference's shipped opening angle (OctreeSearch.cpp:85), over the same device list: every device builds the
 * whole tree and walks its slice, so the records equal the one-device context's in EVERY byte whatever n_dev is. */
static int run_tree(int32_t n, int steps) {
  int32_t devs[64];
  int n_dev = nbody_device_count();
  const char *parts = getenv("NBODY_TEST_PARTS");
  const int shared = parts && atoi(parts) > 1;
  if (shared) n_dev = atoi(parts);
  if (n_dev > 64) n_dev = 64;
  while (n_dev > 1 && n % n_dev != 0) --n_dev;
  for (int k = 0; k < n_dev; ++k) devs[k] = shared ? 0 : k;
  float *posm = malloc(sizeof(float) * 4 * (size_t)n), *vel = malloc(sizeof(float) * 4 * (size_t)n);
  const float centre[3] = {0.f, 0.f, 0.f};
  if (nbody_ic_reference_box(n, 1000.0f, centre, 7u, posm, vel)) return 1;
  nbody_params p;
  nbody_default_params(&p);
  p.n_total = n; p.theta = 1.0f;
  nbody_ctx *one = NULL, *many = NULL;
  ctx_for_err = NULL;
  CHECK(nbody_create(&p, &one));
  CHECK(nbody_create_multi(&p, devs, n_dev, &many));
  ctx_for_err = one;  CHECK(nbody_set_state_soa(one, posm, vel, n));
  ctx_for_err = many; CHECK(nbody_set_state_soa(many, posm, vel, n));
  float size_one = 0, size_many = 0, theta = 0;
  nbody_particle *rec_one = malloc(sizeof(nbody_particle) * (size_t)n), *rec_many = malloc(sizeof(nbody_particle) * (size_t)n);
  for (int s = 0; s < steps; ++s) {
    ctx_for_err = one;  CHECK(nbody_tick(one, 0.01f, &size_one, rec_one, sizeof(nbody_particle)));
    ctx_for_err = many; CHECK(nbody_tick(many, 0.01f, &size_many, rec_many, sizeof(nbody_particle)));
  }
  int32_t nodes_one = 0, nodes_many = 0;
  ctx_for_err = one;  CHECK(nbody_bh_stats(one, &nodes_one, NULL, NULL));
  ctx_for_err = many; CHECK(nbody_bh_stats(many, &nodes_many, NULL, NULL));
  CHECK(nbody_get_theta(many, &theta));
  const int bad = memcmp(rec_one, rec_many, sizeof(nbody_particle) * (size_t)n) != 0 || size_one != size_many || nodes_one != nodes_many ||
                  theta != 1.0f;
  printf("multi parity: N %d theta 1.0 on %d device(s), %d frames: %s (kernel %s, Size %.4f / %.4f, %d / %d nodes)\n", n, n_dev, steps,
         bad ? "MISMATCH" : "bit-identical", nbody_force_kernel_name(many), size_one, size_many, nodes_one, nodes_many);
  nbody_destroy(one);
  nbody_destroy(many);
  free(posm); free(vel); free(rec_one); free(rec_many);
  return bad;
}

int main(void) {
  if (nbody_device_count() <= 0) { printf("multi parity: no HIP device\n"); return 2; }
  if (run_tree(2000, 4)) return 1;                           /* the shipped scene at the shipped opening angle */
  if (run_tree(32768, 3)) return 1;
  if (run(16384, NBODY_PREC_F32, 0.0, 4)) return 1;          /* one-sided kernel */
  if (run(65536, NBODY_PREC_F32, 0.0, 4)) return 1;          /* symmetric kernel (with > 1 device: the all-to-all) */
  if (run(65536, NBODY_PREC_F32_KAHAN, 0.5, 3)) return 1;
  if (run(32768, NBODY_PREC_F64, 0.0, 3)) return 1;
  printf("multi parity: ok\n");
  return 0;
}
