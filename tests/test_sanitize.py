"""Host-side code under AddressSanitizer + UBSan (CPU build only; the GPU pool runs no sanitizers): the oracle's C and
the product's host-only C++ (initial-condition generators, the symmetric pass's work planner) are compiled with -fsanitize=address,undefined into a small
driver and run on the shipped-scene sizes.  The reference has no sanitizer story; its latent hazards —
unbounded recursion on coincident bodies, raw pointers into TArray storage — are the cases exercised here."""
import os
import shutil
import subprocess
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r"""
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "../include/nbody.h"
int oracle_forces_direct_f32(int, const float*, const float*, double, float, int, int, int, float*, int);
int oracle_octree_forces_f32(int, const float*, const float*, const float*, float, float, double, int, float*, float*, int*);
void oracle_kick_drift_f32(int, float*, float*, const float*, float);
float oracle_bounds_f32(int, const float*);
int oracle_octree_f32(int, const float*, const float*, const float*, float, float, double, int, int, float*, float*, int*, float*, int*);
int main(void) {
  const int n = 2000;
  float *posm = malloc(sizeof(float) * 4 * n), *vel = malloc(sizeof(float) * 4 * n);
  float *pos = malloc(sizeof(float) * 3 * n), *v3 = malloc(sizeof(float) * 3 * n), *m = malloc(sizeof(float) * n);
  float *acc = malloc(sizeof(float) * 3 * n), *acc2 = malloc(sizeof(float) * 3 * n);
  float c[3] = {0, 0, 0}, com[3];
  int nodes = 0;
  if (nbody_ic_reference_box(n, 1000.0f, c, 7, posm, vel) != 0) return 1;
  if (nbody_ic_plummer(n, 1000.0, 100.0, 1e4, 7, posm, vel) != 0) return 2;
  if (nbody_ic_reference_box(n, 1000.0f, c, 7, posm, vel) != 0) return 3;
  for (int i = 0; i < n; ++i) { for (int k = 0; k < 3; ++k) { pos[3*i+k] = posm[4*i+k]; v3[3*i+k] = vel[4*i+k]; } m[i] = posm[4*i+3]; }
  if (oracle_forces_direct_f32(n, pos, m, 1e4, 0.0f, 0, 0, n, acc, 1) != 0) return 4;
  float size = oracle_bounds_f32(n, pos);
  for (int frame = 0; frame < 3; ++frame) {
    if (oracle_octree_forces_f32(n, pos, m, c, size, 1.0f, 1e4, 0, acc2, com, &nodes) != 0) return 5;
    oracle_kick_drift_f32(n, pos, v3, acc2, 0.01f);
    c[0] = com[0]; c[1] = com[1]; c[2] = com[2];
    size = oracle_bounds_f32(n, pos);
  }
  /* coincident bodies: the restated Add must stop (rc 1), not recurse forever like the reference */
  pos[3*5] = pos[3*900]; pos[3*5+1] = pos[3*900+1]; pos[3*5+2] = pos[3*900+2];
  if (oracle_octree_forces_f32(n, pos, m, c, size, 1.0f, 1e4, 0, acc2, com, &nodes) != 1) return 6;
  if (nbody_ic_plummer(0, 1, 1, 1, 1, posm, vel) == 0) return 7;
  /* the draw walk (leaf boxes + depth-first order) and the second reading of ComputeMass' division */
  pos[3*5] += 0.5f;
  { float *boxes = malloc(sizeof(float) * 4 * n); int *order = malloc(sizeof(int) * n);
    if (oracle_octree_f32(n, pos, m, c, size, 1.0f, 1e4, 3, 1, acc2, com, &nodes, boxes, order) != 0) return 8;
    long seen = 0; for (int i = 0; i < n; ++i) seen += order[i];
    if (seen != (long)n * (n - 1) / 2) return 9;
    free(boxes); free(order); }
  /* the symmetric pass's work planner (csrc/sym_plan.cpp), headline and ragged / sharded shapes */
  { int32_t n_items = 0; uint64_t pool = 0;
    if (nbody_sym_plan_describe(1 << 20, 0, 0, 4096, 512, 6, 4, 1, &n_items, &pool, NULL, 0) != 0 || n_items < 1000) return 10;
    int32_t *items = malloc(sizeof(int32_t) * 8 * (size_t)n_items);
    if (nbody_sym_plan_describe(1 << 20, 0, 0, 4096, 512, 6, 4, 1, &n_items, &pool, items, n_items) != 0) return 11;
    free(items);
    if (nbody_sym_plan_describe(100003, 0, 0, 2048, 768, 6, 1, 2, &n_items, &pool, NULL, 0) != 0) return 12;
    if (nbody_sym_plan_describe(65536, 16384, 16384, 1024, 1024, 3, 1, 1, &n_items, &pool, NULL, 0) != 0) return 13;
    if (nbody_sym_plan_describe(65536, 100, 300, 1024, 1024, 3, 1, 1, &n_items, &pool, NULL, 0) == 0) return 14;
    /* the even-share planner: the library's mid sizes, a ragged system, more items wanted than a tiny system has steps for */
    if (nbody_sym_plan_describe_even(65536, 4096, 512, &n_items, &pool, NULL, 0) != 0 || n_items != 512) return 15;
    items = malloc(sizeof(int32_t) * 8 * (size_t)n_items);
    if (nbody_sym_plan_describe_even(65536, 4096, 512, &n_items, &pool, items, n_items) != 0) return 16;
    free(items);
    if (nbody_sym_plan_describe_even(100003, 2048, 768, &n_items, &pool, NULL, 0) != 0 || n_items != 768) return 17;
    if (nbody_sym_plan_describe_even(700, 512, 100000, &n_items, &pool, NULL, 0) != 0 || n_items < 2) return 18;
    if (nbody_sym_plan_describe_even(65536, 1000, 512, &n_items, &pool, NULL, 0) == 0) return 19; }
  printf("sanitized run ok, nodes %d, |a0| %g\n", nodes, sqrt(acc[0]*acc[0] + acc[1]*acc[1] + acc[2]*acc[2]));
  free(posm); free(vel); free(pos); free(v3); free(m); free(acc); free(acc2);
  return 0;
}
"""


@pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("g++") is None, reason="needs gcc/g++")
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    drv = tmp_path / "driver.c"
    drv.write_text(textwrap.dedent(DRIVER).replace('"../include/nbody.h"', f'"{ROOT}/include/nbody.h"'))
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-g", "-O1", "-fno-omit-frame-pointer"]
    subprocess.check_call(["gcc", "-std=c11", "-c", *san, "-ffp-contract=off", os.path.join(ROOT, "oracle", "nbody_oracle.c"),
                           "-o", str(tmp_path / "oracle.o")])
    subprocess.check_call(["g++", "-std=c++17", "-c", *san, os.path.join(ROOT, "parallelnbody_amd", "csrc", "ic.cpp"),
                           "-o", str(tmp_path / "ic.o")])
    subprocess.check_call(["g++", "-std=c++17", "-c", *san, os.path.join(ROOT, "parallelnbody_amd", "csrc", "sym_plan.cpp"),
                           "-o", str(tmp_path / "sym_plan.o")])
    subprocess.check_call(["gcc", "-std=c11", "-c", *san, str(drv), "-o", str(tmp_path / "driver.o")])
    exe = tmp_path / "driver"
    subprocess.check_call(["g++", *san, str(tmp_path / "driver.o"), str(tmp_path / "oracle.o"), str(tmp_path / "ic.o"),
                           str(tmp_path / "sym_plan.o"), "-lm", "-o", str(exe)])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sanitized run ok" in out.stdout
