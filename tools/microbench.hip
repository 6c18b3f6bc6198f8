// VALU issue-rate microbenchmark for gfx950: how many cycles one SIMD spends per wave64 instruction for
// the instructions of the pair-law inner loop, at 1/2/4/8 waves per SIMD.  Standalone:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench && tools/microbench
// Output feeds DESIGN.md's roofline section (the issue ceiling of the force kernel).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X X X X X X X X X X X X X X X X

enum { M_FMA = 0, M_PKFMA, M_RSQ, M_PKMUL, M_PKADD, M_MUL, M_CNDMASK, M_CMP, M_MIN, M_MIXS, M_MIXP, M_COUNT };
static const char *kNames[M_COUNT] = {"v_fma_f32", "v_pk_fma_f32", "v_rsq_f32", "v_pk_mul_f32", "v_pk_add_f32",
                                      "v_mul_f32", "v_cndmask_b32", "v_cmp_lt_f32", "v_min_f32",
                                      "mix scalar (12 fma-class + 1 rsq)", "mix packed (12 pk + 2 rsq)"};
// instructions per inner asm block
static const int kInstr[M_COUNT] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 13, 14};

template <int MODE>
__global__ __launch_bounds__(256) void ubench(float *out, int iters) {
  float x = 1.0f + threadIdx.x * 1e-6f, y = 0.999f;
  float a[16];
  f2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = x + i; p[i] = f2{x + i, y + i}; }
  f2 px = f2{x, y}, py = f2{y, x};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MODE == M_FMA) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
      } else if (MODE == M_PKFMA) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(px), "v"(py));
      } else if (MODE == M_RSQ) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
      } else if (MODE == M_PKMUL) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(py));
      } else if (MODE == M_PKADD) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(py));
      } else if (MODE == M_MUL) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
      } else if (MODE == M_CNDMASK) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(y) : );
      } else if (MODE == M_CMP) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(y) : "vcc");
      } else if (MODE == M_MIN) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      } else if (MODE == M_MIXS) {
        // the scalar pair law: 3 sub, 3 fma(r2), rsq, 3 mul, 3 fma  (4 independent copies)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          asm volatile(
              "v_sub_f32 %0, %4, %0\n v_sub_f32 %1, %4, %1\n v_sub_f32 %2, %4, %2\n"
              "v_mul_f32 %3, %0, %0\n v_fma_f32 %3, %1, %1, %3\n v_fma_f32 %3, %2, %2, %3\n"
              "v_rsq_f32 %3, %3\n"
              "v_mul_f32 %0, %3, %3\n v_mul_f32 %1, %3, %5\n v_mul_f32 %2, %0, %1\n"
              "v_fma_f32 %0, %2, %0, %4\n v_fma_f32 %1, %2, %1, %4\n v_fma_f32 %2, %2, %2, %4\n"
              : "+v"(a[4 * i]), "+v"(a[4 * i + 1]), "+v"(a[4 * i + 2]), "+v"(a[4 * i + 3])
              : "v"(x), "v"(y));
      } else if (MODE == M_MIXP) {
        // two pairs per lane, packed: 3 pk_add, 3 pk(r2), 2 rsq, 3 pk_mul, 3 pk_fma
#pragma unroll
        for (int i = 0; i < 4; ++i)
          asm volatile(
              "v_pk_add_f32 %0, %6, %0\n v_pk_add_f32 %1, %6, %1\n v_pk_add_f32 %2, %6, %2\n"
              "v_pk_mul_f32 %3, %0, %0\n v_pk_fma_f32 %3, %1, %1, %3\n v_pk_fma_f32 %3, %2, %2, %3\n"
              "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n"
              "v_pk_mul_f32 %0, %3, %3\n v_pk_mul_f32 %1, %3, %7\n v_pk_mul_f32 %2, %0, %1\n"
              "v_pk_fma_f32 %0, %2, %0, %6\n v_pk_fma_f32 %1, %2, %1, %6\n v_pk_fma_f32 %2, %2, %2, %6\n"
              : "+v"(p[4 * i]), "+v"(p[4 * i + 1]), "+v"(p[4 * i + 2]), "+v"(p[4 * i + 3]), "+v"(a[2 * i]),
                "+v"(a[2 * i + 1])
              : "v"(px), "v"(py));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
double run(float *out, int blocks, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(ubench<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters / 8);   // warm
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(ubench<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  hipEventDestroy(a); hipEventDestroy(b);
  return best;
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  printf("device %s  CUs %d  clockRate %.3f GHz\n", prop.name, cus, ghz);
  float *out;
  hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  const int iters = 20000;
  printf("%-40s %8s %12s %14s %16s\n", "instruction", "waves/SIMD", "ms", "Gwave-instr/s", "cyc/instr/SIMD@clk");
  for (int mode = 0; mode < M_COUNT; ++mode) {
    for (int wps = 1; wps <= 8; wps *= 2) {
      const int blocks = cus * wps;   // 256-thread block = 4 waves = one per SIMD
      double ms = 0;
      switch (mode) {
        case M_FMA: ms = run<M_FMA>(out, blocks, iters); break;
        case M_PKFMA: ms = run<M_PKFMA>(out, blocks, iters); break;
        case M_RSQ: ms = run<M_RSQ>(out, blocks, iters); break;
        case M_PKMUL: ms = run<M_PKMUL>(out, blocks, iters); break;
        case M_PKADD: ms = run<M_PKADD>(out, blocks, iters); break;
        case M_MUL: ms = run<M_MUL>(out, blocks, iters); break;
        case M_CNDMASK: ms = run<M_CNDMASK>(out, blocks, iters); break;
        case M_CMP: ms = run<M_CMP>(out, blocks, iters); break;
        case M_MIN: ms = run<M_MIN>(out, blocks, iters); break;
        case M_MIXS: ms = run<M_MIXS>(out, blocks, iters); break;
        case M_MIXP: ms = run<M_MIXP>(out, blocks, iters); break;
      }
      const double per_wave = (double)iters * 4 * (mode >= M_MIXS ? 4 * kInstr[mode] : kInstr[mode]);
      const double waves = (double)blocks * 4;
      const double ginstr = per_wave * waves / (ms * 1e-3) * 1e-9;
      const double simds = cus * 4.0;
      const double cyc = (ms * 1e-3) * ghz * 1e9 / (per_wave * wps);   // SIMD cycles per wave-instruction
      printf("%-40s %8d %12.3f %14.1f %16.2f\n", kNames[mode], wps, ms, ginstr, cyc);
      (void)simds;
    }
  }
  hipFree(out);
  return 0;
}
