// Issue cost of the instructions of the Barnes-Hut term (kernels_bh_walk.hip force_term) on gfx950: SIMD cycles per wave64
// instruction at 1 / 2 / 4 / 8 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/microbench_f64.hip -o tools/microbench_f64
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

enum { M_FMA64 = 0, M_MUL64, M_RCP64, M_CVT_F64_F32, M_CVT_F32_F64, M_SQRT32, M_RCP32, M_FMA32, M_CNDMASK_FRESH, M_CMP_CNDMASK, M_ADD_U32, M_ASHR, M_COUNT };
static const char *kNames[M_COUNT] = {"v_fma_f64", "v_mul_f64", "v_rcp_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_sqrt_f32", "v_rcp_f32", "v_fma_f32",
                                      "v_cndmask_b32 (vcc written once before)", "v_cmp_lt_f32 + v_cndmask_b32 pairs (per pair)", "v_add_u32", "v_ashrrev_i32"};

template <int MODE>
__global__ __launch_bounds__(256) void ubench(double *out, int iters) {
  double d[8];
  float f[8];
  int u[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { d[i] = 1.0 + threadIdx.x * 1e-9 + i; f[i] = 1.0f + threadIdx.x * 1e-6f + i; u[i] = threadIdx.x + i; }
  const double x = 0.999999, y = 1.0000001;
  const float xf = 0.9999f, yf = 1.0001f;
  asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[0]), "v"(yf) : "vcc");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == M_FMA64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(x), "v"(y));
        else if (MODE == M_MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(y));
        else if (MODE == M_RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
        else if (MODE == M_CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
        else if (MODE == M_CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
        else if (MODE == M_SQRT32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
        else if (MODE == M_RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
        else if (MODE == M_FMA32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(xf), "v"(yf));
        else if (MODE == M_CNDMASK_FRESH) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(yf));
        else if (MODE == M_CMP_CNDMASK) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(yf) : "vcc");
        else if (MODE == M_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        else if (MODE == M_ASHR) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(u[i]));
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += d[i] + f[i] + u[i];
  if (s == 12345.678) out[0] = s;
}

template <int MODE> void run(double *out, double clk_ghz, int cus) {
  for (int wps : {1, 2, 4, 8}) {
    const int iters = 2000;
    dim3 grid(cus * wps), block(256);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(ubench<MODE>, grid, block, 0, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(ubench<MODE>, grid, block, 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double instr_per_wave = 64.0 * iters, waves_per_simd = wps;
    const double cycles = ms * 1e-3 * clk_ghz * 1e9 / (instr_per_wave * waves_per_simd);
    printf("%-48s %d waves/SIMD  %8.3f ms  %6.2f cycles per wave-instruction (at %.2f GHz nominal)\n", kNames[MODE], wps, ms, cycles, clk_ghz);
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const double ghz = p.clockRate * 1e-6;
  double *out;
  hipMalloc(&out, 64);
  printf("# %s, %d CUs, nominal clock %.2f GHz (the SIMD may run below it under load: compare rows, v_fma_f32 is the yardstick)\n", p.name, p.multiProcessorCount, ghz);
  run<M_FMA32>(out, ghz, p.multiProcessorCount);
  run<M_FMA64>(out, ghz, p.multiProcessorCount);
  run<M_MUL64>(out, ghz, p.multiProcessorCount);
  run<M_RCP64>(out, ghz, p.multiProcessorCount);
  run<M_CVT_F64_F32>(out, ghz, p.multiProcessorCount);
  run<M_CVT_F32_F64>(out, ghz, p.multiProcessorCount);
  run<M_SQRT32>(out, ghz, p.multiProcessorCount);
  run<M_RCP32>(out, ghz, p.multiProcessorCount);
  run<M_CNDMASK_FRESH>(out, ghz, p.multiProcessorCount);
  run<M_CMP_CNDMASK>(out, ghz, p.multiProcessorCount);
  run<M_ADD_U32>(out, ghz, p.multiProcessorCount);
  run<M_ASHR>(out, ghz, p.multiProcessorCount);
  return 0;
}
