// Second microbenchmark: does v_rsq_f32 cost more than its 8 cycles when it shares a SIMD with full-rate
// FMAs?  Streams of independent instructions in different rsq:fma groupings, plus the real clock under
// each load from s_memtime / s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>

#include <cstdio>

#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define RSQ(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(r[i]));
#define FMA12 FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) FMA(11)
#define FMA6a FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5)
#define FMA6b FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) FMA(11)

enum { P_1_12 = 0, P_4_48, P_8_96, P_HALF, P_FMA_ONLY, P_RSQ_ONLY, P_COUNT };
static const char *kNames[P_COUNT] = {"(1 rsq, 12 fma) x8", "(4 rsq, 48 fma) x2", "(8 rsq, 96 fma)",
                                      "(6 fma, 1 rsq, 6 fma) x8", "96 fma only", "8 rsq only"};

template <int P>
__global__ __launch_bounds__(256) void ub(float *out, unsigned long long *clk, int iters) {
  float x = 1.0f + threadIdx.x * 1e-6f, y = 0.999f;
  float a[12], r[8];
#pragma unroll
  for (int i = 0; i < 12; ++i) a[i] = x + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = 1.0f + i + x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (P == P_1_12) { RSQ(0) FMA12 RSQ(1) FMA12 RSQ(2) FMA12 RSQ(3) FMA12 RSQ(4) FMA12 RSQ(5) FMA12 RSQ(6) FMA12 RSQ(7) FMA12 }
    if (P == P_4_48) { RSQ(0) RSQ(1) RSQ(2) RSQ(3) FMA12 FMA12 FMA12 FMA12 RSQ(4) RSQ(5) RSQ(6) RSQ(7) FMA12 FMA12 FMA12 FMA12 }
    if (P == P_8_96) { RSQ(0) RSQ(1) RSQ(2) RSQ(3) RSQ(4) RSQ(5) RSQ(6) RSQ(7) FMA12 FMA12 FMA12 FMA12 FMA12 FMA12 FMA12 FMA12 }
    if (P == P_HALF) { FMA6a RSQ(0) FMA6b FMA6a RSQ(1) FMA6b FMA6a RSQ(2) FMA6b FMA6a RSQ(3) FMA6b FMA6a RSQ(4) FMA6b FMA6a RSQ(5) FMA6b FMA6a RSQ(6) FMA6b FMA6a RSQ(7) FMA6b }
    if (P == P_FMA_ONLY) { FMA12 FMA12 FMA12 FMA12 FMA12 FMA12 FMA12 FMA12 }
    if (P == P_RSQ_ONLY) { RSQ(0) RSQ(1) RSQ(2) RSQ(3) RSQ(4) RSQ(5) RSQ(6) RSQ(7) }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += r[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int P>
void run(float *out, unsigned long long *clk, int cus, int wps, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(ub<P>, dim3(cus * wps), dim3(256), 0, 0, out, clk, iters / 4);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(ub<P>, dim3(cus * wps), dim3(256), 0, 0, out, clk, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / (double)h[1] * 0.1;
  // one "unit" = 1 rsq + 12 fma (or 12 fma, or 1 rsq for the pure streams); 8 units per iteration
  const double units = (double)iters * 8.0 * wps;
  const double cyc24 = ms * 1e-3 * 2.4e9 / units;
  const double cycreal = ms * 1e-3 * ghz * 1e9 / units;
  printf("%-28s waves/SIMD %d  %8.3f ms  clock %.3f GHz  cycles/unit: %6.2f @2.4GHz  %6.2f @measured clock\n", kNames[P],
         wps, ms, ghz, cyc24, cycreal);
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
  const int cus = prop.multiProcessorCount;
  float *out; unsigned long long *clk;
  (void)hipMalloc(&out, sizeof(float) * 256 * cus * 8);
  (void)hipMalloc(&clk, 16);
  const int iters = 40000;
  for (int wps : {2, 8}) {
    run<P_1_12>(out, clk, cus, wps, iters);
    run<P_4_48>(out, clk, cus, wps, iters);
    run<P_8_96>(out, clk, cus, wps, iters);
    run<P_HALF>(out, clk, cus, wps, iters);
    run<P_FMA_ONLY>(out, clk, cus, wps, iters);
    run<P_RSQ_ONLY>(out, clk, cus, wps, iters);
  }
  return 0;
}
