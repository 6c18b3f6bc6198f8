// microbench_clock — which of the two counters a wave can read follows the shader clock?
//   s_memtime     (clock64 / __builtin_readcyclecounter): "free-running counter based on the shader core clock" per the ISA manual
//   s_memrealtime (wall_clock64): fixed reference clock, hipDeviceAttributeWallClockRate kHz
// One wave runs a chain of DEPENDENT v_fma_f32 (a fixed number of cycles each, whatever the clock) between two reads of
// both counters; the host times the launch with events.  Run idle and under load (a second stream saturating the VALUs with
// packed FMAs, which pulls the clock down through the power limit): if s_memtime is the shader clock, d(memtime)/d(realtime) x
// reference rate follows the load, and cycles per dependent FMA stays put.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_clock.hip -o tools/microbench_clock && tools/microbench_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void probe(uint64_t *out, float *sink, int iters) {
  uint64_t t0, r0, t1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
  float a = threadIdx.x * 1e-9f, b = 1.0000001f, c = 1e-9f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
  sink[threadIdx.x] = a;
}

typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void load(float *sink, int iters) {
  f2 a[8], b = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
  for (int k = 0; k < 8; ++k) a[k] = f2{threadIdx.x * 1e-9f + k, 1.0f};
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = __builtin_elementwise_fma(a[k], b, c);
  float s = 0; for (int k = 0; k < 8; ++k) s += a[k].x + a[k].y;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  uint64_t *out; float *sink, *sink2;
  CK(hipMalloc(&out, 16)); CK(hipMalloc(&sink, 256)); CK(hipMalloc(&sink2, 4096 * 256 * 4));
  int wall_khz = 0, sclk_khz = 0;
  CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
  CK(hipDeviceGetAttribute(&sclk_khz, hipDeviceAttributeClockRate, 0));
  printf("hipDeviceAttributeWallClockRate %d kHz, hipDeviceAttributeClockRate %d kHz\n", wall_khz, sclk_khz);
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int iters = 200000;               // 12.8 M dependent FMAs
  for (int pass = 0; pass < 6; ++pass) {
    const bool loaded = pass >= 2 && pass < 5;
    if (loaded) hipLaunchKernelGGL(load, dim3(4096), dim3(256), 0, s2, sink2, 3000000 / (pass == 2 ? 4 : 1));
    if (loaded) { hipEvent_t w; CK(hipEventCreate(&w)); CK(hipEventRecord(w, s2)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s1));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, s1, out, sink, iters);
    CK(hipEventRecord(e1, s1));
    CK(hipStreamSynchronize(s1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t h[2]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
    const double real_s = (double)h[1] / (wall_khz * 1e3);
    printf("%s: event %.3f ms | s_memrealtime %.3f ms | s_memtime ticks %llu = %.1f MHz against s_memrealtime | %.3f s_memtime ticks per dependent v_fma_f32\n",
           loaded ? "under load" : "idle      ", ms, real_s * 1e3, (unsigned long long)h[0], (double)h[0] / real_s * 1e-6,
           (double)h[0] / ((double)iters * 64));
    CK(hipStreamSynchronize(s2));
  }
  return 0;
}
