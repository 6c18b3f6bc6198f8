// microbench_floor — what does a launch cost, and does it depend on how many XCDs the stream may use?
// Back-to-back launches of a tiny kernel (one store per workgroup) on streams created with hipExtStreamCreateWithCUMask:
// all 256 CUs, and the CUs of 1, 2, 4 XCDs under two guesses of how mask bits map to XCDs (bit i -> XCD i % 8, or i / 32);
// the kernel notes the XCC_ID of every workgroup so the guess can be checked.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_floor.hip -o tools/microbench_floor && tools/microbench_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void tiny(int *out, int *xcc) {
  if (threadIdx.x == 0) {
    out[blockIdx.x] = blockIdx.x;
    unsigned int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x] = (int)(id & 0xf);
  }
}

int main() {
  int *out, *xcc;
  CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&xcc, 4096 * 4));
  struct Cfg { const char *name; int mode; int xcds; };
  const Cfg cfgs[] = {{"all CUs (plain stream)", 0, 8}, {"mask: bits i with i % 8 < k", 1, 1}, {"mask: bits i with i % 8 < k", 1, 2},
                      {"mask: bits i with i % 8 < k", 1, 4}, {"mask: bits i with i / 32 < k", 2, 1}, {"mask: bits i with i / 32 < k", 2, 2},
                      {"mask: bits i with i / 32 < k", 2, 4}, {"mask: all 256 bits", 3, 8}};
  for (const Cfg &c : cfgs) {
    hipStream_t s;
    if (c.mode == 0) { CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    else {
      uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < 256; ++i) {
        const bool on = c.mode == 3 || (c.mode == 1 ? (i % 8) < c.xcds : (i / 32) < c.xcds);
        if (on) mask[i / 32] |= 1u << (i % 32);
      }
      CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    }
    for (int grid : {1, 125, 1024}) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, out, xcc);
      CK(hipStreamSynchronize(s));
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int w = 0; w < 2000; ++w) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, out, xcc);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      std::vector<int> h(grid);
      CK(hipMemcpy(h.data(), xcc, grid * 4, hipMemcpyDeviceToHost));
      int seen = 0; for (int v : h) seen |= 1 << v;
      printf("%-32s k=%d  grid %4d: %.2f us per launch   XCC ids seen: 0x%02x\n", c.name, c.xcds, grid, best * 1e3 / 2000, seen);
    }
    CK(hipStreamDestroy(s));
  }
  // the same tiny kernel 100 times as a captured graph (what a multi-step call could replay)
  {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int grid : {1, 125, 1024}) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
      for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, out, xcc);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("%-32s      grid %4d: %.2f us per kernel node (graphs of 100 nodes, 20 launches)\n", "hipGraph of 100 launches", grid, best * 1e3 / 2000);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  }
  return 0;
}
