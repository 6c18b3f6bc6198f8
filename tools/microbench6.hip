// microbench6 — the inner loop of forces_sym_pk_kernel (kernels_sym.hip) on its own, taken apart: what does a wave-step
// (64 lanes x NP register pairs meet one j-body) cost in SIMD cycles at 4 waves/SIMD, and which part of it costs what?
// No barriers, no global traffic inside the timed region; LDS reads as in the kernel.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I parallelnbody_amd/csrc tools/microbench6.hip -o tools/microbench6
//   tools/microbench6
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sym_common.h"

using namespace nbody;

enum { V_NODPP = 1, V_NOJ = 2, V_NORSQ = 4, V_NOLDS = 8, V_NOI = 16,
       J_ASM6 = 32,    // j-side: the six v_fmac_f32 of a register pair as one contiguous asm block
       J_ASM12 = 64,   // all twelve of the step (NP = 2) as one block, after the i-side
       J_SIX = 128,    // six running sums (one set per register pair), folded before the dpp moves
       J_PK = 256,
       J_PK6 = 512,
       J_LDS = 1024 }; // with J_PK6: the sums travel through a wave-private LDS row (ds_write_b64 + ds_read_b64, no VALU op)  // packed partial sums that travel as they are (six dpp moves per step), folded after the 64 steps   // packed j-side: 3 v_pk_fma_f32 per pair into (lo, hi) partial sums, folded before the dpp moves

template <int NP, int UNROLL, int V, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void loop_kernel(const float4 *__restrict__ posm, float4 *__restrict__ out, long long *__restrict__ cyc, int rounds) {
  __shared__ float4 sh_pos[2][4][128];
  __shared__ float sh_acc[4][3][256];
  __shared__ f2 exch[4][3][64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  {
    const float4 q = posm[(blockIdx.x * 256 + t) & 65535];
    sh_pos[0][wave][lane] = q; sh_pos[0][wave][lane + 64] = q;
    sh_pos[1][wave][lane] = q; sh_pos[1][wave][lane + 64] = q;
  }
  f2 xi[NP], yi[NP], zi[NP], nmi[NP];
  Acc3pk<false> a[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const float4 pa = posm[(t + 512 * p + 7) & 65535], pb = posm[(t + 512 * p + 263) & 65535];
    xi[p] = f2{pa.x + 3.f, pb.x + 3.f}; yi[p] = f2{pa.y, pb.y}; zi[p] = f2{pa.z, pb.z}; nmi[p] = f2{-pa.w, -pb.w};
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) asm volatile("" ::"v"(xi[p]), "v"(yi[p]), "v"(zi[p]), "v"(nmi[p]));
  __syncthreads();
  const long long w0 = wall_clock64();
  const long long t0 = clock64();
  for (int r = 0; r < rounds; ++r) {
    const int sub = (r + wave) & 3;
    const float4 *sp = &sh_pos[r & 1][sub][lane + 64];
    float jx = 0.f, jy = 0.f, jz = 0.f;
    f2 qx = splat2(0.f), qy = splat2(0.f), qz = splat2(0.f);
    float4 pj0 = sp[0];
#pragma unroll UNROLL
    for (int k = 0; k < 64; ++k) {
      float4 pj;
      if (V & V_NOLDS) { pj = pj0; pj0.x += 1.0f; } else pj = sp[-k];
      f2 dx[NP], dy[NP], dz[NP], w[NP], u[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) { dx[p] = splat2(pj.x) - xi[p]; dy[p] = splat2(pj.y) - yi[p]; dz[p] = splat2(pj.z) - zi[p]; }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        w[p] = dz[p] * dz[p];
        w[p] = fma2(dy[p], dy[p], w[p]);
        w[p] = fma2(dx[p], dx[p], w[p]);
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (V & V_NORSQ) u[p] = w[p] + splat2(1.0f);
        else u[p] = f2{rsq_dev(w[p].x), rsq_dev(w[p].y)};
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        w[p] = u[p] * u[p];
        w[p] = w[p] * u[p];
        if (V & J_PK6) u[p] = w[p] * nmi[p];
        else if (!(V & V_NOJ)) u[p] = mul_swap(w[p], nmi[p]);
        w[p] = mul_bcast_hi(w[p], f2{pj.z, pj.w});
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (!(V & V_NOI)) a[p].add(w[p], dx[p], dy[p], dz[p]);
        else { a[p].x = a[p].x + w[p]; }
      }
      if (V & J_ASM12) {
        static_assert(NP == 2 || !(V & J_ASM12), "");
        asm volatile("v_fmac_f32 %0, %3, %7\n\tv_fmac_f32 %1, %3, %9\n\tv_fmac_f32 %2, %3, %11\n\t"
                     "v_fmac_f32 %0, %5, %13\n\tv_fmac_f32 %1, %5, %15\n\tv_fmac_f32 %2, %5, %17\n\t"
                     "v_fmac_f32 %0, %4, %8\n\tv_fmac_f32 %1, %4, %10\n\tv_fmac_f32 %2, %4, %12\n\t"
                     "v_fmac_f32 %0, %6, %14\n\tv_fmac_f32 %1, %6, %16\n\tv_fmac_f32 %2, %6, %18"
                     : "+v"(jx), "+v"(jy), "+v"(jz)
                     : "v"(u[0].x), "v"(u[0].y), "v"(u[NP - 1].x), "v"(u[NP - 1].y),
                       "v"(dx[0].y), "v"(dx[0].x), "v"(dy[0].y), "v"(dy[0].x), "v"(dz[0].y), "v"(dz[0].x),
                       "v"(dx[NP - 1].y), "v"(dx[NP - 1].x), "v"(dy[NP - 1].y), "v"(dy[NP - 1].x), "v"(dz[NP - 1].y), "v"(dz[NP - 1].x));
      } else if (V & J_ASM6) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
          asm volatile("v_fmac_f32 %0, %3, %5\n\tv_fmac_f32 %1, %3, %7\n\tv_fmac_f32 %2, %3, %9\n\t"
                       "v_fmac_f32 %0, %4, %6\n\tv_fmac_f32 %1, %4, %8\n\tv_fmac_f32 %2, %4, %10"
                       : "+v"(jx), "+v"(jy), "+v"(jz)
                       : "v"(u[p].x), "v"(u[p].y), "v"(dx[p].y), "v"(dx[p].x), "v"(dy[p].y), "v"(dy[p].x), "v"(dz[p].y), "v"(dz[p].x));
      } else if (V & J_SIX) {
        float sx[NP], sy[NP], sz[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          sx[p] = p == 0 ? jx : 0.f; sy[p] = p == 0 ? jy : 0.f; sz[p] = p == 0 ? jz : 0.f;
          if (p == 0) { sx[p] = fmaf(u[p].x, dx[p].y, sx[p]); sy[p] = fmaf(u[p].x, dy[p].y, sy[p]); sz[p] = fmaf(u[p].x, dz[p].y, sz[p]); }
          else        { sx[p] = u[p].x * dx[p].y; sy[p] = u[p].x * dy[p].y; sz[p] = u[p].x * dz[p].y; }
          sx[p] = fmaf(u[p].y, dx[p].x, sx[p]); sy[p] = fmaf(u[p].y, dy[p].x, sy[p]); sz[p] = fmaf(u[p].y, dz[p].x, sz[p]);
        }
        jx = sx[0]; jy = sy[0]; jz = sz[0];
#pragma unroll
        for (int p = 1; p < NP; ++p) { jx += sx[p]; jy += sy[p]; jz += sz[p]; }
      } else if (V & J_PK6) {
#pragma unroll
        for (int p = 0; p < NP; ++p) { qx = fma2(u[p], dx[p], qx); qy = fma2(u[p], dy[p], qy); qz = fma2(u[p], dz[p], qz); }
        if (V & V_NODPP) {
        } else if (V & J_LDS) {
          exch[wave][0][lane] = qx; exch[wave][1][lane] = qy; exch[wave][2][lane] = qz;
          const int from = (lane + 63) & 63;
          qx = exch[wave][0][from]; qy = exch[wave][1][from]; qz = exch[wave][2][from];
        } else {
          qx = f2{wave_ror1(qx.x), wave_ror1(qx.y)}; qy = f2{wave_ror1(qy.x), wave_ror1(qy.y)}; qz = f2{wave_ror1(qz.x), wave_ror1(qz.y)};
        }
      } else if (V & J_PK) {
        // u was made with mul_swap: halves swapped; use a plain product instead
        f2 px = f2{jx, 0.f}, py = f2{jy, 0.f}, pz = f2{jz, 0.f};
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const f2 sj = f2{u[p].y, u[p].x};
          px = fma2(sj, dx[p], px); py = fma2(sj, dy[p], py); pz = fma2(sj, dz[p], pz);
        }
        jx = px.x + px.y; jy = py.x + py.y; jz = pz.x + pz.y;
      } else if (!(V & V_NOJ)) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          jx = fmaf(u[p].x, dx[p].y, jx); jy = fmaf(u[p].x, dy[p].y, jy); jz = fmaf(u[p].x, dz[p].y, jz);
          jx = fmaf(u[p].y, dx[p].x, jx); jy = fmaf(u[p].y, dy[p].x, jy); jz = fmaf(u[p].y, dz[p].x, jz);
        }
      }
      if (!(V & (V_NODPP | J_PK6))) { jx = wave_ror1(jx); jy = wave_ror1(jy); jz = wave_ror1(jz); }
    }
    if (V & J_PK6) { jx = qx.x + qx.y; jy = qy.x + qy.y; jz = qz.x + qz.y; }
    sh_acc[wave][0][sub * 64 + lane] = jx; sh_acc[wave][1][sub * 64 + lane] = jy; sh_acc[wave][2][sub * 64 + lane] = jz;
  }
  const long long t1 = clock64();
  float sx = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) sx += a[p].x.x + a[p].x.y + a[p].y.x + a[p].y.y + a[p].z.x + a[p].z.y;
  out[blockIdx.x * 256 + t] = make_float4(sx, sh_acc[wave][0][t], sh_acc[wave][1][t], sh_acc[wave][2][t]);
  const long long w1 = wall_clock64();
  if (lane == 0) { cyc[blockIdx.x * 4 + wave] = t1 - t0; cyc[4096 + blockIdx.x * 4 + wave] = w0; cyc[8192 + blockIdx.x * 4 + wave] = w1;
    cyc[12288 + blockIdx.x * 4 + wave] = ((long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492); }
}


// calibration in the same run (same clock): ns per wave instruction per SIMD at 4 waves/SIMD, independent accumulators
template <int OP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void calib_kernel(float4 *__restrict__ out, int iters) {
  f2 acc[8], m[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { acc[q] = f2{(float)threadIdx.x, (float)q}; m[q] = f2{1.0f + q * 1e-7f, 1.0f - q * 1e-7f}; }
#pragma unroll
  for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(acc[q]), "+v"(m[q]));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (OP == 0) acc[q] = fma2(acc[q], m[q], m[(q + 3) & 7]);
        if (OP == 1) { acc[q].x = fmaf(m[q].y, m[(q + 3) & 7].x, acc[q].x); }
        if (OP == 2) { acc[q].x = rsq_dev(acc[q].x); }
        if (OP == 3) { acc[q].x = wave_ror1(acc[q].x); }
      }
  }
  f2 sacc = acc[0];
#pragma unroll
  for (int q = 1; q < 8; ++q) sacc = sacc + acc[q];
  out[blockIdx.x * 256 + threadIdx.x] = make_float4(sacc.x, sacc.y, 0.f, 0.f);
}

template <int OP>
void calib(const char *name, float4 *out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((calib_kernel<OP>), dim3(1024), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("calibration %-20s %8.3f ms  -> %.3f ns per wave instruction per SIMD\n", name, best, best * 1e6 / ((double)iters * 32 * 4));
}

template <int NP, int UNROLL, int V, int WAVES = 4>
void run(const char *name, const float4 *posm, float4 *out, long long *cyc) {
  const int blocks = 256 * WAVES, rounds = 400;
  std::vector<long long> h(4096 * 4);
  double best_ns = 1e30, best_cyc = 0, ghz = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL((loop_kernel<NP, UNROLL, V, WAVES>), dim3(blocks), dim3(256), 0, 0, posm, out, cyc, rounds);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    long long wmin = h[4096], wmax = h[8192];
    double csum = 0, wsum = 0;
    for (int q = 0; q < blocks * 4; ++q) {
      if (h[4096 + q] < wmin) wmin = h[4096 + q];
      if (h[8192 + q] > wmax) wmax = h[8192 + q];
      csum += (double)h[q]; wsum += (double)(h[8192 + q] - h[4096 + q]);
    }
    const double span_ns = (wmax - wmin) * 10.0;                                  // wall_clock64: 100 MHz
    const double clk = csum / (wsum * 10.0);                                      // core cycles per ns over the waves' lifetimes
    const double ns = span_ns / ((double)rounds * 64 * WAVES * NP);               // per register-pair step per SIMD
    if (ns < best_ns) { best_ns = ns; ghz = clk; best_cyc = ns * clk; }
  }
  printf("%-44s NP=%d unroll=%d waves/SIMD=%d  per register-pair step: %6.2f ns = %6.1f SIMD cycles at %.3f GHz  (x2 = %6.1f per 8 interactions)\n",
         name, NP, UNROLL, WAVES, best_ns, best_cyc, ghz, 2 * best_cyc);
}

int main() {
  const int n = 65536;
  std::vector<float4> h(n);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
  for (auto &p : h) { p.x = rnd() * 100.f; p.y = rnd() * 100.f; p.z = rnd() * 100.f; p.w = 1.0f + rnd(); }
  float4 *posm, *out; long long *cyc;
  hipMalloc(&posm, n * 16); hipMalloc(&out, 1024 * 256 * 16); hipMalloc(&cyc, 4 * 4096 * 8);
  hipMemcpy(posm, h.data(), n * 16, hipMemcpyHostToDevice);
  run<2, 4, 0>("warm-up", posm, out, cyc);
  run<4, 2, J_PK6>("packed travelling sums (6 dpp)", posm, out, cyc);
  run<4, 2, J_PK6 | J_LDS>("packed sums travelling through LDS", posm, out, cyc);
  run<4, 4, J_PK6 | J_LDS>("packed sums travelling through LDS", posm, out, cyc);
  run<2, 4, J_PK6 | J_LDS>("packed sums travelling through LDS", posm, out, cyc);
  run<4, 2, J_PK6 | V_NODPP>("packed sums, not travelling (floor)", posm, out, cyc);
  run<8, 1, J_PK6, 2>("packed travelling sums (6 dpp)", posm, out, cyc);
  run<8, 2, J_PK6, 2>("packed travelling sums (6 dpp)", posm, out, cyc);
  run<4, 4, J_PK6, 2>("packed travelling sums (6 dpp)", posm, out, cyc);
  run<6, 2, J_PK6, 2>("packed travelling sums (6 dpp)", posm, out, cyc);
  calib<0>("v_pk_fma_f32", out); calib<1>("v_fmac_f32", out); calib<2>("v_rsq_f32", out); calib<3>("v_mov_b32_dpp", out);
  run<2, 4, 0>("full step", posm, out, cyc);
  run<2, 2, 0>("full step", posm, out, cyc);
  run<2, 8, 0>("full step", posm, out, cyc);
  run<1, 4, 0>("full step", posm, out, cyc);
  run<1, 8, 0>("full step", posm, out, cyc);
  run<4, 2, 0>("full step", posm, out, cyc);
  run<4, 4, 0>("full step", posm, out, cyc);
  run<2, 4, J_ASM6>("j-side: 6 fmac per asm block", posm, out, cyc);
  run<2, 4, J_ASM12>("j-side: 12 fmac in one asm block", posm, out, cyc);
  run<2, 4, J_SIX>("j-side: six sums, folded per step", posm, out, cyc);
  run<2, 4, J_PK>("j-side: packed partial sums", posm, out, cyc);
  run<2, 4, J_PK6>("j-side: packed travelling sums (6 dpp)", posm, out, cyc);
  run<4, 2, J_PK6>("j-side: packed travelling sums (6 dpp)", posm, out, cyc);
  run<4, 4, J_PK6>("j-side: packed travelling sums (6 dpp)", posm, out, cyc);
  run<3, 2, J_PK6>("j-side: packed travelling sums (6 dpp)", posm, out, cyc);
  run<3, 4, J_PK6>("j-side: packed travelling sums (6 dpp)", posm, out, cyc);
  run<3, 4, 0>("full step", posm, out, cyc);
  run<6, 2, J_PK6>("j-side: packed travelling sums (6 dpp)", posm, out, cyc);
  run<2, 4, V_NODPP>("no dpp moves", posm, out, cyc);
  run<2, 4, V_NOJ>("no j-side (12 fmac + 2 pk_mul)", posm, out, cyc);
  run<2, 4, V_NOJ | V_NODPP>("no j-side, no dpp", posm, out, cyc);
  run<2, 4, V_NORSQ>("rsq -> pk_add", posm, out, cyc);
  run<2, 4, V_NOLDS>("no LDS reads", posm, out, cyc);
  run<2, 4, V_NOI>("i-side 3 pk_fma -> 1 pk_add", posm, out, cyc);
  run<2, 4, V_NOLDS | V_NODPP>("no LDS, no dpp", posm, out, cyc);
  run<2, 4, V_NOLDS | V_NODPP | V_NOJ>("no LDS, no dpp, no j-side", posm, out, cyc);
  run<2, 4, V_NOLDS | V_NODPP | V_NOJ | V_NORSQ>("no LDS, no dpp, no j-side, no rsq", posm, out, cyc);
  return 0;
}
