// fp64 form of the symmetric (each unordered pair once) force kernel — same structure as kernels_sym.hip
// (work items = an i-set in registers against a strip of 64-body subtiles, 256-body j tiles in LDS, the running j-side
// sum moving from lane to lane with its body, item-private partial-sum segments), with double state and plain v_*_f64 arithmetic:
// there is no packed fp64, a lane simply carries two i-bodies.  Pair law OctreeSearch.h:101-104 in double (build-
// defined: the reference is fp32); rsq = v_rsq_f64 + two Newton steps.
//
// One pair evaluation = 22 fp64 ops (3 sub, 3 for r^2, 6 for rsq, 4 scale factors, 6 accumulate FMAs; 20 in the
// equal-mass form, which has no mass factors) and serves two interactions; the one-sided fp64 kernel spends ~24 per interaction.  The running sums cost six v_mov_b32_dpp per step
// (a double moves as two dwords), shared by the IPT bodies of the lane.
#include "kernels.h"

#include "../../include/nbody.h"
#include "sym_common.h"

namespace nbody {

namespace {


// 1/sqrt(x): v_rsq_f64 seed (~2^-26 relative) and ONE third-order step, y (1 + e/2 + 3/8 e^2) with e = 1 - x y^2:
// the error goes to ~e^3/3 — far below 2^-53 — in five ops where two Newton steps take seven.
__device__ __forceinline__ double rsq64(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y), y, 1.0);
  const double q = e * fma(e, 0.375, 0.5);
  return fma(y, q, y);
}

// One 64-body subtile against the lane's bodies K0 .. IPT-1 (see sym_subtile in kernels_sym.hip): ONE = false, K0 = 0 is
// a symmetric strip; ONE = true is a subtile of the i-set's OWN block that lies in the lanes' slot K0 — the slots above meet
// it symmetrically, slot K0 one-sided (the self pair selected away), the slots below are idle.
// UNI: all masses equal — no mass factor per pair (20 ops instead of 22), the common G m is applied by the update; the
// j-side sums travel with the i side's sign and are negated at home (kernels_sym.hip, sym_subtile).
template <int IPT, int K0, bool ONE, bool BARE, bool SOFT, bool UNI>
__device__ __forceinline__ void sym_subtile64(const double (&xi)[IPT], const double (&yi)[IPT], const double (&zi)[IPT],
                                              const double (&nmi)[IPT], double (&ax)[IPT], double (&ay)[IPT],
                                              double (&az)[IPT], const double2 *sxy, const double2 *szw, double eps2,
                                              double &ox, double &oy, double &oz) {
  double jx = 0.0, jy = 0.0, jz = 0.0;
#pragma unroll 2
  for (int k = 0; k < 64; ++k) {
    const double2 pxy = sxy[-k], pzw = szw[-k];
#pragma unroll
    for (int q = K0; q < IPT; ++q) {
      const bool own = ONE && q == K0;
      const double dx = pxy.x - xi[q], dy = pxy.y - yi[q], dz = pzw.x - zi[q];
      const double r2 = SOFT ? fma(dx, dx, fma(dy, dy, fma(dz, dz, eps2))) : fma(dx, dx, fma(dy, dy, dz * dz));
      double rinv = rsq64(r2);
      if (!SOFT && (!BARE || own)) rinv = (r2 > 0.0) ? rinv : 0.0;
      const double u3 = (rinv * rinv) * rinv;
      const double s_i = UNI ? u3 : u3 * pzw.y;
      ax[q] = fma(s_i, dx, ax[q]); ay[q] = fma(s_i, dy, ay[q]); az[q] = fma(s_i, dz, az[q]);
      if (!own) {
        const double s_j = UNI ? u3 : u3 * nmi[q];
        jx = fma(s_j, dx, jx); jy = fma(s_j, dy, jy); jz = fma(s_j, dz, jz);
      }
    }
    if (IPT - K0 > (ONE ? 1 : 0)) { jx = wave_ror1(jx); jy = wave_ror1(jy); jz = wave_ror1(jz); }
  }
  ox = UNI ? -jx : jx; oy = UNI ? -jy : jy; oz = UNI ? -jz : jz;
}

template <int IPT, int K, bool BARE, bool SOFT, bool UNI>
__device__ __forceinline__ void own_block_subtile64(int slot, const double (&xi)[IPT], const double (&yi)[IPT],
                                                    const double (&zi)[IPT], const double (&nmi)[IPT], double (&ax)[IPT],
                                                    double (&ay)[IPT], double (&az)[IPT], const double2 *sxy,
                                                    const double2 *szw, double eps2, double &ox, double &oy, double &oz) {
  if (slot == K) sym_subtile64<IPT, K, true, BARE, SOFT, UNI>(xi, yi, zi, nmi, ax, ay, az, sxy, szw, eps2, ox, oy, oz);
  else if constexpr (K + 1 < IPT) own_block_subtile64<IPT, K + 1, BARE, SOFT, UNI>(slot, xi, yi, zi, nmi, ax, ay, az, sxy, szw, eps2, ox, oy, oz);
}

// BARE: symmetric strips without a d == 0 guard (see kernels_sym.hip); the one-sided slot of an own-block strip always selects.
// SOFT: eps2 > 0 is added to every r^2, which keeps rsq finite everywhere: no guard at all (BARE is then irrelevant).
// Work items, segments and their summation order: sym_plan.h.  State is read from posm itself (double4 is 32 bytes —
// two LDS-DMA pieces per body — so tiles are staged through registers here, and nothing spills).
// UNI / general: as in forces_sym_pk_kernel — the equal-mass launch runs when *general is clear, the general one when it
// is raised (mass_check_kernel's finding); general == nullptr runs unconditionally.
template <bool BARE, bool SOFT, int IPT, bool UNI>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(IPT == 4 ? 2 : 4, IPT == 4 ? 2 : 4)))
void forces_sym_f64_kernel(const double4 *__restrict__ posm, double4 *__restrict__ pool,
                           const SymItem *__restrict__ items, int n_total, double gscale, double eps2,
                           const int *__restrict__ dup_flag, int run_if_dup, const int *__restrict__ general,
                           int run_if_general, unsigned long long *__restrict__ clk) {
  if (dup_flag != nullptr && ((*dup_flag != 0) ? 1 : 0) != run_if_dup) return;
  if (general != nullptr && ((*general != 0) ? 1 : 0) != run_if_general) return;
  const ClockStamp stamp = clock_begin(clk);   // time_kernels: the shader clock this pass ran at (nbody_kernel_clock; four scalar registers)
  // subtile images, doubled: entries l and l+64 hold body l.  Two 16-byte planes (x, y) and (z, G m) rather than one
  // 32-byte record: a per-lane ds_read_b128 at a 32-byte stride is a 2-way bank conflict (1.1e9 conflict cycles per
  // N = 262144 pass, profiles/r02_pmc_forces_sym_f64_kernel_n262144_ipt4.txt), at a 16-byte stride it is conflict-free
  __shared__ double2 sh_xy[4][128], sh_zw[4][128];   // 8 KB each
  __shared__ double sh_acc[4][3][kJT];     // per-wave j-side sums of the tile (24 KB)

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const SymItem *__restrict__ itp = items + blockIdx.x;          // wave-uniform: scalar loads
  const int i0 = itp->i0, j0 = itp->j0, n_sub = itp->n_sub;
  const unsigned int slot_i = itp->slot_i, slot_j = itp->slot_j;
  const bool own_block = (itp->flags & kSymOneSided) != 0;      // the strip lies inside the i-set's own block
  const int n_tiles = (n_sub + 3) >> 2;
  // zero-mass padding: (padc, padc, padc, 0).  Far away where nothing else keeps it out of the sums (no d == 0 guard,
  // or no mass factor); at the origin otherwise
  const double padc = ((BARE && !SOFT) || UNI) ? kPadFar64 : 0.0;

  double xi[IPT], yi[IPT], zi[IPT], nmi[IPT], ax[IPT], ay[IPT], az[IPT];
#pragma unroll
  for (int q = 0; q < IPT; ++q) {
    const int i = i0 + t + q * kBlock;
    // clamped index, then a select per component: `i < n ? posm[i] : pad` on structs becomes a select of POINTERS with
    // the pad parked in scratch and a flat load
    double4 p = posm[min(i, n_total - 1)];
    if (i >= n_total) { p.x = padc; p.y = padc; p.z = padc; p.w = 0.0; }
    xi[q] = p.x; yi[q] = p.y; zi[q] = p.z; nmi[q] = -gscale * p.w;
    ax[q] = ay[q] = az[q] = 0.0;
  }

  auto fetch = [&](int c) {                                  // thread t owns body j0 + 256 c + t of the strip
    const int j = j0 + c * kJT + t;
    double4 q = posm[min(j, n_total - 1)];
    if (j >= n_total || 4 * c + wave >= n_sub) { q.x = padc; q.y = padc; q.z = padc; q.w = 0.0; }
    return q;
  };
  auto stage = [&](double4 q) {
    const double2 a = make_double2(q.x, q.y), b = make_double2(q.z, q.w * gscale);
    sh_xy[wave][lane] = a; sh_xy[wave][lane + 64] = a;
    sh_zw[wave][lane] = b; sh_zw[wave][lane + 64] = b;
  };
  stage(fetch(0));
  __syncthreads();

  for (int c = 0; c < n_tiles; ++c) {
    const int nsub = min(4, n_sub - 4 * c);
    const bool more = c + 1 < n_tiles;
    double4 nxt;
    if (more) nxt = fetch(c + 1);

    for (int sub = 0; sub < nsub; ++sub) {
      const double2 *sxy = &sh_xy[sub][lane + 64], *szw = &sh_zw[sub][lane + 64];
      double ox, oy, oz;
      if (!own_block)
        sym_subtile64<IPT, 0, false, BARE, SOFT, UNI>(xi, yi, zi, nmi, ax, ay, az, sxy, szw, eps2, ox, oy, oz);
      else   // the subtile's bodies sit in the lanes' slot (offset from the i-set's first body) / 256
        own_block_subtile64<IPT, 0, BARE, SOFT, UNI>((j0 - i0 + (4 * c + sub) * 64) >> 8, xi, yi, zi, nmi, ax, ay, az, sxy, szw, eps2, ox, oy, oz);
      sh_acc[wave][0][sub * 64 + lane] = ox; sh_acc[wave][1][sub * 64 + lane] = oy; sh_acc[wave][2][sub * 64 + lane] = oz;
    }
    __syncthreads();   // every wave is done with the tile images; the tile's j-side sums are complete
    if (t < nsub * 64) {
      double4 o;
      o.x = ((sh_acc[0][0][t] + sh_acc[1][0][t]) + sh_acc[2][0][t]) + sh_acc[3][0][t];
      o.y = ((sh_acc[0][1][t] + sh_acc[1][1][t]) + sh_acc[2][1][t]) + sh_acc[3][1][t];
      o.z = ((sh_acc[0][2][t] + sh_acc[1][2][t]) + sh_acc[2][2][t]) + sh_acc[3][2][t];
      o.w = 0.0;
      pool[(size_t)slot_j + (size_t)c * kJT + t] = o;
    }
    if (more) stage(nxt);
    __syncthreads();   // next tile staged; sh_acc may be rewritten
  }

  double4 *__restrict__ Pi = pool + (size_t)slot_i + t;
#pragma unroll
  for (int q = 0; q < IPT; ++q) { double4 o; o.x = ax[q]; o.y = ay[q]; o.z = az[q]; o.w = 0.0; Pi[q * kBlock] = o; }
  clock_end(clk, stamp);
}

}  // namespace

hipError_t launch_forces_sym64(const SymLaunch &L, hipStream_t s) {
  if (L.n_total <= 0 || L.n_items <= 0 || !L.pool || !L.items) return hipErrorInvalidValue;
  if (L.np != 1 && L.np != 2) return hipErrorInvalidValue;
  dim3 grid(L.n_items), block(kBlock);
  // equal masses: the device decides before every pass (SymLaunch::general, mass_check_kernel); both forms are launched,
  // one of them runs.  uni_host == 0: the host has seen different masses — general launches only.
  int *general = (int *)L.general;
  const bool run_uni = general != nullptr && L.uni_host != 0;
  const int *gate = run_uni ? general : nullptr;
  if (run_uni)
    hipLaunchKernelGGL(mass_check_kernel<double>, dim3((L.n_total + kBlock - 1) / kBlock), block, 0, s, (const double4 *)L.posm,
                       L.n_total, general);
#define NBODY_SYM64_K(BARE, SOFT, IPTV, UNI, FLAG, RUNIF)                                                         \
  hipLaunchKernelGGL((forces_sym_f64_kernel<BARE, SOFT, IPTV, UNI>), grid, block, 0, s, (const double4 *)L.posm,   \
                     (double4 *)L.pool, (const SymItem *)L.items, L.n_total, L.G, L.eps2, (const int *)(FLAG), RUNIF, gate, UNI ? 0 : 1,  \
                     (unsigned long long *)L.clk)
#define NBODY_SYM64_I(BARE, SOFT, IPTV, FLAG, RUNIF)                                                              \
  do { if (run_uni) NBODY_SYM64_K(BARE, SOFT, IPTV, true, FLAG, RUNIF); NBODY_SYM64_K(BARE, SOFT, IPTV, false, FLAG, RUNIF); } while (0)
#define NBODY_SYM64(BARE, SOFT, FLAG, RUNIF)                                                                      \
  do { if (L.np == 2) NBODY_SYM64_I(BARE, SOFT, 4, FLAG, RUNIF); else NBODY_SYM64_I(BARE, SOFT, 2, FLAG, RUNIF); } while (0)
  if (L.eps2 > 0.0) {
    NBODY_SYM64(true, true, nullptr, 0);
  } else if (L.dup_table != nullptr) {
    // the table and its flag words are zero: cleared at creation and by every pass's reduce_j_kernel
    int *flag = (int *)((unsigned long long *)L.dup_table + L.dup_slots);
    hipLaunchKernelGGL(dup_detect_kernel<double>, dim3((L.n_total + kBlock - 1) / kBlock), block, 0, s,
                       (const double4 *)L.posm, L.n_total, (unsigned long long *)L.dup_table,
                       (unsigned int)(L.dup_slots - 1), flag);
    NBODY_SYM64(true, false, flag, 0);
    NBODY_SYM64(false, false, flag, 1);
  } else {
    NBODY_SYM64(false, false, nullptr, 0);
  }
#undef NBODY_SYM64
#undef NBODY_SYM64_I
#undef NBODY_SYM64_K
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((reduce_j_kernel<double, false>), dim3((L.n_total + 63) / 64), block, 0, s,
                     (const double4 *)L.pool, (double4 *)L.send, (const unsigned int *)L.j_ptr,
                     (const unsigned int *)L.j_off, L.n_total, (unsigned long long *)L.dup_table,
                     (L.eps2 == 0.0 && L.dup_table != nullptr) ? L.dup_slots + 8 : 0, 0);
  return hipGetLastError();
}

}  // namespace nbody
