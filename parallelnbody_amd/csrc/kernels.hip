// gfx950 (MI355X, CDNA4) kernels of the N-body hot path.  wave = 64 lanes; no MFMA: the pair law is
// scalar fp32/fp64 FMA work, bounded by the vector-ALU issue rate.
//
// Reference lines restated on the device (paths relative to /root/reference/Source/NBody/):
//   forces_tile_kernel  <- the pair law OctreeSearch.h:101-104 summed over all j, i.e. the loop
//                          OctreeSearch.cpp:83-86 at theta = 0
//   update_kernel       <- OctreeSearch.cpp:28-31 (v += dt*a; x += dt*v)
//   bounds_kernel       <- OctreeSearch.cpp:47-56 (ComputeCubeSize)
#include "kernels.h"

#include "../../include/nbody.h"

namespace nbody {

namespace {

constexpr int kBlock = 256;   // 4 waves of 64 lanes: one per SIMD of a CU

template <typename T> struct V4;
template <> struct V4<float> { using type = float4; };
template <> struct V4<double> { using type = double4; };

__device__ __forceinline__ float rsq_dev(float x) { return __builtin_amdgcn_rsqf(x); }   // v_rsq_f32, 1 ulp
__device__ __forceinline__ double rsq_dev(double x) {
  double y = __builtin_amdgcn_rsq(x);            // v_rsq_f64: ~2^-26 relative
  const double h = 0.5 * x;
  y = y * fma(-h * y, y, 1.5);                   // two Newton steps -> full fp64
  y = y * fma(-h * y, y, 1.5);
  return y;
}

// Plain or Kahan-compensated 3-vector accumulator.
template <typename T, bool KAHAN> struct Acc3;
template <typename T> struct Acc3<T, false> {
  T x = 0, y = 0, z = 0;
  __device__ __forceinline__ void add(T s, T dx, T dy, T dz) {
    x = fma(s, dx, x); y = fma(s, dy, y); z = fma(s, dz, z);
  }
};
template <typename T> struct Acc3<T, true> {
  T x = 0, y = 0, z = 0, cx = 0, cy = 0, cz = 0;
  static __device__ __forceinline__ void kadd(T &sum, T &c, T s, T d) {
    const T yv = fma(s, d, -c);
    const T t = sum + yv;
    c = (t - sum) - yv;
    sum = t;
  }
  __device__ __forceinline__ void add(T s, T dx, T dy, T dz) {
    kadd(x, cx, s, dx); kadd(y, cy, s, dy); kadd(z, cz, s, dz);
  }
};

// One evaluation of the pair law.  mj already carries G.  EXACT = the reference's "d == 0 -> skip"
// (eps2 == 0); otherwise eps2 > 0 keeps rsq finite and coincident pairs contribute s*0 = 0.
template <typename T, bool EXACT, bool KAHAN>
__device__ __forceinline__ void interact(T xi, T yi, T zi, T xj, T yj, T zj, T mj, T eps2, Acc3<T, KAHAN> &a) {
  const T dx = xj - xi, dy = yj - yi, dz = zj - zi;
  T r2;
  if (EXACT) r2 = fma(dx, dx, fma(dy, dy, dz * dz));
  else       r2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, eps2)));
  T rinv = rsq_dev(r2);
  if (EXACT) rinv = (r2 > T(0)) ? rinv : T(0);
  const T rinv2 = rinv * rinv;
  const T s = (mj * rinv) * rinv2;
  a.add(s, dx, dy, dz);
}

// All-pairs partial accelerations.
//   grid.x : i-blocks of kBlock*IPT owned bodies (lane t holds bodies ibase + t + k*kBlock: coalesced)
//   grid.y : j chunks [c*j_chunk, min((c+1)*j_chunk, n_total)); each writes its own partial row
//   LDS    : double-buffered tile of TILE bodies (x,y,z,G*m); every lane reads the same address
//            (broadcast ds_read_b128), one barrier per tile
template <typename T, int IPT, int TILE, bool EXACT, bool KAHAN>
__global__ __launch_bounds__(kBlock) void forces_tile_kernel(const typename V4<T>::type *__restrict__ posm,
                                                             typename V4<T>::type *__restrict__ accp, int n_total,
                                                             int i_begin, int i_count, int j_chunk, T gscale, T eps2) {
  using V = typename V4<T>::type;
  constexpr int LPT = (TILE + kBlock - 1) / kBlock;   // tile elements loaded per lane
  __shared__ V sh[2][TILE];

  const int t = threadIdx.x;
  const int ibase = blockIdx.x * (kBlock * IPT);
  const int c = blockIdx.y;
  const int j0 = c * j_chunk;
  const int j1 = min(j0 + j_chunk, n_total);
  const int ntiles = (j1 > j0) ? (j1 - j0 + TILE - 1) / TILE : 0;

  T xi[IPT], yi[IPT], zi[IPT];
  Acc3<T, KAHAN> a[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int il = min(ibase + t + k * kBlock, i_count - 1);
    const V p = posm[i_begin + il];
    xi[k] = p.x; yi[k] = p.y; zi[k] = p.z;
  }

  V r[LPT];
  auto load_tile = [&](int tile) {
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      const int e = t + l * kBlock;
      if (e < TILE) {
        const int j = j0 + tile * TILE + e;
        if (j < j1) { r[l] = posm[j]; r[l].w *= gscale; }
        else        { r[l].x = 0; r[l].y = 0; r[l].z = 0; r[l].w = 0; }   // zero-mass padding
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      const int e = t + l * kBlock;
      if (e < TILE) sh[buf][e] = r[l];
    }
  };

  if (ntiles > 0) { load_tile(0); store_tile(0); }
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const int buf = tile & 1;
    const bool more = tile + 1 < ntiles;
    if (more) load_tile(tile + 1);          // global loads in flight under the tile's arithmetic
#pragma unroll 8
    for (int jj = 0; jj < TILE; ++jj) {
      const V pj = sh[buf][jj];
#pragma unroll
      for (int k = 0; k < IPT; ++k) interact<T, EXACT, KAHAN>(xi[k], yi[k], zi[k], pj.x, pj.y, pj.z, pj.w, eps2, a[k]);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int il = ibase + t + k * kBlock;
    if (il < i_count) {
      V o; o.x = a[k].x; o.y = a[k].y; o.z = a[k].z; o.w = 0;
      accp[(size_t)c * i_count + il] = o;
    }
  }
}

// Combine the j-chunk partials in chunk order (deterministic), store the acceleration, and — when
// integrate != 0 — apply the reference's update with separate multiply and add (no FMA), exactly
// as FVector's operators do: v = v + dt*a; x = x + dt*v   (OctreeSearch.cpp:29-30).
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }

template <typename T, bool KAHAN>
__global__ __launch_bounds__(kBlock) void update_kernel(typename V4<T>::type *__restrict__ posm,
                                                        typename V4<T>::type *__restrict__ vel,
                                                        typename V4<T>::type *__restrict__ acc,
                                                        const typename V4<T>::type *__restrict__ accp, int i_begin,
                                                        int i_count, int j_split, T dt, int integrate) {
  using V = typename V4<T>::type;
  const int il = blockIdx.x * kBlock + threadIdx.x;
  if (il >= i_count) return;
  T ax = 0, ay = 0, az = 0, cx = 0, cy = 0, cz = 0;
  for (int c = 0; c < j_split; ++c) {
    const V p = accp[(size_t)c * i_count + il];
    if (KAHAN) {
      T yv = p.x - cx; T tt = ax + yv; cx = (tt - ax) - yv; ax = tt;
      yv = p.y - cy; tt = ay + yv; cy = (tt - ay) - yv; ay = tt;
      yv = p.z - cz; tt = az + yv; cz = (tt - az) - yv; az = tt;
    } else {
      ax = add_rn(ax, p.x); ay = add_rn(ay, p.y); az = add_rn(az, p.z);
    }
  }
  V a; a.x = ax; a.y = ay; a.z = az; a.w = 0;
  acc[il] = a;
  if (integrate) {
    V v = vel[il];
    V x = posm[i_begin + il];
    v.x = add_rn(v.x, mul_rn(dt, ax)); v.y = add_rn(v.y, mul_rn(dt, ay)); v.z = add_rn(v.z, mul_rn(dt, az));
    x.x = add_rn(x.x, mul_rn(dt, v.x)); x.y = add_rn(x.y, mul_rn(dt, v.y)); x.z = add_rn(x.z, mul_rn(dt, v.z));
    vel[il] = v;
    posm[i_begin + il] = x;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void bounds_kernel(const typename V4<T>::type *__restrict__ posm, int i_begin,
                                                        int i_count, unsigned int *__restrict__ out_bits) {
  using V = typename V4<T>::type;
  float m = 0.0f;
  for (int il = blockIdx.x * kBlock + threadIdx.x; il < i_count; il += gridDim.x * kBlock) {
    const V p = posm[i_begin + il];
    // GetAbsMax: max(max(|X|,|Y|),|Z|); compared in fp32 like the reference's float Size
    const float v = fmaxf(fmaxf(fabsf((float)p.x), fabsf((float)p.y)), fabsf((float)p.z));
    m = fmaxf(m, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));   // non-negative floats order as uints
}

// fp64 energy diagnostic.  out[0] += sum_i 1/2 m_i v_i^2 (only by blockIdx.y == 0),
// out[1] += sum_i 1/2 m_i phi_i with phi_i = -G sum_j m_j / sqrt(d^2 + eps2), d^2+eps2 == 0 skipped.
template <typename T>
__global__ __launch_bounds__(kBlock) void energy_kernel(const typename V4<T>::type *__restrict__ posm,
                                                        const typename V4<T>::type *__restrict__ vel, int n_total,
                                                        int i_begin, int i_count, int j_chunk, double G, double eps2,
                                                        double *__restrict__ out) {
  using V = typename V4<T>::type;
  __shared__ double4 sh[kBlock];
  __shared__ double red[2][kBlock / 64];
  const int t = threadIdx.x;
  const int il = blockIdx.x * kBlock + t;
  const bool live = il < i_count;
  const int ig = i_begin + min(il, i_count - 1);
  const V pi = posm[ig];
  const double xi = pi.x, yi = pi.y, zi = pi.z, mi = pi.w;
  const int j0 = blockIdx.y * j_chunk;
  const int j1 = min(j0 + j_chunk, n_total);
  double phi = 0.0;
  for (int jt = j0; jt < j1; jt += kBlock) {
    const int j = jt + t;
    double4 q; q.x = 0; q.y = 0; q.z = 0; q.w = 0;
    if (j < j1) { const V p = posm[j]; q.x = p.x; q.y = p.y; q.z = p.z; q.w = p.w; }
    __syncthreads();
    sh[t] = q;
    __syncthreads();
#pragma unroll 4
    for (int jj = 0; jj < kBlock; ++jj) {
      const double4 pj = sh[jj];
      const double dx = pj.x - xi, dy = pj.y - yi, dz = pj.z - zi;
      const double r2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, eps2)));
      double rinv = rsq_dev(r2);
      rinv = (r2 > 0.0 && jt + jj != ig) ? rinv : 0.0;   // no self term, coincident pairs skipped
      phi = fma(pj.w, rinv, phi);
    }
  }
  double pe = live ? -0.5 * G * mi * phi : 0.0;
  double ke = 0.0;
  if (live && blockIdx.y == 0) {
    const V v = vel[il];
    const double vx = v.x, vy = v.y, vz = v.z;
    ke = 0.5 * mi * (vx * vx + vy * vy + vz * vz);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { ke += __shfl_xor(ke, off, 64); pe += __shfl_xor(pe, off, 64); }
  if ((t & 63) == 0) { red[0][t >> 6] = ke; red[1][t >> 6] = pe; }
  __syncthreads();
  if (t == 0) {
    double k = 0, p = 0;
    for (int w = 0; w < kBlock / 64; ++w) { k += red[0][w]; p += red[1][w]; }
    atomicAdd(&out[0], k);
    atomicAdd(&out[1], p);
  }
}

template <typename T, int IPT, int TILE, bool KAHAN>
hipError_t launch_forces_t(const ForceLaunch &L, hipStream_t s) {
  using V = typename V4<T>::type;
  const int iblocks = (L.i_count + kBlock * IPT - 1) / (kBlock * IPT);
  dim3 grid(iblocks, L.j_split), block(kBlock);
  const T gscale = (T)L.G, eps2 = (T)L.eps2;
  if (L.eps2 == 0.0)
    hipLaunchKernelGGL((forces_tile_kernel<T, IPT, TILE, true, KAHAN>), grid, block, 0, s, (const V *)L.posm,
                       (V *)L.accp, L.n_total, L.i_begin, L.i_count, L.j_chunk, gscale, eps2);
  else
    hipLaunchKernelGGL((forces_tile_kernel<T, IPT, TILE, false, KAHAN>), grid, block, 0, s, (const V *)L.posm,
                       (V *)L.accp, L.n_total, L.i_begin, L.i_count, L.j_chunk, gscale, eps2);
  return hipGetLastError();
}

template <typename T, int IPT, bool KAHAN>
hipError_t launch_forces_tile(const ForceLaunch &L, hipStream_t s) {
  switch (L.tile) {
    case 64:  return launch_forces_t<T, IPT, 64, KAHAN>(L, s);
    case 128: return launch_forces_t<T, IPT, 128, KAHAN>(L, s);
    case 256: return launch_forces_t<T, IPT, 256, KAHAN>(L, s);
    case 512: return launch_forces_t<T, IPT, 512, KAHAN>(L, s);
    default:  return hipErrorInvalidValue;
  }
}

template <typename T, bool KAHAN>
hipError_t launch_forces_ipt(const ForceLaunch &L, hipStream_t s) {
  switch (L.ipt) {
    case 1: return launch_forces_tile<T, 1, KAHAN>(L, s);
    case 2: return launch_forces_tile<T, 2, KAHAN>(L, s);
    case 4: return launch_forces_tile<T, 4, KAHAN>(L, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t launch_forces(const ForceLaunch &L, hipStream_t s) {
  if (L.i_count <= 0 || L.n_total <= 0 || L.j_split <= 0 || L.j_chunk <= 0) return hipErrorInvalidValue;
  if (L.j_chunk % L.tile != 0 && L.j_split > 1) return hipErrorInvalidValue;
  switch (L.precision) {
    case NBODY_PREC_F32:       return launch_forces_ipt<float, false>(L, s);
    case NBODY_PREC_F32_KAHAN: return launch_forces_ipt<float, true>(L, s);
    case NBODY_PREC_F64:       return launch_forces_ipt<double, false>(L, s);
    default: return hipErrorInvalidValue;
  }
}

void forces_geometry(const ForceLaunch &L, int *blocks, int *threads) {
  const int iblocks = (L.i_count + kBlock * L.ipt - 1) / (kBlock * L.ipt);
  if (blocks) *blocks = iblocks * L.j_split;
  if (threads) *threads = kBlock;
}

hipError_t launch_update(int precision, void *posm, void *vel, void *acc, const void *accp, int i_begin, int i_count,
                         int j_split, float dt, hipStream_t s) {
  if (i_count <= 0) return hipErrorInvalidValue;
  dim3 grid((i_count + kBlock - 1) / kBlock), block(kBlock);
  const int integrate = dt > 0.0f ? 1 : 0;
  switch (precision) {
    case NBODY_PREC_F32:
      hipLaunchKernelGGL((update_kernel<float, false>), grid, block, 0, s, (float4 *)posm, (float4 *)vel, (float4 *)acc,
                         (const float4 *)accp, i_begin, i_count, j_split, dt, integrate);
      break;
    case NBODY_PREC_F32_KAHAN:
      hipLaunchKernelGGL((update_kernel<float, true>), grid, block, 0, s, (float4 *)posm, (float4 *)vel, (float4 *)acc,
                         (const float4 *)accp, i_begin, i_count, j_split, dt, integrate);
      break;
    case NBODY_PREC_F64:
      hipLaunchKernelGGL((update_kernel<double, false>), grid, block, 0, s, (double4 *)posm, (double4 *)vel,
                         (double4 *)acc, (const double4 *)accp, i_begin, i_count, j_split, (double)dt, integrate);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_bounds(int precision, const void *posm, int i_begin, int i_count, unsigned int *out_bits,
                         hipStream_t s) {
  if (i_count <= 0) return hipErrorInvalidValue;
  int blocks = (i_count + kBlock - 1) / kBlock;
  if (blocks > 2048) blocks = 2048;
  if (precision == NBODY_PREC_F64)
    hipLaunchKernelGGL((bounds_kernel<double>), dim3(blocks), dim3(kBlock), 0, s, (const double4 *)posm, i_begin,
                       i_count, out_bits);
  else
    hipLaunchKernelGGL((bounds_kernel<float>), dim3(blocks), dim3(kBlock), 0, s, (const float4 *)posm, i_begin, i_count,
                       out_bits);
  return hipGetLastError();
}

hipError_t launch_energy(int precision, const void *posm, const void *vel, int n_total, int i_begin, int i_count,
                         double G, double eps2, double *out, hipStream_t s) {
  if (i_count <= 0) return hipErrorInvalidValue;
  const int iblocks = (i_count + kBlock - 1) / kBlock;
  int j_split = 1;
  while (iblocks * j_split < 2048 && n_total / (j_split * 2) >= 4 * kBlock) j_split *= 2;
  int j_chunk = (n_total + j_split - 1) / j_split;
  j_chunk = (j_chunk + kBlock - 1) / kBlock * kBlock;
  j_split = (n_total + j_chunk - 1) / j_chunk;
  dim3 grid(iblocks, j_split), block(kBlock);
  if (precision == NBODY_PREC_F64)
    hipLaunchKernelGGL((energy_kernel<double>), grid, block, 0, s, (const double4 *)posm, (const double4 *)vel, n_total,
                       i_begin, i_count, j_chunk, G, eps2, out);
  else
    hipLaunchKernelGGL((energy_kernel<float>), grid, block, 0, s, (const float4 *)posm, (const float4 *)vel, n_total,
                       i_begin, i_count, j_chunk, G, eps2, out);
  return hipGetLastError();
}

}  // namespace nbody
