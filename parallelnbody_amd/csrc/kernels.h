// Internal launcher interface between the C-ABI (capi.hip) and the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nbody {

struct ForceLaunch {
  const void *posm;     // [n_total] float4 / double4 : x,y,z,m
  void *accp;           // [j_split][i_count] float4 / double4 partial accelerations
  int n_total;
  int i_begin;
  int i_count;
  int tile;             // LDS tile, bodies
  int ipt;              // i-bodies per lane
  int j_split;          // number of j chunks
  int j_chunk;          // bodies per chunk (multiple of tile)
  double G;
  double eps2;          // > 0: softened (also the "floor" mode); == 0: exact d == 0 skip
  int zero_mode;        // for eps2 == 0: 1 = clamp trick (default), 2 = compare+select (A/B only)
  int precision;        // NBODY_PREC_*
};

// All-pairs force partials.  Returns hipSuccess or the launch error.
hipError_t launch_forces(const ForceLaunch &L, hipStream_t s);
// Blocks / threads launch_forces will use for L (for logs).
void forces_geometry(const ForceLaunch &L, int *blocks, int *threads);

// acc[i] = sum_c accp[c][i] in chunk order; if dt > 0 also v += dt*a; x += dt*v (owned slice of posm).
hipError_t launch_update(int precision, void *posm, void *vel, void *acc, const void *accp, int i_begin,
                         int i_count, int j_split, float dt, hipStream_t s);

// Symmetric (each pair once) fp32 force pass of a context that owns all bodies — kernels_sym.hip.
struct SymLaunch {
  const void *posm;     // [n_total] float4
  void *part;           // [2*T][n_pad] float4: rows 0..T-1 i-side sums (slot = partner super tile), T..2T-1 j-side sums
  const void *pairs;    // [n_pairs] int2 (si, sj), si <= sj: one workgroup each
  int n_pairs;
  int n_total;
  int S;                // bodies per super tile (multiple of 256*2*np)
  int T;                // super tiles = ceil(n_total / S)
  int n_pad;            // T * S
  int np;               // register pairs of i-bodies per lane (1 or 2)
  double G;
  double eps2;          // > 0 softened / floor; == 0 exact d == 0 skip (clamp form)
};
hipError_t launch_forces_sym(const SymLaunch &L, hipStream_t s);
hipError_t launch_update_sym(void *posm, void *vel, void *acc, const void *part, int n_total, int S, int T, int n_pad,
                             float dt, hipStream_t s);

// out_bits (uint32, pre-zeroed) = bit pattern of max_i max(|x|,|y|,|z|) over the owned slice.
hipError_t launch_bounds(int precision, const void *posm, int i_begin, int i_count, unsigned int *out_bits,
                         hipStream_t s);

// out_bits (uint32, pre-zeroed) = bit pattern of max_j |m_j| over all n_total bodies.
hipError_t launch_massmax(int precision, const void *posm, int n_total, unsigned int *out_bits, hipStream_t s);

// fp64 energy pieces: out[0] += KE of owned bodies, out[1] += sum_i 1/2 m_i phi_i  (out pre-zeroed).
hipError_t launch_energy(int precision, const void *posm, const void *vel, int n_total, int i_begin, int i_count,
                         double G, double eps2, double *out, hipStream_t s);

}  // namespace nbody
