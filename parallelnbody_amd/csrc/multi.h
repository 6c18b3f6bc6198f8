// One caller thread, several GPUs: the engine behind nbody_create_multi (include/nbody.h).  A Multi owns one ordinary
// context per device — each with an equal contiguous slice of the bodies and a full copy of the positions — plus one
// RCCL communicator per device (ncclCommInitAll: a single process, the reference's game-thread model, OctreeSearch.cpp:
// 21-34).  Per step: force pass on every device; [symmetric algorithm] the j-side sums change hands by grouped
// ncclSend / ncclRecv; kick-drift of the own slice; one in-place ncclAllGather of the positions over xGMI.
// RCCL is loaded with dlopen at creation, so libnbody_amd.so itself does not link against it.
#pragma once
#include <cstdint>
#include <string>

#include "../../include/nbody.h"

namespace nbody {

struct Multi;

int multi_create(const nbody_params *p, const int32_t *devices, int32_t n_dev, Multi **out, std::string *err);
void multi_destroy(Multi *m);
const std::string &multi_error(const Multi *m);
int multi_parts(const Multi *m);
nbody_ctx *multi_part(const Multi *m, int k);                     // the k-th device's context (owned by the Multi)
void multi_slice(const Multi *m, int k, int32_t *i_begin, int32_t *i_count);

int multi_set_particles(Multi *m, const void *aos, size_t stride, int32_t n, bool keep_history = false);
int multi_set_state_soa(Multi *m, const float *posm4, const float *vel4, int32_t n);
int multi_set_state_soa_f64(Multi *m, const double *posm4, const double *vel4, int32_t n);
int multi_forces(Multi *m, float dt);                             // one force pass + update (dt <= 0: accelerations only)
int multi_get_bounds(Multi *m, float *size);
int multi_get_positions(Multi *m, float *xyz, size_t stride, int32_t first, int32_t count);
int multi_get_particles(Multi *m, void *aos, size_t stride);
int multi_get_state_soa(Multi *m, float *posm4, float *vel4, float *acc4);
int multi_get_state_soa_f64(Multi *m, double *posm4, double *vel4, double *acc4);
int multi_energy(Multi *m, double *ke, double *pe);
int multi_synchronize(Multi *m);
int multi_kernel_time(Multi *m, int32_t which, double *total_ms, int64_t *launches);   // slowest device's total
int multi_kernel_time_reset(Multi *m);
int multi_kernel_clock(Multi *m, double *shader_mhz, int32_t *compute_units);   // the slowest clock among the devices
int multi_load_checkpoint(Multi *m, const char *path, int64_t *steps_done);

// theta > 0 (the reference's shipped algorithm, OctreeSearch.cpp:74-89 at Theta = 1.0): every device builds the same tree from the
// replicated positions and walks + integrates its own slice; one in-place all-gather of the positions per frame.
int multi_set_theta(Multi *m, float theta);
int multi_bh_steps(Multi *m, float dt, int nsteps, int *built);   // whole frames; *built: how many were (a refused frame ends the call)
int multi_bh_stats(Multi *m, int32_t *nodes, int32_t *levels, float root_com[3]);
int multi_bh_leaf_boxes(Multi *m, float *boxes, size_t stride);
int multi_bh_leaf_order(Multi *m, int32_t *order);
int multi_bh_root(Multi *m, float root_com[3], int *has_root);

// One device's share of a theta > 0 step in two halves (capi.hip; not part of the C-ABI): queue a frame — dt > 0: tree, walk of the
// own slice, kick-drift; otherwise the accelerations alone — and, later, the one wait with the frames' verdict.
int part_bh_queue_frame(nbody_ctx *c, float dt, bool diagnostic);
int part_bh_collect(nbody_ctx *c, int *status, int *built);
int part_bh_root(nbody_ctx *c, float out[3], int *has_root);

}  // namespace nbody
