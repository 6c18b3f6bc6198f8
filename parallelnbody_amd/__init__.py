"""parallelnbody_amd — MI355X-native all-pairs N-body engine behind the AOctreeSearch surface of
Milias/ParallelNbody.  Hot path: hand-written HIP for gfx950 in csrc/, reached through the C-ABI of
include/nbody.h (libnbody_amd.so).  No CPU fallback."""
from ._lib import (BUF_ACC, BUF_POSM, BUF_VEL, KERNEL_FORCES, KERNEL_UPDATE, PREC_F32, PREC_F32_KAHAN, PREC_F64,
                   NBodyError, build, lib)
from .actor import OctreeSearch
from .engine import PARTICLE_DTYPE, REF_DT, REF_G, NBodyEngine, device_count, ic_plummer, ic_reference_box, sym_plan, sym_plan_even, sym_plan_phased
from .sharded import EngineCreationFailed, ShardedSimulation, partition

__all__ = ["NBodyEngine", "ShardedSimulation", "EngineCreationFailed", "partition", "OctreeSearch", "NBodyError", "PARTICLE_DTYPE", "REF_G", "REF_DT", "ic_plummer",
           "ic_reference_box", "device_count", "sym_plan", "sym_plan_even", "sym_plan_phased", "build", "lib", "PREC_F32", "PREC_F32_KAHAN", "PREC_F64", "BUF_POSM",
           "BUF_VEL", "BUF_ACC", "KERNEL_FORCES", "KERNEL_UPDATE"]
