"""ctypes binding of libnbody_amd.so — the C-ABI declared in include/nbody.h and include/nbody_actor.h.

There is no fallback: if the shared library is missing the import of any compute entry point raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# NBODY_AMD_LIB: another build of the same library (a tuning build from `make variant`, csrc/Makefile) — for A/B measurements;
# the shipped file is never replaced by one
LIB_PATH = os.environ.get("NBODY_AMD_LIB") or os.path.join(_HERE, "libnbody_amd.so")
CSRC = os.path.join(_HERE, "csrc")

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_NOMEM, ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
PREC_F32, PREC_F32_KAHAN, PREC_F64 = 0, 1, 2
BUF_POSM, BUF_VEL, BUF_ACC = 0, 1, 2
KERNEL_FORCES, KERNEL_UPDATE = 0, 1
ZERO_EXACT, ZERO_SELECT, ZERO_FLOOR = 0, 1, 2
ALGO_AUTO, ALGO_TILED, ALGO_SYMMETRIC = 0, 1, 2


class NBodyError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"nbody error {code}: {message}")
        self.code = code


class Params(ctypes.Structure):
    """struct nbody_params (include/nbody.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("n_total", ctypes.c_int32),
        ("i_begin", ctypes.c_int32),
        ("i_count", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("precision", ctypes.c_int32),
        ("G", ctypes.c_double),
        ("eps", ctypes.c_double),
        ("tile", ctypes.c_int32),
        ("i_per_thread", ctypes.c_int32),
        ("j_split", ctypes.c_int32),
        ("time_kernels", ctypes.c_int32),
        ("zero_mode", ctypes.c_int32),
        ("algorithm", ctypes.c_int32),
        ("theta", ctypes.c_float),
        ("bh_div_mode", ctypes.c_int32),
    ]


FLUSH_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p)
DRAW_POINT_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_float)
DRAW_BOX_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_float)

_lib = None


def build(force=False):
    """Compile libnbody_amd.so for gfx950 with hipcc (parallelnbody_amd/csrc/Makefile)."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950). parallelnbody_amd has no CPU or pure-Python compute path.")
    L = ctypes.CDLL(LIB_PATH)
    c_int, c_i32, c_i64, c_f, c_d, vp = ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p
    fp, dp = ctypes.POINTER(c_f), ctypes.POINTER(c_d)
    sz = ctypes.c_size_t

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("nbody_version", c_int)
    sig("nbody_device_count", c_int)
    sig("nbody_default_params", c_int, ctypes.POINTER(Params))
    sig("nbody_create", c_int, ctypes.POINTER(Params), ctypes.POINTER(vp))
    sig("nbody_create_multi", c_int, ctypes.POINTER(Params), ctypes.POINTER(c_i32), c_i32, ctypes.POINTER(vp))
    sig("nbody_destroy", None, vp)
    sig("nbody_last_error", ctypes.c_char_p, vp)
    sig("nbody_set_particles", c_int, vp, vp, sz, c_i32)
    sig("nbody_push_particles", c_int, vp, vp, sz, c_i32)
    sig("nbody_set_state_soa", c_int, vp, fp, fp, c_i32)
    sig("nbody_set_state_soa_f64", c_int, vp, dp, dp, c_i32)
    sig("nbody_compute_forces", c_int, vp)
    sig("nbody_step", c_int, vp, c_f, c_i32)
    sig("nbody_step_begin", c_int, vp)
    sig("nbody_step_end", c_int, vp, c_f)
    sig("nbody_step_begin_local", c_int, vp)
    sig("nbody_step_begin_remote", c_int, vp)
    sig("nbody_exchange_info", c_int, vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(c_i32))
    sig("nbody_bind_exchange", c_int, vp, vp, vp)
    sig("nbody_exchange_read_send", c_int, vp, vp)
    sig("nbody_exchange_write_recv", c_int, vp, vp)
    sig("nbody_set_theta", c_int, vp, c_f)
    sig("nbody_get_theta", c_int, vp, ctypes.POINTER(c_f))
    sig("nbody_bh_stats", c_int, vp, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), fp)
    sig("nbody_bh_leaf_boxes", c_int, vp, fp, sz)
    sig("nbody_bh_leaf_order", c_int, vp, ctypes.POINTER(c_i32))
    sig("nbody_get_bounds", c_int, vp, fp)
    sig("nbody_get_positions", c_int, vp, fp, sz, c_i32, c_i32)
    sig("nbody_get_particles", c_int, vp, vp, sz)
    sig("nbody_tick", c_int, vp, c_f, fp, vp, sz)
    sig("nbody_pin_host_buffer", c_int, vp, vp, sz)
    sig("nbody_unpin_host_buffer", c_int, vp, vp)
    sig("nbody_get_state_soa", c_int, vp, fp, fp, fp)
    sig("nbody_get_state_soa_f64", c_int, vp, dp, dp, dp)
    sig("nbody_energy", c_int, vp, dp, dp)
    sig("nbody_set_stream", c_int, vp, vp)
    sig("nbody_device_ptr", c_int, vp, c_i32, ctypes.POINTER(vp), ctypes.POINTER(sz))
    sig("nbody_bind_device_state", c_int, vp, vp, vp, vp)
    sig("nbody_synchronize", c_int, vp)
    sig("nbody_kernel_time", c_int, vp, c_i32, dp, ctypes.POINTER(c_i64))
    sig("nbody_kernel_time_reset", c_int, vp)
    sig("nbody_kernel_clock", c_int, vp, dp, ctypes.POINTER(c_i32))
    sig("nbody_get_launch_config", c_int, vp, *([ctypes.POINTER(c_i32)] * 5))
    sig("nbody_force_kernel_name", ctypes.c_char_p, vp)
    sig("nbody_get_algorithm", c_int, vp, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32))
    sig("nbody_equal_mass_form", c_int, vp, ctypes.POINTER(c_i32))
    sig("nbody_sym_pool_info", c_int, vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(c_i32))
    sig("nbody_save_checkpoint", c_int, vp, ctypes.c_char_p)
    sig("nbody_load_checkpoint", c_int, vp, ctypes.c_char_p, ctypes.POINTER(c_i64))
    sig("nbody_steps_done", c_int, vp, ctypes.POINTER(c_i64))
    # exported for tests and tuning, not in include/nbody.h
    sig("nbody_debug_bh_sort_counts", c_int, vp, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong))
    sig("nbody_debug_sym_item_clocks", c_int, vp, ctypes.POINTER(ctypes.c_uint64), c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32))
    sig("nbody_debug_bh_clocks", c_int, vp, ctypes.POINTER(ctypes.c_longlong))
    sig("nbody_ic_reference_box", c_int, c_i32, c_f, fp, ctypes.c_uint64, fp, fp)
    sig("nbody_ic_plummer", c_int, c_i32, c_d, c_d, c_d, ctypes.c_uint64, fp, fp)
    sig("nbody_block_pairs_describe", c_i32, c_i32, c_i32)
    sig("nbody_sym_plan_describe", c_int, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, ctypes.POINTER(c_i32),
        ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(c_i32), c_i32)
    sig("nbody_sym_plan_describe_tenths", c_int, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, ctypes.POINTER(c_i32),
        ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(c_i32), c_i32)
    sig("nbody_sym_plan_describe_phased", c_int, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, ctypes.c_uint64, ctypes.POINTER(c_i32),
        ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(c_i32), c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), c_i32)
    sig("nbody_sym_plan_is_even", c_i32, vp)
    sig("nbody_sym_plan_describe_even", c_int, c_i32, c_i32, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(ctypes.c_uint64),
        ctypes.POINTER(c_i32), c_i32)
    # actor mirror (include/nbody_actor.h)
    sig("nbody_actor_create", vp)
    sig("nbody_actor_destroy", None, vp)
    sig("nbody_actor_create_space_points", None, vp, c_i32, c_f)
    sig("nbody_actor_set_particles", None, vp, vp, c_i32)
    sig("nbody_actor_compute_cube_size", None, vp)
    sig("nbody_actor_create_octree", None, vp)
    sig("nbody_actor_tick", None, vp, c_f)
    sig("nbody_actor_clean_particles", None, vp)
    sig("nbody_actor_set_draw_callbacks", None, vp, FLUSH_FN, DRAW_POINT_FN, vp)
    sig("nbody_actor_set_box_callback", None, vp, DRAW_BOX_FN, vp)
    sig("nbody_actor_get_size", c_f, vp)
    sig("nbody_actor_get_initialized", c_i32, vp)
    sig("nbody_actor_num_particles", c_i32, vp)
    sig("nbody_actor_get_ph_delta_time", c_f, vp)
    sig("nbody_actor_set_ph_delta_time", None, vp, c_f)
    sig("nbody_actor_get_show_octree", c_i32, vp)
    sig("nbody_actor_set_show_octree", None, vp, c_i32)
    sig("nbody_actor_set_theta", None, vp, c_f)
    sig("nbody_actor_set_seed", None, vp, ctypes.c_uint64)
    sig("nbody_actor_set_engine", None, vp, c_i32, c_i32, c_d, c_d)
    sig("nbody_actor_set_devices", None, vp, ctypes.POINTER(c_i32), c_i32)
    sig("nbody_actor_last_status", c_i32, vp)
    sig("nbody_actor_get_particles", c_i32, vp, vp, c_i32)
    sig("nbody_actor_particle_data", vp, vp)
    sig("nbody_actor_push_particles", None, vp, vp, c_i32)
    sig("nbody_actor_release_storage", None, vp)
    _lib = L
    return L
