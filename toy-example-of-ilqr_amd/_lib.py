"""ctypes binding of libcilqr_amd.so (the C-ABI declared in include/cilqr_amd.h).

The shared library is built in-tree by ``build.py`` (hipcc, gfx950).  There is no Python or CPU
fallback for the solve path: if the library is missing, importing the solver fails loudly.
"""
import ctypes as C
import os
import pathlib

PKG_DIR = pathlib.Path(__file__).resolve().parent
# CILQR_AMD_LIB: another build of the same library (A/B measurements of kernel changes on one GPU box)
LIB_PATH = pathlib.Path(os.environ["CILQR_AMD_LIB"]) if os.environ.get("CILQR_AMD_LIB") else PKG_DIR / "libcilqr_amd.so"
LIB_PATH_DEV = PKG_DIR / "libcilqr_amd_dev.so"

OK = 0
ERR_BAD_ARG = -1
ERR_OBSTACLE_HORIZON = -2
ERR_DEVICE = -3
ERR_UNSUPPORTED = -4
ERR_NO_DEVICE = -5

MAX_HORIZON = 255
MAX_ALPHA_TRIALS = 20
PROF_SLOTS = 17
DBG_SERIAL_REF_SCAN = 1
DBG_UNIFORM_BACKWARD = 2

STATUS_NAMES = {0: "RUNNING", 1: "CONVERGED", 2: "BACKWARD_PASS_FAIL", 3: "FORWARD_PASS_FAIL",
                4: "FORWARD_PASS_SMALL_STEP"}
END_NAMES = {0: "CONVERGED", 1: "MAX_LAMB", 2: "MAX_ITER", 3: "BAD_INPUT", 4: "NOT_SOLVED"}


class CilqrParams(C.Structure):
    """struct cilqr_params — the scalars CILQRSolver's ctor reads (src/cilqr_solver.cpp:17-83)."""
    _fields_ = [
        ("N", C.c_int32), ("max_iter", C.c_int32), ("solve_type", C.c_int32),
        ("reference_point", C.c_int32), ("use_last_solution", C.c_int32), ("reserved0", C.c_int32),
        ("dt", C.c_double),
        ("w_pos", C.c_double), ("w_vel", C.c_double), ("w_yaw", C.c_double), ("w_acc", C.c_double),
        ("w_stl", C.c_double),
        ("obstacle_exp_q1", C.c_double), ("obstacle_exp_q2", C.c_double),
        ("state_exp_q1", C.c_double), ("state_exp_q2", C.c_double),
        ("alm_rho_init", C.c_double), ("alm_gamma", C.c_double), ("max_rho", C.c_double),
        ("max_mu", C.c_double),
        ("init_lamb", C.c_double), ("lamb_decay", C.c_double), ("lamb_amplify", C.c_double),
        ("max_lamb", C.c_double),
        ("convergence_threshold", C.c_double), ("accept_step_threshold", C.c_double),
        ("wheelbase", C.c_double), ("width", C.c_double), ("length", C.c_double),
        ("velo_max", C.c_double), ("velo_min", C.c_double), ("yaw_lim", C.c_double),
        ("acc_max", C.c_double), ("acc_min", C.c_double), ("stl_lim", C.c_double),
        ("d_safe", C.c_double),
    ]


class CilqrScenarioDesc(C.Structure):
    _fields_ = [
        ("lane_x", C.POINTER(C.c_double)), ("lane_y", C.POINTER(C.c_double)),
        ("lane_yaw", C.POINTER(C.c_double)),
        ("L", C.c_int32), ("M", C.c_int32),
        ("obs", C.POINTER(C.c_double)),
        ("T", C.c_int32), ("reserved0", C.c_int32),
        ("road_borders", C.c_double * 2), ("ref_velo", C.c_double),
    ]


class CilqrTraceRec(C.Structure):
    _fields_ = [("status", C.c_int32), ("trials", C.c_int32), ("accepted", C.c_int32),
                ("alpha_idx", C.c_int32), ("lamb", C.c_double), ("new_J", C.c_double)]


class CilqrResult(C.Structure):
    _fields_ = [("J_init", C.c_double), ("J_final", C.c_double), ("iters", C.c_int32),
                ("end_reason", C.c_int32), ("final_status", C.c_int32), ("ls_trials", C.c_int32),
                ("cost_evals", C.c_int32), ("trace_len", C.c_int32)]


# numpy structured dtypes with the same layout
import numpy as _np

RESULT_DTYPE = _np.dtype([("J_init", "<f8"), ("J_final", "<f8"), ("iters", "<i4"),
                          ("end_reason", "<i4"), ("final_status", "<i4"), ("ls_trials", "<i4"),
                          ("cost_evals", "<i4"), ("trace_len", "<i4")])
TRACE_DTYPE = _np.dtype([("status", "<i4"), ("trials", "<i4"), ("accepted", "<i4"),
                         ("alpha_idx", "<i4"), ("lamb", "<f8"), ("new_J", "<f8")])
assert RESULT_DTYPE.itemsize == C.sizeof(CilqrResult)
assert TRACE_DTYPE.itemsize == C.sizeof(CilqrTraceRec)

_P = C.c_void_p
_I = C.c_int32
_D = C.c_double

# name -> (restype, argtypes); mirrors include/cilqr_amd.h one to one
SIGNATURES = {
    "cilqr_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "cilqr_device_count": (C.c_int, [C.POINTER(C.c_int32)]),
    "cilqr_destroy": (C.c_int, [_P]),
    "cilqr_last_error": (C.c_char_p, []),
    "cilqr_version": (C.c_char_p, []),
    "cilqr_set_params": (C.c_int, [_P, C.POINTER(CilqrParams), _I]),
    "cilqr_set_scenarios": (C.c_int, [_P, C.POINTER(CilqrScenarioDesc), _I]),
    "cilqr_solve_batch": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I]),
    "cilqr_solve": (C.c_int, [_P, _P, C.POINTER(CilqrScenarioDesc), _P, _P, _P, _P]),
    "cilqr_solve_cache_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "cilqr_solve_batch_device": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "cilqr_advance_batch_device": (C.c_int, [_P, _I, _P, _P, _P, _P]),
    "cilqr_closed_loop_batch_device": (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cilqr_last_kernel_ms": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "cilqr_set_batches_in_flight": (C.c_int, [_P, _I]),
    "cilqr_join_device": (C.c_int, [_P, _P]),
    "cilqr_wait": (C.c_int, [_P]),
    "cilqr_slot_kernel_ms": (C.c_int, [_P, _I, C.POINTER(C.c_float)]),
    "cilqr_set_timing": (C.c_int, [_P, _I]),
    "cilqr_set_phase_profiling": (C.c_int, [_P, _I]),
    "cilqr_set_block_timeline": (C.c_int, [_P, _I]),
    "cilqr_get_block_timeline": (C.c_int, [_P, _P, _I]),
    "cilqr_get_phase_cycles": (C.c_int, [_P, _P, _I]),
    "cilqr_set_debug_flags": (C.c_int, [_P, _I]),
    "cilqr_set_helper_mode": (C.c_int, [_P, _I]),
    "cilqr_set_group_mode": (C.c_int, [_P, _I]),
    "cilqr_last_launch_info": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "cilqr_set_rollout_mode": (C.c_int, [_P, _I]),
    "cilqr_set_work_sharing": (C.c_int, [_P, _I]),
    "cilqr_work_sharing_stats": (C.c_int, [_P, _P]),
    "cilqr_set_resume_iters": (C.c_int, [_P, _I]),
    "cilqr_resume_stats": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "cilqr_set_alm_state": (C.c_int, [_P, _I, _P, _P]),
    "cilqr_get_alm_state": (C.c_int, [_P, _I, _P, _P, _P, C.POINTER(_I)]),
    "cilqr_init_traj_batch": (C.c_int, [_P, _I, _P, _P, _P]),
    "cilqr_ref_points_batch": (C.c_int, [_P, _I, _P, _P, _P, _P, _P]),
    "cilqr_total_cost_batch": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _P]),
    "cilqr_forward_pass_batch": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "cilqr_cost_derivatives_batch": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cilqr_backward_pass_batch": (C.c_int, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cilqr_detmath_eval": (C.c_int, [_P, _I, _P, _P, _I, _P]),
    "cilqr_reference_line_build": (C.c_int, [_P, _P, _I, _D, _D, _P, _P, _P, _P, _I, C.POINTER(_I)]),
    "cilqr_reference_line_position": (C.c_int, [_P, _P, _I, _D, _D, _P]),
    "cilqr_perturbed_starts": (C.c_int, [_P, _I, C.c_uint64, C.c_int64, _P]),
    "cilqr_build_routes": (C.c_int, [_P, _P, _I, _P, _I, _D, _P, _I, _D, _D, _P, _I, C.POINTER(_I), _P, _P]),
}

_libs = {}


class CilqrLibraryMissing(ImportError):
    pass


def lib_path(dev=False):
    """libcilqr_amd.so, or — dev=True — libcilqr_amd_dev.so: the same sources built with -DCILQR_DEV_BUILD, which adds
    the testing aids (cilqr_set_debug_flags), the in-kernel cycle accounting (cilqr_set_phase_profiling) and the
    CILQR_TUNE environment switches.  CILQR_AMD_LIB / CILQR_AMD_LIB_DEV name other builds (A/B measurements)."""
    env = os.environ.get("CILQR_AMD_LIB_DEV" if dev else "CILQR_AMD_LIB")
    if env:
        return pathlib.Path(env)
    return PKG_DIR / ("libcilqr_amd_dev.so" if dev else "libcilqr_amd.so")


def load(dev=False):
    """Load the library (once per flavour).  Raises CilqrLibraryMissing if it has not been built."""
    key = bool(dev)
    if key in _libs:
        return _libs[key]
    path = lib_path(dev)
    if not path.exists():
        raise CilqrLibraryMissing(
            f"{path} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The CILQR solve path has no CPU/Python fallback.")
    lib = C.CDLL(str(path), mode=getattr(os, "RTLD_NOW", 2))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _libs[key] = lib
    return lib


def library_info(dev=False):
    """what the loaded library says about itself: version string, and whether it was built with the compiler its kernels were
    validated with (build.py VALIDATED_HIPCC) — only such a build pairs trajectories per wavefront BY DEFAULT (otherwise the
    large-batch launches run one trajectory per wavefront, about half the throughput; cilqr_set_group_mode(2) still selects
    pairs explicitly).  ADVICE r05: surfaced at run time — bench.py prints it, the GPU tests that assert pairs by default ask it."""
    v = load(dev).cilqr_version().decode()
    return {"version": v, "compiler_validated": "NOT the validated" not in v,
            "pairs_per_wavefront_by_default": "NOT the validated" not in v}


class CilqrError(RuntimeError):
    def __init__(self, code, where, lib=None):
        msg = (lib or load()).cilqr_last_error()
        self.code = code
        super().__init__(f"{where} failed with code {code}: {msg.decode() if msg else ''}")


def check(code, where, lib=None):
    if code != OK:
        raise CilqrError(code, where, lib)
