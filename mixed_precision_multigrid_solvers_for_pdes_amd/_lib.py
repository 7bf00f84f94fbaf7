"""ctypes binding of libmghip.so (include/mghip.h).  There is no CPU fallback: a missing library,
a missing symbol or a missing device raises."""
import ctypes as C
import os

import numpy as np

from . import _build

MG_OK = 0
MG_ERR_INVALID_VALUE, MG_ERR_NO_DEVICE, MG_ERR_HIP, MG_ERR_STATE, MG_ERR_ALLOC, MG_ERR_TIMEOUT = -1, -2, -3, -4, -5, -6
MG_PLAN_PHASES = 7
PLAN_PHASE_NAMES = ("legs", "halo_copy", "halo_exchange", "coarse_allgather", "replicated_engine", "allreduce", "other")


class PlanTimeout(RuntimeError):
    """mg_plan_wait gave up (MG_ERR_TIMEOUT): the queued device work is still queued, so the process must not
    synchronise on it again -- report and leave (distributed.bench_main does, with os._exit)."""
MG_F32, MG_F64 = 0, 1
MG_JACOBI, MG_RBGS, MG_LEXGS = 0, 1, 2
MG_CYCLE_V, MG_CYCLE_W, MG_CYCLE_F = 0, 1, 2
MG_PREC_DOUBLE, MG_PREC_SINGLE, MG_PREC_MIXED_LEVELS, MG_PREC_ADAPTIVE, MG_PREC_SINGLE_MANAGED, MG_PREC_DEFECT = 0, 1, 2, 3, 4, 5

CYCLES = {"V": MG_CYCLE_V, "W": MG_CYCLE_W, "F": MG_CYCLE_F}


class MgConfig(C.Structure):
    _fields_ = [
        ("nx", C.c_int32), ("ny", C.c_int32),
        ("x0", C.c_double), ("x1", C.c_double), ("y0", C.c_double), ("y1", C.c_double),
        ("coeff", C.c_double),
        ("max_levels", C.c_int32), ("cycle", C.c_int32), ("pre", C.c_int32), ("post", C.c_int32),
        ("smoother", C.c_int32),
        ("omega", C.c_double),
        ("coarse_tol", C.c_double), ("coarse_maxit", C.c_int32),
        ("precision", C.c_int32),
        ("switch_threshold", C.c_double), ("memory_threshold_gb", C.c_double),
        ("adaptive_reference_rule", C.c_int32),
        ("device", C.c_int32), ("profile", C.c_int32), ("colour_offset", C.c_int32), ("fused", C.c_int32),
        ("tail", C.c_int32), ("fmg_cycles", C.c_int32), ("speculate", C.c_int32), ("coarse_direct", C.c_int32),
        ("mixed_split", C.c_int32),
    ]


class MgStats(C.Structure):
    _fields_ = [
        ("solve_seconds", C.c_double), ("h2d_seconds", C.c_double), ("d2h_seconds", C.c_double),
        ("initial_residual", C.c_double),
        ("precision_switches", C.c_int32), ("last_coarse_sweeps", C.c_int32),
        ("switch_reason", C.c_int32), ("fp32_floor", C.c_double),
    ]


SWITCH_REASONS = {0: None, 1: "threshold", 2: "stagnation", 3: "fp32_floor", 4: "fp32_skipped"}


class MgPlanOp(C.Structure):
    """mg_plan_op (include/mghip.h, "Cycle plans")"""
    _fields_ = [("op", C.c_int32), ("stream", C.c_int32), ("i", C.c_int32 * 24), ("d", C.c_double * 4), ("p", C.c_void_p * 8)]


(MG_PLAN_DOWN_LEG, MG_PLAN_UP_LEG, MG_PLAN_COPY2D, MG_PLAN_ADD_F64, MG_PLAN_GROUP_BEGIN, MG_PLAN_SEND, MG_PLAN_RECV,
 MG_PLAN_GROUP_END, MG_PLAN_ALLGATHER, MG_PLAN_ALLREDUCE_F64, MG_PLAN_COARSE_BEGIN, MG_PLAN_COARSE_CYCLE, MG_PLAN_COARSE_END,
 MG_PLAN_EVENT_RECORD, MG_PLAN_STREAM_WAIT, MG_PLAN_RESULT, MG_PLAN_SPAN_LEG) = range(1, 18)

_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
_pi, _pd = C.POINTER(C.c_int), C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/mghip.h declares
SIGNATURES = {
    "mg_version": (C.c_char_p, []),
    "mg_device_count": (_i, [_pi]),
    "mg_last_error": (C.c_char_p, [_vp]),
    "mg_create": (_i, [C.POINTER(MgConfig), C.POINTER(_vp)]),
    "mg_destroy": (_i, [_vp]),
    "mg_num_levels": (_i, [_vp, _pi]),
    "mg_level_shape": (_i, [_vp, _i, _pi, _pi]),
    "mg_level_timings": (_i, [_vp, _i, _pd]),
    "mg_solve": (_i, [_vp, _vp, _vp, _vp, _i, _d, _i, _pd, _i, _pi, _pi, C.POINTER(C.c_int32), C.POINTER(MgStats)]),
    "mg_iterate": (_i, [_vp, _d, _i, _pd, _i, _pi, _pi, C.POINTER(C.c_int32), C.POINTER(MgStats)]),
    "mg_set_coefficient": (_i, [_vp, _vp, _i]),
    "mg_set_shift": (_i, [_vp, _d]),
    "mg_set_rhs": (_i, [_vp, _vp, _i]),
    "mg_set_solution": (_i, [_vp, _vp, _i]),
    "mg_get_solution": (_i, [_vp, _vp, _i]),
    "mg_cycle": (_i, [_vp, _i]),
    "mg_fmg": (_i, [_vp, _i]),
    "mg_residual_norm": (_i, [_vp, _pd]),
    "mg_set_working_precision": (_i, [_vp, _i]),
    "mg_synchronize": (_i, [_vp]),
    "mg_get_stream": (_i, [_vp, C.POINTER(_vp)]),
    "mg_time_op": (_i, [_vp, _i, _i, _i, _i, _pd]),
    "mg_op_residual": (_i, [_i, _i, _i, _d, _d, _d, _vp, _vp, _vp]),
    "mg_op_residual_mixed": (_i, [_i, _i, _d, _d, _d, _vp, _vp, _vp]),
    "mg_dev_residual_f32in_f64out": (_i, [_i, _i, _i, _i, _d, _d, _d, _vp, _vp, _vp, _vp]),
    "mg_op_apply": (_i, [_i, _i, _i, _d, _d, _d, _vp, _vp]),
    "mg_op_norm": (_i, [_i, _i, _i, _d, _d, _vp, _pd]),
    "mg_op_jacobi": (_i, [_i, _i, _i, _d, _d, _d, _i, _vp, _vp, _vp]),
    "mg_op_rbgs": (_i, [_i, _i, _i, _d, _d, _d, _i, _vp, _vp, _vp]),
    "mg_op_helmholtz": (_i, [_i, _i, _i, _i, _d, _d, _d, _d, _d, _i, _vp, _vp, _vp]),
    "mg_op_residual_var": (_i, [_i, _i, _i, _d, _d, _d, _vp, _vp, _vp, _vp]),
    "mg_op_jacobi_var": (_i, [_i, _i, _i, _d, _d, _d, _i, _vp, _vp, _vp, _vp]),
    "mg_op_rbgs_var": (_i, [_i, _i, _i, _d, _d, _d, _i, _vp, _vp, _vp, _vp]),
    "mg_op_restrict_fw": (_i, [_i, _i, _i, _i, _vp, _vp]),
    "mg_op_prolong_bilinear": (_i, [_i, _i, _i, _i, _vp, _vp]),
    "mg_op_coarse_solve": (_i, [_i, _i, _i, _d, _d, _d, _d, _i, _vp, _vp, _vp, _pi]),
    "mg_dev_jacobi": (_i, [_i, _i, _i, _i, _d, _d, _d, _vp, _vp, _vp, _vp]),
    "mg_dev_rbgs_colour": (_i, [_i, _i, _i, _i, _d, _d, _d, _i, _i, _vp, _vp, _vp]),
    "mg_dev_residual": (_i, [_i, _i, _i, _i, _d, _d, _d, _vp, _vp, _vp, _vp]),
    "mg_dev_sumsq": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mg_dev_restrict_fw": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mg_dev_prolong_add": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mg_set_stream": (_i, [_vp, _vp, _i]),
    "mg_set_rhs_device": (_i, [_vp, _vp, _i, _i]),
    "mg_update_rhs_device": (_i, [_vp, _vp, _i, _i]),
    "mg_zero_solution_device": (_i, [_vp]),
    "mg_get_solution_device": (_i, [_vp, _vp, _i, _i]),
    "mg_dev_convert": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mg_dev_down_leg": (_i, [_i] * 11 + [_d] * 4 + [_i] * 3 + [_vp] * 5 + [_i, C.POINTER(C.c_int)]),
    "mg_dev_up_leg": (_i, [_i] * 13 + [_d] * 4 + [_i] * 2 + [_vp] * 4 + [_i] * 5 + [_vp] * 3),
    "mg_dev_down_leg_var": (_i, [_i] * 11 + [_d] * 4 + [_i] * 3 + [_vp] * 5 + [_i, C.POINTER(C.c_int), _vp, _vp]),
    "mg_dev_up_leg_var": (_i, [_i] * 13 + [_d] * 4 + [_i] * 2 + [_vp] * 4 + [_i] * 5 + [_vp] * 5),
    "mg_dev_span_leg_ok": (_i, [_i] * 6),
    "mg_dev_span_leg": (_i, [_i] * 13 + [_d] * 4 + [_i] * 3 + [_vp] * 6 + [_i] * 4 + [_vp] * 3),
    "mg_dev_var_rdiag": (_i, [_i] * 4 + [_d] * 3 + [_vp] * 3),
    "mg_dev_inject_ring": (_i, [_i] * 11 + [_vp] * 3),
    "mg_dev_scratch_bytes": (_i, [_i, _i, C.POINTER(C.c_int64)]),
    "mg_pitch_elems": (_i, [_i, _i, _pi]),
    "mg_plan_create": (_i, [C.POINTER(MgPlanOp), _i, _vp, _i, C.POINTER(_vp)]),
    "mg_plan_run": (_i, [_vp, _vp, _vp, _pd]),
    "mg_plan_run_async": (_i, [_vp, _vp, _vp]),
    "mg_plan_wait": (_i, [_vp, _pd]),
    "mg_plan_num_ops": (_i, [_vp, _pi]),
    "mg_plan_copy_launches": (_i, [_vp, _pi, _pi]),
    "mg_plan_profile": (_i, [_vp, _i]),
    "mg_plan_phase_times": (_i, [_vp, _pd]),
    "mg_comm_ranks": (_i, [_vp, _pi, _pi]),
    "mg_plan_error": (C.c_char_p, [_vp]),
    "mg_plan_destroy": (_i, [_vp]),
    "mg_comm_unique_id": (_i, [C.c_char_p, _vp]),
    "mg_comm_init": (_i, [C.c_char_p, _vp, _i, _i, _i, C.POINTER(_vp)]),
    "mg_comm_destroy": (_i, [_vp]),
}

_lib = None


def _share_torch_hip_runtime():
    """PyTorch's ROCm wheel bundles its own libamdhip64.so / libhsa-runtime64.so and loads them by path.
    Two HIP runtimes in one process cannot both own the device (the second reports "no ROCm-capable
    device"), so whichever of {torch, libmghip} comes first must bring in the SAME copy: when torch is
    installed but not imported yet, preload its libamdhip64.so so that libmghip.so binds to it by SONAME
    and a later `import torch` finds its own file already mapped."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def library_path():
    return os.environ.get("MGHIP_LIBRARY", _build.LIBPATH)


def load():
    """Load libmghip.so (building it first when the sources are newer and hipcc is present)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if path == _build.LIBPATH and _build.is_stale():
        try:
            _build.build_library()
        except Exception as exc:                      # no compiler on this machine: use what ships
            if not os.path.exists(path):
                raise ImportError(f"libmghip.so is not built and cannot be built here: {exc}") from exc
    if not os.path.exists(path):
        raise ImportError(f"libmghip.so not found at {path}; run `python -m "
                          "mixed_precision_multigrid_solvers_for_pdes_amd._build`")
    _share_torch_hip_runtime()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)                       # AttributeError if the ABI is incomplete
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return MG_F32
    if dt == np.float64:
        return MG_F64
    raise TypeError(f"unsupported dtype {dt}: the multigrid path computes in float32 or float64")


def np_dtype(code):
    return np.float32 if code == MG_F32 else np.float64


def last_error(handle=None):
    msg = load().mg_last_error(handle)
    return msg.decode() if msg else ""


def check(rc, handle=None):
    """Map a status code to the exception the reference raises for the same condition."""
    if rc == MG_OK:
        return
    msg = last_error(handle) or last_error(None) or f"mghip error {rc}"
    if rc in (MG_ERR_INVALID_VALUE, MG_ERR_STATE):
        raise ValueError(msg)
    if rc == MG_ERR_ALLOC:
        raise MemoryError(msg)
    if rc == MG_ERR_NO_DEVICE:
        raise RuntimeError("mghip: no usable HIP device (the multigrid path has no CPU fallback): " + msg)
    raise RuntimeError("mghip: " + msg)


def check_plan(rc, plan=None):
    """status of a mg_plan_* / mg_comm_* call"""
    if rc == MG_OK:
        return
    msg = load().mg_plan_error(plan)
    msg = (msg.decode() if msg else "") or f"mghip plan error {rc}"
    if rc == MG_ERR_INVALID_VALUE:
        raise ValueError(msg)
    if rc == MG_ERR_TIMEOUT:
        raise PlanTimeout("mghip: " + msg)
    raise RuntimeError("mghip: " + msg)


def device_count():
    n = C.c_int(0)
    rc = load().mg_device_count(C.byref(n))
    return n.value if rc == MG_OK else 0


def on_gpu_box():
    """True where the AMD compute device node exists."""
    return os.path.exists("/dev/kfd")


def as_c(a, dtype=None):
    """C-contiguous view/copy of `a` in a supported dtype (never modifies the caller's array)."""
    a = np.asarray(a)
    if dtype is None:
        dtype = a.dtype if a.dtype in (np.float32, np.float64) else np.float64
    return np.ascontiguousarray(a, dtype=dtype)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)
