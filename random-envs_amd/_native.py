"""ctypes binding of the C-ABI declared in include/rex.h (librex_hip.so).

The library is loaded lazily and loudly: if it is missing or cannot be loaded (no ROCm runtime,
not built), every entry point raises -- the product path never falls back to CPU code.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("REX_LIB", "librex_hip.so"))   # REX_LIB: A/B builds while tuning

# every symbol include/rex.h declares (tests check the built library exports all of them)
SYMBOLS = [
    "rex_get_dims", "rex_create", "rex_destroy", "rex_set_dr", "rex_set_dr_training", "rex_set_flags",
    "rex_set_autoreset", "rex_seed", "rex_reset", "rex_step", "rex_get_state", "rex_set_state",
    "rex_get_task", "rex_set_task", "rex_set_random_task", "rex_get_obs", "rex_step_count",
    "rex_get_counters", "rex_enable_timing", "rex_read_timing", "rex_last_error", "rex_version",
    "rex_get_counters_state", "rex_set_counters_state", "rex_sample_task", "rex_set_info_buffer", "rex_export_lane", "rex_get_aux", "rex_set_aux", "rex_replay",
    "rex_get_launch_shape", "rex_set_launch_shape",
]

ENV_KINDS = {"cartpole": 0, "hopper": 1, "halfcheetah": 2, "walker2d": 3, "humanoid": 4}
DR_TYPES = {None: 0, "uniform": 1, "truncnorm": 2, "gaussian": 3, "fullgaussian": 4}


class RexDims(ctypes.Structure):
    _fields_ = [("nq", ctypes.c_int), ("nv", ctypes.c_int), ("act_dim", ctypes.c_int), ("obs_dim", ctypes.c_int),
                ("task_dim", ctypes.c_int), ("frame_skip", ctypes.c_int), ("max_episode_steps", ctypes.c_int),
                ("discrete_action", ctypes.c_int), ("dt", ctypes.c_float), ("act_low", ctypes.c_float),
                ("act_high", ctypes.c_float), ("n_info", ctypes.c_int), ("n_aux", ctypes.c_int)]


class RexError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RexError("librex_hip.so not built (%s missing): run `python -c 'import __graft_entry__ as g; "
                       "g.build()'` -- there is no CPU fallback" % LIB_PATH)
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise RexError("cannot load %s: %s" % (LIB_PATH, e))
    vp, i32, i64, u64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float
    fp = ctypes.POINTER(ctypes.c_float)
    L.rex_get_dims.argtypes = [i32, i32, ctypes.POINTER(RexDims)]
    L.rex_create.argtypes = [i32, i32, i64, i32, u64, i64, ctypes.POINTER(vp)]
    L.rex_destroy.argtypes = [vp]
    L.rex_set_dr.argtypes = [vp, i32, fp, i32, fp]
    L.rex_set_dr_training.argtypes = [vp, i32]
    L.rex_set_flags.argtypes = [vp, i32, i32, f32]
    L.rex_set_autoreset.argtypes = [vp, i32, i32]
    L.rex_seed.argtypes = [vp, u64]
    L.rex_reset.argtypes = [vp, vp, vp, vp]
    L.rex_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.rex_get_state.argtypes = [vp, vp, vp, vp]
    L.rex_set_state.argtypes = [vp, vp, vp, vp]
    L.rex_get_task.argtypes = [vp, vp, vp]
    L.rex_set_task.argtypes = [vp, vp, vp]
    L.rex_set_random_task.argtypes = [vp, vp, vp]
    L.rex_get_obs.argtypes = [vp, vp, vp]
    L.rex_step_count.argtypes = [vp]
    L.rex_step_count.restype = i64
    L.rex_get_counters.argtypes = [vp, ctypes.POINTER(i64)]
    L.rex_get_launch_shape.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    L.rex_set_launch_shape.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    L.rex_enable_timing.argtypes = [vp, i32]
    L.rex_read_timing.argtypes = [vp, fp, i32]
    L.rex_get_counters_state.argtypes = [vp, vp, vp, vp, vp]
    L.rex_set_counters_state.argtypes = [vp, vp, vp, vp, vp]
    L.rex_sample_task.argtypes = [vp, vp, u64, vp]
    L.rex_set_info_buffer.argtypes = [vp, vp]
    L.rex_export_lane.argtypes = [vp, i64, fp, fp, fp]
    L.rex_replay.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rex_get_aux.argtypes = [vp, vp, vp]
    L.rex_set_aux.argtypes = [vp, vp, vp]
    L.rex_last_error.restype = ctypes.c_char_p
    L.rex_version.restype = ctypes.c_char_p
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = lib().rex_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        raise RexError("rex error %d: %s" % (rc, msg))


def exported_symbols():
    """Names from SYMBOLS that the built library exports (no GPU needed: dlsym only)."""
    if not os.path.exists(LIB_PATH):
        raise RexError("librex_hip.so not built")
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = {line.split()[-1] for line in out.splitlines() if line.strip()}
    return [s for s in SYMBOLS if s in names]
