"""ctypes view of include/dpomp.h and the loader of the HIP library.

There is deliberately no CPU implementation behind this module: if
``libdpomp_hip.so`` is missing the import of the filter fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPOMP_HIP_LIB") or os.path.join(HERE, "libdpomp_hip.so")  # (override: experiments, other installs)

ABI_VERSION = 2
MAX_STATES = 32
MAX_EVENTS = 32
MAX_PARAMS = 32
MAX_FACTORS = 4
COMPILED_MAX = 8  # S, E, K up to which compiled-in evaluators exist; larger models are compiled at create time (hipRTC)
FLAG_TABLE_INTERPRET, FLAG_TABLE_COMPILE = 1, 2

OK = 0
EINVAL, EDEVICE, ENOMEM, ESTATE = -1, -2, -3, -4
ST_EVENT_CAP = 1
ST_ZERO_WEIGHT = 2
RATE_LEFT, RATE_RATIO = 0, 1
OBS_GAUSSIAN = 1
OBS_USER = 100
RS_SYSTEMATIC, RS_STRATIFIED, RS_MULTINOMIAL = 1, 2, 3
(BUILTIN_GENERIC, BUILTIN_SI, BUILTIN_SIR, BUILTIN_SEI, BUILTIN_SEIR, BUILTIN_SI_FD, BUILTIN_SIR_FD,
 BUILTIN_SEI_FD, BUILTIN_SEIR_FD, BUILTIN_LOTKA, BUILTIN_ROSSMAC) = range(11)
BUILTIN_USER = 100
DBG_LOGW, DBG_OFFSPRING_END = 1, 2
EVALUATOR_BUILTIN, EVALUATOR_TABLE, EVALUATOR_TABLE_COMPILED, EVALUATOR_SOURCE = 1, 2, 3, 4
STATIC_MAX_BYTES = 16 << 20
STATIC_MAX_SEGMENTS = 4096


class RateTermC(C.Structure):
    _fields_ = [
        ("theta_index", C.c_int32),
        ("form", C.c_int32),
        ("n_factors", C.c_int32),
        ("factor_mask", C.c_uint32 * MAX_FACTORS),
        ("denom_mask", C.c_uint32),
    ]


class ObsModelC(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("state_mask", C.c_uint32),
        ("value_mask", C.c_uint32),
        ("reserved", C.c_int32),
        ("sigma", C.c_double),
    ]


class ModelDescC(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_states", C.c_int32),
        ("n_events", C.c_int32),
        ("n_params", C.c_int32),
        ("builtin_id", C.c_int32),
        ("t0_index", C.c_int32),
        ("initial_condition", C.c_int64 * MAX_STATES),
        ("transition", (C.c_int32 * MAX_STATES) * MAX_EVENTS),
        ("rate", RateTermC * MAX_EVENTS),
        ("obs", ObsModelC),
        ("n_obs", C.c_int32),
        ("n_obs_values", C.c_int32),
        ("obs_time", C.POINTER(C.c_double)),
        ("obs_id", C.POINTER(C.c_int64)),
        ("obs_value", C.POINTER(C.c_int64)),
        ("max_events_per_interval", C.c_int64),
        ("flags", C.c_int32),
        ("reserved", C.c_int32),
    ]


_PROTOTYPES = {
    "dpomp_device_count": (C.c_int, []),
    "dpomp_version": (C.c_char_p, []),
    "dpomp_last_global_error": (C.c_char_p, []),
    "dpomp_create": (C.c_int, [C.POINTER(ModelDescC), C.c_int, C.POINTER(C.c_void_p)]),
    "dpomp_create_with_source": (C.c_int, [C.POINTER(ModelDescC), C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "dpomp_check_rate_source": (C.c_int, [C.POINTER(ModelDescC), C.c_char_p, C.c_char_p, C.c_int64]),
    "dpomp_destroy": (None, [C.c_void_p]),
    "dpomp_last_error": (C.c_char_p, [C.c_void_p]),
    "dpomp_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dpomp_set_resampler": (C.c_int, [C.c_void_p, C.c_int]),
    "dpomp_set_ess_threshold": (C.c_int, [C.c_void_p, C.c_double]),
    "dpomp_set_propagate_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "dpomp_pf_loglik": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int64, C.c_uint64,
                                  C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "dpomp_pf_loglik_static": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int64, C.c_uint64,
                                         C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "dpomp_abi_sizeof": (C.c_int, [C.c_int]),
    "dpomp_get_evaluator": (C.c_int, [C.c_void_p]),
    "dpomp_set_generic_jit": (C.c_int, [C.c_int]),
    "dpomp_pop_alloc": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]),
    "dpomp_pop_free": (None, [C.c_void_p, C.c_void_p]),
    "dpomp_pf_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_int32,
                                   C.c_uint64, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "dpomp_pop_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "dpomp_pop_take": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "dpomp_pop_device_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "dpomp_pop_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "dpomp_pop_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "dpomp_simulate": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int64, C.c_uint64,
                                 C.c_int64, C.POINTER(C.c_int64)]),
    "dpomp_simulate_events": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_uint64, C.c_int64, C.c_int64,
                                        C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int64)]),
    "dpomp_debug_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "dpomp_selftest_numerics": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dpomp_selftest_division": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dpomp_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "dpomp_kernel_count": (C.c_int, []),
    "dpomp_kernel_name": (C.c_char_p, [C.c_int]),
    "dpomp_get_kernel_stat": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "dpomp_get_null_bracket_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "dpomp_get_lane_utilisation": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "dpomp_last_event_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "dpomp_debug_alloc_count": (C.c_longlong, []),
    "dpomp_static_max_bytes": (C.c_int64, []),
}

_lib = None


class DpompLibraryMissing(RuntimeError):
    pass


def _bind_to_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` (same soname as /opt/rocm's).  Two HIP
    runtimes in one process cannot both own the device ("No HIP GPUs are available" for the second), so
    when torch is installed this library binds to torch's copy, whichever of the two is loaded first.
    Without torch (e.g. from Julia) the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # already in the process; the soname lookup below resolves to it
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def load_library() -> C.CDLL:
    """Load libdpomp_hip.so (built by ``__graft_entry__.build()`` / ``build.py``)."""
    global _lib
    if _lib is not None:
        return _lib
    _bind_to_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise DpompLibraryMissing(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for the particle filter.")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _PROTOTYPES.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the library does not export it
        except AttributeError:
            if os.environ.get("DPOMP_HIP_LIB"):  # an experimental / older build loaded on purpose (tools/ab_probe.py)
                continue
            raise
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def exported_names():
    return sorted(_PROTOTYPES)
