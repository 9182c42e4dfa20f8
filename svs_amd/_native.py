"""
ctypes binding of libsvs_amd.so (include/svs_amd.h).

The product path has NO CPU fallback: if the HIP library has not been built, or
no MI355X is visible, the calls below raise.  (The numpy oracle lives under
/oracle and is test infrastructure; nothing here imports it.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# (SVS_AMD_LIB: another build of the same library, for same-box A/B runs of tools/)
_LIB_PATH = os.environ.get("SVS_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libsvs_amd.so")

SVS_OK = 0
SVS_ERR_INVALID = -1
SVS_ERR_SHAPE = -2
SVS_ERR_DEVICE = -3
SVS_ERR_NOMEM = -4
SVS_ERR_UNSUPPORTED = -5

DTYPE_F32 = 0
DTYPE_F16 = 1
DTYPE_FP8 = 2


class IndexInfo(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("d", C.c_int32),
        ("ld", C.c_int32),
        ("dtype", C.c_int32),
        ("device", C.c_int32),
        ("row_offset", C.c_int64),
        ("hbm_bytes", C.c_int64),
        ("n_masked", C.c_int64),
    ]


class Timing(C.Structure):
    _fields_ = [
        ("score_ms_sum", C.c_double),
        ("select_ms_sum", C.c_double),
        ("launches", C.c_int64),
        ("dominant_ms_sum", C.c_double),
    ]


# every symbol include/svs_amd.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "svs_version": (C.c_char_p, []),
    "svs_last_error": (C.c_char_p, []),
    "svs_device_count": (C.c_int32, []),
    "svs_device_memory": (C.c_int32, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "svs_index_create": (C.c_int32, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.POINTER(_P)]),
    "svs_index_create_from_device": (C.c_int32, [_P, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.POINTER(_P)]),
    "svs_index_append": (C.c_int32, [_P, _P, C.c_int64]),
    "svs_index_append_from_device": (C.c_int32, [_P, _P, C.c_int64, C.c_int64]),
    "svs_index_reserve": (C.c_int32, [_P, C.c_int64]),
    "svs_index_staging_acquire": (C.c_int32, [_P, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "svs_index_staging_commit": (C.c_int32, [_P, C.c_int64]),
    "svs_index_staging_finish": (C.c_int32, [_P]),
    "svs_index_set_coalesce": (C.c_int32, [_P, C.c_int32]),
    "svs_index_coalesce_stats": (C.c_int32, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "svs_index_coalesce_sizes": (C.c_int32, [_P, C.POINTER(C.c_int64), C.c_int32]),
    "svs_multi_create": (C.c_int32, [_P, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(_P)]),
    "svs_multi_search": (C.c_int32, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.POINTER(C.c_int32)]),
    "svs_multi_retain": (C.c_int32, [_P]),
    "svs_multi_release": (C.c_int32, [_P]),
    "svs_multi_info": (C.c_int32, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "svs_multi_set_coalesce": (C.c_int32, [_P, C.c_int32]),
    "svs_multi_coalesce_stats": (C.c_int32, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "svs_multi_shard": (C.c_int32, [_P, C.c_int32, C.POINTER(_P)]),
    "svs_index_mask_rows": (C.c_int32, [_P, _P, C.c_int64]),
    "svs_index_retain": (C.c_int32, [_P]),
    "svs_index_release": (C.c_int32, [_P]),
    "svs_index_info": (C.c_int32, [_P, C.POINTER(IndexInfo)]),
    "svs_index_search": (C.c_int32, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.POINTER(C.c_int32)]),
    "svs_index_search_device": (C.c_int32, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.POINTER(C.c_int32), _P]),
    "svs_index_scores_n": (C.c_int32, [_P, _P, C.c_int32, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "svs_index_top_pairs": (C.c_int32, [_P, C.c_int32, _P, _P, _P, C.POINTER(C.c_int32)]),
    "svs_index_debug_dequant": (C.c_int32, [_P, C.c_int64, C.c_int64, _P]),
    "svs_index_debug_query": (C.c_int32, [_P, _P, C.c_int32, _P]),
    "svs_index_set_timing": (C.c_int32, [_P, C.c_int32]),
    "svs_index_get_timing": (C.c_int32, [_P, C.POINTER(Timing)]),
    "svs_index_set_variant": (C.c_int32, [_P, C.c_int32]),
}

# svs_amd/csrc/internal.h: hooks for this repo's own tests, tools and bench (not part of the boundary)
INTERNAL = {
    "svs_internal_coalesce_hold": (C.c_int32, [_P, C.c_int32]),
    "svs_internal_tune": (C.c_int32, [C.c_int32, C.c_int64]),
    "svs_internal_host_phases": (C.c_int32, [C.POINTER(C.c_double), C.c_int32]),
}

_lib: Optional[C.CDLL] = None
_quick: Optional[C.PyDLL] = None

# Entry points that neither block nor take a lock (a few loads, or one atomic add): bound a second time through
# PyDLL, which keeps the GIL across the call.  CDLL drops and re-takes it around EVERY call, and with many Python
# threads inside retrieve() each re-take is a hand-off (futex wake + context switch, 5-10 us in which no thread
# runs Python): four of them per search -- info, retain, search, release -- against one (the search itself).
QUICK = ("svs_index_info", "svs_index_retain", "svs_index_release", "svs_last_error", "svs_multi_info")


def lib_path() -> str:
    return _LIB_PATH


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own
    libamdhip64 (SONAME libamdhip64.so.7, looked up as plain "libamdhip64.so"); if
    this library pulled in the system copy first, a later ``import torch`` would
    load a SECOND runtime and fail with "No HIP GPUs are available", and stream
    handles could not be shared.  So when such a wheel is installed and torch is
    not loaded yet, its runtime is loaded first and libsvs_amd binds to it.
    Opt out with SVS_AMD_SYSTEM_HIP=1."""
    import sys
    if os.environ.get("SVS_AMD_SYSTEM_HIP") or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001 -- fall back to the system runtime
        pass


def load() -> C.CDLL:
    """Loads the HIP library (once).  Raises RuntimeError when it is missing --
    there is deliberately nothing to fall back to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"svs_amd: HIP library not built ({_LIB_PATH}); run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C svs_amd/csrc`. There is no CPU fallback."
        )
    _share_torch_hip_runtime()
    lib = C.CDLL(_LIB_PATH)  # CDLL releases the GIL around every call
    for name, (res, args) in list(SIGNATURES.items()) + list(INTERNAL.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def quick():
    """The same library bound through PyDLL, for the entry points in QUICK (see there).  svs_index_release through
    this binding is for a reference taken around ONE call (the owner still holds its own, so it is not the last and
    frees nothing); an owner's release -- which may free gigabytes of HBM -- goes through load()."""
    global _quick
    if _quick is not None:
        return _quick
    load()
    # (SVS_AMD_QUICK=0: the GIL-releasing binding for these too -- the A/B of tools/coalesce_bench.py)
    ql = C.PyDLL(_LIB_PATH) if os.environ.get("SVS_AMD_QUICK", "1") != "0" else C.CDLL(_LIB_PATH)
    for name in QUICK:
        fn = getattr(ql, name)
        fn.restype, fn.argtypes = SIGNATURES[name]
    _quick = ql
    return ql


def last_error() -> str:
    return (quick().svs_last_error() or b"").decode("utf-8", "replace")   # (thread-local in the library: same thread, no hand-off)


def check(rc: int) -> None:
    """Status -> the exception the reference raises at the same spot
    (include/svs_amd.h, 'Conventions')."""
    if rc == SVS_OK:
        return
    msg = last_error()
    if rc in (SVS_ERR_SHAPE, SVS_ERR_INVALID):
        raise ValueError(msg)
    if rc == SVS_ERR_NOMEM:
        raise MemoryError(msg)
    if rc == SVS_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


def device_count() -> int:
    return int(load().svs_device_count())


def device_memory(device: int = 0):
    """(free, total) HBM bytes."""
    f, t = C.c_int64(0), C.c_int64(0)
    check(load().svs_device_memory(int(device), C.byref(f), C.byref(t)))
    return f.value, t.value
