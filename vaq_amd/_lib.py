"""ctypes binding of include/vaqhip.h (vaq_amd/lib/libvaqhip.so).

The library is loaded lazily and the load FAILS LOUDLY when the shared object
is missing: there is no Python/NumPy fallback for any entry point.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("VAQHIP_LIB") or os.path.join(_HERE, "lib", "libvaqhip.so")

# every symbol include/vaqhip.h declares
SYMBOLS = [
    "vaqhip_index_create", "vaqhip_index_create_ex", "vaqhip_index_destroy", "vaqhip_index_set_codes_u16",
    "vaqhip_index_set_codes_u16_device", "vaqhip_index_add_codes_u16", "vaqhip_index_add_codes_u16_device",
    "vaqhip_index_set_ti_clusters", "vaqhip_index_set_method",
    "vaqhip_search", "vaqhip_search_projected",
    "vaqhip_search_device", "vaqhip_search_staged_supported", "vaqhip_search_begin_device",
    "vaqhip_search_finish_device", "vaqhip_build_lut", "vaqhip_project", "vaqhip_merge_topk_device",
    "vaqhip_merge_topk_strided_device",
    "vaqhip_encode", "vaqhip_encode_device", "vaqhip_refine", "vaqhip_refine_device",
    "vaqhip_index_info", "vaqhip_set_option", "vaqhip_last_timing", "vaqhip_last_error",
    "vaqhip_version", "vaqhip_device_count",
    "vaqhip_multi_create", "vaqhip_multi_destroy", "vaqhip_multi_set_codes_u16", "vaqhip_multi_add_codes_u16",
    "vaqhip_multi_search", "vaqhip_multi_search_device", "vaqhip_multi_set_ti_clusters", "vaqhip_multi_set_method", "vaqhip_multi_set_option",
    "vaqhip_multi_get_info", "vaqhip_multi_shard", "vaqhip_multi_last_error",
]
MAX_DEVICES = 16

ERROR_NAMES = {
    -1: "EINVAL", -2: "EUNSUPPORTED", -3: "ENODEVICE", -4: "ENOMEM", -5: "EHIP",
    -6: "ERANGE", -7: "ESTATE",
}


class VaqHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"vaqhip error {code} ({ERROR_NAMES.get(code, '?')}): {msg}")
        self.code = code


class Info(C.Structure):
    _fields_ = [("D", C.c_int), ("M", C.c_int), ("L", C.c_int), ("max_bits", C.c_int),
                ("total_bits", C.c_int), ("code_bytes", C.c_int), ("algo_code_bytes", C.c_int),
                ("lut_floats", C.c_int), ("N", C.c_int64), ("id_base", C.c_int64),
                ("device_id", C.c_int), ("layout", C.c_int), ("ti_clusters", C.c_int),
                ("ti_segments", C.c_int), ("methods", C.c_uint), ("visit", C.c_float)]


class Timing(C.Structure):
    _fields_ = [("project_ms", C.c_float), ("lut_ms", C.c_float), ("seed_ms", C.c_float),
                ("scan_ms", C.c_float),
                ("merge_ms", C.c_float), ("n_searches", C.c_int), ("queries_per_pass", C.c_int), ("slices", C.c_int),
                ("workgroups", C.c_int), ("passes", C.c_int), ("lds_bytes", C.c_int),
                ("seed_slices", C.c_int), ("early_abandon", C.c_int),
                ("best_first", C.c_int), ("deferred_queries", C.c_int),
                ("bucket_major", C.c_int)]


class MultiInfo(C.Structure):
    _fields_ = [("n_devices", C.c_int), ("exchange", C.c_int), ("N", C.c_int64), ("id_base", C.c_int64),
                ("device_ids", C.c_int * 16), ("shard_rows", C.c_int64 * 16), ("last_search_ms", C.c_float),
                ("last_exchange_ms", C.c_float), ("last_merge_ms", C.c_float)]


_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Return the loaded C-ABI library; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: build it with `python -m vaq_amd.build` "
            "(hipcc, gfx950). vaq_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7 /
    # libhsa-runtime64 under torch/lib, and a second copy (the system one this
    # library's RUNPATH points at) cannot open the GPU once the first has.  Both
    # carry the soname libamdhip64.so.7, so importing torch first makes the
    # dynamic linker bind libvaqhip.so to the runtime torch already loaded.
    # Without torch (a plain C/C++ host) the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.vaqhip_index_create.argtypes = [C.POINTER(vp), i32, i32, C.POINTER(i32),
                                      C.POINTER(C.POINTER(C.c_float)), vp, i32]
    L.vaqhip_index_create_ex.argtypes = [C.POINTER(vp), i32, i32, C.POINTER(i32),
                                         C.POINTER(C.POINTER(C.c_float)), vp, i32, C.c_uint]
    L.vaqhip_index_destroy.argtypes = [vp]
    L.vaqhip_index_destroy.restype = None
    L.vaqhip_index_set_codes_u16.argtypes = [vp, vp, i64, i64]
    L.vaqhip_index_set_codes_u16_device.argtypes = [vp, vp, i64, i64, vp]
    L.vaqhip_index_add_codes_u16.argtypes = [vp, vp, i64]
    L.vaqhip_index_add_codes_u16_device.argtypes = [vp, vp, i64, vp]
    L.vaqhip_index_set_ti_clusters.argtypes = [vp, vp, i32, i32]
    L.vaqhip_index_set_method.argtypes = [vp, C.c_uint, C.c_float]
    L.vaqhip_search.argtypes = [vp, vp, i32, i32, vp, vp]
    L.vaqhip_search_projected.argtypes = [vp, vp, i32, i32, vp, vp]
    L.vaqhip_search_device.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    L.vaqhip_search_staged_supported.argtypes = [vp, i32, i32]
    L.vaqhip_search_begin_device.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.vaqhip_search_finish_device.argtypes = [vp, vp, vp]
    L.vaqhip_build_lut.argtypes = [vp, vp, i32, i32, vp]
    L.vaqhip_project.argtypes = [vp, vp, i64, vp]
    L.vaqhip_merge_topk_device.argtypes = [i32, vp, vp, i32, i32, i32, vp, vp, vp]
    L.vaqhip_encode.argtypes = [vp, vp, i64, i32, vp]
    L.vaqhip_encode_device.argtypes = [vp, vp, i64, i32, vp, vp]
    L.vaqhip_refine.argtypes = [i32, vp, i32, i32, vp, i64, vp, i32, i32, vp, vp]
    L.vaqhip_refine_device.argtypes = [i32, vp, i32, i32, vp, vp, i32, i32, vp, vp, vp]
    L.vaqhip_merge_topk_strided_device.argtypes = [i32, vp, vp, i32, i64, i64, i32, i32, vp, vp, vp]
    L.vaqhip_index_info.argtypes = [vp, C.POINTER(Info)]
    L.vaqhip_set_option.argtypes = [vp, C.c_char_p, i64]
    L.vaqhip_last_timing.argtypes = [vp, C.POINTER(Timing)]
    L.vaqhip_last_error.restype = C.c_char_p
    L.vaqhip_multi_create.argtypes = [C.POINTER(vp), i32, i32, C.POINTER(i32), C.POINTER(C.POINTER(C.c_float)), vp,
                                      i32, C.POINTER(i32), C.c_uint]
    L.vaqhip_multi_destroy.argtypes = [vp]
    L.vaqhip_multi_destroy.restype = None
    L.vaqhip_multi_set_codes_u16.argtypes = [vp, vp, i64, i64]
    L.vaqhip_multi_add_codes_u16.argtypes = [vp, vp, i64]
    L.vaqhip_multi_search.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.vaqhip_multi_search_device.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    L.vaqhip_multi_set_ti_clusters.argtypes = [vp, vp, i32, i32]
    L.vaqhip_multi_set_method.argtypes = [vp, C.c_uint, C.c_float]
    L.vaqhip_multi_set_option.argtypes = [vp, C.c_char_p, i64]
    L.vaqhip_multi_get_info.argtypes = [vp, C.POINTER(MultiInfo)]
    L.vaqhip_multi_shard.argtypes = [vp, i32]
    L.vaqhip_multi_shard.restype = vp
    L.vaqhip_multi_last_error.restype = C.c_char_p
    for name in SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int:
            fn.restype = C.c_int
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        msg = load().vaqhip_last_error()
        raise VaqHipError(rc, msg.decode() if msg else "")
    return rc


def check_multi(rc: int) -> int:
    if rc < 0:
        msg = load().vaqhip_multi_last_error()
        raise VaqHipError(rc, msg.decode() if msg else "")
    return rc
