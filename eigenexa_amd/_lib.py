"""ctypes binding of libeigenexa_amd.so (the C-ABI declared in include/eigenexa_amd.h).

There is no CPU fallback: if the HIP library is missing or no GPU is visible the calls fail loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EIGX_LIB: alternative build of the same library (tools/: the diagnostic build with in-kernel stamps)
LIB_PATH = os.environ.get("EIGX_LIB") or os.path.join(_HERE, "lib", "libeigenexa_amd.so")

_c_double_p = C.POINTER(C.c_double)
_c_int_p = C.POINTER(C.c_int)

# name -> (restype, argtypes); mirrors include/eigenexa_amd.h one to one
SIGNATURES = {
    "eigx_init": (C.c_int, [C.c_int]),
    "eigx_init_multi": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_char]),
    "eigx_get_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "eigx_get_device_count": (C.c_int, []),
    "eigx_get_comm": (C.c_int, [C.POINTER(C.c_int)] * 4),
    "eigx_comm_seconds": (C.c_double, []),
    "eigx_comm_info": (C.c_int, [C.c_char_p, C.c_int]),
    "eigx_rccl_selftest": (C.c_int, []),
    "eigx_free": (C.c_int, []),
    "eigx_get_version": (C.c_int, [_c_int_p, C.c_char_p, C.c_char_p]),
    "eigx_get_procs": (C.c_int, [_c_int_p, _c_int_p, _c_int_p]),
    "eigx_get_id": (C.c_int, [_c_int_p, _c_int_p, _c_int_p]),
    "eigx_get_errinfo": (C.c_int, [C.POINTER(C.c_int64)]),
    "eigx_get_matdims": (C.c_int, [C.c_int, _c_int_p, _c_int_p, C.c_int, C.c_int, C.c_char]),
    "eigx_matdims_for_grid": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char, _c_int_p, _c_int_p]),
    "eigx_held_bytes": (C.c_int64, []),
    "eigx_held_bytes_named": (C.c_int64, [C.c_char_p]),
    "eigx_transpose_plan": (C.c_int, [C.c_int] * 6 + [C.POINTER(C.c_int)] * 5),
    "eigx_memory_internal": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "eigx_loop_start": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "eigx_loop_end": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "eigx_translate_l2g": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "eigx_translate_g2l": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "eigx_owner_node": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "eigx_owner_index": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "eigx_sx": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                          C.c_int, C.c_int, C.c_char]),
    "eigx_s": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                         C.c_int, C.c_int, C.c_char]),
    "eigx_sx_dev": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                              C.c_int, C.c_int, C.c_char]),
    "eigx_s_dev": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                             C.c_int, C.c_int, C.c_char]),
    "eigx_solve_bc": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_char]),
    "eigx_solve_bc_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_char]),
    "eigx_numroc": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "eigx_h": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                         C.c_int, C.c_int, C.c_char]),
    "eigx_h_dev": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                             C.c_int, C.c_int, C.c_char]),
    "eigx_set_grid_dims": (C.c_int, [C.c_int, C.c_int]),
    "eigx_band_reduce_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int]),
    "eigx_band_dc_dev": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_int]),
    "eigx_gev": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "eigx_gev_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "eigx_band_bisect_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "eigx_trbak_dev": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "eigx_dgemm_dev": (C.c_int, [C.c_char, C.c_char, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p,
                                 C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int]),
    "eigx_dgemm_gather_dev": (C.c_int, [C.c_char, C.c_char, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p]),
    "eigx_get_timers": (C.c_int, [_c_double_p]),
    "eigx_profile": (C.c_int, [C.c_int]),
    "eigx_profile_read": (C.c_int, [_c_double_p]),
    "eigx_profile_read_kinds": (C.c_int, [_c_double_p, C.c_int]),
    "eigx_tune": (C.c_int, [C.c_int, C.c_int]),
    "eigx_device_synchronize": (C.c_int, []),
    "eigx_malloc_dev": (C.c_void_p, [C.c_int64]),
    "eigx_free_dev": (C.c_int, [C.c_void_p]),
    "eigx_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "eigx_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
}

_lib = None


def load():
    """Load the shared library (building is __graft_entry__.build()'s job)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with tools/build_lib.sh (hipcc --offload-arch=gfx950). "
            "eigenexa_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")
