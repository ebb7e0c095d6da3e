"""ctypes binding of libdeff_amd.so (include/deff_amd.h).

There is no CPU fallback: if the HIP library is missing this module raises at
load time, and every entry point raises DeffError on a non-zero return code.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# DEFF_AMD_LIB: another build of the same library (kernel experiments, A/B runs on one box)
LIB_PATH = os.environ.get("DEFF_AMD_LIB") or os.path.join(_PKG, "libdeff_amd.so")

KERNEL_AUTO, KERNEL_EXPLICIT, KERNEL_SCALAR, KERNEL_MATFREE, KERNEL_MATFREE_TB = range(5)
KERNEL_NAMES = {"auto": 0, "explicit": 1, "scalar": 2, "matfree": 3, "matfree_tb": 4}

# every symbol include/deff_amd.h declares (tests check the library exports them all)
SYMBOLS = [
    "deff_version", "deff_last_error", "deff_error_string", "deff_device_count",
    "deff_create", "deff_create_batch", "deff_batch_size", "deff_recommended_batch", "deff_destroy", "deff_mesh", "deff_set_kernel", "deff_get_kernel",
    "deff_set_tuning", "deff_get_plan", "deff_set_image", "deff_synth_image", "deff_get_image",
    "deff_load_jpeg_gray", "deff_free", "deff_assemble_2phase", "deff_assemble_3phase", "deff_flood_fill", "deff_assemble_from_D", "deff_set_system", "deff_get_system",
    "deff_init_linear", "deff_set_field", "deff_get_field", "deff_solve", "deff_solve_batch", "deff_sweeps",
    "deff_slab_group_create", "deff_slab_group_destroy", "deff_slab_group_layout", "deff_slab_group_set_tuning", "deff_slab_group_get_plan",
    "deff_slab_group_set_image", "deff_slab_group_synth_image", "deff_slab_group_assemble_2phase",
    "deff_slab_group_init_linear", "deff_slab_group_set_field", "deff_slab_group_get_field",
    "deff_slab_group_sweeps", "deff_slab_group_flux", "deff_slab_group_solve",
    "deff_rccl_unique_id", "deff_slab_rank_create", "deff_slab_rank_create_custom", "deff_slab_rank_destroy", "deff_slab_rank_layout",
    "deff_slab_rank_window", "deff_slab_rank_context", "deff_slab_rank_set_tuning", "deff_slab_rank_set_image_window",
    "deff_slab_rank_synth_image", "deff_slab_rank_assemble_3phase", "deff_slab_group_assemble_3phase", "deff_slab_rank_get_field", "deff_slab_rank_sweeps", "deff_slab_rank_solve",
    "deff_solve_stream", "deff_get_slot_field", "deff_debug_tb_stamps", "deff_flux", "deff_residual", "deff_residual_slot", "deff_residual_D", "deff_set_progress", "deff_last_launches", "deff_device_field", "deff_synchronize",
]


class DeffError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"deff_amd error {code}: {msg}")
        self.code = code


class Result(C.Structure):
    _fields_ = [("iters", C.c_int64), ("checks", C.c_int64), ("deff_raw", C.c_double),
                ("conv", C.c_double), ("loop_ms", C.c_double)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_int64, C.c_double, C.c_double, C.c_void_p)
NEXT_IMAGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_int64))
IMAGE_DONE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int, C.POINTER(Result))
_cdp = C.POINTER(C.c_double)
HOST_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _cdp, _cdp, _cdp, _cdp, C.c_size_t)
HOST_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _cdp, _cdp, C.c_size_t)
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_lib = None


def load():
    """Load libdeff_amd.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C effectivediffusivityfvm_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    ctx = C.c_void_p
    L.deff_version.restype = C.c_char_p
    L.deff_last_error.restype = C.c_char_p
    L.deff_error_string.restype = C.c_char_p
    L.deff_error_string.argtypes = [C.c_int]
    L.deff_device_count.argtypes = [C.POINTER(C.c_int)]
    L.deff_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(ctx)]
    L.deff_create_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ctx)]
    L.deff_batch_size.argtypes = [ctx, C.POINTER(C.c_int)]
    L.deff_recommended_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int)]
    L.deff_destroy.argtypes = [ctx]
    L.deff_mesh.argtypes = [ctx, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                            C.POINTER(C.c_double)]
    L.deff_set_kernel.argtypes = [ctx, C.c_int]
    L.deff_get_kernel.argtypes = [ctx, C.POINTER(C.c_int)]
    L.deff_set_tuning.argtypes = [ctx, C.c_char_p, C.c_int]
    L.deff_get_plan.argtypes = [ctx, C.c_char_p, C.POINTER(C.c_int)]
    L.deff_set_image.argtypes = [ctx, _u8p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.deff_synth_image.argtypes = [ctx, C.c_uint64, C.c_uint64]
    L.deff_get_image.argtypes = [ctx, _u8p]
    L.deff_load_jpeg_gray.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]
    L.deff_free.argtypes = [C.c_void_p]
    L.deff_free.restype = None
    L.deff_assemble_2phase.argtypes = [ctx, C.c_double, C.c_double, C.c_double, C.c_double]
    L.deff_assemble_3phase.argtypes = [ctx, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_double,
                                       C.c_double]
    L.deff_flood_fill.argtypes = [np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS"), C.c_int, C.c_int,
                                  C.POINTER(C.c_int)]
    L.deff_assemble_from_D.argtypes = [ctx, _dp, C.c_void_p, C.c_double, C.c_double]
    L.deff_set_system.argtypes = [ctx, _dp, _dp, C.c_void_p, C.c_double, C.c_double]
    L.deff_get_system.argtypes = [ctx, _dp, _dp]
    L.deff_init_linear.argtypes = [ctx, C.c_double, C.c_double]
    L.deff_set_field.argtypes = [ctx, _dp]
    L.deff_get_field.argtypes = [ctx, _dp]
    L.deff_solve.argtypes = [ctx, C.c_double, C.c_double, C.c_int64, C.c_int64, C.POINTER(Result),
                             C.c_void_p, C.c_void_p]
    L.deff_solve_batch.argtypes = [ctx, C.c_double, C.c_double, C.c_int64, C.c_int64, C.POINTER(Result),
                                   C.c_void_p, C.c_void_p]
    L.deff_sweeps.argtypes = [ctx, C.c_int64, C.c_double, C.POINTER(C.c_float)]
    L.deff_flux.argtypes = [ctx, C.POINTER(C.c_double), C.c_void_p, C.c_void_p]
    L.deff_residual.argtypes = [ctx, C.POINTER(C.c_double), C.POINTER(C.c_float)]
    L.deff_residual_slot.argtypes = [ctx, C.c_int, C.POINTER(C.c_double)]
    L.deff_residual_D.argtypes = [ctx, _dp, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_float)]
    L.deff_set_progress.argtypes = [ctx, PROGRESS_FN, C.c_void_p]
    L.deff_last_launches.argtypes = [ctx, C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    ip = C.POINTER(C.c_int)
    L.deff_slab_group_create.argtypes = [C.c_int, ip, C.c_int, C.c_int, C.POINTER(ctx)]
    L.deff_slab_group_destroy.argtypes = [ctx]
    L.deff_slab_group_layout.argtypes = [ctx, ip, ip]
    L.deff_slab_group_set_tuning.argtypes = [ctx, C.c_char_p, C.c_int]
    L.deff_slab_rank_set_tuning.argtypes = [ctx, C.c_char_p, C.c_int]
    L.deff_slab_group_get_plan.argtypes = [ctx, C.c_int, C.c_char_p, C.POINTER(C.c_int)]
    L.deff_slab_group_set_image.argtypes = [ctx, _u8p]
    L.deff_slab_group_synth_image.argtypes = [ctx, C.c_uint64, C.c_uint64]
    L.deff_slab_group_assemble_2phase.argtypes = [ctx, C.c_double, C.c_double, C.c_double, C.c_double]
    L.deff_slab_group_init_linear.argtypes = [ctx, C.c_double, C.c_double]
    L.deff_slab_group_set_field.argtypes = [ctx, _dp]
    L.deff_slab_group_get_field.argtypes = [ctx, _dp]
    L.deff_slab_group_sweeps.argtypes = [ctx, C.c_int64, C.c_double, C.POINTER(C.c_float)]
    L.deff_slab_group_flux.argtypes = [ctx, C.POINTER(C.c_double), C.c_void_p, C.c_void_p]
    L.deff_slab_group_solve.argtypes = [ctx, C.c_double, C.c_double, C.c_int64, C.c_int64, C.POINTER(Result),
                                        C.c_void_p, C.c_void_p]
    L.deff_rccl_unique_id.argtypes = [C.c_char_p]
    L.deff_slab_rank_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(ctx)]
    L.deff_slab_rank_create_custom.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, HOST_EXCHANGE_FN,
                                               HOST_ALLGATHER_FN, C.c_void_p, C.POINTER(ctx)]
    L.deff_slab_rank_destroy.argtypes = [ctx]
    L.deff_slab_rank_layout.argtypes = [ctx, ip, ip]
    L.deff_slab_rank_window.argtypes = [ctx, ip, ip]
    L.deff_slab_rank_context.argtypes = [ctx, C.POINTER(ctx)]
    L.deff_slab_rank_set_image_window.argtypes = [ctx, _u8p]
    L.deff_slab_rank_synth_image.argtypes = [ctx, C.c_uint64, C.c_uint64]
    L.deff_slab_rank_assemble_3phase.argtypes = [ctx, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_double, C.c_double]
    L.deff_slab_group_assemble_3phase.argtypes = [ctx, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_double, C.c_double]
    L.deff_slab_rank_get_field.argtypes = [ctx, _dp]
    L.deff_slab_rank_sweeps.argtypes = [ctx, C.c_int64, C.c_double, C.POINTER(C.c_float)]
    L.deff_slab_rank_solve.argtypes = [ctx, C.c_double, C.c_double, C.c_int64, C.c_int64, C.POINTER(Result),
                                       C.c_void_p, C.c_void_p]
    L.deff_solve_stream.argtypes = [ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                    C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int64, NEXT_IMAGE_FN,
                                    IMAGE_DONE_FN, C.c_void_p]
    L.deff_get_slot_field.argtypes = [ctx, C.c_int, _dp]
    L.deff_debug_tb_stamps.argtypes = [ctx, C.c_double, C.c_void_p, C.POINTER(C.c_int)]
    L.deff_device_field.argtypes = [ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.deff_synchronize.argtypes = [ctx]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise DeffError(rc, load().deff_last_error().decode("utf-8", "replace"))
