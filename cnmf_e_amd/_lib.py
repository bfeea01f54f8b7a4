"""ctypes binding of the C ABI declared in include/cnmfe.h.

There is no CPU fallback: if the HIP shared library is missing or does not load,
importing this module raises.  Build it with ``python -m cnmf_e_amd.build``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CNMFE_LIB: another build of the same ABI (A/B runs of kernel variants, scripts/build_variant.py); the default is the in-tree library
LIB_PATH = os.environ.get("CNMFE_LIB") or os.path.join(_HERE, "libcnmfe_hip.so")


class _Lib:
    """The shared library, loaded and bound at first attribute access (Engine.__init__ is the first user).  Importing the package on a
    checkout without the .so therefore works for the host logic (sources2d on a test double); anything that touches the engine raises --
    there is still no CPU fallback."""
    _dll = None

    def _load(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "cnmf_e_amd: %s not found -- the HIP engine is mandatory (no CPU fallback). "
                "Run `python -m cnmf_e_amd.build` (needs hipcc)." % LIB_PATH)
        dll = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(dll, name)          # AttributeError here == missing export
            fn.restype = res
            fn.argtypes = args
        _Lib._dll = dll
        return dll

    def __getattr__(self, name):
        return getattr(_Lib._dll or self._load(), name)


lib = _Lib()

c_ctx = C.c_void_p
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
u8p = C.POINTER(C.c_uint8)

class DeconvOpts(C.Structure):
    """struct cnmfe_deconv_opts (include/cnmfe.h)"""
    _fields_ = [("type", C.c_int32), ("method", C.c_int32), ("smin", C.c_double), ("lambda_", C.c_double),
                ("max_tau", C.c_double), ("optimize_b", C.c_int32), ("optimize_pars", C.c_int32), ("maxIter", C.c_int32)]


# every symbol include/cnmfe.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "cnmfe_last_error": (C.c_char_p, []),
    "cnmfe_version": (C.c_char_p, []),
    "cnmfe_create": (c_ctx, [C.c_int]),
    "cnmfe_destroy": (None, [c_ctx]),
    "cnmfe_patch_create": (C.c_int, [c_ctx, C.c_int, i32p, i32p, C.c_int32, C.c_int32, C.c_int64]),
    "cnmfe_upload_block": (C.c_int, [c_ctx, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64]),
    "cnmfe_get_ymean": (C.c_int, [c_ctx, C.c_int, f64p]),
    "cnmfe_ring_init": (C.c_int, [c_ctx, C.c_int, C.c_int32, C.c_int32]),
    "cnmfe_ring_solve_stats": (C.c_int, [c_ctx, C.c_int, i64p]),
    "cnmfe_fit_reserve": (C.c_int, [c_ctx, C.c_int]),
    "cnmfe_ring_nnz": (C.c_int, [c_ctx, C.c_int, i64p, i32p]),
    "cnmfe_ring_get_csr": (C.c_int, [c_ctx, C.c_int, i64p, i32p, f32p]),
    "cnmfe_ring_set_values": (C.c_int, [c_ctx, C.c_int, f32p]),
    "cnmfe_ring_first_run": (C.c_int, [c_ctx, C.c_int, C.POINTER(C.c_int)]),
    "cnmfe_b0_get": (C.c_int, [c_ctx, C.c_int, f32p]),
    "cnmfe_b0_set": (C.c_int, [c_ctx, C.c_int, f32p]),
    "cnmfe_fit_ring_model": (C.c_int, [c_ctx, C.c_int, C.c_int32, i64p, i32p, f32p, f32p, C.c_int,
                                       C.c_double, C.c_int, f32p, i64p]),
    "cnmfe_background_ssub": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_int32, C.c_int32, i64p, i32p, f32p, f32p, C.c_int, f32p]),
    "cnmfe_reconstruct_background_ssub": (C.c_int, [c_ctx, C.c_int, f32p, C.c_int64, C.c_int64, f32p, C.c_int]),
    "cnmfe_compute_rss_ssub": (C.c_int, [c_ctx, C.c_int, C.c_int32, i64p, i32p, f32p, f32p, C.c_int, f32p, C.POINTER(C.c_double)]),
    "cnmfe_estimate_noise": (C.c_int, [c_ctx, C.c_int, C.c_int64, f32p]),
    "cnmfe_stitch_finish_async": (C.c_int, [c_ctx, C.c_int, f32p]),
    "cnmfe_update_spatial_fetch": (C.c_int, [c_ctx, f32p, C.c_int64]),
    "cnmfe_update_spatial_fetch_connected": (C.c_int, [c_ctx, C.c_int32, C.c_int32, C.c_int32, i64p, i32p, f32p, u8p]),
    "cnmfe_stitch_wait": (C.c_int, [c_ctx]),
    "cnmfe_host_alloc": (C.c_void_p, [C.c_size_t]),
    "cnmfe_host_free": (None, [C.c_void_p]),
    "cnmfe_csc_drop_zeros": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "cnmfe_update_spatial_fetch_async": (C.c_int, [c_ctx, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "cnmfe_ticket_wait": (C.c_int, [c_ctx, C.c_int64]),
    "cnmfe_update_spatial_fetch_connected_async": (C.c_int, [c_ctx, C.c_int32, C.c_int32, C.c_int32, i64p, i32p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "cnmfe_hals_temporal_job": (C.c_int, [c_ctx, C.c_int, C.c_int32, i64p, i32p, f32p, f32p, C.c_int, C.c_int32, C.POINTER(DeconvOpts), f32p, i32p]),
    "cnmfe_temporal_jobs_sweep": (C.c_int, [c_ctx]),
    "cnmfe_stitch_add_job": (C.c_int, [c_ctx, C.c_int32, C.c_int32, i32p]),
    "cnmfe_copy_generation": (C.c_int, [c_ctx, C.POINTER(C.c_int64)]),
    "cnmfe_copy_wait": (C.c_int, [c_ctx, C.c_int64]),
    "cnmfe_csc_select_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "cnmfe_csc_select_block_patch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "cnmfe_csc_bbox": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "cnmfe_set_noise": (C.c_int, [c_ctx, C.c_int, f32p]),
    "cnmfe_patch_derive": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_int32, C.c_int]),
    "cnmfe_fit_ring_model_ssub": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_int32, i64p, i32p, f32p, f32p, C.c_int,
                                            C.c_double, C.c_int, i64p]),
    "cnmfe_residual_ssub": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_int32, C.c_int32, i64p, i32p, f32p, f32p, C.c_int, C.c_void_p, C.c_int]),
    "cnmfe_residual": (C.c_int, [c_ctx, C.c_int, C.c_int32, i64p, i32p, f32p, f32p, C.c_int, C.c_void_p, C.c_int]),
    "cnmfe_get_sn": (C.c_int, [c_ctx, C.c_int, f32p]),
    "cnmfe_traces_bind": (C.c_int, [c_ctx, C.c_int32, C.c_int64, f32p, C.c_int]),
    "cnmfe_update_spatial": (C.c_int, [c_ctx, C.c_int, C.c_int, C.c_int32, i64p, i32p, f32p, f32p, C.c_int,
                                       i64p, i32p, f32p, C.c_int32, f32p]),
    "cnmfe_fast_temporal": (C.c_int, [c_ctx, C.c_int, C.c_int32, i64p, i32p, f32p, C.c_int, f32p, f32p]),
    "cnmfe_reconstruct_background": (C.c_int, [c_ctx, C.c_int, f32p, f32p, C.c_int64, C.c_int64, f32p, C.c_int]),
    "cnmfe_compute_rss": (C.c_int, [c_ctx, C.c_int, C.c_int32, i64p, i32p, f32p, f32p, C.c_int, f32p, f32p, C.POINTER(C.c_double)]),
    "cnmfe_hals_temporal": (C.c_int, [c_ctx, C.c_int, C.c_int32, i64p, i32p, f32p, f32p, C.c_int, C.c_int32,
                                      f32p, f32p, f32p]),
    "cnmfe_hals_temporal_deconv": (C.c_int, [c_ctx, C.c_int, C.c_int32, i64p, i32p, f32p, f32p, C.c_int, C.c_int32,
                                             C.POINTER(DeconvOpts), f32p, f32p, f32p, f32p, f32p, f32p]),
    "cnmfe_deconv_temporal": (C.c_int, [c_ctx, C.c_int32, C.c_int64, f32p, C.c_int, C.POINTER(DeconvOpts), f32p, f32p, f32p, f32p]),
    "cnmfe_deconv_temporal_bound": (C.c_int, [c_ctx, C.POINTER(DeconvOpts), f32p, f32p, f32p, f32p, f32p]),
    "cnmfe_post_process_spatial": (C.c_int, [c_ctx, C.c_int32, C.c_int32, C.c_int32, i64p, i32p, f32p, u8p]),
    "cnmfe_stitch_begin": (C.c_int, [c_ctx, C.c_int32, C.c_int64]),
    "cnmfe_stitch_dims": (C.c_int, [c_ctx, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "cnmfe_stitch_add": (C.c_int, [c_ctx, C.c_int32, i32p]),
    "cnmfe_stitch_buffer": (C.c_int, [c_ctx, C.POINTER(f32p), i64p]),
    "cnmfe_stitch_buffer_stream": (C.c_int, [c_ctx, C.POINTER(f32p), i64p, C.POINTER(C.c_void_p)]),
    "cnmfe_footprint_moments": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 10),
    "cnmfe_search_ellipse": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32,
                                       C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cnmfe_csc_from_triplets": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cnmfe_stitch_finish": (C.c_int, [c_ctx, C.c_int, f32p, C.c_int]),
    "cnmfe_stitch_temporal": (C.c_int, [C.POINTER(c_ctx), C.c_int, C.c_int, f32p, C.c_int]),
    "cnmfe_profile_enable": (C.c_int, [c_ctx, C.c_int]),
    "cnmfe_profile_reset": (C.c_int, [c_ctx]),
    "cnmfe_profile_count": (C.c_int, [c_ctx]),
    "cnmfe_profile_get": (C.c_int, [c_ctx, C.c_int, C.c_char_p, C.c_int, f64p, i64p]),
    "cnmfe_synchronize": (C.c_int, [c_ctx]),
    "cnmfe_set_option": (C.c_int, [c_ctx, C.c_char_p, C.c_int64]),
}

# enums of include/cnmfe.h
F32, F64, U16, U8, F16 = 0, 1, 2, 3, 4
HOST, DEVICE = 0, 1
COLMAJOR, ROWMAJOR, BOUND, BOUND_ROWS = 0, 1, 2, 3
SPATIAL_HALS, SPATIAL_HALS_THRESH, SPATIAL_NNLS = 0, 1, 2


class CnmfeError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != 0:
        raise CnmfeError("cnmfe error %d: %s" % (rc, lib.cnmfe_last_error().decode("utf-8", "replace")))
