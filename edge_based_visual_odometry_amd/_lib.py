"""ctypes binding of the C ABI in include/ebvo_hip.h (libebvo_hip.so, built in-tree by hipcc).

There is no CPU fallback: if the shared library is missing this module raises, and if no HIP
device is usable ``ebvo_ctx_create`` fails and :class:`EbvoError` is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EBVO_LIB") or os.path.join(_HERE, "libebvo_hip.so")   # EBVO_LIB: another build of the same ABI (A/B runs)

EDGE_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("theta", "<f8"), ("index", "<i4"), ("pad", "<i4")])
assert EDGE_DTYPE.itemsize == 32

EBVO_OK = 0
EBVO_ERR_ARG = -1
EBVO_ERR_CAPACITY = -2
EBVO_ERR_HIP = -3
EBVO_ERR_NOMEM = -4
EBVO_ERR_STATE = -5

STAGE_EPIPOLAR = 1
STAGE_DISPARITY = 2
STAGE_ORIENTATION = 4
STAGE_ALL = 7

MAX_KERNELS = 24
TOED_STRICT, TOED_HYBRID = 0, 1

# every symbol include/ebvo_hip.h declares (checked by tests/test_abi_symbols.py)
ABI_SYMBOLS = (
    "ebvo_strerror", "ebvo_last_error", "ebvo_abi_version", "ebvo_ctx_create", "ebvo_ctx_destroy",
    "ebvo_set_toed_mode", "ebvo_get_toed_mode", "ebvo_toed_fallbacks", "ebvo_graph_launches", "ebvo_toed_stats", "ebvo_toed_screen_audit", "ebvo_stereo_upload_async", "ebvo_host_register", "ebvo_host_unregister", "ebvo_ingest_stats",
    "ebvo_stereo_fetch_compact_begin", "ebvo_stereo_fetch_compact_end", "ebvo_stereo_pushed_view",
    "ebvo_toed", "ebvo_toed_pair", "ebvo_epipolar_lines", "ebvo_epi_candidates", "ebvo_epi_candidates_staged", "ebvo_ncc_pairs",
    "ebvo_edge_patches", "ebvo_ncc_patches", "ebvo_ncc_quads", "ebvo_stereo_default_params", "ebvo_finalize_default_params",
    "ebvo_stereo_upload", "ebvo_stereo_run", "ebvo_stereo_fetch", "ebvo_stereo_set_slots",
    "ebvo_stereo_upload_slot", "ebvo_stereo_submit", "ebvo_stereo_wait", "ebvo_stereo_fetch_slot", "ebvo_profile_enable",
    "ebvo_profile_reset", "ebvo_profile_get", "ebvo_fp64_peak",
    "ebvo_gn_default_params", "ebvo_sobel_gradients", "ebvo_gn_refine_stereo", "ebvo_stereo_refine",
    "ebvo_stereo_fetch_refined", "ebvo_gn_refine_temporal", "ebvo_finalize_pairs", "ebvo_bnb_test", "ebvo_keep_best", "ebvo_epipolar_shift", "ebvo_cluster_rows", "ebvo_stereo_finalize", "ebvo_stereo_finalize_submit", "ebvo_stereo_finalize_wait", "ebvo_stereo_fetch_final",
    "ebvo_debug_set", "ebvo_stereo_fetch_begin", "ebvo_stereo_fetch_end",
    "ebvo_undistort", "ebvo_stereo_set_undistort", "ebvo_sift_descriptors", "ebvo_sift_min_distances",
    "ebvo_toed_resident", "ebvo_epi_candidates_resident", "ebvo_ncc_pairs_resident",
    "ebvo_temporal_default_params", "ebvo_temporal_set_keyframe", "ebvo_temporal_match", "ebvo_temporal_match_submit", "ebvo_temporal_match_wait", "ebvo_temporal_fetch", "ebvo_temporal_fetch_final",
)


class StereoParams(C.Structure):
    _fields_ = [("F21", C.c_double * 9), ("epi_thr", C.c_double), ("max_disp", C.c_double),
                ("orient_thr_deg", C.c_double), ("ncc_thr", C.c_double), ("stage_mask", C.c_int),
                ("reserved", C.c_int)]


PAIR_PACK = 8                         # ... into device staging: the compact fetch is one copy
PAIR_PUSH, PAIR_PUSH_THETA = 2, 4   # ... the chain ends by writing the compact results into page-locked host memory
PAIR_NO_SIMS = 1     # StereoParams.reserved: the resident pipeline stores best + keep only (ebvo_hip.h EBVO_PAIR_NO_SIMS)


class StereoCalib(C.Structure):
    _fields_ = [("K_left", C.c_double * 9), ("K_right", C.c_double * 9), ("R21", C.c_double * 9), ("T21", C.c_double * 3)]


class GnParams(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("tol", C.c_double), ("huber_delta", C.c_double)]


class FinalizeParams(C.Structure):
    _fields_ = [("bnb_ratio", C.c_double), ("ncc_thr", C.c_double), ("gn", GnParams), ("use_sift", C.c_int),
                ("reserved", C.c_int), ("sift_thr", C.c_double), ("bnb_sift", C.c_double)]


class FinalizeCounts(C.Structure):
    _fields_ = [("n_ncc", C.c_int32), ("n_bnb", C.c_int32), ("n_clusters", C.c_int32), ("n_ncc2", C.c_int32),
                ("n_final", C.c_int32), ("n_sift", C.c_int32)]


class StereoCounts(C.Structure):
    _fields_ = [("n_left", C.c_int32), ("n_right", C.c_int32), ("n_total_left", C.c_int32),
                ("n_total_right", C.c_int32), ("n_pairs", C.c_int64), ("n_matches", C.c_int64)]


class UndistortParams(C.Structure):
    _fields_ = [("K_left", C.c_double * 4), ("K_right", C.c_double * 4), ("dist_left", C.c_double * 5),
                ("dist_right", C.c_double * 5), ("n_dist", C.c_int), ("reserved", C.c_int)]


class TemporalParams(C.Structure):
    _fields_ = [("cell_size", C.c_int), ("stages", C.c_int), ("grid_radius", C.c_double), ("orient_thr_deg", C.c_double),
                ("ncc_thr", C.c_double), ("sift_thr", C.c_double), ("bnb_ncc", C.c_double), ("bnb_sift", C.c_double),
                ("gn", GnParams)]


class TemporalCounts(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("n_cf", C.c_int32), ("n_candidates", C.c_int64), ("n_kept", C.c_int64),
                ("n_sift", C.c_int64), ("n_bnb_ncc", C.c_int64), ("n_bnb_sift", C.c_int64), ("n_refined_valid", C.c_int64),
                ("n_final", C.c_int64)]


class StereoView(C.Structure):
    _fields_ = [("left", C.c_void_p), ("right", C.c_void_p), ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p),
                ("sims", C.c_void_p), ("best", C.c_void_p), ("keep", C.c_void_p), ("n_left", C.c_int32),
                ("n_right", C.c_int32), ("n_pairs", C.c_int64)]


class ToedView(C.Structure):
    _fields_ = [("edges", C.c_void_p), ("all4", C.c_void_p), ("n_kept", C.c_int32), ("n_total", C.c_int32),
                ("tag", C.c_uint64), ("t_conv", C.c_double), ("t_nms", C.c_double)]


class CandidatesView(C.Structure):
    _fields_ = [("row_ptr", C.c_void_p), ("col_idx", C.c_void_p), ("orient_ok", C.c_void_p), ("n_pairs", C.c_int64),
                ("row_ptr_final", C.c_void_p), ("col_idx_final", C.c_void_p), ("n_final", C.c_int64)]


class NccView(C.Structure):
    _fields_ = [("left_patches", C.c_void_p), ("sims", C.c_void_p), ("best", C.c_void_p), ("keep", C.c_void_p),
                ("n_left", C.c_int32), ("n_pairs", C.c_int64)]


NCC_WANT_LEFT_PATCHES, NCC_WANT_SIMS = 1, 2

FETCH_EDGES, FETCH_CSR, FETCH_BEST, FETCH_KEEP, FETCH_SIMS, FETCH_DEFAULT, FETCH_ALL = 1, 2, 4, 8, 16, 15, 31


class CompactView(C.Structure):
    _fields_ = [("left_xy", C.c_void_p), ("right_xy", C.c_void_p), ("left_theta", C.c_void_p), ("right_theta", C.c_void_p),
                ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p), ("best", C.c_void_p), ("keep_bits", C.c_void_p),
                ("n_left", C.c_int32), ("n_right", C.c_int32), ("n_pairs", C.c_int64), ("n_matches", C.c_int64)]


COMPACT_XY, COMPACT_THETA, COMPACT_CSR, COMPACT_BEST, COMPACT_KEEP_BITS, COMPACT_DEFAULT, COMPACT_ALL = 1, 2, 4, 8, 16, 29, 31


class ScreenAudit(C.Structure):
    _fields_ = [("n_candidates", C.c_int32), ("n_maxima", C.c_int32), ("n_kept", C.c_int32), ("n_neighbour_points", C.c_int32),
                ("max_err_gx", C.c_double), ("max_err_gy", C.c_double), ("max_err_mag", C.c_double),
                ("max_err_mag_neighbours", C.c_double), ("bound_g", C.c_double), ("bound_mag", C.c_double),
                ("bound_slope", C.c_double), ("tol_mag", C.c_double), ("tol_slope", C.c_double)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_double), ("launches", C.c_int64)]


class EbvoError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: status {status}" + (f" ({detail})" if detail else ""))


_lib = None


def load_library() -> C.CDLL:
    """dlopen libebvo_hip.so (raises if it was not built) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with __graft_entry__.build() or "
            "`make -C edge_based_visual_odometry_amd/csrc` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    ssz = C.c_ssize_t
    lib.ebvo_strerror.restype = C.c_char_p
    lib.ebvo_strerror.argtypes = [i32]
    lib.ebvo_last_error.restype = C.c_char_p
    lib.ebvo_last_error.argtypes = [vp]
    lib.ebvo_abi_version.restype = i32
    lib.ebvo_ctx_create.argtypes = [i32, i32, i32, C.POINTER(vp)]
    lib.ebvo_ctx_destroy.restype = None
    lib.ebvo_ctx_destroy.argtypes = [vp]
    lib.ebvo_set_toed_mode.argtypes = [vp, i32]
    lib.ebvo_get_toed_mode.argtypes = [vp]
    lib.ebvo_toed_fallbacks.argtypes = [vp]
    lib.ebvo_toed_fallbacks.restype = i64
    lib.ebvo_graph_launches.argtypes = [vp]
    lib.ebvo_graph_launches.restype = i64
    lib.ebvo_toed_stats.argtypes = [vp, i32, vp]
    lib.ebvo_toed_screen_audit.argtypes = [vp, vp, i32, i32, ssz, C.POINTER(ScreenAudit)]
    lib.ebvo_toed.argtypes = [vp, vp, i32, i32, ssz, vp, i32, C.POINTER(i32), C.POINTER(i32), vp, i32,
                              C.POINTER(dbl), C.POINTER(dbl)]
    lib.ebvo_toed_pair.argtypes = [vp, vp, vp, i32, i32, ssz, ssz, vp, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    lib.ebvo_epipolar_lines.argtypes = [vp, vp, i32, vp]
    lib.ebvo_epi_candidates.argtypes = [vp, vp, i32, vp, i32, vp, dbl, dbl, dbl, i32, vp, vp, i64,
                                        C.POINTER(i64)]
    lib.ebvo_epi_candidates_staged.restype = i32
    lib.ebvo_epi_candidates_staged.argtypes = [vp, vp, i32, vp, i32, vp, dbl, dbl, dbl, vp, vp, vp, i64, C.POINTER(i64)]
    lib.ebvo_ncc_pairs.argtypes = [vp, vp, vp, i32, i32, ssz, ssz, vp, i32, vp, vp, dbl, vp, vp, vp, vp]
    lib.ebvo_edge_patches.argtypes = [vp, vp, i32, i32, ssz, vp, i32, vp]
    lib.ebvo_ncc_patches.argtypes = [vp, vp, vp, i32, vp]
    lib.ebvo_ncc_quads.argtypes = [vp, vp, vp, vp, vp, i32, dbl, vp, vp, vp]
    lib.ebvo_stereo_default_params.restype = None
    lib.ebvo_finalize_default_params.restype = None
    lib.ebvo_finalize_default_params.argtypes = [C.c_void_p]
    lib.ebvo_stereo_default_params.argtypes = [C.POINTER(StereoParams)]
    lib.ebvo_stereo_upload.argtypes = [vp, vp, vp, i32, i32, ssz, ssz]
    lib.ebvo_toed_resident.argtypes = [vp, i32, vp, i32, i32, ssz, i32, C.POINTER(ToedView)]
    lib.ebvo_epi_candidates_resident.argtypes = [vp, C.c_uint64, C.c_uint64, vp, dbl, dbl, dbl, i32, i32, C.POINTER(CandidatesView)]
    lib.ebvo_ncc_pairs_resident.argtypes = [vp, C.c_uint64, C.c_uint64, vp, vp, i32, i32, ssz, ssz, vp, vp, dbl, i32,
                                            C.POINTER(NccView)]
    lib.ebvo_stereo_run.argtypes = [vp, C.POINTER(StereoParams), C.POINTER(StereoCounts)]
    lib.ebvo_stereo_fetch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ebvo_stereo_set_slots.argtypes = [vp, i32]
    lib.ebvo_stereo_upload_slot.argtypes = [vp, i32, vp, vp, i32, i32, ssz, ssz]
    lib.ebvo_stereo_submit.argtypes = [vp, i32, C.POINTER(StereoParams)]
    lib.ebvo_stereo_upload_async.argtypes = [vp, i32, vp, vp, i32, i32, ssz, ssz]
    lib.ebvo_host_register.argtypes = [vp, vp, C.c_size_t]
    lib.ebvo_host_unregister.argtypes = [vp, vp]
    lib.ebvo_ingest_stats.argtypes = [vp, vp]
    lib.ebvo_stereo_fetch_compact_begin.argtypes = [vp, i32, i32]
    lib.ebvo_stereo_fetch_compact_end.argtypes = [vp, i32, C.POINTER(CompactView)]
    lib.ebvo_stereo_pushed_view.argtypes = [vp, i32, C.POINTER(CompactView)]
    lib.ebvo_stereo_wait.argtypes = [vp, i32, C.POINTER(StereoCounts)]
    lib.ebvo_stereo_fetch_slot.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ebvo_profile_enable.argtypes = [vp, i32]
    lib.ebvo_profile_reset.argtypes = [vp]
    lib.ebvo_profile_get.argtypes = [vp, C.POINTER(KernelTime), C.POINTER(i32)]
    lib.ebvo_fp64_peak.argtypes = [vp, i32, C.POINTER(dbl), C.POINTER(dbl)]
    lib.ebvo_debug_set.argtypes = [vp, i32, i32]
    lib.ebvo_stereo_fetch_begin.argtypes = [vp, i32, i32]
    lib.ebvo_sift_descriptors.argtypes = [vp, vp, i32, i32, ssz, vp, i32, vp]
    lib.ebvo_sift_min_distances.argtypes = [vp, vp, i32, vp, vp, vp]
    lib.ebvo_temporal_default_params.restype = None
    lib.ebvo_temporal_default_params.argtypes = [C.POINTER(TemporalParams)]
    lib.ebvo_temporal_set_keyframe.argtypes = [vp, i32]
    lib.ebvo_temporal_match.argtypes = [vp, i32, C.POINTER(TemporalParams), C.POINTER(TemporalCounts)]
    lib.ebvo_temporal_match_submit.argtypes = [vp, i32, C.POINTER(TemporalParams)]
    lib.ebvo_temporal_match_wait.argtypes = [vp, i32, C.POINTER(TemporalCounts)]
    lib.ebvo_temporal_fetch.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    lib.ebvo_temporal_fetch_final.restype = i32
    lib.ebvo_temporal_fetch_final.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ebvo_undistort.argtypes = [vp, vp, i32, i32, ssz, vp, vp, i32, vp, ssz]
    lib.ebvo_stereo_set_undistort.argtypes = [vp, C.POINTER(UndistortParams)]
    lib.ebvo_stereo_fetch_end.argtypes = [vp, i32, C.POINTER(StereoView)]
    lib.ebvo_gn_default_params.restype = None
    lib.ebvo_gn_default_params.argtypes = [C.POINTER(GnParams)]
    lib.ebvo_sobel_gradients.argtypes = [vp, vp, i32, i32, ssz, vp, vp]
    lib.ebvo_gn_refine_temporal.argtypes = [vp, vp, vp, i32, i32, ssz, ssz, vp, vp, vp, i32, C.POINTER(GnParams), vp, vp, vp,
                                            vp]
    lib.ebvo_bnb_test.argtypes = [vp, vp, i32, vp, dbl, i32, vp, vp]
    lib.ebvo_keep_best.argtypes = [vp, vp, i32, vp, vp, vp]
    lib.ebvo_epipolar_shift.argtypes = [vp, vp, vp, vp, i32, vp]
    lib.ebvo_cluster_rows.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp]
    lib.ebvo_stereo_finalize.argtypes = [vp, i32, C.POINTER(FinalizeParams), C.POINTER(StereoCalib), C.POINTER(FinalizeCounts)]
    lib.ebvo_stereo_finalize_submit.argtypes = [vp, i32, C.POINTER(FinalizeParams), C.POINTER(StereoCalib)]
    lib.ebvo_stereo_finalize_wait.argtypes = [vp, i32, C.POINTER(FinalizeCounts)]
    lib.ebvo_stereo_fetch_final.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.ebvo_finalize_pairs.argtypes = [vp, C.POINTER(StereoCalib), vp, vp, i32, vp]
    lib.ebvo_stereo_refine.argtypes = [vp, i32, C.POINTER(GnParams)]
    lib.ebvo_stereo_fetch_refined.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    lib.ebvo_gn_refine_stereo.argtypes = [vp, vp, vp, i32, i32, ssz, ssz, vp, i32, vp, vp, vp, C.POINTER(GnParams),
                                          vp, vp, vp, vp, vp, vp]
    _lib = lib
    return lib


def ptr(a):
    """void* of a numpy array (None -> NULL)."""
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)
