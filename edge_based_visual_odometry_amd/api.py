"""Thin Python view of the C ABI, used by the tests and by bench.py.

The product's host side is C++ (include/ebvo/*.hpp adapters mirroring the reference classes);
this module only marshals numpy arrays to the same entry points, so that the parity tests read
like calls to the reference: ``get_Third_Order_Edges``, ``apply_NCC_Filtering`` ...
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import EDGE_DTYPE, EbvoError, StereoCounts, StereoParams, ptr


def _u8(img: np.ndarray) -> np.ndarray:
    if img.dtype != np.uint8 or img.ndim != 2:
        raise TypeError("image must be a 2-D uint8 array (CV_8UC1)")
    if img.strides[1] != 1:
        img = np.ascontiguousarray(img)
    return img


def _edges(e: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(e, dtype=EDGE_DTYPE)


@dataclass
class ToedResult:
    edges: np.ndarray      # EDGE_DTYPE, toed_edges
    n_total: int           # Total_Num_Of_TOED
    all4: np.ndarray | None  # subpix_edge_pts_final rows (x, y, theta, mag)
    time_conv: float
    time_nms: float


class Context:
    """One HIP device workspace (``ebvo_ctx``).  Not thread-safe; one per process and GPU."""

    def __init__(self, max_h: int, max_w: int, device: int = 0, toed_mode: str | None = None):
        self.lib = _lib.load_library()
        self._ctx = C.c_void_p()
        rc = self.lib.ebvo_ctx_create(device, max_h, max_w, C.byref(self._ctx))
        if rc != 0:
            raise EbvoError(rc, "ebvo_ctx_create", self.lib.ebvo_strerror(rc).decode())
        self.max_h, self.max_w, self.device = max_h, max_w, device
        if toed_mode is not None:
            self.set_toed_mode(toed_mode)

    def set_toed_mode(self, mode: str):
        """'strict': direct-form convolution everywhere; 'hybrid': separable screen + exact re-evaluation."""
        m = {"strict": _lib.TOED_STRICT, "hybrid": _lib.TOED_HYBRID}[mode]
        self._check(self.lib.ebvo_set_toed_mode(self._ctx, m), "ebvo_set_toed_mode")

    def toed_stats(self, slot: int = 0) -> dict:
        out = np.zeros(8, dtype=np.int32)
        self._check(self.lib.ebvo_toed_stats(self._ctx, slot, ptr(out)), "ebvo_toed_stats")
        return {"left": dict(n_total=int(out[0]), n_kept=int(out[1]), n_candidates=int(out[2]), n_neighbour_points=int(out[3])),
                "right": dict(n_total=int(out[4]), n_kept=int(out[5]), n_candidates=int(out[6]),
                              n_neighbour_points=int(out[7]))}

    def toed_screen_audit(self, img: np.ndarray) -> dict:
        """ebvo_toed_screen_audit: max |screen - exact| of the hybrid detector's FP32 screen on this image, with its budget."""
        img = _u8(img)
        h, w = img.shape
        a = _lib.ScreenAudit()
        self._check(self.lib.ebvo_toed_screen_audit(self._ctx, ptr(img), h, w, img.strides[0], C.byref(a)),
                    "ebvo_toed_screen_audit")
        return {name: getattr(a, name) for name, _ in a._fields_}

    @property
    def toed_fallbacks(self) -> int:
        """Hybrid TOED runs the library repeated on the strict path (screened candidates > max_h * max_w)."""
        return int(self.lib.ebvo_toed_fallbacks(self._ctx))

    @property
    def graph_launches(self) -> int:
        """Pairs that ebvo_stereo_submit launched as a captured hipGraph."""
        return int(self.lib.ebvo_graph_launches(self._ctx))

    @property
    def toed_mode(self) -> str:
        return "hybrid" if self.lib.ebvo_get_toed_mode(self._ctx) == _lib.TOED_HYBRID else "strict"

    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.ebvo_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int, where: str):
        if rc != 0:
            raise EbvoError(rc, where, self.lib.ebvo_strerror(rc).decode() + "; " +
                            self.lib.ebvo_last_error(self._ctx).decode())

    # -- ThirdOrderEdgeDetectionCPU::get_Third_Order_Edges ---------------------------------
    def toed(self, img: np.ndarray, want_all: bool = False, cap: int | None = None) -> ToedResult:
        img = _u8(img)
        h, w = img.shape
        cap = h * w if cap is None else cap
        out = np.zeros(cap, dtype=EDGE_DTYPE)
        all4 = np.zeros((cap, 4), dtype=np.float64) if want_all else None
        nk, nt = C.c_int(), C.c_int()
        tc, tn = C.c_double(), C.c_double()
        rc = self.lib.ebvo_toed(self._ctx, ptr(img), h, w, img.strides[0], ptr(out), cap, C.byref(nk),
                                C.byref(nt), ptr(all4), cap if want_all else 0, C.byref(tc), C.byref(tn))
        if rc == _lib.EBVO_ERR_CAPACITY:
            raise EbvoError(rc, f"ebvo_toed (need kept={nk.value}, total={nt.value})")
        self._check(rc, "ebvo_toed")
        return ToedResult(out[: nk.value].copy(), nt.value,
                          all4[: nt.value].copy() if want_all else None, tc.value, tn.value)

    def toed_pair(self, left: np.ndarray, right: np.ndarray):
        left, right = _u8(left), _u8(right)
        h, w = left.shape
        assert right.shape == (h, w)
        cap = h * w
        ol, orr = np.zeros(cap, dtype=EDGE_DTYPE), np.zeros(cap, dtype=EDGE_DTYPE)
        nk, nt = (C.c_int * 2)(), (C.c_int * 2)()
        rc = self.lib.ebvo_toed_pair(self._ctx, ptr(left), ptr(right), h, w, left.strides[0], right.strides[0],
                                     ptr(ol), ptr(orr), cap, nk, nt)
        self._check(rc, "ebvo_toed_pair")
        return ol[: nk[0]].copy(), orr[: nk[1]].copy(), (nt[0], nt[1])

    # -- Stereo_Matches::CalculateEpipolarLine ----------------------------------------------
    def epipolar_lines(self, F: np.ndarray, edges: np.ndarray) -> np.ndarray:
        F = np.ascontiguousarray(F, dtype=np.float64).reshape(9)
        edges = _edges(edges)
        lines = np.zeros((len(edges), 3), dtype=np.float64)
        self._check(self.lib.ebvo_epipolar_lines(ptr(F), ptr(edges), len(edges), ptr(lines)), "ebvo_epipolar_lines")
        return lines

    # -- apply_Epipolar_Line_Distance_Filtering / Disparity / orientation -------------------
    def epi_candidates(self, L, R, lines, epi_thr=0.5, max_disp=25.0, orient_thr_deg=10.0,
                       stage_mask=_lib.STAGE_ALL):
        L, R = _edges(L), _edges(R)
        lines = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 3)
        assert len(lines) == len(L)
        row_ptr = np.zeros(len(L) + 1, dtype=np.int32)
        n = C.c_int64()
        rc = self.lib.ebvo_epi_candidates(self._ctx, ptr(L), len(L), ptr(R), len(R), ptr(lines), epi_thr, max_disp,
                                          orient_thr_deg, stage_mask, ptr(row_ptr), None, 0, C.byref(n))
        self._check(rc, "ebvo_epi_candidates(size)")
        col = np.zeros(max(1, n.value), dtype=np.int32)
        rc = self.lib.ebvo_epi_candidates(self._ctx, ptr(L), len(L), ptr(R), len(R), ptr(lines), epi_thr, max_disp,
                                          orient_thr_deg, stage_mask, ptr(row_ptr), ptr(col), len(col), C.byref(n))
        self._check(rc, "ebvo_epi_candidates")
        return row_ptr, col[: n.value].copy()

    def epi_candidates_staged(self, L, R, lines, epi_thr=0.5, max_disp=25.0, orient_thr_deg=10.0):
        """One search for the three geometric stages: the (epipolar AND disparity) list + the orientation flag of every
        listed pair.  One call: the list is sized generously first and fetched again only if it did not fit."""
        L, R = _edges(L), _edges(R)
        lines = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 3)
        assert len(lines) == len(L)
        row_ptr = np.zeros(len(L) + 1, dtype=np.int32)
        n = C.c_int64()
        cap = max(1024, 24 * len(L))
        for _ in range(2):
            col, ok = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.uint8)
            rc = self.lib.ebvo_epi_candidates_staged(self._ctx, ptr(L), len(L), ptr(R), len(R), ptr(lines), epi_thr, max_disp,
                                                     orient_thr_deg, ptr(row_ptr), ptr(col), ptr(ok), cap, C.byref(n))
            if rc != _lib.EBVO_ERR_CAPACITY:
                break
            cap = int(n.value)
        self._check(rc, "ebvo_epi_candidates_staged")
        return row_ptr, col[: n.value].copy(), ok[: n.value].copy()

    # -- Stereo_Matches::apply_NCC_Filtering --------------------------------------------------
    def ncc_pairs(self, imgL, imgR, L, Rc, row_ptr, thr=0.6, want_left_patches=False):
        imgL, imgR = _u8(imgL), _u8(imgR)
        h, w = imgL.shape
        L, Rc = _edges(L), _edges(Rc)
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        npairs = int(row_ptr[-1]) if len(row_ptr) else 0
        assert len(Rc) == npairs
        sims = np.zeros((npairs, 4), dtype=np.float64)
        best = np.zeros(npairs, dtype=np.float64)
        keep = np.zeros(npairs, dtype=np.uint8)
        lp = np.zeros((len(L), 2, 49), dtype=np.float32) if want_left_patches else None
        rc = self.lib.ebvo_ncc_pairs(self._ctx, ptr(imgL), ptr(imgR), h, w, imgL.strides[0], imgR.strides[0], ptr(L),
                                     len(L), ptr(Rc), ptr(row_ptr), thr, ptr(lp), ptr(sims), ptr(best), ptr(keep))
        self._check(rc, "ebvo_ncc_pairs")
        return sims, best, keep, lp

    # -- the resident stage-wise calls (what a stage produced stays on the device under a tag) ------------------------
    @staticmethod
    def _view(ptr_, count, dtype):
        """copy of `count` items of `dtype` at a page-locked address the library returned"""
        if not ptr_ or count == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_ubyte * (int(count) * np.dtype(dtype).itemsize)).from_address(ptr_)
        return np.frombuffer(buf, dtype=dtype, count=int(count)).copy()

    def toed_resident(self, img: np.ndarray, which: int, want_all: bool = False):
        """ebvo_toed_resident: (edges, n_total, all4 or None, tag)"""
        img = _u8(img)
        h, w = img.shape
        v = _lib.ToedView()
        self._check(self.lib.ebvo_toed_resident(self._ctx, which, ptr(img), h, w, img.strides[0], int(want_all), C.byref(v)),
                    "ebvo_toed_resident")
        all4 = self._view(v.all4, v.n_total * 4, np.float64).reshape(-1, 4) if want_all else None
        return self._view(v.edges, v.n_kept, EDGE_DTYPE), v.n_total, all4, int(v.tag)

    def epi_candidates_resident(self, tag_left, tag_right, lines, epi_thr=0.5, max_disp=25.0, orient_thr_deg=10.0,
                                stage_mask=_lib.STAGE_ALL, staged=False):
        """ebvo_epi_candidates_resident: (row_ptr, col_idx[, orient_ok]); staged = the (epipolar AND disparity) list + flags"""
        lines = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 3)
        v = _lib.CandidatesView()
        mask = (_lib.STAGE_EPIPOLAR | _lib.STAGE_DISPARITY) if staged else stage_mask
        self._check(self.lib.ebvo_epi_candidates_resident(self._ctx, tag_left, tag_right, ptr(lines), epi_thr, max_disp,
                                                          orient_thr_deg, mask, int(staged), C.byref(v)),
                    "ebvo_epi_candidates_resident")
        rp = self._view(v.row_ptr, len(lines) + 1, np.int32)
        ci = self._view(v.col_idx, v.n_pairs, np.int32)
        if not staged:
            return rp, ci
        # the list the flags select (after apply_orientation_filter), formed on the device as well
        self.last_final_lists = (self._view(v.row_ptr_final, len(lines) + 1, np.int32), self._view(v.col_idx_final, v.n_final, np.int32))
        return rp, ci, self._view(v.orient_ok, v.n_pairs, np.uint8)

    def ncc_pairs_resident(self, tag_left, tag_right, imgL, imgR, row_ptr, col_idx, thr=0.6, want_left_patches=False,
                           want_sims=True):
        """ebvo_ncc_pairs_resident: (sims or None, best, keep, left_patches or None)"""
        imgL, imgR = _u8(imgL), _u8(imgR)
        h, w = imgL.shape
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        want = (_lib.NCC_WANT_LEFT_PATCHES if want_left_patches else 0) | (_lib.NCC_WANT_SIMS if want_sims else 0)
        v = _lib.NccView()
        self._check(self.lib.ebvo_ncc_pairs_resident(self._ctx, tag_left, tag_right, ptr(imgL), ptr(imgR), h, w, imgL.strides[0],
                                                     imgR.strides[0], ptr(row_ptr), ptr(col_idx), thr, want, C.byref(v)),
                    "ebvo_ncc_pairs_resident")
        sims = self._view(v.sims, v.n_pairs * 4, np.float64).reshape(-1, 4) if want_sims else None
        lp = self._view(v.left_patches, v.n_left * 98, np.float32).reshape(-1, 2, 49) if want_left_patches else None
        return sims, self._view(v.best, v.n_pairs, np.float64), self._view(v.keep, v.n_pairs, np.uint8), lp

    # -- cv::undistort (src/Pipeline.cpp:78-79) ---------------------------------------------------------
    def undistort(self, img, K, dist):
        img = _u8(img)
        h, w = img.shape
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(4)
        dist = np.ascontiguousarray(dist, dtype=np.float64).reshape(-1)
        out = np.zeros((h, w), dtype=np.uint8)
        self._check(self.lib.ebvo_undistort(self._ctx, ptr(img), h, w, img.strides[0], ptr(K), ptr(dist), len(dist), ptr(out),
                                            out.strides[0]), "ebvo_undistort")
        return out

    def set_undistort(self, K_left=None, dist_left=None, K_right=None, dist_right=None):
        """Resident pipeline: undistort the uploaded pair (TOED / refinement on the result, NCC on the raw images).
        No arguments: off."""
        if K_left is None:
            self._check(self.lib.ebvo_stereo_set_undistort(self._ctx, None), "ebvo_stereo_set_undistort")
            return
        p = _lib.UndistortParams()
        n = len(dist_left)
        assert len(dist_right) == n and n in (4, 5)
        for k in range(4):
            p.K_left[k], p.K_right[k] = float(K_left[k]), float(K_right[k])
        for k in range(n):
            p.dist_left[k], p.dist_right[k] = float(dist_left[k]), float(dist_right[k])
        p.n_dist = n
        self._check(self.lib.ebvo_stereo_set_undistort(self._ctx, C.byref(p)), "ebvo_stereo_set_undistort")

    # -- util_compute_Img_Gradients (include/utility.h:131-141) -----------------------------------
    def sobel_gradients(self, img):
        img = _u8(img)
        h, w = img.shape
        gx = np.zeros((h, w), dtype=np.float32)
        gy = np.zeros((h, w), dtype=np.float32)
        self._check(self.lib.ebvo_sobel_gradients(self._ctx, ptr(img), h, w, img.strides[0], ptr(gx), ptr(gy)),
                    "ebvo_sobel_gradients")
        return gx, gy

    # -- Stereo_Matches::refine_edge_disparity (src/Stereo_Matches.cpp:1290-1358) ----------------
    def gn_refine_stereo(self, imgL, imgR, L, lines, row_ptr, cand_xy, max_iter=None, tol=None, huber_delta=None):
        """Photometric Gauss-Newton refinement of every (left edge, candidate) pair along the epipolar line.
        Returns dict(alpha, score, confidence, validity, iters, refined_xy)."""
        from ._lib import GnParams
        imgL, imgR = _u8(imgL), _u8(imgR)
        h, w = imgL.shape
        L = _edges(L)
        lines = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 3)
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        cand_xy = np.ascontiguousarray(cand_xy, dtype=np.float64).reshape(-1, 2)
        n = int(row_ptr[-1]) if len(row_ptr) else 0
        assert len(cand_xy) == n and len(lines) == len(L) and len(row_ptr) == len(L) + 1
        p = GnParams()
        self.lib.ebvo_gn_default_params(p)
        if max_iter is not None:
            p.max_iter = int(max_iter)
        if tol is not None:
            p.tol = float(tol)
        if huber_delta is not None:
            p.huber_delta = float(huber_delta)
        out = dict(alpha=np.zeros(n), score=np.zeros(n), confidence=np.zeros(n), validity=np.zeros(n, dtype=np.uint8),
                   iters=np.zeros(n, dtype=np.int32), refined_xy=np.zeros((n, 2)))
        rc = self.lib.ebvo_gn_refine_stereo(self._ctx, ptr(imgL), ptr(imgR), h, w, imgL.strides[0], imgR.strides[0],
                                            ptr(L), len(L), ptr(lines), ptr(row_ptr), ptr(cand_xy), C.byref(p),
                                            ptr(out["alpha"]), ptr(out["score"]), ptr(out["confidence"]),
                                            ptr(out["validity"]), ptr(out["iters"]), ptr(out["refined_xy"]))
        self._check(rc, "ebvo_gn_refine_stereo")
        return out

    def _gn_params(self, max_iter=None, tol=None, huber_delta=None):
        from ._lib import GnParams
        p = GnParams()
        self.lib.ebvo_gn_default_params(p)
        if max_iter is not None:
            p.max_iter = int(max_iter)
        if tol is not None:
            p.tol = float(tol)
        if huber_delta is not None:
            p.huber_delta = float(huber_delta)
        return p

    # -- Temporal_Matches::min_Edge_Photometric_Residual_by_Gauss_Newton (src/Temporal_Matches.cpp:735-851) ------
    def gn_refine_temporal(self, imgKF, imgCF, kf, cf, init_disp, **kw):
        imgKF, imgCF = _u8(imgKF), _u8(imgCF)
        h, w = imgKF.shape
        kf, cf = _edges(kf), _edges(cf)
        init_disp = np.ascontiguousarray(init_disp, dtype=np.float64).reshape(-1, 2)
        n = len(kf)
        assert len(cf) == n and len(init_disp) == n
        p = self._gn_params(**kw)
        out = dict(disp=np.zeros((n, 2)), score=np.zeros(n), validity=np.zeros(n, dtype=np.uint8),
                   iters=np.zeros(n, dtype=np.int32))
        rc = self.lib.ebvo_gn_refine_temporal(self._ctx, ptr(imgKF), ptr(imgCF), h, w, imgKF.strides[0], imgCF.strides[0],
                                              ptr(kf), ptr(cf), ptr(init_disp), n, C.byref(p), ptr(out["disp"]),
                                              ptr(out["score"]), ptr(out["validity"]), ptr(out["iters"]))
        self._check(rc, "ebvo_gn_refine_temporal")
        return out

    # -- stage glue: apply_Best_Nearly_Best_Test / apply_Lowe_Ratio_Test / shift_Edge_to_Epipolar_Line ---------------
    def bnb_test(self, row_ptr, scores, ratio_thr, higher_is_better=True):
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        nL = len(row_ptr) - 1
        cnt = np.zeros(nL, dtype=np.int32)
        order = np.full(len(scores), -1, dtype=np.int32)
        self._check(self.lib.ebvo_bnb_test(self._ctx, ptr(row_ptr), nL, ptr(scores), ratio_thr, int(higher_is_better),
                                           ptr(cnt), ptr(order)), "ebvo_bnb_test")
        return cnt, order

    def keep_best(self, row_ptr, scores):
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        scores = np.ascontiguousarray(scores, dtype=np.float64)
        nL = len(row_ptr) - 1
        cnt = np.zeros(nL, dtype=np.int32)
        order = np.full(len(scores), -1, dtype=np.int32)
        self._check(self.lib.ebvo_keep_best(self._ctx, ptr(row_ptr), nL, ptr(scores), ptr(cnt), ptr(order)),
                    "ebvo_keep_best")
        return cnt, order

    def epipolar_shift(self, cand, lines, row_ptr):
        cand = _edges(cand)
        lines = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 3)
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        out = np.zeros(len(cand), dtype=EDGE_DTYPE)
        self._check(self.lib.ebvo_epipolar_shift(self._ctx, ptr(cand), ptr(lines), ptr(row_ptr), len(row_ptr) - 1,
                                                 ptr(out)), "ebvo_epipolar_shift")
        return out

    def cluster_rows(self, cand, row_ptr, by_orientation=False, skip_single=True):
        """EdgeClusterer per row: (new_count, centres, cluster_of)."""
        cand = _edges(cand)
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        nL = len(row_ptr) - 1
        cnt = np.zeros(nL, dtype=np.int32)
        centres = np.zeros(len(cand), dtype=EDGE_DTYPE)
        cluster_of = np.full(len(cand), -1, dtype=np.int32)
        self._check(self.lib.ebvo_cluster_rows(self._ctx, ptr(cand), ptr(row_ptr), nL, int(by_orientation), int(skip_single),
                                               ptr(cnt), ptr(centres), ptr(cluster_of)), "ebvo_cluster_rows")
        return cnt, centres, cluster_of

    # -- write_finalized_stereo_edge_pairs_to_file, numeric body (src/Stereo_Matches.cpp:1656-1699) -----------------
    def finalize_pairs(self, K_left, K_right, R21, T21, left, right) -> np.ndarray:
        from ._lib import StereoCalib
        cal = StereoCalib()
        for name, v, n in (("K_left", K_left, 9), ("K_right", K_right, 9), ("R21", R21, 9), ("T21", T21, 3)):
            a = np.ascontiguousarray(v, dtype=np.float64).reshape(n)
            getattr(cal, name)[:] = a.tolist()
        left, right = _edges(left), _edges(right)
        assert len(left) == len(right)
        out = np.zeros((len(left), 16))
        self._check(self.lib.ebvo_finalize_pairs(self._ctx, C.byref(cal), ptr(left), ptr(right), len(left), ptr(out)),
                    "ebvo_finalize_pairs")
        return out

    # -- cv::SIFT::compute at the +-8 px points of every edge / apply_SIFT_filtering's score ---------------------------
    def sift_descriptors(self, img, edges) -> np.ndarray:
        img = _u8(img)
        h, w = img.shape
        edges = _edges(edges)
        out = np.zeros((len(edges), 2, 128), dtype=np.float32)
        self._check(self.lib.ebvo_sift_descriptors(self._ctx, ptr(img), h, w, img.strides[0], ptr(edges), len(edges), ptr(out)),
                    "ebvo_sift_descriptors")
        return out

    def sift_min_distances(self, left_desc, cand_desc, row_ptr) -> np.ndarray:
        left_desc = np.ascontiguousarray(left_desc, dtype=np.float32).reshape(-1, 2, 128)
        cand_desc = np.ascontiguousarray(cand_desc, dtype=np.float32).reshape(-1, 2, 128)
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        assert len(row_ptr) == len(left_desc) + 1 and int(row_ptr[-1]) == len(cand_desc)
        out = np.zeros(len(cand_desc))
        self._check(self.lib.ebvo_sift_min_distances(self._ctx, ptr(left_desc), len(left_desc), ptr(cand_desc), ptr(row_ptr),
                                                     ptr(out)), "ebvo_sift_min_distances")
        return out

    def _finalize_args(self, calib, bnb_ratio, ncc_thr, use_sift, sift_thr, bnb_sift, gn):
        from ._lib import FinalizeParams, StereoCalib
        p = FinalizeParams()
        p.bnb_ratio, p.ncc_thr, p.gn = bnb_ratio, ncc_thr, self._gn_params(**gn)
        p.use_sift, p.sift_thr, p.bnb_sift = int(bool(use_sift)), sift_thr, bnb_sift
        cal = None
        if calib is not None:
            cal = StereoCalib()
            for name, v, n in zip(("K_left", "K_right", "R21", "T21"), calib, (9, 9, 9, 3)):
                getattr(cal, name)[:] = np.ascontiguousarray(v, dtype=np.float64).reshape(n).tolist()
        return p, cal

    def stereo_finalize_submit(self, calib=None, slot=0, bnb_ratio=0.9, ncc_thr=0.6, use_sift=False, sift_thr=500.0,
                               bnb_sift=0.4, **gn):
        """Enqueue the chain after the first NCC pass on the slot's stream and return (ebvo_stereo_finalize_submit)."""
        p, cal = self._finalize_args(calib, bnb_ratio, ncc_thr, use_sift, sift_thr, bnb_sift, gn)
        self._check(self.lib.ebvo_stereo_finalize_submit(self._ctx, slot, C.byref(p), C.byref(cal) if cal is not None else None),
                    "ebvo_stereo_finalize_submit")
        self._fin_opts = getattr(self, "_fin_opts", {})
        self._fin_opts[slot] = (calib is not None, bool(use_sift))

    def stereo_finalize_wait(self, slot=0, fetch=True):
        """Wait for the chain of `slot`; returns (counts dict, dict(left_index, right, score[, rows]) or None)."""
        from ._lib import FinalizeCounts
        cnt = FinalizeCounts()
        self._check(self.lib.ebvo_stereo_finalize_wait(self._ctx, slot, C.byref(cnt)), "ebvo_stereo_finalize_wait")
        has_rows, use_sift = self._fin_opts.get(slot, (False, False))
        keys = ("n_sift",) * use_sift + ("n_ncc", "n_bnb", "n_clusters", "n_ncc2", "n_final")
        counts = {k: getattr(cnt, k) for k in keys}
        if not fetch:
            return counts, None
        n = cnt.n_final
        out = dict(left_index=np.zeros(n, dtype=np.int32), right=np.zeros(n, dtype=EDGE_DTYPE), score=np.zeros(n))
        rows = np.zeros((n, 16)) if has_rows else None
        self._check(self.lib.ebvo_stereo_fetch_final(self._ctx, slot, ptr(out["left_index"]), ptr(out["right"]),
                                                     ptr(out["score"]), ptr(rows)), "ebvo_stereo_fetch_final")
        if rows is not None:
            out["rows"] = rows
        return counts, out

    def stereo_finalize(self, calib=None, slot=0, bnb_ratio=0.9, ncc_thr=0.6, use_sift=False, sift_thr=500.0, bnb_sift=0.4,
                        **gn):
        """BNB -> shift -> refine -> cluster -> NCC -> best on the resident pair.  calib = (K_left, K_right, R21, T21) adds the
        output-file rows.  Returns (counts dict, dict(left_index, right, score[, rows]))."""
        self.stereo_finalize_submit(calib, slot, bnb_ratio, ncc_thr, use_sift, sift_thr, bnb_sift, **gn)
        return self.stereo_finalize_wait(slot)

    def stereo_refine(self, counts, slot=0, **kw):
        """Refine every kept match of the resident pair on the device; returns the per-pair outputs (n_pairs entries,
        validity 255 where the pair was not a kept match)."""
        p = self._gn_params(**kw)
        self._check(self.lib.ebvo_stereo_refine(self._ctx, slot, C.byref(p)), "ebvo_stereo_refine")
        n = int(counts.n_pairs)
        out = dict(alpha=np.zeros(n), score=np.zeros(n), confidence=np.zeros(n), validity=np.zeros(n, dtype=np.uint8),
                   iters=np.zeros(n, dtype=np.int32), refined_xy=np.zeros((n, 2)))
        self._check(self.lib.ebvo_stereo_fetch_refined(self._ctx, slot, ptr(out["alpha"]), ptr(out["score"]),
                                                       ptr(out["confidence"]), ptr(out["validity"]), ptr(out["iters"]),
                                                       ptr(out["refined_xy"])), "ebvo_stereo_fetch_refined")
        return out

    # -- Utility::get_edge_patches ------------------------------------------------------------
    def edge_patches(self, img, edges) -> np.ndarray:
        img = _u8(img)
        h, w = img.shape
        edges = _edges(edges)
        out = np.zeros((len(edges), 2, 49), dtype=np.float32)
        self._check(self.lib.ebvo_edge_patches(self._ctx, ptr(img), h, w, img.strides[0], ptr(edges), len(edges),
                                               ptr(out)), "ebvo_edge_patches")
        return out

    # -- Utility::get_patch_similarity / MatlabNCCComputer::computeNCC -----------------------
    def ncc_patches(self, A, B) -> np.ndarray:
        A = np.ascontiguousarray(A, dtype=np.float32).reshape(-1, 49)
        B = np.ascontiguousarray(B, dtype=np.float32).reshape(-1, 49)
        assert A.shape == B.shape
        sim = np.zeros(len(A), dtype=np.float64)
        self._check(self.lib.ebvo_ncc_patches(self._ctx, ptr(A), ptr(B), len(A), ptr(sim)), "ebvo_ncc_patches")
        return sim

    # -- Temporal_Matches::apply_NCC_filtering_quads ------------------------------------------
    def ncc_quads(self, kfL, kfR, cfL, cfR, thr=0.8):
        arrs = [np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 98) for a in (kfL, kfR, cfL, cfR)]
        n = len(arrs[0])
        sl, sr = np.zeros(n), np.zeros(n)
        keep = np.zeros(n, dtype=np.uint8)
        self._check(self.lib.ebvo_ncc_quads(self._ctx, *[ptr(a) for a in arrs], n, thr, ptr(sl), ptr(sr), ptr(keep)),
                    "ebvo_ncc_quads")
        return sl, sr, keep

    # -- device-resident pipeline --------------------------------------------------------------
    def default_params(self, F21=None) -> StereoParams:
        p = StereoParams()
        self.lib.ebvo_stereo_default_params(C.byref(p))
        if F21 is not None:
            F = np.ascontiguousarray(F21, dtype=np.float64).reshape(9)
            for k in range(9):
                p.F21[k] = float(F[k])
        return p

    def set_slots(self, n: int):
        self._check(self.lib.ebvo_stereo_set_slots(self._ctx, n), "ebvo_stereo_set_slots")

    def stereo_upload(self, left, right, slot: int = 0):
        left, right = _u8(left), _u8(right)
        h, w = left.shape
        assert right.shape == (h, w)
        self._check(self.lib.ebvo_stereo_upload_slot(self._ctx, slot, ptr(left), ptr(right), h, w, left.strides[0],
                                                     right.strides[0]), "ebvo_stereo_upload_slot")

    def stereo_upload_async(self, left, right, slot: int = 0):
        """ebvo_stereo_upload_async: returns once the copies are enqueued (page-locked sources: host_register); the slot's next
        submission waits for them on the device.  The arrays must stay alive and unchanged until the pair has been waited for."""
        assert left.dtype == np.uint8 and right.dtype == np.uint8 and left.flags.c_contiguous and right.flags.c_contiguous
        h, w = left.shape
        assert right.shape == (h, w)
        self._check(self.lib.ebvo_stereo_upload_async(self._ctx, slot, ptr(left), ptr(right), h, w, left.strides[0],
                                                      right.strides[0]), "ebvo_stereo_upload_async")

    def host_register(self, arr: np.ndarray):
        """page-lock a numpy array (a frame ring) for asynchronous uploads"""
        self._check(self.lib.ebvo_host_register(self._ctx, ptr(arr), arr.nbytes), "ebvo_host_register")

    def ingest_stats(self) -> dict:
        out = np.zeros(2, dtype=np.int64)
        self._check(self.lib.ebvo_ingest_stats(self._ctx, ptr(out)), "ebvo_ingest_stats")
        return {"pull_uploads": int(out[0]), "stream_uploads": int(out[1])}

    def host_unregister(self, arr: np.ndarray):
        self._check(self.lib.ebvo_host_unregister(self._ctx, ptr(arr)), "ebvo_host_unregister")

    def stereo_run(self, params: StereoParams) -> StereoCounts:
        c = StereoCounts()
        self._check(self.lib.ebvo_stereo_run(self._ctx, C.byref(params), C.byref(c)), "ebvo_stereo_run")
        return c

    def stereo_submit(self, params: StereoParams, slot: int = 0):
        self._check(self.lib.ebvo_stereo_submit(self._ctx, slot, C.byref(params)), "ebvo_stereo_submit")

    def stereo_wait(self, slot: int = 0) -> StereoCounts:
        c = StereoCounts()
        self._check(self.lib.ebvo_stereo_wait(self._ctx, slot, C.byref(c)), "ebvo_stereo_wait")
        return c

    def stereo_fetch(self, counts: StereoCounts, patches: bool = False, slot: int = 0):
        nL, nR, npairs = counts.n_left, counts.n_right, counts.n_pairs
        left = np.zeros(nL, dtype=EDGE_DTYPE)
        right = np.zeros(nR, dtype=EDGE_DTYPE)
        row_ptr = np.zeros(nL + 1, dtype=np.int32)
        col = np.zeros(npairs, dtype=np.int32)
        sims = np.zeros((npairs, 4))
        best = np.zeros(npairs)
        keep = np.zeros(npairs, dtype=np.uint8)
        lp = np.zeros((nL, 2, 49), dtype=np.float32) if patches else None
        self._check(self.lib.ebvo_stereo_fetch_slot(self._ctx, slot, ptr(left), ptr(right), ptr(row_ptr), ptr(col),
                                                    ptr(sims), ptr(best), ptr(keep), ptr(lp)), "ebvo_stereo_fetch_slot")
        return dict(left=left, right=right, row_ptr=row_ptr, col_idx=col, sims=sims, best=best, keep=keep,
                    left_patches=lp)

    # -- temporal quads against the keyframe (src/Temporal_Matches.cpp:335-469) -----------------------------------
    def temporal_set_keyframe(self, slot: int = 0):
        self._check(self.lib.ebvo_temporal_set_keyframe(self._ctx, slot), "ebvo_temporal_set_keyframe")

    def temporal_match_submit(self, slot: int = 0, **kw):
        """Enqueue the candidate + NCC stages of the slot's final mates against the keyframe (ebvo_temporal_match_submit)."""
        p = _lib.TemporalParams()
        self.lib.ebvo_temporal_default_params(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        self._check(self.lib.ebvo_temporal_match_submit(self._ctx, slot, C.byref(p)), "ebvo_temporal_match_submit")
        self._tq_stages = getattr(self, "_tq_stages", {})
        self._tq_stages[slot] = int(p.stages)

    def temporal_match_wait(self, slot: int = 0, fetch: bool = True):
        c = _lib.TemporalCounts()
        self._check(self.lib.ebvo_temporal_match_wait(self._ctx, slot, C.byref(c)), "ebvo_temporal_match_wait")
        return self._temporal_results(slot, c, self._tq_stages.get(slot, 0), fetch)

    def temporal_match(self, slot: int = 0, fetch: bool = True, **kw):
        self.temporal_match_submit(slot, **kw)
        return self.temporal_match_wait(slot, fetch)

    def _temporal_results(self, slot, c, stages, fetch):
        counts = dict(n_kf=c.n_kf, n_cf=c.n_cf, n_candidates=c.n_candidates, n_kept=c.n_kept)
        if stages:
            counts.update(n_sift=c.n_sift, n_bnb_ncc=c.n_bnb_ncc, n_bnb_sift=c.n_bnb_sift, n_refined_valid=c.n_refined_valid,
                          n_final=c.n_final)
        if not fetch:
            return counts, None
        n = c.n_candidates
        out = dict(row_ptr=np.zeros(c.n_kf + 1, dtype=np.int32), col_idx=np.zeros(n, dtype=np.int32), sim_left=np.zeros(n),
                   sim_right=np.zeros(n), keep=np.zeros(n, dtype=np.uint8))
        self._check(self.lib.ebvo_temporal_fetch(self._ctx, slot, ptr(out["row_ptr"]), ptr(out["col_idx"]), ptr(out["sim_left"]),
                                                 ptr(out["sim_right"]), ptr(out["keep"])), "ebvo_temporal_fetch")
        if stages:
            m = c.n_final
            fin = dict(row_ptr=np.zeros(c.n_kf + 1, dtype=np.int32), cf_index=np.zeros(m, dtype=np.int32),
                       left=np.zeros(m, dtype=EDGE_DTYPE), right=np.zeros(m, dtype=EDGE_DTYPE), ncc_left=np.zeros(m),
                       sift_left=np.zeros(m), score_left=np.zeros(m), score_right=np.zeros(m), valid=np.zeros(m, dtype=np.uint8))
            self._check(self.lib.ebvo_temporal_fetch_final(self._ctx, slot, *(ptr(fin[k]) for k in (
                "row_ptr", "cf_index", "left", "right", "ncc_left", "sift_left", "score_left", "score_right", "valid"))),
                "ebvo_temporal_fetch_final")
            out["final"] = fin
        return counts, out

    def stereo_fetch_begin(self, slot: int = 0, what: int = _lib.FETCH_DEFAULT):
        """Enqueue the device-to-host copies of a finished pair's results into the slot's page-locked staging."""
        self._check(self.lib.ebvo_stereo_fetch_begin(self._ctx, slot, what), "ebvo_stereo_fetch_begin")

    def stereo_fetch_end(self, slot: int = 0) -> dict:
        """Wait for the copies and return numpy VIEWS of the page-locked arrays (valid until the slot is reused)."""
        v = _lib.StereoView()
        self._check(self.lib.ebvo_stereo_fetch_end(self._ctx, slot, C.byref(v)), "ebvo_stereo_fetch_end")

        def view(ptr_, count, dtype):
            if not ptr_ or count == 0:
                return None if not ptr_ else np.zeros(0, dtype=dtype)
            nbytes = int(count) * np.dtype(dtype).itemsize
            buf = (C.c_ubyte * nbytes).from_address(ptr_)
            return np.frombuffer(buf, dtype=dtype, count=int(count))

        sims = view(v.sims, v.n_pairs * 4, np.float64)
        return dict(left=view(v.left, v.n_left, EDGE_DTYPE), right=view(v.right, v.n_right, EDGE_DTYPE),
                    row_ptr=view(v.row_ptr, v.n_left + 1, np.int32), col_idx=view(v.col_idx, v.n_pairs, np.int32),
                    sims=None if sims is None else sims.reshape(-1, 4), best=view(v.best, v.n_pairs, np.float64),
                    keep=view(v.keep, v.n_pairs, np.uint8))

    def stereo_fetch_compact_begin(self, slot: int = 0, what: int = _lib.COMPACT_DEFAULT):
        self._check(self.lib.ebvo_stereo_fetch_compact_begin(self._ctx, slot, what), "ebvo_stereo_fetch_compact_begin")

    def stereo_pushed_view(self, slot: int = 0) -> dict:
        """the compact results a pair submitted with PAIR_PUSH left in the slot's page-locked arena (no copy, no wait)"""
        v = _lib.CompactView()
        self._check(self.lib.ebvo_stereo_pushed_view(self._ctx, slot, C.byref(v)), "ebvo_stereo_pushed_view")
        return self._compact_dict(v)

    def stereo_fetch_compact_end(self, slot: int = 0) -> dict:
        """numpy VIEWS of the compact arrays: (x, y) pairs, orientations, CSR, best, keep as a bit mask (bit k & 31 of word k >> 5)"""
        v = _lib.CompactView()
        self._check(self.lib.ebvo_stereo_fetch_compact_end(self._ctx, slot, C.byref(v)), "ebvo_stereo_fetch_compact_end")
        return self._compact_dict(v)

    @staticmethod
    def _compact_dict(v) -> dict:

        def view(ptr_, count, dtype):
            if not ptr_:
                return None
            nbytes = int(count) * np.dtype(dtype).itemsize
            if nbytes == 0:
                return np.zeros(0, dtype=dtype)
            return np.frombuffer((C.c_ubyte * nbytes).from_address(ptr_), dtype=dtype, count=int(count))

        lxy, rxy = view(v.left_xy, 2 * v.n_left, np.float64), view(v.right_xy, 2 * v.n_right, np.float64)
        return dict(left_xy=None if lxy is None else lxy.reshape(-1, 2), right_xy=None if rxy is None else rxy.reshape(-1, 2),
                    left_theta=view(v.left_theta, v.n_left, np.float64), right_theta=view(v.right_theta, v.n_right, np.float64),
                    row_ptr=view(v.row_ptr, v.n_left + 1, np.int32), col_idx=view(v.col_idx, v.n_pairs, np.int32),
                    best=view(v.best, v.n_pairs, np.float64), keep_bits=view(v.keep_bits, 2 * ((v.n_pairs + 63) // 64), np.uint32),
                    n_pairs=int(v.n_pairs), n_matches=int(v.n_matches))

    def debug_set(self, key: int, value: int):
        """Test hooks (ebvo_debug_set): 0 = attempts of the regrow loop, 1 = force N overflowed results, 2 = number of
        lanes (streams the submitted pairs are dealt to when more slots are in use; 0 = one stream per slot), 3 = the
        profiler instruments one stage alone (index in profile_get()'s order + 1; 0 = every stage), 4 / 5 = launch layout
        of the refinements (1 = one thread per pair always / threshold of the eight-lanes layout); 10 = pair chain as a
        hipGraph; 11 / 12 / 17 = grids of the exact centre / mags / NCC tile kernels in blocks; 13 / 14 / 18 = A/B switches;
        15 = bit mask of kernels launched twice, 16 = the chain ends after stage N (measurement only): include/ebvo_hip.h."""
        self._check(self.lib.ebvo_debug_set(self._ctx, key, value), "ebvo_debug_set")

    # -- profiling -----------------------------------------------------------------------------
    def profile_enable(self, on: bool = True, every: int = 1):
        """every = N > 1: bracket only every N-th pair submitted to the device pipeline."""
        self._check(self.lib.ebvo_profile_enable(self._ctx, (max(1, every) if on else 0)), "ebvo_profile_enable")

    def profile_reset(self):
        self._check(self.lib.ebvo_profile_reset(self._ctx), "ebvo_profile_reset")

    def profile_get(self) -> dict:
        arr = (_lib.KernelTime * _lib.MAX_KERNELS)()
        n = C.c_int()
        self._check(self.lib.ebvo_profile_get(self._ctx, arr, C.byref(n)), "ebvo_profile_get")
        return {arr[k].name.decode(): (arr[k].ms, arr[k].launches) for k in range(n.value)}

    def fp64_peak(self, iters: int = 20):
        a, b = C.c_double(), C.c_double()
        self._check(self.lib.ebvo_fp64_peak(self._ctx, iters, C.byref(a), C.byref(b)), "ebvo_fp64_peak")
        return a.value, b.value
