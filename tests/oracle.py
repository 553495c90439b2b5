"""ctypes view of oracle/libebvo_oracle.so -- the CPU restatement of the reference (checker only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.environ.get("EBVO_ORACLE_LIB", os.path.join(ORACLE_DIR, "libebvo_oracle.so"))  # override: sanitizer build

EDGE_DTYPE = np.dtype([("x", "<f8"), ("y", "<f8"), ("theta", "<f8"), ("index", "<i4"), ("pad", "<i4")])
PORTABLE, LIBM = 0, 1
STAGE_EPIPOLAR, STAGE_DISPARITY, STAGE_ORIENTATION, STAGE_ALL = 1, 2, 4, 7

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-C", ORACLE_DIR])
        L = C.CDLL(LIB)
        L.orc_edge_hash.restype = C.c_uint64
        L.orc_fnv1a64.restype = C.c_uint64
        L.orc_patch_similarity.restype = C.c_double
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def toed(img, math_mode=PORTABLE, nthreads=0, want_all=False, want_maps=False):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    cap = h * w
    kept = np.zeros(cap, dtype=EDGE_DTYPE)
    all4 = np.zeros((cap, 4)) if want_all else None
    maps = np.zeros((5, 2 * h, 2 * w)) if want_maps else None
    nk, nt = C.c_int(), C.c_int()
    tc, tn = C.c_double(), C.c_double()
    rc = lib().orc_toed(_p(img), h, w, C.c_ssize_t(img.strides[0]), math_mode, nthreads, _p(kept), cap, _p(all4),
                        cap if want_all else 0, C.byref(nk), C.byref(nt), _p(maps), C.byref(tc), C.byref(tn))
    assert rc == 0, rc
    return dict(edges=kept[: nk.value].copy(), n_total=nt.value, all4=None if all4 is None else all4[: nt.value].copy(),
                maps=maps, t_conv=tc.value, t_nms=tn.value)


def edge_hash(edges, with_theta):
    e = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
    return f"{lib().orc_edge_hash(_p(e), len(e), int(with_theta)):016x}"


def epipolar_lines(F, edges):
    F = np.ascontiguousarray(F, dtype=np.float64).reshape(9)
    e = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
    out = np.zeros((len(e), 3))
    lib().orc_epipolar_lines(_p(F), _p(e), len(e), _p(out))
    return out


def filter_pairs(L, R, row_ptr, col_idx, max_disp=25.0, orient_thr_deg=10.0, stage_mask=2, nthreads=0):
    """a later geometric stage (2 = disparity, 4 = orientation) applied to existing lists: keep flag per listed pair"""
    L = np.ascontiguousarray(L, dtype=EDGE_DTYPE)
    R = np.ascontiguousarray(R, dtype=EDGE_DTYPE)
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
    keep = np.zeros(len(col_idx), dtype=np.uint8)
    rc = lib().orc_filter_pairs(_p(L), len(L), _p(R), _p(row_ptr), _p(col_idx), C.c_double(max_disp), C.c_double(orient_thr_deg),
                                stage_mask, nthreads, _p(keep))
    assert rc == 0
    return keep


def epi_candidates(L, R, lines, epi_thr=0.5, max_disp=25.0, orient_thr_deg=10.0, stage_mask=STAGE_ALL, nthreads=0):
    L = np.ascontiguousarray(L, dtype=EDGE_DTYPE)
    R = np.ascontiguousarray(R, dtype=EDGE_DTYPE)
    lines = np.ascontiguousarray(lines, dtype=np.float64)
    row_ptr = np.zeros(len(L) + 1, dtype=np.int32)
    n = C.c_int64()
    f = lib().orc_epi_candidates
    args = (_p(L), len(L), _p(R), len(R), _p(lines), C.c_double(epi_thr), C.c_double(max_disp),
            C.c_double(orient_thr_deg), stage_mask, nthreads, _p(row_ptr))
    f(*args, None, C.c_int64(0), C.byref(n))
    col = np.zeros(max(1, n.value), dtype=np.int32)
    rc = f(*args, _p(col), C.c_int64(len(col)), C.byref(n))
    assert rc == 0
    return row_ptr, col[: n.value].copy()


def edge_patches(img, edges, math_mode=PORTABLE, nthreads=0):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    e = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
    out = np.zeros((len(e), 2, 49), dtype=np.float32)
    lib().orc_edge_patches(_p(img), h, w, C.c_ssize_t(img.strides[0]), _p(e), len(e), math_mode, nthreads, _p(out))
    return out


def patch_similarity(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(49)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(49)
    return lib().orc_patch_similarity(_p(a), _p(b))


def ncc_patches(A, B, nthreads=0):
    A = np.ascontiguousarray(A, dtype=np.float32).reshape(-1, 49)
    B = np.ascontiguousarray(B, dtype=np.float32).reshape(-1, 49)
    out = np.zeros(len(A))
    lib().orc_ncc_patches(_p(A), _p(B), len(A), nthreads, _p(out))
    return out


def ncc_pairs(imgL, imgR, L, Rc, row_ptr, thr=0.6, math_mode=PORTABLE, nthreads=0):
    imgL = np.ascontiguousarray(imgL, dtype=np.uint8)
    imgR = np.ascontiguousarray(imgR, dtype=np.uint8)
    h, w = imgL.shape
    L = np.ascontiguousarray(L, dtype=EDGE_DTYPE)
    Rc = np.ascontiguousarray(Rc, dtype=EDGE_DTYPE)
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    n = int(row_ptr[-1])
    lp = np.zeros((len(L), 2, 49), dtype=np.float32)
    sims = np.zeros((n, 4))
    best = np.zeros(n)
    keep = np.zeros(n, dtype=np.uint8)
    lib().orc_ncc_pairs(_p(imgL), _p(imgR), h, w, C.c_ssize_t(imgL.strides[0]), C.c_ssize_t(imgR.strides[0]), _p(L),
                        len(L), _p(Rc), _p(row_ptr), math_mode, nthreads, C.c_double(thr), _p(lp), _p(sims), _p(best),
                        _p(keep))
    return sims, best, keep, lp


def sobel_gradients(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    gx = np.zeros((h, w), dtype=np.float32)
    gy = np.zeros((h, w), dtype=np.float32)
    lib().orc_sobel_gradients(_p(img), h, w, C.c_ssize_t(img.strides[0]), _p(gx), _p(gy))
    return gx, gy


def gn_refine_stereo(imgL, imgR, L, lines, row_ptr, cand_xy, max_iter=20, tol=1e-3, huber_delta=3.0, math_mode=PORTABLE,
                     nthreads=0):
    imgL = np.ascontiguousarray(imgL, dtype=np.uint8)
    imgR = np.ascontiguousarray(imgR, dtype=np.uint8)
    h, w = imgL.shape
    L = np.ascontiguousarray(L, dtype=EDGE_DTYPE)
    lines = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 3)
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    cand_xy = np.ascontiguousarray(cand_xy, dtype=np.float64).reshape(-1, 2)
    n = int(row_ptr[-1])
    out = dict(alpha=np.zeros(n), score=np.zeros(n), confidence=np.zeros(n), validity=np.zeros(n, dtype=np.uint8),
               iters=np.zeros(n, dtype=np.int32), refined_xy=np.zeros((n, 2)))
    lib().orc_gn_refine_stereo(_p(imgL), _p(imgR), h, w, C.c_ssize_t(imgL.strides[0]), C.c_ssize_t(imgR.strides[0]),
                               _p(L), _p(lines), _p(row_ptr), len(L), _p(cand_xy), int(max_iter), C.c_double(tol),
                               C.c_double(huber_delta), math_mode, nthreads, _p(out["alpha"]), _p(out["score"]),
                               _p(out["confidence"]), _p(out["validity"]), _p(out["iters"]), _p(out["refined_xy"]))
    return out


def gn_refine_temporal(imgKF, imgCF, kf, cf, init_disp, max_iter=20, tol=1e-3, huber_delta=3.0, math_mode=PORTABLE,
                       nthreads=0):
    imgKF = np.ascontiguousarray(imgKF, dtype=np.uint8)
    imgCF = np.ascontiguousarray(imgCF, dtype=np.uint8)
    h, w = imgKF.shape
    kf = np.ascontiguousarray(kf, dtype=EDGE_DTYPE)
    cf = np.ascontiguousarray(cf, dtype=EDGE_DTYPE)
    init_disp = np.ascontiguousarray(init_disp, dtype=np.float64).reshape(-1, 2)
    n = len(kf)
    out = dict(disp=np.zeros((n, 2)), score=np.zeros(n), validity=np.zeros(n, dtype=np.uint8),
               iters=np.zeros(n, dtype=np.int32))
    lib().orc_gn_refine_temporal(_p(imgKF), _p(imgCF), h, w, C.c_ssize_t(imgKF.strides[0]),
                                 C.c_ssize_t(imgCF.strides[0]), _p(kf), _p(cf), _p(init_disp), n, int(max_iter),
                                 C.c_double(tol), C.c_double(huber_delta), math_mode, nthreads, _p(out["disp"]),
                                 _p(out["score"]), _p(out["validity"]), _p(out["iters"]))
    return out


def bnb_test(row_ptr, scores, ratio_thr, higher_is_better=True, always_sorted=False):
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    cnt = np.zeros(len(row_ptr) - 1, dtype=np.int32)
    order = np.full(len(scores), -1, dtype=np.int32)
    lib().orc_bnb_test(_p(row_ptr), len(row_ptr) - 1, _p(scores), C.c_double(ratio_thr), int(higher_is_better) | (2 if always_sorted else 0), _p(cnt),
                       _p(order))
    return cnt, order


def keep_best(row_ptr, scores):
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    cnt = np.zeros(len(row_ptr) - 1, dtype=np.int32)
    order = np.full(len(scores), -1, dtype=np.int32)
    lib().orc_keep_best(_p(row_ptr), len(row_ptr) - 1, _p(scores), _p(cnt), _p(order))
    return cnt, order


def epipolar_shift(cand, lines, row_ptr, math_mode=PORTABLE):
    cand = np.ascontiguousarray(cand, dtype=EDGE_DTYPE)
    lines = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 3)
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    out = np.zeros(len(cand), dtype=EDGE_DTYPE)
    lib().orc_epipolar_shift(_p(cand), _p(lines), _p(row_ptr), len(row_ptr) - 1, math_mode, _p(out))
    return out


def cluster_rows(cand, row_ptr, by_orientation=False, skip_single=True, math_mode=PORTABLE):
    cand = np.ascontiguousarray(cand, dtype=EDGE_DTYPE)
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    cnt = np.zeros(len(row_ptr) - 1, dtype=np.int32)
    centres = np.zeros(len(cand), dtype=EDGE_DTYPE)
    cluster_of = np.full(len(cand), -1, dtype=np.int32)
    lib().orc_cluster_rows(_p(cand), _p(row_ptr), len(row_ptr) - 1, int(by_orientation), int(skip_single), math_mode, _p(cnt),
                           _p(centres), _p(cluster_of))
    return cnt, centres, cluster_of


def finalize_pairs(K_left, K_right, R21, T21, left, right, math_mode=PORTABLE):
    arrs = [np.ascontiguousarray(a, dtype=np.float64).reshape(-1) for a in (K_left, K_right, R21, T21)]
    left = np.ascontiguousarray(left, dtype=EDGE_DTYPE)
    right = np.ascontiguousarray(right, dtype=EDGE_DTYPE)
    out = np.zeros((len(left), 16))
    lib().orc_finalize_pairs(*[_p(a) for a in arrs], _p(left), _p(right), len(left), math_mode, _p(out))
    return out


def ncc_quads(kfL, kfR, cfL, cfR, thr=0.8, nthreads=0):
    arrs = [np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 98) for a in (kfL, kfR, cfL, cfR)]
    n = len(arrs[0])
    sl, sr = np.zeros(n), np.zeros(n)
    keep = np.zeros(n, dtype=np.uint8)
    lib().orc_ncc_quads(*[_p(a) for a in arrs], n, nthreads, C.c_double(thr), _p(sl), _p(sr), _p(keep))
    return sl, sr, keep


def atan2_v(y, x, math_mode):
    y = np.ascontiguousarray(y, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(len(y))
    lib().orc_atan2_v(_p(y), _p(x), len(y), math_mode, _p(out))
    return out


def sincos_v(t, math_mode):
    t = np.ascontiguousarray(t, dtype=np.float64)
    s, c = np.zeros(len(t)), np.zeros(len(t))
    lib().orc_sincos_v(_p(t), len(t), math_mode, _p(s), _p(c))
    return s, c


def exp_v(x, math_mode):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(len(x))
    lib().orc_exp_v(_p(x), len(x), math_mode, _p(out))
    return out


def undistort(img, K, dist):
    """cv::undistort(img, K, dist) restated (oracle/ebvo_oracle.c: orc_undistort).  K = (fx, fy, cx, cy)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(4)
    dist = np.ascontiguousarray(dist, dtype=np.float64).reshape(-1)
    out = np.zeros_like(img)
    lib().orc_undistort(_p(img), h, w, C.c_ssize_t(img.strides[0]), _p(K), _p(dist), len(dist), _p(out),
                        C.c_ssize_t(out.strides[0]))
    return out


def sift_base(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), dtype=np.float32)
    lib().orc_sift_base(_p(img), h, w, C.c_ssize_t(img.strides[0]), _p(out))
    return out


def sift_descriptors(img, edges, math_mode=PORTABLE, nthreads=0):
    """cv::SIFT descriptors at the +-8 px points of every edge: (n, 2, 128) float32, values 0..255."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    e = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
    out = np.zeros((len(e), 2, 128), dtype=np.float32)
    lib().orc_sift_descriptors(_p(img), h, w, C.c_ssize_t(img.strides[0]), _p(e), len(e), math_mode, nthreads, _p(out))
    return out


def sift_min_distances(left_desc, cand_desc, row_ptr):
    left_desc = np.ascontiguousarray(left_desc, dtype=np.float32).reshape(-1, 2, 128)
    cand_desc = np.ascontiguousarray(cand_desc, dtype=np.float32).reshape(-1, 2, 128)
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    out = np.zeros(len(cand_desc))
    lib().orc_sift_min_distances(_p(left_desc), _p(cand_desc), _p(row_ptr), len(row_ptr) - 1, _p(out))
    return out


def temporal_candidates(kfL, kfR, cfL, cfR, img_w, img_h, cell=15, radius=30.0, orient_thr_deg=10.0):
    arrs = [np.ascontiguousarray(a, dtype=EDGE_DTYPE) for a in (kfL, kfR, cfL, cfR)]
    n_kf, n_cf = len(arrs[0]), len(arrs[2])
    row_ptr = np.zeros(n_kf + 1, dtype=np.int32)
    n = C.c_int64()
    f = lib().orc_temporal_candidates
    args = (_p(arrs[0]), _p(arrs[1]), n_kf, _p(arrs[2]), _p(arrs[3]), n_cf, img_w, img_h, cell, C.c_double(radius),
            C.c_double(orient_thr_deg), _p(row_ptr))
    f(*args, None, C.c_int64(0), C.byref(n))
    col = np.zeros(max(1, n.value), dtype=np.int32)
    assert f(*args, _p(col), C.c_int64(len(col)), C.byref(n)) == 0
    return row_ptr, col[: n.value].copy()
