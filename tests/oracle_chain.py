"""Oracle-side chain: Stereo_Matches::get_Stereo_Edge_Pairs (src/Stereo_Matches.cpp:1360-1540) composed from the
functions of the CPU restatement (oracle/ebvo_oracle.c), stage for stage, plus the rows of the output file
(:1656-1699).  Test infrastructure: the device-resident chain (ebvo_stereo_finalize) is compared with THIS, not with
the HIP entry points.

The SIFT filter and the BNB test on SIFT distances (:1410-1414, :1452) are included with sift=True (the oracle's restated
cv::SIFT descriptors); without it the chain is the reference's with those two stages left out, which is what
ebvo_stereo_finalize computes with use_sift = 0.
"""
from __future__ import annotations

import numpy as np

from tests import oracle as orc


def csr_select(rp, cnt, order=None):
    """Indices (into the pair arrays of CSR `rp`) of the first cnt[i] entries of order[rp[i]:] per row, and the new
    row_ptr.  order None: identity (the first cnt[i] pairs of each row)."""
    rp = np.asarray(rp, dtype=np.int64)
    cnt = np.asarray(cnt, dtype=np.int64)
    new_rp = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    total = int(new_rp[-1])
    if total == 0:
        return np.zeros(0, dtype=np.int64), new_rp
    pos = np.repeat(rp[:-1], cnt) + (np.arange(total) - np.repeat(new_rp[:-1].astype(np.int64), cnt))
    return (pos if order is None else np.asarray(order, dtype=np.int64)[pos]), new_rp


def rows_of(rp):
    rp = np.asarray(rp, dtype=np.int64)
    return np.repeat(np.arange(len(rp) - 1), np.diff(rp))


def filter_rows(rp, mask):
    """CSR after dropping the pairs with mask == 0 (order kept): new row_ptr."""
    rows = rows_of(rp)[mask]
    return np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=len(rp) - 1))]).astype(np.int32)


def stereo_edge_pairs(left_img, right_img, F, calib=None, bnb_ratio=0.9, ncc_thr=0.6, stage1=None, right_img_undist=None,
                      left_img_undist=None, sift=False, sift_thr=500.0, bnb_sift=0.4, cluster_args=(True, False)):
    """Runs the whole chain on the oracle.  stage1 = dict(left, right, row_ptr, col_idx, best, keep) may carry the
    results of TOED + candidates + first NCC pass if the caller already has them (they are oracle outputs too).
    left/right_img_undist: the undistorted images TOED and the refinement run on (None: the same as the raw ones, which
    is the KITTI / ETH3D case: zero distortion)."""
    lu = left_img if left_img_undist is None else left_img_undist
    ru = right_img if right_img_undist is None else right_img_undist
    if stage1 is None:
        L = orc.toed(lu)["edges"]
        R = orc.toed(ru)["edges"]
        lines = orc.epipolar_lines(F, L)
        rp, ci = orc.epi_candidates(L, R, lines)                          # :1374, :1387, :1399
        sims, best, keep, _ = orc.ncc_pairs(left_img, right_img, L, R[ci], rp, ncc_thr)   # :1427 (raw images, :562)
    else:
        L, R, rp, ci, best, keep = (stage1[k] for k in ("left", "right", "row_ptr", "col_idx", "best", "keep"))
        lines = orc.epipolar_lines(F, L)
    nL = len(L)
    counts = {}
    conf = None
    if sift:
        # augment_Edge_Data (:1410) + apply_SIFT_filtering (:1414), on the UNDISTORTED images: pairs whose smallest
        # descriptor distance is >= sift_thr are dropped before the NCC filter; the distance is carried as
        # refine_confidences (:757, :602).  NCC scores are per pair, so masking the already scored pairs is the same.
        dl, dr = orc.sift_descriptors(lu, L), orc.sift_descriptors(ru, R)
        d = orc.sift_min_distances(dl, dr[ci], rp)
        ok = d < sift_thr
        counts["n_sift"] = int(ok.sum())
        keep = (keep.astype(bool) & ok).astype(np.uint8)
        conf = d
    k = keep.astype(bool)
    cand = R[ci[k]].copy()
    cand["index"] = 0
    score = best[k]
    rp = filter_rows(rp, k)
    if conf is not None:
        conf = conf[k]
    counts["n_ncc"] = len(cand)
    # Best-Nearly-Best on NCC (:1440), then on the SIFT distances (:1452)
    cnt, order = orc.bnb_test(rp, score, bnb_ratio, True)
    idx, rp = csr_select(rp, cnt, order)
    cand, score = cand[idx], score[idx]
    if conf is not None:
        conf = conf[idx]
        cnt, order = orc.bnb_test(rp, conf, bnb_sift, False)
        idx, rp = csr_select(rp, cnt, order)
        cand, score, conf = cand[idx], score[idx], conf[idx]
    counts["n_bnb"] = len(cand)
    # shift to the epipolar line (:1465), refine along it (:1468)
    cand = orc.epipolar_shift(cand, lines, rp)
    ref = orc.gn_refine_stereo(lu, ru, L, lines, rp, np.stack([cand["x"], cand["y"]], 1))
    cand = cand.copy()
    cand["x"], cand["y"] = ref["refined_xy"][:, 0], ref["refined_xy"][:, 1]
    # shift again and cluster by orientation, single-candidate rows included (:1483 as the arguments bind)
    # (cluster_args = (False, True) is the cluster-only reading of that call, kept to show that a fixture tells them apart)
    by_orientation, skip_single = cluster_args
    if by_orientation:
        cand = orc.epipolar_shift(cand, lines, rp)
    cnt, centres, _ = orc.cluster_rows(cand, rp, by_orientation, skip_single)
    idx, rp = csr_select(rp, cnt, None)
    cand = centres[idx]
    counts["n_clusters"] = len(cand)
    # second NCC pass on the cluster centres (:1500), best survivor per row (:1513), rows with a match (:1526)
    _, best2, keep2, _ = orc.ncc_pairs(left_img, right_img, L, cand, rp, ncc_thr)
    k2 = keep2.astype(bool)
    rp = filter_rows(rp, k2)
    cand, best2 = cand[k2], best2[k2]
    counts["n_ncc2"] = len(cand)
    cnt, order = orc.keep_best(rp, best2)
    idx, _ = csr_select(rp, cnt, order)
    left_index = np.flatnonzero(np.asarray(cnt) > 0).astype(np.int32)
    right, fscore = cand[idx], best2[idx]
    counts["n_final"] = len(right)
    rows16 = orc.finalize_pairs(*calib, L[left_index], right) if calib is not None else None
    return dict(counts=counts, left_index=left_index, right=right, score=fscore, rows=rows16, left=L, right_edges=R,
                lines=lines)
