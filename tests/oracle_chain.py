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


def _select_rows(row_ptr, cnt, order):
    """(row_ptr of the selection, source index of every selected pair): order[row_ptr[i] + k], k < cnt[i]"""
    rp = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    idx = np.concatenate([order[row_ptr[i]:row_ptr[i] + cnt[i]] for i in range(len(cnt))] or [np.zeros(0, np.int32)]).astype(np.int64)
    return rp, idx


def temporal_edge_pairs(kfL, kfR, cfL, cfR, kf_imgs, cf_imgs, rp, ci, sim_left, keep, sift_thr=200.0, bnb_ncc=0.8,
                        bnb_sift=0.8, max_iter=20, tol=1e-3, huber=3.0):
    """Temporal_Matches::get_Temporal_Edge_Pairs_from_Quads after the NCC filter (src/Temporal_Matches.cpp:196-215) as a chain
    of oracle functions.  kfL / kfR / cfL / cfR: the stereo mates (left TOED edge, final right edge) of the keyframe and of
    the current frame; kf_imgs / cf_imgs = (undistorted left, undistorted right); rp / ci: candidate quads per keyframe mate;
    sim_left / keep: their left NCC maximum and the NCC keep flag."""
    n_kf = len(kfL)
    # SIFT descriptors of the mates: left edge on the undistorted left image, final right edge on the undistorted right one
    dkL, dkR = orc.sift_descriptors(kf_imgs[0], kfL), orc.sift_descriptors(kf_imgs[1], kfR)
    dcL, dcR = orc.sift_descriptors(cf_imgs[0], cfL), orc.sift_descriptors(cf_imgs[1], cfR)
    # the quads that passed the NCC filter
    cnt = np.array([int(keep[rp[i]:rp[i + 1]].sum()) for i in range(n_kf)], dtype=np.int32)
    order = np.full(len(ci), -1, dtype=np.int32)
    for i in range(n_kf):
        k = np.flatnonzero(keep[rp[i]:rp[i + 1]]) + rp[i]
        order[rp[i]:rp[i] + len(k)] = k
    rp0, idx = _select_rows(rp, cnt, order)
    cf, simL = ci[idx], sim_left[idx]
    counts = {}
    # apply_SIFT_filtering_quads (:471-515)
    siftL, siftR = orc.sift_min_distances(dkL, dcL[cf], rp0), orc.sift_min_distances(dkR, dcR[cf], rp0)
    ok = (siftL < sift_thr) & (siftR < sift_thr)
    cnt = np.array([int(ok[rp0[i]:rp0[i + 1]].sum()) for i in range(n_kf)], dtype=np.int32)
    order = np.full(len(cf), -1, dtype=np.int32)
    for i in range(n_kf):
        k = np.flatnonzero(ok[rp0[i]:rp0[i + 1]]) + rp0[i]
        order[rp0[i]:rp0[i] + len(k)] = k
    rp1, idx = _select_rows(rp0, cnt, order)
    cf, simL, siftL, siftR = cf[idx], simL[idx], siftL[idx], siftR[idx]
    counts["n_sift"] = len(cf)
    # apply_best_nearly_best_filtering_quads on the NCC, then on the SIFT scores (:517-570)
    for name, thr, scores_of, higher in (("n_bnb_ncc", bnb_ncc, lambda: simL, True), ("n_bnb_sift", bnb_sift, lambda: siftL, False)):
        c, o = orc.bnb_test(rp1, scores_of(), thr, higher, always_sorted=True)
        rp1, idx = _select_rows(rp1, c, o)
        cf, simL, siftL, siftR = cf[idx], simL[idx], siftL[idx], siftR[idx]
        counts[name] = len(cf)
    kf = np.repeat(np.arange(n_kf), np.diff(rp1))
    # apply_photometric_refinement_quads (:572-634)
    out = {}
    cen = {}
    for cam, kE, cE, kimg, cimg in (("L", kfL, cfL, kf_imgs[0], cf_imgs[0]), ("R", kfR, cfR, kf_imgs[1], cf_imgs[1])):
        ke, ce = kE[kf], cE[cf]
        init = np.stack([ke["x"] - ce["x"], ke["y"] - ce["y"]], 1)
        r = orc.gn_refine_temporal(kimg, cimg, ke, ce, init, max_iter, tol, huber)
        c = ce.copy()
        v = r["validity"] == 1
        c["x"][v] = ke["x"][v] - r["disp"][v, 0]
        c["y"][v] = ke["y"][v] - r["disp"][v, 1]
        out[cam], cen[cam] = r, c
    valid = ((out["L"]["validity"] == 1) & (out["R"]["validity"] == 1)).astype(np.uint8)
    counts["n_refined_valid"] = int(valid.sum())
    # apply_temporal_edge_clustering_quads (:636-733)
    ncl, centres, cluster_of = orc.cluster_rows(cen["L"], rp1, True, True)
    f_rows, f_src, f_L, f_R = [0], [], [], []
    for i in range(n_kf):
        b, n = rp1[i], rp1[i + 1] - rp1[i]
        if n < 2:
            for k in range(n):
                f_src.append(b)
                f_L.append(cen["L"][b])
                f_R.append(cen["R"][b])
            f_rows.append(len(f_src))
            continue
        loc = np.stack([cen["L"]["x"][b:b + n], cen["L"]["y"][b:b + n]], 1)
        for c in range(ncl[i]):
            members = [m for m in range(n) if cluster_of[b + m] == c]
            rights, best = [], -1
            for m in members:
                d = np.sqrt((loc[m, 0] - loc[:, 0]) * (loc[m, 0] - loc[:, 0]) + (loc[m, 1] - loc[:, 1]) * (loc[m, 1] - loc[:, 1]))
                closest = int(np.argmin(d))                              # the first smallest: `d < closest_dist`
                rights.append(cen["R"][b + closest])
                best = closest
            right = rights[0].copy()
            if len(rights) > 1:
                sx = sy = st = 0.0
                for e in rights:
                    sx += float(e["x"])
                    sy += float(e["y"])
                    st += float(e["theta"])
                right["x"], right["y"], right["theta"] = sx / len(rights), sy / len(rights), st / len(rights)
            f_src.append(b + best)
            f_L.append(centres[b + c])
            f_R.append(right)
        f_rows.append(len(f_src))
    f_src = np.array(f_src, dtype=np.int64)
    counts["n_final"] = len(f_src)
    return dict(counts=counts, row_ptr=np.array(f_rows, dtype=np.int32), cf_index=cf[f_src].astype(np.int32),
                left=np.array(f_L, dtype=orc.EDGE_DTYPE) if f_L else np.zeros(0, orc.EDGE_DTYPE),
                right=np.array(f_R, dtype=orc.EDGE_DTYPE) if f_R else np.zeros(0, orc.EDGE_DTYPE), ncc_left=simL[f_src],
                sift_left=siftL[f_src], score_left=out["L"]["score"][f_src], score_right=out["R"]["score"][f_src],
                valid=valid[f_src])


def temporal_reference(kfL, kfR, cfL, cfR, kf_imgs, cf_imgs, w, h, chain=True, ncc_thr=0.8):
    """One frame of Temporal_Matches::get_Temporal_Edge_Pairs_from_Quads on the oracle, from the stereo mates of the keyframe
    and of the current frame (src/Temporal_Matches.cpp:168-218): grid + orientation candidates (:335-414), NCC on the
    stored patches (:416-469), and with chain=True every later stage (temporal_edge_pairs above).
    kf_imgs / cf_imgs = (RAW left, undistorted left, undistorted right): the left patches of a mate are sampled from the raw
    left image (src/Stereo_Matches.cpp:562), the right ones from the undistorted right image (:1580-1582)."""
    rp, ci = orc.temporal_candidates(kfL, kfR, cfL, cfR, w, h)
    pkL, pkR = orc.edge_patches(kf_imgs[0], kfL), orc.edge_patches(kf_imgs[2], kfR)
    pcL, pcR = orc.edge_patches(cf_imgs[0], cfL), orc.edge_patches(cf_imgs[2], cfR)
    rows = rows_of(rp)
    sl, sr, keep = orc.ncc_quads(pkL[rows], pkR[rows], pcL[ci], pcR[ci], ncc_thr)
    ref = dict(row_ptr=rp, col_idx=ci, sim_left=sl, sim_right=sr, keep=keep,
               counts=dict(n_kf=len(kfL), n_cf=len(cfL), n_candidates=len(ci), n_kept=int(keep.sum())))
    if chain:
        ref["final"] = temporal_edge_pairs(kfL, kfR, cfL, cfR, kf_imgs[1:], cf_imgs[1:], rp, ci, sl, keep)
        ref["counts"].update(ref["final"]["counts"])
    return ref


def _same(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype.kind == "f":
        ua = a.view(np.uint64 if a.dtype == np.float64 else np.uint32)
        ub = b.view(np.uint64 if b.dtype == np.float64 else np.uint32)
        return bool(((ua == ub) | (np.isnan(a) & np.isnan(b))).all())
    return bool((a == b).all())


def _same_edges(a, b):
    return len(a) == len(b) and all(_same(a[f].copy(), b[f].copy()) for f in ("x", "y", "theta")) and bool((a["index"] == b["index"]).all())


def temporal_problems(counts, q, ref):
    """What of ebvo_temporal_match's results (counts, the arrays of ebvo_temporal_fetch / _fetch_final as api.temporal_match
    returns them) differs from temporal_reference's -- every quad, bit for bit.  Empty list = parity."""
    bad = []
    for k, v in ref["counts"].items():
        if counts.get(k) != v:
            bad.append(f"count {k}: {counts.get(k)} != oracle {v}")
    for k in ("row_ptr", "col_idx", "sim_left", "sim_right", "keep"):
        if not _same(q[k], ref[k]):
            bad.append(f"{k} differs from the oracle")
    if "final" in ref:
        fin, rf = q["final"], ref["final"]
        for k in ("row_ptr", "cf_index", "ncc_left", "sift_left", "score_left", "score_right", "valid"):
            if not _same(fin[k], rf[k]):
                bad.append(f"final.{k} differs from the oracle")
        for k in ("left", "right"):
            if not _same_edges(fin[k], rf[k]):
                bad.append(f"final.{k} centres differ from the oracle")
    return bad
