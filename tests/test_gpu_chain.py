"""ebvo_stereo_finalize (BNB -> shift -> refine -> cluster -> NCC -> best on the resident pair) equals the same stages
chained through the host-buffer entry points on the fetched data -- each of which is tested against the oracle."""
import numpy as np
import pytest

from edge_based_visual_odometry_amd import synth
from tests import oracle as orc
from tests import oracle_chain
from tests.util import assert_bit_equal, assert_edges_equal

pytestmark = pytest.mark.gpu

F_KITTI = synth.fundamental_for("kitti")


def _select(rp, cnt, order, *arrays):
    """Apply a (new_count, order) selection to per-pair arrays: new row_ptr + gathered arrays."""
    idx = np.concatenate([order[rp[i]:rp[i] + cnt[i]] for i in range(len(cnt))]) if cnt.sum() else np.zeros(0, np.int64)
    new_rp = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    return new_rp, [a[idx] for a in arrays]


def _host_chain(ctx, l, r, o, calib, F=F_KITTI):
    L, R = o["left"], o["right"]
    nL = len(L)
    lines = ctx.epipolar_lines(F, L)
    keep = o["keep"].astype(bool)
    rows = np.repeat(np.arange(nL), np.diff(o["row_ptr"]))[keep]
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=nL))]).astype(np.int32)
    cand = R[o["col_idx"][keep]].copy()
    cand["index"] = 0
    score = o["best"][keep]
    counts = dict(n_ncc=len(cand))
    cnt, order = ctx.bnb_test(rp, score, 0.9, True)
    rp, (cand,) = _select(rp, cnt, order, cand)
    counts["n_bnb"] = len(cand)
    cand = ctx.epipolar_shift(cand, lines, rp)
    ref = ctx.gn_refine_stereo(l, r, L, lines, rp, np.stack([cand["x"], cand["y"]], 1))
    cand["x"], cand["y"] = ref["refined_xy"].T
    cand = ctx.epipolar_shift(cand, lines, rp)                       # :1483 binds shift = true, cluster = true
    cnt, centres, _ = ctx.cluster_rows(cand, rp, True, False)
    rp2 = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    cand = np.concatenate([centres[rp[i]:rp[i] + cnt[i]] for i in range(nL)]) if cnt.sum() else cand[:0]
    rp = rp2
    counts["n_clusters"] = len(cand)
    _, best, keep2, _ = ctx.ncc_pairs(l, r, L, cand, rp, 0.6)
    rows = np.repeat(np.arange(nL), np.diff(rp))[keep2.astype(bool)]
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=nL))]).astype(np.int32)
    cand, best = cand[keep2.astype(bool)], best[keep2.astype(bool)]
    counts["n_ncc2"] = len(cand)
    cnt, order = ctx.keep_best(rp, best)
    sel = [order[rp[i]] for i in range(nL) if cnt[i]]
    left_index = np.array([i for i in range(nL) if cnt[i]], dtype=np.int32)
    right, score = cand[sel], best[sel]
    counts["n_final"] = len(right)
    rows16 = ctx.finalize_pairs(*calib, L[left_index], right)
    return counts, left_index, right, score, rows16


@pytest.mark.parametrize("shape", [(96, 160), (200, 320)])
def test_device_chain_equals_chained_entry_points(ctx, shape):
    from edge_based_visual_odometry_amd._lib import EbvoError, EBVO_ERR_STATE
    l, r = synth.stereo_pair("s2", *shape)
    cal = synth.CALIB["kitti"]
    fx, fy, cx, cy = cal["K"]
    K = [fx, 0, cx, 0, fy, cy, 0, 0, 1]
    calib = (K, K, cal["R21"], cal["T21"])
    ctx.stereo_upload(l, r)
    with pytest.raises(EbvoError) as ei:
        ctx.stereo_finalize(calib)                                   # nothing has run yet
    assert ei.value.status == EBVO_ERR_STATE
    c = ctx.stereo_run(ctx.default_params(F_KITTI))
    counts, fin = ctx.stereo_finalize(calib)
    o = ctx.stereo_fetch(c)                                          # the run's own results are still intact
    assert c.n_matches == counts["n_ncc"] == int(o["keep"].sum())
    hc, left_index, right, score, rows16 = _host_chain(ctx, l, r, o, calib)
    assert counts == hc
    assert counts["n_ncc"] > counts["n_bnb"] >= counts["n_clusters"] >= counts["n_ncc2"] >= counts["n_final"] > 0
    assert_bit_equal(fin["left_index"], left_index, "left_index")
    assert_edges_equal(fin["right"], right, "right centre")
    assert_bit_equal(fin["score"], score, "score")
    assert_bit_equal(fin["rows"], rows16, "rows")
    # ... and the chain of ORACLE functions (tests/oracle_chain.py), bit for bit
    ref = oracle_chain.stereo_edge_pairs(l, r, F_KITTI, calib)
    assert counts == ref["counts"]
    assert_bit_equal(fin["left_index"], ref["left_index"], "left_index vs oracle chain")
    assert_edges_equal(fin["right"], ref["right"], "right centre vs oracle chain")
    assert_bit_equal(fin["score"], ref["score"], "score vs oracle chain")
    assert_bit_equal(fin["rows"], ref["rows"], "rows vs oracle chain")
    # the matches are the generator's disparity
    d = o["left"]["x"][fin["left_index"]] - fin["right"]["x"]
    assert np.median(np.abs(d - 12.0)) < 0.1
    # without calibration: pairs only
    ctx.stereo_upload(l, r)
    ctx.stereo_run(ctx.default_params(F_KITTI))
    counts2, fin2 = ctx.stereo_finalize(None)
    assert counts2 == counts and "rows" not in fin2
    assert_edges_equal(fin2["right"], right)


def test_device_chain_slanted_epipolar_lines(ctx):
    """EuRoC calibration (non-rectified): the shift, refinement and output rows run along slanted epipolar lines."""
    F = synth.fundamental_for("euroc")
    ce = synth.CALIB["euroc"]
    Kl = [ce["K"][0], 0, ce["K"][2], 0, ce["K"][1], ce["K"][3], 0, 0, 1]
    Kr = [ce["K_right"][0], 0, ce["K_right"][2], 0, ce["K_right"][1], ce["K_right"][3], 0, 0, 1]
    calib = (Kl, Kr, ce["R21"], ce["T21"])
    l, r = synth.stereo_pair("s2", 160, 240)
    ctx.stereo_upload(l, r)
    c = ctx.stereo_run(ctx.default_params(F))
    counts, fin = ctx.stereo_finalize(calib)
    o = ctx.stereo_fetch(c)
    hc, left_index, right, score, rows16 = _host_chain(ctx, l, r, o, calib, F)
    assert counts == hc and counts["n_final"] > 0
    assert_bit_equal(fin["left_index"], left_index)
    assert_edges_equal(fin["right"], right)
    assert_bit_equal(fin["score"], score)
    assert_bit_equal(fin["rows"], rows16)
