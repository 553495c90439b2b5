"""Temporal quads on the resident pairs (ebvo_temporal_set_keyframe / _match): candidate lists against the brute-force
restatement of the grid + orientation filters (src/Temporal_Matches.cpp:335-414), and the NCC of every candidate quad on
the mates' stored patches against the oracle's apply_NCC_filtering_quads (:416-469) -- every quad, bit for bit."""
import numpy as np
import pytest

from edge_based_visual_odometry_amd import synth
from tests import oracle as orc
from tests.util import assert_bit_equal, assert_edges_equal

pytestmark = pytest.mark.gpu


def _frame(ctx, h, w, k, F, calib, undist=None):
    """frame k of SURVEY 8(d) config 3: scene 7, noise seeds (2k + 1, 2k + 2), global shift of k px"""
    l, r = synth.stereo_pair("s2", h, w, scene=7, noise_base=2 * k, disparity=9)
    l, r = np.roll(l, k, axis=1), np.roll(r, k, axis=1)
    ctx.stereo_upload(l, r)
    ctx.stereo_run(ctx.default_params(F))
    counts, fin = ctx.stereo_finalize(calib)
    return l, r, fin


@pytest.mark.parametrize("undist", [False, True])
def test_temporal_quads_equal_oracle(ctx, undist):
    h, w = 240, 376
    ce = synth.CALIB["euroc"]
    K = tuple(v / 2 for v in ce["K"])
    Kr = tuple(v / 2 for v in ce["K_right"])
    F = synth.fundamental_21(K, Kr, ce["R21"], ce["T21"])
    calib = ([K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1], [Kr[0], 0, Kr[2], 0, Kr[1], Kr[3], 0, 0, 1], ce["R21"], ce["T21"])
    if undist:
        ctx.set_undistort(K, ce["dist"], Kr, ce["dist_right"])
    try:
        l0, r0, kf = _frame(ctx, h, w, 0, F, calib)
        ctx.temporal_set_keyframe()
        # keyframe mates as the host sees them: left TOED edge of every final pair + its right centre
        ctx.stereo_upload(l0, r0)
        c0 = ctx.stereo_run(ctx.default_params(F))
        left0 = ctx.stereo_fetch(c0)["left"]
        _, kf = ctx.stereo_finalize(calib)
        kfL, kfR = left0[kf["left_index"]], kf["right"]
        l3, r3, _ = _frame(ctx, h, w, 3, F, calib)
        c3 = ctx.stereo_run(ctx.default_params(F))
        left3 = ctx.stereo_fetch(c3)["left"]
        _, cf = ctx.stereo_finalize(calib)
        cfL, cfR = left3[cf["left_index"]], cf["right"]
        counts, q = ctx.temporal_match()
    finally:
        if undist:
            ctx.set_undistort()
    assert counts["n_kf"] == len(kfL) and counts["n_cf"] == len(cfL) and counts["n_kf"] > 1000
    orp, oci = orc.temporal_candidates(kfL, kfR, cfL, cfR, w, h)
    assert_bit_equal(q["row_ptr"], orp, "row_ptr")
    assert_bit_equal(q["col_idx"], oci, "col_idx")
    assert counts["n_candidates"] == len(oci) > 10 * counts["n_kf"]
    # stored patches: left from the RAW left image, right from the UNDISTORTED right image
    ru0 = orc.undistort(r0, Kr, ce["dist_right"]) if undist else r0
    ru3 = orc.undistort(r3, Kr, ce["dist_right"]) if undist else r3
    pkL, pkR = orc.edge_patches(l0, kfL), orc.edge_patches(ru0, kfR)
    pcL, pcR = orc.edge_patches(l3, cfL), orc.edge_patches(ru3, cfR)
    rows = np.repeat(np.arange(len(kfL)), np.diff(orp))
    sl, sr, keep = orc.ncc_quads(pkL[rows], pkR[rows], pcL[oci], pcR[oci], 0.8)
    assert_bit_equal(q["sim_left"], sl, "sim_left")
    assert_bit_equal(q["sim_right"], sr, "sim_right")
    assert_bit_equal(q["keep"], keep, "keep")
    assert counts["n_kept"] == int(keep.sum()) > 100
    # the scene moved by 3 px: the kept quads are that motion
    dx = cfL["x"][oci[keep.astype(bool)]] - kfL["x"][rows[keep.astype(bool)]]
    assert abs(np.median(dx) - 3.0) < 1.0


@pytest.mark.parametrize("undist", [False, True])
def test_temporal_chain_equals_oracle_chain(ctx, undist):
    """get_Temporal_Edge_Pairs_from_Quads stage for stage (src/Temporal_Matches.cpp:184-215): grid + orientation candidates,
    NCC, SIFT filter, Best-Nearly-Best on the NCC and on the SIFT scores, photometric refinement of both cameras, edge
    clustering -- the device chain against the chain of oracle functions (tests/oracle_chain.py), every final quad."""
    from tests import oracle_chain
    h, w = 240, 376
    ce = synth.CALIB["euroc"]
    K = tuple(v / 2 for v in ce["K"])
    Kr = tuple(v / 2 for v in ce["K_right"])
    F = synth.fundamental_21(K, Kr, ce["R21"], ce["T21"])
    calib = ([K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1], [Kr[0], 0, Kr[2], 0, Kr[1], Kr[3], 0, 0, 1], ce["R21"], ce["T21"])
    if undist:
        ctx.set_undistort(K, ce["dist"], Kr, ce["dist_right"])
    try:
        l0, r0, _ = _frame(ctx, h, w, 0, F, calib)
        c0 = ctx.stereo_run(ctx.default_params(F))
        left0 = ctx.stereo_fetch(c0)["left"]
        _, kf = ctx.stereo_finalize(calib, use_sift=not undist)   # with the SIFT stages the mates' left descriptors are
        ctx.temporal_set_keyframe()                               # picked from the chain's, otherwise computed: same bits
        kfL, kfR = left0[kf["left_index"]], kf["right"]
        l3, r3, _ = _frame(ctx, h, w, 2, F, calib)
        c3 = ctx.stereo_run(ctx.default_params(F))
        left3 = ctx.stereo_fetch(c3)["left"]
        _, cf = ctx.stereo_finalize(calib, use_sift=not undist)
        cfL, cfR = left3[cf["left_index"]], cf["right"]
        counts, q = ctx.temporal_match(stages=1)
    finally:
        if undist:
            ctx.set_undistort()
    und = (lambda img, k, d: orc.undistort(img, k, d)) if undist else (lambda img, k, d: img)
    kf_imgs = (und(l0, K, ce["dist"]), und(r0, Kr, ce["dist_right"]))
    cf_imgs = (und(l3, K, ce["dist"]), und(r3, Kr, ce["dist_right"]))
    ref = oracle_chain.temporal_edge_pairs(kfL, kfR, cfL, cfR, kf_imgs, cf_imgs, q["row_ptr"], q["col_idx"], q["sim_left"],
                                           q["keep"])
    got = {k: counts[k] for k in ref["counts"]}
    assert got == ref["counts"] and counts["n_final"] > 300 and counts["n_sift"] < counts["n_kept"]
    fin = q["final"]
    assert_bit_equal(fin["row_ptr"], ref["row_ptr"], "row_ptr")
    assert_bit_equal(fin["cf_index"], ref["cf_index"], "cf_index")
    assert_edges_equal(fin["left"], ref["left"], "left centres")
    assert_edges_equal(fin["right"], ref["right"], "right centres")
    for k in ("ncc_left", "sift_left", "score_left", "score_right", "valid"):
        assert_bit_equal(fin[k], ref[k], k)
    # the scene moved by 2 px between the keyframe and this frame
    rows = np.repeat(np.arange(len(kfL)), np.diff(fin["row_ptr"]))
    v = fin["valid"].astype(bool)
    assert v.mean() > 0.5 and abs(np.median(fin["left"]["x"][v] - kfL["x"][rows[v]]) - 2.0) < 0.5


def test_temporal_state_machine(ctx):
    from edge_based_visual_odometry_amd._lib import EbvoError, EBVO_ERR_STATE
    l, r = synth.stereo_pair("s2", 96, 160)
    F = synth.fundamental_for("kitti")
    ctx.stereo_upload(l, r)
    ctx.stereo_run(ctx.default_params(F))
    with pytest.raises(EbvoError) as ei:
        ctx.temporal_set_keyframe()                     # not finalized
    assert ei.value.status == EBVO_ERR_STATE
    ctx.stereo_finalize(None)
    ctx.temporal_set_keyframe()
    counts, q = ctx.temporal_match()                    # a frame against itself
    assert counts["n_kf"] == counts["n_cf"] and counts["n_kept"] >= counts["n_kf"] * 0.9
    with pytest.raises(EbvoError) as ei:                # the chain after the NCC filter was not run for this pair
        ctx.lib.ebvo_temporal_fetch_final.restype = int
        ctx._check(ctx.lib.ebvo_temporal_fetch_final(ctx._ctx, 0, None, None, None, None, None, None, None, None, None), "fetch_final")
    assert ei.value.status == EBVO_ERR_STATE
    counts, q = ctx.temporal_match(stages=1)            # ... now it was: a frame against itself keeps itself (the
    assert counts["n_final"] >= counts["n_kf"] * 0.8    # refinement stops at once: "outlier", fewer than two iterations)
    rows = np.repeat(np.arange(counts["n_kf"]), np.diff(q["final"]["row_ptr"]))
    assert (q["final"]["cf_index"] == rows).mean() > 0.8
    ctx.stereo_upload(l, r)
    with pytest.raises(EbvoError):
        ctx.temporal_match()                            # the new pair has no final mates yet
    with pytest.raises(EbvoError):
        ctx._check(ctx.lib.ebvo_temporal_fetch_final(ctx._ctx, 0, None, None, None, None, None, None, None, None, None), "fetch_final")


def test_temporal_chain_with_nothing_to_match(ctx):
    """Empty ends of the chain: a current frame without a single mate in reach of the keyframe's, and a keyframe without
    mates -- counts of zero, an empty final list, no error."""
    h, w = 96, 160
    F = synth.fundamental_for("kitti")
    l, r = synth.stereo_pair("s2", h, w)
    ctx.stereo_upload(l, r)
    ctx.stereo_run(ctx.default_params(F))
    _, kf = ctx.stereo_finalize(None)
    assert len(kf["left_index"]) > 100
    ctx.temporal_set_keyframe()
    flat = np.full((h, w), 128, dtype=np.uint8)           # no edges at all in the current frame
    ctx.stereo_upload(flat, flat)
    c = ctx.stereo_run(ctx.default_params(F))
    assert c.n_left == 0
    ctx.stereo_finalize(None)
    counts, q = ctx.temporal_match(stages=1)
    assert counts["n_cf"] == 0 and counts["n_final"] == 0 and len(q["final"]["cf_index"]) == 0
    assert_bit_equal(q["final"]["row_ptr"], np.zeros(counts["n_kf"] + 1, dtype=np.int32))
    # the mirrored scene: candidates exist by location, none survives orientation + NCC + SIFT all the way ... or some do; either
    # way the lists are consistent
    ctx.stereo_upload(l[:, ::-1].copy(), r[:, ::-1].copy())
    ctx.stereo_run(ctx.default_params(F))
    ctx.stereo_finalize(None)
    counts, q = ctx.temporal_match(stages=1)
    f = q["final"]
    assert f["row_ptr"][-1] == counts["n_final"] == len(f["cf_index"]) and (np.diff(f["row_ptr"]) >= 0).all()
    assert counts["n_final"] <= counts["n_bnb_sift"] <= counts["n_bnb_ncc"] <= counts["n_sift"] <= counts["n_kept"]
    # a keyframe without mates
    ctx.stereo_upload(flat, flat)
    ctx.stereo_run(ctx.default_params(F))
    ctx.stereo_finalize(None)
    ctx.temporal_set_keyframe()
    ctx.stereo_upload(l, r)
    ctx.stereo_run(ctx.default_params(F))
    ctx.stereo_finalize(None)
    counts, q = ctx.temporal_match(stages=1)
    assert counts["n_kf"] == 0 and counts["n_final"] == 0 and len(q["final"]["row_ptr"]) == 1
