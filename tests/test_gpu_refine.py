"""Photometric Gauss-Newton refinement (ebvo_gn_refine_stereo, ebvo_sobel_gradients) through the C ABI vs the oracle.
Every output -- alpha, score, confidence, refined location, validity, iteration count -- is bit-exact (same scalar IEEE
operations in the same order; sin / cos / exp are the routines of csrc/ebvo_math.h on both sides)."""
import numpy as np
import pytest

from edge_based_visual_odometry_amd import synth
from tests import oracle as orc
from tests.util import assert_bit_equal

pytestmark = pytest.mark.gpu

F_KITTI = synth.fundamental_for("kitti")


def _compare(out, ref):
    assert_bit_equal(out["validity"], ref["validity"], "validity")
    assert_bit_equal(out["iters"], ref["iters"], "iters")
    assert_bit_equal(out["alpha"], ref["alpha"], "alpha")
    assert_bit_equal(out["score"], ref["score"], "score")
    assert_bit_equal(out["refined_xy"], ref["refined_xy"], "refined_xy")
    assert_bit_equal(out["confidence"], ref["confidence"], "confidence")


@pytest.fixture(params=["rows", "threads", "rows_per_iteration", "rows_three_waves"])
def gn_layout(request, ctx):
    """The launch layouts of the stereo Gauss-Newton iterations (developer keys of ebvo_debug_set): eight lanes per pair in one
    persistent launch (what the library picks below ~49 k active pairs; built for two or three waves per SIMD), the same
    layout as a launch per iteration, and one thread per pair (what larger problems run).  Same bits every way."""
    key, value = {"rows": (4, 0), "threads": (4, 1), "rows_per_iteration": (7, 1), "rows_three_waves": (8, 3)}[request.param]
    ctx.debug_set(key, value)
    yield request.param
    ctx.debug_set(4, 0)
    ctx.debug_set(7, 0)
    ctx.debug_set(8, 2)


@pytest.mark.parametrize("shape", [(48, 64), (96, 160), (120, 200)])
def test_sobel_equals_oracle(ctx, shape):
    img = synth.s2_image(*shape)
    gx, gy = ctx.sobel_gradients(img)
    ox, oy = orc.sobel_gradients(img)
    assert_bit_equal(gx, ox, "gx")
    assert_bit_equal(gy, oy, "gy")


def test_refine_pipeline_matches_small(ctx, gn_layout):
    """The reference's use: every kept NCC match of a pair, candidate = the right TOED edge."""
    l, r = synth.stereo_pair("s2", 96, 160)
    ctx.stereo_upload(l, r)
    c = ctx.stereo_run(ctx.default_params(F_KITTI))
    o = ctx.stereo_fetch(c)
    keep = o["keep"].astype(bool)
    rows = np.repeat(np.arange(c.n_left), np.diff(o["row_ptr"]))[keep]
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=c.n_left))]).astype(np.int32)
    cand = np.stack([o["right"]["x"][o["col_idx"][keep]], o["right"]["y"][o["col_idx"][keep]]], 1)
    assert rp[-1] == len(cand) == c.n_matches > 100
    lines = ctx.epipolar_lines(F_KITTI, o["left"])
    out = ctx.gn_refine_stereo(l, r, o["left"], lines, rp, cand)
    ref = orc.gn_refine_stereo(l, r, o["left"], lines, rp, cand)
    _compare(out, ref)
    assert (out["validity"] == 1).mean() > 0.5


@pytest.mark.parametrize("cfg", ["euroc", "eth3d"])
def test_refine_slanted_lines_borders_and_params(ctx, cfg, gn_layout):
    """Slanted epipolar lines (EuRoC calibration), candidates on and beyond the image border (clamped sampling),
    non-default parameters, empty rows."""
    F = synth.fundamental_for(cfg)
    l, r = synth.stereo_pair("s2", 120, 200)
    L = ctx.toed(l).edges
    rng = np.random.default_rng(3)
    L = L[rng.choice(len(L), 600, replace=False)]
    L = L[np.argsort(L["index"])]
    lines = ctx.epipolar_lines(F, L)
    per = rng.integers(0, 4, len(L))                               # 0..3 candidates per left edge
    rp = np.concatenate([[0], np.cumsum(per)]).astype(np.int32)
    rows = np.repeat(np.arange(len(L)), per)
    cand = np.stack([L["x"][rows] - 12.0 + rng.uniform(-2, 2, len(rows)), L["y"][rows] + rng.uniform(-1, 1, len(rows))], 1)
    cand[::17] = rng.uniform(-8, 8, (len(cand[::17]), 2))          # near / outside the top-left corner
    cand[5::23, 0] = 199.0 + rng.uniform(-3, 6, len(cand[5::23]))  # right border
    for kw in ({}, dict(max_iter=3, tol=1e-2, huber_delta=1.0), dict(max_iter=1)):
        out = ctx.gn_refine_stereo(l, r, L, lines, rp, cand, **kw)
        ref = orc.gn_refine_stereo(l, r, L, lines, rp, cand, **{**dict(max_iter=20, tol=1e-3, huber_delta=3.0), **kw})
        _compare(out, ref)


def test_refine_degenerate(ctx, gn_layout):
    flat = np.full((48, 64), 77, dtype=np.uint8)
    L = np.zeros(2, dtype=orc.EDGE_DTYPE)
    L["x"], L["y"], L["theta"] = [20.0, 30.0], [20.0, 25.0], [0.3, -1.2]
    lines = orc.epipolar_lines(F_KITTI, L)
    rp = np.array([0, 1, 3], dtype=np.int32)
    cand = np.array([[15.0, 20.0], [25.0, 25.0], [-5.0, 200.0]])
    out = ctx.gn_refine_stereo(flat, flat, L, lines, rp, cand)
    _compare(out, orc.gn_refine_stereo(flat, flat, L, lines, rp, cand))
    assert np.all(out["validity"] == 2) and np.all(np.isnan(out["score"]))
    empty = ctx.gn_refine_stereo(flat, flat, L, lines, np.zeros(3, dtype=np.int32), np.zeros((0, 2)))
    assert len(empty["alpha"]) == 0


def test_refine_full_size_property(ctx):
    """KITTI shape, every kept match of the pipeline (~4.7e5 pairs): the refined disparity of the valid fits is the
    generator's 12 px; rerun gives identical bits."""
    h, w = synth.SHAPES["kitti"]
    l, r = synth.stereo_pair("s2", h, w)
    ctx.stereo_upload(l, r)
    c = ctx.stereo_run(ctx.default_params(F_KITTI))
    o = ctx.stereo_fetch(c)
    keep = o["keep"].astype(bool)
    rows = np.repeat(np.arange(c.n_left), np.diff(o["row_ptr"]))[keep]
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=c.n_left))]).astype(np.int32)
    cand = np.stack([o["right"]["x"][o["col_idx"][keep]], o["right"]["y"][o["col_idx"][keep]]], 1)
    lines = ctx.epipolar_lines(F_KITTI, o["left"])
    out = ctx.gn_refine_stereo(l, r, o["left"], lines, rp, cand)
    again = ctx.gn_refine_stereo(l, r, o["left"], lines, rp, cand)
    for k in out:
        assert_bit_equal(out[k], again[k], k)
    ok = out["validity"] == 1
    assert ok.mean() > 0.5
    disp = o["left"]["x"][rows][ok] - out["refined_xy"][ok, 0]
    true = np.abs(o["left"]["x"][rows][ok] - cand[ok, 0] - 12.0) < 1.0          # candidates that were the true mate
    assert true.mean() > 0.5
    assert np.median(np.abs(disp[true] - 12.0)) < 0.1
    assert np.all(out["refined_xy"][:, 1] == cand[:, 1])                        # rectified: rows do not move


@pytest.mark.parametrize("shape", [(96, 160), (376, 1241)])
def test_device_resident_refine_equals_host_buffer_call(ctx, shape):
    """ebvo_stereo_refine on the resident pair == ebvo_gn_refine_stereo fed with the fetched kept matches."""
    from edge_based_visual_odometry_amd._lib import EbvoError, EBVO_ERR_STATE
    l, r = synth.stereo_pair("s2", *shape)
    ctx.stereo_upload(l, r)
    with pytest.raises(EbvoError) as ei:
        ctx.stereo_refine(type("C", (), {"n_pairs": 0})())          # nothing has run on the pair yet
    assert ei.value.status == EBVO_ERR_STATE
    c = ctx.stereo_run(ctx.default_params(F_KITTI))
    dev = ctx.stereo_refine(c)
    o = ctx.stereo_fetch(c)
    keep = o["keep"].astype(bool)
    assert np.all(dev["validity"][~keep] == 255) and np.all(dev["validity"][keep] != 255)
    rows = np.repeat(np.arange(c.n_left), np.diff(o["row_ptr"]))[keep]
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=c.n_left))]).astype(np.int32)
    cand = np.stack([o["right"]["x"][o["col_idx"][keep]], o["right"]["y"][o["col_idx"][keep]]], 1)
    host = ctx.gn_refine_stereo(l, r, o["left"], ctx.epipolar_lines(F_KITTI, o["left"]), rp, cand)
    for k in host:
        assert_bit_equal(dev[k][keep], host[k], k)
    xy_all = np.stack([o["right"]["x"][o["col_idx"]], o["right"]["y"][o["col_idx"]]], 1)
    assert_bit_equal(dev["refined_xy"][~keep], xy_all[~keep], "unrefined locations")


def _temporal_scene(h, w, sx=3, sy=2):
    kf_img = synth.s2_image(h + 16, w + 16, scene=7, noise_seed=1)
    cf_img = np.ascontiguousarray(kf_img[8 - sy:8 - sy + h, 8 - sx:8 - sx + w])
    kf_img = np.ascontiguousarray(kf_img[8:8 + h, 8:8 + w])
    return kf_img, cf_img


def _compare_temporal(out, ref):
    for k in ("validity", "iters", "disp", "score"):
        assert_bit_equal(out[k], ref[k], k)


@pytest.mark.parametrize("shape", [(96, 160), (120, 200)])
def test_temporal_refine_equals_oracle(ctx, shape, gn_layout):
    """ebvo_gn_refine_temporal vs the restatement of src/Temporal_Matches.cpp:735-851: bit-exact disparity, score,
    validity and iteration count, incl. items starting on / beyond the border and non-default parameters."""
    h, w = shape
    kf_img, cf_img = _temporal_scene(h, w)
    kf = ctx.toed(kf_img).edges
    cfe = ctx.toed(cf_img).edges
    rng = np.random.default_rng(9)
    n = 1500
    kf = kf[rng.choice(len(kf), n, replace=True)]
    cf = cfe[rng.choice(len(cfe), n, replace=True)].copy()
    good = rng.random(n) < 0.7                                            # 70 %: the true mate, the rest: unrelated edges
    cf["x"][good], cf["y"][good], cf["theta"][good] = kf["x"][good] + 3, kf["y"][good] + 2, kf["theta"][good]
    init = np.stack([kf["x"] - cf["x"], kf["y"] - cf["y"]], 1) + rng.uniform(-1.5, 1.5, (n, 2))
    init[::31] += rng.uniform(-40, 40, (len(init[::31]), 2))             # patches clamped at the border
    for kw in ({}, dict(max_iter=4, tol=5e-2, huber_delta=1.5), dict(max_iter=1)):
        out = ctx.gn_refine_temporal(kf_img, cf_img, kf, cf, init, **kw)
        ref = orc.gn_refine_temporal(kf_img, cf_img, kf, cf, init, **{**dict(max_iter=20, tol=1e-3, huber_delta=3.0), **kw})
        _compare_temporal(out, ref)
    assert (out["validity"] != 1).all()                                   # one iteration: always "outlier"
    empty = ctx.gn_refine_temporal(kf_img, cf_img, kf[:0], cf[:0], init[:0])
    assert len(empty["score"]) == 0


def test_temporal_refine_flat_image(ctx, gn_layout):
    """Zero gradients: H is only the accumulated 1e-6 regulariser, the update is exactly zero, one iteration."""
    flat = np.full((64, 96), 90, dtype=np.uint8)
    kf = np.zeros(3, dtype=orc.EDGE_DTYPE)
    kf["x"], kf["y"], kf["theta"] = [30.0, 40.0, 50.0], [30.0, 20.0, 40.0], [0.1, 1.0, -2.0]
    init = np.array([[1.0, -2.0], [0.0, 0.0], [5.5, 3.25]])
    out = ctx.gn_refine_temporal(flat, flat, kf, kf, init)
    _compare_temporal(out, orc.gn_refine_temporal(flat, flat, kf, kf, init))
    assert np.array_equal(out["disp"], init) and np.all(out["iters"] == 1) and np.all(out["score"] == 0.0)


def test_finalize_pairs_equals_oracle_and_geometry(ctx):
    """ebvo_finalize_pairs vs the restatement of src/Stereo_Matches.cpp:1656-1699 (bit-exact), plus the geometry it
    must satisfy on the synthetic rectified pair: depth f B / d, unit tangents, projected tangent along the edge."""
    l, r = synth.stereo_pair("s2", 120, 200)
    ctx.stereo_upload(l, r)
    c = ctx.stereo_run(ctx.default_params(F_KITTI))
    o = ctx.stereo_fetch(c)
    keep = o["keep"].astype(bool)
    rows = np.repeat(np.arange(c.n_left), np.diff(o["row_ptr"]))[keep]
    L, R = o["left"][rows], o["right"][o["col_idx"][keep]]
    cal = synth.CALIB["kitti"]
    fx, fy, cx, cy = cal["K"]
    K = [fx, 0, cx, 0, fy, cy, 0, 0, 1]
    out = ctx.finalize_pairs(K, K, cal["R21"], cal["T21"], L, R)
    ref = orc.finalize_pairs(K, K, cal["R21"], cal["T21"], L, R)
    assert_bit_equal(out, ref, "finalize_pairs")
    assert_bit_equal(out[:, 0], L["x"]) and assert_bit_equal(out[:, 5], R["theta"])
    d = L["x"] - R["x"]
    ok = np.abs(d - 12.0) < 0.5
    assert ok.mean() > 0.5
    # |Gamma.z| = f B / disparity; with the reference's T21 = (+0.54, 0, 0) (config/kitti.yaml) and x_R = x_L - d its
    # formula (src/utility.cpp:95-102) returns the depth with a minus sign -- reproduced, not corrected
    assert np.allclose(-out[ok, 8], fx * 0.54 / d[ok], rtol=1e-9)
    assert np.allclose(np.linalg.norm(out[:, 9:12], axis=1), 1.0, atol=1e-12)
    assert np.allclose(np.linalg.norm(out[:, 12:14], axis=1)[ok], 1.0, atol=1e-6)
    # other calibration (non-identity rotation) and an empty list
    ce = synth.CALIB["euroc"]
    Kl = [ce["K"][0], 0, ce["K"][2], 0, ce["K"][1], ce["K"][3], 0, 0, 1]
    Kr = [ce["K_right"][0], 0, ce["K_right"][2], 0, ce["K_right"][1], ce["K_right"][3], 0, 0, 1]
    assert_bit_equal(ctx.finalize_pairs(Kl, Kr, ce["R21"], ce["T21"], L[:500], R[:500]),
                     orc.finalize_pairs(Kl, Kr, ce["R21"], ce["T21"], L[:500], R[:500]), "euroc calib")
    assert ctx.finalize_pairs(K, K, cal["R21"], cal["T21"], L[:0], R[:0]).shape == (0, 16)
