"""cv::undistort on the device (ebvo_undistort, and inside the resident pipeline via ebvo_stereo_set_undistort) against the
oracle's restatement: byte-exact.  With undistortion on, TOED and the refinement run on the undistorted images while
both NCC passes sample the RAW ones, as the reference does (src/Pipeline.cpp:78-99 vs src/Stereo_Matches.cpp:562-563)."""
import numpy as np
import pytest

from edge_based_visual_odometry_amd import synth
from tests import oracle as orc
from tests import oracle_chain
from tests.util import assert_bit_equal, assert_edges_equal

pytestmark = pytest.mark.gpu

CE = synth.CALIB["euroc"]


@pytest.mark.parametrize("shape", [(480, 752), (120, 200), (97, 131)])
@pytest.mark.parametrize("cam", ["left", "right"])
def test_undistort_equals_oracle(ctx, shape, cam):
    img = synth.s2_image(*shape, noise_seed=3)
    K, d = (CE["K"], CE["dist"]) if cam == "left" else (CE["K_right"], CE["dist_right"])
    if shape != (480, 752):                         # keep the principal point inside the smaller test images
        K = (K[0] * shape[1] / 752, K[1] * shape[0] / 480, K[2] * shape[1] / 752, K[3] * shape[0] / 480)
    got = ctx.undistort(img, K, d)
    ref = orc.undistort(img, K, d)
    assert (got == ref).all(), f"{int((got != ref).sum())} pixels differ"
    assert (got != img).mean() > 0.3
    assert (ctx.undistort(img, K, [0, 0, 0, 0]) == img).all()      # zero distortion: identity
    got5 = ctx.undistort(img, K, list(d) + [0.01])                  # k3
    assert (got5 == orc.undistort(img, K, list(d) + [0.01])).all()
    wide = np.zeros((shape[0], shape[1] + 13), dtype=np.uint8)      # strided input
    wide[:, :shape[1]] = img
    assert (ctx.undistort(wide[:, :shape[1]], K, d) == ref).all()


def test_resident_pipeline_with_undistortion(ctx):
    h, w = 240, 376
    K = (CE["K"][0] / 2, CE["K"][1] / 2, CE["K"][2] / 2, CE["K"][3] / 2)
    Kr = (CE["K_right"][0] / 2, CE["K_right"][1] / 2, CE["K_right"][2] / 2, CE["K_right"][3] / 2)
    F = synth.fundamental_21(K, Kr, CE["R21"], CE["T21"])
    l, r = synth.stereo_pair("s2", h, w, disparity=9)
    lu, ru = orc.undistort(l, K, CE["dist"]), orc.undistort(r, Kr, CE["dist_right"])
    kl = [K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1]
    kr = [Kr[0], 0, Kr[2], 0, Kr[1], Kr[3], 0, 0, 1]
    calib = (kl, kr, CE["R21"], CE["T21"])
    ctx.set_undistort(K, CE["dist"], Kr, CE["dist_right"])
    try:
        ctx.stereo_upload(l, r)
        c = ctx.stereo_run(ctx.default_params(F))
        out = ctx.stereo_fetch(c, patches=True)
        counts, fin = ctx.stereo_finalize(calib)
    finally:
        ctx.set_undistort()
    # stage by stage on the oracle: TOED on the undistorted images, candidates, NCC on the RAW images
    L, R = orc.toed(lu)["edges"], orc.toed(ru)["edges"]
    assert_edges_equal(out["left"], L, "left edges (undistorted image)")
    assert_edges_equal(out["right"], R, "right edges (undistorted image)")
    lines = orc.epipolar_lines(F, L)
    rp, ci = orc.epi_candidates(L, R, lines)
    assert_bit_equal(out["row_ptr"], rp) and assert_bit_equal(out["col_idx"], ci)
    sims, best, keep, lp = orc.ncc_pairs(l, r, L, R[ci], rp)
    assert_bit_equal(out["sims"], sims, "sims (raw images)")
    assert_bit_equal(out["keep"], keep, "keep")
    assert_bit_equal(out["left_patches"], lp, "left patches (raw image)")
    sims_u, _, _, _ = orc.ncc_pairs(lu, ru, L, R[ci], rp)
    assert not np.array_equal(sims_u, sims)                          # the distinction is observable in this fixture
    ref = oracle_chain.stereo_edge_pairs(l, r, F, calib, left_img_undist=lu, right_img_undist=ru)
    assert counts == ref["counts"] and counts["n_final"] > 100
    assert_bit_equal(fin["left_index"], ref["left_index"], "left_index")
    assert_edges_equal(fin["right"], ref["right"], "right centre")
    assert_bit_equal(fin["score"], ref["score"], "score")
    assert_bit_equal(fin["rows"], ref["rows"], "rows")
    # switched off again: the same raw pair gives the plain result
    ctx.stereo_upload(l, r)
    c2 = ctx.stereo_run(ctx.default_params(F))
    assert_edges_equal(ctx.stereo_fetch(c2)["left"], orc.toed(l)["edges"])
