"""Fixed-scale SIFT descriptors and the SIFT stages of the chain on the device vs the oracle's restatement of cv::SIFT
(parity unpinned against OpenCV itself; bit-exact between the two restatements: same float operations in the same order,
the histogram accumulated in sample order)."""
import numpy as np
import pytest

from edge_based_visual_odometry_amd import synth
from tests import oracle as orc
from tests import oracle_chain
from tests.util import assert_bit_equal, assert_edges_equal

pytestmark = pytest.mark.gpu

F_KITTI = synth.fundamental_for("kitti")


@pytest.mark.parametrize("shape", [(96, 160), (200, 320)])
def test_descriptors_and_distances_equal_oracle(ctx, shape):
    l, r = synth.stereo_pair("s2", *shape)
    L, R = orc.toed(l)["edges"], orc.toed(r)["edges"]
    dl, dr = ctx.sift_descriptors(l, L), ctx.sift_descriptors(r, R)
    assert_bit_equal(dl, orc.sift_descriptors(l, L), "left descriptors")
    assert_bit_equal(dr, orc.sift_descriptors(r, R), "right descriptors")
    rp, ci = orc.epi_candidates(L, R, orc.epipolar_lines(F_KITTI, L))
    d = ctx.sift_min_distances(dl, dr[ci], rp)
    assert_bit_equal(d, orc.sift_min_distances(dl, dr[ci], rp), "min distances")
    assert 0.3 < (d < 500).mean() < 1.0


def test_descriptors_near_the_border_and_odd_orientations(ctx):
    """Keypoints whose 11 x 11 window leaves the image (samples skipped, r > 0 && r < rows - 1 ...), orientations on the
    wrap (kp.angle == 0 -> 360 - angle == 360 -> 0) and negative orientations (kp.angle < 0)."""
    img = synth.s2_image(64, 80, noise_seed=5)
    e = np.zeros(12, dtype=orc.EDGE_DTYPE)
    e["x"] = [2.0, 77.5, 40.0, 40.0, 9.3, 70.2, 40.5, 40.5, 1.0, 79.0, 30.25, 55.75]
    e["y"] = [3.0, 60.5, 1.5, 62.0, 9.9, 54.1, 30.5, 30.5, 63.0, 0.0, 20.0, 41.0]
    e["theta"] = [0.0, -0.0, np.pi, -np.pi / 2, 1e-9, 3.1, -3.1, 0.7853981633974483, 2.0, -2.0, 1.5707963267948966, -1e-7]
    assert_bit_equal(ctx.sift_descriptors(img, e), orc.sift_descriptors(img, e), "descriptors")


@pytest.mark.parametrize("shape", [(120, 200), (200, 320)])
def test_chain_with_sift_equals_oracle_chain(ctx, shape):
    l, r = synth.stereo_pair("s2", *shape)
    cal = synth.CALIB["kitti"]
    K = [cal["K"][0], 0, cal["K"][2], 0, cal["K"][1], cal["K"][3], 0, 0, 1]
    calib = (K, K, cal["R21"], cal["T21"])
    ctx.stereo_upload(l, r)
    ctx.stereo_run(ctx.default_params(F_KITTI))
    counts, fin = ctx.stereo_finalize(calib, use_sift=True)
    ref = oracle_chain.stereo_edge_pairs(l, r, F_KITTI, calib, sift=True)
    assert counts == ref["counts"]
    assert counts["n_sift"] > counts["n_ncc"] > counts["n_bnb"] >= counts["n_clusters"] >= counts["n_final"] > 0
    assert_bit_equal(fin["left_index"], ref["left_index"], "left_index")
    assert_edges_equal(fin["right"], ref["right"], "right centre")
    assert_bit_equal(fin["score"], ref["score"], "score")
    assert_bit_equal(fin["rows"], ref["rows"], "rows")
    # the SIFT stages change the result: the chain without them keeps more candidates per row
    counts0, _ = ctx.stereo_finalize(calib)
    assert counts0["n_bnb"] > counts["n_bnb"]
