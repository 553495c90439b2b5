"""Full-size temporal parity (BASELINE configs[2] / configs[3]): keyframe = frame 0 of the SURVEY 8(d) config-3 sequence
(scene 7, noise seeds (2k + 1, 2k + 2), k px of global motion), a later frame matched against it at 752x480 WITH
undistortion (EuRoC) and at 942x489 (ETH3D delivery_area).  Every candidate quad of the grid + orientation search
(src/Temporal_Matches.cpp:335-414), both NCC maxima and the keep flag of every quad (:416-469), and every final quad of
the chain after the NCC filter (:184-215: SIFT filter, both Best-Nearly-Best tests, refinement of both cameras, clustering)
against the oracle-side chain (tests/oracle_chain.py) -- bit for bit, no strided subsets."""
import numpy as np
import pytest

from edge_based_visual_odometry_amd import synth
from tests import oracle as orc
from tests import oracle_chain

pytestmark = pytest.mark.gpu

CASES = {
    # name: (config, frame index of the current frame, undistortion on)
    "euroc-752x480-undistort": ("euroc", 3, True),
    "eth3d-942x489": ("eth3d", 2, False),
}


def sequence_frame(cfg, k, rank=0):
    """frame k of the sequence bench.py --workload euroc replays (sequence_bench)"""
    h, w = synth.SHAPES[cfg]
    l, r = synth.stereo_pair("s2", h, w, scene=7 + rank, noise_base=100 * rank + 2 * k, disparity=9)
    return np.roll(l, k, axis=1), np.roll(r, k, axis=1)


def calib_of(cfg):
    c = synth.CALIB[cfg]
    kl = [c["K"][0], 0, c["K"][2], 0, c["K"][1], c["K"][3], 0, 0, 1]
    kr = [c["K_right"][0], 0, c["K_right"][2], 0, c["K_right"][1], c["K_right"][3], 0, 0, 1]
    return kl, kr, c["R21"], c["T21"]


def mates_of(ctx, imgs, params, calib, slot=0):
    """the stereo mates of a frame as the host sees them: left TOED edge of every final pair + its right centre"""
    ctx.stereo_upload(*imgs, slot=slot)
    ctx.stereo_submit(params, slot=slot)
    c = ctx.stereo_wait(slot=slot)
    left = ctx.stereo_fetch(c, slot=slot)["left"]
    _, fin = ctx.stereo_finalize(calib, slot=slot, use_sift=True)
    return left[fin["left_index"]], fin["right"]


def image_triple(imgs, cal, undist):
    """(raw left, undistorted left, undistorted right) on the oracle"""
    if not undist:
        return imgs[0], imgs[0], imgs[1]
    return imgs[0], orc.undistort(imgs[0], cal["K"], cal["dist"]), orc.undistort(imgs[1], cal["K_right"], cal["dist_right"])


@pytest.mark.parametrize("case", list(CASES))
def test_temporal_frame_equals_oracle_at_full_size(ctx, case):
    if ctx.toed_mode != "hybrid":
        pytest.skip("the temporal stages do not depend on the detector mode; the full-size oracle run is made once")
    cfg, k, undist = CASES[case]
    h, w = synth.SHAPES[cfg]
    cal = synth.CALIB[cfg]
    F = synth.fundamental_for(cfg)
    calib = calib_of(cfg)
    params = ctx.default_params(F)
    if undist:
        ctx.set_undistort(cal["K"], cal["dist"], cal["K_right"], cal["dist_right"])
    try:
        f0, fk = sequence_frame(cfg, 0), sequence_frame(cfg, k)
        kfL, kfR = mates_of(ctx, f0, params, calib)
        ctx.temporal_set_keyframe()
        cfL, cfR = mates_of(ctx, fk, params, calib)
        counts, q = ctx.temporal_match(stages=1)
    finally:
        if undist:
            ctx.set_undistort()
    ref = oracle_chain.temporal_reference(kfL, kfR, cfL, cfR, image_triple(f0, cal, undist), image_triple(fk, cal, undist), w, h)
    assert oracle_chain.temporal_problems(counts, q, ref) == []
    assert counts["n_kf"] > 10000 and counts["n_candidates"] > 10 * counts["n_kf"] and counts["n_final"] > 1000
    # the scene moved by k px between the keyframe and this frame: the valid final quads are that motion
    rows = oracle_chain.rows_of(q["final"]["row_ptr"])
    v = q["final"]["valid"].astype(bool)
    assert v.mean() > 0.5 and abs(np.median(q["final"]["left"]["x"][v] - kfL["x"][rows[v]]) - k) < 0.5
