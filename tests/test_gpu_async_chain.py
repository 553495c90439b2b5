"""The enqueue-only chains (ebvo_stereo_finalize_submit / _wait, ebvo_temporal_match_submit / _wait): several slots in flight,
no host synchronisation between the stages of a chain -- same bits as the one-call forms, which are themselves checked
against the oracle chain (tests/test_gpu_chain.py, test_gpu_fullsize.py, test_gpu_temporal.py)."""
import numpy as np
import pytest

from edge_based_visual_odometry_amd import _lib, synth
from edge_based_visual_odometry_amd._lib import EbvoError
from edge_based_visual_odometry_amd.api import Context
from tests.util import assert_bit_equal, assert_edges_equal

pytestmark = pytest.mark.gpu

H, W = 240, 376


def _setup():
    ce = synth.CALIB["euroc"]
    K = tuple(v / 2 for v in ce["K"])
    Kr = tuple(v / 2 for v in ce["K_right"])
    F = synth.fundamental_21(K, Kr, ce["R21"], ce["T21"])
    calib = ([K[0], 0, K[2], 0, K[1], K[3], 0, 0, 1], [Kr[0], 0, Kr[2], 0, Kr[1], Kr[3], 0, 0, 1], ce["R21"], ce["T21"])
    frames = []
    for k in range(4):
        l, r = synth.stereo_pair("s2", H, W, scene=7, noise_base=2 * k, disparity=9)
        frames.append((np.roll(l, k, axis=1), np.roll(r, k, axis=1)))
    return F, calib, frames


def _same_final(a, b):
    assert_bit_equal(a["left_index"], b["left_index"], "left_index")
    assert_edges_equal(a["right"], b["right"], "right")
    assert_bit_equal(a["score"], b["score"], "score")
    assert_bit_equal(a["rows"], b["rows"], "rows")


def _same_quads(a, b, final):
    for k in ("row_ptr", "col_idx", "sim_left", "sim_right", "keep"):
        assert_bit_equal(a[k], b[k], k)
    if final:
        for k in ("row_ptr", "cf_index", "ncc_left", "sift_left", "score_left", "score_right", "valid"):
            assert_bit_equal(a["final"][k], b["final"][k], "final." + k)
        assert_edges_equal(a["final"]["left"], b["final"]["left"], "final.left")
        assert_edges_equal(a["final"]["right"], b["final"]["right"], "final.right")


@pytest.mark.parametrize("use_sift", [False, True])
@pytest.mark.parametrize("stages", [0, 1])
def test_pipelined_frames_equal_one_call_forms(use_sift, stages):
    F, calib, frames = _setup()
    with Context(H, W, device=0) as ctx:
        ctx.set_slots(4)
        params = ctx.default_params(F)
        # reference: frame after frame, every call waited for
        ref = []
        for k, (l, r) in enumerate(frames):
            ctx.stereo_upload(l, r, slot=k)
            ctx.stereo_submit(params, slot=k)
            ctx.stereo_wait(slot=k)
            fc, fin = ctx.stereo_finalize(calib, slot=k, use_sift=use_sift)
            if k == 0:
                ctx.temporal_set_keyframe(slot=0)
                ref.append((fc, fin, None, None))
            else:
                tc, q = ctx.temporal_match(slot=k, stages=stages)
                ref.append((fc, fin, tc, q))
        assert ref[1][0]["n_final"] > 1000 and ref[1][2]["n_kept"] > 1000
        # the same frames (still resident) with every stage enqueued before any result is read
        for k in range(1, 4):
            ctx.stereo_submit(params, slot=k)
        for k in range(1, 4):
            ctx.stereo_wait(slot=k)
            ctx.stereo_finalize_submit(calib, slot=k, use_sift=use_sift)
        got_fin = {}
        for k in range(1, 4):
            got_fin[k] = ctx.stereo_finalize_wait(slot=k)
            ctx.temporal_match_submit(slot=k, stages=stages)
        for k in (3, 1, 2):                                   # any order
            tc, q = ctx.temporal_match_wait(slot=k)
            assert got_fin[k][0] == ref[k][0]
            _same_final(got_fin[k][1], ref[k][1])
            assert tc == ref[k][2]
            _same_quads(q, ref[k][3], bool(stages))


def test_quad_buffers_too_small_are_regrown():
    F, calib, frames = _setup()
    with Context(H, W, device=0) as ctx:
        ctx.set_slots(2)
        params = ctx.default_params(F)
        for k in range(2):
            ctx.stereo_upload(*frames[k], slot=k)
            ctx.stereo_submit(params, slot=k)
            ctx.stereo_wait(slot=k)
            ctx.stereo_finalize(calib, slot=k)
        ctx.temporal_set_keyframe(slot=0)
        tc, q = ctx.temporal_match(slot=1, stages=1)          # sizes the buffers for this frame
        assert tc["n_candidates"] > 10000
        ctx.debug_set(6, 64)                                  # ... now pretend they were sized for 64 quads
        tc2, q2 = ctx.temporal_match(slot=1, stages=1)
        assert tc2 == tc
        _same_quads(q2, q, True)
        tc3, q3 = ctx.temporal_match(slot=1, stages=0)        # and once more with the regrown buffers
        assert tc3["n_kept"] == tc["n_kept"]
        assert_bit_equal(q3["keep"], q["keep"], "keep")


def test_a_slot_with_a_chain_in_flight_accepts_nothing_else():
    F, calib, frames = _setup()
    with Context(H, W, device=0) as ctx:
        params = ctx.default_params(F)
        ctx.stereo_upload(*frames[0])
        ctx.stereo_run(params)

        def refused(call):
            with pytest.raises(EbvoError) as ei:
                call()
            assert ei.value.status == _lib.EBVO_ERR_STATE

        refused(lambda: ctx.stereo_finalize_wait())           # nothing submitted
        refused(lambda: ctx.temporal_match_wait())
        ctx.stereo_finalize_submit(calib)
        refused(lambda: ctx.stereo_finalize_submit(calib))
        refused(lambda: ctx.stereo_upload(*frames[1]))
        refused(lambda: ctx.stereo_submit(params))
        refused(lambda: ctx.temporal_set_keyframe())          # no final mates yet
        refused(lambda: ctx.toed(frames[0][0]))               # host-buffer calls work on slot 0
        fc, fin = ctx.stereo_finalize_wait()
        assert fc["n_final"] == len(fin["left_index"]) > 1000
        refused(lambda: ctx.stereo_finalize_wait())
        refused(lambda: ctx.temporal_match_submit())          # no keyframe
        ctx.temporal_set_keyframe()
        ctx.temporal_match_submit(stages=1)
        refused(lambda: ctx.temporal_match_submit())
        refused(lambda: ctx.stereo_finalize_submit(calib))
        refused(lambda: ctx.stereo_upload(*frames[1]))
        tc, q = ctx.temporal_match_wait()
        assert tc["n_kf"] == tc["n_cf"] == fc["n_final"] and tc["n_final"] > 0.8 * tc["n_kf"]
        # empty ends: a pair without edges runs through both chains
        flat = np.full((H, W), 77, dtype=np.uint8)
        ctx.stereo_upload(flat, flat)
        ctx.stereo_run(params)
        ctx.stereo_finalize_submit(calib, use_sift=True)
        fc, fin = ctx.stereo_finalize_wait()
        assert fc["n_final"] == 0 and len(fin["left_index"]) == 0
        ctx.temporal_match_submit(stages=1)
        tc, q = ctx.temporal_match_wait()
        assert tc["n_cf"] == 0 and tc["n_final"] == 0 and len(q["final"]["row_ptr"]) == tc["n_kf"] + 1


def test_pipelined_frames_with_undistortion_equal_one_call_forms():
    """the raw / undistorted split of the resident pipeline through the enqueue-only chains"""
    F, calib, frames = _setup()
    ce = synth.CALIB["euroc"]
    K = tuple(v / 2 for v in ce["K"])
    Kr = tuple(v / 2 for v in ce["K_right"])
    with Context(H, W, device=0) as ctx:
        ctx.set_slots(3)
        ctx.set_undistort(K, ce["dist"], Kr, ce["dist_right"])
        params = ctx.default_params(F)
        ref = []
        for k in range(3):
            ctx.stereo_upload(*frames[k], slot=k)
            ctx.stereo_submit(params, slot=k)
            ctx.stereo_wait(slot=k)
            fc, fin = ctx.stereo_finalize(calib, slot=k, use_sift=True)
            if k == 0:
                ctx.temporal_set_keyframe(slot=0)
                ref.append((fc, fin, None, None))
            else:
                ref.append((fc, fin) + ctx.temporal_match(slot=k, stages=1))
        for k in (1, 2):
            ctx.stereo_submit(params, slot=k)
        for k in (1, 2):
            ctx.stereo_wait(slot=k)
            ctx.stereo_finalize_submit(calib, slot=k, use_sift=True)
        fins = {k: ctx.stereo_finalize_wait(slot=k) for k in (2, 1)}
        for k in (1, 2):
            ctx.temporal_match_submit(slot=k, stages=1)
        for k in (1, 2):
            tc, q = ctx.temporal_match_wait(slot=k)
            assert fins[k][0] == ref[k][0] and tc == ref[k][2]
            _same_final(fins[k][1], ref[k][1])
            _same_quads(q, ref[k][3], True)
        assert ref[1][2]["n_final"] > 300
