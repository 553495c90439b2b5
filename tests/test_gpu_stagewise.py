"""The resident stage-wise calls (ebvo_toed_resident / ebvo_epi_candidates_resident / ebvo_ncc_pairs_resident): main_VO's
stage-after-stage sequence (src/Pipeline.cpp:24-29, :93-97; src/Stereo_Matches.cpp:1374-1427) with the edge lists staying
on the device between the stages.  Same bits as the host-buffer entry points and as the oracle; a stale tag is refused."""
import numpy as np
import pytest

from edge_based_visual_odometry_amd import _lib, synth
from edge_based_visual_odometry_amd._lib import EbvoError
from tests import oracle as orc
from tests.util import assert_bit_equal, assert_edges_equal

pytestmark = pytest.mark.gpu


def _pair(h, w, **kw):
    return synth.stereo_pair("s2", h, w, **kw)


@pytest.mark.parametrize("shape", [(96, 160), (240, 376)])
@pytest.mark.parametrize("left_ws", [0, 1])
def test_stagewise_resident_equals_host_buffer_calls_and_oracle(ctx, shape, left_ws):
    h, w = shape
    l, r = _pair(h, w)
    F = synth.fundamental_for("kitti")
    eL, ntL, a4L, tagL = ctx.toed_resident(l, left_ws, want_all=True)
    eR, ntR, a4R, tagR = ctx.toed_resident(r, 1 - left_ws, want_all=True)
    assert tagL and tagR and tagL != tagR
    oL, oR = orc.toed(l, want_all=True), orc.toed(r, want_all=True)
    assert_edges_equal(eL, oL["edges"], "left")
    assert_edges_equal(eR, oR["edges"], "right")
    assert ntL == oL["n_total"] and ntR == oR["n_total"]
    assert_bit_equal(a4L, oL["all4"], "subpix_edge_pts_final left")
    assert_bit_equal(a4R, oR["all4"], "subpix_edge_pts_final right")
    lines = ctx.epipolar_lines(F, eL)
    # the three stages fused, and staged (list under epipolar + disparity, orientation as flags)
    rp, ci = ctx.epi_candidates_resident(tagL, tagR, lines)
    orp, oci = orc.epi_candidates(oL["edges"], oR["edges"], lines)
    assert_bit_equal(rp, orp, "row_ptr")
    assert_bit_equal(ci, oci, "col_idx")
    rp2, ci2, ok = ctx.epi_candidates_resident(tagL, tagR, lines, staged=True)
    orp2, oci2 = orc.epi_candidates(oL["edges"], oR["edges"], lines, stage_mask=3)
    assert_bit_equal(rp2, orp2, "staged row_ptr")
    assert_bit_equal(ci2, oci2, "staged col_idx")
    assert_bit_equal(ci2[ok.astype(bool)], oci, "flagged pairs = the fused list")
    rpf, cif = ctx.last_final_lists                                    # ... which the staged call also returns, formed on the device
    assert_bit_equal(rpf, orp, "row_ptr_final")
    assert_bit_equal(cif, oci, "col_idx_final")
    # NCC by index into the resident right edges, with the left patches
    sims, best, keep, lp = ctx.ncc_pairs_resident(tagL, tagR, l, r, rp, ci, want_left_patches=True)
    osims, obest, okeep, _ = orc.ncc_pairs(l, r, oL["edges"], oR["edges"][oci], orp)
    assert_bit_equal(sims, osims, "sims")
    assert_bit_equal(best, obest, "best")
    assert_bit_equal(keep, okeep, "keep")
    assert_bit_equal(lp, orc.edge_patches(l, oL["edges"]), "left patches")
    # ... and without the optional arrays
    sims2, best2, keep2, lp2 = ctx.ncc_pairs_resident(tagL, tagR, l, r, rp, ci, want_sims=False)
    assert sims2 is None and lp2 is None
    assert_bit_equal(best2, obest, "best")
    assert_bit_equal(keep2, okeep, "keep")
    # a subsequence of the lists (what the SIFT filter leaves, src/Stereo_Matches.cpp:1414): rows shrink, order kept
    sel = np.ones(len(ci), dtype=bool)
    sel[::3] = False
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    rp3 = np.concatenate([[0], np.cumsum(np.bincount(rows[sel], minlength=len(rp) - 1))]).astype(np.int32)
    sims3, best3, keep3, _ = ctx.ncc_pairs_resident(tagL, tagR, l, r, rp3, ci[sel])
    assert_bit_equal(sims3, osims[sel], "sims of the sub-list")
    assert_bit_equal(keep3, okeep[sel], "keep of the sub-list")


def test_ncc_samples_the_images_it_is_given(ctx):
    """TOED ran on the undistorted pair, the NCC samples the RAW pair (src/Stereo_Matches.cpp:562-563)"""
    h, w = 120, 200
    l, r = _pair(h, w)
    raw_l, raw_r = _pair(h, w, noise_base=6)              # stand-ins for the raw images: same scene, other pixels
    F = synth.fundamental_for("kitti")
    eL, _, _, tagL = ctx.toed_resident(l, 0)
    eR, _, _, tagR = ctx.toed_resident(r, 1)
    rp, ci = ctx.epi_candidates_resident(tagL, tagR, ctx.epipolar_lines(F, eL))
    sims, best, keep, lp = ctx.ncc_pairs_resident(tagL, tagR, raw_l, raw_r, rp, ci, want_left_patches=True)
    osims, obest, okeep, _ = orc.ncc_pairs(raw_l, raw_r, eL, eR[ci], rp)
    assert_bit_equal(sims, osims, "sims")
    assert_bit_equal(keep, okeep, "keep")
    assert_bit_equal(lp, orc.edge_patches(raw_l, eL), "left patches")
    assert not np.array_equal(osims, orc.ncc_pairs(l, r, eL, eR[ci], rp)[0])


def test_stale_tags_are_refused(ctx):
    h, w = 96, 160
    l, r = _pair(h, w)
    F = synth.fundamental_for("kitti")
    eL, _, _, tagL = ctx.toed_resident(l, 0)
    eR, _, _, tagR = ctx.toed_resident(r, 1)
    lines = ctx.epipolar_lines(F, eL)
    rp, ci = ctx.epi_candidates_resident(tagL, tagR, lines)

    def refused(call):
        with pytest.raises(EbvoError) as ei:
            call()
        assert ei.value.status == _lib.EBVO_ERR_STATE

    refused(lambda: ctx.epi_candidates_resident(tagL, tagL, lines))          # one list cannot be both sides
    refused(lambda: ctx.epi_candidates_resident(tagL, tagR + 100, lines))    # never issued
    refused(lambda: ctx.epi_candidates_resident(0, tagR, lines))
    _, _, _, tagL2 = ctx.toed_resident(l, 0)                                 # workspace 0 reused: the old tag is gone
    refused(lambda: ctx.epi_candidates_resident(tagL, tagR, lines))
    rp_b, ci_b = ctx.epi_candidates_resident(tagL2, tagR, lines)
    assert_bit_equal(rp_b, rp, "row_ptr")
    assert_bit_equal(ci_b, ci, "col_idx")
    ctx.epi_candidates(eL, eR, lines)                                        # host-buffer calls that run no detector leave
    ctx.sift_descriptors(l, eL[:50])                                         # the resident edge lists alone
    rp_c, ci_c = ctx.epi_candidates_resident(tagL2, tagR, lines)
    assert_bit_equal(ci_c, ci, "col_idx")
    ctx.toed(l)                                                              # ... the detector overwrites them
    refused(lambda: ctx.epi_candidates_resident(tagL2, tagR, lines))
    refused(lambda: ctx.ncc_pairs_resident(tagL2, tagR, l, r, rp, ci))
    _, _, _, tagL3 = ctx.toed_resident(l, 0)
    _, _, _, tagR3 = ctx.toed_resident(r, 1)
    ctx.stereo_upload(l, r)                                                  # ... and so does a pair uploaded into slot 0
    refused(lambda: ctx.epi_candidates_resident(tagL3, tagR3, lines))
    l2, _ = _pair(64, 96)
    _, _, _, tagL4 = ctx.toed_resident(l, 0)
    _, _, _, tagR4 = ctx.toed_resident(l2, 1)                                # another size: the other workspace is dropped
    refused(lambda: ctx.epi_candidates_resident(tagL4, tagR4, lines))


def test_malformed_lists_are_refused_before_any_launch(ctx):
    h, w = 96, 160
    l, r = _pair(h, w)
    F = synth.fundamental_for("kitti")
    eL, _, _, tagL = ctx.toed_resident(l, 0)
    eR, _, _, tagR = ctx.toed_resident(r, 1)
    rp, ci = ctx.epi_candidates_resident(tagL, tagR, ctx.epipolar_lines(F, eL))
    for bad_ci in (np.where(np.arange(len(ci)) == 5, len(eR), ci), np.where(np.arange(len(ci)) == 7, -1, ci)):
        with pytest.raises(EbvoError) as ei:
            ctx.ncc_pairs_resident(tagL, tagR, l, r, rp, bad_ci.astype(np.int32))
        assert ei.value.status == _lib.EBVO_ERR_ARG
    bad_rp = rp.copy()
    bad_rp[3], bad_rp[4] = rp[4] + 1, rp[3]
    with pytest.raises(EbvoError) as ei:
        ctx.ncc_pairs_resident(tagL, tagR, l, r, bad_rp, ci)
    assert ei.value.status == _lib.EBVO_ERR_ARG
    sims, _, _, _ = ctx.ncc_pairs_resident(tagL, tagR, l, r, rp, ci)         # the tags survived the refused calls
    assert len(sims) == len(ci)


def test_images_without_edges(ctx):
    h, w = 96, 160
    flat = np.full((h, w), 90, dtype=np.uint8)
    l, _ = _pair(h, w)
    eL, _, _, tagL = ctx.toed_resident(l, 0)
    eR, nt, _, tagR = ctx.toed_resident(flat, 1)
    assert len(eR) == 0 and nt == 0
    lines = ctx.epipolar_lines(synth.fundamental_for("kitti"), eL)
    rp, ci = ctx.epi_candidates_resident(tagL, tagR, lines)
    assert len(ci) == 0 and not rp.any() and len(rp) == len(eL) + 1
    rp_s, ci_s, ok_s = ctx.epi_candidates_resident(tagL, tagR, lines, staged=True)
    assert len(ci_s) == 0 and len(ok_s) == 0 and not ctx.last_final_lists[0].any() and len(ctx.last_final_lists[1]) == 0
    sims, best, keep, lp = ctx.ncc_pairs_resident(tagL, tagR, l, flat, rp, ci, want_left_patches=True)
    assert len(best) == 0 and lp.shape == (len(eL), 2, 49)
    rp, ci = ctx.epi_candidates_resident(tagR, tagL, np.zeros((0, 3)))       # no left edges
    assert len(rp) == 1 and len(ci) == 0


@pytest.mark.parametrize("cfg", ["euroc", "eth3d"])
def test_stagewise_resident_with_strided_images_and_slanted_lines(ctx, cfg):
    """non-rectified calibration (slanted epipolar lines) and images that are views into wider buffers (cv::Mat ROI: step > cols)"""
    h, w = 200, 312
    l, r = _pair(h, w, disparity=9)
    wide_l = np.zeros((h, w + 24), dtype=np.uint8)
    wide_r = np.full((h, w + 40), 255, dtype=np.uint8)
    wide_l[:, 8:8 + w] = l
    wide_r[:, 16:16 + w] = r
    vl, vr = wide_l[:, 8:8 + w], wide_r[:, 16:16 + w]          # strides w + 24 / w + 40
    assert not vl.flags["C_CONTIGUOUS"]
    F = synth.fundamental_for(cfg)
    eL, _, _, tagL = ctx.toed_resident(vl, 0)
    eR, _, _, tagR = ctx.toed_resident(vr, 1)
    oL, oR = orc.toed(l)["edges"], orc.toed(r)["edges"]
    assert_edges_equal(eL, oL, "left")
    assert_edges_equal(eR, oR, "right")
    lines = ctx.epipolar_lines(F, eL)
    rp, ci = ctx.epi_candidates_resident(tagL, tagR, lines)
    orp, oci = orc.epi_candidates(oL, oR, lines)
    assert_bit_equal(rp, orp, "row_ptr")
    assert_bit_equal(ci, oci, "col_idx")
    assert len(ci) > len(eL)
    sims, best, keep, lp = ctx.ncc_pairs_resident(tagL, tagR, vl, vr, rp, ci, want_left_patches=True)
    osims, obest, okeep, _ = orc.ncc_pairs(l, r, oL, oR[oci], orp)
    assert_bit_equal(sims, osims, "sims")
    assert_bit_equal(keep, okeep, "keep")
    assert_bit_equal(lp, orc.edge_patches(l, oL), "left patches")
