"""GPU parity of the candidate search + geometric filters (ebvo_epi_candidates).

Bar: identical CSR (row_ptr and ascending right indices) to the brute-force CPU oracle, for every
stage mask, on rectified and slanted epipolar geometry, and for edge lists in arbitrary order."""
import numpy as np
import pytest

from edge_based_visual_odometry_amd import synth
from edge_based_visual_odometry_amd._lib import EDGE_DTYPE
from tests import oracle as orc
from tests.util import assert_bit_equal

pytestmark = pytest.mark.gpu

F_KITTI = synth.fundamental_21(synth.CALIB["kitti"]["K"], synth.CALIB["kitti"]["K"], synth.CALIB["kitti"]["R21"],
                               synth.CALIB["kitti"]["T21"])
# a non-rectified geometry (slanted epipolar lines): small rotation + vertical baseline component
_c, _s = np.cos(0.03), np.sin(0.03)
F_SLANT = synth.fundamental_21((450.0, 450.0, 100.0, 60.0), (460.0, 455.0, 98.0, 63.0),
                               ((_c, -_s, 0.0), (_s, _c, 0.0), (0.0, 0.0, 1.0)), (0.11, 0.02, 0.003))


def _edges_pair(ctx, h, w, kind="s2"):
    l, r = synth.stereo_pair(kind, h, w)
    return ctx.toed(l).edges, ctx.toed(r).edges


@pytest.mark.parametrize("mask", [1, 2, 4, 3, 5, 6, 7])
@pytest.mark.parametrize("F", [F_KITTI, F_SLANT], ids=["rectified", "slanted"])
def test_candidates_match_bruteforce(ctx, mask, F):
    L, R = _edges_pair(ctx, 120, 200)
    assert len(L) > 500 and len(R) > 500
    if mask in (2, 4, 6):          # without the epipolar stage the lists are huge: subsample the left side
        L = L[::7]
    lines = orc.epipolar_lines(F, L)
    assert_bit_equal(ctx.epipolar_lines(F, L), lines, "lines")
    rp, ci = orc.epi_candidates(L, R, lines, stage_mask=mask)
    grp, gci = ctx.epi_candidates(L, R, lines, stage_mask=mask)
    assert_bit_equal(grp, rp, "row_ptr")
    assert_bit_equal(gci, ci, "col_idx")
    assert len(ci) > 0


def test_candidates_full_size_subset(ctx):
    """KITTI shape: all right edges, a strided subset of left edges (the oracle is O(nL*nR))."""
    l, r = synth.stereo_pair("s2", 376, 1241)
    L, R, _ = ctx.toed_pair(l, r)
    Ls = L[::97]
    lines = orc.epipolar_lines(F_KITTI, Ls)
    rp, ci = orc.epi_candidates(Ls, R, lines)
    grp, gci = ctx.epi_candidates(Ls, R, lines)
    assert_bit_equal(grp, rp, "row_ptr")
    assert_bit_equal(gci, ci, "col_idx")
    assert len(ci) > len(Ls)


def test_candidates_arbitrary_order_and_thresholds(ctx):
    """The box pre-filter must stay exact for edge lists that are not in raster order."""
    rng = np.random.default_rng(5)
    L, R = _edges_pair(ctx, 96, 160)
    R = R[rng.permutation(len(R))].copy()
    L = L[rng.permutation(len(L))][:400].copy()
    lines = orc.epipolar_lines(F_SLANT, L)
    for thr, disp, orient in ((0.5, 25.0, 10.0), (2.0, 8.0, 45.0), (0.05, 60.0, 1.0)):
        rp, ci = orc.epi_candidates(L, R, lines, thr, disp, orient)
        grp, gci = ctx.epi_candidates(L, R, lines, thr, disp, orient)
        assert_bit_equal(grp, rp, "row_ptr")
        assert_bit_equal(gci, ci, "col_idx")


def test_candidates_random_points(ctx):
    rng = np.random.default_rng(9)
    n = 3000
    def rand_edges(n):
        e = np.zeros(n, dtype=EDGE_DTYPE)
        e["x"], e["y"] = rng.uniform(0, 300, n), rng.uniform(0, 200, n)
        e["theta"] = rng.uniform(-np.pi, np.pi, n)
        e["index"] = np.arange(n)
        return e
    L, R = rand_edges(500), rand_edges(n)
    lines = rng.normal(size=(len(L), 3)) * np.array([1e-3, 1e-3, 0.2])
    rp, ci = orc.epi_candidates(L, R, lines, 1.0, 40.0, 20.0)
    grp, gci = ctx.epi_candidates(L, R, lines, 1.0, 40.0, 20.0)
    assert_bit_equal(grp, rp, "row_ptr")
    assert_bit_equal(gci, ci, "col_idx")


def test_candidates_empty_and_degenerate(ctx):
    L, R = _edges_pair(ctx, 64, 96)
    lines = orc.epipolar_lines(F_KITTI, L)
    rp, ci = ctx.epi_candidates(L[:0], R, lines[:0])
    assert list(rp) == [0] and len(ci) == 0
    rp, ci = ctx.epi_candidates(L, R[:0], lines)
    assert (rp == 0).all() and len(ci) == 0
    zero = np.zeros_like(lines)          # a = b = 0: distance is NaN/inf, nothing passes the epipolar test
    rp, ci = ctx.epi_candidates(L, R, zero)
    orp, oci = orc.epi_candidates(L, R, zero)
    assert_bit_equal(rp, orp)
    assert len(ci) == len(oci) == 0


def test_candidates_rows_longer_than_the_staging_area(ctx):
    """Rows with more than 64 candidates (the counting pass stages 64 per row; longer rows are redone by the fill pass),
    mixed with short rows in the same 64-edge tiles, plus NaN and duplicated right edges."""
    L, R = _edges_pair(ctx, 120, 200)
    R = R.copy()
    R["x"][5::97] = np.nan                                   # NaN fails every predicate
    R = np.concatenate([R, R[100:140]])                      # exact duplicates (same location, later index)
    R["index"] = np.arange(len(R))
    lines = orc.epipolar_lines(F_KITTI, L)
    for thr, disp, mask in ((3.0, 25.0, 1), (0.5, 60.0, 3), (0.5, 25.0, 7)):
        rp, ci = orc.epi_candidates(L, R, lines, thr, disp, 10.0, stage_mask=mask)
        grp, gci = ctx.epi_candidates(L, R, lines, thr, disp, 10.0, stage_mask=mask)
        assert_bit_equal(grp, rp, "row_ptr")
        assert_bit_equal(gci, ci, "col_idx")
        n = np.diff(rp)
        if mask == 1:
            assert n.min() > 64                              # every row is refilled
        if mask == 3:
            assert n.max() > 64 and n.min() <= 64            # staged and refilled rows side by side


@pytest.mark.parametrize("cfg", ["kitti", "euroc"])
def test_staged_candidates_give_all_three_stages_from_one_search(ctx, cfg):
    """ebvo_epi_candidates_staged: the (epipolar AND disparity) list of the oracle, and the pairs flagged orient_ok are, in
    order, the oracle's list for all three stages."""
    l, r = synth.stereo_pair("s2", 120, 200)
    L, R = orc.toed(l)["edges"], orc.toed(r)["edges"]
    lines = orc.epipolar_lines(synth.fundamental_for(cfg), L)
    for thr in ((0.5, 25.0, 10.0), (1.5, 12.0, 30.0)):
        rp, ci, ok = ctx.epi_candidates_staged(L, R, lines, *thr)
        rp2, ci2 = orc.epi_candidates(L, R, lines, *thr, stage_mask=3)
        assert_bit_equal(rp, rp2, "row_ptr (epipolar + disparity)")
        assert_bit_equal(ci, ci2, "col_idx (epipolar + disparity)")
        rp3, ci3 = orc.epi_candidates(L, R, lines, *thr, stage_mask=7)
        rows = np.repeat(np.arange(len(L)), np.diff(rp))
        keep = ok.astype(bool)
        assert_bit_equal(ci[keep], ci3, "flagged pairs == the list of all three stages")
        assert_bit_equal(np.bincount(rows[keep], minlength=len(L)).astype(np.int32), np.diff(rp3).astype(np.int32), "rows")
        if cfg == "kitti":   # the euroc F leaves this small synthetic pair without candidates: parity of the empty lists only
            assert 0 < keep.sum() < len(keep)
    # empty inputs
    rp, ci, ok = ctx.epi_candidates_staged(L[:0], R, lines[:0])
    assert len(rp) == 1 and len(ci) == 0 and len(ok) == 0
