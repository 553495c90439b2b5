"""GPU parity of patch sampling and NCC scoring (ebvo_edge_patches, ebvo_ncc_pairs,
ebvo_ncc_patches, ebvo_ncc_quads).

Bar: patches and all four similarities bit-identical to the CPU oracle (which fixes one canonical
reduction order, so 'within 1e-5' is met with zero difference), keep flags identical, NaN and
-1.0 sentinels reproduced."""
import numpy as np
import pytest

from edge_based_visual_odometry_amd import synth
from edge_based_visual_odometry_amd._lib import EDGE_DTYPE
from tests import oracle as orc
from tests.util import assert_bit_equal

pytestmark = pytest.mark.gpu

F_KITTI = synth.fundamental_21(synth.CALIB["kitti"]["K"], synth.CALIB["kitti"]["K"], synth.CALIB["kitti"]["R21"],
                               synth.CALIB["kitti"]["T21"])


def _mk_edges(x, y, th):
    e = np.zeros(len(x), dtype=EDGE_DTYPE)
    e["x"], e["y"], e["theta"] = x, y, th
    e["index"] = np.arange(len(x))
    return e


def test_edge_patches_bit_exact(ctx):
    img = synth.s2_image(120, 200, 7, 1, 0)
    edges = ctx.toed(img).edges
    assert_bit_equal(ctx.edge_patches(img, edges), orc.edge_patches(img, edges), "patches")


def test_edge_patches_nan_rules(ctx):
    """Out-of-image corners and exactly-integer coordinates give NaN (include/utility.h:95-103)."""
    img = synth.s2_image(64, 96, 7, 1, 0)
    e = _mk_edges(np.array([2.3, 93.9, 40.0, 40.5, 50.25, 10.5]), np.array([30.2, 30.7, 1.2, 62.9, 20.0, 8.0]),
                  np.array([0.3, -2.0, 1.1, 3.0, 0.0, np.pi / 2]))
    got, ref = ctx.edge_patches(img, e), orc.edge_patches(img, e)
    assert_bit_equal(got, ref, "patches")
    assert np.isnan(got).any() and not np.isnan(got).all()
    # theta = 0 at integer y: the whole grid has integer y offsets -> every sample is NaN
    assert np.isnan(got[4]).all()


def _pairs_for(ctx, l, r, F, sub=None):
    L, R, _ = ctx.toed_pair(l, r)
    if sub:
        L = L[::sub]
    lines = orc.epipolar_lines(F, L)
    rp, ci = ctx.epi_candidates(L, R, lines)
    return L, R, rp, ci


@pytest.mark.parametrize("shape,sub", [((120, 200), None), ((376, 1241), 23)])
def test_ncc_pairs_bit_exact(ctx, shape, sub):
    l, r = synth.stereo_pair("s2", *shape)
    L, R, rp, ci = _pairs_for(ctx, l, r, F_KITTI, sub)
    Rc = R[ci]
    assert len(ci) > 100
    sims, best, keep, lp = ctx.ncc_pairs(l, r, L, Rc, rp, want_left_patches=True)
    osims, obest, okeep, olp = orc.ncc_pairs(l, r, L, Rc, rp)
    assert_bit_equal(lp, olp, "left_edge_patches")
    assert_bit_equal(sims, osims, "sims")
    assert_bit_equal(best, obest, "best")
    assert_bit_equal(keep, okeep, "keep")
    assert keep.sum() > 0 and keep.sum() < len(keep)
    # against exact arithmetic: the north-star tolerance
    a, b = olp[0, 0].astype(np.float64), orc.edge_patches(r, Rc[rp[0]:rp[0] + 1])[0, 0].astype(np.float64)
    if rp[1] > rp[0] and not (np.isnan(a).any() or np.isnan(b).any()):
        da, db = a - a.mean(), b - b.mean()
        assert abs(sims[0, 0] - (da @ db) / np.sqrt((da @ da) * (db @ db))) <= 1e-5


def test_ncc_pairs_sentinels_and_nan(ctx):
    """Flat patches -> -1.0; NaN in pp poisons the max, NaN in a later term is ignored
    (std::max({...}) semantics, src/Stereo_Matches.cpp:596)."""
    rng = np.random.default_rng(2)
    imgL = synth.s2_image(80, 120, 7, 1, 0)
    imgR = imgL.copy()
    imgR[:, 60:] = 90                      # flat right half: zero-variance patches
    x = np.concatenate([rng.uniform(12, 108, 40), [3.0, 117.2, 60.3]])
    y = np.concatenate([rng.uniform(12, 68, 40), [40.2, 40.1, 4.9]])
    th = rng.uniform(-np.pi, np.pi, len(x))
    L = _mk_edges(x, y, th)
    Rc = _mk_edges(np.concatenate([x[:20] + 0.3, x[20:40] * 0 + 90.7, x[40:]]), y + 0.1, th + 0.05)
    rp = np.arange(len(L) + 1, dtype=np.int32)
    sims, best, keep, lp = ctx.ncc_pairs(imgL, imgR, L, Rc, rp, want_left_patches=True)
    osims, obest, okeep, olp = orc.ncc_pairs(imgL, imgR, L, Rc, rp)
    assert_bit_equal(sims, osims, "sims")
    assert_bit_equal(best, obest, "best")
    assert_bit_equal(keep, okeep, "keep")
    assert (sims == -1.0).any() and np.isnan(sims).any() and np.isnan(best).any()


def test_ncc_pairs_ragged_rows(ctx):
    """Rows with zero candidates and rows with many; cluster centres (non-TOED edges) as candidates."""
    rng = np.random.default_rng(4)
    l, r = synth.stereo_pair("s2", 100, 160)
    L = ctx.toed(l).edges[:300]
    counts = rng.integers(0, 6, len(L))
    counts[::5] = 0
    rp = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    n = int(rp[-1])
    li = np.repeat(np.arange(len(L)), counts)
    Rc = _mk_edges(L["x"][li] - 12 + rng.normal(0, 0.7, n), L["y"][li] + rng.normal(0, 0.3, n),
                   L["theta"][li] + rng.normal(0, 0.05, n))
    sims, best, keep, _ = ctx.ncc_pairs(l, r, L, Rc, rp)
    osims, obest, okeep, _ = orc.ncc_pairs(l, r, L, Rc, rp)
    assert_bit_equal(sims, osims, "sims")
    assert_bit_equal(keep, okeep, "keep")


def test_ncc_stored_patches_and_quads(ctx):
    rng = np.random.default_rng(6)
    n = 500
    A = (rng.integers(0, 256, (n, 49)) + rng.random((n, 49))).astype(np.float32)
    B = (0.6 * A + rng.normal(0, 25, (n, 49))).astype(np.float32)
    A[3] = 12.0                      # zero variance -> -1
    B[7, 11] = np.nan                # NaN flows through
    assert_bit_equal(ctx.ncc_patches(A, B), orc.ncc_patches(A, B), "ncc_patches")
    assert ctx.ncc_patches(A, B)[3] == -1.0 and np.isnan(ctx.ncc_patches(A, B)[7])
    q = [(rng.integers(0, 256, (n, 2, 49)) + rng.random((n, 2, 49))).astype(np.float32) for _ in range(2)]
    kfL, kfR = q
    cfL = (kfL[:, ::-1] * 0.8 + rng.normal(0, 10, kfL.shape)).astype(np.float32)
    cfR = (kfR * 0.9 + rng.normal(0, 1, kfR.shape) * rng.uniform(5, 150, (n, 1, 1))).astype(np.float32)
    sl, sr, keep = ctx.ncc_quads(kfL, kfR, cfL, cfR, 0.8)
    osl, osr, okeep = orc.ncc_quads(kfL, kfR, cfL, cfR, 0.8)
    assert_bit_equal(sl, osl, "sim_left")
    assert_bit_equal(sr, osr, "sim_right")
    assert_bit_equal(keep, okeep, "keep")
    assert 0 < keep.sum() < n


def test_ncc_empty(ctx):
    l, r = synth.stereo_pair("s2", 64, 96)
    L = ctx.toed(l).edges[:10]
    rp = np.zeros(len(L) + 1, dtype=np.int32)
    sims, best, keep, lp = ctx.ncc_pairs(l, r, L, L[:0], rp, want_left_patches=True)
    assert len(sims) == 0 and lp.shape == (10, 2, 49)
    assert_bit_equal(lp, orc.edge_patches(l, L))
    assert len(ctx.ncc_patches(np.zeros((0, 49), np.float32), np.zeros((0, 49), np.float32))) == 0
