"""The FP32 screen of the hybrid detector, MEASURED (ebvo_toed_screen_audit): at every candidate the screen's gx, gy, |g|
against the exact stage's values, and |g| at every neighbour grid point -- on the three full-size workloads and on full-size
saturating images (0 / 255 stripes along both axes and the diagonal, and checkerboards, periods 1..19 px: transitions
aligned with the sign changes of the 17- / 19-tap kernels).  Asserted: observed error <= the worst-case budget
(tools/screen_error_bound.py, compiled into toed_kernels.hip) AND hybrid == strict, bit for bit, on the same image
(the decision itself: src/toed/cpu_toed.cpp:406-483)."""
import numpy as np
import pytest

from edge_based_visual_odometry_amd import _lib, synth
from edge_based_visual_odometry_amd.api import Context
from tests.util import assert_bit_equal, assert_edges_equal

pytestmark = pytest.mark.gpu

SHAPES = {"kitti": (376, 1241), "euroc": (480, 752), "eth3d": (489, 942)}


@pytest.fixture(scope="module")
def c():
    ctx = Context(max_h=512, max_w=1280, device=0, toed_mode="hybrid")
    yield ctx
    ctx.close()


def both_modes(c, img):
    c.set_toed_mode("hybrid")
    f0 = c.toed_fallbacks
    cap = 4 * img.size                              # an image of ties has more maxima than pixels
    hy = c.toed(img, want_all=True, cap=cap)
    fell_back = c.toed_fallbacks != f0
    c.set_toed_mode("strict")
    st = c.toed(img, want_all=True, cap=cap)
    c.set_toed_mode("hybrid")
    assert_edges_equal(hy.edges, st.edges)
    assert hy.n_total == st.n_total
    assert_bit_equal(hy.all4, st.all4, "subpix_edge_pts_final")
    return hy, fell_back


@pytest.fixture(scope="module")
def big():
    """a context whose candidate buffers hold EVERY grid point of a KITTI-size image (cap = max_h * max_w >= 4 H W): the screen
    cannot overflow, so that images of ties -- where the sector rule flags every point with |g| > 2 -- are audited, not skipped"""
    ctx = Context(max_h=800, max_w=2500, device=0, toed_mode="hybrid")
    yield ctx
    ctx.close()


def check_audit(a, what):
    assert a["bound_g"] == pytest.approx(7.67e-5) and a["bound_mag"] == pytest.approx(1.22e-4), what
    assert a["tol_mag"] >= 2 * a["bound_mag"] and a["tol_slope"] >= 2 * a["bound_slope"], what
    assert a["max_err_gx"] <= a["bound_g"], (what, a)
    assert a["max_err_gy"] <= a["bound_g"], (what, a)
    assert a["max_err_mag"] <= a["bound_mag"], (what, a)
    assert a["max_err_mag_neighbours"] <= a["bound_mag"], (what, a)
    assert a["n_maxima"] <= a["n_candidates"], (what, a)


@pytest.mark.parametrize("name", list(SHAPES))
def test_screen_error_on_the_full_size_workloads(c, name):
    h, w = SHAPES[name]
    l, r = synth.stereo_pair("s2", h, w)
    for img in (l, r):
        a = c.toed_screen_audit(img)
        check_audit(a, name)
        hy, fell_back = both_modes(c, img)
        assert not fell_back
        assert (a["n_maxima"], a["n_kept"]) == (hy.n_total, len(hy.edges))
        # the audit is not vacuous, and the screen is far tighter on images than its worst case
        assert a["n_candidates"] > 50000 and 0 < a["max_err_mag"] < 0.25 * a["bound_mag"]
        # what the tolerances cost: candidates the exact NMS rejects
        assert a["n_candidates"] - a["n_maxima"] < 0.02 * a["n_maxima"], a


def saturating(kind, period, h, w):
    y, x = np.mgrid[0:h, 0:w]
    if kind == "vertical":
        b = (x // period) & 1
    elif kind == "horizontal":
        b = (y // period) & 1
    elif kind == "diagonal":
        b = ((x + y) // period) & 1
    else:
        b = ((x // period) + (y // period)) & 1
    return (b * 255).astype(np.uint8)


@pytest.mark.parametrize("kind", ["vertical", "horizontal", "diagonal", "checkerboard"])
def test_screen_error_on_full_size_saturating_images(big, c, kind):
    h, w = SHAPES["kitti"]
    worst = dict(gx=0.0, gy=0.0, mag=0.0, nb=0.0)
    most = 0
    overflowed = 0
    for period in range(1, 20):
        img = saturating(kind, period, h, w)
        a = big.toed_screen_audit(img)                # never overflows here: every period is audited
        check_audit(a, (kind, period))
        most = max(most, a["n_candidates"])
        worst = dict(gx=max(worst["gx"], a["max_err_gx"]), gy=max(worst["gy"], a["max_err_gy"]),
                     mag=max(worst["mag"], a["max_err_mag"]), nb=max(worst["nb"], a["max_err_mag_neighbours"]))
        hy, fell_back = both_modes(big, img)          # the decision itself: hybrid == strict, the exact stage over all candidates
        assert not fell_back and (a["n_maxima"], a["n_kept"]) == (hy.n_total, len(hy.edges))
        # ... and in the context sized for ordinary images, where an image of ties overflows the candidate buffers: the library
        # falls back to the strict path and returns the same edges
        try:
            hy2, fb2 = both_modes(c, img)
        except _lib.EbvoError as e:                   # more maxima than the small context holds at all (max_h * max_w): both
            assert e.status == _lib.EBVO_ERR_CAPACITY, e   # modes refuse alike; the roomy context above has covered the image
            overflowed += 1
            continue
        assert_edges_equal(hy2.edges, hy.edges)
        overflowed += bool(fb2)
    assert most > 100000, (kind, most)                # the images do exercise the exact stage
    print(f"{kind}: worst |screen - exact| gx {worst['gx']:.3e} gy {worst['gy']:.3e} |g| {worst['mag']:.3e} "
          f"neighbours {worst['nb']:.3e}; most candidates {most}; {overflowed} of 19 periods overflow a 512 x 1280 context")
