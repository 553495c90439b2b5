"""GPU parity of the third-order edge detector, through the C ABI (ebvo_toed / ebvo_toed_pair).

Bar: bit-exact (x, y, theta, index), counts and the sub-pixel magnitude list against the CPU
oracle in portable-math mode; the reference's own known-answer "xyi" hashes at full size."""
import numpy as np
import pytest

from edge_based_visual_odometry_amd import synth
from tests import oracle as orc
from tests.util import assert_bit_equal, assert_edges_equal, case_id, kat_cases, kat_image

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", kat_cases(), ids=case_id)
def test_toed_matches_reference_known_answers(ctx, case):
    """x, y, index of every edge hash to the value recorded from the reference source."""
    img = kat_image(case)
    assert synth.img_fnv(img) == case["img_fnv"]
    r = ctx.toed(img)
    assert len(r.edges) == case["kept"]
    assert r.n_total == case["total"]
    assert orc.edge_hash(r.edges, False) == case["xyi"]
    if "e0" in case:
        assert (r.edges[0]["x"], r.edges[0]["y"]) == tuple(case["e0"][:2])
        assert abs(r.edges[0]["theta"] - case["e0"][2]) <= 1e-15 * abs(case["e0"][2]) * 4


@pytest.mark.parametrize("shape,kind", [((48, 64), "s1"), ((48, 64), "s2"), ((100, 131), "s2"), ((33, 37), "s2"),
                                        ((376, 1241), "s2"), ((480, 752), "s1")])
def test_toed_bit_exact_vs_oracle(ctx, shape, kind):
    h, w = shape
    img = synth.s1_image(h, w, 5, 3) if kind == "s1" else synth.s2_image(h, w, 11, 4, 2)
    ref = orc.toed(img, math_mode=orc.PORTABLE, want_all=True)
    got = ctx.toed(img, want_all=True)
    assert got.n_total == ref["n_total"]
    assert_edges_equal(got.edges, ref["edges"])
    assert_bit_equal(got.all4, ref["all4"], "subpix_edge_pts_final")
    assert got.time_conv > 0 and got.time_nms > 0


def test_toed_pair_equals_two_single_calls(ctx):
    l, r = synth.stereo_pair("s2", 120, 200)
    el, er, nt = ctx.toed_pair(l, r)
    a, b = ctx.toed(l), ctx.toed(r)
    assert_edges_equal(el, a.edges, "left")
    assert_edges_equal(er, b.edges, "right")
    assert nt == (a.n_total, b.n_total)


def test_toed_flat_and_saturated_images_have_no_edges(ctx):
    for v in (0, 255, 77):
        r = ctx.toed(np.full((64, 96), v, dtype=np.uint8))
        assert len(r.edges) == 0 and r.n_total == 0


def test_toed_strided_input(ctx):
    big = synth.s2_image(80, 160, 3, 9, 0)
    view = big[:, 16:144]            # stride 160, width 128
    assert not view.flags["C_CONTIGUOUS"]
    a = ctx.toed(view)
    b = ctx.toed(np.ascontiguousarray(view))
    assert_edges_equal(a.edges, b.edges)


def test_toed_capacity_error_reports_required_size(ctx):
    from edge_based_visual_odometry_amd._lib import EbvoError, EBVO_ERR_CAPACITY
    img = synth.s2_image(64, 96, 7, 1, 0)
    full = ctx.toed(img)
    assert len(full.edges) > 4
    with pytest.raises(EbvoError) as ei:
        ctx.toed(img, cap=3)
    assert ei.value.status == EBVO_ERR_CAPACITY
    assert f"kept={len(full.edges)}" in str(ei.value)


def test_toed_rejects_bad_sizes(ctx):
    from edge_based_visual_odometry_amd._lib import EbvoError, EBVO_ERR_ARG
    with pytest.raises(EbvoError) as ei:
        ctx.toed(np.zeros((ctx.max_h + 1, 64), dtype=np.uint8))
    assert ei.value.status == EBVO_ERR_ARG
    with pytest.raises(EbvoError):
        ctx.toed(np.zeros((16, 16), dtype=np.uint8))


def test_toed_idempotent(ctx):
    img = synth.s2_image(376, 1241, 7, 1, 0)
    a, b = ctx.toed(img), ctx.toed(img)
    assert_edges_equal(a.edges, b.edges)


def _stress_images():
    rng = np.random.default_rng(42)
    h, w = 96, 128
    yy, xx = np.mgrid[0:h, 0:w]
    imgs = {
        "uniform_noise": rng.integers(0, 256, (h, w)).astype(np.uint8),
        "ramp_x": np.clip(xx * 2, 0, 255).astype(np.uint8),                       # constant gradient: |g| ties everywhere
        "ramp_diag": np.clip(xx + yy, 0, 255).astype(np.uint8),                    # gx == gy: ambiguous sector
        "checker4": (((xx // 4 + yy // 4) % 2) * 200 + 20).astype(np.uint8),
        "checker1": (((xx + yy) % 2) * 255).astype(np.uint8),
        "stripes_v": ((xx // 6 % 2) * 180 + 30).astype(np.uint8),                 # gy == 0 exactly
        "stripes_h": ((yy // 5 % 2) * 180 + 30).astype(np.uint8),                 # gx == 0 exactly
        "disc": (((xx - 64) ** 2 + (yy - 48) ** 2 < 30 ** 2) * 150 + 50).astype(np.uint8),
        "step_plus_noise": (np.where(xx < 64, 60, 190) + rng.integers(-3, 4, (h, w))).astype(np.uint8),
        "s1_other_seed": synth.s1_image(h, w, 17, 5),
        "s2_other_scene": synth.s2_image(h, w, 23, 9, 3),
        "border_blob": np.pad(np.full((h - 8, w - 8), 220, np.uint8), 4, constant_values=10),
    }
    return imgs


@pytest.mark.parametrize("name", sorted(_stress_images()))
def test_hybrid_equals_strict_on_adversarial_images(name):
    """The separable screen of the hybrid detector must never lose a pixel the exact test accepts: exact ties,
    axis-aligned and diagonal gradients, 1-px textures, structures touching the border.  Strict mode is checked
    against the CPU oracle, hybrid mode against strict -- both bit for bit."""
    from edge_based_visual_odometry_amd.api import Context
    img = _stress_images()[name]
    ref = orc.toed(img, math_mode=orc.PORTABLE, want_all=True)
    with Context(128, 160, toed_mode="strict") as cs, Context(128, 160, toed_mode="hybrid") as ch:
        a, b = cs.toed(img, want_all=True), ch.toed(img, want_all=True)
    assert a.n_total == ref["n_total"] == b.n_total
    assert_edges_equal(a.edges, ref["edges"], "strict")
    assert_edges_equal(b.edges, a.edges, "hybrid")
    assert_bit_equal(b.all4, a.all4, "subpix_edge_pts_final")


@pytest.mark.parametrize("name", ["checker4", "checker1", "uniform_noise"])
def test_hybrid_screen_overflow_falls_back_to_strict(name):
    """The candidate buffers of the hybrid detector hold max_h * max_w entries; on an image of ties the screen can flag
    more grid points than that.  The library must notice and repeat the image on the strict path -- in a context sized
    exactly to the image -- instead of dropping candidates."""
    from edge_based_visual_odometry_amd.api import Context
    img = _stress_images()[name]
    h, w = img.shape
    ref = orc.toed(img, math_mode=orc.PORTABLE, want_all=True)
    with Context(h, w, toed_mode="hybrid") as ch:
        b = ch.toed(img, want_all=True)
        st = ch.toed_stats()["left"]
        fell_back = ch.toed_fallbacks
        lb, rb, _ = ch.toed_pair(img, img[:, ::-1].copy())
    assert b.n_total == ref["n_total"]
    assert_edges_equal(b.edges, ref["edges"], "hybrid, tight context")
    assert_bit_equal(b.all4, ref["all4"], "subpix_edge_pts_final")
    assert_edges_equal(lb, ref["edges"], "pair call, left")
    if name == "checker4":
        assert fell_back >= 1, (fell_back, st)  # 27 k screened candidates for 12 k entries


def test_pipeline_falls_back_to_strict_when_the_screen_overflows():
    """Same on the enqueue-only pair pipeline: the result record carries the overflow and ebvo_stereo_wait repeats the
    pair with the strict detector."""
    from edge_based_visual_odometry_amd.api import Context
    img = _stress_images()["checker4"]
    right = np.roll(img, -3, axis=1)
    h, w = img.shape
    out = {}
    for mode in ("strict", "hybrid"):
        with Context(h, w, toed_mode=mode) as c:
            p = c.default_params()
            c.stereo_upload(img, right)
            c.stereo_submit(p)
            counts = c.stereo_wait()
            res = c.stereo_fetch(counts)
            fb_first = c.toed_fallbacks
            # the SAME images again (a replayed pair): the slot remembers that the screen overflowed and goes strict at once --
            # no second hybrid pass, no second fallback -- until new images are uploaded
            for _ in range(3):
                c.stereo_submit(p)
                again = c.stereo_wait()
                assert (again.n_left, again.n_pairs, again.n_matches) == (counts.n_left, counts.n_pairs, counts.n_matches)
            assert c.toed_fallbacks == fb_first
            out[mode] = (counts, res, fb_first)
            # new images: the override is gone, the hybrid detector runs (and does not overflow on an ordinary image)
            from edge_based_visual_odometry_amd import synth as _synth
            l2, r2 = _synth.stereo_pair("s2", h, w)
            c.stereo_upload(l2, r2)
            c.stereo_submit(p)
            c.stereo_wait()
            assert c.toed_fallbacks == fb_first
    cs, rs, _ = out["strict"]
    ch, rh, fb = out["hybrid"]
    assert fb == 1
    assert (cs.n_left, cs.n_right, cs.n_pairs, cs.n_matches) == (ch.n_left, ch.n_right, ch.n_pairs, ch.n_matches)
    assert cs.n_left > 0
    for k in ("left", "right", "row_ptr", "col_idx", "sims", "best", "keep"):
        assert_bit_equal(rh[k], rs[k], k)


def _random_images(seed, n):
    """Images built to put many gradient magnitudes within the screen's tolerance of each other: smooth ramps and blobs
    perturbed by one or two grey levels, low-contrast textures, blurred noise, plateaus."""
    rng = np.random.default_rng(seed)
    h, w = 64, 96
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = []
    for k in range(n):
        kind = k % 6
        if kind == 0:      # a ramp in a random direction plus +-1 grey level of noise
            a = rng.uniform(0, 2 * np.pi)
            img = 100 + (np.cos(a) * xx + np.sin(a) * yy) * rng.uniform(0.3, 2.5) + rng.integers(-1, 2, (h, w))
        elif kind == 1:    # two-level texture, contrast 2 .. 12 grey levels
            img = 120 + rng.integers(0, 2, (h, w)) * rng.integers(2, 13)
        elif kind == 2:    # blurred noise (box filter), strong
            z = rng.integers(0, 256, (h + 8, w + 8)).astype(np.float64)
            c = np.cumsum(np.cumsum(z, 0), 1)
            s = int(rng.integers(2, 7))
            img = (c[s:, s:] - c[:-s, s:] - c[s:, :-s] + c[:-s, :-s])[:h, :w] / (s * s)
        elif kind == 3:    # concentric rings: every gradient direction, exact symmetries
            r = np.hypot(xx - w / 2 + rng.uniform(-1, 1), yy - h / 2 + rng.uniform(-1, 1))
            img = 128 + 100 * np.cos(r / rng.uniform(1.5, 5.0))
        elif kind == 4:    # plateaus with steps of 1 .. 4 grey levels (|g| just around the threshold of 2)
            img = 90 + (xx // rng.integers(3, 9) + yy // rng.integers(3, 9)) * rng.integers(1, 5)
        else:              # saturated blobs
            img = np.clip(rng.normal(128, 90, (h, w)), 0, 255)
        out.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
    return out


def test_hybrid_equals_strict_on_random_near_tie_images():
    """The FP32 screen against the strict detector on 240 images that crowd the relaxed test's tolerances; strict mode is the
    one checked against the oracle (every image here too, at a sixth of the count)."""
    from edge_based_visual_odometry_amd.api import Context
    imgs = _random_images(20260101, 240)
    h, w = imgs[0].shape
    total = 0
    with Context(h, w, toed_mode="strict") as cs, Context(h, w, toed_mode="hybrid") as ch:
        for k, img in enumerate(imgs):
            a, b = cs.toed(img, want_all=True), ch.toed(img, want_all=True)
            assert a.n_total == b.n_total, (k, a.n_total, b.n_total)
            assert_edges_equal(b.edges, a.edges, f"image {k}")
            assert_bit_equal(b.all4, a.all4, f"image {k}: subpix_edge_pts_final")
            total += a.n_total
            if k % 6 == k // 40:  # one of each kind against the oracle as well
                ref = orc.toed(img, math_mode=orc.PORTABLE, want_all=True)
                assert a.n_total == ref["n_total"]
                assert_edges_equal(a.edges, ref["edges"], f"image {k} vs oracle")
        assert total > 100000, total  # the images do produce edges
        assert ch.toed_fallbacks < len(imgs) // 2  # ... and most of them through the screen, not through the fallback
