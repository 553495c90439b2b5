"""Sequence-realistic ingest (round 4): ebvo_stereo_upload_async from a page-locked frame ring, issued a pair ahead of its
submission, and the compact result fetch.  Both must hand back exactly what the synchronous upload and the full fetch do
(the frame loop of cmd/main_VO.cpp:99-113 / src/Pipeline.cpp:77-99 around the same per-pair path)."""
import numpy as np
import pytest

from edge_based_visual_odometry_amd import _lib, synth
from edge_based_visual_odometry_amd.api import Context
from tests.util import assert_bit_equal, assert_edges_equal

pytestmark = pytest.mark.gpu

H, W = 200, 320
F = synth.fundamental_for("kitti")


@pytest.fixture(scope="module", params=["strict", "hybrid"])
def c(request):
    ctx = Context(H, W, device=0, toed_mode=request.param)
    ctx.set_slots(4)
    yield ctx
    ctx.close()


def _ring(n):
    return [tuple(np.ascontiguousarray(im) for im in synth.stereo_pair("s2", H, W, scene=3 + k, noise_base=20 * k, disparity=10))
            for k in range(n)]


def _reference(c, ring):
    """every pair of the ring through the synchronous upload and the full fetch on slot 0"""
    p = c.default_params(F)
    ref = []
    for l, r in ring:
        c.stereo_upload(l, r, slot=0)
        c.stereo_submit(p, slot=0)
        cnt = c.stereo_wait(slot=0)
        ref.append((cnt, c.stereo_fetch(cnt, slot=0)))
    return ref


def _same(out, ref):
    cnt, full = ref
    assert_edges_equal(out["left"], full["left"])
    assert_edges_equal(out["right"], full["right"])
    assert_bit_equal(out["row_ptr"], full["row_ptr"], "row_ptr")
    assert_bit_equal(out["col_idx"], full["col_idx"], "col_idx")
    assert_bit_equal(out["sims"], full["sims"], "sims")
    assert_bit_equal(out["keep"], full["keep"], "keep")


@pytest.mark.parametrize("registered", [True, False])
def test_async_upload_a_pair_ahead_equals_synchronous_upload(c, registered):
    ring = _ring(5)
    ref = _reference(c, ring)
    if registered:
        for pair in ring:
            for im in pair:
                c.host_register(im)
    before = c.ingest_stats()
    try:
        p = c.default_params(F)
        S, inflight, steps = 4, 3, 14
        # three pairs in flight on four slots; the images of step i + 1 go up while step i runs
        for k in range(inflight):
            c.stereo_upload_async(*ring[k % 5], slot=k)
            c.stereo_submit(p, slot=k)
        uploaded, ahead = inflight, inflight % S
        c.stereo_upload_async(*ring[uploaded % 5], slot=ahead)
        uploaded += 1
        for done in range(steps):
            k = done % S
            cnt = c.stereo_wait(slot=k)
            if ahead is not None:
                c.stereo_submit(p, slot=ahead)
                ahead = None
            out = c.stereo_fetch(cnt, slot=k)
            _same(out, ref[done % 5])
            assert (cnt.n_pairs, cnt.n_matches) == (ref[done % 5][0].n_pairs, ref[done % 5][0].n_matches)
            if uploaded < steps:
                c.stereo_upload_async(*ring[uploaded % 5], slot=k)
                uploaded += 1
                ahead = k
        after = c.ingest_stats()
        form = "pull_uploads" if registered else "stream_uploads"       # registered memory takes the pull form, pageable the stream
        other = "stream_uploads" if registered else "pull_uploads"
        assert after[form] - before[form] == uploaded and after[other] == before[other]
    finally:
        if registered:
            for pair in ring:
                for im in pair:
                    c.host_unregister(im)


def test_async_upload_then_host_buffer_call_on_slot_0_is_ordered(c):
    """a host-buffer entry point (slot 0's workspace) right behind an asynchronous upload into slot 0: the library drains the
    upload before it overwrites the image buffers"""
    ring = _ring(2)
    c.stereo_upload_async(*ring[0], slot=0)
    e = c.toed(ring[1][0]).edges                      # uploads ring[1][0] into slot 0's first image buffer
    c.stereo_upload(*ring[1], slot=0)
    cnt = c.stereo_run(c.default_params(F))
    out = c.stereo_fetch(cnt)
    assert_edges_equal(out["left"], e)


def test_pageable_images_may_be_overwritten_as_soon_as_the_async_upload_returns(c):
    """Images outside every registered range are read BEFORE ebvo_stereo_upload_async returns (into the slot's page-locked
    staging): the caller may overwrite or free them at once.  (Round 4: the runtime read a pageable source when the copy got
    its turn on the device -- a GPU memory fault at a freed host address, once in five full test runs.)"""
    ring = _ring(3)
    ref = _reference(c, ring)
    p = c.default_params(F)
    before = c.ingest_stats()
    for k, (l, r) in enumerate(ring):
        slot = 1 + (k & 1)
        lt, rt = l.copy(), r.copy()                    # pageable, never registered
        c.stereo_upload_async(lt, rt, slot=slot)
        lt[:] = 0                                      # overwritten ...
        rt[:] = 255
        del lt, rt                                     # ... and dropped before the pair is even submitted
        c.stereo_submit(p, slot=slot)
        cnt = c.stereo_wait(slot=slot)
        assert (cnt.n_left, cnt.n_right, cnt.n_pairs, cnt.n_matches) == \
            (ref[k][0].n_left, ref[k][0].n_right, ref[k][0].n_pairs, ref[k][0].n_matches)
        _same(c.stereo_fetch(cnt, slot=slot), ref[k])
    after = c.ingest_stats()
    assert after["stream_uploads"] - before["stream_uploads"] == 3 and after["pull_uploads"] == before["pull_uploads"]


def test_compact_fetch_equals_full_fetch(c):
    ring = _ring(2)
    ref = _reference(c, ring)
    p = c.default_params(F)
    for what in (_lib.COMPACT_DEFAULT, _lib.COMPACT_ALL, _lib.COMPACT_KEEP_BITS, _lib.COMPACT_XY | _lib.COMPACT_THETA):
        for k, (l, r) in enumerate(ring):
            c.stereo_upload(l, r, slot=1)
            c.stereo_submit(p, slot=1)
            cnt = c.stereo_wait(slot=1)
            c.stereo_fetch_compact_begin(slot=1, what=what)
            with pytest.raises(_lib.EbvoError):        # the arena holds the compact selection: the full view is refused
                c.stereo_fetch_end(slot=1)
            v = c.stereo_fetch_compact_end(slot=1)
            full = ref[k][1]
            assert v["n_pairs"] == cnt.n_pairs and v["n_matches"] == cnt.n_matches == int(full["keep"].sum())
            if what & _lib.COMPACT_XY:
                for side in ("left", "right"):
                    assert_bit_equal(v[side + "_xy"][:, 0], full[side]["x"], side + ".x")
                    assert_bit_equal(v[side + "_xy"][:, 1], full[side]["y"], side + ".y")
            else:
                assert v["left_xy"] is None
            if what & _lib.COMPACT_THETA:
                assert_bit_equal(v["left_theta"], full["left"]["theta"], "left.theta")
                assert_bit_equal(v["right_theta"], full["right"]["theta"], "right.theta")
            else:
                assert v["left_theta"] is None
            if what & _lib.COMPACT_CSR:
                assert_bit_equal(v["row_ptr"], full["row_ptr"], "row_ptr")
                assert_bit_equal(v["col_idx"], full["col_idx"], "col_idx")
            if what & _lib.COMPACT_BEST:
                assert_bit_equal(v["best"], full["best"], "best")
            if what & _lib.COMPACT_KEEP_BITS:
                bits = np.unpackbits(v["keep_bits"].view(np.uint8), bitorder="little")[: cnt.n_pairs]
                assert_bit_equal(bits, full["keep"], "keep bits")
                assert int(v["keep_bits"].view(np.uint8)[(cnt.n_pairs + 7) // 8:].sum()) == 0     # padding bits are clear
    # the full fetch still works on the same slot afterwards
    c.stereo_fetch_begin(slot=1)
    out = c.stereo_fetch_end(slot=1)
    assert_bit_equal(out["keep"], ref[1][1]["keep"], "keep")


def test_no_sims_flag_stores_best_and_keep_only(c):
    """EBVO_PAIR_NO_SIMS: the four similarities are not written (the reference keeps their maximum only,
    src/Stereo_Matches.cpp:596-600); best, keep and the counts are unchanged, asking for sims is a state error"""
    ring = _ring(1)
    ref = _reference(c, ring)[0]
    p = c.default_params(F)
    p.reserved = _lib.PAIR_NO_SIMS
    c.stereo_upload(*ring[0], slot=2)
    for _ in range(4):                                 # direct launches, then the captured graph of this flag
        c.stereo_submit(p, slot=2)
        cnt = c.stereo_wait(slot=2)
    assert (cnt.n_pairs, cnt.n_matches) == (ref[0].n_pairs, ref[0].n_matches)
    with pytest.raises(_lib.EbvoError):
        c.stereo_fetch(cnt, slot=2)                    # asks for sims
    with pytest.raises(_lib.EbvoError):
        c.stereo_fetch_begin(slot=2, what=_lib.FETCH_ALL)
    c.stereo_fetch_begin(slot=2, what=_lib.FETCH_DEFAULT)
    out = c.stereo_fetch_end(slot=2)
    assert_bit_equal(out["best"], ref[1]["best"], "best")
    assert_bit_equal(out["keep"], ref[1]["keep"], "keep")
    assert_bit_equal(out["col_idx"], ref[1]["col_idx"], "col_idx")
    p.reserved = 0                                     # and back: the scores are there again
    c.stereo_submit(p, slot=2)
    cnt = c.stereo_wait(slot=2)
    assert_bit_equal(c.stereo_fetch(cnt, slot=2)["sims"], ref[1]["sims"], "sims")


@pytest.mark.parametrize("stream_form", [0, 1])
def test_both_forms_of_the_asynchronous_upload(c, stream_form):
    """ebvo_debug_set 13: 0 = the pair's chain pulls the images from page-locked memory itself (default), 1 = copies on the
    context's upload stream + event; strided sources and a replayed submission included"""
    ring = _ring(2)
    ref = _reference(c, ring)
    wide = [tuple(np.ascontiguousarray(np.pad(im, ((0, 0), (0, 24)))) for im in pair) for pair in ring]   # stride = W + 24
    for pair in wide:
        for im in pair:
            c.host_register(im)
    c.debug_set(13, stream_form)
    try:
        p = c.default_params(F)
        for k in range(2):
            l, r = (im[:, :W] for im in wide[k])
            assert l.strides[0] == W + 24
            # (api.stereo_upload_async wants contiguous arrays: the strided call goes through the C entry directly)
            from edge_based_visual_odometry_amd.api import ptr
            c._check(c.lib.ebvo_stereo_upload_async(c._ctx, 3, ptr(wide[k][0]), ptr(wide[k][1]), H, W, W + 24, W + 24), "upload_async")
            for _ in range(2):                         # the second submission replays the same images
                c.stereo_submit(p, slot=3)
                cnt = c.stereo_wait(slot=3)
                _same(c.stereo_fetch(cnt, slot=3), ref[k])
    finally:
        c.debug_set(13, 0)
        c.stereo_upload(*ring[0], slot=3)              # the slot no longer refers to the registered memory
        for pair in wide:
            for im in pair:
                c.host_unregister(im)


@pytest.mark.parametrize("theta", [False, True])
@pytest.mark.parametrize("mode", ["push", "pack"])
def test_pushed_results_equal_full_fetch(c, theta, mode):
    """EBVO_PAIR_PUSH: the pair's chain writes the compact results into page-locked host memory itself; EBVO_PAIR_PACK: into
    device staging, fetched by ONE copy (ebvo_stereo_fetch_compact_begin / _end).  Either way what the host reads equals the
    full fetch -- over direct launches, the captured graph, a regrown pair buffer"""
    ring = _ring(3)
    ref = _reference(c, ring)
    p = c.default_params(F)
    p.reserved = (_lib.PAIR_PUSH if mode == "push" else _lib.PAIR_PACK) | _lib.PAIR_NO_SIMS | (_lib.PAIR_PUSH_THETA if theta else 0)
    for rep in range(5):                               # the third submission of a slot is a graph launch
        for k, (l, r) in enumerate(ring):
            c.stereo_upload(l, r, slot=1)
            if rep == 3 and k == 1:
                c.debug_set(1, 1)                      # the next result is treated as overflowed: matching half re-enqueued, arena kept in step
            c.stereo_submit(p, slot=1)
            cnt = c.stereo_wait(slot=1)
            if mode == "push":
                v = c.stereo_pushed_view(slot=1)
            else:
                c.stereo_fetch_compact_begin(slot=1, what=_lib.COMPACT_ALL if theta else _lib.COMPACT_DEFAULT)
                v = c.stereo_fetch_compact_end(slot=1)
            full = ref[k][1]
            assert (v["n_pairs"], v["n_matches"]) == (cnt.n_pairs, cnt.n_matches) == (ref[k][0].n_pairs, ref[k][0].n_matches)
            for side in ("left", "right"):
                assert_bit_equal(v[side + "_xy"][:, 0], full[side]["x"], side + ".x")
                assert_bit_equal(v[side + "_xy"][:, 1], full[side]["y"], side + ".y")
                if theta:
                    assert_bit_equal(v[side + "_theta"], full[side]["theta"], side + ".theta")
                else:
                    assert v[side + "_theta"] is None
            assert_bit_equal(v["row_ptr"], full["row_ptr"], "row_ptr")
            assert_bit_equal(v["col_idx"], full["col_idx"], "col_idx")
            assert_bit_equal(v["best"], full["best"], "best")
            bits = np.unpackbits(v["keep_bits"].view(np.uint8), bitorder="little")[: cnt.n_pairs]
            assert_bit_equal(bits, full["keep"], "keep bits")
    p.reserved = 0                                     # a pair without the flag leaves no pushed view
    c.stereo_submit(p, slot=1)
    cnt = c.stereo_wait(slot=1)
    with pytest.raises(_lib.EbvoError):
        c.stereo_pushed_view(slot=1)
    c.stereo_fetch_compact_begin(slot=1)               # ... and the compact fetch packs on the copy stream as before
    v = c.stereo_fetch_compact_end(slot=1)
    assert_bit_equal(v["best"], ref[2][1]["best"], "best")
    p.reserved = _lib.PAIR_PUSH | _lib.PAIR_PACK       # one destination at a time
    with pytest.raises(_lib.EbvoError):
        c.stereo_submit(p, slot=1)
