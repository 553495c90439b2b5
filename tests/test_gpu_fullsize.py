"""Full-size, oracle-checked runs of the resident pipeline: the EXACT pair bench.py times (S2, scene 7, noise 1 / 2,
shift 12, 1241x376) and the EuRoC / ETH3D configurations, every left edge, every candidate pair -- no strided subsets.

  * both edge lists bit-equal to the oracle (x, y, theta, index) and to the reference's known-answer xyi hashes;
  * the ENTIRE candidate CSR (all 126,184 rows at KITTI size) against the brute-force search of the oracle;
  * all four NCC scores, best and keep of every pair; the counts bench.py prints (581,657 / 472,947);
  * ebvo_stereo_finalize against the ORACLE-side chain (tests/oracle_chain.py), not against the HIP entry points.

The oracle side runs once per configuration (brute force: ~10-20 s on the GPU box's host cores) and is shared by both
detector modes.
"""
import functools

import numpy as np
import pytest

from edge_based_visual_odometry_amd import synth
from tests import oracle as orc
from tests import oracle_chain
from tests.util import assert_bit_equal, assert_edges_equal, kat_cases

pytestmark = pytest.mark.gpu

PAIRS = {
    # name: (config, generator arguments of synth.stereo_pair)
    "kitti": ("kitti", dict(scene=7, noise_base=0, disparity=12)),     # bench.py's pair, SURVEY 8(d) config 2
    "euroc": ("euroc", dict(scene=11, noise_base=4, disparity=9)),     # config 3 shape, slanted epipolar lines
    "eth3d": ("eth3d", dict(scene=11, noise_base=4, disparity=9)),     # config 4 shape
}


def _calib(cfg):
    c = synth.CALIB[cfg]
    kl = [c["K"][0], 0, c["K"][2], 0, c["K"][1], c["K"][3], 0, 0, 1]
    kr = [c["K_right"][0], 0, c["K_right"][2], 0, c["K_right"][1], c["K_right"][3], 0, 0, 1]
    return kl, kr, c["R21"], c["T21"]


@functools.lru_cache(maxsize=None)
def _oracle(name):
    cfg, args = PAIRS[name]
    h, w = synth.SHAPES[cfg]
    F = synth.fundamental_for(cfg)
    l, r = synth.stereo_pair("s2", h, w, **args)
    L, R = orc.toed(l)["edges"], orc.toed(r)["edges"]
    lines = orc.epipolar_lines(F, L)
    rp, ci = orc.epi_candidates(L, R, lines)
    sims, best, keep, _ = orc.ncc_pairs(l, r, L, R[ci], rp)
    return dict(l=l, r=r, F=F, left=L, right=R, row_ptr=rp, col_idx=ci, sims=sims, best=best, keep=keep)


@functools.lru_cache(maxsize=None)
def _oracle_chain(name, by_orientation=True, skip_single=False, sift=False):
    o = _oracle(name)
    return oracle_chain.stereo_edge_pairs(o["l"], o["r"], o["F"], _calib(PAIRS[name][0]), stage1=o,
                                          cluster_args=(by_orientation, skip_single), sift=sift)


@pytest.mark.parametrize("name", list(PAIRS))
def test_resident_pipeline_equals_oracle_everywhere(ctx, name):
    o = _oracle(name)
    ctx.stereo_upload(o["l"], o["r"])
    c = ctx.stereo_run(ctx.default_params(o["F"]))
    out = ctx.stereo_fetch(c)
    assert_edges_equal(out["left"], o["left"], "left edges")
    assert_edges_equal(out["right"], o["right"], "right edges")
    assert_bit_equal(out["row_ptr"], o["row_ptr"], "row_ptr")          # every row
    assert_bit_equal(out["col_idx"], o["col_idx"], "col_idx")          # every candidate pair, ascending right index
    assert_bit_equal(out["sims"], o["sims"], "sims")                   # pp, nn, pn, np of every pair
    assert_bit_equal(out["best"], o["best"], "best")
    assert_bit_equal(out["keep"], o["keep"], "keep")
    assert c.n_pairs == len(o["col_idx"]) and c.n_matches == int(o["keep"].sum())
    if name == "kitti":
        # what bench.py prints for its workload
        assert (c.n_left, c.n_right, c.n_pairs, c.n_matches) == (126184, 126340, 581657, 472947)
        kat = {(k["gen"], k["h"], k["w"], tuple(sorted(k["args"].items()))): k for k in kat_cases()}
        for side, edges, args in (("left", out["left"], dict(scene=7, noise_seed=1, shift=0)),
                                  ("right", out["right"], dict(scene=7, noise_seed=2, shift=12))):
            k = kat[("s2", 376, 1241, tuple(sorted(args.items())))]
            assert orc.edge_hash(edges, with_theta=False) == k["xyi"], side


@pytest.mark.parametrize("name", ["kitti", "euroc"])
def test_resident_chain_equals_oracle_chain(ctx, name):
    """get_Stereo_Edge_Pairs after the candidate stages (no SIFT): BNB -> shift -> refine -> shift + cluster by
    orientation -> NCC -> best -> non-empty rows -> output rows, device vs the chain of oracle functions."""
    o = _oracle(name)
    ref = _oracle_chain(name)
    ctx.stereo_upload(o["l"], o["r"])
    ctx.stereo_run(ctx.default_params(o["F"]))
    counts, fin = ctx.stereo_finalize(_calib(PAIRS[name][0]))
    assert counts == ref["counts"]
    assert_bit_equal(fin["left_index"], ref["left_index"], "left_index")
    assert_edges_equal(fin["right"], ref["right"], "right centre")
    assert_bit_equal(fin["score"], ref["score"], "score")
    assert_bit_equal(fin["rows"], ref["rows"], "rows")
    # the fixture separates the reference's clustering call (shift + by orientation, single rows included) from the
    # cluster-only reading: near-coincident candidates more than 20 degrees apart exist in it
    other = _oracle_chain(name, False, True)
    assert other["counts"]["n_clusters"] != ref["counts"]["n_clusters"] or not np.array_equal(
        other["right"]["x"], ref["right"]["x"])


def test_resident_chain_with_sift_equals_oracle_chain_at_kitti_size(ctx):
    """get_Stereo_Edge_Pairs stage for stage, the SIFT filter (500) and the Best-Nearly-Best test on the SIFT distances
    (0.4) included: ~252 k descriptors per image, every candidate pair scored."""
    o = _oracle("kitti")
    ref = _oracle_chain("kitti", sift=True)
    ctx.stereo_upload(o["l"], o["r"])
    ctx.stereo_run(ctx.default_params(o["F"]))
    counts, fin = ctx.stereo_finalize(_calib("kitti"), use_sift=True)
    assert counts == ref["counts"]
    assert_bit_equal(fin["left_index"], ref["left_index"], "left_index")
    assert_edges_equal(fin["right"], ref["right"], "right centre")
    assert_bit_equal(fin["score"], ref["score"], "score")
    assert_bit_equal(fin["rows"], ref["rows"], "rows")
    d = o["left"]["x"][fin["left_index"]] - fin["right"]["x"]
    assert np.median(np.abs(d - 12.0)) < 0.1
