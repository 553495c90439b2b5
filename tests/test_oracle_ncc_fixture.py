"""Pin the NCC restatement against the only numeric NCC fixture the reference ships:
test/ncc_debug_frame1_edge8 (patch PNGs + patch_statistics.txt), committed here as JSON by
tests/golden/make_ncc_fixture.py.  Order of the four scores: PP, MM, PM, MP
(src/Stereo_Matches.cpp:592-595)."""
import json
import os

import numpy as np

from tests import oracle as orc
from tests.util import GOLDEN


def test_ncc_restatement_reproduces_reference_fixture():
    d = json.load(open(os.path.join(GOLDEN, "ncc_debug_frame1_edge8.json")))
    prev = d["patches"]["prev"]
    assert len(d["expected_vs_prev"]) == 6
    for name, exp in d["expected_vs_prev"].items():
        c = d["patches"][name]
        got = [orc.patch_similarity(prev["plus"], c["plus"]), orc.patch_similarity(prev["minus"], c["minus"]),
               orc.patch_similarity(prev["plus"], c["minus"]), orc.patch_similarity(prev["minus"], c["plus"])]
        got.append(max(got))
        assert np.max(np.abs(np.array(got) - np.array(exp))) <= d["tolerance"], (name, got, exp)


def test_ncc_against_exact_math():
    """|canonical arithmetic - exact NCC| <= 1e-5 (the north-star tolerance) on random patches."""
    rng = np.random.default_rng(3)
    for _ in range(200):
        a = rng.integers(0, 256, 49).astype(np.float32) + rng.random(49).astype(np.float32)
        b = (0.5 * a + rng.normal(0, 20, 49)).astype(np.float32)
        got = orc.patch_similarity(a, b)
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        da, db = a64 - a64.mean(), b64 - b64.mean()
        exact = (da @ db) / np.sqrt((da @ da) * (db @ db))
        assert abs(got - exact) <= 1e-5


def test_ncc_sentinels():
    flat = np.full(49, 37.0, dtype=np.float32)
    ramp = np.arange(49, dtype=np.float32)
    assert orc.patch_similarity(flat, ramp) == -1.0     # src/utility.cpp:170-172
    assert orc.patch_similarity(ramp, flat) == -1.0
    assert orc.patch_similarity(ramp, ramp) > 0.999999
    nanp = ramp.copy()
    nanp[5] = np.nan
    assert np.isnan(orc.patch_similarity(nanp, ramp))   # NaN is not < 1e-10: flows through
