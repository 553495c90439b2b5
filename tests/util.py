import json
import os

import numpy as np

from edge_based_visual_odometry_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def kat_cases():
    return json.load(open(os.path.join(GOLDEN, "toed_kat.json")))["cases"]


def kat_image(case):
    gen = synth.s1_image if case["gen"] == "s1" else synth.s2_image
    return gen(case["h"], case["w"], **case["args"])


def case_id(case):
    return f'{case["gen"]}-{case["w"]}x{case["h"]}-' + "-".join(str(v) for v in case["args"].values())


def bits(a):
    """float array -> integer bit patterns (so NaN == NaN and -0.0 != +0.0)."""
    a = np.ascontiguousarray(a)
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def assert_bit_equal(a, b, what=""):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if a.dtype.kind == "f":
        # identical bit patterns, except that any NaN matches any NaN (x86 and gfx950 produce
        # default NaNs of opposite sign; the reference only ever tests NaN-ness)
        bad = (bits(a) != bits(b)) & ~(np.isnan(a) & np.isnan(b))
    else:
        bad = a != b
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} elements differ; first at {np.argwhere(bad)[0]}"


def assert_edges_equal(a, b, what="edges"):
    assert len(a) == len(b), f"{what}: {len(a)} vs {len(b)}"
    for f in ("x", "y", "theta"):
        assert_bit_equal(a[f], b[f], f"{what}.{f}")
    assert (a["index"] == b["index"]).all(), f"{what}.index"
