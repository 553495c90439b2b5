"""orc_temporal_candidates against a direct Python reading of the reference: SpatialGrid insertion
(src/Temporal_Matches.cpp:25-40), getCandidatesWithinRadius (include/Dataset.h:92-113: dy outer, dx inner, a cell's
indices in insertion order), the right-grid membership test and both orientation gates (:335-414) -- same candidates in
the SAME ORDER, edges outside the grid included."""
import math

import numpy as np

from tests import oracle as orc


def _mates(rng, n, w, h):
    e = np.zeros(n, dtype=orc.EDGE_DTYPE)
    e["x"] = rng.uniform(-20, w + 20, n)          # some edges outside the grid on every side
    e["y"] = rng.uniform(-20, h + 20, n)
    e["theta"] = rng.uniform(-math.pi, math.pi, n)
    e["index"] = np.arange(n)
    return e


def _cpp_int_div(a, b):
    return int(a / b) if a * b < 0 else a // b    # C++ integer division truncates towards zero


def _reading(kfL, kfR, cfL, cfR, w, h, cell, radius, thr):
    gw, gh = (w + cell - 1) // cell, (h + cell - 1) // cell
    sr = int(math.ceil(radius / cell))

    def cell_of(e):
        return _cpp_int_div(int(e["x"]), cell), _cpp_int_div(int(e["y"]), cell)

    def grid(edges):
        g = {}
        for j, e in enumerate(edges):
            cx, cy = cell_of(e)
            if 0 <= cx < gw and 0 <= cy < gh:
                g.setdefault((cx, cy), []).append(j)
        return g

    def query(g, e):
        qx, qy = cell_of(e)
        out = []
        for dy in range(-sr, sr + 1):
            for dx in range(-sr, sr + 1):
                nx, ny = qx + dx, qy + dy
                if 0 <= nx < gw and 0 <= ny < gh:
                    out += g.get((nx, ny), [])
        return out

    def close(a, b):
        od = abs((a - b) * (180.0 / math.pi))
        if od > 180.0:
            od = 360.0 - od
        return od < thr or abs(od - 180.0) < thr

    gl, gr = grid(cfL), grid(cfR)
    rp, ci = [0], []
    for i in range(len(kfL)):
        right_set = set(query(gr, kfR[i]))
        for j in query(gl, kfL[i]):
            if j in right_set and close(kfL[i]["theta"], cfL[j]["theta"]) and close(kfR[i]["theta"], cfR[j]["theta"]):
                ci.append(j)
        rp.append(len(ci))
    return np.array(rp, dtype=np.int32), np.array(ci, dtype=np.int32)


def test_candidates_in_the_order_of_the_reference():
    rng = np.random.default_rng(5)
    w, h, cell = 200, 130, 15
    cfL = _mates(rng, 1500, w, h)
    cfR = cfL.copy()
    cfR["x"] -= rng.uniform(0, 40, len(cfR))      # a disparity: the right cells differ from the left ones
    cfR["theta"] += rng.normal(0, 0.05, len(cfR))
    pick = rng.choice(len(cfL), 300, replace=False)
    kfL, kfR = cfL[pick].copy(), cfR[pick].copy()
    kfL["x"] += rng.normal(0, 6, len(pick))
    kfL["y"] += rng.normal(0, 6, len(pick))
    kfR["x"] += rng.normal(0, 6, len(pick))
    kfR["y"] += rng.normal(0, 6, len(pick))
    kfL["theta"] += rng.normal(0, 0.1, len(pick))
    for radius, thr in ((30.0, 10.0), (15.0, 45.0), (31.0, 180.0)):
        rp, ci = orc.temporal_candidates(kfL, kfR, cfL, cfR, w, h, cell, radius, thr)
        want_rp, want_ci = _reading(kfL, kfR, cfL, cfR, w, h, cell, radius, thr)
        assert np.array_equal(rp, want_rp) and np.array_equal(ci, want_ci)
        assert len(ci) > len(kfL)
    # the order is not the ascending index order: rows exist whose candidates go back in index (cell-major walk)
    assert any((np.diff(ci[rp[i]:rp[i + 1]]) < 0).any() for i in range(len(kfL)))
