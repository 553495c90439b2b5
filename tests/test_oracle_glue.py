"""Stage-glue restatements (oracle/): Best-Nearly-Best test, keep-best, epipolar shift -- checked against a direct
Python reading of src/Stereo_Matches.cpp:789-862, :916-964, :26-89 (PARITY UNPINNED by reference fixtures)."""
import math

import numpy as np

from edge_based_visual_odometry_amd import synth
from tests import oracle as orc


def _rows(rng, n_rows=400, max_len=24):
    lens = rng.integers(0, max_len, n_rows)
    lens[::7] = 1
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return rp, int(rp[-1])


def _bnb_python(rp, sc, thr, higher):
    cnt, order = [], np.full(len(sc), -1, dtype=np.int32)
    for i in range(len(rp) - 1):
        b, n = rp[i], rp[i + 1] - rp[i]
        idx = list(range(n))
        keep = n
        if n >= 2:
            # std::sort with the reference's comparator; Python's sort is stable like libstdc++'s small-row insertion sort
            idx.sort(key=lambda a: -sc[b + a] if higher else sc[b + a])
            best = sc[b + idx[0]]
            keep = 1
            for j in range(n - 1):
                nxt = sc[b + idx[j + 1]]
                if best == 0:
                    break
                ratio = nxt / best if higher else best / nxt
                if ratio >= thr:
                    keep += 1
                else:
                    break
            if keep == n:
                idx = list(range(n))
        cnt.append(keep)
        order[b:b + n] = [b + a for a in idx]
    return np.array(cnt, dtype=np.int32), order


def test_bnb_matches_python_reading():
    rng = np.random.default_rng(1)
    rp, n = _rows(rng)
    for higher, thr in ((True, 0.9), (False, 0.4)):
        sc = rng.uniform(0.3, 1.0, n) if higher else rng.uniform(50, 400, n)
        sc[::11] = sc[1::11][: len(sc[::11])] if len(sc[1::11]) >= len(sc[::11]) else sc[::11]   # a few exact ties
        sc[5] = 0.0
        cnt, order = orc.bnb_test(rp, sc, thr, higher)
        pc, po = _bnb_python(rp, sc, thr, higher)
        assert np.array_equal(cnt, pc)
        for i in range(len(rp) - 1):
            b = rp[i]
            if rp[i + 1] - b <= 16:                           # std::sort == a stable sort only up to 16 entries (ebvo_sort.h)
                assert np.array_equal(order[b:b + cnt[i]], po[b:b + pc[i]])
            else:                                             # same scores in the same places, ties possibly permuted
                assert np.array_equal(sc[order[b:b + cnt[i]]], sc[po[b:b + pc[i]]])
            if cnt[i] == rp[i + 1] - b:                       # nothing dropped: original order
                assert np.array_equal(order[b:rp[i + 1]], np.arange(b, rp[i + 1]))


def test_temporal_bnb_variant_always_returns_sorted_rows():
    """Temporal_Matches::apply_best_nearly_best_filtering_quads (src/Temporal_Matches.cpp:517-570): the same ratio test, but a
    row of two or more quads is rebuilt from the sorted indices whether or not something was dropped."""
    rng = np.random.default_rng(8)
    rp, n = _rows(rng)
    for higher, thr in ((True, 0.8), (False, 0.8)):
        sc = rng.uniform(0.5, 1.0, n) if higher else rng.uniform(20, 200, n)
        sc[7] = 0.0
        cnt, order = orc.bnb_test(rp, sc, thr, higher, always_sorted=True)
        pc, po = _bnb_python(rp, sc, thr, higher)                   # the stereo reading: same counts, same kept prefix
        assert np.array_equal(cnt, pc)
        untouched = 0
        for i in range(len(rp) - 1):
            b, m = rp[i], rp[i + 1] - rp[i]
            kept = order[b:b + cnt[i]]
            if m >= 2:                                              # :531-541 sorted, :553-558 rebuilt from indices[0 .. keep)
                want = sorted(range(b, b + m), key=lambda k: (-sc[k] if higher else sc[k]))[:cnt[i]]
                if m <= 16:
                    assert list(kept) == want
                else:
                    assert np.array_equal(sc[kept], sc[want])
                untouched += int(cnt[i] == m and list(kept) == list(range(b, b + m)))
            else:
                assert list(kept) == list(range(b, b + m))          # :524 `if (n < 2) continue`
        assert untouched < (np.diff(rp) >= 2).sum()                 # rows that lost nothing were still reordered


def test_std_sort_restatement_equals_libstdcxx(tmp_path):
    """csrc/ebvo_sort.h against the real std::sort of this toolchain (tests/cpp/sort_check.cpp): ties, both comparators,
    lengths up to 5000."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "sort_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(root, "tests", "cpp", "sort_check.cpp"),
                           "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout


def test_bnb_long_rows_with_ties_follow_std_sort():
    """Rows of more than 16 candidates with tied scores at the cut: the survivors are the ones libstdc++'s std::sort puts
    first, not the ones a stable sort would (the two differ on this input)."""
    rng = np.random.default_rng(21)
    lens = rng.integers(17, 120, 300)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    sc = rng.choice([0.95, 0.9, 0.86, 0.7, 0.5], rp[-1])            # heavy ties, several of them at the 0.9 ratio cut
    cnt, order = orc.bnb_test(rp, sc, 0.9, True)
    stable_differs = 0
    for i in range(len(lens)):
        b, n = rp[i], lens[i]
        kept = order[b:b + cnt[i]]
        assert len(set(kept.tolist())) == cnt[i] and ((kept >= b) & (kept < b + n)).all()
        s = sc[kept]
        assert (np.diff(s) <= 0).all()                               # descending
        best = s[0]
        assert best == sc[b:b + n].max() and (s / best >= 0.9).all()
        assert cnt[i] == int((sc[b:b + n] / best >= 0.9).sum())      # ties never change HOW MANY survive here
        stable = np.argsort(-sc[b:b + n], kind="stable")[:cnt[i]] + b
        stable_differs += int(not np.array_equal(stable, kept))
    assert stable_differs > 0


def test_keep_best():
    rng = np.random.default_rng(2)
    rp, n = _rows(rng)
    sc = rng.uniform(-1.5, 1.0, n)
    cnt, order = orc.keep_best(rp, sc)
    for i in range(len(rp) - 1):
        b, m = rp[i], rp[i + 1] - rp[i]
        assert cnt[i] == (1 if m else 0)
        if m:
            best, mx = 0, -1.0
            for j in range(m):
                if sc[b + j] > mx:
                    mx, best = sc[b + j], j
            assert order[b] == b + best


def test_epipolar_shift_geometry():
    """Shifted candidates lie on their epipolar line (when they moved at all), within the reference's displacement
    thresholds, and the small-normal-distance branch is the orthogonal projection."""
    F = synth.fundamental_for("euroc")                         # slanted lines
    l, r = synth.stereo_pair("s2", 96, 160)
    L = orc.toed(l)["edges"][:300]
    R = orc.toed(r)["edges"]
    lines = orc.epipolar_lines(F, L)
    rng = np.random.default_rng(3)
    per = rng.integers(0, 4, len(L))
    rp = np.concatenate([[0], np.cumsum(per)]).astype(np.int32)
    cand = R[rng.integers(0, len(R), rp[-1])].copy()
    rows = np.repeat(np.arange(len(L)), per)
    a, b, c = lines[rows].T
    # put a third of them close to their line (normal distance < 0.4) to exercise the projection branch
    near = np.arange(len(cand)) % 3 == 0
    d = (a * cand["x"] + b * cand["y"] + c) / (a * a + b * b)
    cand["x"][near] -= (a * d)[near] * 0.999
    cand["y"][near] -= (b * d)[near] * 0.999
    out = orc.epipolar_shift(cand, lines, rp, math_mode=orc.LIBM)
    moved = (out["x"] != cand["x"]) | (out["y"] != cand["y"])
    res = np.abs(a * out["x"] + b * out["y"] + c) / np.sqrt(a * a + b * b)
    assert moved[near].all() and np.all(res[moved] < 1e-9)
    disp = np.hypot(out["x"] - cand["x"], out["y"] - cand["y"])
    assert np.all(disp[moved] < 3.0 + 1e-12)
    assert np.all(out["theta"][near] == cand["theta"][near])
    turned = out["theta"] != cand["theta"]
    assert np.allclose(np.abs(out["theta"] - cand["theta"])[turned], 0.174533)
    assert np.all(out["index"] == 0)
    # portable and libm modes agree to rounding
    out2 = orc.epipolar_shift(cand, lines, rp, math_mode=orc.PORTABLE)
    same_branch = (out2["theta"] == out["theta"]) & ((out2["x"] != cand["x"]) == moved)
    assert same_branch.mean() > 0.99
    assert np.allclose(out2["x"][same_branch], out["x"][same_branch], rtol=0, atol=1e-9)


def _cluster_python(E, by_orientation):
    """EdgeClusterer::performClustering, src/EdgeClusterer.cpp:119-302, read directly."""
    n = len(E)
    lab = list(range(n))
    thr = math.radians(20.0)

    def gauss(label):
        idx = [i for i in range(n) if lab[i] == label]
        sx = sy = 0.0
        for i in idx:
            sx += E["x"][i]
            sy += E["y"][i]
        cx, cy = sx / len(idx), sy / len(idx)
        tot = 0.0
        for i in idx:
            tot += math.sqrt((E["x"][i] - cx) ** 2 + (E["y"][i] - cy) ** 2)
        mean = tot / len(idx)
        wx = wy = wt = w = 0.0
        for i in idx:
            d = math.sqrt((E["x"][i] - cx) ** 2 + (E["y"][i] - cy) ** 2)
            g = math.exp(-0.5 * ((d - mean) / 2.0) ** 2)
            wx += g * E["x"][i]
            wy += g * E["y"][i]
            wt += g * E["theta"][i]
            w += g
        return wx / w, wy / w, wt / w

    merged = True
    while merged:
        merged = False
        for i in range(n):
            md, nearest = float("inf"), -1
            for j in range(n):
                if lab[i] != lab[j]:
                    dist = math.sqrt((E["x"][i] - E["x"][j]) ** 2 + (E["y"][i] - E["y"][j]) ** 2)
                    ok = dist < md and dist < 1
                    if by_orientation:
                        ok = ok and abs(E["theta"][i] - E["theta"][j]) < thr
                    if ok:
                        md, nearest = dist, j
            if nearest != -1:
                old, new = lab[nearest], lab[i]
                if lab.count(old) + lab.count(new) <= 10:
                    lab = [new if v == old else v for v in lab]
                    merged = True
                    break
    uniq = sorted(set(lab))
    return [uniq.index(v) for v in lab], [gauss(u) for u in uniq]


def test_cluster_rows_matches_python_reading():
    rng = np.random.default_rng(6)
    lens = rng.integers(0, 18, 300)
    lens[::9] = 1
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cand = np.zeros(rp[-1], dtype=orc.EDGE_DTYPE)
    # candidates along a line with gaps around the 1-px threshold, a few dense groups that hit the size cap
    for i in range(len(lens)):
        b, n = rp[i], lens[i]
        x = np.cumsum(rng.choice([0.2, 0.6, 0.95, 1.05, 2.5], n)) + 50
        cand["x"][b:b + n] = x[rng.permutation(n)]
        cand["y"][b:b + n] = 30 + rng.uniform(-0.2, 0.2, n)
        cand["theta"][b:b + n] = rng.choice([0.3, 0.5, 1.2], n) + rng.uniform(-0.05, 0.05, n)
    for by_orient in (False, True):
        for skip in (True, False):
            cnt, centres, cof = orc.cluster_rows(cand, rp, by_orient, skip, orc.LIBM)   # math.exp below is glibc's
            for i in range(len(lens)):
                b, n = rp[i], lens[i]
                if n == 0:
                    assert cnt[i] == 0
                    continue
                if n == 1 and skip:
                    assert cnt[i] == 1 and centres[b] == cand[b]
                    continue
                lab, cen = _cluster_python(cand[b:b + n], by_orient)
                assert cnt[i] == len(cen)
                assert list(cof[b:b + n]) == lab
                for c, (gx, gy, gt) in enumerate(cen):
                    assert centres["x"][b + c] == gx and centres["y"][b + c] == gy and centres["theta"][b + c] == gt
            sizes = np.bincount(np.repeat(np.arange(len(lens)), lens) * 64 + cof)
            assert sizes.max() <= 10                                   # MAX_CLUSTER_SIZE
