"""Wall time of every call of the stage-wise drop-in sequence (host-buffer entry points) on the KITTI-shaped S2 pair."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from edge_based_visual_odometry_amd import synth
from edge_based_visual_odometry_amd.api import Context
h, w = synth.SHAPES["kitti"]; F = synth.fundamental_for("kitti")
l, r = synth.stereo_pair("s2", h, w)
with Context(h, w) as ctx:
    for rep in range(3):
        t = [time.perf_counter()]
        eL, eR, _ = ctx.toed_pair(l, r); t.append(time.perf_counter())
        lines = ctx.epipolar_lines(F, eL); t.append(time.perf_counter())
        rp2, ci2, ok = ctx.epi_candidates_staged(eL, eR, lines); t.append(time.perf_counter())
        kept_before = np.concatenate([[0], np.cumsum(ok, dtype=np.int64)])   # the binding's host step: drop the unflagged candidates
        rp = kept_before[rp2].astype(np.int32)
        ci = ci2[ok.view(np.bool_)]; cand = eR[ci]; t.append(time.perf_counter())
        out = ctx.ncc_pairs(l, r, eL, cand, rp, want_left_patches=True); t.append(time.perf_counter())
        out2 = ctx.ncc_pairs(l, r, eL, cand, rp, want_left_patches=False); t.append(time.perf_counter())
        names = ["toed_pair", "lines", "candidates_staged", "host row filter", "ncc_pairs+patches", "ncc_pairs"]
        print(rep, {n: round((b - a) * 1e3, 2) for n, a, b in zip(names, t, t[1:])}, len(ci2), len(ci))
