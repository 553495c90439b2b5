import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from edge_based_visual_odometry_amd import synth
from edge_based_visual_odometry_amd.api import Context
from tests import oracle as orc
F = synth.fundamental_21(synth.CALIB["kitti"]["K"], synth.CALIB["kitti"]["K"], synth.CALIB["kitti"]["R21"], synth.CALIB["kitti"]["T21"])
with Context(512, 1280) as ctx:
    l, r = synth.stereo_pair("s2", 120, 200)
    L, R = ctx.toed(l).edges, ctx.toed(r).edges
    lines = orc.epipolar_lines(F, L)
    for mask in (1, 7):
        rp, ci = orc.epi_candidates(L, R, lines, stage_mask=mask)
        grp, gci = ctx.epi_candidates(L, R, lines, stage_mask=mask)
        ec, gc = np.diff(rp), np.diff(grp)
        bad = np.nonzero(ec != gc)[0]
        print("mask", mask, "nL", len(L), "nR", len(R), "bad rows", len(bad), "expected total", rp[-1], "got", grp[-1])
        for i in bad[:5]:
            e = set(ci[rp[i]:rp[i+1]]); g = set(gci[grp[i]:grp[i+1]])
            print("  row", i, "exp", ec[i], "got", gc[i], "missing", sorted(e-g)[:10], "extra", sorted(g-e)[:10])
            print("     L", L[i]["x"], L[i]["y"], "line", lines[i], "R0", R[0]["x"], R[0]["y"], "R1", R[1]["x"], R[1]["y"])
        allmiss = np.concatenate([np.array(sorted(set(ci[rp[i]:rp[i+1]]) - set(gci[grp[i]:grp[i+1]])), dtype=np.int64) for i in bad[:400]]) if len(bad) else np.array([])
        if len(allmiss):
            u, c = np.unique(allmiss, return_counts=True)
            print("   distinct missing right edges:", len(u), "first", u[:20], "chunks", np.unique(u // 16)[:20])
