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
    def run(sel, Rs, tag):
        Ls, ls = L[sel], lines[sel]
        rp, ci = orc.epi_candidates(Ls, Rs, ls, stage_mask=1)
        grp, gci = ctx.epi_candidates(Ls, Rs, ls, stage_mask=1)
        print(tag, "expected", rp[-1], "got", grp[-1], "rowdiff", (np.diff(grp) - np.diff(rp))[:16])
    run(np.array([2]), R, "single row 2, all R")
    run(np.array([2]), R[:16], "single row 2, R[:16]")
    run(np.array([2]), R[:64], "single row 2, R[:64]")
    run(np.array([2]), R[:1024], "single row 2, R[:1024]")
    run(np.array([2]), R[:1040], "single row 2, R[:1040]")
    run(np.arange(14), R[:1024], "rows 0..13, R[:1024]")
    run(np.arange(14), R[:4096], "rows 0..13, R[:4096]")
    run(np.arange(14), R[:4112], "rows 0..13, R[:4112]")
    run(np.arange(14), R, "rows 0..13, R all")
