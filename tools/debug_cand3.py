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
    rp, ci = orc.epi_candidates(L, R, lines, stage_mask=1)
    grp, gci = ctx.epi_candidates(L, R, lines, stage_mask=1)
    for i in (2, 12, 13, 300, 301):
        e = ci[rp[i]:rp[i+1]]; g = set(gci[grp[i]:grp[i+1]])
        print("row", i, "L", round(L[i]["x"],1), round(L[i]["y"],2))
        print("   ", " ".join(f"{k}:{R[k]['x']:.0f},{R[k]['y']:.1f}{'' if k in g else '!'}" for k in e))
