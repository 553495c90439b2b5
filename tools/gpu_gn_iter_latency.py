"""Run ON the GPU box: duration of the refinement stage of the one-pass chain against the iteration cap (the pairs that never
converge run every iteration, one after the other: the slope is the latency of one iteration of one pair)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from edge_based_visual_odometry_amd import synth
from edge_based_visual_odometry_amd.api import Context

for cfg, disp in (("euroc", 9), ("kitti", 12)):
    h, w = synth.SHAPES[cfg]
    F = synth.fundamental_for(cfg)
    l, r = synth.stereo_pair("s2", h, w, disparity=disp)
    with Context(h, w) as c:
        c.stereo_upload(l, r)
        c.stereo_run(c.default_params(F))
        c.stereo_finalize(None, use_sift=True)
        for mi in (1, 2, 4, 8, 12, 16, 20):
            c.profile_reset(); c.profile_enable(True)
            for _ in range(3):
                counts, _ = c.stereo_finalize(None, use_sift=True, max_iter=mi)
            c.profile_enable(False)
            gn = c.profile_get()["gn_refine"][0] / 3
            print(f"{cfg}: max_iter {mi:2d}: refinement stage {gn * 1e3:7.1f} us  (pairs {counts['n_bnb']})")
