"""Wall time of ebvo_stereo_finalize on the KITTI-shaped S2 pair."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from edge_based_visual_odometry_amd import synth
from edge_based_visual_odometry_amd.api import Context

h, w = synth.SHAPES["kitti"]
F = synth.fundamental_for("kitti")
cal = synth.CALIB["kitti"]
fx, fy, cx, cy = cal["K"]
K = [fx, 0, cx, 0, fy, cy, 0, 0, 1]
l, r = synth.stereo_pair("s2", h, w)
with Context(h, w, toed_mode="hybrid") as c:
    c.stereo_upload(l, r)
    c.stereo_run(c.default_params(F))
    c.stereo_finalize((K, K, cal["R21"], cal["T21"]))
    for _ in range(2):
        c.stereo_upload(l, r)
        t0 = time.perf_counter(); cnt = c.stereo_run(c.default_params(F)); t1 = time.perf_counter()
        c.profile_reset(); c.profile_enable(True)
        counts, fin = c.stereo_finalize((K, K, cal["R21"], cal["T21"]))
        t2 = time.perf_counter()
        c.profile_enable(False)
    prof = c.profile_get()
    print("run %.2f ms, finalize (incl. fetch of %d final pairs) %.2f ms" % ((t1 - t0) * 1e3, counts["n_final"], (t2 - t1) * 1e3), counts)
    print({k: round(v[0], 3) for k, v in prof.items() if v[1]})
    c.stereo_upload(l, r)
    c.stereo_run(c.default_params(F))
    c.profile_reset(); c.profile_enable(True)
    t0 = time.perf_counter()
    counts_s, fin_s = c.stereo_finalize((K, K, cal["R21"], cal["T21"]), use_sift=True)
    t1 = time.perf_counter()
    c.profile_enable(False)
    print("with the SIFT stages: finalize %.2f ms" % ((t1 - t0) * 1e3), counts_s)
    print({k: round(v[0], 3) for k, v in c.profile_get().items() if v[1]})
    d = fin["rows"][:, 0] - fin["rows"][:, 3]
    print("median |disparity - 12| =", float(np.median(np.abs(d - 12))), "median |depth| =", float(np.median(np.abs(fin["rows"][:, 8]))))
