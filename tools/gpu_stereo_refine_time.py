"""Wall time of ebvo_stereo_refine (the call alone, results left on the device) on the KITTI-shaped S2 pair."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_based_visual_odometry_amd import synth
from edge_based_visual_odometry_amd.api import Context

h, w = synth.SHAPES["kitti"]
F = synth.fundamental_for("kitti")
l, r = synth.stereo_pair("s2", h, w)
with Context(h, w, toed_mode="hybrid") as c:
    c.stereo_upload(l, r)
    cnt = c.stereo_run(c.default_params(F))
    p = c._gn_params()
    for _ in range(2):
        c._check(c.lib.ebvo_stereo_refine(c._ctx, 0, C.byref(p)), "ebvo_stereo_refine")
    c.profile_reset()
    c.profile_enable(True)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        c._check(c.lib.ebvo_stereo_refine(c._ctx, 0, C.byref(p)), "ebvo_stereo_refine")
    dt = (time.perf_counter() - t0) / n
    c.profile_enable(False)
    prof = c.profile_get()
    print("ebvo_stereo_refine: %.3f ms per call, %d pairs (%d kept)" % (dt * 1e3, cnt.n_pairs, cnt.n_matches))
    print({k: (round(v[0] / n, 3), v[1] // n) for k, v in prof.items() if v[1]})
