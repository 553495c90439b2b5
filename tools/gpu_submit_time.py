import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_based_visual_odometry_amd import synth
from edge_based_visual_odometry_amd.api import Context
F = synth.fundamental_for("kitti")
l, r = synth.stereo_pair("s2", 376, 1241)
for mode in ("hybrid", "strict"):
    for prof in (False, True):
        with Context(376, 1241, toed_mode=mode) as c:
            c.stereo_upload(l, r); p = c.default_params(F)
            for _ in range(3): c.stereo_run(p)
            c.profile_enable(prof)
            ts = []
            for _ in range(20):
                t0 = time.perf_counter(); c.stereo_submit(p); t1 = time.perf_counter(); c.stereo_wait(); t2 = time.perf_counter()
                ts.append((t1 - t0, t2 - t0))
            ts.sort()
            print(mode, "profiling" if prof else "no-prof", "submit median %.3f ms" % (ts[10][0] * 1e3), "pair median %.3f ms" % (sorted(t[1] for t in ts)[10] * 1e3))
