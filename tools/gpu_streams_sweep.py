"""Throughput of the resident loop against the number of pairs in flight, and the host time of one submit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edge_based_visual_odometry_amd import synth
from edge_based_visual_odometry_amd.api import Context

h, w = synth.SHAPES["kitti"]
F = synth.fundamental_for("kitti")
l, r = synth.stereo_pair("s2", h, w)
N = 150
for S in (1, 2, 3, 4, 5, 6, 8):
    with Context(h, w, toed_mode="hybrid") as c:
        c.set_slots(S)
        p = c.default_params(F)
        for k in range(S):
            c.stereo_upload(l, r, slot=k)
            for _ in range(3):
                c.stereo_submit(p, slot=k); c.stereo_wait(slot=k)
        t_sub = 0.0
        t0 = time.perf_counter()
        sub = done = 0
        while sub < S:
            a = time.perf_counter(); c.stereo_submit(p, slot=sub % S); t_sub += time.perf_counter() - a; sub += 1
        while done < N:
            k = done % S
            c.stereo_wait(slot=k); done += 1
            if sub < N:
                a = time.perf_counter(); c.stereo_submit(p, slot=k); t_sub += time.perf_counter() - a; sub += 1
        dt = time.perf_counter() - t0
        print(f"slots {S}: {N / dt:8.1f} pairs/s  {dt / N * 1e3:6.3f} ms/pair   host submit {t_sub / N * 1e6:6.1f} us/pair", flush=True)
