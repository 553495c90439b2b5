"""PCIe-inclusive rate: a frame loop that uploads a NEW stereo pair per step (host buffers -> HBM), runs the hot path,
and optionally fetches the results, with 3 pairs in flight.  Never used as bench.py's `value` (that one starts with the
inputs resident); quoted in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from edge_based_visual_odometry_amd import synth
from edge_based_visual_odometry_amd.api import Context

h, w = synth.SHAPES["kitti"]
F = synth.fundamental_for("kitti")
pool = [synth.stereo_pair("s2", h, w, noise_base=10 * k) for k in range(6)]
N, S = 240, 3
with Context(h, w, toed_mode="hybrid") as c:
    c.set_slots(S)
    p = c.default_params(F)
    for mode in ("resident", "upload", "upload+fetch"):
        for k in range(S):
            c.stereo_upload(*pool[k], slot=k)
            c.stereo_submit(p, slot=k); c.stereo_wait(slot=k)
        t0 = time.perf_counter()
        sub = done = 0
        while sub < S:
            if mode != "resident":
                c.stereo_upload(*pool[sub % len(pool)], slot=sub % S)
            c.stereo_submit(p, slot=sub % S); sub += 1
        nbytes = 0
        while done < N:
            k = done % S
            cnt = c.stereo_wait(slot=k); done += 1
            if mode == "upload+fetch":
                out = c.stereo_fetch(cnt, slot=k)
                nbytes += sum(v.nbytes for v in out.values() if v is not None)
            if sub < N:
                if mode != "resident":
                    c.stereo_upload(*pool[sub % len(pool)], slot=k)
                c.stereo_submit(p, slot=k); sub += 1
        dt = time.perf_counter() - t0
        print(f"{mode:14s} {N / dt:8.1f} pairs/s  {dt / N * 1e3:6.3f} ms/pair" + (f"  ({nbytes / N / 1e6:.1f} MB fetched per pair)" if nbytes else ""))
