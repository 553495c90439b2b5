#!/usr/bin/env python3
"""Run ON the GPU box: where the host thread of the pipelined sequence loop spends its time -- in the enqueue-only calls
(launch overhead) or in the waits (the device is the bottleneck).  EuRoC-size frames, bench.py's pipeline."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edge_based_visual_odometry_amd import synth  # noqa: E402
from edge_based_visual_odometry_amd.api import Context  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "euroc"
STAGES = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # 1: every stage of the temporal chain (its tail runs inside temporal_wait)
ONLY = tuple(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else None   # one lag setting (for a kernel trace)
H, W = synth.SHAPES[cfg]
cal = synth.CALIB[cfg]
F = synth.fundamental_for(cfg)
calib = ([cal["K"][0], 0, cal["K"][2], 0, cal["K"][1], cal["K"][3], 0, 0, 1],
         [cal["K_right"][0], 0, cal["K_right"][2], 0, cal["K_right"][1], cal["K_right"][3], 0, 0, 1], cal["R21"], cal["T21"])
n_frames = 16
ctx = Context(H, W)
ctx.set_slots(n_frames)
if "dist" in cal:
    ctx.set_undistort(cal["K"], cal["dist"], cal["K_right"], cal["dist_right"])
params = ctx.default_params(F)
for k in range(n_frames):
    l, r = synth.stereo_pair("s2", H, W, scene=7, noise_base=2 * k, disparity=9 if cfg != "kitti" else 12)
    ctx.stereo_upload(np.roll(l, k, axis=1), np.roll(r, k, axis=1), slot=k)
ctx.stereo_submit(params, slot=0)
ctx.stereo_wait(slot=0)
ctx.stereo_finalize(calib, slot=0, use_sift=True)
ctx.temporal_set_keyframe(slot=0)
acc = dict(stereo_submit=0.0, stereo_wait=0.0, finalize_submit=0.0, finalize_wait=0.0, temporal_submit=0.0, temporal_wait=0.0)


def timed(key, fn, *a, **k):
    t = time.perf_counter()
    r = fn(*a, **k)
    acc[key] += time.perf_counter() - t
    return r


def pipeline(slots, lag_b=1, lag_c=2, lag_d=3):
    n = len(slots)
    for i in range(n + lag_d):
        if i < n:
            timed("stereo_submit", ctx.stereo_submit, params, slot=slots[i])
        if 0 <= i - lag_b < n:
            timed("stereo_wait", ctx.stereo_wait, slot=slots[i - lag_b])
            timed("finalize_submit", ctx.stereo_finalize_submit, calib, slot=slots[i - lag_b], use_sift=True)
        if 0 <= i - lag_c < n:
            timed("finalize_wait", ctx.stereo_finalize_wait, slot=slots[i - lag_c], fetch=False)
            timed("temporal_submit", ctx.temporal_match_submit, slot=slots[i - lag_c], stages=STAGES)
        if 0 <= i - lag_d < n:
            timed("temporal_wait", ctx.temporal_match_wait, slot=slots[i - lag_d], fetch=False)


pipeline(list(range(n_frames)))
for k in acc:
    acc[k] = 0.0
steps = 96
for lags in ((ONLY,) if ONLY else ((1, 2, 3), (1, 3, 4), (1, 4, 5), (2, 4, 5), (2, 5, 6), (2, 6, 8), (3, 8, 10))):
    for k in acc:
        acc[k] = 0.0
    t0 = time.perf_counter()
    pipeline([i % n_frames for i in range(steps)], *lags)
    dt = time.perf_counter() - t0
    print(f"{cfg} lags {lags}: {steps / dt:.1f} frames/s, {dt / steps * 1e3:.3f} ms per frame; host ms per frame: " +
          ", ".join(f"{k} {v / steps * 1e3:.3f}" for k, v in acc.items()))
