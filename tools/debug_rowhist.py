"""Histogram of candidates per left edge on the bench workload (sizing of the candidate staging area)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from edge_based_visual_odometry_amd import synth
from edge_based_visual_odometry_amd.api import Context

h, w = synth.SHAPES["kitti"]
l, r = synth.stereo_pair("s2", h, w)
with Context(h, w, toed_mode="hybrid") as c:
    c.stereo_upload(l, r)
    p = c.default_params(synth.fundamental_for("kitti"))
    cnt = c.stereo_run(p)
    out = c.stereo_fetch(cnt)
rp = out["row_ptr"].astype(np.int64)
n = np.diff(rp)
print("rows", len(n), "pairs", rp[-1], "mean", n.mean(), "max", n.max())
for t in (8, 16, 24, 32, 48, 64):
    rows = (n > t).mean()
    tiles = [(n[i:i + 256] > t).any() for i in range(0, len(n), 256)]
    w64 = [(n[i:i + 64] > t).any() for i in range(0, len(n), 64)]
    print(f"> {t}: rows {rows:.4f} tiles256 {np.mean(tiles):.3f} waves64 {np.mean(w64):.3f} pairs-in-such-rows {n[n > t].sum() / n.sum():.3f}")
