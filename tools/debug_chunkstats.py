"""How many 16-edge chunks of the right image meet one left edge's search region (sizing of the candidate walk)."""
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
L, R = out["left"], out["right"]
lx, ly = L["x"], L["y"]
rx, ry = R["x"], R["y"]
print("nL", len(L), "nR", len(R))
print("right y monotone violations", int((np.diff(ry) < 0).sum()), "max back-step", float(-np.diff(ry).min()))
for CH in (8, 16, 32):
    nch = (len(R) + CH - 1) // CH
    pad = nch * CH - len(R)
    X = np.concatenate([rx, np.full(pad, rx[-1])]).reshape(nch, CH)
    Y = np.concatenate([ry, np.full(pad, ry[-1])]).reshape(nch, CH)
    bx0, bx1, by0, by1 = X.min(1), X.max(1), Y.min(1), Y.max(1)
    print(f"chunk {CH}: mean box width {np.mean(bx1 - bx0):.1f} height {np.mean(by1 - by0):.2f}")
    D, band = 25.0, 0.5
    tot = 0
    sample = np.arange(0, len(L), 7)
    per = []
    for i in sample:
        m = (bx1 >= lx[i] - D) & (bx0 <= lx[i] + D) & (by1 >= ly[i] - band) & (by0 <= ly[i] + band)
        per.append(int(m.sum()))
    per = np.array(per)
    print(f"   chunks per left edge: mean {per.mean():.2f} p50 {np.median(per):.0f} p90 {np.percentile(per, 90):.0f} max {per.max()}"
          f"  -> pair tests per edge {per.mean() * CH:.0f}")
    # per wave: max over 64 consecutive sampled... use true consecutive lanes
    sel = np.arange(0, min(len(L), 64 * 200))
    pw = []
    for i in sel:
        m = (bx1 >= lx[i] - D) & (bx0 <= lx[i] + D) & (by1 >= ly[i] - band) & (by0 <= ly[i] + band)
        pw.append(int(m.sum()))
    pw = np.array(pw).reshape(-1, 64)
    print(f"   per-wave max chunks: mean {pw.max(1).mean():.2f}")
