#!/bin/bash
# Run ON the GPU box: builds tools/stagewise_profile.cpp against the in-tree library and prints the per-call split of the
# resident stage-wise sequence on the KITTI-size S2 pair.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
LIB=$ROOT/edge_based_visual_odometry_amd
g++ -std=c++17 -O2 -I include tools/stagewise_profile.cpp -o /tmp/stagewise_profile -L "$LIB" -lebvo_hip -Wl,-rpath,"$LIB" -Wl,-rpath,/opt/rocm/lib
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from edge_based_visual_odometry_amd import synth
l, r = synth.stereo_pair("s2", 376, 1241, scene=7, noise_base=0, disparity=12)
l.tofile('/tmp/l.raw'); r.tofile('/tmp/r.raw')
PY
/tmp/stagewise_profile /tmp/l.raw /tmp/r.raw 376 1241 ${1:-20}
