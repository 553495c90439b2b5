#!/bin/bash
# Run ON the GPU box: tools/boundary_bench.cpp with the library's stage-wise trace on (EBVO_TRACE_STAGEWISE), KITTI-size pair.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
LIB=$ROOT/edge_based_visual_odometry_amd
g++ -std=c++17 -O2 -I include tools/boundary_bench.cpp -o /tmp/boundary_bench -L "$LIB" -lebvo_hip -Wl,-rpath,"$LIB" -Wl,-rpath,/opt/rocm/lib
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from edge_based_visual_odometry_amd import synth
l, r = synth.stereo_pair("s2", 376, 1241, scene=7, noise_base=0, disparity=12)
l.tofile('/tmp/l.raw'); r.tofile('/tmp/r.raw')
PY
EBVO_TOED_MODE=hybrid EBVO_TRACE_STAGEWISE=1 /tmp/boundary_bench /tmp/l.raw /tmp/r.raw 376 1241 ${1:-6} 2>&1 | tail -${2:-20}
EBVO_TOED_MODE=hybrid /tmp/boundary_bench /tmp/l.raw /tmp/r.raw 376 1241 ${1:-6}
