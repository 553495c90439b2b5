#!/bin/bash
# Run ON the GPU box: the row layout of the stereo refinement compiled for 2 / 3 / 4 waves per SIMD, EuRoC sequence bench.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for wv in 2 3 4; do
  touch edge_based_visual_odometry_amd/csrc/refine_kernels.hip
  make -s -C edge_based_visual_odometry_amd/csrc EXTRA=-DGN_ROWS_WAVES=$wv > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  echo "GN_ROWS_WAVES=$wv euroc: $(python3 bench.py --workload euroc --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.load(sys.stdin); print(round(d["value"],1), d["unit"], round(d["full_temporal_chain_frames_per_s"],1), {k: round(v["ms_per_step"],3) for k,v in d.get("kernels",{}).items() if "gn" in k})')"
done
touch edge_based_visual_odometry_amd/csrc/refine_kernels.hip
make -s -C edge_based_visual_odometry_amd/csrc > /dev/null 2>&1
