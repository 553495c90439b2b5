#!/bin/bash
# Run ON the GPU box: ebvo_stereo_refine wall time for the layout threshold (EBVO_GN_ROWS_BELOW) and, rebuilt, GN_TAP_ROWS.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for rows in 1 2; do
  touch edge_based_visual_odometry_amd/csrc/refine_kernels.hip
  make -s -C edge_based_visual_odometry_amd/csrc EXTRA=-DGN_TAP_ROWS=$rows > /dev/null 2>&1 || { echo "build failed"; exit 1; }
  for thr in 0 16384 32768 65536; do
    echo "GN_TAP_ROWS=$rows EBVO_GN_ROWS_BELOW=$thr: $(EBVO_GN_ROWS_BELOW=$thr python3 tools/gpu_stereo_refine_time.py 2>&1 | head -1)"
  done
done
touch edge_based_visual_odometry_amd/csrc/refine_kernels.hip
make -s -C edge_based_visual_odometry_amd/csrc > /dev/null 2>&1
