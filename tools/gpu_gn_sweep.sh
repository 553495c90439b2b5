#!/bin/bash
# Run ON the GPU box: the layout threshold of the stereo refinement (EBVO_GN_ROWS_BELOW) at KITTI size (one call of
# ebvo_stereo_refine) and on the EuRoC sequence bench (small refinements every frame).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for thr in 0 8192 16384 32768 49152 65536; do
  echo "EBVO_GN_ROWS_BELOW=$thr kitti: $(EBVO_GN_ROWS_BELOW=$thr python3 tools/gpu_stereo_refine_time.py 2>&1 | head -1)"
  echo "EBVO_GN_ROWS_BELOW=$thr euroc: $(EBVO_GN_ROWS_BELOW=$thr python3 bench.py --workload euroc --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.load(sys.stdin); print(round(d["value"],1), d["unit"], {k: round(v,1) for k,v in d.items() if k.endswith("frames_per_s")})')"
done
