#!/bin/bash
# Run ON the GPU box: the small measurements DESIGN.md cites besides the rocprofv3 summaries.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=gpurun_out/evidence
mkdir -p $OUT
( echo "# tools/ubench/bank_conflict"; timeout -k 5 60 tools/ubench/bank_conflict; echo; echo "# tools/ubench/issue_mix"; timeout -k 5 60 tools/ubench/issue_mix ) > $OUT/r02_ubench_valu_issue.txt 2>&1
python3 tools/gpu_lanes_sweep.py 4:3 6:3 8:3 12:3 5:4 6:4 8:4 12:4 8:1 8:2 8:5 8:6 > $OUT/r02_lanes_sweep.txt 2>&1
bash tools/gpu_gn_launches.sh > $OUT/r02_gn_launches.txt 2>&1
python3 tools/gpu_stereo_refine_time.py >> $OUT/r02_gn_launches.txt 2>&1
# wave placement of the exact kernels: needs the traced build (on this box's copy only)
make -C edge_based_visual_odometry_amd/csrc clean > /dev/null 2>&1
make -C edge_based_visual_odometry_amd/csrc EXTRA=-DEBVO_WAVE_TRACE > /dev/null 2>&1
python3 tools/gpu_wave_trace.py > $OUT/r02_wave_trace.txt 2>&1
ls -la $OUT
