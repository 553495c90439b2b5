#!/bin/bash
# Run ON the GPU box: kernel trace of the sequence loop with EVERY stage of the temporal chain (tools/gpu_pipeline_hosttime.py
# euroc 1 1,2,3), summarised into gpurun_out/profiles_<tag>/<tag>_kernel_stats_hybrid.txt.   usage: tools/gpu_trace_full_chain.sh <tag>
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export EBVO_PROFILES_DST=$ROOT/gpurun_out/profiles_${TAG}
mkdir -p "$EBVO_PROFILES_DST" gpurun_out/prof_${TAG}
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/prof_${TAG}/trace_hybrid" -o out -- \
    python3 "$ROOT/tools/gpu_pipeline_hosttime.py" euroc 1 1,2,3 > "$EBVO_PROFILES_DST/${TAG}_loops.txt" 2>&1 ) || { echo "trace FAILED"; tail -5 "$EBVO_PROFILES_DST/${TAG}_loops.txt"; exit 1; }
echo "== trace_hybrid: rocprofv3 --kernel-trace --stats -- python3 tools/gpu_pipeline_hosttime.py euroc 1 1,2,3" > gpurun_out/prof_${TAG}/log.txt
python3 tools/rocprof_summary.py ${TAG} hybrid | head -60
grep "frames/s" "$EBVO_PROFILES_DST/${TAG}_loops.txt"
rm -rf gpurun_out/prof_${TAG}
