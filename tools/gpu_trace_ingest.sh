#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel trace of the streamed-ingest loops (tools/gpu_ingest_diag.py): the pull kernel at the head
# of a pair's chain and the pack kernel at its end next to the other kernels.  usage: tools/gpu_trace_ingest.sh <tag> [pack|push]
set -u
TAG=${1:-r04_ingest}
FORM=${2:-pack}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export EBVO_PROFILES_DST=${EBVO_PROFILES_DST:-$ROOT/gpurun_out/profiles_${TAG}}
mkdir -p "$EBVO_PROFILES_DST" gpurun_out/prof_${TAG}
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/prof_${TAG}/trace_hybrid" -o out -- \
    python3 "$ROOT/tools/gpu_ingest_diag.py" "$FORM" > "$EBVO_PROFILES_DST/${TAG}_loops.txt" 2>&1 ) || { echo "trace FAILED"; tail -5 "$EBVO_PROFILES_DST/${TAG}_loops.txt"; exit 1; }
echo "== trace_hybrid: rocprofv3 --kernel-trace --stats -- python3 tools/gpu_ingest_diag.py $FORM" > gpurun_out/prof_${TAG}/log.txt
python3 tools/rocprof_summary.py ${TAG} hybrid | head -40
rm -rf gpurun_out/prof_${TAG}
