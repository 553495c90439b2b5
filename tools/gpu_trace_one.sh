#!/bin/bash
# Run ON the GPU box: one rocprofv3 kernel trace of bench.py with the given arguments, summarised into
# gpurun_out/profiles_<tag>/<tag>_kernel_stats_hybrid.txt.   usage: tools/gpu_trace_one.sh <tag> <bench.py arguments ...>
# --no-transfer-legs --no-ingest --no-cpu-baseline are always appended: the boundary leg would build and start a GPU child process from
# inside the profiled program (which the profiler's preload has already put on the GPU), as tools/gpu_profile.sh explains.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export EBVO_PROFILES_DST=$ROOT/gpurun_out/profiles_${TAG}
mkdir -p "$EBVO_PROFILES_DST" gpurun_out/prof_${TAG}
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/prof_${TAG}/trace_hybrid" -o out -- \
    python3 "$ROOT/bench.py" "$@" --no-transfer-legs --no-ingest --no-cpu-baseline > "$EBVO_PROFILES_DST/${TAG}_bench_traced.json" 2> "$EBVO_PROFILES_DST/${TAG}_bench_traced.err" ) || { echo "trace FAILED"; tail -5 "$EBVO_PROFILES_DST/${TAG}_bench_traced.err"; exit 1; }
echo "== trace_hybrid: rocprofv3 --kernel-trace --stats -- python3 bench.py $* --no-transfer-legs --no-ingest --no-cpu-baseline" > gpurun_out/prof_${TAG}/log.txt
python3 tools/rocprof_summary.py ${TAG} hybrid | head -70
rm -rf gpurun_out/prof_${TAG}
