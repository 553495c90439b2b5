#!/bin/bash
# Run ON the GPU box: tools/gpu_profile.sh + on-box summary; only the text summaries come back (gpurun_out/profiles_<tag>/).
TAG=${1:-wip}
MODE=${2:-hybrid}
[ $# -gt 0 ] && shift; [ $# -gt 0 ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export EBVO_PROFILES_DST=$ROOT/gpurun_out/profiles_${TAG}
mkdir -p "$EBVO_PROFILES_DST"
bash tools/gpu_profile.sh ${TAG} ${MODE} "$@" > "$EBVO_PROFILES_DST/run.log" 2>&1; cp gpurun_out/prof_${TAG}/log.txt "$EBVO_PROFILES_DST/commands.log"
python3 tools/rocprof_summary.py ${TAG} ${MODE} _wip >> "$EBVO_PROFILES_DST/summary_errors.log" 2>&1
rm -rf gpurun_out/prof_${TAG}
head -12 "$EBVO_PROFILES_DST/${TAG}_kernel_stats_${MODE}_wip.txt"
