#!/bin/bash
# Run ON the GPU box: the two HBM counter passes (FETCH_SIZE, WRITE_SIZE) + a kernel trace of the headline pair, summarised.
# usage: tools/gpu_pmc_hbm_quick.sh <tag>
set -u
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-verify --no-transfer-legs --no-ingest --toed-mode hybrid --streams 1"
for pass in "trace_hybrid 40 --kernel-trace --stats" "pmc_fetch_hybrid 8 --kernel-trace --pmc FETCH_SIZE" "pmc_write_hybrid 8 --kernel-trace --pmc WRITE_SIZE"; do
    set -- $pass
    name=$1; steps=$2; shift 2
    echo "== $name: rocprofv3 $* -- python3 bench.py --steps $steps --warmup 2 $COMMON" >> "$OUT/log.txt"
    rocprofv3 "$@" -d "$OUT/$name" -o out -- python3 "$ROOT/bench.py" --steps "$steps" --warmup 2 $COMMON >> "$OUT/log.txt" 2>&1 || echo "FAILED $name"
done
cd "$ROOT"
EBVO_PROFILES_DST=$ROOT/gpurun_out/profiles_$TAG python3 tools/rocprof_summary.py $TAG hybrid x > /dev/null
head -9 gpurun_out/profiles_$TAG/${TAG}_kernel_stats_hybridx.txt
grep -E "toed_exact_centre|toed_exact_decide|toed_exact_mags|sum over" gpurun_out/profiles_$TAG/${TAG}_pmc_hbm_hybridx.txt
rm -rf "$OUT"
