#!/bin/bash
# Run ON the GPU box: everything profiles/ holds for a round.  usage: tools/gpu_round_profiles.sh <tag>
# The rocpd databases are summarised on the box and deleted (gpurun copies back at most 64 MiB): what returns is
# gpurun_out/profiles_<tag>/*.txt|json, to be copied into profiles/.
TAG=${1:-r04}
PART=${2:-all}   # "prof": the rocprofv3 passes; "bench": the bench lines and timing tools; "all": both (may exceed one gpurun call)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export EBVO_PROFILES_DST=$ROOT/gpurun_out/profiles_${TAG}
mkdir -p "$EBVO_PROFILES_DST"
sum() { python3 tools/rocprof_summary.py "$@" > /dev/null 2>> "$EBVO_PROFILES_DST/summary_errors.log"; }
if [ "$PART" != "bench" ]; then
bash tools/gpu_profile.sh ${TAG} hybrid > /dev/null 2>&1; sum ${TAG} hybrid; cp gpurun_out/prof_${TAG}/log.txt "$EBVO_PROFILES_DST/${TAG}_commands_hybrid.txt"; rm -rf gpurun_out/prof_${TAG}
bash tools/gpu_profile.sh ${TAG}s strict > /dev/null 2>&1; sum ${TAG}s strict; rm -rf gpurun_out/prof_${TAG}s
bash tools/gpu_profile.sh ${TAG}_eth3d hybrid --workload eth3d > /dev/null 2>&1; sum ${TAG}_eth3d hybrid; rm -rf gpurun_out/prof_${TAG}_eth3d
bash tools/gpu_profile_chain.sh ${TAG}_chain > /dev/null 2>&1; sum ${TAG}_chain chain; rm -rf gpurun_out/prof_${TAG}_chain
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/prof_${TAG}_euroc/trace_hybrid" -o out -- \
    python3 "$ROOT/bench.py" --workload euroc --steps 32 --warmup 2 --no-verify > "$EBVO_PROFILES_DST/${TAG}_bench_euroc_traced.json" 2> /dev/null )
echo "== trace_hybrid: rocprofv3 --kernel-trace --stats -- python3 bench.py --workload euroc --steps 32 --warmup 2 --no-verify" > gpurun_out/prof_${TAG}_euroc/log.txt
sum ${TAG}_euroc hybrid; rm -rf gpurun_out/prof_${TAG}_euroc
bash tools/gpu_timeline.sh ${TAG} > /dev/null 2>&1; cp gpurun_out/timeline_${TAG}/summary.txt "$EBVO_PROFILES_DST/${TAG}_timeline_slots.txt"; rm -rf gpurun_out/timeline_${TAG}
fi
if [ "$PART" = "prof" ]; then ls -la "$EBVO_PROFILES_DST"; exit 0; fi
# bench lines of the round (untraced)
python3 bench.py --no-cpu-baseline --no-transfer-legs > /dev/null 2>&1   # the first run on a fresh box reads low (clocks)
python3 bench.py > "$EBVO_PROFILES_DST/${TAG}_bench_kitti.json" 2> /dev/null
python3 bench.py --steps 20 --warmup 5 > "$EBVO_PROFILES_DST/${TAG}_bench_kitti_steps20.json" 2> /dev/null   # the driver's command line
python3 bench.py --gpus 2 --steps 100 --no-cpu-baseline --no-transfer-legs > "$EBVO_PROFILES_DST/${TAG}_bench_kitti_2ranks_1gpu.json" 2> /dev/null
python3 bench.py --streams 1 --no-cpu-baseline --no-transfer-legs > "$EBVO_PROFILES_DST/${TAG}_bench_kitti_1slot.json" 2> /dev/null
python3 bench.py --toed-mode strict --no-cpu-baseline --no-transfer-legs > "$EBVO_PROFILES_DST/${TAG}_bench_kitti_strict.json" 2> /dev/null
python3 bench.py --workload eth3d --no-cpu-baseline > "$EBVO_PROFILES_DST/${TAG}_bench_eth3d.json" 2> /dev/null
python3 bench.py --workload euroc --steps 64 --warmup 4 > "$EBVO_PROFILES_DST/${TAG}_bench_euroc.json" 2> /dev/null
python3 tools/gpu_gn_ab.py > "$EBVO_PROFILES_DST/${TAG}_gn_layouts.txt" 2>&1
python3 tools/gpu_pipeline_hosttime.py euroc > "$EBVO_PROFILES_DST/${TAG}_pipeline_hosttime.txt" 2>&1
bash tools/gpu_stagewise_profile.sh 20 > "$EBVO_PROFILES_DST/${TAG}_stagewise_calls.txt" 2>&1
bash tools/gpu_boundary_trace.sh 20 3 >> "$EBVO_PROFILES_DST/${TAG}_stagewise_calls.txt" 2>&1
python3 tools/gpu_chain_time.py > "$EBVO_PROFILES_DST/${TAG}_chain_time.txt" 2>&1
python3 tools/gpu_streams_sweep.py > "$EBVO_PROFILES_DST/${TAG}_slots_sweep.txt" 2>&1
python3 tools/gpu_ab_round4.py > "$EBVO_PROFILES_DST/${TAG}_ab_switches.txt" 2>&1
python3 tools/gpu_prefix_chain.py > "$EBVO_PROFILES_DST/${TAG}_prefix_chain.txt" 2>&1      # what every stage costs with six pairs in flight
python3 tools/gpu_marginal_cost.py > "$EBVO_PROFILES_DST/${TAG}_marginal_cost.txt" 2>&1
python3 tools/gpu_ingest_diag.py pack > "$EBVO_PROFILES_DST/${TAG}_ingest_forms.txt" 2>&1
python3 tools/gpu_ingest_diag.py push >> "$EBVO_PROFILES_DST/${TAG}_ingest_forms.txt" 2>&1
ls -la "$EBVO_PROFILES_DST"
du -sh gpurun_out
