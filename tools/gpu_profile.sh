#!/bin/bash
# Run ON the GPU box (through gpurun): rocprofv3 kernel trace + PMC passes of bench.py, one pair in flight.
#   usage: tools/gpu_profile.sh <tag> [toed-mode] [extra bench.py arguments ...]
# Writes rocpd databases under gpurun_out/prof_<tag>/ ; tools/rocprof_summary.py turns them into profiles/<tag>_*.txt.
# Counters are collected in their own passes (--kernel-trace + --pmc only): FETCH_SIZE and WRITE_SIZE separately (TCC
# slots), SQ counters at most eight per pass.  The program itself follows `--` (python3 directly: no env / shell
# wrapper, rocprofv3's preloaded library initialises the GPU before the program starts).
set -u
TAG=${1:-r02}
MODE=${2:-hybrid}
[ $# -gt 0 ] && shift; [ $# -gt 0 ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# --no-transfer-legs --no-ingest: the trace covers the resident loop only (the boundary leg would start g++ and a GPU child under the
# profiler's preload)
COMMON="--no-cpu-baseline --no-verify --no-transfer-legs --no-ingest --toed-mode $MODE --streams 1 $*"
FAILED=0
pass() { # name, bench steps, rocprofv3 options ...
    local name=$1 steps=$2
    shift 2
    echo "== $name: rocprofv3 $* -- python3 bench.py --steps $steps --warmup 2 $COMMON" | tee -a "$OUT/log.txt"
    rocprofv3 "$@" -d "$OUT/$name" -o out -- python3 "$ROOT/bench.py" --steps "$steps" --warmup 2 $COMMON >> "$OUT/log.txt" 2>&1 ||
        { echo "   FAILED: $name" | tee -a "$OUT/log.txt"; FAILED=1; }
}
pass trace_$MODE 40 --kernel-trace --stats
pass pmc_fetch_$MODE 8 --kernel-trace --pmc FETCH_SIZE
pass pmc_write_$MODE 8 --kernel-trace --pmc WRITE_SIZE
pass pmc_sq1_$MODE 8 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass pmc_sq2_$MODE 8 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VALU SQ_INSTS_SMEM
pass pmc_sq3_$MODE 8 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
pass pmc_grbm_$MODE 8 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY
ls "$OUT"
if [ $FAILED -ne 0 ]; then echo "gpu_profile.sh: at least one pass failed (see $OUT/log.txt)"; tail -30 "$OUT/log.txt"; exit 1; fi
