#!/bin/bash
# Run ON the GPU box: kernel trace + SQ counters of the resident chain (ebvo_stereo_run + ebvo_stereo_finalize).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
OUT=$ROOT/gpurun_out/prof_${TAG}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
    local name=$1
    shift
    echo "== $name: rocprofv3 $* -- python3 tools/gpu_chain_time.py" | tee -a "$OUT/log.txt"
    rocprofv3 "$@" -d "$OUT/$name" -o out -- python3 "$ROOT/tools/gpu_chain_time.py" >> "$OUT/log.txt" 2>&1 || echo "   FAILED: $name"
}
pass trace_chain --kernel-trace --stats
pass pmc_sq1_chain --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass pmc_sq2_chain --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VALU SQ_INSTS_SMEM
pass pmc_sq3_chain --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
pass pmc_grbm_chain --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY
