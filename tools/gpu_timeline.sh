#!/bin/bash
# Run ON the GPU box: kernel traces of the resident loop with 3, 4 and 6 pairs in flight + their overlap statistics.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/timeline_${1:-r02}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for S in 3 4 6; do
    rocprofv3 --kernel-trace -d "$OUT/s$S" -o out -- python3 "$ROOT/bench.py" --steps 120 --warmup 3 --streams $S --no-verify \
        --no-cpu-baseline --no-transfer-legs --no-ingest > "$OUT/s$S.json" 2> "$OUT/s$S.err"
    DB=$(find "$OUT/s$S" -name '*_results.db' | head -1)
    echo "== $S pairs in flight: $(python3 -c "import json;print(round(json.load(open('$OUT/s$S.json'))['value']))") pairs/s (traced)" | tee -a "$OUT/summary.txt"
    python3 "$ROOT/tools/rocprof_timeline.py" "$DB" 0.25 0.8 | tee -a "$OUT/summary.txt"
done
