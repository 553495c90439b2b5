#!/bin/bash
# Run ON the GPU box: durations of the 20 gn_iter launches of one ebvo_stereo_refine (KITTI S2 pair), in launch order.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/gn_launches
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o out -- python3 "$ROOT/tools/gpu_stereo_refine_time.py" > "$OUT/run.log" 2>&1
CSV=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)
python3 - "$CSV" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gn_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-44:]          # the last refine call: init kernels + 20 iterations x two layouts
for r in last:
    print("%-40s %8.1f us  grid %s" % (r["Kernel_Name"][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", "?")))
PY
rm -rf "$OUT/trace"
