#!/bin/bash
# Run ON the GPU box: A/B of two builds of the library in one call, interleaved (box-to-box drift is 2-3 %, minute-to-minute
# ~1 %): bench.py's resident loop with EBVO_LIB pointing at either build.   usage: tools/gpu_ab_libs.sh <out.txt> <base.so> [reps] [bench args]
set -u
OUT=$1; BASE=$2; REPS=${3:-3}; shift; shift; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
ARGS="--steps 300 --warmup 10 --no-cpu-baseline --no-transfer-legs --no-ingest --no-verify $*"
echo "# python3 bench.py $ARGS   (A = $BASE, B = the tree's library), interleaved x $REPS" > "$OUT"
for rep in $(seq 1 "$REPS"); do
  for which in A B; do
    if [ $which = A ]; then export EBVO_LIB=$ROOT/$BASE; else unset EBVO_LIB; fi
    python3 bench.py $ARGS 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which rep $rep  value %.1f  sustained_300 %s  short_warmup %s' % (d['value'], d.get('value_sustained_300'), d.get('value_short_warmup')))" >> "$OUT" || { echo "$which rep $rep FAILED" >> "$OUT"; exit 1; }
  done
done
cat "$OUT"
