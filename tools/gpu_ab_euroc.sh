#!/bin/bash
# Run ON the GPU box: A/B of two builds of the library on the sequence workload and on the KITTI-size chain, interleaved.
# usage: tools/gpu_ab_euroc.sh <out.txt> <base.so> [reps]
OUT=$1; BASE=$2; REPS=${3:-2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
echo "# python3 bench.py --workload euroc --no-cpu-baseline --no-verify and tools/gpu_chain_time.py   (A = $BASE, B = the tree's library), interleaved x $REPS" > "$OUT"
for rep in $(seq 1 "$REPS"); do
  for which in A B; do
    if [ $which = A ]; then export EBVO_LIB=$ROOT/$BASE; else unset EBVO_LIB; fi
    python3 bench.py --workload euroc --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which rep $rep  euroc value %.1f  full chain %.1f  one at a time %.1f  gn_refine %.3f ms  sift %.3f ms' % (d['value'], d['full_temporal_chain_frames_per_s'], d['one_frame_at_a_time_frames_per_s'], d['kernels']['gn_refine']['ms_per_step'], d['kernels']['sift']['ms_per_step']))" >> "$OUT"
    python3 tools/gpu_chain_time.py 2>/dev/null | grep "finalize" | sed "s/^/$which rep $rep  kitti chain: /" | cut -c1-120 >> "$OUT"
  done
done
cat "$OUT"
