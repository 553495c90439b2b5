#!/bin/bash
# Run ON the GPU box: samples the shader clock and the power while the resident pair loop runs (25,000 pairs, ~8 s).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
python3 bench.py --steps 25000 --warmup 5 --no-cpu-baseline --no-transfer-legs --no-ingest --no-verify > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
BP=$!
sleep 2.5   # import + warm-up
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use" | tr '\n' ';'; echo
  sleep 0.15
  kill -0 $BP 2>/dev/null || break
done
wait $BP
python3 -c "
import json; d=json.loads(open('gpurun_out/clock_bench.json').read().strip().splitlines()[-1]); print('value', round(d['value'],1), 'ms_per_step', round(d['ms_per_step'],4))"
echo "idle:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ';'; echo
