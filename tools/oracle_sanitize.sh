#!/bin/bash
# The oracle (test infrastructure) under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU: the whole
# `-m "not gpu"` suite plus the oracle-side stereo chains (tools/oracle_sanitize_chain.py).  Rebuilds the normal
# library afterwards.  Prints the number of sanitizer reports (0 expected).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/oracle" || exit 1
gcc -O1 -g -std=gnu99 -fopenmp -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o libebvo_oracle.so ebvo_oracle.c -lm || exit 1
cd "$ROOT"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 OMP_NUM_THREADS=4
python -m pytest tests -q -s -m "not gpu" > /tmp/oracle_sanitize.log 2>&1
python tools/oracle_sanitize_chain.py >> /tmp/oracle_sanitize.log 2>&1
unset LD_PRELOAD
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' /tmp/oracle_sanitize.log)"
tail -3 /tmp/oracle_sanitize.log
make -s -C oracle clean && make -s -C oracle
