#!/bin/bash
# usage: tools/isa_stats.sh <file.hip> <kernel-name-substring>
# Compiles for gfx950 with the product flags, prints register use and an instruction histogram.
set -e
SRC=$(realpath "$1"); PAT="$2"
OUT=$(mktemp -d /tmp/isa.XXXXXX)
( cd "$OUT" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -save-temps $EXTRA -c "$SRC" -o x.o 2>/dev/null )
S=$(ls "$OUT"/*gfx950.s)
SYM=$(grep -oE "^_Z[A-Za-z0-9_]*${PAT}[A-Za-z0-9_]*:" "$S" | head -1 | tr -d ':')
echo "kernel: $SYM   asm: $S"
awk -v sym="$SYM:" '$1==sym{f=1} f{print} f&&/s_endpgm/{exit}' "$S" > "$OUT/k.s"
echo "lines: $(wc -l < "$OUT/k.s")"
for k in v_mul_f64 v_add_f64 v_fma_f64 s_load_dword ds_read ds_write v_accvgpr scratch_ s_waitcnt s_cbranch global_load global_store; do
  printf "%-14s %s\n" $k $(grep -c "$k" "$OUT/k.s" || true)
done
grep -E "${SYM}\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size)" "$S" | sed 's/.*\.set //'
awk -v sym="$SYM" '$0 ~ sym && /\.size/{f=1} f&&/Occupancy/{print; exit}' "$S"
