#!/bin/bash
# GPU box: the timing-only builds on one layer (wrong results by construction: only the timing lines matter)
OUT=${1:-gpurun_out/mx_ablate.txt}
mkdir -p $(dirname $OUT)
for m in 0 1 2 4 8 16 7 24; do
  echo "== ablate $m ==" >> $OUT
  timeout -k 10 200 tools/probes/conv_mx_r512_probe_a$m 256 56 56 512 256 8 2>&1 | grep "stamps\|median" >> $OUT
done
cat $OUT
