#!/bin/bash
# GPU box: in-kernel cycle stamps of the third structure (diagnostic build, -DUNET_R512_STAMPS=1)
OUT=${1:-gpurun_out/t448_stamps.txt}
mkdir -p $(dirname $OUT)
shift
for P in "$@"; do
{
echo "#### $P"
for args in "256 224 224 64 64 6 0" "256 224 224 128 64 6 0" "256 112 112 128 128 6 0" "256 112 112 64 128 6 0"; do
  echo "-- $args"
  timeout -k 10 300 tools/probes/$P $args || echo "FAILED rc=$?"
done
} >> $OUT 2>&1
done
grep -v "^N \|float64\|differ" $OUT
