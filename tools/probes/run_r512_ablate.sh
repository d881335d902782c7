#!/bin/bash
OUT=${1:-gpurun_out/r512_ablate.txt}
mkdir -p $(dirname $OUT)
shift
for P in "$@"; do
{
echo "#### $P"
timeout -k 10 300 tools/probes/$P 256 56 56 512 256 5
timeout -k 10 300 tools/probes/$P 256 28 28 1024 512 5
timeout -k 10 300 tools/probes/$P 256 112 112 128 128 5 2
} >> $OUT 2>&1
done
grep -v "^N \|float64\|halfs" $OUT
