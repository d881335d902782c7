#!/bin/bash
OUT=${1:-gpurun_out/r512_stamps.txt}
mkdir -p $(dirname $OUT)
shift
for P in "$@"; do
{
echo "#### $P"
timeout -k 10 300 tools/probes/$P 256 56 56 128 256 6 || echo "FAILED rc=$?"
timeout -k 10 300 tools/probes/$P 256 56 56 512 256 6 || echo "FAILED rc=$?"
timeout -k 10 300 tools/probes/$P 256 28 28 1024 512 6 || echo "FAILED rc=$?"
timeout -k 10 300 tools/probes/$P 256 14 14 1024 1024 6 || echo "FAILED rc=$?"
timeout -k 10 300 tools/probes/$P 256 112 112 128 128 6 2 || echo "FAILED rc=$?"
} >> $OUT 2>&1
done
grep -v "^N \|float64" $OUT
