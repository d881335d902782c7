#!/bin/bash
# GPU box: the third structure's 256-channel form (8-row tiles, optional tall-image tiling) against the first structure
# (bitwise) and the second's 256-channel form (timing)
P=${P:-tools/probes/conv_x3_t448_probe}
OUT=${1:-gpurun_out/t448_c4_probe.txt}
mkdir -p $(dirname $OUT)
{
echo "== correctness =="
for args in "2 56 56 64 256 0 0" "3 28 28 128 512 0 0" "3 28 28 64 256 0 1" "2 56 56 64 256 0 1" "1 8 28 64 256 0 0" \
            "4 12 28 64 256 0 0" "5 20 56 64 256 0 0" "2 24 84 192 256 0 1" "7 28 28 64 512 0 1" "3 14 28 64 256 0 0" \
            "2 112 112 64 128 0 0 2" "2 112 112 64 64 0 1 1"; do
  echo "-- $args"
  timeout -k 10 120 $P $args || echo "FAILED rc=$? ($args)"
done
} > $OUT 2>&1
if grep -q FAILED $OUT; then echo "correctness failed"; grep -B6 FAILED $OUT | tail -60; exit 1; fi
{
echo "== timing, batch 256 =="
for args in "256 56 56 128 256 10 0" "256 56 56 256 256 8 0" "256 56 56 512 256 6 0" "256 28 28 256 512 8 0" \
            "256 28 28 512 512 8 0" "256 28 28 1024 512 6 0" "256 56 56 256 256 8 1" "256 28 28 512 512 8 1"; do
  echo "-- $args"
  timeout -k 10 300 $P $args || echo "FAILED rc=$? ($args)"
done
} >> $OUT 2>&1
grep -v "float64" $OUT | tail -70
