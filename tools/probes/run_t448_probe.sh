#!/bin/bash
# GPU box: correctness cases, then layer timings at batch 256, of tools/probes/conv_x3_t448_probe (built in the container)
P=${P:-tools/probes/conv_x3_t448_probe}
OUT=${1:-gpurun_out/t448_probe.txt}
mkdir -p $(dirname $OUT)
{
echo "== correctness =="
for args in "2 112 112 64 64 0 0" "2 112 112 64 64 0 1" "2 112 112 64 64 0 2" "3 56 56 128 64 0 0" "2 40 84 64 64 0 0" \
            "2 40 28 64 64 0 1" "2 112 112 64 128 0 0" "2 112 112 128 128 0 1" "3 24 56 64 256 0 0" "5 16 28 192 128 0 1" \
            "2 64 64 64 64 0 0" "2 48 96 64 64 0 1" "2 32 32 64 64 0 2" "2 64 64 64 128 0 0"; do
  echo "-- $args"
  timeout -k 10 120 $P $args || echo "FAILED rc=$? ($args)"
done
} > $OUT 2>&1
if grep -q FAILED $OUT; then echo "correctness failed"; grep -B6 FAILED $OUT | tail -60; exit 1; fi
{
echo "== timing, batch 256 =="
for args in "256 224 224 64 64 8 1" "256 112 112 64 128 10 0" "256 112 112 128 128 8 1" "256 112 112 256 128 8 0" \
            "256 112 112 128 128 8 0" "256 224 224 128 64 6 0" "256 224 224 64 64 8 2" "256 224 224 64 64 8 0"; do
  echo "-- $args"
  timeout -k 10 300 $P $args || echo "FAILED rc=$? ($args)"
done
} >> $OUT 2>&1
grep -v "float64" $OUT | tail -90
