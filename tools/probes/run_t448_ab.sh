#!/bin/bash
# GPU box: A/B of two builds of tools/probes/conv_x3_t448_probe (stamps builds), same shapes, interleaved per shape
OUT=${1:-gpurun_out/t448_ab.txt}
A=$2
B=$3
mkdir -p $(dirname $OUT)
{
for args in "3 40 56 64 64 0 0" "2 48 56 64 128 0 1" "3 28 28 64 256 0 1" "2 22 28 64 64 0 0"; do
  echo "-- correctness $args"
  timeout -k 10 120 tools/probes/$A $args || echo "FAILED rc=$? ($A $args)"
  timeout -k 10 120 tools/probes/$B $args || echo "FAILED rc=$? ($B $args)"
done
} > $OUT 2>&1
if grep -q FAILED $OUT; then echo "correctness failed"; grep -B6 FAILED $OUT | tail -40; exit 1; fi
{
for args in "256 224 224 64 64 6 0" "256 224 224 128 64 6 0" "256 112 112 128 128 6 0" "256 112 112 64 128 6 0" "256 56 56 128 256 6 0" "256 28 28 512 512 6 0"; do
  for P in $A $B $A $B; do
    echo "-- $P $args"
    timeout -k 10 300 tools/probes/$P $args | grep "stamps\|third :" || echo "FAILED rc=$?"
  done
done
} >> $OUT 2>&1
grep -v "^N \|float64\|differ" $OUT | tail -80
