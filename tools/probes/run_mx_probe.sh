#!/bin/bash
# GPU box: correctness cases, then layer timings, of tools/probes/conv_mx_r512_probe (built in the container)
P=tools/probes/conv_mx_r512_probe
OUT=${1:-gpurun_out/mx_probe.txt}
mkdir -p $(dirname $OUT)
{
echo "== correctness =="
timeout -k 10 120 $P 2 56 56 64 256 || echo "FAILED rc=$?"
timeout -k 10 120 $P 3 28 28 128 512 || echo "FAILED rc=$?"
timeout -k 10 120 $P 5 14 14 64 256 || echo "FAILED rc=$?"
timeout -k 10 120 $P 1 112 112 64 256 || echo "FAILED rc=$?"
} > $OUT 2>&1
if grep -q "FAILED\|HIP error" $OUT; then echo "correctness failed"; tail -30 $OUT; exit 1; fi
{
echo "== timing, batch 256 =="
timeout -k 10 300 $P 256 56 56 128 256 12 || echo "FAILED rc=$?"
timeout -k 10 300 $P 256 56 56 256 256 12 || echo "FAILED rc=$?"
timeout -k 10 300 $P 256 56 56 512 256 8 || echo "FAILED rc=$?"
timeout -k 10 300 $P 256 28 28 512 512 12 || echo "FAILED rc=$?"
timeout -k 10 300 $P 256 28 28 1024 512 8 || echo "FAILED rc=$?"
timeout -k 10 300 $P 256 14 14 1024 1024 12 || echo "FAILED rc=$?"
} >> $OUT 2>&1
tail -60 $OUT
