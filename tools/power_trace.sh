#!/bin/bash
# GPU box: socket power and clocks (rocm-smi, read only) sampled while bench.py's inference step runs back to back.
# usage: tools/power_trace.sh <out dir> [bench args...]
OUT=${1:-gpurun_out/power}; shift
mkdir -p $OUT
rocm-smi --showpower --showclocks --showmaxpower > $OUT/idle.txt 2>&1
python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline --latency-iters 0 --no-check --other-tier-steps 0 --q8-steps 0 --train-steps 0 --bf16-steps 0 --int8-steps 0 --large-steps 0 "$@" > $OUT/bench.json 2> $OUT/bench.err &
BP=$!
sleep 25   # import torch + build of the plans + warm-up
for i in $(seq 1 40); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk" >> $OUT/trace.txt
  echo "--" >> $OUT/trace.txt
  sleep 0.25
  kill -0 $BP 2>/dev/null || break
done
wait $BP
echo "bench rc=$?" >> $OUT/trace.txt
tail -c 600 $OUT/bench.json
