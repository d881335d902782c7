#!/bin/bash
# GPU box: socket power and shader clock (rocm-smi, read only) while the batch-64 training step runs back to back, with the
# weight gradients in line (UNET_TRAIN_SIDE=0) and on the side stream (1).  usage: tools/train_power_trace.sh <out dir>
OUT=${1:-gpurun_out/train_power}
mkdir -p $OUT
for MODE in 0 1; do
  UNET_TRAIN_SIDE=$MODE python3 - > $OUT/steps_mode$MODE.txt 2>&1 <<'PY' &
import sys, time
import torch
sys.path.insert(0, ".")
from unet_lane_detection_amd import state as S
from unet_lane_detection_amd.trainer import UNetTrainer
tr = UNetTrainer(S.seeded_state_dict(seed=0), device=0, lr=1e-4)
x = torch.from_numpy(S.synthetic_frames(64, seed=100)).cuda()
t = torch.from_numpy(S.synthetic_targets(64, seed=100)).cuda()
for _ in range(5):
    tr.step(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 14.0:
    for _ in range(20):
        tr.step(x, t)
    torch.cuda.synchronize()
    n += 20
print("steps %d, %.3f ms per step" % (n, (time.perf_counter() - t0) / n * 1e3))
PY
  BP=$!
  sleep 12   # import torch, workspace, warm-up
  for i in $(seq 1 28); do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" >> $OUT/trace_mode$MODE.txt
    echo "--" >> $OUT/trace_mode$MODE.txt
    sleep 0.25
    kill -0 $BP 2>/dev/null || break
  done
  wait $BP
  cat $OUT/steps_mode$MODE.txt | tail -1
done
