"""Loss trajectory of the training step on one fixed synthetic batch, fp16 split-operand convolutions against the
exact-fp32 kernels (unet_set_train_x3): the two should track each other step for step.  GPU box only.
usage: python tools/train_trajectory.py [--steps 40] [--batch 16]"""
import argparse
import sys

import torch

sys.path.insert(0, ".")
from unet_lane_detection_amd import _lib, state as S
from unet_lane_detection_amd.trainer import UNetTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--lr", type=float, default=1e-3)
args = ap.parse_args()
lib = _lib.load(build_if_missing=False)
frames = torch.from_numpy(S.synthetic_frames(args.batch, seed=3))
targets = torch.from_numpy(S.synthetic_targets(args.batch, seed=3))
curves = {}
for mode in (1, 0):
    lib.unet_set_train_x3(mode)
    tr = UNetTrainer(S.seeded_state_dict(seed=0), device=0, lr=args.lr)
    losses = []
    for _ in range(args.steps):
        tr.forward_backward(frames, targets)
        losses.append(float(tr.loss.item()))
        tr.optimizer_step()
    curves[mode] = losses
    tr.release()
lib.unet_set_train_x3(1)
print("step  f16x3-convs   fp32-convs   rel.diff")
for i, (a, b) in enumerate(zip(curves[1], curves[0])):
    if i < 5 or i % 5 == 4:
        print(f"{i:4d}  {a:11.6f}  {b:11.6f}  {abs(a - b) / max(abs(b), 1e-12):9.2e}")
worst = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(curves[1], curves[0]))
print(f"largest relative difference over {args.steps} steps: {worst:.2e}; final losses {curves[1][-1]:.6f} / {curves[0][-1]:.6f}")
assert all(x == x and abs(x) < 1e6 for x in curves[1]), "non-finite loss"
