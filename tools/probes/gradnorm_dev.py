"""GPU box: gradient norms of the reference's batch-4 golden step (tests/golden/modelA_train_step_b4.npz) on the HIP
training path - deviation from the reference's fp32 run AND from its float64 run (modelA_train_step_b4_f64.npz).
usage: [UNET_TRAIN_FUSED_STATS=0|1] [UNET_TRAIN_X3=0|1] python tools/probes/gradnorm_dev.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from unet_lane_detection_amd import state as S
from unet_lane_detection_amd.trainer import UNetTrainer
g = np.load("tests/golden/modelA_train_step_b4.npz")
g64 = np.load("tests/golden/modelA_train_step_b4_f64.npz")
tr = UNetTrainer(S.seeded_state_dict(seed=0), device=0, lr=1e-4)
tr.forward_backward(torch.from_numpy(S.synthetic_frames(4, seed=3)), torch.from_numpy(S.synthetic_targets(4, seed=3)))
print("env FUSED_STATS=%s TRAIN_X3=%s: loss diff %.3e" % (os.environ.get("UNET_TRAIN_FUSED_STATS"), os.environ.get("UNET_TRAIN_X3"),
                                                           abs(float(tr.loss.item()) - float(g["loss"]))))
gd = tr.grad_dict()
devs = []
for k in g.files:
    if k.startswith("gradnorm/"):
        ref = float(g[k]); ref64 = float(g64["gradnorm64/" + k[9:]]); got = float(gd[k[9:]].double().norm().item())
        devs.append((abs(got - ref64) / max(ref64, 1e-6), abs(got - ref) / max(ref, 1e-6), abs(ref - ref64) / max(ref64, 1e-6), k[9:], got))
print("worst |ours - f64| %.3e, worst |ours - ref fp32| %.3e, worst |ref fp32 - f64| %.3e" %
      (max(d[0] for d in devs), max(d[1] for d in devs), max(d[2] for d in devs)))
devs.sort(reverse=True)
for d in devs[:5]: print("  vs f64 %.3e  vs ref32 %.3e  (ref32 vs f64 %.3e)  %s  %.6g" % d)
