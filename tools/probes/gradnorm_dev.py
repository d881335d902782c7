import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from unet_lane_detection_amd import state as S
from unet_lane_detection_amd.trainer import UNetTrainer
g = np.load("tests/golden/modelA_train_step_b4.npz")
tr = UNetTrainer(S.seeded_state_dict(seed=0), device=0, lr=1e-4)
tr.forward_backward(torch.from_numpy(S.synthetic_frames(4, seed=3)), torch.from_numpy(S.synthetic_targets(4, seed=3)))
print("loss diff", abs(float(tr.loss.item()) - float(g["loss"])))
gd = tr.grad_dict()
devs = []
for k in g.files:
    if k.startswith("gradnorm/"):
        ref = float(g[k]); got = float(gd[k[9:]].double().norm().item())
        devs.append((abs(got - ref) / max(ref, 1e-6), k[9:], got, ref))
devs.sort(reverse=True)
for d in devs[:6]: print("%.3e %s %.6g %.6g" % d)
