"""Single-frame forward: per-kernel time (hipEvent pairs) and host-side wall time, per tier.  GPU box only.
usage: python tools/latency_breakdown.py [--tier f16x3|fp32] [--batch 1]"""
import argparse
import collections
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from unet_lane_detection_amd import state as S
from unet_lane_detection_amd.model import UNetHIP

ap = argparse.ArgumentParser()
ap.add_argument("--tier", default="f16x3")
ap.add_argument("--batch", type=int, default=1)
args = ap.parse_args()
m = UNetHIP(S.seeded_state_dict(seed=0), device=0)
frames = torch.from_numpy(S.synthetic_frames(args.batch, seed=1)).cuda()
for _ in range(5):
    m.run_u8(frames, return_probs=True, precision=args.tier)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    m.run_u8(frames, return_probs=True, precision=args.tier)
torch.cuda.synchronize()
print("device-resident forward, no profiling: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
m.profile(True)
m.run_u8(frames, return_probs=True, precision=args.tier)
torch.cuda.synchronize()
recs = m.profile_records()
m.profile(False)
tot = sum(r[1] for r in recs)
print("sum of kernel times: %.3f ms over %d launches" % (tot, len(recs)))
for i, (name, ms, fl, by) in enumerate(recs):
    print("%2d %-28s %8.1f us  %8.1f GF  %6.1f TF/s" % (i, name, ms * 1e3, fl / 1e9, fl / max(ms, 1e-9) / 1e9))
