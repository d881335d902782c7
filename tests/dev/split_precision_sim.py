"""Dev study (CPU, not a test): logit error of the split-operand tiers against the fp32 oracle on the
reference frame.  Every conv operand is replaced by hi + lo (bf16 or fp16, round to nearest) and the product
hi*hi + hi*lo + lo*hi is accumulated in fp32 (torch CPU conv)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as O          # noqa: E402
from unet_lane_detection_amd import state as S  # noqa: E402


def split(x, dt):
    hi = x.to(dt).float()
    lo = (x - hi).to(dt).float()
    return hi, lo


def make_ops(dt, terms=3, wino=False):
    def conv(x, w, **kw):
        xh, xl = split(x, dt)
        wh, wl = split(w, dt)
        y = F.conv2d(xh, wh, **kw) + F.conv2d(xh, wl, **kw) + F.conv2d(xl, wh, **kw)
        if terms == 4:
            y = y + F.conv2d(xl, wl, **kw)
        return y

    def convt(x, w, b):
        xh, xl = split(x, dt)
        wh, wl = split(w, dt)
        return (F.conv_transpose2d(xh, wh, None, stride=2) + F.conv_transpose2d(xh, wl, None, stride=2)
                + F.conv_transpose2d(xl, wh, None, stride=2)) + b[None, :, None, None]
    return conv, convt


def wino_conv(x, w, dt):
    """F(2x2,3x3) with split operands in the Winograd domain (transforms in fp32 / double for weights)."""
    N, C, H, W = x.shape
    K = w.shape[0]
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
    U = torch.einsum('ia,kcab,jb->kcij', G, w.double(), G).float()          # K,C,4,4
    xp = F.pad(x, (1, 1, 1, 1))
    patches = xp.unfold(2, 4, 2).unfold(3, 4, 2)                              # N,C,H/2,W/2,4,4
    V = torch.einsum('ia,nctuab,jb->nctuij', BT, patches, BT)                 # fp32 adds
    Vh, Vl = split(V, dt)
    Uh, Ul = split(U, dt)
    M = (torch.einsum('nctuij,kcij->nktuij', Vh, Uh) + torch.einsum('nctuij,kcij->nktuij', Vh, Ul)
         + torch.einsum('nctuij,kcij->nktuij', Vl, Uh))
    Y = torch.einsum('ai,nktuij,bj->nktuab', AT, M, AT)                       # N,K,H/2,W/2,2,2
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)


def run(tag, conv3, convt, sd, x, ref):
    O_conv3x3, O_up = O.conv3x3, O.upconv2x2
    O.conv3x3 = conv3
    O.upconv2x2 = convt
    try:
        with torch.no_grad():
            y = O.forward(sd, x).numpy()[0, 0]
    finally:
        O.conv3x3, O.upconv2x2 = O_conv3x3, O_up
    d = np.abs(y - ref)
    flips = int(((y > 0) != (ref > 0)).sum())
    print(f"{tag:34s} max|dlogit| {d.max():.3e}  rms {np.sqrt((d**2).mean()):.3e}  mask flips {flips}", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(8)
    g = np.load(os.path.join(ROOT, "tests/golden/modelA_frame_001410.npz"))
    frame = np.fromfile(os.path.join(ROOT, "tests/golden/frame_001410_rgb_u8.bin"), dtype=np.uint8).reshape(1, 224, 224, 3)
    sd = O.to_torch_state(S.seeded_state_dict(seed=0))
    x = O.normalize_u8_nhwc(frame)
    ref = g["logits"]
    with torch.no_grad():
        y0 = O.forward(sd, x).numpy()[0, 0]
    print("oracle vs golden", np.abs(y0 - ref).max())
    for dt, name in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        c, t = make_ops(dt)
        run(f"{name} x3 direct", lambda a, w: c(a, w, padding=1), t, sd, x, ref)
        if "--wino" in sys.argv:
            run(f"{name} x3 winograd", lambda a, w, dt=dt: (wino_conv(a, w, dt) if a.shape[1] >= 16 else c(a, w, padding=1)), t, sd, x, ref)
