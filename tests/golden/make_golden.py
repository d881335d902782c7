#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container, where /root/reference is mounted.  The
reference's float network exists only as a code block in its README
(README.md:1418-1481); this script extracts that block as
text at run time, executes it against torch-CPU and records inputs/outputs.
No reference source is written into this repository: the committed artefacts
are data (inputs, expected outputs, scalar summaries).

Usage:  python tests/golden/make_golden.py [--reference /root/reference]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from unet_lane_detection_amd.state import (  # noqa: E402
    DEFAULT_FEATURES, seeded_state_dict, synthetic_frames, synthetic_targets)

README_UNET_LINES = (1418, 1481)  # `import torch` .. `return self.output(x)`


def load_reference_unet(ref_root):
    with open(os.path.join(ref_root, "README.md"), encoding="utf-8") as f:
        lines = f.read().split("\n")
    lo, hi = README_UNET_LINES
    src = "\n".join(lines[lo - 1:hi])
    ns = {}
    exec(compile(src, "reference:README.md", "exec"), ns)  # noqa: S102 - the reference oracle
    return ns["UNet"]


README_BCEDICE_LINES = (1855, 1893)  # `class BCEDiceLoss(nn.Module):` .. `return total_loss, bce_loss, dice_loss`


def load_reference_bcedice(ref_root):
    with open(os.path.join(ref_root, "README.md"), encoding="utf-8") as f:
        lines = f.read().split("\n")
    lo, hi = README_BCEDICE_LINES
    ns = {"torch": torch, "nn": torch.nn}
    exec(compile("\n".join(lines[lo - 1:hi]), "reference:README.md", "exec"), ns)  # noqa: S102
    return ns["BCEDiceLoss"]


README_DICE_LINES = (2115, 2120)  # `def compute_dice(pred, target, smooth=1e-6):` .. its return


def load_reference_compute_dice(ref_root):
    with open(os.path.join(ref_root, "README.md"), encoding="utf-8") as f:
        lines = f.read().split("\n")
    lo, hi = README_DICE_LINES
    ns = {"torch": torch}
    exec(compile("\n".join(lines[lo - 1:hi]), "reference:README.md", "exec"), ns)  # noqa: S102
    return ns["compute_dice"]


def make_bcedice(ref_root):
    """Loss values and logit gradient of the reference's BCEDiceLoss(0.5, 0.5, pos_weight=3) (README.md:2169-2170)."""
    BCEDiceLoss = load_reference_bcedice(ref_root)
    g = torch.Generator().manual_seed(21)
    x = (torch.randn(3, 1, 24, 40, generator=g) * 2.5).requires_grad_(True)
    t = (torch.rand(3, 1, 24, 40, generator=g) < 0.15).float()
    crit = BCEDiceLoss(bce_weight=0.5, dice_weight=0.5, pos_weight=torch.tensor([3.0]))
    total, bce, dice = crit(x, t)
    total.backward()
    # Dice metric of the validation loop on the same logits (README.md:2103-2104, :2115-2120)
    metric = load_reference_compute_dice(ref_root)(torch.sigmoid(x.detach()) > 0.5, t)
    np.savez_compressed(os.path.join(HERE, "bcedice.npz"), x=x.detach().numpy(), t=t.numpy(),
                        total=np.float32(total.item()), bce=np.float32(bce.item()), dice=np.float32(dice.item()),
                        gx=x.grad.numpy(), dice_metric=np.float32(metric.item()))
    print("bcedice golden written:", total.item(), bce.item(), dice.item())


def to_t(sd):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def make_adamw(ref_root):
    """Two steps of the training loop as the reference configures it (README.md:2060-2084 with
    BCEDiceLoss(0.5, 0.5, pos_weight=3) :2169-2170 and optim.AdamW :2173-2174) on the tiny config, for two
    (lr, weight_decay) pairs: the reference's (1e-4, 1e-4) - whose decay factor 1 - 1e-8 rounds to 1 in fp32 -
    and an amplified (1e-2, 1e-1) that makes the decoupled decay visible."""
    UNet = load_reference_unet(ref_root)
    BCEDiceLoss = load_reference_bcedice(ref_root)
    feats = [4, 8]
    sd = seeded_state_dict(feats, seed=1)
    rng = np.random.default_rng(17)
    xb = rng.standard_normal((4, 3, 32, 32)).astype(np.float32)
    tb = (rng.random((4, 1, 32, 32)) < 0.085).astype(np.float32)
    out = {"input": xb, "target": tb}
    for tag, lr, wd in (("ref", 1e-4, 1e-4), ("amp", 1e-2, 1e-1)):
        m = UNet(3, 1, features=feats)
        m.load_state_dict(to_t(sd), strict=True)
        m.train()
        opt = torch.optim.AdamW(m.parameters(), lr=lr, weight_decay=wd)
        crit = BCEDiceLoss(bce_weight=0.5, dice_weight=0.5, pos_weight=torch.tensor([3.0]))
        for step in range(2):
            opt.zero_grad()
            total, bce, dice = crit(m(torch.from_numpy(xb)), torch.from_numpy(tb))
            total.backward()
            opt.step()
            out[f"{tag}/loss{step}"] = np.array([total.item(), bce.item(), dice.item()], dtype=np.float64)
        out[f"{tag}/lr"], out[f"{tag}/wd"] = np.float64(lr), np.float64(wd)
        for k, v in m.state_dict().items():
            out[f"{tag}/post/{k}"] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "tiny_f4_8_adamw2.npz"), **out)
    print("adamw golden written:", {k: out[k] for k in out if "loss" in k})


def normalize(frames_u8):
    mean = torch.tensor([123.675, 116.28, 103.53])
    std = torch.tensor([58.395, 57.12, 57.375])
    x = (torch.from_numpy(frames_u8).float() - mean) / std
    return x.permute(0, 3, 1, 2).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--only", default="", help="'bcedice' / 'adamw': regenerate only that fixture")
    args = ap.parse_args()
    if args.only == "bcedice":
        make_bcedice(args.reference)
        return
    if args.only == "adamw":
        make_adamw(args.reference)
        return
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    UNet = load_reference_unet(args.reference)

    # ---- sanity pins quoted by the reference itself -------------------------
    model = UNet(in_channels=3, out_channels=1)
    n_params = sum(p.numel() for p in model.parameters())
    assert n_params == 31_037_633, n_params          # README.md:2288
    assert len(model.state_dict()) == 118

    # ---- (1) tiny config: exact op order, concat order, convT layout ---------
    feats = [4, 8]
    sd = seeded_state_dict(feats, seed=1)
    tiny = UNet(3, 1, features=feats)
    tiny.load_state_dict(to_t(sd), strict=True)
    tiny.eval()
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2, 3, 32, 32)).astype(np.float32)
    with torch.no_grad():
        y = tiny(torch.from_numpy(x)).numpy()
    out = {"input": x, "logits": y}
    out.update({"sd/" + k: v for k, v in sd.items()})
    np.savez_compressed(os.path.join(HERE, "tiny_f4_8_eval.npz"), **out)

    # tiny config, one training step (train-mode BN, BCE-with-logits, Adam lr 1e-4)
    tiny.train()
    xb = rng.standard_normal((4, 3, 32, 32)).astype(np.float32)
    tb = (rng.random((4, 1, 32, 32)) < 0.085).astype(np.float32)
    opt = torch.optim.Adam(tiny.parameters(), lr=1e-4)
    crit = torch.nn.BCEWithLogitsLoss()
    opt.zero_grad()
    logits = tiny(torch.from_numpy(xb))
    loss = crit(logits, torch.from_numpy(tb))
    loss.backward()
    grads = {k: p.grad.detach().numpy().copy() for k, p in tiny.named_parameters()}
    opt.step()
    post = {k: v.detach().numpy().copy() for k, v in tiny.state_dict().items()}
    out = {"input": xb, "target": tb, "loss": np.float32(loss.item()),
           "logits": logits.detach().numpy()}
    out.update({"grad/" + k: v for k, v in grads.items()})
    out.update({"post/" + k: v for k, v in post.items()})
    np.savez_compressed(os.path.join(HERE, "tiny_f4_8_train_step.npz"), **out)

    # ---- (2) model A on the reference's own test frame -----------------------
    from PIL import Image
    frame = np.asarray(Image.open(os.path.join(args.reference, "test_images", "frame_001410.jpg")).convert("RGB"))
    assert frame.shape == (224, 224, 3) and frame.dtype == np.uint8
    frame.tofile(os.path.join(HERE, "frame_001410_rgb_u8.bin"))
    sdA = seeded_state_dict(DEFAULT_FEATURES, seed=0)
    model.load_state_dict(to_t(sdA), strict=True)
    model.eval()
    with torch.no_grad():
        logitsA = model(normalize(frame[None])).numpy()[0, 0]
    mask = (1.0 / (1.0 + np.exp(-logitsA)) > 0.5).astype(np.uint8) * 255   # src/unet.py:63-67
    near = int((np.abs(logitsA) < 1e-3).sum())
    np.savez_compressed(os.path.join(HERE, "modelA_frame_001410.npz"),
                        logits=logitsA, mask=mask, near_zero_1e3=np.int64(near),
                        n_params=np.int64(n_params))

    # model A, 2 synthetic frames (bench-style input) with per-stage summaries
    frames = synthetic_frames(2, seed=0)
    acts = {}
    hooks = []
    names = {}
    for i, m in enumerate(model.encoder_blocks):
        names[m] = f"enc{i}"
    names[model.bottleneck] = "bottleneck"
    for j in range(4):
        names[model.decoder_blocks[2 * j]] = f"up{j}"
        names[model.decoder_blocks[2 * j + 1]] = f"dec{j}"
    for m, nm in names.items():
        hooks.append(m.register_forward_hook(lambda mod, i, o, nm=nm: acts.__setitem__(nm, o.detach())))
    with torch.no_grad():
        lg = model(normalize(frames)).numpy()
    for h in hooks:
        h.remove()
    out = {"logits": lg[:, 0]}
    for nm, a in acts.items():
        out[f"stat/{nm}/mean"] = np.float64(a.double().mean().item())
        out[f"stat/{nm}/absmax"] = np.float64(a.abs().max().item())
        out[f"stat/{nm}/l2"] = np.float64(a.double().pow(2).sum().sqrt().item())
        # a thin slice is enough to localise a wrong stage: first frame, 8 channels, one row
        out[f"slice/{nm}"] = a[0, :8, a.shape[2] // 2, :].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "modelA_synth2.npz"), **out)

    # ---- (4) model A, batch-4 training step (summaries only; weights are 124 MB)
    model.load_state_dict(to_t(sdA), strict=True)
    model.train()
    xb = normalize(synthetic_frames(4, seed=3))
    tb = torch.from_numpy(synthetic_targets(4, seed=3))
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    opt.zero_grad()
    lg = model(xb)
    loss = torch.nn.BCEWithLogitsLoss()(lg, tb)
    loss.backward()
    out = {"loss": np.float64(loss.item())}
    for k, p in model.named_parameters():
        out["gradnorm/" + k] = np.float64(p.grad.double().norm().item())
    opt.step()
    for k, v in model.state_dict().items():
        out["postsum/" + k] = np.float64(v.double().sum().item())
    np.savez_compressed(os.path.join(HERE, "modelA_train_step_b4.npz"), **out)

    # ---- (3) per-op vectors (torch.nn modules the reference composes) --------
    g = torch.Generator().manual_seed(11)
    ops = {}
    xin = torch.randn(2, 5, 6, 8, generator=g)
    conv = torch.nn.Conv2d(5, 7, 3, padding=1, bias=False)
    ops["conv3x3/x"], ops["conv3x3/w"] = xin.numpy(), conv.weight.detach().numpy()
    ops["conv3x3/y"] = conv(xin).detach().numpy()
    ct = torch.nn.ConvTranspose2d(5, 3, kernel_size=2, stride=2)
    ops["convt/w"], ops["convt/b"] = ct.weight.detach().numpy(), ct.bias.detach().numpy()
    ops["convt/y"] = ct(xin).detach().numpy()
    ops["maxpool/y"] = torch.nn.MaxPool2d(2, 2)(xin).numpy()
    bn = torch.nn.BatchNorm2d(5)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g)
        bn.bias.uniform_(-0.5, 0.5, generator=g)
        bn.running_mean.uniform_(-0.5, 0.5, generator=g)
        bn.running_var.uniform_(0.5, 1.5, generator=g)
    for k in ("weight", "bias", "running_mean", "running_var"):
        ops[f"bn/{k}"] = getattr(bn, k).detach().numpy().copy()
    bn.eval()
    ops["bn/y_eval"] = bn(xin).detach().numpy()
    bn.train()
    xg = xin.clone().requires_grad_(True)
    yb = bn(xg)
    gy = torch.randn(yb.shape, generator=g)
    yb.backward(gy)
    ops["bn/y_train"], ops["bn/gy"] = yb.detach().numpy(), gy.numpy()
    ops["bn/gx"], ops["bn/gw"], ops["bn/gb"] = xg.grad.numpy(), bn.weight.grad.numpy(), bn.bias.grad.numpy()
    ops["bn/new_mean"], ops["bn/new_var"] = bn.running_mean.numpy().copy(), bn.running_var.numpy().copy()
    lgt = torch.randn(2, 1, 6, 8, generator=g).mul(3).requires_grad_(True)
    tgt = (torch.rand(2, 1, 6, 8, generator=g) < 0.3).float()
    l = torch.nn.BCEWithLogitsLoss()(lgt, tgt)
    l.backward()
    ops["bce/x"], ops["bce/t"] = lgt.detach().numpy(), tgt.numpy()
    ops["bce/loss"], ops["bce/gx"] = np.float32(l.item()), lgt.grad.numpy()
    p = torch.nn.Parameter(torch.randn(64, generator=g))
    o = torch.optim.Adam([p], lr=1e-4)
    ops["adam/p0"] = p.detach().numpy().copy()
    gs = []
    for _ in range(3):
        p.grad = torch.randn(64, generator=g)
        gs.append(p.grad.numpy().copy())
        o.step()
    ops["adam/g"], ops["adam/p3"] = np.stack(gs), p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **ops)

    make_bcedice(args.reference)
    make_adamw(args.reference)
    print("near-zero logits (<1e-3) on test frame:", near, "of", logitsA.size)
    print("mask positive fraction:", float((mask > 0).mean()))
    for f in sorted(os.listdir(HERE)):
        print(f"{os.path.getsize(os.path.join(HERE, f)):>9d}  {f}")


if __name__ == "__main__":
    main()
