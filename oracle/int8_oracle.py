"""CPU oracle of the int8 tier (model B).  TEST INFRASTRUCTURE ONLY.

Integer-exact restatement of the quantised forward pass: quantise -> int32 accumulate -> requantise, following the
scheme the reference documents for its RKNN conversion (README.md:3106-3116 `asymmetric_quantized-8`, `channel`;
README.md:3370-3383 q = round(r / scale) + zero_point, r ~= (q - zero_point) * scale) and the network structure of
the float model (README.md:1460-1481; the deployed blob fuses BatchNorm into the convolutions and the sigmoid into
the head).

**Parity unpinned**: the arithmetic really lives in Rockchip's rknn-toolkit2 / the RK3588 NPU runtime, which is
absent here; the shipped .rknn blobs cannot be executed and the reference holds no quantised output for this path.
What this oracle pins is the HIP tier against an independent integer restatement of the same documented scheme (bit
for bit), and the quantiser against the README's formulas.

Exactness: all integer sums are formed with float64 convolutions of integer-valued tensors (every partial sum stays
far below 2^53), then cast to int64.  The requantisation multiplies float32(acc) by a float32 multiplier (one IEEE
multiply) and rounds half to even: numpy and the HIP kernel perform the same two IEEE operations.

Only tests/ (and a future cpu_baseline leg) may import this module.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

QMIN, QMAX = -128, 127


# ---- README.md:3370-3383 -------------------------------------------------------------------------------------
def affine_params(r_min, r_max):
    """scale = (r_max - r_min) / 255 with the range widened to contain 0; zero_point = round(-128 - r_min / scale)."""
    r_min, r_max = min(float(r_min), 0.0), max(float(r_max), 0.0)
    if r_max - r_min < 1e-30:
        return 1.0, 0
    scale = (r_max - r_min) / 255.0
    return scale, int(min(max(round(QMIN - r_min / scale), QMIN), QMAX))


def quantize(r, scale, zp):
    return np.clip(np.rint(np.asarray(r, dtype=np.float64) / scale) + zp, QMIN, QMAX).astype(np.int8)


def dequantize(q, scale, zp):
    return (np.asarray(q, dtype=np.float64) - zp) * scale


# ---- integer operators ----------------------------------------------------------------------------------------
def requantize(acc, bias_q, mult, y_zp, relu):
    """acc (N,C,H,W) int64, bias_q / mult per channel -> int8: clamp(rint(float32(acc + bias) * float32(mult)) + zp)."""
    t = (acc + np.asarray(bias_q, dtype=np.int64)[None, :, None, None]).astype(np.float32)
    y = np.rint(t * np.asarray(mult, dtype=np.float32)[None, :, None, None]).astype(np.int64) + int(y_zp)
    return np.clip(y, int(y_zp) if relu else QMIN, QMAX).astype(np.int8)


def conv3x3_acc(x_q, x_zp, w_q, w_zp):
    """sum over the 3x3 window and input channels of (qx - zx)(qw - zw[co]); positions outside the image contribute 0
    (zero padding in the real domain, README.md:1452).  x_q (N,C,H,W) int8, w_q (O,C,3,3) int8 -> (N,O,H,W) int64."""
    x = torch.from_numpy(x_q.astype(np.float64) - float(x_zp))
    w = torch.from_numpy(w_q.astype(np.float64) - np.asarray(w_zp, dtype=np.float64)[:, None, None, None])
    return F.conv2d(x, w, padding=1).numpy().round().astype(np.int64)


def upconv2x2_acc(x_q, x_zp, w_q, w_zp):
    """ConvTranspose2d k2 s2 (README.md:1442): w_q (I,O,2,2), zero points per OUTPUT channel."""
    x = torch.from_numpy(x_q.astype(np.float64) - float(x_zp))
    w = torch.from_numpy(w_q.astype(np.float64) - np.asarray(w_zp, dtype=np.float64)[None, :, None, None])
    return F.conv_transpose2d(x, w, stride=2).numpy().round().astype(np.int64)


def maxpool2x2(x_q):
    n, c, h, w = x_q.shape
    return x_q.reshape(n, c, h // 2, 2, w // 2, 2).max(axis=(3, 5))


def quantize_input(frames_u8, lut):
    """(N,H,W,3) uint8 -> (N,3,H,W) int8 through the per-channel table (normalisation + quantisation)."""
    f = np.asarray(frames_u8)
    return np.stack([np.asarray(lut)[c][f[..., c]] for c in range(3)], axis=1).astype(np.int8)


# ---- the forward pass (README.md:1460-1481 on integers) ----------------------------------------------------------
def forward(qm, frames_u8, taps=None):
    """qm: the flat dict produced by the product's quantiser (data only).  Returns float32 logits (N,1,H,W)."""
    feats = [int(f) for f in qm["features"]]
    d = len(feats)

    def unit(key, x, transposed=False):
        fn = upconv2x2_acc if transposed else conv3x3_acc
        acc = fn(x, int(qm[key + ".x_zp"]), qm[key + ".w_q"], qm[key + ".w_zp"])
        y = requantize(acc, qm[key + ".bias_q"], qm[key + ".mult"], int(qm[key + ".y_zp"]), bool(int(qm[key + ".relu"])))
        if taps is not None:
            taps[key] = y
        return y

    x = quantize_input(frames_u8, qm["input.lut"])
    if taps is not None:
        taps["input"] = x
    skips = []
    for l in range(d):
        x = unit(f"encoder_blocks.{l}.0", x)
        x = unit(f"encoder_blocks.{l}.3", x)
        skips.append(x)
        x = maxpool2x2(x)
    x = unit("bottleneck.0", x)
    x = unit("bottleneck.3", x)
    for j in range(d):
        up = unit(f"decoder_blocks.{2 * j}", x, transposed=True)
        x = np.concatenate([skips[d - 1 - j], up], axis=1)            # skip first (README.md:1478)
        x = unit(f"decoder_blocks.{2 * j + 1}.0", x)
        x = unit(f"decoder_blocks.{2 * j + 1}.3", x)
    # head: 1x1 conv, real-valued logits (the blob's ConvSigmoid applies the sigmoid to them)
    xz = x.astype(np.int64) - int(qm["output.x_zp"])
    wz = qm["output.w_q"].astype(np.int64).reshape(1, -1) - np.asarray(qm["output.w_zp"], dtype=np.int64).reshape(1, 1)
    acc = np.einsum("nchw,oc->nohw", xz, wz) + np.asarray(qm["output.bias_q"], dtype=np.int64)[None, :, None, None]
    return (acc.astype(np.float32) * np.asarray(qm["output.mult"], dtype=np.float32)[None, :, None, None]).astype(np.float32)


# ---- float activations ranges for calibration checks (README.md:3046-3078: min/max over calibration frames) -------
def float_ranges(sd_float, frames_u8):
    """Per-tensor (min, max) of the float model's activations on `frames_u8`, named as the product's
    quant.tensor_names(): the CPU counterpart of the device calibration pass."""
    from . import unet_oracle as O
    sd = O.to_torch_state(sd_float)
    d = len(O.infer_features(sd))
    taps = {}
    x = O.normalize_u8_nhwc(frames_u8)
    with torch.no_grad():
        O.forward(sd, x, taps=taps)
    rng = {"input": (float(x.min()), float(x.max()))}

    def mm(*ts):
        return (min(float(t.min()) for t in ts), max(float(t.max()) for t in ts))

    # taps: 'z/<prefix>.<conv>' raw conv outputs, 'enc{i}' / 'bottleneck' / 'up{j}' / 'dec{j}' block outputs
    for l in range(d):
        rng[f"cat{l}"] = mm(taps[f"enc{l}"], taps[f"up{d - 1 - l}"])
    for j in range(d):
        rng[f"dec{j}.b"] = mm(taps[f"dec{j}"])
    rng["bott.b"] = mm(taps["bottleneck"])
    # first conv of every block: recompute BN + ReLU of the tapped raw conv output
    def first_act(prefix):
        z = taps[f"z/{prefix}.0"]
        return torch.relu(O.bn_eval(z, sd[f"{prefix}.1.weight"], sd[f"{prefix}.1.bias"], sd[f"{prefix}.1.running_mean"],
                                    sd[f"{prefix}.1.running_var"]))
    for l in range(d):
        rng[f"enc{l}.a"] = mm(first_act(f"encoder_blocks.{l}"))
    rng["bott.a"] = mm(first_act("bottleneck"))
    for j in range(d):
        rng[f"dec{j}.a"] = mm(first_act(f"decoder_blocks.{2 * j + 1}"))
    return rng
