"""Post-training quantisation of the float U-Net to the int8 tier.

What the reference does (README.md:3039-3272, `convert_rknn.py`): the float model goes through
`rknn.config(mean_values, std_values, quantized_dtype='asymmetric_quantized-8', quantized_algorithm='normal',
quantized_method='channel')` (README.md:3106-3116) and `rknn.build(do_quantization=True, dataset=...)` with ~100
calibration frames (README.md:3046-3078); the quantisation rule it documents is (README.md:3370-3383)

    q = round(r / scale) + zero_point,   scale = (r_max - r_min) / 255,   r ~= (q - zero_point) * scale

with q in [-128, 127].  The conversion itself happens inside Rockchip's closed toolkit, so its exact choices
(rounding, range handling, operator fusion) are not in the reference: this module restates the documented scheme.
Parity against the shipped .rknn blobs is **unpinned** (they cannot be executed and hold no float weights).

Scheme implemented here (the integer-exact contract of the int8 tier; oracle/int8_oracle.py restates the forward):
  * activations: per-tensor asymmetric int8; (r_min, r_max) from calibration ("normal" = min/max), widened to contain
    0 so that zero is exactly representable (padding, ReLU);
  * weights: per-output-channel ('channel') asymmetric int8 of the BatchNorm-folded weights (the blob's 14 ConvRelu
    nodes carry folded weight/bias pairs);
  * bias: int32 in units of x_scale * w_scale[co];
  * the two halves of a concat share one (scale, zero_point): both producers requantise into the concat tensor;
  * input: (u8 - mean) / std quantised per tensor through a 256-entry table per channel (the normalisation lives in
    the blob, README.md:3110-3111);
  * requantisation: q_out = clamp(rint(float32(acc + bias_q) * float32(x_scale * w_scale[co] / y_scale)) + y_zp),
    float32 multiply and round-half-even, ReLU as the lower clamp y_zp;
  * head (ConvSigmoid in the blob): logits = float32(acc + bias_q) * float32(x_scale * w_scale), then sigmoid.
"""
from __future__ import annotations

import numpy as np

from .state import BN_EPS, INPUT_MEAN, INPUT_STD

QMIN, QMAX = -128, 127


def affine_params(r_min, r_max):
    """(scale, zero_point) of an asymmetric int8 tensor whose real range is [r_min, r_max] widened to contain 0
    (README.md:3376-3379).  Returns (float64 scale, int zero_point)."""
    r_min = min(float(r_min), 0.0)
    r_max = max(float(r_max), 0.0)
    if r_max - r_min < 1e-30:
        return 1.0, 0
    scale = (r_max - r_min) / 255.0
    zp = int(np.clip(np.rint(QMIN - r_min / scale), QMIN, QMAX))
    return scale, zp


def quantize(r, scale, zp):
    """q = clamp(round(r / scale) + zp) (README.md:3370)."""
    return np.clip(np.rint(np.asarray(r, dtype=np.float64) / scale) + zp, QMIN, QMAX).astype(np.int8)


def dequantize(q, scale, zp):
    return (np.asarray(q, dtype=np.float64) - zp) * scale


def quantize_weight_per_channel(w, axis=0):
    """Per-output-channel asymmetric int8 ('channel', README.md:3116): returns (w_q int8, scale float64[O], zp int32[O])."""
    w = np.asarray(w, dtype=np.float64)
    wm = np.moveaxis(w, axis, 0).reshape(w.shape[axis], -1)
    scales = np.empty(wm.shape[0], dtype=np.float64)
    zps = np.empty(wm.shape[0], dtype=np.int32)
    q = np.empty_like(wm, dtype=np.int8)
    for o in range(wm.shape[0]):
        scales[o], zps[o] = affine_params(wm[o].min(), wm[o].max())
        q[o] = quantize(wm[o], scales[o], zps[o])
    wq = np.moveaxis(q.reshape((w.shape[axis],) + tuple(np.delete(w.shape, axis))), 0, axis)
    return np.ascontiguousarray(wq), scales, zps


def fold_bn(sd, prefix, conv_i, bn_i):
    """Conv + eval BatchNorm -> (w_fold, b_fold), float64 (what 'ConvRelu' nodes of the blob carry)."""
    w = np.asarray(sd[f"{prefix}.{conv_i}.weight"], dtype=np.float64)
    g = np.asarray(sd[f"{prefix}.{bn_i}.weight"], dtype=np.float64)
    b = np.asarray(sd[f"{prefix}.{bn_i}.bias"], dtype=np.float64)
    m = np.asarray(sd[f"{prefix}.{bn_i}.running_mean"], dtype=np.float64)
    v = np.asarray(sd[f"{prefix}.{bn_i}.running_var"], dtype=np.float64)
    s = g / np.sqrt(v + BN_EPS)
    return w * s[:, None, None, None], b - m * s


def tensor_names(depth):
    """Activation tensors that carry quantisation parameters, in the order the calibration pass reports them
    (unet_forward_u8_ranges): input, per level the first encoder conv and the concat tensor (skip half = second encoder
    conv, upper half = the decoder's transposed conv), the two bottleneck convs, per decoder step its two convs."""
    names = ["input"]
    for l in range(depth):
        names += [f"enc{l}.a", f"cat{l}"]
    names += ["bott.a", "bott.b"]
    for j in range(depth):
        names += [f"dec{j}.a", f"dec{j}.b"]
    return names


def conv_units(features):
    """(state_dict prefix, conv index, bn index, input tensor, output tensor) of the 3x3 conv units in forward order."""
    d = len(features)
    units = []
    for l in range(d):
        src = "input" if l == 0 else f"cat{l - 1}.pool"
        units.append((f"encoder_blocks.{l}", 0, 1, src, f"enc{l}.a"))
        units.append((f"encoder_blocks.{l}", 3, 4, f"enc{l}.a", f"cat{l}"))
    units.append(("bottleneck", 0, 1, f"cat{d - 1}.pool", "bott.a"))
    units.append(("bottleneck", 3, 4, "bott.a", "bott.b"))
    for j in range(d):
        l = d - 1 - j
        units.append((f"decoder_blocks.{2 * j + 1}", 0, 1, f"cat{l}", f"dec{j}.a"))
        units.append((f"decoder_blocks.{2 * j + 1}", 3, 4, f"dec{j}.a", f"dec{j}.b"))
    return units


def quantize_model(state_dict, ranges, input_mean=INPUT_MEAN, input_std=INPUT_STD):
    """Float state_dict + calibrated activation ranges {tensor name: (r_min, r_max)} -> quantised model: a flat dict
    of numpy arrays (int8 weights, int32 zero points / biases, float32 multipliers) keyed '<unit>.<field>', the unit
    names being the reference's state_dict prefixes ('encoder_blocks.0.0', 'decoder_blocks.0', 'output', ...)."""
    feats = []
    while f"encoder_blocks.{len(feats)}.0.weight" in state_dict:
        feats.append(int(np.asarray(state_dict[f"encoder_blocks.{len(feats)}.0.weight"]).shape[0]))
    d = len(feats)
    q = {"features": np.asarray(feats, dtype=np.int32)}
    tq = {}
    for name in tensor_names(d):
        lo, hi = ranges[name]
        tq[name] = affine_params(lo, hi)
    for l in range(d):                      # a max-pooled tensor keeps its source's parameters
        tq[f"cat{l}.pool"] = tq[f"cat{l}"]

    # input: (u8 - mean) / std -> int8 through a table per channel
    s_in, z_in = tq["input"]
    lut = np.empty((3, 256), dtype=np.int8)
    for c in range(3):
        lut[c] = quantize((np.arange(256, dtype=np.float64) - input_mean[c]) / input_std[c], s_in, z_in)
    q["input.lut"] = lut
    q["input.zp"] = np.int32(z_in)
    q["input.scale"] = np.float32(s_in)

    def put_unit(key, w_q, w_s, w_z, b_fold, xname, yname, relu):
        s_x, z_x = tq[xname]
        q[key + ".w_q"] = w_q
        q[key + ".w_zp"] = w_z.astype(np.int32)
        q[key + ".w_scale"] = w_s.astype(np.float32)
        q[key + ".bias_q"] = np.rint(np.asarray(b_fold, dtype=np.float64) / (s_x * w_s)).astype(np.int64).clip(
            -2**31, 2**31 - 1).astype(np.int32)
        q[key + ".x_zp"] = np.int32(z_x)
        if yname is not None:
            s_y, z_y = tq[yname]
            q[key + ".mult"] = (s_x * w_s / s_y).astype(np.float32)
            q[key + ".y_zp"] = np.int32(z_y)
            q[key + ".y_scale"] = np.float32(s_y)
        else:
            q[key + ".mult"] = (s_x * w_s).astype(np.float32)      # head: real-valued logits
        q[key + ".relu"] = np.int32(1 if relu else 0)

    for prefix, ci, bi, xname, yname in conv_units(feats):
        w_fold, b_fold = fold_bn(state_dict, prefix, ci, bi)
        w_q, w_s, w_z = quantize_weight_per_channel(w_fold, axis=0)
        put_unit(f"{prefix}.{ci}", w_q, w_s, w_z, b_fold, xname, yname, True)
    for j in range(d):
        l = d - 1 - j
        w = np.asarray(state_dict[f"decoder_blocks.{2 * j}.weight"], dtype=np.float64)      # (I, O, 2, 2)
        b = np.asarray(state_dict[f"decoder_blocks.{2 * j}.bias"], dtype=np.float64)
        w_q, w_s, w_z = quantize_weight_per_channel(w, axis=1)
        src = "bott.b" if j == 0 else f"dec{j - 1}.b"
        put_unit(f"decoder_blocks.{2 * j}", w_q, w_s, w_z, b, src, f"cat{l}", False)
    w = np.asarray(state_dict["output.weight"], dtype=np.float64)
    w_q, w_s, w_z = quantize_weight_per_channel(w, axis=0)
    put_unit("output", w_q, w_s, w_z, np.asarray(state_dict["output.bias"], dtype=np.float64), f"dec{d - 1}.b", None,
             False)
    return q


def save_quantized(path, qmodel):
    np.savez_compressed(path, **qmodel)


def load_quantized(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}
