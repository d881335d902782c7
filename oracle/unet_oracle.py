"""CPU oracle for the U-Net lane-segmentation hot path.  TEST INFRASTRUCTURE ONLY.

This is a restatement (not a copy) of the reference's float network and of the
pre/post-processing around it, written with torch-CPU functional ops.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import it; the product path (`unet_lane_detection_amd`) never does and
fails loudly when its HIP library is missing.

Parity pin: the reference's own tests pin nothing for this path (SURVEY.md
section 4).  The oracle is pinned against the reference's `UNet` class itself
(reference README.md:1418-1481), executed in the build container by
`tests/golden/make_golden.py`; the resulting vectors are committed under
`tests/golden/` and re-checked by `tests/test_oracle_golden.py`.

Every function cites the reference lines it follows (paths relative to the
reference root).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
INPUT_MEAN = (123.675, 116.28, 103.53)   # README.md:3110
INPUT_STD = (58.395, 57.12, 57.375)      # README.md:3111


def to_torch_state(sd):
    """numpy/torch state dict -> dict of torch CPU tensors (fp32, int64 counters)."""
    out = {}
    for k, v in sd.items():
        t = torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v.detach().cpu()
        out[k] = t.clone()
    return out


def infer_features(sd):
    """Recover the `features` list from encoder conv shapes (README.md:1424)."""
    feats = []
    i = 0
    while f"encoder_blocks.{i}.0.weight" in sd:
        feats.append(int(sd[f"encoder_blocks.{i}.0.weight"].shape[0]))
        i += 1
    return feats


# ----------------------------------------------------------------------------
# a9: input normalisation baked into the deployed blob (README.md:3110-3111)
# ----------------------------------------------------------------------------
def normalize_u8_nhwc(frames_u8):
    """(N,H,W,3) uint8 RGB -> (N,3,H,W) float32, (x - mean) / std per channel."""
    x = torch.as_tensor(np.asarray(frames_u8)).to(torch.float32)
    mean = torch.tensor(INPUT_MEAN, dtype=torch.float32)
    std = torch.tensor(INPUT_STD, dtype=torch.float32)
    x = (x - mean) / std
    return x.permute(0, 3, 1, 2).contiguous()


# ----------------------------------------------------------------------------
# a1-a3: `_conv_block` (README.md:1449-1458)
# ----------------------------------------------------------------------------
def conv3x3(x, w):
    """3x3 cross-correlation, stride 1, pad 1, no bias (README.md:1452, :1455)."""
    return F.conv2d(x, w, bias=None, stride=1, padding=1)


def bn_eval(x, gamma, beta, mean, var):
    """Eval BatchNorm2d: (x-mu)/sqrt(var+eps)*gamma+beta (README.md:1453)."""
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * gamma)[None, :, None, None] + beta[None, :, None, None]


def bn_train(x, gamma, beta, run_mean, run_var, count):
    """Training BatchNorm2d: batch mean, biased var for normalisation; running
    stats updated with momentum 0.1 and the unbiased variance (README.md:1453).
    Returns (y, new_run_mean, new_run_var, new_count)."""
    n = x.shape[0] * x.shape[2] * x.shape[3]
    mean = x.mean(dim=(0, 2, 3))
    var_b = x.var(dim=(0, 2, 3), unbiased=False)
    y = (x - mean[None, :, None, None]) * torch.rsqrt(var_b + BN_EPS)[None, :, None, None]
    y = y * gamma[None, :, None, None] + beta[None, :, None, None]
    with torch.no_grad():
        var_u = var_b * (n / max(n - 1, 1))
        new_mean = (1 - BN_MOMENTUM) * run_mean + BN_MOMENTUM * mean
        new_var = (1 - BN_MOMENTUM) * run_var + BN_MOMENTUM * var_u
    return y, new_mean.detach(), new_var.detach(), count + 1


def double_conv(x, sd, prefix, training=False, new_stats=None, taps=None):
    """Conv-BN-ReLU twice (README.md:1449-1458)."""
    for conv_i, bn_i in ((0, 1), (3, 4)):
        x = conv3x3(x, sd[f"{prefix}.{conv_i}.weight"])
        if taps is not None:
            taps[f"z/{prefix}.{conv_i}"] = x
        g, b = sd[f"{prefix}.{bn_i}.weight"], sd[f"{prefix}.{bn_i}.bias"]
        m, v = sd[f"{prefix}.{bn_i}.running_mean"], sd[f"{prefix}.{bn_i}.running_var"]
        if training:
            x, nm, nv, nc = bn_train(x, g, b, m, v, sd[f"{prefix}.{bn_i}.num_batches_tracked"])
            if new_stats is not None:
                new_stats[f"{prefix}.{bn_i}.running_mean"] = nm
                new_stats[f"{prefix}.{bn_i}.running_var"] = nv
                new_stats[f"{prefix}.{bn_i}.num_batches_tracked"] = nc
        else:
            x = bn_eval(x, g, b, m, v)
        x = torch.relu(x)  # README.md:1454, :1457
    return x


def maxpool2x2(x):
    """MaxPool2d(2,2), floor mode (README.md:1429, :1467)."""
    return F.max_pool2d(x, kernel_size=2, stride=2)


def upconv2x2(x, w, b):
    """ConvTranspose2d(k=2,s=2): y[co,2i+a,2j+b] = sum_ci x[ci,i,j] W[ci,co,a,b] + bias
    (README.md:1442, :1476)."""
    return F.conv_transpose2d(x, w, b, stride=2)


def head1x1(x, w, b):
    """1x1 conv with bias producing logits (README.md:1447, :1481)."""
    return F.conv2d(x, w, b)


# ----------------------------------------------------------------------------
# a0-a7: the forward pass (README.md:1460-1481)
# ----------------------------------------------------------------------------
def forward(sd, x, training=False, new_stats=None, taps=None):
    """x: (N,Cin,H,W) float32 -> logits (N,Cout,H,W).  H, W must be divisible
    by 2**len(features) (the reference's cat raises otherwise)."""
    feats = infer_features(sd)
    depth = len(feats)
    if x.shape[2] % (1 << depth) or x.shape[3] % (1 << depth):
        raise ValueError(f"H and W must be multiples of {1 << depth}, got {tuple(x.shape)}")
    skips = []
    for i in range(depth):                                   # README.md:1464-1467
        x = double_conv(x, sd, f"encoder_blocks.{i}", training, new_stats, taps)
        if taps is not None:
            taps[f"enc{i}"] = x
        skips.append(x)
        x = maxpool2x2(x)
    x = double_conv(x, sd, "bottleneck", training, new_stats, taps)  # README.md:1470
    if taps is not None:
        taps["bottleneck"] = x
    for j in range(depth):                                   # README.md:1475-1479
        x = upconv2x2(x, sd[f"decoder_blocks.{2 * j}.weight"], sd[f"decoder_blocks.{2 * j}.bias"])
        if taps is not None:
            taps[f"up{j}"] = x
        x = torch.cat([skips[depth - 1 - j], x], dim=1)      # skip first, README.md:1478
        x = double_conv(x, sd, f"decoder_blocks.{2 * j + 1}", training, new_stats, taps)
        if taps is not None:
            taps[f"dec{j}"] = x
    return head1x1(x, sd["output.weight"], sd["output.bias"])  # README.md:1481


# ----------------------------------------------------------------------------
# Boundary (src/py_utils/rknn_executor.py:26-38) and caller semantics (src/unet.py)
# ----------------------------------------------------------------------------
def container_run(sd, inputs):
    """What `RKNN_model_container.run` returns for the float network: a list with
    one (N,1,H,W) float32 array of probabilities (the deployed blob fuses the
    sigmoid into its last conv).  `inputs` is a uint8 NHWC
    array or a list holding one (src/py_utils/rknn_executor.py:31-34)."""
    if not isinstance(inputs, (list, tuple)):
        inputs = [inputs]
    x = normalize_u8_nhwc(inputs[0])
    with torch.no_grad():
        logits = forward(sd, x)
    return [torch.sigmoid(logits).numpy()]


def postprocess_output(output, threshold=0.5):
    """src/unet.py:44-72 without the final cv2.resize (identity when the frame is
    already 224x224): slice [0,0], sigmoid only if the values look like logits,
    threshold, scale to {0,255} uint8."""
    mask = output[0] if isinstance(output, (list, tuple)) else output
    mask = np.asarray(mask)
    if mask.ndim == 4:
        mask = mask[0, 0]                                    # src/unet.py:53-54
    elif mask.ndim == 3:
        mask = mask[0]
    if mask.dtype == np.int8:
        mask = mask.astype(np.float32)                       # src/unet.py:59-60
    if mask.max() > 1.0 or mask.min() < 0.0:                 # src/unet.py:63-64
        mask = 1 / (1 + np.exp(-mask))
    return (mask > threshold).astype(np.uint8) * 255         # src/unet.py:67


def logits_to_mask(logits):
    """Binary mask in the logit domain: sigmoid(x) > 0.5  <=>  x > 0."""
    return (np.asarray(logits) > 0).astype(np.uint8) * 255


def mask_iou(a, b):
    a = np.asarray(a) > 0
    b = np.asarray(b) > 0
    union = np.logical_or(a, b).sum()
    return 1.0 if union == 0 else float(np.logical_and(a, b).sum()) / float(union)


# ----------------------------------------------------------------------------
# a12: BCE-with-logits (README.md:1694-1709), a13: Adam step (README.md:2071-2079)
# ----------------------------------------------------------------------------
def bce_with_logits(logits, target):
    """mean over all elements of max(x,0) - x*t + log1p(exp(-|x|)) (pos_weight=None)."""
    x, t = logits, target
    return (torch.clamp(x, min=0) - x * t + torch.log1p(torch.exp(-torch.abs(x)))).mean()


def bce_with_logits_grad(logits, target):
    """d(mean BCE)/d(logits) = (sigmoid(x) - t) / numel."""
    return (torch.sigmoid(logits) - target) / logits.numel()


def bce_dice_loss(logits, target, bce_weight=0.5, dice_weight=0.5, pos_weight=None, smooth=1e-6):
    """BCEDiceLoss of the training script (README.md:1855-1893): returns (total, bce, dice)."""
    x, t = logits, target.float()
    pw = 1.0 if pos_weight is None else float(pos_weight)
    log_s = F.logsigmoid(x)
    log_1ms = F.logsigmoid(-x)
    bce = (-(pw * t * log_s + (1 - t) * log_1ms)).mean()
    s = torch.sigmoid(x).reshape(-1)
    tf = t.reshape(-1)
    inter = (s * tf).sum()
    dice = (2.0 * inter + smooth) / (s.sum() + tf.sum() + smooth)
    dl = 1 - dice
    return bce_weight * bce + dice_weight * dl, bce, dl


def compute_dice(pred, target, smooth=1e-6):
    """Dice metric of the validation loop (README.md:2115-2120)."""
    p = pred.reshape(-1).float()
    t = target.reshape(-1).float()
    inter = (p * t).sum()
    return (2.0 * inter + smooth) / (p.sum() + t.sum() + smooth)


PARAM_KINDS = ("weight", "bias")


def is_parameter(key):
    return key.endswith(".weight") or key.endswith(".bias")


def loss_and_grads(sd, x, target, loss_fn=None):
    """One training-mode forward/backward with autograd over the functional
    restatement above.  Returns (loss, grads dict, new BN buffers, logits)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if is_parameter(k)}
    full = dict(sd)
    full.update(params)
    new_stats = {}
    logits = forward(full, x, training=True, new_stats=new_stats)
    loss = bce_with_logits(logits, target) if loss_fn is None else loss_fn(logits, target)
    keys = list(params.keys())
    gs = torch.autograd.grad(loss, [params[k] for k in keys])
    return loss.detach(), dict(zip(keys, gs)), new_stats, logits.detach()


def adam_step(p, g, m, v, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
              decoupled=False):
    """torch.optim.Adam / AdamW update, single tensor, `step` counted from 1.
    decoupled=True gives AdamW (README.md:2173); False gives Adam (BASELINE.json)."""
    if weight_decay != 0.0:
        if decoupled:
            p = p * (1 - lr * weight_decay)
        else:
            g = g + weight_decay * p
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def train_step(sd, opt, x, target, lr=1e-4, weight_decay=0.0, decoupled=False):
    """zero_grad -> forward -> BCE -> backward -> Adam (README.md:2060-2084).
    `opt` = {"step": int, "m": {k: tensor}, "v": {k: tensor}}; returns
    (loss, new_sd, new_opt, grads)."""
    loss, grads, new_stats, _ = loss_and_grads(sd, x, target)
    step = opt["step"] + 1
    new_sd = dict(sd)
    new_sd.update(new_stats)
    new_opt = {"step": step, "m": {}, "v": {}}
    for k, g in grads.items():
        m = opt["m"].get(k, torch.zeros_like(g))
        v = opt["v"].get(k, torch.zeros_like(g))
        p, m, v = adam_step(sd[k], g, m, v, step, lr=lr, weight_decay=weight_decay, decoupled=decoupled)
        new_sd[k], new_opt["m"][k], new_opt["v"][k] = p, m, v
    return loss, new_sd, new_opt, grads
