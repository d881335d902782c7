"""State-dict naming and a seeded weight recipe for the lane-segmentation U-Net.

The reference ships no float weights, so parity tests
regenerate identical weights on both sides from this recipe.  Key names and
tensor shapes follow the state_dict of the reference's `UNet`
(reference README.md:1424-1447):

  encoder_blocks.{i}.{0,3}.weight                       (O, I, 3, 3)
  encoder_blocks.{i}.{1,4}.{weight,bias,running_mean,running_var,num_batches_tracked}
  bottleneck.{0,1,3,4}.*                                same pattern
  decoder_blocks.{2j}.{weight (I, O, 2, 2), bias}       ConvTranspose2d
  decoder_blocks.{2j+1}.{0,1,3,4}.*                     DoubleConv
  output.{weight (out, f0, 1, 1), bias}

Only numpy is used here, so the recipe is bit-reproducible on any machine
(PCG64 stream + float32 casts).
"""
from __future__ import annotations

import numpy as np

DEFAULT_FEATURES = (64, 128, 256, 512)
BN_EPS = 1e-5  # nn.BatchNorm2d default, reference README.md:1453
BN_MOMENTUM = 0.1

# Normalisation baked into the shipped blob (reference
# README.md:3110-3111): (u8 - mean) / std per RGB channel.
INPUT_MEAN = (123.675, 116.28, 103.53)
INPUT_STD = (58.395, 57.12, 57.375)


def double_conv_keys(prefix: str):
    """Keys of one `_conv_block` Sequential (reference README.md:1449-1458)."""
    keys = []
    for conv_i, bn_i in ((0, 1), (3, 4)):
        keys.append(f"{prefix}.{conv_i}.weight")
        for leaf in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            keys.append(f"{prefix}.{bn_i}.{leaf}")
    return keys


def state_dict_spec(features=DEFAULT_FEATURES, in_channels=3, out_channels=1):
    """Ordered list of (key, shape, kind) in the order nn.Module registers them.

    kind is one of conv3 / bn_w / bn_b / bn_mean / bn_var / bn_count / convt_w /
    convt_b / head_w / head_b.
    """
    features = list(features)
    spec = []

    def add_double(prefix, cin, cout):
        for conv_i, bn_i, ci in ((0, 1, cin), (3, 4, cout)):
            spec.append((f"{prefix}.{conv_i}.weight", (cout, ci, 3, 3), "conv3"))
            spec.append((f"{prefix}.{bn_i}.weight", (cout,), "bn_w"))
            spec.append((f"{prefix}.{bn_i}.bias", (cout,), "bn_b"))
            spec.append((f"{prefix}.{bn_i}.running_mean", (cout,), "bn_mean"))
            spec.append((f"{prefix}.{bn_i}.running_var", (cout,), "bn_var"))
            spec.append((f"{prefix}.{bn_i}.num_batches_tracked", (), "bn_count"))

    cin = in_channels
    for i, f in enumerate(features):
        add_double(f"encoder_blocks.{i}", cin, f)
        cin = f
    for j, f in enumerate(reversed(features)):
        spec.append((f"decoder_blocks.{2 * j}.weight", (2 * f, f, 2, 2), "convt_w"))
        spec.append((f"decoder_blocks.{2 * j}.bias", (f,), "convt_b"))
        add_double(f"decoder_blocks.{2 * j + 1}", 2 * f, f)
    add_double("bottleneck", features[-1], 2 * features[-1])
    spec.append(("output.weight", (out_channels, features[0], 1, 1), "head_w"))
    spec.append(("output.bias", (out_channels,), "head_b"))
    return spec


def num_parameters(features=DEFAULT_FEATURES, in_channels=3, out_channels=1) -> int:
    n = 0
    for _, shape, kind in state_dict_spec(features, in_channels, out_channels):
        if kind in ("bn_mean", "bn_var", "bn_count"):
            continue
        n += int(np.prod(shape)) if shape else 1
    return n


def seeded_state_dict(features=DEFAULT_FEATURES, seed=0, in_channels=3, out_channels=1):
    """Deterministic 'trained-like' weights as a dict of numpy arrays.

    Conv weights are He-normal so activations stay O(1) through the 23 layers;
    BatchNorm affine and running statistics are randomised around (1, 0, 0, 1)
    so that folding mistakes (gamma vs. beta, mean vs. var) show up in parity.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for key, shape, kind in state_dict_spec(features, in_channels, out_channels):
        if kind == "conv3":
            fan_in = shape[1] * 9
            v = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        elif kind == "convt_w":
            v = rng.standard_normal(shape) * np.sqrt(1.0 / shape[0])
        elif kind == "head_w":
            v = rng.standard_normal(shape) * np.sqrt(4.0 / shape[1])
        elif kind == "bn_w":
            v = 1.0 + 0.2 * rng.uniform(-1.0, 1.0, shape)
        elif kind in ("bn_b", "bn_mean", "convt_b"):
            v = 0.1 * rng.uniform(-1.0, 1.0, shape)
        elif kind == "head_b":
            v = -0.25 + 0.1 * rng.uniform(-1.0, 1.0, shape)
        elif kind == "bn_var":
            v = rng.uniform(0.6, 1.4, shape)
        elif kind == "bn_count":
            sd[key] = np.array(0, dtype=np.int64)
            continue
        else:  # pragma: no cover
            raise AssertionError(kind)
        sd[key] = np.ascontiguousarray(v, dtype=np.float32)
    return sd


def synthetic_frames(n, h=224, w=224, seed=0):
    """uint8 NHWC RGB frames, uniform [0,255]."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)


def synthetic_targets(n, h=224, w=224, seed=0, p=0.085):
    """Bernoulli(p) lane masks as float32 (N,1,H,W)."""
    rng = np.random.default_rng(seed + 10_000)
    return (rng.random((n, 1, h, w)) < p).astype(np.float32)
