"""int8 tier (model B) on the CPU: the quantiser against the formulas the reference
documents (README.md:3370-3383), the integer oracle against brute-force loops, and the quantised forward against the
float model it was derived from.  Parity against the shipped .rknn blobs is unpinned (oracle/int8_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import int8_oracle as Q
from oracle import unet_oracle as O
from unet_lane_detection_amd import quant, state as S

FEATS_B = [32, 64, 128]        # the deployed blob's widths


def test_affine_formula_of_the_readme():
    # scale = (r_max - r_min) / 255, q = round(r / scale) + zp, r ~= (q - zp) * scale   (README.md:3370-3383)
    for lo, hi in [(-2.1179, 2.64), (0.0, 7.3), (-0.4, 0.0), (0.2, 3.0), (-5.0, -1.0)]:
        s, z = quant.affine_params(lo, hi)
        assert (s, z) == Q.affine_params(lo, hi)
        lo0, hi0 = min(lo, 0.0), max(hi, 0.0)
        assert abs(s - (hi0 - lo0) / 255.0) < 1e-15 and -128 <= z <= 127
        r = np.linspace(lo0, hi0, 1001)
        q = quant.quantize(r, s, z)
        assert np.array_equal(q, Q.quantize(r, s, z))
        assert q.min() >= -128 and q.max() <= 127
        assert np.abs(quant.dequantize(q, s, z) - r).max() <= 0.5 * s + 1e-12 + abs(quant.dequantize(z, s, z))
        assert quant.quantize(0.0, s, z) == z                     # zero is exactly representable
    assert quant.affine_params(0.0, 0.0) == (1.0, 0)


def test_per_channel_weights():
    rng = np.random.default_rng(0)
    w = rng.standard_normal((5, 7, 3, 3)) * rng.uniform(0.1, 3.0, size=(5, 1, 1, 1))
    w[2] = np.abs(w[2])                                            # a channel without negative weights
    wq, ws, wz = quant.quantize_weight_per_channel(w, axis=0)
    assert wq.dtype == np.int8 and wq.shape == w.shape and ws.shape == (5,) and wz.dtype == np.int32
    for o in range(5):
        s, z = Q.affine_params(w[o].min(), w[o].max())
        assert abs(ws[o] - s) < 1e-15 and wz[o] == z
        assert np.abs(Q.dequantize(wq[o], s, z) - w[o]).max() <= 0.5 * s + 1e-12
    wt = rng.standard_normal((6, 4, 2, 2))                          # ConvTranspose2d layout (I, O, 2, 2)
    wq, ws, wz = quant.quantize_weight_per_channel(wt, axis=1)
    assert ws.shape == (4,)
    for o in range(4):
        assert np.abs(Q.dequantize(wq[:, o], ws[o], wz[o]) - wt[:, o]).max() <= 0.5 * ws[o] + 1e-12


def test_integer_conv_oracle_is_exact():
    rng = np.random.default_rng(1)
    x = rng.integers(-128, 128, size=(2, 3, 5, 6), dtype=np.int8)
    w = rng.integers(-128, 128, size=(4, 3, 3, 3), dtype=np.int8)
    wz = rng.integers(-128, 128, size=4).astype(np.int32)
    xz = -37
    got = Q.conv3x3_acc(x, xz, w, wz)
    want = np.zeros((2, 4, 5, 6), dtype=np.int64)
    for n in range(2):
        for o in range(4):
            for y in range(5):
                for xx in range(6):
                    s = 0
                    for c in range(3):
                        for ky in range(3):
                            for kx in range(3):
                                yy, xc = y + ky - 1, xx + kx - 1
                                if 0 <= yy < 5 and 0 <= xc < 6:
                                    s += (int(x[n, c, yy, xc]) - xz) * (int(w[o, c, ky, kx]) - int(wz[o]))
                    want[n, o, y, xx] = s
    assert np.array_equal(got, want)
    wt = rng.integers(-128, 128, size=(3, 2, 2, 2), dtype=np.int8)
    wtz = np.array([5, -9], dtype=np.int32)
    up = Q.upconv2x2_acc(x, xz, wt, wtz)
    for n, o, y, xx, a, b in [(0, 0, 0, 0, 0, 0), (1, 1, 4, 5, 1, 1), (0, 1, 2, 3, 1, 0)]:
        s = sum((int(x[n, c, y, xx]) - xz) * (int(wt[c, o, a, b]) - int(wtz[o])) for c in range(3))
        assert up[n, o, 2 * y + a, 2 * xx + b] == s
    # requantisation: one float32 multiply, round half to even, clamp; ReLU = lower clamp at the zero point
    acc = np.array([[[[-300, 250, 1000, 40000, 6]]]], dtype=np.int64)
    y = Q.requantize(acc, np.array([0]), np.array([0.25], dtype=np.float32), -20, True)
    assert y.tolist() == [[[[-20, 42, 127, 127, -18]]]]              # 62.5 -> 62 (even), 1.5 -> 2


def _model_b(seed=0, size=64, n=3):
    sdn = S.seeded_state_dict(FEATS_B, seed=seed)
    frames = S.synthetic_frames(n, size, size, seed=seed + 1)
    ranges = Q.float_ranges(sdn, frames)
    return sdn, frames, ranges, quant.quantize_model(sdn, ranges)


def test_quantised_model_b_tracks_the_float_model():
    sdn, frames, ranges, qm = _model_b()
    assert set(quant.tensor_names(3)) == set(ranges)
    assert qm["encoder_blocks.0.0.w_q"].shape == (32, 3, 3, 3) and qm["decoder_blocks.0.w_q"].shape == (256, 128, 2, 2)
    # concat halves share one (scale, zero point): skip producer and transposed conv requantise into the same tensor
    assert qm["encoder_blocks.2.3.y_zp"] == qm["decoder_blocks.0.y_zp"]
    assert qm["encoder_blocks.2.3.y_scale"] == qm["decoder_blocks.0.y_scale"]
    taps = {}
    logits = Q.forward(qm, frames, taps=taps)
    assert logits.shape == (3, 1, 64, 64) and logits.dtype == np.float32
    with torch.no_grad():
        ref = O.forward(O.to_torch_state(sdn), O.normalize_u8_nhwc(frames)).numpy()
    err = np.abs(logits - ref)
    span = float(ref.max() - ref.min())
    print("int8 vs float logits: max %.3f mean %.4f of span %.2f" % (err.max(), err.mean(), span))
    assert err.mean() < 0.05 * span and err.max() < 0.5 * span
    assert O.mask_iou(logits > 0, ref > 0) > 0.85
    # every activation tensor really uses its int8 range (a scale / zero-point mix-up collapses it)
    for k, v in taps.items():
        assert int(v.max()) - int(v.min()) > 60, k


def test_quantised_model_roundtrip(tmp_path):
    _, frames, _, qm = _model_b(seed=3, size=32, n=1)
    p = tmp_path / "model_b_int8.npz"
    quant.save_quantized(p, qm)
    qm2 = quant.load_quantized(p)
    assert set(qm2) == set(qm)
    assert np.array_equal(Q.forward(qm, frames), Q.forward(qm2, frames))
