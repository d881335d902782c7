"""Range safety of the split-operand (f16x3) tier.

The tier stores activations as fp16 hi + lo planes (|v| <= 65504, 22 significant bits only while the lo part is a normal
fp16 number); the reference's network is plain fp32 (README.md:1449-1458) and has no such limits.  The tier therefore
(a) stores every activation scaled by a per-channel power of two derived from the BatchNorm parameters
(csrc/unet_x3.inc, ActScale), and (b) reports an activation that still leaves the range (UNET_ERR_RANGE from
unet_device_error) so that the caller re-runs the frames on the exact-fp32 tier instead of receiving clamped results.
These tests hold both to the fp32 acceptance (tests/test_x3_gpu.py: logits within 2e-4 of the oracle, masks identical
off ties) on checkpoints and frames chosen to stress the range: BatchNorm gammas spread over eight decades per channel,
channels with gamma = 0, all-0 and all-255 frames, and one checkpoint that does leave the range."""
import copy
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O
from unet_lane_detection_amd import _lib
from unet_lane_detection_amd import state as S

pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-4


def _frame(golden_dir):
    return np.fromfile(os.path.join(golden_dir, "frame_001410_rgb_u8.bin"), dtype=np.uint8).reshape(1, 224, 224, 3)


def _oracle_logits(sd, frames_u8):
    with torch.no_grad():
        return O.forward(O.to_torch_state(sd), O.normalize_u8_nhwc(frames_u8)).numpy()[:, 0]


def _conv1_blocks(features):
    """(prefix of a DoubleConv) for every block: its first BatchNorm is '<prefix>.1', the convolution that reads that
    activation '<prefix>.3'"""
    d = len(features)
    return ["encoder_blocks.%d" % l for l in range(d)] + ["bottleneck"] + ["decoder_blocks.%d" % (2 * j + 1) for j in range(d)]


def _spread_gammas(sd, rng, lo=-4.0, hi=4.0):
    """Every DoubleConv's inner activation scaled per channel by s = 10^U(lo, hi) (BatchNorm gamma and beta times s)
    with the next convolution's input channels divided by s: ReLU is positively homogeneous, so the network computes
    the same function while that activation spans eight decades across its channels."""
    sd = copy.deepcopy(sd)
    for p in _conv1_blocks(O.infer_features(sd)):
        c = sd[p + ".1.weight"].shape[0]
        s = (10.0 ** rng.uniform(lo, hi, c)).astype(np.float32)
        sd[p + ".1.weight"] = sd[p + ".1.weight"] * s
        sd[p + ".1.bias"] = sd[p + ".1.bias"] * s
        sd[p + ".3.weight"] = (sd[p + ".3.weight"] / s[None, :, None, None]).astype(np.float32)
    return sd


def _check(model, sd, frames, tag):
    ref = _oracle_logits(sd, frames)
    logits, mask = model.run_u8(torch.from_numpy(frames).cuda(), return_mask=True, precision="f16x3")
    assert model.device_error() == 0
    got = logits.cpu().numpy()[:, 0]
    err = np.abs(got - ref).max()
    print("%s: f16x3 max |dlogit| %.3e (logits in [%.2f, %.2f])" % (tag, err, ref.min(), ref.max()))
    assert err < LOGIT_TOL * max(1.0, np.abs(ref).max() / 8.0), err
    sure = np.abs(ref) > LOGIT_TOL * max(1.0, np.abs(ref).max() / 8.0)
    assert np.array_equal(mask.cpu().numpy()[sure] > 0, ref[sure] > 0)


def test_x3_gammas_over_eight_decades(golden_dir):
    from unet_lane_detection_amd.model import UNetHIP
    sd = _spread_gammas(S.seeded_state_dict(seed=0), np.random.default_rng(11))
    m = UNetHIP(sd, device=0)
    try:
        _check(m, sd, _frame(golden_dir), "gammas 1e-4..1e4")
    finally:
        m.release()


def test_x3_constant_frames_and_dead_channels(golden_dir):
    from unet_lane_detection_amd.model import UNetHIP
    sd = S.seeded_state_dict(seed=0)
    rng = np.random.default_rng(5)
    for p in _conv1_blocks(O.infer_features(sd)):      # a quarter of the channels: gamma = 0 (a constant beta after ReLU),
        c = sd[p + ".1.weight"].shape[0]               # some of those with beta = 0 too (dead)
        idx = rng.permutation(c)[: c // 4]
        sd[p + ".1.weight"][idx] = 0.0
        sd[p + ".1.bias"][idx[: len(idx) // 2]] = 0.0
    m = UNetHIP(sd, device=0)
    try:
        _check(m, sd, _frame(golden_dir), "gamma = 0 channels")
        _check(m, sd, np.zeros((1, 224, 224, 3), np.uint8), "all-0 frame")
        _check(m, sd, np.full((1, 224, 224, 3), 255, np.uint8), "all-255 frame")
    finally:
        m.release()


def _out_of_range_checkpoint():
    """The first BatchNorm's running variance is 1e-14 of what its input really has: xhat is ~1e6, far beyond what
    gamma and beta predict, and the activation leaves the fp16 range whatever power of two it was stored with."""
    sd = S.seeded_state_dict(seed=0)
    sd["encoder_blocks.0.1.running_var"] = np.full_like(sd["encoder_blocks.0.1.running_var"], 1e-14)
    # the second convolution's weights absorb the scale so that the fp32 network stays finite and well conditioned
    sd["encoder_blocks.0.3.weight"] = (sd["encoder_blocks.0.3.weight"] * 1e-6).astype(np.float32)
    return sd


def test_x3_out_of_range_is_reported_not_returned(golden_dir):
    from unet_lane_detection_amd.model import UNetHIP
    sd = _out_of_range_checkpoint()
    frames = _frame(golden_dir)
    m = UNetHIP(sd, device=0)
    try:
        m.run_u8(torch.from_numpy(frames).cuda(), precision="f16x3")
        assert m.device_error() == _lib.UNET_ERR_RANGE
        assert m.device_error() == 0                       # reported once
        logits = m.run_u8(torch.from_numpy(frames).cuda(), precision="fp32")
        assert m.device_error() == 0
        ref = _oracle_logits(sd, frames)
        err = np.abs(logits.cpu().numpy()[:, 0] - ref).max()
        assert err < LOGIT_TOL * max(1.0, np.abs(ref).max() / 8.0), err
    finally:
        m.release()


def test_container_falls_back_to_fp32_or_raises(golden_dir, monkeypatch, tmp_path):
    """Through the reference's container interface (src/py_utils/rknn_executor.py:26-38): under the automatic tier the
    call is served by the fp32 tier (same probabilities as the oracle), with the tier forced it raises - it never
    returns clamped results with a success status."""
    from unet_lane_detection_amd.py_utils.rknn_executor import RKNN_model_container
    sd = _out_of_range_checkpoint()
    path = os.path.join(tmp_path, "oor.npz")
    np.savez(path, **sd)
    frames = _frame(golden_dir)
    ref = 1.0 / (1.0 + np.exp(-_oracle_logits(sd, frames).astype(np.float64)))
    monkeypatch.delenv("UNET_HIP_TIER", raising=False)
    c = RKNN_model_container(path, "rk3588", "0")
    assert c.precision == "f16x3"
    out = c.run([frames])
    assert c.precision == "fp32"
    assert np.abs(out[0][:, 0] - ref).max() < 1e-4
    out2 = c.run([frames])                                  # stays on fp32, graph path included
    assert np.abs(out2[0][:, 0] - ref).max() < 1e-4
    c.release()
    monkeypatch.setenv("UNET_HIP_TIER", "f16x3")
    c = RKNN_model_container(path, "rk3588", "0")
    with pytest.raises(RuntimeError):
        c.run([frames])
    c.release()


def test_error_block_is_reported_once_and_calls_recover(golden_dir):
    """A kernel-side failure record fails ONE call (the next entry point, or unet_device_error) and is cleared with
    the report: later frames are served normally (the record used to stay set for ever when a forward
    itself returned it)."""
    from unet_lane_detection_amd.model import UNetHIP
    m = UNetHIP(S.seeded_state_dict(seed=0), device=0)
    frames = torch.from_numpy(_frame(golden_dir)).cuda()
    try:
        a = m.run_u8(frames, precision="f16x3")
        assert m.device_error() == 0
        assert m._lib.unet_debug_set_error_block(m._h, 0, 1) == 0
        with pytest.raises(_lib.UnetError):
            m.run_u8(frames, precision="f16x3")            # the forward reports the record ...
        b = m.run_u8(frames, precision="f16x3")            # ... once
        assert m.device_error() == 0
        assert torch.equal(a, b)
        assert m._lib.unet_debug_set_error_block(m._h, 1, 1) == 0
        assert m.device_error() == _lib.UNET_ERR_RANGE
        assert m.device_error() == 0
    finally:
        m.release()


def test_container_fallback_keeps_graphs_and_workspaces_apart(golden_dir, monkeypatch):
    """the tiers have separate workspaces in the library, each growing with the largest batch IT has seen.  After
    a range fallback the container's workspace high-water mark (taken on the f16x3 tier) must not vouch for the fp32
    tier's workspace: 8 frames on f16x3, a forced range report, then 1-frame and 2-frame calls alternating - the 2-frame
    call re-allocates the fp32 workspace and the 1-frame graph captured before it pointed into the freed one.  Every call
    must return what the direct (graph-free) path returns."""
    from unet_lane_detection_amd.py_utils.rknn_executor import RKNN_model_container
    monkeypatch.delenv("UNET_HIP_TIER", raising=False)
    monkeypatch.setenv("UNET_HIP_RANGE_RETRY", "0")          # stay on the fp32 tier
    one = _frame(golden_dir)
    frames8 = np.concatenate([np.roll(one, k * 7, axis=2) for k in range(8)], axis=0)
    c = RKNN_model_container("seed:0", "rk3588", "0")
    monkeypatch.setenv("UNET_HIP_GRAPH", "0")
    d = RKNN_model_container("seed:0", "rk3588", "0")         # the direct path, fp32 tier throughout
    d.precision, d._auto_tier = "fp32", False
    try:
        assert c.precision == "f16x3"
        c.run([frames8])                                      # the f16x3 workspace now holds 8 frames
        assert c.model._lib.unet_debug_set_error_block(c.model._h, 1, 1) == 0
        out = c.run([frames8[:1]])                            # range report -> served by the fp32 tier
        assert c.precision == "fp32" and c.range_fallbacks == 1
        assert np.array_equal(out[0], d.run([frames8[:1]])[0])
        for k in range(3):
            a1 = c.run([frames8[:1]])[0]                      # graph of the 1-frame shape
            a2 = c.run([frames8[2:4]])[0]                     # a larger batch: the fp32 workspace grows
            a3 = c.run([frames8[1:2]])[0]                     # the 1-frame graph again
            assert np.array_equal(a1, d.run([frames8[:1]])[0]), k
            assert np.array_equal(a2, d.run([frames8[2:4]])[0]), k
            assert np.array_equal(a3, d.run([frames8[1:2]])[0]), k
    finally:
        c.release()
        d.release()


def test_container_returns_to_f16x3_after_clean_frames(golden_dir, monkeypatch):
    """One out-of-range frame must not cost the node its frame rate for good (the fp32 tier is 2.1x slower): after
    UNET_HIP_RANGE_RETRY clean frames on the fp32 tier the container tries the f16x3 tier again; a second report doubles
    the wait."""
    from unet_lane_detection_amd.py_utils.rknn_executor import RKNN_model_container
    monkeypatch.delenv("UNET_HIP_TIER", raising=False)
    monkeypatch.setenv("UNET_HIP_RANGE_RETRY", "3")
    one = _frame(golden_dir)
    c = RKNN_model_container("seed:0", "rk3588", "0")
    try:
        base = c.run([one])[0]
        assert c.precision == "f16x3"
        assert c.model._lib.unet_debug_set_error_block(c.model._h, 1, 1) == 0
        out = c.run([one])[0]                                 # report: this frame comes from the fp32 tier
        assert c.precision == "fp32" and c.range_fallbacks == 1
        assert np.abs(out - base).max() < 1e-4
        c.run([one]); c.run([one])                            # frames 2 and 3 of the wait (the re-run counted as 1)
        assert c.precision == "f16x3"                          # back
        again = c.run([one])[0]
        assert c.precision == "f16x3" and np.array_equal(again, base)
        assert c.model._lib.unet_debug_set_error_block(c.model._h, 1, 1) == 0
        c.run([one])                                          # second report: the wait is now 6 frames
        assert c.precision == "fp32" and c.range_fallbacks == 2
        for _ in range(4):
            c.run([one])
        assert c.precision == "fp32"
        c.run([one])
        assert c.precision == "f16x3"
    finally:
        c.release()
