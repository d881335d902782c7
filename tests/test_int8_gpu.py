"""int8 tier of model B on the GPU: bit-exact against the integer oracle
(oracle/int8_oracle.py) - logits, masks and every intermediate int8 tensor - and the device calibration pass against
the float oracle's activation ranges.  Parity with the Rockchip runtime is unpinned (see the oracle's header)."""
import numpy as np
import pytest
import torch

from oracle import int8_oracle as Q
from oracle import unet_oracle as O
from unet_lane_detection_amd import quant, state as S

pytestmark = pytest.mark.gpu

FEATS_B = [32, 64, 128]


@pytest.fixture(scope="module")
def model_b():
    """Float model B (seeded weights), its HIP fp32 handle, calibrated ranges and the quantised model."""
    from unet_lane_detection_amd.int8 import calibrate
    from unet_lane_detection_amd.model import UNetHIP
    sdn = S.seeded_state_dict(FEATS_B, seed=0)
    fm = UNetHIP(sdn, device=0)
    calib = S.synthetic_frames(6, 64, 64, seed=1)
    ranges = calibrate(fm, torch.from_numpy(calib), batch=4)
    yield sdn, fm, calib, ranges, quant.quantize_model(sdn, ranges)
    fm.release()


def test_device_calibration_matches_float_oracle(model_b):
    sdn, fm, calib, ranges, _ = model_b
    ref = Q.float_ranges(sdn, calib)
    assert set(ranges) == set(ref)
    for k in ref:
        span = ref[k][1] - ref[k][0]
        assert abs(ranges[k][0] - ref[k][0]) < 1e-4 * span + 1e-5, (k, ranges[k], ref[k])
        assert abs(ranges[k][1] - ref[k][1]) < 1e-4 * span + 1e-5, (k, ranges[k], ref[k])


TENSOR_OF_UNIT = {"encoder_blocks.%d.0": "enc%d.a", "decoder_blocks.%d.0": None}


# the last two: batches with thousands of pixel tiles per layer (ragged tile grid in the second)
@pytest.mark.parametrize("n,h,w,seed", [(2, 64, 64, 3), (3, 40, 72, 4), (1, 224, 224, 5), (5, 8, 8, 6),
                                        (6, 224, 224, 7), (6, 200, 232, 8)])
def test_int8_forward_is_bit_exact(model_b, n, h, w, seed):
    from unet_lane_detection_amd.int8 import UNetInt8
    _, _, _, _, qm = model_b
    net = UNetInt8(qm, device=0)
    frames = S.synthetic_frames(n, h, w, seed=seed)
    taps = {}
    ref = Q.forward(qm, frames, taps=taps)
    logits, probs, mask = net.run_u8(torch.from_numpy(frames).cuda(), return_probs=True, return_mask=True)
    torch.cuda.synchronize()
    d = len(FEATS_B)
    # every intermediate tensor, in forward order: the first mismatch names the failing layer
    checks = [("im2col", None, 0)]
    for l in range(d):
        checks += [(f"enc{l}.a", taps[f"encoder_blocks.{l}.0"], l), (f"cat{l}.pool", Q.maxpool2x2(taps[f"encoder_blocks.{l}.3"]), l + 1)]
    checks += [("bott.a", taps["bottleneck.0"], d), ("bott.b", taps["bottleneck.3"], d)]
    for j in range(d):
        l = d - 1 - j
        cat = np.concatenate([taps[f"encoder_blocks.{l}.3"], taps[f"decoder_blocks.{2 * j}"]], axis=1)
        checks += [(f"cat{l}", cat, l), (f"dec{j}.a", taps[f"decoder_blocks.{2 * j + 1}.0"], l),
                   (f"dec{j}.b", taps[f"decoder_blocks.{2 * j + 1}.3"], l)]
    for name, want, level in checks:
        got = net.read_tensor(name, n, h >> level, w >> level)
        if want is None:      # im2col rows: centre tap (k = 12, 13, 14) is the quantised input itself
            assert np.array_equal(got[:, 12:15], taps["input"]), name
            continue
        assert got.shape == want.shape, (name, got.shape, want.shape)
        bad = int((got != want).sum())
        assert bad == 0, f"{name}: {bad} of {want.size} int8 values differ (max |d| {np.abs(got.astype(int) - want.astype(int)).max()})"
    lg = logits.cpu().numpy()
    assert np.array_equal(lg, ref), np.abs(lg - ref).max()
    assert np.array_equal(mask.cpu().numpy(), ((ref[:, 0] > 0) * 255).astype(np.uint8))
    assert np.abs(probs.cpu().numpy() - 1 / (1 + np.exp(-ref.astype(np.float64)))).max() < 1e-6
    net.release()


def test_int8_tracks_the_float_tier(model_b):
    from unet_lane_detection_amd.int8 import UNetInt8
    _, fm, _, _, qm = model_b
    net = UNetInt8(qm, device=0)
    frames = torch.from_numpy(S.synthetic_frames(4, 64, 64, seed=9)).cuda()
    lf, mf = fm.run_u8(frames, return_mask=True)
    lq, mq = net.run_u8(frames, return_mask=True)
    err = (lf - lq).abs()
    span = (lf.max() - lf.min()).item()
    print("int8 vs fp32 tier: max %.3f mean %.4f of logit span %.2f, mask IoU %.4f"
          % (err.max().item(), err.mean().item(), span, O.mask_iou(mq.cpu().numpy(), mf.cpu().numpy())))
    assert err.mean().item() < 0.05 * span
    assert O.mask_iou(mq.cpu().numpy(), mf.cpu().numpy()) > 0.85
    net.release()


def test_int8_model_file_roundtrip_and_errors(model_b, tmp_path):
    from unet_lane_detection_amd import _lib
    from unet_lane_detection_amd.int8 import UNetInt8
    _, _, _, _, qm = model_b
    p = tmp_path / "lane_unet_int8.npz"
    quant.save_quantized(p, qm)
    net = UNetInt8.from_file(p, device=0)
    frames = torch.from_numpy(S.synthetic_frames(1, 32, 32, seed=2)).cuda()
    a = net.run_u8(frames)
    assert np.array_equal(a.cpu().numpy(), Q.forward(qm, frames.cpu().numpy()))
    with pytest.raises(_lib.UnetError):
        net.run_u8(torch.zeros((1, 20, 24, 3), dtype=torch.uint8).cuda())      # 20 is not a multiple of 8
    net.release()
    broken = dict(qm)
    del broken["bottleneck.3.w_q"]
    with pytest.raises(_lib.UnetError):
        UNetInt8(broken, device=0)


def test_container_runs_a_quantised_model_file(model_b, tmp_path):
    """The drop-in container (reference src/py_utils/rknn_executor.py:4-42) given a quantised model file - the
    counterpart of the int8 .rknn blob the reference loads - runs the int8 tier and keeps the container contract."""
    from unet_lane_detection_amd.py_utils.rknn_executor import RKNN_model_container
    _, _, _, _, qm = model_b
    p = tmp_path / "lane_unet_int8.npz"
    quant.save_quantized(p, qm)
    frame = S.synthetic_frames(1, 224, 224, seed=11)
    c = RKNN_model_container(str(p), "rk3588", "0")
    assert c.precision == "int8"
    out = c.run(inputs=[frame])
    assert isinstance(out, list) and out[0].shape == (1, 1, 224, 224) and out[0].dtype == np.float32
    ref = Q.forward(qm, frame)
    assert np.abs(out[0] - 1 / (1 + np.exp(-ref.astype(np.float64)))).max() < 1e-6
    assert np.array_equal(O.postprocess_output(out), ((ref[0, 0] > 0) * 255).astype(np.uint8))
    c.release()
    assert c.run([frame]) == []
    c.release()
