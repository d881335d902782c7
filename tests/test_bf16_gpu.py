"""bf16 tier (BASELINE.json configs[2]): bf16 storage, fp32 accumulate.  A separate accuracy tier - the fp32
1e-3 logit bound does not apply (SURVEY.md section 7 'Hard parts'); the bounds asserted here are measured
bands with head-room, stated so that a regression in the kernel (wrong tap, wrong channel order) cannot
hide inside them: a layout bug produces O(1) logit errors, bf16 rounding through 23 layers ~1e-2."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O
from unet_lane_detection_amd import state as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def modelA():
    from unet_lane_detection_amd.model import UNetHIP
    m = UNetHIP(S.seeded_state_dict(seed=0), device=0)
    yield m
    m.release()


def test_bf16_reference_frame(modelA, golden_dir):
    g = np.load(os.path.join(golden_dir, "modelA_frame_001410.npz"))
    frame = np.fromfile(os.path.join(golden_dir, "frame_001410_rgb_u8.bin"), dtype=np.uint8).reshape(1, 224, 224, 3)
    logits, mask = modelA.run_u8(torch.from_numpy(frame).cuda(), return_mask=True, precision="bf16")
    lg = logits.cpu().numpy()[0, 0]
    err = np.abs(lg - g["logits"])
    print("bf16 logit error: max %.4f mean %.5f ; logit std %.3f" % (err.max(), err.mean(), g["logits"].std()))
    assert err.max() < 0.35 and err.mean() < 0.03
    iou = O.mask_iou(mask.cpu().numpy()[0], g["mask"])
    print("bf16 mask IoU vs fp32 reference mask: %.5f" % iou)
    assert iou > 0.985
    sure = np.abs(g["logits"]) > 0.35
    assert np.array_equal(mask.cpu().numpy()[0][sure], g["mask"][sure])


def test_bf16_batch_and_fp32_agree(modelA):
    frames = torch.from_numpy(S.synthetic_frames(3, seed=12)).cuda()
    a = modelA.run_u8(frames, precision="fp32")
    b = modelA.run_u8(frames, precision="bf16")
    d = (a - b).abs()
    assert d.max().item() < 0.6 and d.mean().item() < 0.05
    # relative L2 error of the logit field
    assert (d.pow(2).sum().sqrt() / a.pow(2).sum().sqrt()).item() < 0.02


def test_bf16_other_shape():
    from unet_lane_detection_amd.model import UNetHIP
    feats = [32, 64]
    sdn = S.seeded_state_dict(feats, seed=3)
    m = UNetHIP(sdn, device=0)
    frames = S.synthetic_frames(5, 40, 56, seed=1)
    with torch.no_grad():
        ref = O.forward(O.to_torch_state(sdn), O.normalize_u8_nhwc(frames))
    got = m.run_u8(torch.from_numpy(frames).cuda(), precision="bf16").cpu()
    d = (got - ref).abs()
    assert d.max().item() < 0.3 and d.mean().item() < 0.03
    m.release()
