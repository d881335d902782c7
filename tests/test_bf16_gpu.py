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
    assert err.max() < 0.3 and err.mean() < 0.033       # measured 0.143 / 0.0165: twice that
    iou = O.mask_iou(mask.cpu().numpy()[0], g["mask"])
    print("bf16 mask IoU vs fp32 reference mask: %.5f" % iou)
    assert iou > 0.985
    sure = np.abs(g["logits"]) > 0.3
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


def _with_persistent(mode, fn):
    from unet_lane_detection_amd import _lib
    lib = _lib.load()
    prev = lib.unet_set_bf16_persistent(mode)
    try:
        return fn()
    finally:
        lib.unet_set_bf16_persistent(prev)


def test_bf16_persistent_kernel_batch(modelA):
    """The persistent wave-specialised kernel (csrc/conv_bf16_ws.h; levels 0 and 1 when forced on) against the
    2x2-wave kernel on the same frames: the two differ only in fp32 summation order inside a layer, i.e. in
    which activations land on the other side of a bf16 rounding boundary (in practice both accumulate chunk by
    chunk, tap by tap, and come out bit-identical).  A halo / border / channel-order bug gives O(1) differences."""
    frames = torch.from_numpy(S.synthetic_frames(12, seed=21)).cuda()
    ref = modelA.run_u8(frames, precision="fp32")
    base, mbase = _with_persistent(0, lambda: modelA.run_u8(frames, return_mask=True, precision="bf16"))
    for mode, launches in ((-1, 3), (1, 7)):
        modelA.profile(True)
        got, mgot = _with_persistent(mode, lambda: modelA.run_u8(frames, return_mask=True, precision="bf16"))
        names = [r[0] for r in modelA.profile_records()]
        modelA.profile(False)
        # automatic: the three 64-channel level-0 layers; forced: level 1 (112x112) as well
        assert names.count("conv3x3_ws_bf16") == launches, names
        # transposed convolutions: the wave-specialised kernel for all four when forced (mode 1); automatically the
        # one-wave-per-SIMD kernel (csrc/upconv_bf16_r512.h) where there is a work item for half of the CUs - three of
        # the four at batch 12
        assert names.count("upconv2x2_ws_bf16") == (4 if mode == 1 else 0), names
        assert names.count("upconv2x2_r512_bf16") == (0 if mode == 1 else 3), names
        d = (got - base).abs()
        print("persistent mode %d vs 2x2-wave kernel: max %.4f mean %.5f" % (mode, d.max().item(), d.mean().item()))
        assert d.max().item() < 0.3 and d.mean().item() < 0.02
        assert O.mask_iou(mgot.cpu().numpy(), mbase.cpu().numpy()) > 0.99
        e = (got - ref).abs()
        assert e.max().item() < 0.6 and e.mean().item() < 0.05
        # borders carry the zero-page halo: compare them separately
        for sl in (np.s_[:, :, 0, :], np.s_[:, :, -1, :], np.s_[:, :, :, 0], np.s_[:, :, :, -1]):
            assert (got[sl] - base[sl]).abs().max().item() < 0.3


def test_bf16_persistent_kernel_partial_tiles():
    """Width 240 = 7.5 tiles of 32 columns, height 208 = 13 tile rows: partial tiles and per-image borders of the
    persistent kernel, against the CPU oracle."""
    from unet_lane_detection_amd.model import UNetHIP
    sdn = S.seeded_state_dict(seed=0)
    m = UNetHIP(sdn, device=0)
    frames = S.synthetic_frames(3, 208, 240, seed=5)
    with torch.no_grad():
        ref = O.forward(O.to_torch_state(sdn), O.normalize_u8_nhwc(frames))
    fr = torch.from_numpy(frames).cuda()
    base = _with_persistent(0, lambda: m.run_u8(fr, precision="bf16")).cpu()
    m.profile(True)
    got = _with_persistent(1, lambda: m.run_u8(fr, precision="bf16")).cpu()
    assert [r[0] for r in m.profile_records()].count("conv3x3_ws_bf16") == 3
    m.profile(False)
    d = (got - base).abs()
    e = (got - ref).abs()
    print("partial tiles: vs 2x2-wave max %.4f mean %.5f ; vs oracle max %.4f mean %.5f"
          % (d.max().item(), d.mean().item(), e.max().item(), e.mean().item()))
    assert d.max().item() < 0.3 and d.mean().item() < 0.02
    assert e.max().item() < 0.6 and e.mean().item() < 0.05
    m.release()


def test_bf16_r512_kernel_is_bit_identical(modelA):
    """The one-wave-per-SIMD kernel (csrc/conv_bf16_r512.h: 8 x 28 / 16 x 14 tiles, weights straight from L2) forced
    onto every layer it supports (14 of 18 at 224 x 224: the 112 x 112 ... 14 x 14 levels) against the 2x2-wave
    kernel: same chunk / tap accumulation order, so the same logits bit for bit; a batch of 5 leaves the tall-image
    tiling of the 28 x 28 and 14 x 14 levels with a partial last tile."""
    frames = torch.from_numpy(S.synthetic_frames(5, seed=33)).cuda()
    base = _with_persistent(0, lambda: modelA.run_u8(frames, precision="bf16"))
    modelA.profile(True)
    got = _with_persistent(2, lambda: modelA.run_u8(frames, precision="bf16"))
    names = [r[0] for r in modelA.profile_records()]
    modelA.profile(False)
    assert names.count("conv3x3_r512_bf16") == 14, names
    assert names.count("upconv2x2_r512_bf16") == 4, names          # ... and the four transposed convolutions with it
    assert torch.equal(got, base)
    auto = _with_persistent(-1, lambda: modelA.run_u8(frames, precision="bf16"))
    assert torch.equal(auto, base)
