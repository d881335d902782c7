"""Split-operand (f16x3) tier: fp16 hi + lo operands, three MFMAs per product (csrc/conv_x3_ws.h).

It claims the accuracy class of the exact-fp32 tier, so it is held to the fp32 acceptance (BASELINE.json north_star:
logits within 1e-3, asserted at 2e-4; mask identical wherever |logit| exceeds that; IoU >= 1 - 1e-4) against the
reference's golden vectors, plus per-operator checks against the CPU oracle on every tile shape and border case."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O
from unet_lane_detection_amd import state as S

pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-4


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.fixture(scope="module")
def lib():
    from unet_lane_detection_amd import _lib
    return _lib.load(build_if_missing=False)


@pytest.fixture(scope="module")
def modelA():
    from unet_lane_detection_amd.model import UNetHIP
    m = UNetHIP(S.seeded_state_dict(seed=0), device=0)
    yield m
    m.release()


# (n, cin, cout, h, w): full tiles, partial columns, rows past the image bottom, maps smaller than a tile,
# several channel tiles and chunk pairs, both tile shapes
CONV_CASES = [(2, 64, 64, 16, 32), (1, 64, 128, 8, 32), (3, 128, 64, 14, 14), (2, 64, 64, 28, 28),
              (1, 192, 64, 24, 40), (2, 64, 192, 10, 50), (1, 256, 256, 14, 14), (5, 64, 64, 6, 6),
              (1, 64, 64, 56, 56), (1, 128, 128, 18, 34)]


@pytest.mark.parametrize("tw", [0, 16, 32])
@pytest.mark.parametrize("n,cin,cout,h,w", CONV_CASES)
def test_conv3x3_x3_vs_oracle(lib, n, cin, cout, h, w, tw):
    g = torch.Generator().manual_seed(cin * 7 + cout + h + w)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.3
    for relu in (1, 0):
        ref = O.conv3x3(x, wt) * scale[None, :, None, None] + shift[None, :, None, None]
        if relu:
            ref = torch.relu(ref)
        xd = x.permute(0, 2, 3, 1).contiguous().cuda()
        y = torch.full((n, h, w, cout), float("nan"), device="cuda")
        rc = lib.unet_op_conv3x3_x3(0, _p(xd), n, h, w, cin, C.c_void_p(wt.numpy().ctypes.data),
                                    C.c_void_p(scale.numpy().ctypes.data), C.c_void_p(shift.numpy().ctypes.data),
                                    cout, relu, tw, _p(y), None, None)
        assert rc == 0
        got = y.cpu().permute(0, 3, 1, 2)
        err = (got - ref).abs().max().item()
        # the split keeps 22 bits per operand: the bound is the fp32 kernels' (2e-5 of the output range)
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (err, relu)


def _conv_x3(lib, xd, wt, scale, shift, n, h, w, cin, cout, relu, tw, pool=False):
    y = torch.full((n, h, w, cout), float("nan"), device="cuda")
    yp = torch.full((n, h // 2, w // 2, cout), float("nan"), device="cuda") if pool else None
    rc = lib.unet_op_conv3x3_x3(0, _p(xd), n, h, w, cin, C.c_void_p(wt.numpy().ctypes.data),
                                C.c_void_p(scale.numpy().ctypes.data), C.c_void_p(shift.numpy().ctypes.data),
                                cout, relu, tw, _p(y), _p(yp), None)
    assert rc == 0, rc
    return y, yp


# Second kernel structure (csrc/conv_x3_r512.h): 8 x 28 and 16 x 14 pixel tiles whose fragments straddle tile rows.
# (n, cin, cout, h, w, tw): whole tiles, rows past the image bottom, the batch tiled as one tall image (heights 28, 14,
# 20, 12 with several images, a last tile that ends inside an image), one and several channel groups, 2-wave and
# 4-wave channel layouts (tw + 200), a single chunk pair and many chunks
R512_CASES = [(2, 64, 256, 56, 56, 28), (3, 128, 512, 28, 28, 28), (5, 64, 256, 14, 14, 14), (1, 64, 256, 8, 28, 28),
              (2, 64, 128, 112, 112, 28), (3, 64, 128, 28, 28, 28), (7, 64, 384, 14, 14, 14), (1, 192, 256, 20, 84, 28),
              (3, 64, 256, 20, 28, 28), (2, 64, 256, 56, 56, 228), (3, 64, 512, 14, 14, 214), (1, 512, 256, 16, 28, 28),
              (4, 64, 256, 12, 28, 28), (1, 64, 128, 5, 28, 28), (3, 64, 128, 7, 14, 14),
              # the 7 x 32, 14 x 16 and 28 x 8 tiles (widths that are no multiple of 28: the 640 x 640 configuration's levels)
              (2, 64, 256, 21, 64, 332), (3, 128, 256, 10, 32, 332), (2, 64, 256, 28, 48, 316), (5, 64, 512, 9, 16, 316),
              (2, 64, 256, 56, 24, 308), (3, 64, 256, 20, 40, 308), (1, 64, 256, 3, 8, 308),
              # ... and the 7 x 32 tile in its two-wave form (Cout = 128 at the 320 x 320 level)
              (2, 64, 128, 21, 64, 532), (3, 128, 128, 10, 32, 532), (1, 64, 384, 14, 96, 532)]


@pytest.mark.parametrize("n,cin,cout,h,w,tw", R512_CASES)
def test_conv3x3_x3_r512_vs_oracle_and_first_structure(lib, n, cin, cout, h, w, tw):
    g = torch.Generator().manual_seed(cin * 5 + cout + h * 3 + w + n)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.3
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    for relu in (1, 0):
        ref = O.conv3x3(x, wt) * scale[None, :, None, None] + shift[None, :, None, None]
        if relu:
            ref = torch.relu(ref)
        y, _ = _conv_x3(lib, xd, wt, scale, shift, n, h, w, cin, cout, relu, tw)
        err = (y.cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (err, relu)
        # the same accumulation order as the first structure: bit for bit the same planes
        y1, _ = _conv_x3(lib, xd, wt, scale, shift, n, h, w, cin, cout, relu, 16 if (w == 14 or tw == 316) else 32)  # first structure
        assert torch.equal(y, y1)


@pytest.mark.parametrize("n,cin,cout,h,w,tw", [(2, 64, 256, 56, 56, 28), (3, 64, 128, 28, 28, 28),
                                                (4, 64, 256, 14, 14, 14)])
def test_conv3x3_x3_r512_pool(lib, n, cin, cout, h, w, tw):
    g = torch.Generator().manual_seed(h * w + cin + 1)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    scale, shift = torch.ones(cout), torch.zeros(cout)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    y, yp = _conv_x3(lib, xd, wt, scale, shift, n, h, w, cin, cout, 1, tw, pool=True)
    want = O.maxpool2x2(y.cpu().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert torch.equal(yp.cpu(), want)
    ref = torch.relu(O.conv3x3(x, wt))
    assert (y.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


# Third kernel structure (csrc/conv_x3_t448.h): 16 x 28 / 16 x 32 pixel tiles made of 4 x 4-pixel fragments, for the layers
# with 64 / 128 output channels.  (n, cin, cout, h, w, tw): whole tiles, rows past the image bottom (h = 40, 20, 6), maps of
# one tile column and of several, one and several channel groups, the one-wave (cout = 64, 192) and two-wave (cout = 128,
# 256) channel layouts, one chunk pair and many chunks, both tile widths
T448_CASES = [(2, 64, 64, 112, 112, 628), (1, 128, 64, 56, 56, 628), (2, 64, 64, 40, 84, 628), (2, 64, 128, 48, 56, 628),
              (3, 64, 256, 24, 56, 628), (5, 192, 128, 16, 28, 628), (1, 64, 192, 32, 28, 628), (3, 64, 64, 6, 28, 628),
              (2, 64, 64, 64, 64, 632), (2, 64, 128, 48, 96, 632), (1, 128, 64, 20, 32, 632), (1, 512, 64, 16, 32, 632),
              # the 256-channel form (tw 728): 8 x 28 tiles, four waves along the channels; heights that are no multiple of 8
              # with several images run as one tall image (28, 12, 20, 14: image boundaries inside a tile, at every row
              # of a 4 x 4 fragment block that can be one, and a last tile that ends inside an image)
              (2, 64, 256, 56, 56, 728), (3, 128, 512, 28, 28, 728), (1, 64, 256, 8, 28, 728), (4, 64, 256, 12, 28, 728),
              (5, 64, 256, 20, 56, 728), (2, 192, 256, 24, 84, 728), (3, 64, 256, 14, 28, 728), (1, 64, 512, 5, 28, 728),
              (6, 64, 256, 4, 28, 728)]


@pytest.mark.parametrize("n,cin,cout,h,w,tw", T448_CASES)
def test_conv3x3_x3_t448_vs_oracle_and_first_structure(lib, n, cin, cout, h, w, tw):
    g = torch.Generator().manual_seed(cin * 3 + cout + h * 5 + w + n)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.3
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    for relu in (1, 0):
        ref = O.conv3x3(x, wt) * scale[None, :, None, None] + shift[None, :, None, None]
        if relu:
            ref = torch.relu(ref)
        y, _ = _conv_x3(lib, xd, wt, scale, shift, n, h, w, cin, cout, relu, tw)
        err = (y.cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (err, relu)
        y1, _ = _conv_x3(lib, xd, wt, scale, shift, n, h, w, cin, cout, relu, 32)   # first structure
        assert torch.equal(y, y1)       # the same accumulation order: bit for bit the same planes


@pytest.mark.parametrize("n,cin,cout,h,w,tw", [(2, 64, 64, 112, 112, 628), (3, 64, 128, 48, 56, 628), (2, 128, 64, 40, 28, 628),
                                                (2, 64, 64, 48, 96, 632), (1, 64, 192, 22, 64, 632),
                                                (2, 64, 256, 56, 56, 728), (3, 64, 256, 28, 28, 728), (7, 64, 512, 28, 28, 728),
                                                (5, 128, 256, 12, 28, 728)])
def test_conv3x3_x3_t448_fused_pool(lib, n, cin, cout, h, w, tw):
    """EPI 1 of the third structure: both partners of a 2 x 2 window sit in one 4 x 4 fragment (DPP, no second fragment)."""
    g = torch.Generator().manual_seed(h * w + cin + 2)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    y, yp = _conv_x3(lib, xd, wt, scale, shift, n, h, w, cin, cout, 1, tw, pool=True)
    want = O.maxpool2x2(y.cpu().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert torch.equal(yp.cpu(), want)     # the pooled tensor is exactly the 2x2 max of the stored activation
    y1, yp1 = _conv_x3(lib, xd, wt, scale, shift, n, h, w, cin, cout, 1, 32, pool=True)   # first structure, fused too
    assert torch.equal(y, y1) and torch.equal(yp, yp1)
    ref = torch.relu(O.conv3x3(x, wt) * scale[None, :, None, None] + shift[None, :, None, None])
    assert (y.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("n,cin,h,w,tw", [(2, 64, 112, 112, 628), (1, 128, 40, 56, 628), (2, 64, 48, 64, 632), (3, 64, 6, 28, 628)])
def test_conv3x3_x3_head_fused_t448_vs_oracle_and_first_structure(lib, n, cin, h, w, tw):
    """The network's last two layers (reference README.md:1456-1457, :1447, :1481): 3x3 convolution to 64 channels + BN +
    ReLU with the 1x1 head in its epilogue, third structure against the oracle and bit for bit against the first."""
    g = torch.Generator().manual_seed(h + w + cin + 9)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(64, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    hw, hb = torch.randn(64, generator=g) * 0.2, 0.37
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    outs = []
    for t in (tw, 32):
        lg = torch.full((n, h, w), float("nan"), device="cuda")
        rc = lib.unet_op_conv3x3_x3_head(0, _p(xd), n, h, w, cin, C.c_void_p(wt.numpy().ctypes.data),
                                         C.c_void_p(scale.numpy().ctypes.data), C.c_void_p(shift.numpy().ctypes.data), 1, t,
                                         C.c_void_p(hw.numpy().ctypes.data), hb, _p(lg), None)
        assert rc == 0, rc
        outs.append(lg)
    assert torch.equal(outs[0], outs[1])
    act = torch.relu(O.conv3x3(x, wt) * scale[None, :, None, None] + shift[None, :, None, None])
    ref = (act * hw[None, :, None, None]).sum(1) + hb
    assert (outs[0].cpu() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def test_conv3x3_x3_t448_rejects_unsupported_shapes(lib):
    x = torch.zeros(1, 8, 40, 64, device="cuda")
    wt, scale, shift = torch.zeros(64, 64, 3, 3), torch.ones(64), torch.zeros(64)
    y = torch.zeros(1, 8, 40, 64, device="cuda")
    rc = lib.unet_op_conv3x3_x3(0, _p(x), 1, 8, 40, 64, C.c_void_p(wt.numpy().ctypes.data),
                                C.c_void_p(scale.numpy().ctypes.data), C.c_void_p(shift.numpy().ctypes.data),
                                64, 1, 628, _p(y), None, None)
    assert rc != 0   # W = 40 is neither a multiple of 28 nor of 32


def test_conv3x3_x3_r512_rejects_unsupported_shapes(lib):
    x = torch.zeros(1, 8, 32, 64, device="cuda")
    wt, scale, shift = torch.zeros(128, 64, 3, 3), torch.ones(128), torch.zeros(128)
    y = torch.zeros(1, 8, 32, 128, device="cuda")
    rc = lib.unet_op_conv3x3_x3(0, _p(x), 1, 8, 32, 64, C.c_void_p(wt.numpy().ctypes.data),
                                C.c_void_p(scale.numpy().ctypes.data), C.c_void_p(shift.numpy().ctypes.data),
                                128, 1, 28, _p(y), None, None)
    assert rc != 0   # W = 32 is not a multiple of 28


@pytest.mark.parametrize("tw", [16, 32])
@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 64, 16, 32), (1, 64, 128, 28, 28), (3, 64, 64, 14, 14),
                                             (1, 128, 64, 20, 36)])
def test_conv3x3_x3_fused_pool(lib, n, cin, cout, h, w, tw):
    g = torch.Generator().manual_seed(h * w + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    scale, shift = torch.ones(cout), torch.zeros(cout)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    y = torch.full((n, h, w, cout), float("nan"), device="cuda")
    yp = torch.full((n, h // 2, w // 2, cout), float("nan"), device="cuda")
    rc = lib.unet_op_conv3x3_x3(0, _p(xd), n, h, w, cin, C.c_void_p(wt.numpy().ctypes.data),
                                C.c_void_p(scale.numpy().ctypes.data), C.c_void_p(shift.numpy().ctypes.data),
                                cout, 1, tw, _p(y), _p(yp), None)
    assert rc == 0
    # the pooled tensor is exactly the 2x2 max of the stored activation (the split is monotonic)
    want = O.maxpool2x2(y.cpu().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert torch.equal(yp.cpu(), want)
    ref = torch.relu(O.conv3x3(x, wt))
    assert (y.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 128, 64, 7, 7), (1, 1024, 512, 14, 14), (3, 64, 64, 5, 9),
                                             (1, 256, 128, 28, 28)])
def test_upconv2x2_x3_vs_oracle(lib, n, cin, cout, h, w):
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 2, 2, generator=g) * (1.0 / cin) ** 0.5
    b = torch.randn(cout, generator=g) * 0.2
    ref = O.upconv2x2(x, wt, b)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    y = torch.full((n, 2 * h, 2 * w, cout), float("nan"), device="cuda")
    rc = lib.unet_op_upconv2x2_x3(0, _p(xd), n, h, w, cin, C.c_void_p(wt.numpy().ctypes.data),
                                  C.c_void_p(b.numpy().ctypes.data), cout, _p(y), None)
    assert rc == 0
    err = (y.cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


# the one-wave-per-SIMD structure (csrc/upconv_x3_r512.h), forced: full and ragged last pixel tiles, rows shorter than a
# 16-pixel fragment (w = 14: a fragment crosses two row ends), one to eight channel tiles, 2 to 8 stages
@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 128, 64, 7, 7), (1, 1024, 512, 14, 14), (3, 256, 128, 28, 28),
                                             (5, 128, 128, 14, 14), (2, 512, 256, 9, 11), (1, 128, 64, 112, 112)])
def test_upconv2x2_x3_r512_vs_oracle_and_first_structure(lib, n, cin, cout, h, w):
    g = torch.Generator().manual_seed(cin + cout + h + 5)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 2, 2, generator=g) * (1.0 / cin) ** 0.5
    b = torch.randn(cout, generator=g) * 0.2
    ref = O.upconv2x2(x, wt, b)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    ys = []
    for mode in (1, 0):
        y = torch.full((n, 2 * h, 2 * w, cout), float("nan"), device="cuda")
        prev = lib.unet_set_x3_upconv_r512(mode)
        try:
            rc = lib.unet_op_upconv2x2_x3(0, _p(xd), n, h, w, cin, C.c_void_p(wt.numpy().ctypes.data),
                                          C.c_void_p(b.numpy().ctypes.data), cout, _p(y), None)
        finally:
            lib.unet_set_x3_upconv_r512(prev)
        assert rc == 0
        err = (y.cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (mode, err)
        ys.append(y)
    assert torch.equal(ys[0], ys[1])      # same accumulation order: bit for bit the same planes


def test_x3_modelA_reference_frame(modelA, golden_dir):
    g = np.load(os.path.join(golden_dir, "modelA_frame_001410.npz"))
    frame = np.fromfile(os.path.join(golden_dir, "frame_001410_rgb_u8.bin"), dtype=np.uint8).reshape(1, 224, 224, 3)
    logits, probs, mask = modelA.run_u8(torch.from_numpy(frame).cuda(), return_probs=True, return_mask=True,
                                        precision="f16x3")
    assert modelA.device_error() == 0
    logits = logits.cpu().numpy()[0, 0]
    err = np.abs(logits - g["logits"]).max()
    print("f16x3 reference frame: max |dlogit| %.3e" % err)
    assert err < LOGIT_TOL, err
    mask = mask.cpu().numpy()[0]
    sure = np.abs(g["logits"]) > LOGIT_TOL
    assert np.array_equal(mask[sure], g["mask"][sure])          # bit-exact mask away from ties
    assert O.mask_iou(mask, g["mask"]) >= 1 - 1e-4
    p = probs.cpu().numpy()[0, 0]
    assert np.abs(p - 1 / (1 + np.exp(-g["logits"].astype(np.float64)))).max() < 1e-5


def test_x3_synthetic_frames_and_f32_entry(modelA, golden_dir):
    g = np.load(os.path.join(golden_dir, "modelA_synth2.npz"))
    frames = S.synthetic_frames(2, seed=0)
    a = modelA.run_u8(torch.from_numpy(frames).cuda(), precision="f16x3")
    err = np.abs(a.cpu().numpy()[:, 0] - g["logits"]).max()
    print("f16x3 synthetic frames: max |dlogit| %.3e" % err)
    assert err < LOGIT_TOL
    b = modelA.forward(O.normalize_u8_nhwc(frames).cuda(), precision="f16x3")     # forward(image) -> logits
    assert (a - b).abs().max().item() < 1e-4
    c = modelA.run_u8(torch.from_numpy(frames).cuda(), precision="fp32")
    print("f16x3 vs exact-fp32 tier: max |diff| %.3e" % (a - c).abs().max().item())


def test_x3_batch256_every_frame(modelA, golden_dir):
    """configs[1] at its benchmark size through this tier: every one of 256 frames against the reference's logits."""
    g = np.load(os.path.join(golden_dir, "modelA_synth2.npz"))
    ref = torch.from_numpy(g["logits"]).cuda()
    frames = torch.from_numpy(S.synthetic_frames(2, seed=0)).cuda().repeat(128, 1, 1, 1).contiguous()
    logits, mask = modelA.run_u8(frames, return_mask=True, precision="f16x3")
    assert modelA.device_error() == 0
    lg = logits[:, 0].view(128, 2, 224, 224)
    assert (lg - ref[None]).abs().max().item() < LOGIT_TOL
    assert torch.equal(lg, lg[:1].expand_as(lg))
    sure = (ref.abs() > LOGIT_TOL)[None].expand(128, -1, -1, -1)
    want = ((ref > 0).to(torch.uint8) * 255)[None].expand(128, -1, -1, -1)
    assert torch.equal(mask.view(128, 2, 224, 224)[sure], want[sure])


def test_x3_other_shapes_vs_oracle():
    """Sizes that are not multiples of the pixel tiles (partial columns, rows past the bottom, 16x16-tile levels), a
    3-level model whose head cannot fuse (Cout of the last layer != 64 is not the case here; depth differs)."""
    from unet_lane_detection_amd.model import UNetHIP
    for feats, shapes in (([64, 128, 256], [(2, 72, 104), (1, 160, 160), (3, 8, 8)]), ([64, 128], [(2, 40, 56)])):
        sdn = S.seeded_state_dict(feats, seed=4)
        m = UNetHIP(sdn, device=0)
        sd = O.to_torch_state(sdn)
        for (n, h, w) in shapes:
            frames = S.synthetic_frames(n, h, w, seed=h + w)
            with torch.no_grad():
                ref = O.forward(sd, O.normalize_u8_nhwc(frames))
            got = m.run_u8(torch.from_numpy(frames).cuda(), precision="f16x3").cpu()
            assert (got - ref).abs().max().item() < LOGIT_TOL, (feats, n, h, w)
        m.release()


def test_x3_640_frame(modelA):
    frames = S.synthetic_frames(1, 640, 640, seed=21)
    sd = O.to_torch_state(S.seeded_state_dict(seed=0))
    with torch.no_grad():
        ref = O.forward(sd, O.normalize_u8_nhwc(frames))
    got = modelA.run_u8(torch.from_numpy(frames).cuda(), precision="f16x3").cpu()
    assert (got - ref).abs().max().item() < LOGIT_TOL


def test_x3_rejects_unsupported_widths():
    from unet_lane_detection_amd import _lib
    from unet_lane_detection_amd.model import UNetHIP
    m = UNetHIP(S.seeded_state_dict([32, 64, 128], seed=1), device=0)     # model B widths: 32 is not a multiple of 64
    with pytest.raises(_lib.UnetError):
        m.run_u8(torch.from_numpy(S.synthetic_frames(1, 64, 64, seed=0)).cuda(), precision="f16x3")
    m.release()
