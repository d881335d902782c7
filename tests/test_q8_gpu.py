"""The "f16q8" tier (csrc/conv_q8_r512.h): the f16x3 split-operand product with its two cross terms on the fp8 matrix
pipe.  It is an accuracy tier of its own: held to BASELINE.json's north_star tolerance (logits within 1e-3 of the
reference's, reference README.md:1449-1458 being plain fp32), not to the 2e-4 the f16x3 tier is tested to; the tolerances
below are written against the measured figures (profiles/r03/mx_experiments.md)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O
from unet_lane_detection_amd import state as S

pytestmark = pytest.mark.gpu

LOGIT_TOL_Q8 = 1e-3      # BASELINE.json: pre-sigmoid logits within 1e-3


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.fixture(scope="module")
def lib():
    from unet_lane_detection_amd import _lib
    return _lib.load(build_if_missing=False)


def _conv(lib, xd, wt, scale, shift, n, h, w, cin, cout, relu, tw, pool=False):
    y = torch.full((n, h, w, cout), float("nan"), device="cuda")
    yp = torch.full((n, h // 2, w // 2, cout), float("nan"), device="cuda") if pool else None
    rc = lib.unet_op_conv3x3_x3(0, _p(xd), n, h, w, cin, C.c_void_p(wt.numpy().ctypes.data),
                                C.c_void_p(scale.numpy().ctypes.data), C.c_void_p(shift.numpy().ctypes.data),
                                cout, relu, tw, _p(y), _p(yp), None)
    return rc, y, yp


# (n, cin, cout, h, w, tw): whole 8 x 28 tiles, the batch tiled as one tall image (28, 14), rows past the image bottom,
# one and several channel groups / chunk pairs
Q8_CASES = [(2, 64, 256, 56, 56, 428), (3, 128, 512, 28, 28, 428), (5, 64, 256, 14, 14, 414), (1, 64, 256, 20, 28, 428),
            (4, 192, 256, 12, 28, 428), (1, 256, 256, 14, 14, 414), (2, 64, 512, 8, 84, 428)]


@pytest.mark.parametrize("n,cin,cout,h,w,tw", Q8_CASES)
def test_conv3x3_q8_vs_oracle_and_f16x3(lib, n, cin, cout, h, w, tw):
    """Activations as the tier's planes hold them (4 sigma ~ 512 ... 1024: the fp8 shifts of the q plane are chosen for
    that range, csrc/unet_x3.inc act_from_bn).  Per product the cross terms are kept to 2^-5 of 2^-11: the sum is off by
    ~2^-17 sum |w x| at most, measured 1e-5 of the output range; the f16x3 kernel on the same operands is 10 x closer."""
    g = torch.Generator().manual_seed(cin * 3 + cout + h * 5 + w + n)
    x = torch.relu(torch.randn(n, cin, h, w, generator=g)) * 200.0
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5 / 200.0
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.3
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    for relu in (1, 0):
        ref = O.conv3x3(x, wt) * scale[None, :, None, None] + shift[None, :, None, None]
        if relu:
            ref = torch.relu(ref)
        rc, y, _ = _conv(lib, xd, wt, scale, shift, n, h, w, cin, cout, relu, tw)
        assert rc == 0, rc
        top = max(1.0, ref.abs().max().item())
        err = (y.cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
        assert err < 1e-4 * top, (err, top, relu)
        rc, y3, _ = _conv(lib, xd, wt, scale, shift, n, h, w, cin, cout, relu, 28 if tw == 428 else 14)
        assert rc == 0, rc
        err3 = (y3.cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
        assert err3 < 2e-5 * top
        print("q8 %.2e  f16x3 %.2e  (output range %.2f)" % (err, err3, top))


def test_conv3x3_q8_saturating_planes_stay_finite(lib):
    """Activations far above the range the q plane's shifts are chosen for (|x| up to ~2e4 where fp8(x / 8) saturates at
    3584): the fp8 copies saturate - they are clamped before the conversion, never NaN - and the layer degrades towards
    plain-fp16 accuracy (cross terms partly lost: ~2^-11 of the product) instead of failing."""
    g = torch.Generator().manual_seed(5)
    n, cin, cout, h, w = 2, 64, 256, 16, 28
    x = torch.relu(torch.randn(n, cin, h, w, generator=g)) * 5000.0
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5 / 5000.0
    scale, shift = torch.ones(cout), torch.zeros(cout)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    ref = torch.relu(O.conv3x3(x, wt))
    rc, y, _ = _conv(lib, xd, wt, scale, shift, n, h, w, cin, cout, 1, 428)
    assert rc == 0
    assert torch.isfinite(y).all()
    err = (y.cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
    print("saturating q plane: max err %.2e of range %.2f" % (err, ref.abs().max().item()))
    assert err < 2e-3 * max(1.0, ref.abs().max().item())


def test_conv3x3_q8_pool_and_rejection(lib):
    g = torch.Generator().manual_seed(11)
    n, cin, cout, h, w = 2, 64, 256, 56, 56
    x = torch.relu(torch.randn(n, cin, h, w, generator=g)) * 200.0
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5 / 200.0
    scale, shift = torch.ones(cout), torch.zeros(cout)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    rc, y, yp = _conv(lib, xd, wt, scale, shift, n, h, w, cin, cout, 1, 428, pool=True)
    assert rc == 0
    assert torch.equal(yp.cpu(), O.maxpool2x2(y.cpu().permute(0, 3, 1, 2)).permute(0, 2, 3, 1))
    # shapes the kernel does not take are refused, not run on something else: Cout not a multiple of 256, width 14 with 428
    rc, _, _ = _conv(lib, xd, wt[:128], scale[:128], shift[:128], n, h, w, cin, 128, 1, 428)
    assert rc != 0
    x14 = xd[:, :14, :14].contiguous()
    rc, _, _ = _conv(lib, x14, wt, scale, shift, n, 14, 14, cin, cout, 1, 428)
    assert rc != 0


def test_q8_modelA_batch256_vs_reference_golden_and_f16x3(golden_dir):
    """Whole network at the benchmark batch (256: the eight C >= 128 layers at 56x56 and 28x28 take the tier's kernel where
    its 256-channel tiles fill the CUs evenly): every frame within 1e-3 of the reference's golden logits, masks identical
    wherever |logit| exceeds that, the tier's kernels really ran, and the f16x3 tier on the same frames is untouched by the
    switch."""
    from unet_lane_detection_amd.model import UNetHIP
    g = np.load(os.path.join(golden_dir, "modelA_synth2.npz"))
    ref = torch.from_numpy(g["logits"]).cuda()                       # (2,224,224)
    frames = torch.from_numpy(S.synthetic_frames(2, seed=0)).cuda().repeat(128, 1, 1, 1).contiguous()
    m = UNetHIP(S.seeded_state_dict(seed=0), device=0)
    try:
        x3_before = m.run_u8(frames, precision="f16x3")[:, 0].clone()
        m.profile(True)
        lq, mask = m.run_u8(frames, return_mask=True, precision="f16q8")
        names = [r[0] for r in m.profile_records()]
        m.profile(False)
        assert m.device_error() == 0
        assert sum(nm.startswith("conv3x3_q8_f16q8") for nm in names) == 8, names
        # the first convolution of four blocks hands its output's q plane straight to the second (no conversion pass)
        assert sum(nm.startswith("conv3x3_q8_f16q8") and "_q" in nm[16:] for nm in names) == 4, names
        # ... the pooling passes of the two 56x56 / 28x28 encoder blocks and the two transposed convolutions write the q
        # planes of the concat buffers and of the pooled tensor; one conversion pass is left (the 56x56 level's input
        # comes from a layer of the first structure)
        assert names.count("maxpool2x2_planes_q8") == 2 and names.count("upconv2x2_r512_f16x3_q") == 2, names
        assert names.count("planes_to_q8") == 1, names
        x3_after = m.run_u8(frames, precision="f16x3")[:, 0]
        assert torch.equal(x3_before, x3_after)
        lg = lq[:, 0].view(128, 2, 224, 224)
        err = (lg - ref[None]).abs().amax(dim=(1, 2, 3))
        print("f16q8 batch 256: max |dlogit| vs golden %.3e, vs f16x3 %.3e" %
              (err.max().item(), (lq[:, 0] - x3_after).abs().max().item()))
        assert err.max().item() < LOGIT_TOL_Q8, err.max().item()
        assert torch.equal(lg, lg[:1].expand_as(lg))                 # batch position does not matter, bit for bit
        sure = (ref.abs() > LOGIT_TOL_Q8)[None].expand(128, -1, -1, -1)
        want = ((ref > 0).to(torch.uint8) * 255)[None].expand(128, -1, -1, -1)
        assert torch.equal(mask.view(128, 2, 224, 224)[sure], want[sure])
        got = mask.view(128, 2, 224, 224)[0] > 0
        inter = (got & (ref > 0)).sum().item()
        union = (got | (ref > 0)).sum().item()
        print("mask IoU vs reference on the two frames: %.6f (%d of %d pixels differ)" %
              (inter / max(1, union), int((got != (ref > 0)).sum().item()), got.numel()))
        assert inter / max(1, union) > 1.0 - 5e-4
    finally:
        m.release()
