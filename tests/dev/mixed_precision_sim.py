"""Dev study (CPU, not a test): logit error on the reference frame of split-operand schemes that spend FEWER than three
fp16 MFMAs per product.  The main term w_hi x_hi stays an fp16 MFMA; the two cross terms (w_lo x_hi + w_hi x_lo) are
rounded to a narrower format (OCP fp8 e4m3 / e5m2, MX fp6 e2m3, fp4 e2m1 with a power-of-two scale) before they are
multiplied - on gfx950 v_mfma_scale_f32_16x16x128_f8f6f4 multiplies fp8 at twice and fp6 / fp4 at four times the fp16
rate, so the schemes cost 2.0 (fp8) and 1.5 (fp6) MFMA units per product against f16x3's 3.0.

usage: python tests/dev/mixed_precision_sim.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as O          # noqa: E402
from unet_lane_detection_amd import state as S  # noqa: E402


def f16(x):
    return x.to(torch.float16).float()


def q_fp8(x, shift, dt=torch.float8_e4m3fn):
    """round(x * 2^shift) to fp8 (saturating), back to fp32 in the original scale"""
    lim = 448.0 if dt == torch.float8_e4m3fn else 57344.0
    return (x * 2.0 ** shift).clamp(-lim, lim).to(dt).float() * 2.0 ** -shift


def q_small(x, shift, mant, emin, emax):
    """generic tiny float: `mant` explicit mantissa bits, normal exponents emin..emax, subnormals, saturating"""
    v = x * 2.0 ** shift
    a = v.abs()
    e = torch.floor(torch.log2(a.clamp_min(1e-38))).clamp(emin, emax)
    ulp = 2.0 ** (e - mant)
    q = torch.round(a / ulp) * ulp
    q = q.clamp_max((2.0 - 2.0 ** -mant) * 2.0 ** emax)
    return torch.sign(v) * q * 2.0 ** -shift


def q_fp6(x, shift):      # e2m3: 1.0 .. 7.5 normal, subnormals 0.125 steps
    return q_small(x, shift, 3, 0, 2)


def q_fp4(x, shift):      # e2m1
    return q_small(x, shift, 1, 0, 2)


def block_shift(x, dim, top):
    """per-32-block (along `dim`) power-of-two shift that brings the block's amax just under 2^top (MX E8M0 scale)"""
    shp = list(x.shape)
    c = shp[dim]
    assert c % 32 == 0 or c < 32
    b = min(32, c)
    xs = x.abs().unflatten(dim, (c // b, b))
    am = xs.amax(dim=dim + 1, keepdim=True).clamp_min(2.0 ** -60)
    sh = top - 1 - torch.floor(torch.log2(am))
    return sh.expand_as(xs).flatten(dim, dim + 1)


def make_scheme(kind):
    """returns conv(x, w, conv_fn) -> y where conv_fn(a, b) is the linear op on (activation, weight)"""
    def apply(x, w, lin):
        xh, wh = f16(x), f16(w)
        xl, wl = x - xh, w - wh
        main = lin(xh, wh)
        if kind == "f16x1":
            return main
        if kind == "f16x3":
            return main + lin(xh, f16(wl)) + lin(f16(xl), wh)
        if kind == "f16x2_wlo":       # drops w_hi x_lo
            return main + lin(xh, f16(wl))
        if kind == "f16x2_xlo":       # drops w_lo x_hi
            return main + lin(f16(xl), wh)
        if kind.startswith("f16x3_lo"):   # f16x3 with the lo parts rounded to a k-bit significand (fp16 MFMAs throughout)
            k = int(kind[len("f16x3_lo"):])

            def cut(v):
                m, e = torch.frexp(v)
                return torch.ldexp(torch.round(m * 2.0 ** k) / 2.0 ** k, e)
            return main + lin(xh, f16(cut(wl))) + lin(f16(cut(xl)), wh)
        # weights: per-output-channel power-of-two normalisation (as prescale_pow2 does) is assumed, emulated here by a
        # per-tensor shift that brings the largest |w| under 2^0
        wmax = float(w.abs().max())
        ws = -int(np.floor(np.log2(wmax))) - 1
        xmax = float(x.abs().max())
        xs = -int(np.floor(np.log2(max(xmax, 1e-30)))) - 1
        if kind in ("fp8", "fp8_e5m2"):
            dt = torch.float8_e4m3fn if kind == "fp8" else torch.float8_e5m2
            top = 8 if kind == "fp8" else 15
            # hi copies: largest value at 2^(top-1); lo parts are 2^-11 of that, shifted 11 further
            xh8, wh8 = q_fp8(x, xs + top - 1, dt), q_fp8(w, ws + top - 1, dt)
            xl8, wl8 = q_fp8(xl, xs + top - 1 + 11, dt), q_fp8(wl, ws + top - 1 + 11, dt)
            return main + lin(xh8, wl8) + lin(xl8, wh8)
        if kind in ("mxfp6", "mxfp4", "mxfp8"):
            q = {"mxfp6": q_fp6, "mxfp4": q_fp4, "mxfp8": lambda v, s: q_fp8(v, s)}[kind]
            top = 8 if kind == "mxfp8" else 3     # block amax goes under 2^top
            xh8 = q(x, block_shift(x, 1, top))
            xl8 = q(xl, block_shift(xl, 1, top))
            wh8 = q(w, block_shift(w, 1 if lin.cin_dim == 1 else 0, top))
            wl8 = q(wl, block_shift(wl, 1 if lin.cin_dim == 1 else 0, top))
            return main + lin(xh8, wl8) + lin(xl8, wh8)
        raise ValueError(kind)
    return apply


class Lin:
    def __init__(self, fn, cin_dim):
        self.fn, self.cin_dim = fn, cin_dim

    def __call__(self, a, b):
        return self.fn(a, b)


def run(tag, kind, sd, x, ref, first_exact=True):
    sch = make_scheme(kind)
    conv_lin = Lin(lambda a, b: F.conv2d(a, b, padding=1), 1)
    up_lin = Lin(lambda a, b: F.conv_transpose2d(a, b, None, stride=2), 0)
    x3 = make_scheme("f16x3")

    def conv3(a, w):
        if a.shape[1] < 32:     # the first convolution (3 channels, 0.6 % of the flops) stays on three terms
            return x3(a, w, conv_lin)
        return sch(a, w, conv_lin)

    def convt(a, w, b):
        return sch(a, w, up_lin) + b[None, :, None, None]

    O_conv3x3, O_up = O.conv3x3, O.upconv2x2
    O.conv3x3, O.upconv2x2 = conv3, convt
    try:
        with torch.no_grad():
            y = O.forward(sd, x).numpy()[0, 0]
    finally:
        O.conv3x3, O.upconv2x2 = O_conv3x3, O_up
    d = np.abs(y - ref)
    flips = int(((y > 0) != (ref > 0)).sum())
    print(f"{tag:44s} max|dlogit| {d.max():.3e}  rms {np.sqrt((d**2).mean()):.3e}  mask flips {flips}", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(8)
    g = np.load(os.path.join(ROOT, "tests/golden/modelA_frame_001410.npz"))
    frame = np.fromfile(os.path.join(ROOT, "tests/golden/frame_001410_rgb_u8.bin"), dtype=np.uint8).reshape(1, 224, 224, 3)
    sd = O.to_torch_state(S.seeded_state_dict(seed=0))
    x = O.normalize_u8_nhwc(frame)
    ref = g["logits"]
    print("logit range", ref.min(), ref.max(), " |logit| < 1e-3:", int((np.abs(ref) < 1e-3).sum()))
    kinds = sys.argv[1:] or ["f16x3", "f16x1", "f16x2_wlo", "f16x2_xlo", "fp8", "fp8_e5m2", "mxfp8", "mxfp6", "mxfp4"]
    for k in kinds:
        run(k, k, sd, x, ref)
