"""Dev aid (GPU box): where the one-wave-per-SIMD transposed convolution differs from the oracle."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from oracle import unet_oracle as O
from unet_lane_detection_amd import _lib

lib = _lib.load(build_if_missing=False)


def run(n, cin, cout, h, w, mode):
    g = torch.Generator().manual_seed(cin + cout + h + 5)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 2, 2, generator=g) * (1.0 / cin) ** 0.5
    b = torch.randn(cout, generator=g) * 0.2
    ref = O.upconv2x2(x, wt, b).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    y = torch.full((n, 2 * h, 2 * w, cout), float("nan"), device="cuda")
    prev = lib.unet_set_x3_upconv_r512(mode)
    rc = lib.unet_op_upconv2x2_x3(0, C.c_void_p(xd.data_ptr()), n, h, w, cin, C.c_void_p(wt.numpy().ctypes.data),
                                  C.c_void_p(b.numpy().ctypes.data), cout, C.c_void_p(y.data_ptr()), None)
    lib.unet_set_x3_upconv_r512(prev)
    y = y.cpu()
    d = (y - ref).abs()
    bad = (d > 1e-3) | torch.isnan(d)
    print(f"shape n{n} cin{cin} cout{cout} {h}x{w} mode {mode}: rc {rc} nan {int(torch.isnan(y).sum())} bad {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        bn = bad.view(n, h, 2, w, 2, cout)          # n, y, a, x, b, c
        print("  bad by (a,b):", bn.sum(dim=(0, 1, 3, 5)).tolist())
        print("  bad by channel block of 16:", bn.view(n, h, 2, w, 2, cout // 16, 16).sum(dim=(0, 1, 2, 3, 4, 6)).tolist())
        pix = bn.sum(dim=(2, 4, 5)).view(-1)       # per input pixel
        print("  bad input pixels (first 40 flags):", (pix[:40] > 0).int().tolist(), " total bad pixels", int((pix > 0).sum()), "of", pix.numel())


for shp in [(1, 128, 64, 16, 16), (2, 128, 64, 7, 7), (1, 128, 64, 14, 14), (1, 256, 64, 16, 16), (1, 128, 128, 16, 16)]:
    run(*shp, mode=1)
run(1, 128, 64, 16, 16, mode=0)
