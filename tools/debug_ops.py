import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from oracle import unet_oracle as O
from unet_lane_detection_amd import _lib
lib = _lib.load()
def p(t): return C.c_void_p(t.data_ptr())
for (n, cin, cout, h, w) in [(3, 64, 64, 12, 16), (3, 64, 128, 12, 16), (3, 128, 64, 12, 16), (3, 128, 128, 6, 8), (1, 64, 64, 12, 16), (3, 32, 32, 12, 16), (3, 64, 64, 12, 12)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w, generator=g); dz = torch.randn(n, cout, h, w, generator=g)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).requires_grad_(True)
    y = O.conv3x3(x, wt); y.backward(dz)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda(); dzd = dz.permute(0, 2, 3, 1).contiguous().cuda()
    dw = torch.zeros(cout, cin, 3, 3, device='cuda')
    rc = lib.unet_op_wgrad3x3(0, p(dzd), p(xd), n, h, w, cin, cout, p(dw), None)
    e1 = (dw.cpu() - wt.grad).abs().max().item() / wt.grad.abs().max().item()
    yd = torch.zeros(n, h, w, cout, device='cuda')
    ones = np.ones(cout, np.float32); zeros = np.zeros(cout, np.float32); wn = wt.detach().numpy()
    rc2 = lib.unet_op_conv3x3(0, p(xd), n, h, w, cin, wn.ctypes.data_as(C.c_void_p), ones.ctypes.data_as(C.c_void_p), zeros.ctypes.data_as(C.c_void_p), cout, 0, p(yd), None)
    e2 = (yd.cpu().permute(0, 3, 1, 2) - y.detach()).abs().max().item() / y.abs().max().item()
    print((n, cin, cout, h, w), "wgrad rel", f"{e1:.2e}", "conv rel", f"{e2:.2e}", rc, rc2)
