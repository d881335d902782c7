"""int8 tier of the deployed network ("model B": features [32, 64, 128], sigmoid head) on the HIP path
(SURVEY.md section 8 row f4).  Stands where the reference's quantised .rknn blob stands behind
`RKNN_model_container.run` (src/py_utils/rknn_executor.py:26-38): uint8 NHWC frames in, probabilities out.

  ranges = calibrate(float_model, frames)                  # device pass over calibration frames (README.md:3046-3078)
  qmodel = quant.quantize_model(state_dict, ranges)        # README.md:3106-3116 scheme, host arithmetic on weights
  net = UNetInt8(qmodel, device=0); net.run_u8(frames)

All arithmetic on activations runs in libunet_hip.so (csrc/conv_i8.h); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, quant
from .model import _logit, _pack


def calibrate(float_model, frames_u8, batch=8):
    """Per-tensor activation ranges of the float network over `frames_u8` ((N,H,W,3) uint8, host or device):
    {tensor name: (min, max)} for quant.quantize_model.  Runs the fp32 HIP tier with a min/max probe behind every
    layer (unet_forward_u8_ranges); frames are processed `batch` at a time and the ranges merged."""
    lib = _lib.load()
    h = float_model._h
    nt = int(lib.unet_num_range_tensors(h))
    names = quant.tensor_names(len(float_model.features))
    assert nt == len(names), (nt, names)
    frames = torch.as_tensor(frames_u8)
    lo = np.full(nt, np.inf)
    hi = np.full(nt, -np.inf)
    buf = (C.c_float * (2 * nt))()
    for i in range(0, frames.shape[0], batch):
        f = frames[i:i + batch].to(float_model.device).contiguous()
        n, hh, ww, _ = f.shape
        rc = lib.unet_forward_u8_ranges(h, C.c_void_p(f.data_ptr()), n, hh, ww, buf,
                                        C.c_void_p(torch.cuda.current_stream(float_model.device).cuda_stream))
        _lib.check(rc, "unet_forward_u8_ranges", h)
        r = np.frombuffer(buf, dtype=np.float32).reshape(nt, 2)
        lo = np.minimum(lo, r[:, 0])
        hi = np.maximum(hi, r[:, 1])
    return {name: (float(lo[i]), float(hi[i])) for i, name in enumerate(names)}


class UNetInt8:
    """Quantised U-Net forward on one MI355X (i8 MFMA)."""

    def __init__(self, qmodel, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("UNetInt8 needs a HIP device; there is no CPU fallback")
        self._lib = _lib.load()
        self.features = [int(f) for f in qmodel["features"]]
        self.device = torch.device("cuda", int(device))
        feats = (C.c_int * len(self.features))(*self.features)
        h = C.c_void_p()
        _lib.check(self._lib.unet_i8_create(len(self.features), feats, int(device), C.byref(h)), "unet_i8_create")
        self._h = h
        self.multiple = 1 << len(self.features)
        for k, v in qmodel.items():
            if k == "features":
                continue
            a = np.ascontiguousarray(np.asarray(v))
            if a.dtype == np.float64:
                a = a.astype(np.float32)
            if a.dtype == np.int64:
                a = a.astype(np.int32)
            self._check(self._lib.unet_i8_load(h, k.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes), f"unet_i8_load({k})")
        self._check(self._lib.unet_i8_finalize(h), "unet_i8_finalize")

    @classmethod
    def from_file(cls, path, device=0):
        return cls(quant.load_quantized(path), device=device)

    def _check(self, rc, where):
        if rc != 0:
            msg = self._lib.unet_i8_last_error(self._h)
            raise _lib.UnetError(rc, where, msg.decode() if msg else "")

    def _require_live(self):
        if self._h is None:
            raise RuntimeError("UNetInt8 has been released")

    def run_u8(self, frames, return_probs=False, return_mask=False, threshold=0.5):
        """frames: (N,H,W,3) uint8 RGB on this device -> logits (N,1,H,W) float32 [, probabilities, mask]."""
        self._require_live()
        if frames.dim() != 4 or frames.shape[-1] != 3 or frames.dtype != torch.uint8:
            raise ValueError("frames must be (N,H,W,3) uint8")
        frames = frames.to(self.device).contiguous()
        n, h, w, _ = frames.shape
        logits = torch.empty((n, 1, h, w), dtype=torch.float32, device=self.device)
        probs = torch.empty_like(logits) if return_probs else None
        mask = torch.empty((n, h, w), dtype=torch.uint8, device=self.device) if return_mask else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        rc = self._lib.unet_i8_forward_u8(self._h, p(frames), n, h, w, p(logits), p(probs), p(mask), _logit(threshold),
                                          C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        self._check(rc, "unet_i8_forward_u8")
        return _pack(logits, probs, mask, return_probs, return_mask)

    def read_tensor(self, name, n, h, w):
        """int8 activation tensor `name` of the last forward as an (N,C,h,w) numpy array (parity tests); h, w are that
        tensor's spatial size."""
        self._require_live()
        cap = n * h * w * 512
        buf = np.empty(cap, dtype=np.int8)
        c = C.c_int()
        self._check(self._lib.unet_i8_read_tensor(self._h, name.encode(), buf.ctypes.data_as(C.c_void_p), cap, C.byref(c)),
                    f"unet_i8_read_tensor({name})")
        return buf[:n * h * w * c.value].reshape(n, h, w, c.value).transpose(0, 3, 1, 2).copy()

    def release(self):
        if getattr(self, "_h", None) is not None:
            torch.cuda.synchronize(self.device)
            self._lib.unet_i8_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass
