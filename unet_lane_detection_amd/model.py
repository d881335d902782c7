"""Host-side mirror of the reference's float model API on top of the C ABI.

`UNetHIP.forward(image) -> logits` keeps the signature of the reference's
`UNet.forward` (reference README.md:1460-1481): (N,3,H,W) float32 in,
(N,1,H,W) float32 pre-sigmoid logits out.  `run_u8` is the container-side
entry (uint8 NHWC frames, normalisation fused on device).  PyTorch is used
for device memory and streams only; every arithmetic step runs in
libunet_hip.so.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .state import DEFAULT_FEATURES, INPUT_MEAN, INPUT_STD


def _as_f32_host(v):
    if torch.is_tensor(v):
        v = v.detach().to("cpu", torch.float32).contiguous().numpy()
    return np.ascontiguousarray(np.asarray(v), dtype=np.float32)


def infer_features(state_dict):
    feats = []
    while f"encoder_blocks.{len(feats)}.0.weight" in state_dict:
        feats.append(int(state_dict[f"encoder_blocks.{len(feats)}.0.weight"].shape[0]))
    if not feats:
        raise ValueError("state_dict has no encoder_blocks.*.0.weight keys")
    return feats


class UNetHIP:
    """U-Net forward on one MI355X through libunet_hip.so."""

    def __init__(self, state_dict=None, features=None, device=0, in_channels=3, out_channels=1):
        if not torch.cuda.is_available():
            raise RuntimeError("UNetHIP needs a HIP device; there is no CPU fallback")
        self._lib = _lib.load()
        if features is None:
            features = infer_features(state_dict) if state_dict is not None else list(DEFAULT_FEATURES)
        self.features = [int(f) for f in features]
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        cfg = _lib.UnetConfig()
        cfg.in_channels, cfg.out_channels, cfg.depth = in_channels, out_channels, len(self.features)
        for i, f in enumerate(self.features):
            cfg.features[i] = f
        cfg.device = self.device_index
        for i in range(3):
            cfg.input_mean[i] = INPUT_MEAN[i]
            cfg.input_std[i] = INPUT_STD[i]
        h = C.c_void_p()
        _lib.check(self._lib.unet_create(C.byref(cfg), C.byref(h)), "unet_create")
        self._h = h
        self.multiple = 1 << len(self.features)
        if state_dict is not None:
            self.load_state_dict(state_dict)

    @classmethod
    def from_checkpoint(cls, model_path, device=0):
        """Model from a float checkpoint file: bare state_dict, the reference's wrapped form
        {'model_state_dict': ...} (README.md:2208-2213), .npz, or 'seed:<int>' test weights."""
        from .py_utils.rknn_executor import load_float_state_dict
        return cls(load_float_state_dict(model_path), device=device)

    # ---- parameters ------------------------------------------------------------------
    def param_names(self):
        n = self._lib.unet_num_params(self._h)
        return [self._lib.unet_param_name(self._h, i).decode() for i in range(n)]

    def load_state_dict(self, state_dict, strict=True):
        """Accepts the reference module's state_dict (tensors or numpy arrays); integer
        `num_batches_tracked` entries are ignored (inference does not use them)."""
        self._require_live()
        names = self.param_names()
        missing = [k for k in names if k not in state_dict]
        extra = [k for k in state_dict if k not in names and not k.endswith("num_batches_tracked")]
        if strict and (missing or extra):
            raise KeyError(f"state_dict mismatch: missing={missing[:4]}... unexpected={extra[:4]}...")
        for k in names:
            if k in state_dict:
                a = _as_f32_host(state_dict[k]).reshape(-1)
                rc = self._lib.unet_load_param(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), a.size)
                _lib.check(rc, f"unet_load_param({k})", self._h)
        _lib.check(self._lib.unet_finalize(self._h), "unet_finalize", self._h)

    # ---- forward ---------------------------------------------------------------------
    def _require_live(self):
        if self._h is None:
            raise RuntimeError("UNetHIP has been released")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reserve(self, n, h, w):
        self._require_live()
        _lib.check(self._lib.unet_reserve(self._h, n, h, w), "unet_reserve", self._h)

    def workspace_bytes(self, n, h, w):
        return int(self._lib.unet_workspace_bytes(self._h, n, h, w))

    def _outputs(self, n, h, w, want_probs, want_mask):
        logits = torch.empty((n, 1, h, w), dtype=torch.float32, device=self.device)
        probs = torch.empty_like(logits) if want_probs else None
        mask = torch.empty((n, h, w), dtype=torch.uint8, device=self.device) if want_mask else None
        return logits, probs, mask

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    PRECISIONS = ("fp32", "f16x3", "f16q8", "bf16")

    def forward(self, image, return_probs=False, return_mask=False, threshold=0.5, precision="fp32"):
        """image: (N,3,H,W) float32 on this device, already normalised -> logits (N,1,H,W).
        precision "fp32" (exact fp32 MFMA) or "f16x3" (split operands, same accuracy class)."""
        self._require_live()
        if image.dim() != 4:
            raise ValueError("image must be (N,C,H,W)")
        image = image.to(self.device, torch.float32).contiguous()
        n, c, h, w = image.shape
        logits, probs, mask = self._outputs(n, h, w, return_probs, return_mask)
        fns = {"fp32": self._lib.unet_forward_f32, "f16x3": self._lib.unet_forward_f32_x3}
        if precision not in fns:
            raise ValueError(f"precision={precision!r}: forward() takes one of {sorted(fns)}")
        fn = fns[precision]
        rc = fn(self._h, self._ptr(image), n, h, w, self._ptr(logits), self._ptr(probs),
                self._ptr(mask), _logit(threshold), self._stream())
        _lib.check(rc, f"unet_forward_f32[{precision}]", self._h)
        return _pack(logits, probs, mask, return_probs, return_mask)

    __call__ = forward

    def run_u8(self, frames, return_probs=False, return_mask=False, threshold=0.5, precision="fp32"):
        """frames: (N,H,W,3) uint8 RGB on this device, un-normalised -> logits (N,1,H,W).
        precision "fp32" (default: exact fp32 MFMA), "f16x3" (fp16 hi + lo operands, three MFMAs per product: the
        same accuracy class at 3/16 of the MFMA cost), "f16q8" (f16x3 with the cross terms of the wide 3x3 convolutions
        in fp8: logits within 1e-3, its own tier) or "bf16" (bf16 storage, fp32 accumulate: its own tier)."""
        self._require_live()
        if frames.dim() != 4 or frames.shape[-1] != 3 or frames.dtype != torch.uint8:
            raise ValueError("frames must be (N,H,W,3) uint8")
        frames = frames.to(self.device).contiguous()
        n, h, w, _ = frames.shape
        logits, probs, mask = self._outputs(n, h, w, return_probs, return_mask)
        fns = {"fp32": self._lib.unet_forward_u8, "f16x3": self._lib.unet_forward_u8_x3,
               "f16q8": self._lib.unet_forward_u8_x3, "bf16": self._lib.unet_forward_u8_bf16}
        if precision not in fns:
            raise ValueError(f"precision={precision!r}: run_u8() takes one of {sorted(fns)}")
        fn = fns[precision]
        # "f16q8": the f16x3 forward with the cross terms of its wide 3x3 convolutions on the fp8 matrix pipe - a
        # thread-local switch of the library (include/unet_hip.h), set for this call of this thread
        prev = self._lib.unet_set_x3_cross_fp8(1 if precision == "f16q8" else 0) if precision in ("f16x3", "f16q8") else None
        try:
            rc = fn(self._h, self._ptr(frames), n, h, w, self._ptr(logits), self._ptr(probs),
                    self._ptr(mask), _logit(threshold), self._stream())
        finally:
            if prev is not None:
                self._lib.unet_set_x3_cross_fp8(prev)
        _lib.check(rc, f"unet_forward_u8[{precision}]", self._h)
        return _pack(logits, probs, mask, return_probs, return_mask)

    # ---- per-launch timing ------------------------------------------------------------
    def profile(self, on=True):
        """Bracket every kernel launch of later forward calls with HIP events (clears old records)."""
        self._require_live()
        _lib.check(self._lib.unet_profile_enable(self._h, 1 if on else 0), "unet_profile_enable", self._h)

    def profile_records(self):
        """[(kernel name, ms, algorithmic flops, algorithmic bytes)] since profile(True)."""
        self._require_live()
        out = []
        name = C.create_string_buffer(64)
        ms, fl, by = C.c_double(), C.c_double(), C.c_double()
        for i in range(self._lib.unet_profile_count(self._h)):
            self._lib.unet_profile_get(self._h, i, name, 64, C.byref(ms), C.byref(fl), C.byref(by))
            out.append((name.value.decode(), ms.value, fl.value, by.value))
        return out

    def device_error(self, current_stream_only=False):
        """Synchronise the device (or, with `current_stream_only`, just torch's current stream on it - enough when every
        forward since the last call was launched there) and return the status of every launch on this handle since the
        last call, clearing it: 0 = ok, UNET_ERR_HIP after a kernel-side failure (a timed-out wave-progress wait),
        UNET_ERR_RANGE (7) when an f16x3 forward met an activation beyond the fp16 range - those results are not at fp32
        parity and the frames should be re-run with precision="fp32"."""
        self._require_live()
        if current_stream_only:
            return int(self._lib.unet_device_error_on(self._h, self._stream()))
        return int(self._lib.unet_device_error(self._h))

    def release(self):
        if getattr(self, "_h", None) is not None:
            self._lib.unet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def _logit(t):
    t = float(t)
    if t <= 0.0:
        return -math.inf
    if t >= 1.0:
        return math.inf
    return math.log(t / (1.0 - t))


def _pack(logits, probs, mask, want_probs, want_mask):
    if not want_probs and not want_mask:
        return logits
    out = [logits]
    if want_probs:
        out.append(probs)
    if want_mask:
        out.append(mask)
    return tuple(out)
