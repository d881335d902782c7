"""Host mirror of the reference's ROS callback with the camera stage on the GPU.

Reference: `LaneSegmentationROS` (src/unet_ros_node.py:231-311) subscribes to a `sensor_msgs/Image`, converts it with
a cv_bridge stand-in (src/tool.py:10-52), warps it with a fixed 4-point perspective matrix to 1055x685, converts to
RGB, runs `RKNNLaneInference.predict` (resize to 224x224, network, threshold, resize back; src/unet.py:24-97) and
publishes the mask as a mono8 Image.  Here the message bytes go to the GPU once and the mask bytes come back once;
warp + resize + colour swap are one kernel (`unet_ipm_prestage_u8`), the mask's way back another (`unet_resize_u8`).
rospy is not needed: `ImageMsg` carries the wire fields of sensor_msgs/Image.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib

# the reference's calibration (src/unet_ros_node.py:238-258)
REF_SRC_POINTS = ((29, 347), (619, 368), (202, 238), (422, 248))
REF_DST_POINTS = ((300, 580), (755, 580), (300, 100), (755, 100))
REF_WARP_SIZE = (1055, 685)   # (width, height)


@dataclass
class ImageMsg:
    """Wire fields of sensor_msgs/Image (src/tool.py:43-52 fills exactly these)."""
    height: int
    width: int
    encoding: str
    data: bytes
    step: int = 0
    is_bigendian: int = 0
    header: object = None
    extra: dict = field(default_factory=dict)

    def __post_init__(self):
        if not self.step:
            self.step = self.width * (3 if self.encoding in ("bgr8", "rgb8") else 1)


def get_perspective_transform(src, dst):
    """cv2.getPerspectiveTransform(src, dst): the 3x3 matrix mapping four source points onto four destination
    points (src/unet_ros_node.py:255), 8x8 linear system in float64."""
    src = np.asarray(src, dtype=np.float64).reshape(4, 2)
    dst = np.asarray(dst, dtype=np.float64).reshape(4, 2)
    a = np.zeros((8, 8))
    b = np.zeros(8)
    for i in range(4):
        (x, y), (u, v) = src[i], dst[i]
        a[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        a[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i], b[i + 4] = u, v
    return np.append(np.linalg.solve(a, b), 1.0).reshape(3, 3)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class CameraStage:
    """The two GPU kernels behind the C ABI, on torch device buffers."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("the camera stage needs a HIP device; there is no CPU fallback")
        self._lib = _lib.load()
        self.device = torch.device("cuda", int(device))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def prestage(self, img, matrix, warp_size, out_size=(224, 224), encoding="bgr8", step=None):
        """img: (H,W,3) uint8 device tensor (or a flat byte tensor with `step`); matrix: 3x3 source->destination;
        warp_size / out_size: (width, height).  Returns (out_h, out_w, 3) RGB uint8 on the device."""
        if encoding not in ("bgr8", "rgb8"):
            raise ValueError(f"Unsupported encoding: {encoding}")
        img = img.to(self.device).contiguous()
        if img.dim() == 3:
            h, w = int(img.shape[0]), int(img.shape[1])
            step = w * 3 if step is None else step
        else:
            raise ValueError("img must be (H,W,3) uint8")
        minv = np.ascontiguousarray(np.linalg.inv(np.asarray(matrix, dtype=np.float64)))
        out = torch.empty((out_size[1], out_size[0], 3), dtype=torch.uint8, device=self.device)
        rc = self._lib.unet_ipm_prestage_u8(self.device.index, _ptr(img), h, w, int(step), 1 if encoding == "bgr8" else 0,
                                            minv.ctypes.data_as(C.POINTER(C.c_double)), int(warp_size[0]),
                                            int(warp_size[1]), int(out_size[0]), int(out_size[1]), _ptr(out),
                                            self._stream())
        _lib.check(rc, "unet_ipm_prestage_u8")
        return out

    def resize(self, img, out_size):
        """cv2.resize(img, (width, height)) of a uint8 (H,W) or (H,W,C) device tensor."""
        img = img.to(self.device).contiguous()
        cn = 1 if img.dim() == 2 else int(img.shape[2])
        h, w = int(img.shape[0]), int(img.shape[1])
        shape = (out_size[1], out_size[0]) if img.dim() == 2 else (out_size[1], out_size[0], cn)
        out = torch.empty(shape, dtype=torch.uint8, device=self.device)
        rc = self._lib.unet_resize_u8(self.device.index, _ptr(img), h, w, cn, int(out_size[0]), int(out_size[1]), _ptr(out),
                                      self._stream())
        _lib.check(rc, "unet_resize_u8")
        return out


class LanePipelineGPU:
    """image_callback of the reference node (src/unet_ros_node.py:296-311) without rospy: ImageMsg -> mono8 ImageMsg."""

    def __init__(self, model, threshold=0.5, src_points=REF_SRC_POINTS, dst_points=REF_DST_POINTS,
                 warp_size=REF_WARP_SIZE, input_size=(224, 224), precision="fp32", on_error="raise"):
        """on_error: "raise" propagates every exception; "reference" reproduces the reference's failure handling:
        an inference failure (exception or empty output from the model) yields an all-zero mask of the warped size
        (`RKNNLaneInference.predict`, src/unet.py:81-92) and any other failure of the callback is logged and
        nothing is published (`image_callback`'s try/except, src/unet_ros_node.py:293-338: process returns None)."""
        if on_error not in ("raise", "reference"):
            raise ValueError(on_error)
        self.on_error = on_error
        self.last_error = None
        self.model = model                       # UNetHIP
        self.stage = CameraStage(model.device.index)
        self.threshold = threshold
        self.matrix = get_perspective_transform(src_points, dst_points)
        self.warp_size = tuple(warp_size)
        self.input_size = tuple(input_size)
        self.precision = precision

    def msg_to_device(self, msg):
        if msg.encoding not in ("bgr8", "rgb8"):
            raise ValueError(f"Unsupported encoding: {msg.encoding}")
        if msg.is_bigendian:
            pass   # 8-bit channels: byte order does not apply
        need = msg.step * msg.height
        buf = np.frombuffer(msg.data, dtype=np.uint8)
        if buf.size < need or msg.step < msg.width * 3:
            raise ValueError("Image data shorter than step * height")
        rows = torch.from_numpy(buf[:need].copy()).reshape(msg.height, msg.step)
        return rows.to(self.stage.device, non_blocking=True)

    def process(self, msg):
        """One callback: ImageMsg -> mono8 ImageMsg (see __init__ for the failure modes)."""
        if self.on_error == "raise":
            return self._process(msg, False)
        try:
            return self._process(msg, True)
        except Exception as e:   # src/unet_ros_node.py:337-338
            self.last_error = e
            print(f"Error in lane segmentation: {e}")
            return None

    def _run_checked(self, frame):
        """One forward plus the device's verdict on it (synchronises): a kernel-side failure raises instead of
        handing out stale pixels; a range report of the f16x3 tier (include/unet_hip.h, UNET_ERR_RANGE) re-runs the
        frame on the exact-fp32 tier and keeps the pipeline there."""
        for _ in range(2):
            out = self.model.run_u8(frame, return_mask=True, threshold=self.threshold, precision=self.precision)
            rc = self.model.device_error()
            if rc == 0:
                return out
            if rc == _lib.UNET_ERR_RANGE and self.precision == "f16x3":
                print("unet_hip: an activation left the fp16 range of the f16x3 tier; continuing on the fp32 tier")
                self.precision = "fp32"
                continue
            raise _lib.UnetError(rc, "unet_device_error")
        raise _lib.UnetError(rc, "unet_device_error")

    def _infer(self, frame, guarded):
        """predict (src/unet.py:74-97): mask of the network-input size, or None after a guarded failure."""
        if not guarded:
            return self._run_checked(frame)[1]
        try:
            out = self._run_checked(frame)
            if out is None or len(out) < 2:
                print("Warning: Model inference returned empty output")   # src/unet.py:85-87
                return None
            return out[1]
        except Exception as e:                    # src/unet.py:89-92
            self.last_error = e
            print(f"Inference error: {e}")
            return None

    def _process(self, msg, guarded):
        rows = self.msg_to_device(msg)
        # (H, step) bytes viewed as (H, W, 3) with a row pitch: the kernel takes the pitch separately
        lib = self.stage._lib
        out_w, out_h = self.input_size
        frame = torch.empty((1, out_h, out_w, 3), dtype=torch.uint8, device=self.stage.device)
        minv = np.ascontiguousarray(np.linalg.inv(self.matrix))
        rc = lib.unet_ipm_prestage_u8(self.stage.device.index, _ptr(rows), msg.height, msg.width, msg.step,
                                      1 if msg.encoding == "bgr8" else 0, minv.ctypes.data_as(C.POINTER(C.c_double)),
                                      self.warp_size[0], self.warp_size[1], out_w, out_h, _ptr(frame),
                                      self.stage._stream())
        _lib.check(rc, "unet_ipm_prestage_u8")
        mask = self._infer(frame, guarded)
        if mask is None:   # np.zeros(original_shape, uint8): original_shape is the warped image's (src/unet.py:31, :87)
            host = np.zeros((self.warp_size[1], self.warp_size[0]), dtype=np.uint8)
        else:
            host = self.stage.resize(mask[0], self.warp_size).cpu().numpy()
        return ImageMsg(height=host.shape[0], width=host.shape[1], encoding="mono8", data=host.tobytes(),
                        step=host.shape[1], is_bigendian=0, header=msg.header)
