"""Drop-in for the reference's model container (reference src/py_utils/rknn_executor.py:4-42).

Same class name, constructor and methods, so that
`from py_utils.rknn_executor import RKNN_model_container` in the reference's
callers (src/unet.py:12, src/unet_ros_node.py:18) resolves to this module when
this directory's parent is put on sys.path ahead of the reference's own
`py_utils`.  Behind it sits the MI355X HIP path instead of
the Rockchip NPU runtime.

Behavioural contract kept from the reference:
  * `RKNN_model_container(model_path, target=None, device_id=None)`; `target`
    is accepted and ignored, `device_id` (a string such as '0') selects the
    HIP device  (rknn_executor.py:5, src/unet.py:21);
  * `run(inputs)` wraps a bare array in a list (rknn_executor.py:31-34), takes
    the un-normalised uint8 NHWC frame the caller builds (src/unet.py:39-40)
    and returns a list whose first element is a float32 (N,1,H,W) array;
    like the deployed blob (last op ConvSigmoid) the values are
    probabilities, so the caller's logits guard (src/unet.py:63) stays inert;
  * after `release()`, `run` prints the reference's error line and returns []
    (rknn_executor.py:27-29); `release()` may be called twice
    (src/unet.py:148-150 calls it again from `__del__`).
"""
from __future__ import annotations

import os
import threading

import numpy as np
import torch

from .._lib import UNET_ERR_RANGE
from ..model import UNetHIP
from ..state import DEFAULT_FEATURES, seeded_state_dict

# Arithmetic tier behind the container, from the environment (the reference's constructor has no argument for it):
#   UNET_HIP_TIER = auto (default) | f16x3 | fp32 | bf16
# auto = the split-operand fp16 tier (fp32-level accuracy, the fastest tier that meets the fp32 parity bar) when the
# model's widths allow it (multiples of 64), else exact fp32.  A quantised model file (*.npz written by
# unet_lane_detection_amd.quant.save_quantized, recognised by its 'input.lut' array) always runs on the int8 tier.
# The f16x3 tier stores activations as fp16 planes; the reference's network is plain fp32 (README.md:1449-1458) and has
# no range limit.  The kernels report an activation beyond the fp16 range (UNET_ERR_RANGE from device_error): under
# `auto` the container then re-runs those frames on the exact-fp32 tier and serves the next RANGE_RETRY_FRAMES frames
# from it (one saturated camera frame must not cost a ROS node its frame rate for the rest of its life: the fp32 tier is
# 2.1x slower); after that many clean frames it tries the f16x3 tier again, and every further range report doubles the
# wait (a checkpoint that is out of range on every frame ends up asking once per RANGE_RETRY_MAX frames).  With the tier
# forced to f16x3 it raises (the caller's predict() turns that into a zero mask, src/unet.py:81-92) rather than return
# clamped results.
TIER_ENV = "UNET_HIP_TIER"
RANGE_RETRY_ENV = "UNET_HIP_RANGE_RETRY"   # frames on the fp32 tier before f16x3 is tried again; 0 = stay on fp32
RANGE_RETRY_FRAMES = 64
RANGE_RETRY_MAX = 1 << 16
# UNET_HIP_GRAPH = 1 (default) | 0: small host batches (<= GRAPH_MAX_FRAMES frames, the reference's one-frame calls) are
# served by replaying a captured HIP graph of the forward pass (one graph per input shape: ~35 kernel launches become
# one), with pinned staging buffers for the frame and the probabilities.
GRAPH_ENV = "UNET_HIP_GRAPH"
GRAPH_MAX_FRAMES = 8
GRAPH_MAX_SHAPES = 8     # captured graphs kept at a time (oldest dropped first)


def _pick_tier(features):
    """-> (tier, chosen automatically)"""
    t = os.environ.get(TIER_ENV, "auto").lower()
    if t not in ("auto", "f16x3", "fp32", "bf16"):
        raise ValueError(f"{TIER_ENV}={t!r}: expected auto, f16x3, fp32 or bf16")
    if t == "auto":
        return ("f16x3" if all(f % 64 == 0 for f in features) and 2 * features[-1] <= 1024 else "fp32"), True
    return t, False


def load_float_state_dict(model_path):
    """`model_path` -> state_dict.  Accepted: a torch checkpoint holding either the bare
    state_dict or the reference's wrapped form {'model_state_dict': ...} (reference
    README.md:2208-2213, :2876-2880), an .npz of arrays, or 'seed:<int>[:f0,f1,...]' for the
    reproducible test weights (unet_lane_detection_amd.state.seeded_state_dict)."""
    if isinstance(model_path, dict):
        return model_path
    p = str(model_path)
    if p.startswith("seed:"):
        parts = p.split(":")
        feats = [int(x) for x in parts[2].split(",")] if len(parts) > 2 else list(DEFAULT_FEATURES)
        return seeded_state_dict(feats, seed=int(parts[1]))
    if not os.path.exists(p):
        raise FileNotFoundError(p)
    if p.endswith(".npz"):
        with np.load(p, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    obj = torch.load(p, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "model_state_dict" in obj:
        obj = obj["model_state_dict"]
    return obj


class RKNN_model_container:
    def __init__(self, model_path, target=None, device_id=None) -> None:
        print('--> Init runtime environment')
        try:
            dev = int(device_id) if device_id not in (None, "") else 0
            quantised = None
            if isinstance(model_path, (str, os.PathLike)) and str(model_path).endswith(".npz") and os.path.exists(str(model_path)):
                with np.load(str(model_path), allow_pickle=False) as z:
                    if "input.lut" in z.files:
                        quantised = {k: z[k] for k in z.files}
            if quantised is not None:       # the deployed form: an int8 model (the reference loads an int8 .rknn blob)
                from ..int8 import UNetInt8
                self.model = UNetInt8(quantised, device=dev)
                self.precision, self._auto_tier = "int8", False
            else:
                self.model = UNetHIP(load_float_state_dict(model_path), device=dev)
                self.precision, self._auto_tier = _pick_tier(self.model.features)
        except Exception as e:  # reference: print + exit(ret) on init failure (rknn_executor.py:16-18)
            print('Init runtime environment failed')
            raise SystemExit(f"unet_hip init failed: {e}")
        print('done')
        self.target = target
        self.rknn = self.model  # attribute name the reference uses for "is it alive"
        # {input shape: (graph, device input, device probs, pinned input, pinned output)}.  Graphs point into the model's
        # workspace, which only ever grows and does so with the pixel count of a batch: the cache is dropped when a batch
        # larger than anything seen so far comes in (`_ws_px`), and kept otherwise, so a caller alternating between shapes
        # replays instead of re-capturing
        self._graphs = {}
        self._ws_px = 0
        self._lock = threading.Lock()   # run() owns the pinned buffers and the graphs: one caller at a time
        # range fallback (see RANGE_RETRY_ENV): frames still to serve from the fp32 tier, and the current wait
        self._retry_base = max(0, int(os.environ.get(RANGE_RETRY_ENV, RANGE_RETRY_FRAMES)))
        self._retry_wait = self._retry_base
        self._fp32_frames_left = 0
        self._clean_x3_frames = 0       # clean f16x3 frames since the last return to the tier
        self.range_fallbacks = 0        # how often the f16x3 tier was left (diagnostics, tests)
        self._use_graph = os.environ.get(GRAPH_ENV, "1") != "0" and self.precision != "int8"

    def _captured(self, shape):
        """The replayable forward for host batches of `shape`, captured on first use; None if capture is not possible
        (the direct path then serves the call: same kernels, launched one by one)."""
        g = self._graphs.get(shape)
        if g is not None or not self._use_graph:
            return g
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # a process group's watchdog thread polls events from another thread, which stream capture does not
            # tolerate: inside a distributed job the kernels are launched directly
            self._use_graph = False
            return None
        dev = self.model.device
        self._note_pixels(shape[0] * shape[1] * shape[2])
        if len(self._graphs) >= GRAPH_MAX_SHAPES:
            self._graphs.pop(next(iter(self._graphs)))
        try:
            gin = torch.zeros(shape, dtype=torch.uint8, device=dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):       # warm-up: workspace, kernel attributes, zero pages all exist afterwards
                for _ in range(2):
                    self.model.run_u8(gin, return_probs=True, precision=self.precision)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                _, probs = self.model.run_u8(gin, return_probs=True, precision=self.precision)
            hin = torch.empty(shape, dtype=torch.uint8).pin_memory()
            hout = torch.empty(tuple(probs.shape), dtype=torch.float32).pin_memory()
            g = (graph, gin, probs, hin, hout)
            self._graphs[shape] = g
            return g
        except Exception as e:   # noqa: BLE001 - any capture failure: serve directly from now on
            print(f"unet_hip: HIP graph capture unavailable ({e}); launching kernels directly")
            self._use_graph = False
            self._graphs.clear()
            return None

    def _note_pixels(self, px):
        """A batch of `px` pixels is about to run: if it is the largest so far the workspace is going to be re-allocated
        and every captured graph points into the old one."""
        if px > self._ws_px:
            self._graphs.clear()
            self._ws_px = px

    def _device_status(self, what):
        """Status of everything launched since the last check (synchronises).  -> True when the frames have to be run
        again because the f16x3 tier left its range and the tier was chosen automatically (the container is then on
        the fp32 tier); raises on any other failure (the caller's predict() turns that into a zero mask)."""
        # everything this container launches goes to torch's current stream of the device: waiting for that stream is
        # enough, and does not stall other streams (a camera stage, a second model)
        rc = self.model.device_error(current_stream_only=hasattr(self.model, "_stream"))
        if rc == 0:
            return False
        if rc == UNET_ERR_RANGE and self.precision == "f16x3" and self._auto_tier:
            again = f"for the next {self._retry_wait} frames" if self._retry_wait else "from now on"
            print(f"unet_hip: an activation left the fp16 range of the f16x3 tier; re-running on the fp32 tier ({again})")
            self._set_tier("fp32")
            self.range_fallbacks += 1
            self._fp32_frames_left = self._retry_wait
            self._retry_wait = min(2 * self._retry_wait, RANGE_RETRY_MAX)
            return True
        raise RuntimeError(f"unet_hip {what} failed on the device (status {rc})")

    def _set_tier(self, tier):
        """Switch the arithmetic tier.  Each tier has its own workspace in the library (the f16x3 planes, the fp32
        tensors), each growing with the largest batch IT has seen: the captured graphs point into the old tier's, and the
        high-water mark that decides when a workspace is about to be re-allocated starts again for the new one."""
        self.precision = tier
        self._graphs.clear()
        self._ws_px = 0

    def _served(self, nframes):
        """`nframes` frames came back clean.  On the fp32 tier after a range fallback: count them down and return to
        the f16x3 tier when the wait is over."""
        if not self._auto_tier:
            return
        if self.precision == "fp32" and self._fp32_frames_left > 0:
            self._fp32_frames_left -= nframes
            if self._fp32_frames_left <= 0:
                self._fp32_frames_left = 0
                self._clean_x3_frames = 0
                self._set_tier("f16x3")
        elif self.precision == "f16x3" and self._retry_wait != self._retry_base:
            # as many clean frames on the tier as the last wait was long: the next report starts from the base wait again
            self._clean_x3_frames += nframes
            if self._clean_x3_frames >= self._retry_wait:
                self._retry_wait = self._retry_base

    def run(self, inputs):
        with self._lock:
            return self._run(inputs)

    def _run(self, inputs):
        if self.rknn is None:
            print("ERROR: rknn has been released")
            return []
        x = inputs[0] if isinstance(inputs, (list, tuple)) else inputs   # a bare array counts as [array]
        if torch.is_tensor(x):
            frames = x
        else:
            x = np.asarray(x)
            if x.ndim == 3:
                x = x[None]
            if x.dtype != np.uint8:
                x = x.astype(np.uint8)  # the caller keeps uint8 for the quantised blob (src/unet.py:36-37)
            x = np.ascontiguousarray(x)
            for _ in range(2):   # second pass: the same frames on the fp32 tier after a range report
                g = self._captured(tuple(x.shape)) if x.ndim == 4 and x.shape[0] <= GRAPH_MAX_FRAMES and x.shape[-1] == 3 else None
                if g is None:
                    break
                graph, gin, probs, hin, hout = g
                hin.copy_(torch.from_numpy(x))
                gin.copy_(hin, non_blocking=True)
                graph.replay()
                hout.copy_(probs, non_blocking=True)
                if not self._device_status("inference"):   # synchronises
                    res = [hout.numpy().copy()]
                    self._served(int(x.shape[0]))
                    return res
            frames = torch.from_numpy(x)
        # the direct path may grow the model's workspace: graphs captured against the old one must not be replayed
        if frames.dim() == 4:
            self._note_pixels(int(frames.shape[0]) * int(frames.shape[1]) * int(frames.shape[2]))
        else:
            self._graphs.clear()
        frames = frames.to(self.model.device, non_blocking=True)
        if self.precision == "int8":
            _, probs = self.model.run_u8(frames, return_probs=True)
            return [probs.cpu().numpy()]
        for _ in range(2):
            _, probs = self.model.run_u8(frames, return_probs=True, precision=self.precision)
            out = probs.cpu().numpy()
            if not self._device_status("inference"):
                self._served(int(frames.shape[0]) if frames.dim() == 4 else 1)
                return [out]
        raise RuntimeError("unet_hip inference failed on the device (range report on the fp32 tier)")

    def release(self):
        if self.rknn is not None:
            self._graphs.clear()
            self.rknn.release()
        self.rknn = None
