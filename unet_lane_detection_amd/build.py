"""Compile the HIP library in-tree: unet_lane_detection_amd/csrc/libunet_hip.so.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build
container and the resulting .so travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libunet_hip.so")
SOURCES = ["unet_hip.cpp"]
HEADERS = ["igemm_f32.h", "elementwise.h", "train_kernels.h", "wgrad_f32.h", "wgrad_wino_f32.h", "wgrad_gemm_f32.h", "camera_stage.h", "wino_f32.h", "lds_dma.h", "igemm_bf16.h", "conv_bf16_ws.h", "conv_first_bf16x3.h", "upconv_bf16_ws.h", "unet_bf16.inc", "unet_train.inc", os.path.join(ROOT, "include", "unet_hip.h")]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    cmd = [hipcc_path(), "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result", "-Wno-unused-value", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    cmd += os.environ.get("UNET_HIPCC_FLAGS", "").split()   # e.g. -DUNET_WS_STAMPS=1 (diagnostic builds only)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
