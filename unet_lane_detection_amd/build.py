"""Compile the HIP library in-tree: unet_lane_detection_amd/csrc/libunet_hip.so.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build
container and the resulting .so travels to the GPU box with the snapshot.

Staleness is decided by content, not by mtimes: the SHA-256 over every source
the library is built from (csrc/*.cpp, *.h, *.inc, include/unet_hip.h), the
compiler flags and the hipcc version string is stored next to the .so in
libunet_hip.so.srchash; the library is rebuilt whenever that digest differs.
On a box without hipcc (nothing to rebuild with) a shipped .so whose digest
matches is used as is and one that does not match is an error.
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, "libunet_hip.so")
HASHFILE = LIB + ".srchash"
SOURCES = ["unet_hip.cpp"]
FLAGS = ["-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
         "-Wno-unused-value"]


def hipcc_path(required: bool = True):
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    if required:
        raise RuntimeError("hipcc not found (set HIPCC)")
    return None


def dependency_files():
    deps = []
    for pat in ("*.cpp", "*.h", "*.inc"):
        deps += glob.glob(os.path.join(CSRC, pat))
    deps.append(os.path.join(ROOT, "include", "unet_hip.h"))
    return sorted(deps)


def source_digest() -> str:
    h = hashlib.sha256()
    for path in dependency_files():
        h.update(os.path.relpath(path, ROOT).encode())
        with open(path, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    h.update(" ".join(FLAGS + os.environ.get("UNET_HIPCC_FLAGS", "").split()).encode())
    return h.hexdigest()


def _stored_digest():
    try:
        with open(HASHFILE) as f:
            return f.read().strip()
    except OSError:
        return None


def is_stale() -> bool:
    return not os.path.exists(LIB) or _stored_digest() != source_digest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    cc = hipcc_path(required=False)
    if cc is None:
        raise RuntimeError(f"{LIB} is missing or does not match the sources, and there is no hipcc to rebuild it")
    cmd = [cc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    cmd += os.environ.get("UNET_HIPCC_FLAGS", "").split()   # e.g. -DUNET_WS_STAMPS=1 (diagnostic builds only)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    with open(HASHFILE, "w") as f:
        f.write(source_digest() + "\n")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
