"""Compile the HIP library in-tree: unet_lane_detection_amd/csrc/libunet_hip.so.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build
container and the resulting .so travels to the GPU box with the snapshot.

Staleness is decided by content, not by mtimes: the SHA-256 over every source
the library is built from (csrc/*.cpp, *.h, *.inc, include/unet_hip.h) and the
compiler flags, plus the first line of `hipcc --version`, are stored next to
the .so in libunet_hip.so.srchash; the library is rebuilt whenever either
differs.  On a box without hipcc (nothing to rebuild with) a shipped .so whose
source digest matches is used as is and one that does not match is an error.
The build runs under a file lock into a temporary file that replaces the
library atomically, so concurrent ranks never load a half-written file.
"""
from __future__ import annotations

import fcntl
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


_HIPCC_VERSION = None


def hipcc_version() -> str:
    """The "HIP version: ..." line of `hipcc --version` (part of the digest: a library built by another compiler is stale); empty on a
    box without hipcc, where the shipped library's recorded digest is compared with the same empty string left out."""
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        cc = hipcc_path(required=False)
        v = ""
        if cc:
            try:
                out = subprocess.run([cc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
                lines = out.stdout.strip().splitlines() or [""]
                # the "HIP version:" line, not simply the first one: under rocprofv3 the profiler's preloaded tool
                # prints its own log lines ahead of hipcc's
                v = next((ln for ln in lines if ln.startswith("HIP version")), lines[0])
            except (OSError, subprocess.SubprocessError):
                v = ""
        _HIPCC_VERSION = v
    return _HIPCC_VERSION


def source_digest() -> str:
    h = hashlib.sha256()
    for path in dependency_files():
        h.update(os.path.relpath(path, ROOT).encode())
        with open(path, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    h.update(" ".join(FLAGS + os.environ.get("UNET_HIPCC_FLAGS", "").split()).encode())
    return h.hexdigest()


def _stored():
    """(source digest, compiler version line) recorded next to the library, or (None, None)"""
    try:
        with open(HASHFILE) as f:
            lines = f.read().splitlines()
        return (lines[0].strip() if lines else None), (lines[1].strip() if len(lines) > 1 else "")
    except OSError:
        return None, None


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    digest, compiler = _stored()
    if digest != source_digest():
        return True
    # the compiler is only compared where there is one to rebuild with (the GPU box may lack hipcc)
    return bool(hipcc_path(required=False)) and compiler != hipcc_version()


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    cc = hipcc_path(required=False)
    if cc is None:
        raise RuntimeError(f"{LIB} is missing or does not match the sources, and there is no hipcc to rebuild it")
    # several ranks may get here at once (bench.py --gpus N, tests/dp_rehearsal.py): one builds, into a temporary file
    # that replaces the library atomically; the others wait on the lock and find it fresh
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale():
                return LIB
            tmp = f"{LIB}.tmp.{os.getpid()}"
            cmd = [cc] + FLAGS + ["-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
            cmd += os.environ.get("UNET_HIPCC_FLAGS", "").split()   # e.g. -DUNET_WS_STAMPS=1 (diagnostic builds only)
            if verbose:
                print(" ".join(cmd), flush=True)
            try:
                subprocess.run(cmd, check=True, cwd=CSRC)
                with open(HASHFILE + ".tmp", "w") as f:
                    f.write(source_digest() + "\n" + hipcc_version() + "\n")
                os.replace(tmp, LIB)
                os.replace(HASHFILE + ".tmp", HASHFILE)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
