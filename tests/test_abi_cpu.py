"""CPU-side checks of the C ABI: the library loads, exports every symbol the header declares,
and its host logic (spec enumeration, shape checks) behaves.  No kernels run here."""
import ctypes as C
import os
import re

import pytest

from unet_lane_detection_amd import _lib, state as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "unet_hip.h")).read()
    declared = set(re.findall(r"\b(unet_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/unet_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_version_string(lib):
    assert b"gfx950" in lib.unet_version()


def test_invalid_create_arguments(lib):
    h = C.c_void_p()
    assert lib.unet_create(None, C.byref(h)) == 1
    cfg = _lib.UnetConfig()
    cfg.in_channels, cfg.out_channels, cfg.depth = 3, 1, 0
    assert lib.unet_create(C.byref(cfg), C.byref(h)) == 1
    cfg.depth = 2
    cfg.features[0], cfg.features[1] = 6, 8          # channels must be multiples of 4
    assert lib.unet_create(C.byref(cfg), C.byref(h)) == 1


def test_param_spec_matches_reference_state_dict(lib):
    """The library expects exactly the reference module's float state_dict keys and sizes."""
    import numpy as np
    cfg = _lib.UnetConfig()
    cfg.in_channels, cfg.out_channels, cfg.depth = 3, 1, 4
    for i, f in enumerate(S.DEFAULT_FEATURES):
        cfg.features[i] = f
    h = C.c_void_p()
    assert lib.unet_create(C.byref(cfg), C.byref(h)) == 0
    n = lib.unet_num_params(h)
    got = {lib.unet_param_name(h, i).decode(): lib.unet_param_numel(h, i) for i in range(n)}
    want = {k: int(np.prod(shape)) if shape else 1 for k, shape, kind in S.state_dict_spec() if kind != "bn_count"}
    assert got == want
    assert sum(v for k, v in got.items() if "running_" not in k) == 31_037_633   # reference README.md:2288
    # wrong size and unknown name are rejected with the documented codes
    buf = (C.c_float * 4)()
    assert lib.unet_load_param(h, b"output.bias", buf, 4) == 2
    assert lib.unet_load_param(h, b"nope.weight", buf, 4) == 6
    assert lib.unet_finalize(h) == 3                   # parameters missing
    assert lib.unet_workspace_bytes(h, 1, 225, 224) == 0   # 225 is not a multiple of 16
    assert lib.unet_workspace_bytes(h, 256, 224, 224) > 10 * 2**30
    assert lib.unet_destroy(h) == 0


def test_act_scale_of_nearly_dead_bn_channel(lib):
    """f16x3 tier, host logic (csrc/unet_x3.inc, act_from_bn): the per-channel power-of-two activation scale puts
    4 |gamma| + |beta| into [512, 1024) - and must stay finite for a channel that is almost, but not exactly, dead
    (the unclamped 2^(10 - e) was +inf below 2^-118: inf / NaN scale and shift, a range report on every frame)."""
    import math
    for gamma, beta in ((1.0, 0.0), (0.02, -0.3), (3e-5, 0.0), (250.0, 10.0)):
        s = lib.unet_debug_act_scale(gamma, beta)
        m = (4 * abs(gamma) + abs(beta)) * s
        assert 512.0 <= m < 1024.0 and math.log2(s) == int(math.log2(s)), (gamma, beta, s)
    assert lib.unet_debug_act_scale(0.0, 0.0) == 1.0                    # a dead channel is stored as it is
    for gamma, beta in ((1e-38, 0.0), (0.0, 1e-40), (3e-36, 1e-37)):    # nearly dead: finite, no scaling
        s = lib.unet_debug_act_scale(gamma, beta)
        assert math.isfinite(s) and s == 1.0, (gamma, beta, s)
    s = lib.unet_debug_act_scale(1e-20, 0.0)                            # tiny but alive: clamped to 2^40
    assert s == 2.0 ** 40
    assert lib.unet_debug_act_scale(1e30, 0.0) == 2.0 ** -40
