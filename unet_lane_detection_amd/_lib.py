"""ctypes binding of libunet_hip.so (C ABI: include/unet_hip.h).

There is no fallback: if the library is missing or does not load, importing a
product path raises.  (The oracle under oracle/ is test infrastructure and is
never reached from here.)
"""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB, build_library, is_stale

UNET_MAX_DEPTH = 6


class UnetConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32),
        ("out_channels", C.c_int32),
        ("depth", C.c_int32),
        ("features", C.c_int32 * UNET_MAX_DEPTH),
        ("device", C.c_int32),
        ("input_mean", C.c_float * 3),
        ("input_std", C.c_float * 3),
    ]


STATUS = {0: "UNET_OK", 1: "UNET_ERR_INVALID_ARG", 2: "UNET_ERR_SHAPE", 3: "UNET_ERR_STATE", 4: "UNET_ERR_HIP",
          5: "UNET_ERR_NOMEM", 6: "UNET_ERR_UNKNOWN_PARAM", 7: "UNET_ERR_RANGE"}
UNET_ERR_HIP = 4     # a HIP runtime call or a kernel-side check failed
UNET_ERR_RANGE = 7   # f16x3 tier: an activation left the fp16 range (include/unet_hip.h); re-run on the fp32 tier

# name -> (restype, argtypes); every symbol include/unet_hip.h declares
SIGNATURES = {
    "unet_create": (C.c_int, [C.POINTER(UnetConfig), C.POINTER(C.c_void_p)]),
    "unet_load_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "unet_finalize": (C.c_int, [C.c_void_p]),
    "unet_num_params": (C.c_int, [C.c_void_p]),
    "unet_param_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "unet_param_numel": (C.c_size_t, [C.c_void_p, C.c_int]),
    "unet_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "unet_reserve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "unet_forward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_float, C.c_void_p]),
    "unet_forward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_float, C.c_void_p]),
    "unet_forward_u8_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_float, C.c_void_p]),
    "unet_forward_u8_x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_float, C.c_void_p]),
    "unet_forward_f32_x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_float, C.c_void_p]),
    "unet_op_conv3x3_x3": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "unet_op_conv3x3_x3_head": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "unet_op_upconv2x2_x3": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "unet_i8_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "unet_i8_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "unet_i8_finalize": (C.c_int, [C.c_void_p]),
    "unet_i8_forward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_float, C.c_void_p]),
    "unet_i8_read_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "unet_i8_destroy": (C.c_int, [C.c_void_p]),
    "unet_i8_last_error": (C.c_char_p, [C.c_void_p]),
    "unet_num_range_tensors": (C.c_int, [C.c_void_p]),
    "unet_forward_u8_ranges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "unet_destroy": (C.c_int, [C.c_void_p]),
    "unet_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "unet_profile_count": (C.c_int, [C.c_void_p]),
    "unet_profile_get": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double),
                                   C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "unet_train_param_numel": (C.c_size_t, [C.c_void_p]),
    "unet_train_buffer_numel": (C.c_size_t, [C.c_void_p]),
    "unet_train_layout": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "unet_train_attach": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "unet_train_set_loss": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]),
    "unet_train_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "unet_train_forward_backward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "unet_train_forward_backward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "unet_train_adam_step": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_int, C.c_float, C.c_void_p]),
    "unet_train_debug_snapshot": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "unet_op_wgrad3x3": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p]),
    "unet_op_wgrad3x3_x3": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_int, C.c_void_p]),
    "unet_train_repack": (C.c_int, [C.c_void_p, C.c_void_p]),
    "unet_set_train_x3": (C.c_int, [C.c_int]),
    "unet_set_train_side": (C.c_int, [C.c_int]),
    "unet_train_set_comm_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "unet_train_grad_split": (C.c_size_t, [C.c_void_p]),
    "unet_dice_metric": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_void_p,
                                   C.c_void_p]),
    "unet_device_error": (C.c_int, [C.c_void_p]),
    "unet_device_error_on": (C.c_int, [C.c_void_p, C.c_void_p]),
    "unet_device_status_to": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "unet_debug_set_error_block": (C.c_int, [C.c_void_p, C.c_int, C.c_uint]),
    "unet_debug_act_scale": (C.c_float, [C.c_float, C.c_float]),
    "unet_last_error": (C.c_char_p, [C.c_void_p]),
    "unet_version": (C.c_char_p, []),
    "unet_set_winograd": (C.c_int, [C.c_int]),
    "unet_set_bf16_persistent": (C.c_int, [C.c_int]),
    "unet_set_x3_upconv_r512": (C.c_int, [C.c_int]),
    "unet_set_x3_cross_fp8": (C.c_int, [C.c_int]),
    "unet_ipm_prestage_u8": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "unet_resize_u8": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p]),
    "unet_op_conv3x3": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "unet_op_upconv2x2": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_void_p, C.c_void_p]),
    "unet_op_conv1x1": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_void_p]),
    "unet_op_maxpool2x2": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "unet_op_head1x1": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float,
                                  C.c_void_p, C.c_void_p]),
}

_lib = None


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load (building first if the .so is absent and hipcc is available)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64; it must be the first HIP runtime in the process, or a second
    # copy pulled in through libunet_hip.so's RPATH-less dependency fails to see the device.
    import torch  # noqa: F401
    path = os.environ.get("UNET_HIP_LIB", LIB)   # override: A/B timing builds of the same ABI
    if path == LIB and is_stale():
        # content hash of the sources differs from the one the .so was built from (or there is no .so)
        if not build_if_missing:
            raise RuntimeError(f"{LIB} is missing or older than its sources; run `python -m unet_lane_detection_amd.build`")
        build_library()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class UnetError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: {STATUS.get(code, code)}" + (f" ({detail})" if detail else ""))


def check(code, where, handle=None):
    if code != 0:
        detail = ""
        if handle:
            msg = load().unet_last_error(handle)
            detail = msg.decode() if msg else ""
        raise UnetError(code, where, detail)
