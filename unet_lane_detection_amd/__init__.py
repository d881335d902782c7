"""MI355X-native U-Net lane-segmentation path (HIP kernels behind a C ABI).

Public surface:
  UNetHIP                      forward(image) -> logits, run_u8(frames)
  RKNN_model_container         drop-in for the reference's model container
  seeded_state_dict, ...       reproducible weights / synthetic inputs
"""
from .state import (DEFAULT_FEATURES, INPUT_MEAN, INPUT_STD, num_parameters, seeded_state_dict,  # noqa: F401
                    state_dict_spec, synthetic_frames, synthetic_targets)


def __getattr__(name):  # lazy: importing the package must not need torch.cuda or the .so
    if name == "UNetHIP":
        from .model import UNetHIP
        return UNetHIP
    if name == "RKNN_model_container":
        from .py_utils.rknn_executor import RKNN_model_container
        return RKNN_model_container
    raise AttributeError(name)
