"""Camera stage on the GPU (csrc/camera_stage.h through the C ABI) against oracle/camera_oracle.py: bit-exact.
The oracle restates OpenCV's 8-bit algorithms; parity against cv2 itself is unpinned (see its header)."""
import numpy as np
import pytest
import torch

from oracle import camera_oracle as CO
from oracle import unet_oracle as O
from unet_lane_detection_amd import ros_bridge as RB
from unet_lane_detection_amd import state as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stage():
    return RB.CameraStage(0)


@pytest.mark.parametrize("shape,enc", [((480, 640), "bgr8"), ((376, 672), "rgb8"), ((720, 1280), "bgr8")])
def test_prestage_reference_calibration(stage, shape, enc):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, size=shape + (3,), dtype=np.uint8)
    m = RB.get_perspective_transform(RB.REF_SRC_POINTS, RB.REF_DST_POINTS)
    ref = CO.prestage(img, m, 1055, 685, 224, 224, bgr_in=(enc == "bgr8"))
    got = stage.prestage(torch.from_numpy(img), m, (1055, 685), (224, 224), encoding=enc).cpu().numpy()
    assert np.array_equal(got, ref)


def test_prestage_without_resize_and_other_sizes(stage):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, size=(120, 160, 3), dtype=np.uint8)
    m = RB.get_perspective_transform(((5, 100), (150, 110), (40, 30), (120, 35)), ((20, 90), (110, 90), (20, 10), (110, 10)))
    # warp size == output size: cv2.resize is a copy, the result is the warp itself
    ref = CO.prestage(img, m, 131, 97, 131, 97, bgr_in=True)
    got = stage.prestage(torch.from_numpy(img), m, (131, 97), (131, 97)).cpu().numpy()
    assert np.array_equal(got, ref)
    assert np.array_equal(got, CO.warp_perspective(img, m, 131, 97)[..., ::-1])
    # up-scaling resize after the warp
    ref = CO.prestage(img, m, 64, 48, 200, 150, bgr_in=False)
    got = stage.prestage(torch.from_numpy(img), m, (64, 48), (200, 150), encoding="rgb8").cpu().numpy()
    assert np.array_equal(got, ref)
    with pytest.raises(ValueError):
        stage.prestage(torch.from_numpy(img), m, (64, 48), encoding="mono8")


@pytest.mark.parametrize("src,dst", [((224, 224), (1055, 685)), ((685, 1055), (224, 224)), ((31, 17), (31, 17)),
                                      ((5, 9), (64, 3))])
def test_resize_u8(stage, src, dst):
    rng = np.random.default_rng(src[0] + dst[0])
    mask = (rng.random(src) > 0.7).astype(np.uint8) * 255
    got = stage.resize(torch.from_numpy(mask), dst).cpu().numpy()
    assert np.array_equal(got, CO.resize_linear(mask, dst[0], dst[1]))
    rgb = rng.integers(0, 256, size=src + (3,), dtype=np.uint8)
    got = stage.resize(torch.from_numpy(rgb), dst).cpu().numpy()
    assert np.array_equal(got, CO.resize_linear(rgb, dst[0], dst[1]))


def test_pipeline_message_to_mask_message():
    """ImageMsg (bgr8 with a padded row pitch) -> mono8 ImageMsg, against oracle pre-stage -> oracle network ->
    threshold -> oracle post-stage.  Pixels whose probability is within 1e-4 of the threshold may differ."""
    from unet_lane_detection_amd.model import UNetHIP
    feats = [8, 16]
    sdn = S.seeded_state_dict(feats, seed=4)
    model = UNetHIP(sdn, device=0)
    pipe = RB.LanePipelineGPU(model, threshold=0.5)
    rng = np.random.default_rng(3)
    h, w, step = 480, 640, 640 * 3 + 16
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    rows = np.zeros((h, step), dtype=np.uint8)
    rows[:, :w * 3] = img.reshape(h, w * 3)
    msg = RB.ImageMsg(height=h, width=w, encoding="bgr8", data=rows.tobytes(), step=step, header="hdr")
    out = pipe.process(msg)
    assert (out.height, out.width, out.encoding, out.step, out.header) == (685, 1055, "mono8", 1055, "hdr")
    got = np.frombuffer(out.data, dtype=np.uint8).reshape(685, 1055)
    frame = CO.prestage(img, pipe.matrix, 1055, 685, 224, 224, bgr_in=True)
    with torch.no_grad():
        logits = O.forward(O.to_torch_state(sdn), O.normalize_u8_nhwc(frame[None]))[0, 0].numpy()
    prob = 1.0 / (1.0 + np.exp(-logits))
    small = (prob > 0.5).astype(np.uint8) * 255
    ref = CO.poststage(small, 1055, 685)
    unsure = CO.poststage((np.abs(prob - 0.5) < 1e-4).astype(np.uint8) * 255, 1055, 685) > 0
    assert np.array_equal(got[~unsure], ref[~unsure])
    with pytest.raises(ValueError):
        pipe.process(RB.ImageMsg(height=h, width=w, encoding="16UC1", data=rows.tobytes(), step=step))
    model.release()


def test_camera_abi_rejects_bad_arguments():
    import ctypes as C
    from unet_lane_detection_amd import _lib
    lib = _lib.load(build_if_missing=False)
    img = torch.zeros((8, 8, 3), dtype=torch.uint8, device="cuda")
    out = torch.zeros((4, 4, 3), dtype=torch.uint8, device="cuda")
    eye = (C.c_double * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1)
    p = lambda t: C.c_void_p(t.data_ptr())
    # row pitch shorter than a row, null image, empty output
    assert lib.unet_ipm_prestage_u8(0, p(img), 8, 8, 8 * 3 - 1, 1, eye, 8, 8, 4, 4, p(out), None) != 0
    assert lib.unet_ipm_prestage_u8(0, None, 8, 8, 24, 1, eye, 8, 8, 4, 4, p(out), None) != 0
    assert lib.unet_ipm_prestage_u8(0, p(img), 8, 8, 24, 1, eye, 8, 8, 0, 4, p(out), None) != 0
    assert lib.unet_resize_u8(0, p(img), 8, 8, 0, 4, 4, p(out), None) != 0
    assert lib.unet_ipm_prestage_u8(0, p(img), 8, 8, 24, 1, eye, 8, 8, 4, 4, p(out), None) == 0
    torch.cuda.synchronize()


def test_persistent_switch_returns_previous_setting():
    from unet_lane_detection_amd import _lib
    lib = _lib.load(build_if_missing=False)
    first = lib.unet_set_bf16_persistent(0)
    assert lib.unet_set_bf16_persistent(1) == 0
    assert lib.unet_set_bf16_persistent(-1) == 1
    assert lib.unet_set_bf16_persistent(first) == -1
