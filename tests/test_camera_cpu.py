"""Camera-stage oracle (oracle/camera_oracle.py) on the CPU: properties and hand-computed values of the restated
OpenCV 8-bit algorithms, and the host-side perspective solve of the product.  cv2 is not installed and the
reference holds no fixture for this stage: parity against cv2 itself is unpinned."""
import numpy as np

from oracle import camera_oracle as CO
from unet_lane_detection_amd import ros_bridge as RB


def test_perspective_transform_maps_the_reference_points():
    m = CO.get_perspective_transform(RB.REF_SRC_POINTS, RB.REF_DST_POINTS)
    assert np.allclose(m, RB.get_perspective_transform(RB.REF_SRC_POINTS, RB.REF_DST_POINTS), rtol=0, atol=1e-12)
    for (x, y), (u, v) in zip(RB.REF_SRC_POINTS, RB.REF_DST_POINTS):
        p = m @ np.array([x, y, 1.0])
        assert np.allclose(p[:2] / p[2], (u, v), atol=1e-9)
    assert m[2, 2] == 1.0


def test_identity_warp_and_same_size_resize_are_copies():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    assert np.array_equal(CO.warp_perspective(img, np.eye(3), 53, 37), img)
    assert np.array_equal(CO.resize_linear(img, 53, 37), img)
    # integer shift: destination (x, y) reads source (x - 5, y - 3); outside is the constant border 0
    shift = np.array([[1, 0, 5], [0, 1, 3], [0, 0, 1]], dtype=np.float64)
    w = CO.warp_perspective(img, shift, 53, 37)
    assert np.array_equal(w[3:, 5:], img[:-3, :-5]) and not w[:3].any() and not w[:, :5].any()


def test_half_pixel_shift_rounds_like_the_fixed_point_table():
    img = np.array([[[10], [20], [31]]], dtype=np.uint8)   # one row
    half = np.array([[1, 0, 0.5], [0, 1, 0], [0, 0, 1]], dtype=np.float64)   # dst x reads src x - 0.5
    w = CO.warp_perspective(img, half, 3, 1)[0, :, 0]
    # x=0: (0*16 + 10*16)*32 ... = 5 ; x=1: 15 ; x=2: (20+31)/2 = 25.5 -> (51*16*32 + 16384) >> 15 = 26
    assert list(w) == [5, 15, 26]


def test_resize_hand_computed():
    src = np.array([[0, 255]], dtype=np.uint8)
    out = CO.resize_linear(src, 4, 1)[0]
    # dx=1: f = 0.25 -> coefficients 1536 / 512 -> D = 130560 -> ((2048 * (D >> 4)) >> 16 + 2) >> 2 = 64
    assert list(out) == [0, 64, 191, 255]
    const = np.full((5, 7, 3), 200, dtype=np.uint8)
    assert (CO.resize_linear(const, 31, 17) == 200).all() and (CO.resize_linear(const, 3, 2) == 200).all()


def test_prestage_shapes_and_channel_order():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
    m = CO.get_perspective_transform(RB.REF_SRC_POINTS, RB.REF_DST_POINTS)
    a = CO.prestage(img, m, 1055, 685, 224, 224, bgr_in=True)
    b = CO.prestage(img[..., ::-1].copy(), m, 1055, 685, 224, 224, bgr_in=False)
    assert a.shape == (224, 224, 3) and a.dtype == np.uint8 and np.array_equal(a, b)
    back = CO.poststage((a[..., 0] > 127).astype(np.uint8) * 255, 1055, 685)
    assert back.shape == (685, 1055)
