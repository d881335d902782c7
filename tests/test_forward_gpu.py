"""End-to-end parity of the HIP forward path against golden vectors produced by the reference's
own UNet class (tests/golden/make_golden.py) and against the CPU oracle.

Tolerances (BASELINE.json north_star): pre-sigmoid logits within 1e-3 in fp32 (we assert 2e-4);
binary mask identical wherever |logit| exceeds the logit tolerance; mask IoU >= 1 - 1e-4.
"""
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O
from unet_lane_detection_amd import state as S

pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-4


@pytest.fixture(scope="module", params=[1, 0], ids=["winograd", "direct"], autouse=True)
def conv_algo(request):
    """Every test of this module runs under both 3x3 algorithms (Winograd F(2x2,3x3) and direct implicit GEMM)."""
    from unet_lane_detection_amd import _lib
    lib = _lib.load(build_if_missing=False)
    prev = lib.unet_set_winograd(request.param)
    yield request.param
    lib.unet_set_winograd(prev)


@pytest.fixture(scope="module")
def modelA(conv_algo):
    from unet_lane_detection_amd.model import UNetHIP
    m = UNetHIP(S.seeded_state_dict(seed=0), device=0)
    yield m
    m.release()


def test_tiny_config_matches_reference_golden(golden_dir):
    from unet_lane_detection_amd.model import UNetHIP
    g = np.load(os.path.join(golden_dir, "tiny_f4_8_eval.npz"))
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd/")}
    m = UNetHIP(sd, device=0)
    y = m.forward(torch.from_numpy(g["input"]).cuda()).cpu().numpy()
    m.release()
    assert y.shape == g["logits"].shape
    assert np.abs(y - g["logits"]).max() < 5e-5


def test_modelA_reference_frame(modelA, golden_dir):
    g = np.load(os.path.join(golden_dir, "modelA_frame_001410.npz"))
    frame = np.fromfile(os.path.join(golden_dir, "frame_001410_rgb_u8.bin"), dtype=np.uint8).reshape(1, 224, 224, 3)
    logits, probs, mask = modelA.run_u8(torch.from_numpy(frame).cuda(), return_probs=True, return_mask=True)
    logits = logits.cpu().numpy()[0, 0]
    err = np.abs(logits - g["logits"]).max()
    assert err < LOGIT_TOL, err
    mask = mask.cpu().numpy()[0]
    sure = np.abs(g["logits"]) > LOGIT_TOL
    assert np.array_equal(mask[sure], g["mask"][sure])          # bit-exact mask away from ties
    assert O.mask_iou(mask, g["mask"]) >= 1 - 1e-4
    p = probs.cpu().numpy()[0, 0]
    assert np.abs(p - 1 / (1 + np.exp(-g["logits"].astype(np.float64)))).max() < 1e-5


def test_modelA_synthetic_frames_vs_reference(modelA, golden_dir):
    g = np.load(os.path.join(golden_dir, "modelA_synth2.npz"))
    frames = torch.from_numpy(S.synthetic_frames(2, seed=0)).cuda()
    logits = modelA.run_u8(frames).cpu().numpy()[:, 0]
    assert np.abs(logits - g["logits"]).max() < LOGIT_TOL


def test_forward_f32_nchw_equals_u8_path(modelA):
    frames = S.synthetic_frames(3, seed=5)
    a = modelA.run_u8(torch.from_numpy(frames).cuda())
    b = modelA.forward(O.normalize_u8_nhwc(frames).cuda())
    assert (a - b).abs().max().item() < 1e-4


def test_batch_independence_and_oracle(modelA):
    """Frames are independent: a frame's logits do not depend on its batch position or batch size,
    and match the CPU oracle run on that frame alone."""
    frames = S.synthetic_frames(5, seed=9)
    full = modelA.run_u8(torch.from_numpy(frames).cuda()).cpu()
    one = modelA.run_u8(torch.from_numpy(frames[3:4]).cuda()).cpu()
    assert torch.equal(full[3:4], one) or (full[3:4] - one).abs().max().item() < 1e-5
    sd = O.to_torch_state(S.seeded_state_dict(seed=0))
    with torch.no_grad():
        ref = O.forward(sd, O.normalize_u8_nhwc(frames[3:4]))
    assert (one - ref).abs().max().item() < LOGIT_TOL


def test_other_resolutions(golden_dir):
    """H, W only need to be multiples of 2^depth: 64x96 and 160x160 on a 3-level model."""
    from unet_lane_detection_amd.model import UNetHIP
    feats = [16, 32, 64]
    sdn = S.seeded_state_dict(feats, seed=4)
    m = UNetHIP(sdn, device=0)
    sd = O.to_torch_state(sdn)
    for (n, h, w) in [(2, 64, 96), (1, 160, 160), (3, 8, 8)]:
        x = torch.randn(n, 3, h, w, generator=torch.Generator().manual_seed(h))
        with torch.no_grad():
            ref = O.forward(sd, x)
        got = m.forward(x.cuda()).cpu()
        assert (got - ref).abs().max().item() < LOGIT_TOL, (n, h, w)
    with pytest.raises(Exception):
        m.forward(torch.zeros(1, 3, 20, 24).cuda())   # 20 is not a multiple of 8
    m.release()


def test_container_drop_in(golden_dir):
    """RKNN_model_container semantics (reference src/py_utils/rknn_executor.py:4-42) and the
    caller-side post-processing of src/unet.py:44-72 on top of it."""
    from unet_lane_detection_amd.py_utils.rknn_executor import RKNN_model_container
    g = np.load(os.path.join(golden_dir, "modelA_frame_001410.npz"))
    frame = np.fromfile(os.path.join(golden_dir, "frame_001410_rgb_u8.bin"), dtype=np.uint8).reshape(1, 224, 224, 3)
    c = RKNN_model_container("seed:0", "rk3588", "0")
    assert c.precision == "f16x3"                             # auto: the fastest tier that meets the fp32 parity bar
    out = c.run(inputs=[frame])
    assert isinstance(out, list) and out[0].shape == (1, 1, 224, 224) and out[0].dtype == np.float32
    assert out[0].min() >= 0.0 and out[0].max() <= 1.0       # probabilities: the caller's guard stays inert
    out2 = c.run(frame)                                        # bare array is wrapped (rknn_executor.py:31-34)
    assert np.array_equal(out[0], out2[0])
    mask = O.postprocess_output(out)
    sure = np.abs(g["logits"]) > LOGIT_TOL
    assert np.array_equal(mask[sure], g["mask"][sure])
    c.release()
    assert c.run([frame]) == []                                # rknn_executor.py:27-29
    c.release()                                                # idempotent (src/unet.py:148-150)


def test_container_graph_replay_matches_direct_launches(monkeypatch):
    """Small host batches are served by replaying a captured HIP graph of the forward pass (one graph per input shape).
    The replay must give the bits the one-by-one launches give, across shape changes (each re-captures; a larger shape
    may re-allocate the workspace) and after a large batch has gone through the direct path in between."""
    from unet_lane_detection_amd.py_utils.rknn_executor import RKNN_model_container, GRAPH_MAX_FRAMES
    frames = S.synthetic_frames(GRAPH_MAX_FRAMES + 8, seed=5)
    monkeypatch.setenv("UNET_HIP_GRAPH", "0")
    direct = RKNN_model_container("seed:0", "rk3588", "0")
    assert not direct._use_graph
    monkeypatch.setenv("UNET_HIP_GRAPH", "1")
    c = RKNN_model_container("seed:0", "rk3588", "0")
    assert c._use_graph
    for lo, hi in [(0, 1), (1, 2), (0, 3), (3, 4), (0, GRAPH_MAX_FRAMES + 8), (4, 5), (2, 2 + GRAPH_MAX_FRAMES), (5, 6)]:
        a = c.run([frames[lo:hi]])[0]
        b = direct.run([frames[lo:hi]])[0]
        assert a.shape == (hi - lo, 1, 224, 224)
        assert np.array_equal(a, b), (lo, hi)
        if hi - lo <= GRAPH_MAX_FRAMES:
            assert c._use_graph and tuple(frames[lo:hi].shape) in c._graphs, "the graph path did not serve this call"
    c.release()
    direct.release()


def test_container_keeps_a_graph_per_shape():
    """A caller alternating between shapes replays instead of re-capturing: graphs are kept per input shape and dropped
    only when a batch larger than any before comes in (the workspace they point into is then re-allocated)."""
    from unet_lane_detection_amd.py_utils.rknn_executor import RKNN_model_container
    frames = S.synthetic_frames(6, seed=9)
    c = RKNN_model_container("seed:0", "rk3588", "0")
    k4, k1, k6 = (4, 224, 224, 3), (1, 224, 224, 3), (6, 224, 224, 3)
    a4 = c.run([frames[:4]])[0]
    g4 = c._graphs[k4][0]
    a1 = c.run([frames[:1]])[0]                     # a smaller batch: the 4-frame graph stays
    assert k4 in c._graphs and k1 in c._graphs
    b4 = c.run([frames[:4]])[0]
    assert c._graphs[k4][0] is g4 and np.array_equal(a4, b4)
    assert np.array_equal(c.run([frames[:1]])[0], a1)
    c.run([frames[:6]])                             # the largest so far: everything captured before is dropped
    assert k6 in c._graphs and k4 not in c._graphs and k1 not in c._graphs
    assert np.array_equal(c.run([frames[:4]])[0], a4)
    c.release()


def test_config5_640x640_frame(modelA):
    """BASELINE.json configs[4]: 640x640 input (large-input path); one frame against the CPU oracle."""
    frames = S.synthetic_frames(1, 640, 640, seed=21)
    sd = O.to_torch_state(S.seeded_state_dict(seed=0))
    with torch.no_grad():
        ref = O.forward(sd, O.normalize_u8_nhwc(frames))
    got = modelA.run_u8(torch.from_numpy(frames).cuda()).cpu()
    assert (got - ref).abs().max().item() < LOGIT_TOL


def test_model_B_configuration():
    """The deployed blob's architecture (3 levels, base 32, sigmoid fused into the
    head, ~1.92 M parameters) is a configuration of the same code path: features=[32,64,128] + probabilities."""
    from unet_lane_detection_amd.model import UNetHIP
    feats = [32, 64, 128]
    n_params = S.num_parameters(feats)
    assert 1.90e6 < n_params < 1.95e6, n_params
    sdn = S.seeded_state_dict(feats, seed=11)
    m = UNetHIP(sdn, device=0)
    frames = S.synthetic_frames(2, seed=17)
    logits, probs = m.run_u8(torch.from_numpy(frames).cuda(), return_probs=True)
    with torch.no_grad():
        ref = O.forward(O.to_torch_state(sdn), O.normalize_u8_nhwc(frames))
    assert (logits.cpu() - ref).abs().max().item() < LOGIT_TOL
    p = probs.cpu()
    assert p.min().item() >= 0.0 and p.max().item() <= 1.0       # blob metadata: output range [1.28e-6, 1.0]
    assert (p - torch.sigmoid(ref)).abs().max().item() < 1e-5
    m.release()


def test_error_paths_and_edge_shapes():
    """Boundary behaviour at the Python mirror: shape errors raise with the C ABI's code, release is idempotent,
    calls after release raise, the smallest legal input (16x16, N=1) and a ragged batch (N=7) work."""
    from unet_lane_detection_amd import _lib
    from unet_lane_detection_amd.model import UNetHIP
    m = UNetHIP(S.seeded_state_dict(seed=0), device=0)
    with pytest.raises(_lib.UnetError) as e:
        m.run_u8(torch.zeros((1, 225, 224, 3), dtype=torch.uint8).cuda())      # 225 not a multiple of 16
    assert e.value.code == 2
    with pytest.raises(ValueError):
        m.run_u8(torch.zeros((1, 224, 224, 4), dtype=torch.uint8).cuda())      # not RGB
    with pytest.raises(ValueError):
        m.run_u8(torch.zeros((224, 224, 3), dtype=torch.uint8).cuda())         # no batch dimension
    sd = O.to_torch_state(S.seeded_state_dict(seed=0))
    for (n, h, w) in [(1, 16, 16), (7, 32, 48)]:
        frames = S.synthetic_frames(n, h, w, seed=n)
        with torch.no_grad():
            ref = O.forward(sd, O.normalize_u8_nhwc(frames))
        got = m.run_u8(torch.from_numpy(frames).cuda()).cpu()
        assert (got - ref).abs().max().item() < LOGIT_TOL, (n, h, w)
    m.release()
    m.release()                                                                 # idempotent
    with pytest.raises(RuntimeError):
        m.run_u8(torch.zeros((1, 224, 224, 3), dtype=torch.uint8).cuda())
