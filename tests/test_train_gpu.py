"""Training-step parity: HIP forward(train)/backward/Adam against golden vectors from the reference's
UNet + torch.optim.Adam (tests/golden/make_golden.py) and against the CPU oracle's autograd."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O
from unet_lane_detection_amd import state as S

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(1e-12, np.abs(b).max())


@pytest.mark.parametrize("wino", [1, 0])
@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 4, 8, 16, 16), (1, 16, 32, 14, 14), (3, 64, 64, 28, 28),
                                             (2, 128, 64, 16, 48), (1, 32, 80, 7, 9), (2, 64, 128, 56, 56),
                                             (2, 64, 64, 20, 36), (5, 128, 128, 14, 14), (1, 64, 192, 2, 2),
                                             (3, 64, 64, 6, 10)])
def test_wgrad3x3(n, cin, cout, h, w, wino):
    """Both algorithms: Winograd F(3x3,2x2) (channel counts multiples of 64 on even maps; image boundaries inside
    a tile group, ragged tile grids, maps smaller than a group) and the direct MFMA kernel."""
    from unet_lane_detection_amd import _lib
    lib = _lib.load(build_if_missing=False)
    prev = lib.unet_set_winograd(wino)
    try:
        _wgrad_case(lib, n, cin, cout, h, w)
    finally:
        lib.unet_set_winograd(prev)


def _wgrad_case(lib, n, cin, cout, h, w):
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    dz = torch.randn(n, cout, h, w, generator=g)
    wt = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    O.conv3x3(x, wt).backward(dz)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    dzd = dz.permute(0, 2, 3, 1).contiguous().cuda()
    dw = torch.full((cout, cin, 3, 3), float("nan"), device="cuda")
    rc = lib.unet_op_wgrad3x3(0, C.c_void_p(dzd.data_ptr()), C.c_void_p(xd.data_ptr()), n, h, w, cin, cout,
                              C.c_void_p(dw.data_ptr()), None)
    assert rc == 0
    assert _rel(dw.cpu().numpy(), wt.grad.numpy()) < 2e-5


@pytest.mark.parametrize("scaled,magnitude", [(0, 1.0), (1, 1.0), (1, 3e-8)])
@pytest.mark.parametrize("n,cin,cout,h,w", [(3, 64, 64, 28, 28), (2, 128, 64, 16, 48), (2, 64, 128, 56, 56),
                                             (2, 64, 64, 20, 36), (5, 128, 128, 14, 14), (1, 64, 192, 2, 2),
                                             (3, 64, 64, 6, 10), (1, 64, 64, 224, 224), (7, 256, 128, 4, 17)])
def test_wgrad3x3_f16x3(n, cin, cout, h, w, scaled, magnitude):
    """The split-operand weight-gradient kernel (wgrad_x3_ws.h): widths that are not multiples of the 16-column strip,
    maps smaller than a K-step, more splits than K-steps, gradients far below the fp16 range (scaled path)."""
    from unet_lane_detection_amd import _lib
    lib = _lib.load(build_if_missing=False)
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(n, cin, h, w, generator=g)
    dz = torch.randn(n, cout, h, w, generator=g) * magnitude
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x.double(), wt, padding=1).backward(dz.double())
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    dzd = dz.permute(0, 2, 3, 1).contiguous().cuda()
    dw = torch.full((cout, cin, 3, 3), float("nan"), device="cuda")
    rc = lib.unet_op_wgrad3x3_x3(0, C.c_void_p(dzd.data_ptr()), C.c_void_p(xd.data_ptr()), n, h, w, cin, cout,
                                 C.c_void_p(dw.data_ptr()), scaled, None)
    assert rc == 0
    assert _rel(dw.cpu().numpy(), wt.grad.numpy()) < 2e-5


def _check_grads(tr, ref_grads, loss_ref, gtol):
    assert abs(float(tr.loss.item()) - loss_ref) < 2e-5 * max(1.0, abs(loss_ref))
    got = {k: v.detach().cpu().numpy() for k, v in tr.grad_dict().items()}
    worst = max((_rel(got[k], ref_grads[k]), k) for k in ref_grads)
    assert worst[0] < gtol, worst
    return worst


def test_tiny_train_step_matches_reference_golden(golden_dir):
    from unet_lane_detection_amd.trainer import UNetTrainer
    g = np.load(os.path.join(golden_dir, "tiny_f4_8_train_step.npz"))
    tr = UNetTrainer(S.seeded_state_dict([4, 8], seed=1), device=0, lr=1e-4)
    logits = tr.forward_backward(torch.from_numpy(g["input"]), torch.from_numpy(g["target"]), return_logits=True)
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < 5e-5
    ref_grads = {k[5:]: g[k] for k in g.files if k.startswith("grad/")}
    _check_grads(tr, ref_grads, float(g["loss"]), 5e-4)
    tr.optimizer_step()
    sd = tr.state_dict()
    for k in g.files:
        if not k.startswith("post/"):
            continue
        name = k[5:]
        ref = g[k]
        if name.endswith("num_batches_tracked"):
            assert int(sd[name]) == int(ref)
            continue
        # Adam's first step moves every weight by ~lr*sign(g); the band covers rounding of m/sqrt(v)
        assert np.abs(sd[name].numpy().astype(np.float64) - ref).max() < 3e-6, name
    tr.release()


def _relu_margin(sd, x):
    """Smallest |BatchNorm output| feeding a ReLU in a train-mode forward.  The gradient is discontinuous
    where that value crosses zero, so a pixel sitting within fp32 rounding noise of the kink may take either
    branch on either side (observed: one such pixel moved a bias gradient by 7 %).  Parity of the backward
    pass is only defined away from those ties; the tests pick inputs with a clear margin."""
    taps = {}
    with torch.no_grad():
        O.forward(sd, x, training=True, new_stats={}, taps=taps)
    m = float("inf")
    for k, z in taps.items():
        if not k.startswith("z/"):
            continue
        prefix, conv_i = k[2:].rsplit(".", 1)
        bn = f"{prefix}.{int(conv_i) + 1}"
        mu = z.mean(dim=(0, 2, 3), keepdim=True)
        var = z.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        y = (z - mu) * torch.rsqrt(var + O.BN_EPS) * sd[bn + ".weight"][None, :, None, None] \
            + sd[bn + ".bias"][None, :, None, None]
        m = min(m, float(y.abs().min()))
    return m


def _tie_free_frames(sd_t, n, hh, ww, margin=2e-6, good=5e-6, max_seeds=32):
    """A synthetic batch (seeds 2, 3, ...) whose train-mode forward keeps every BatchNorm output away from the ReLU
    kink: the first seed with a margin above `good`, else the seed with the largest margin among `max_seeds`, which
    must still exceed `margin` (fp32 reassociation noise is ~5e-7 here).  Returns (frames, seeds rejected).  Finding
    none is a hard failure, never a skip: the gradient parity tests must run."""
    best = (-1.0, None, 0)
    for seed in range(2, 2 + max_seeds):
        frames = S.synthetic_frames(n, hh, ww, seed=seed)
        m = _relu_margin(sd_t, O.normalize_u8_nhwc(frames))
        if m > best[0]:
            best = (m, frames, seed - 2)
        if m > good:
            break
    if best[0] <= margin:
        pytest.fail(f"no tie-free input among {max_seeds} seeds (largest margin {best[0]:.2e} <= {margin:.0e})")
    print(f"tie-free input: margin {best[0]:.2e}")
    return best[1], best[2]


@pytest.mark.parametrize("feats,shape", [([16, 32, 64], (3, 24, 32)), ([8, 16], (2, 32, 48)), ([32, 64], (5, 28, 28)),
                                         ([16, 32, 64], (3, 48, 64)),
                                         # widths that are multiples of 64: every 3x3 unit but the first and both
                                         # transposed convolutions on the fp16 split-operand kernels, activations and
                                         # gradients in operand form ("planes mode": heights divisible by 2^(depth+1))
                                         ([64, 128], (2, 32, 48)), ([64, 128, 256], (1, 16, 32)),
                                         ([64, 128], (1, 16, 40)),     # one image, width not a multiple of 16
                                         # ... and the same widths on maps where planes mode does not apply (odd
                                         # bottleneck height: fp32 weight gradient there, per-consumer splits)
                                         ([64, 128], (2, 28, 28))])
def test_mid_config_grads_vs_oracle(feats, shape):
    from unet_lane_detection_amd.trainer import UNetTrainer
    n, hh, ww = shape
    if hh % (1 << len(feats)) or ww % (1 << len(feats)):
        hh, ww = hh // (1 << len(feats)) * (1 << len(feats)), ww // (1 << len(feats)) * (1 << len(feats))
    sdn = S.seeded_state_dict(feats, seed=6)
    sd_t = O.to_torch_state(sdn)
    frames, rejected = _tie_free_frames(sd_t, n, hh, ww)
    print(f"tie-free input: seed {2 + rejected} after {rejected} rejected seeds")
    tgt = torch.from_numpy(S.synthetic_targets(n, hh, ww, seed=2))
    loss, grads, new_stats, logits = O.loss_and_grads(sd_t, O.normalize_u8_nhwc(frames), tgt)
    tr = UNetTrainer(sdn, device=0)
    lg = tr.forward_backward(torch.from_numpy(frames), tgt, return_logits=True)
    assert (lg.cpu() - logits).abs().max().item() < 1e-4
    ref = {k: v.numpy() for k, v in grads.items()}
    _check_grads(tr, ref, float(loss), 1e-3)
    sd = tr.state_dict()
    for k, v in new_stats.items():
        if k.endswith("num_batches_tracked"):
            continue
        assert _rel(sd[k].numpy(), v.numpy()) < 1e-5, k
    tr.release()


@pytest.fixture(params=[1, 0], ids=["f16x3_convs", "fp32_convs"])
def train_conv_mode(request):
    """Both settings of unet_set_train_x3: forward / input-gradient convolutions on the split-operand fp16 kernel
    (default) or on the exact-fp32 kernels."""
    from unet_lane_detection_amd import _lib
    lib = _lib.load(build_if_missing=False)
    prev = lib.unet_set_train_x3(request.param)
    yield request.param
    lib.unet_set_train_x3(prev)


GRADNORM_K = 2.0    # see test_modelA_batch4_step_vs_reference_golden


def test_modelA_batch4_step_vs_reference_golden(golden_dir, train_conv_mode):
    from unet_lane_detection_amd.trainer import UNetTrainer
    g = np.load(os.path.join(golden_dir, "modelA_train_step_b4.npz"))
    tr = UNetTrainer(S.seeded_state_dict(seed=0), device=0, lr=1e-4)
    tr.profile(True)
    tr.forward_backward(torch.from_numpy(S.synthetic_frames(4, seed=3)),
                        torch.from_numpy(S.synthetic_targets(4, seed=3)))
    assert abs(float(tr.loss.item()) - float(g["loss"])) < 1e-5
    names = [r[0] for r in tr.profile_records()]
    tr.profile(False)
    # 17 of the 18 forward convolutions and all 17 input-gradient convolutions have channel counts that are multiples of 64
    assert names.count("conv3x3_f16x3") == (17 if train_conv_mode else 0), names
    assert names.count("dgrad3x3_f16x3") == (17 if train_conv_mode else 0), names
    # ... the 17 weight gradients and the four transposed convolutions (forward, weight and input gradient) with them
    assert names.count("wgrad3x3_f16x3") == (17 if train_conv_mode else 0), names
    # (the forward transposed convolution runs on csrc/upconv_x3_r512.h where there is a work item for half the CUs)
    assert names.count("upconv2x2_ws_f16x3") + names.count("upconv2x2_r512_f16x3") == (4 if train_conv_mode else 0), names
    for label in ("wgrad1x1_f16x3", "upconv_dgrad_f16x3"):
        assert names.count(label) == (4 if train_conv_mode else 0), (label, names)
    gd = tr.grad_dict()
    # Gradient norms.  The bound is derived, not fitted: tests/golden/modelA_train_step_b4_f64.npz holds the norms of the
    # same step with the reference's network run in float64 (make_golden_f64.py).  The BatchNorm-weight gradients are
    # sums of many cancelling terms that depend on which near-zero BatchNorm outputs fall on which side of the ReLU, so
    # two correct fp32 implementations differ at the 1e-3 level: the reference's OWN fp32 run is up to D = 1.14e-3 from
    # its float64 run.  An implementation passes if every norm is within GRADNORM_K x D of the float64 value (D taken
    # over all tensors: which tensor a given rounding hits hardest is a matter of chance) - i.e. it is as close to the true
    # gradient as the reference's fp32 arithmetic is, within a factor GRADNORM_K.
    g64 = np.load(os.path.join(golden_dir, "modelA_train_step_b4_f64.npz"))
    D = max(abs(float(g["gradnorm/" + k[11:]]) - float(g64[k])) / max(float(g64[k]), 1e-6)
            for k in g64.files if k.startswith("gradnorm64/"))
    assert 5e-4 < D < 3e-3, D           # the fixture pair itself (1.14e-3)
    worst = worst32 = 0.0
    for k in g.files:
        if k.startswith("gradnorm/"):
            ref64 = float(g64["gradnorm64/" + k[9:]])
            got = float(gd[k[9:]].double().norm().item())
            worst = max(worst, abs(got - ref64) / max(ref64, 1e-6))
            worst32 = max(worst32, abs(got - float(g[k])) / max(float(g[k]), 1e-6))
            assert abs(got - ref64) <= GRADNORM_K * D * max(ref64, 1e-6), (k, got, ref64, D)
    print("worst gradient-norm deviation from the float64 run %.2e (the reference's fp32 run: %.2e), from the reference's "
          "fp32 run %.2e (convs: %s)" % (worst, D, worst32, "f16x3" if train_conv_mode else "fp32"))
    tr.optimizer_step()
    sd = tr.state_dict()
    for k in g.files:
        if k.startswith("postsum/") and not k.endswith("num_batches_tracked"):
            ref = float(g[k])
            t = sd[k[8:]]
            got = float(t.double().sum().item())
            # Adam's first step moves every element by ~lr*sign(g) = 1e-4; an element whose gradient is within
            # rounding noise of zero may flip sign (2e-4 in the sum): allow 0.1 % of a tensor's elements to do so
            flips = max(10.0, 1e-3 * t.numel())
            assert abs(got - ref) <= 1e-5 * max(1.0, abs(ref)) + 2e-4 * flips, (k, got, ref)
    tr.release()


@pytest.mark.parametrize("feats,shape", [(None, (4, 224, 224)),          # model A: every unit but the first on the side stream
                                         ([64, 128], (2, 28, 28))],      # odd bottleneck: fp32 weight gradients there, which
                                                                         # share the split-K slab with the side stream's
                         ids=["modelA_b4", "f64_128_28x28"])
def test_side_stream_weight_gradients_are_bit_identical(feats, shape):
    """unet_set_train_side: the weight-gradient kernels in line (0), on the handle's side stream forked when the unit's
    dZ exists (1, default) or behind its input-gradient convolution (2).  Only the order of independent launches
    differs: gradients after one backward pass and parameters after three optimizer steps must match bit for bit, with
    the dZ planes alternating between two buffers and the split-K slab shared by every weight gradient."""
    from unet_lane_detection_amd import _lib
    from unet_lane_detection_amd.trainer import UNetTrainer
    lib = _lib.load(build_if_missing=False)
    n, hh, ww = shape
    frames = torch.from_numpy(S.synthetic_frames(n, hh, ww, seed=3))
    tgt = torch.from_numpy(S.synthetic_targets(n, hh, ww, seed=3))
    prev = lib.unet_set_train_side(-1)
    assert prev in (0, 1, 2)
    got = {}
    try:
        for mode in (0, 1, 2):
            assert lib.unet_set_train_side(mode) in (0, 1, 2) and lib.unet_set_train_side(-1) == mode
            tr = UNetTrainer(S.seeded_state_dict(seed=0) if feats is None else S.seeded_state_dict(feats, seed=0), device=0,
                             lr=1e-4)
            tr.forward_backward(frames, tgt)
            torch.cuda.synchronize()
            assert tr.device_error() == 0
            grads = tr.grads.clone()
            for _ in range(3):
                tr.step(frames, tgt)
            torch.cuda.synchronize()
            got[mode] = (grads, tr.params.clone(), float(tr.loss.item()))
            tr.release()
    finally:
        lib.unet_set_train_side(prev)
    assert float(got[0][0].abs().max()) > 0
    for mode in (1, 2):
        assert torch.equal(got[mode][0], got[0][0]), mode
        assert torch.equal(got[mode][1], got[0][1]), mode
        assert got[mode][2] == got[0][2], mode


def test_training_reduces_loss():
    """Ten Adam steps on one fixed batch lower the BCE loss (end-to-end sanity of the sign conventions)."""
    from unet_lane_detection_amd.trainer import UNetTrainer
    feats = [16, 32, 64]
    tr = UNetTrainer(S.seeded_state_dict(feats, seed=8), device=0, lr=1e-3)
    frames = torch.from_numpy(S.synthetic_frames(4, 32, 32, seed=4))
    tgt = torch.from_numpy(S.synthetic_targets(4, 32, 32, seed=4))
    losses = [float(tr.step(frames, tgt).item()) for _ in range(10)]
    assert losses[-1] < losses[0] * 0.9, losses
    tr.release()


def test_bce_dice_loss_and_grads_vs_oracle():
    """The training script's BCEDiceLoss(0.5, 0.5, pos_weight=3) (reference README.md:2169-2170)."""
    from unet_lane_detection_amd.trainer import UNetTrainer
    feats = [16, 32]
    sdn = S.seeded_state_dict(feats, seed=9)
    sd_t = O.to_torch_state(sdn)
    n, hh, ww = 2, 32, 32
    frames, _ = _tie_free_frames(sd_t, n, hh, ww)
    tgt = torch.from_numpy(S.synthetic_targets(n, hh, ww, seed=5))
    fn = lambda lg, t: O.bce_dice_loss(lg, t, 0.5, 0.5, pos_weight=3.0)[0]
    loss, grads, _, logits = O.loss_and_grads(sd_t, O.normalize_u8_nhwc(frames), tgt, loss_fn=fn)
    total, bce, dice = O.bce_dice_loss(logits, tgt, 0.5, 0.5, pos_weight=3.0)
    tr = UNetTrainer(sdn, device=0)
    tr.set_loss("bce_dice", 0.5, 0.5, 3.0)
    tr.forward_backward(torch.from_numpy(frames), tgt)
    lt = tr.loss_terms.cpu().numpy()
    assert abs(lt[0] - total.item()) < 2e-5 and abs(lt[1] - bce.item()) < 2e-5 and abs(lt[2] - dice.item()) < 2e-5
    _check_grads(tr, {k: v.numpy() for k, v in grads.items()}, float(loss), 1e-3)
    tr.release()


def test_checkpoint_resume_is_bitwise(tmp_path):
    """Save after 2 steps in the reference's checkpoint format, resume in a fresh trainer, take 2 more steps:
    identical to 4 uninterrupted steps (deterministic reductions, no float atomics)."""
    from unet_lane_detection_amd.trainer import UNetTrainer
    feats = [16, 32]
    sdn = S.seeded_state_dict(feats, seed=3)
    frames = torch.from_numpy(S.synthetic_frames(2, 32, 32, seed=1))
    tgt = torch.from_numpy(S.synthetic_targets(2, 32, 32, seed=1))
    a = UNetTrainer(sdn, device=0, lr=1e-3)
    for _ in range(2):
        a.step(frames, tgt)
    path = os.path.join(tmp_path, "ck.pth")
    a.save_checkpoint(path, epoch=1, best_dice=0.25)
    for _ in range(2):
        a.step(frames, tgt)
    b = UNetTrainer(sdn, device=0, lr=5e-2)            # different lr: must be overwritten by the checkpoint
    rest = b.load_checkpoint(path)
    assert rest["epoch"] == 1 and rest["best_dice"] == 0.25 and b.step_count == 2 and b.lr == 1e-3
    for _ in range(2):
        b.step(frames, tgt)
    assert torch.equal(a.params, b.params) and torch.equal(a.bn, b.bn)
    assert torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
    a.release()
    b.release()


@pytest.mark.parametrize("tag", ["ref", "amp"])
def test_adamw_bcedice_two_steps_vs_reference_golden(golden_dir, tag):
    """The loop as the reference trains (README.md:2060-2084): BCEDiceLoss(0.5, 0.5, pos_weight=3) (:2169-2170) +
    optim.AdamW(lr, weight_decay) (:2173-2174), two steps on the tiny config, against tensors produced by those
    reference classes (tests/golden/make_golden.py --only adamw).  'ref' = (1e-4, 1e-4), whose decay factor
    1 - 1e-8 rounds to 1 in fp32; 'amp' = (1e-2, 1e-1) makes the decoupled decay worth ~3e-4 per element, 10x the
    tolerance, so Adam-with-L2 (decoupled=False) cannot pass."""
    from unet_lane_detection_amd.trainer import UNetTrainer
    g = np.load(os.path.join(golden_dir, "tiny_f4_8_adamw2.npz"))
    lr, wd = float(g[f"{tag}/lr"]), float(g[f"{tag}/wd"])
    x, t = torch.from_numpy(g["input"]), torch.from_numpy(g["target"])

    def run(decoupled):
        tr = UNetTrainer(S.seeded_state_dict([4, 8], seed=1), device=0, lr=lr, weight_decay=wd, decoupled=decoupled)
        tr.set_loss("bce_dice", 0.5, 0.5, 3.0)
        losses = []
        for _ in range(2):
            tr.step(x, t)
            losses.append(tr.loss_terms[:3].cpu().numpy().astype(np.float64))
        sd = tr.state_dict()
        tr.release()
        return sd, losses

    sd, losses = run(True)
    for step in range(2):
        assert np.abs(losses[step] - g[f"{tag}/loss{step}"]).max() < 2e-5, (step, losses[step])
    tol = 6e-6 if tag == "ref" else 3e-5
    worst = 0.0
    for k in g.files:
        if k.startswith(f"{tag}/post/") and not k.endswith("num_batches_tracked"):
            name = k[len(tag) + 6:]
            d = np.abs(sd[name].numpy().astype(np.float64) - g[k]).max()
            worst = max(worst, d)
            assert d < tol, (name, d)
    print(f"AdamW {tag}: worst post-step difference {worst:.2e} (tolerance {tol:.0e})")
    if tag == "amp":   # the L2 form of weight decay lands far outside the band
        sd_l2, _ = run(False)
        k = "encoder_blocks.0.3.weight"
        assert np.abs(sd_l2[k].numpy().astype(np.float64) - g[f"amp/post/{k}"]).max() > 10 * tol


def test_dice_metric_vs_reference_golden_and_oracle(golden_dir):
    """Device Dice metric (reference README.md:2103-2104, :2115-2120) against the value the reference's own
    compute_dice gave on the fixture logits, and against the oracle on a batch-sized random case."""
    from unet_lane_detection_amd.trainer import UNetTrainer
    tr = UNetTrainer(S.seeded_state_dict([4, 8], seed=1), device=0)
    g = np.load(os.path.join(golden_dir, "bcedice.npz"))
    d = tr.dice_metric(torch.from_numpy(g["x"]), torch.from_numpy(g["t"]))
    assert abs(float(d.item()) - float(g["dice_metric"])) < 1e-6
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(8, 1, 224, 224, generator=gen) * 3
    x[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1e-3, -1e-3])          # the threshold itself is not a positive
    t = (torch.rand(8, 1, 224, 224, generator=gen) < 0.085).float()
    ref = O.compute_dice(torch.sigmoid(x) > 0.5, t)
    d = tr.dice_metric(x, t)
    assert abs(float(d.item()) - float(ref.item())) < 1e-6
    sums = tr._metric.cpu().numpy()                                  # exact integer counts
    p = (x > 0).float()
    assert sums[1] == float((p * t).sum()) and sums[2] == float(p.sum()) and sums[3] == float(t.sum())
    for thr in (0.3, 0.7):
        ref = O.compute_dice(torch.sigmoid(x.double()) > thr, t)
        assert abs(float(tr.dice_metric(x, t, threshold=thr).item()) - float(ref.item())) < 1e-5
    # empty prediction and empty target: smooth / smooth = 1
    z = torch.zeros(1, 1, 16, 16)
    assert abs(float(tr.dice_metric(z - 1, z).item()) - 1.0) < 1e-6
    tr.release()


def test_checkpoint_resume_keeps_loss_and_adamw(tmp_path):
    """Resume of a run configured the reference's way (BCEDiceLoss + AdamW): the loss configuration and the
    decoupled flag must survive load_checkpoint - bitwise equal to the uninterrupted run."""
    from unet_lane_detection_amd.trainer import UNetTrainer
    feats = [16, 32]
    sdn = S.seeded_state_dict(feats, seed=3)
    frames = torch.from_numpy(S.synthetic_frames(2, 32, 32, seed=1))
    tgt = torch.from_numpy(S.synthetic_targets(2, 32, 32, seed=1))

    def make(decoupled):
        tr = UNetTrainer(sdn, device=0, lr=1e-2, weight_decay=0.1, decoupled=decoupled)
        tr.set_loss("bce_dice", 0.5, 0.5, 3.0)
        return tr

    a = make(True)
    for _ in range(2):
        a.step(frames, tgt)
    path = os.path.join(tmp_path, "ck.pth")
    a.save_checkpoint(path, epoch=3)
    for _ in range(2):
        a.step(frames, tgt)
    b = make(False)                      # wrong flag on purpose: the checkpoint's param_group says AdamW
    b.load_checkpoint(path)
    assert b.decoupled is True and b.weight_decay == 0.1
    for _ in range(2):
        b.step(frames, tgt)
    assert torch.equal(a.loss_terms, b.loss_terms) and float(b.loss_terms[2].item()) > 0     # dice term alive
    assert torch.equal(a.params, b.params) and torch.equal(a.bn, b.bn)
    assert torch.equal(a.exp_avg, b.exp_avg) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)
    # and the file loads into torch.optim.AdamW as AdamW
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert ck["optimizer_state_dict"]["param_groups"][0].get("decoupled_weight_decay") is True
    a.release()
    b.release()
