"""Host-side logic of the training loop pieces that need no GPU: LR schedule, checkpoint format, loss
restatement against the torch modules the reference composes."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O
from unet_lane_detection_amd import checkpoint, schedules, state as S


def test_cosine_warm_restarts_matches_torch():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-4, weight_decay=1e-4)          # reference README.md:2173
    ref = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=10, T_mult=2)   # README.md:2177
    mine = schedules.CosineAnnealingWarmRestarts(1e-4, T_0=10, T_mult=2)
    assert abs(mine.get_lr() - opt.param_groups[0]["lr"]) < 1e-15
    for _ in range(75):                                                # crosses restarts at 10, 30, 70
        opt.step()
        ref.step()
        assert abs(mine.step() - opt.param_groups[0]["lr"]) < 1e-12


def test_bce_dice_matches_torch_composition():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 1, 16, 16, generator=g) * 3
    t = (torch.rand(2, 1, 16, 16, generator=g) < 0.2).float()
    total, bce, dice = O.bce_dice_loss(x, t, 0.5, 0.5, pos_weight=3.0)
    ref_bce = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([3.0]))(x, t)
    s = torch.sigmoid(x).view(-1)
    ref_dice = 1 - (2 * (s * t.view(-1)).sum() + 1e-6) / (s.sum() + t.sum() + 1e-6)
    assert abs(bce.item() - ref_bce.item()) < 1e-6 and abs(dice.item() - ref_dice.item()) < 1e-6
    assert abs(total.item() - (0.5 * ref_bce + 0.5 * ref_dice).item()) < 1e-6


def test_optimizer_state_loads_into_torch_adam(tmp_path):
    """A checkpoint written in the reference's layout loads into torch.optim.Adam over parameters of the
    reference's shapes, and comes back unchanged through our loader."""
    feats = [4, 8]
    sd = S.seeded_state_dict(feats, seed=1)
    entries, off = [], 0
    for key, shape, kind in S.state_dict_spec(feats):
        if kind in ("bn_mean", "bn_var", "bn_count"):
            continue
        n = int(np.prod(shape))
        entries.append((key, off, n, tuple(shape)))
        off += n
    g = torch.Generator().manual_seed(2)
    m, v = torch.randn(off, generator=g), torch.rand(off, generator=g)
    osd = checkpoint.optimizer_state_dict(entries, m, v, step=7, lr=1e-4, betas=(0.9, 0.999), eps=1e-8,
                                          weight_decay=0.0, decoupled=False)
    path = os.path.join(tmp_path, "best_model.pth")
    checkpoint.save(path, {k: torch.from_numpy(np.asarray(a)) for k, a in sd.items()}, epoch=3,
                    optimizer_state=osd, best_dice=0.5)
    msd, rest = checkpoint.load(path)
    assert set(msd) == set(sd) and rest["epoch"] == 3 and rest["best_dice"] == 0.5
    params = [torch.nn.Parameter(torch.zeros(e[3])) for e in entries]
    opt = torch.optim.Adam(params, lr=1e-3)
    opt.load_state_dict(rest["optimizer_state_dict"])                 # torch accepts the layout
    assert opt.param_groups[0]["lr"] == 1e-4
    assert torch.equal(opt.state[params[3]]["exp_avg"].reshape(-1), m[entries[3][1]:entries[3][1] + entries[3][2]])
    m2, v2 = torch.zeros(off), torch.zeros(off)
    step, group = checkpoint.load_optimizer_state(opt.state_dict(), entries, m2, v2)
    assert step == 7 and torch.equal(m2, m) and torch.equal(v2, v)
    # bare state_dict form (reference README.md:2231)
    bare = os.path.join(tmp_path, "last_model.pth")
    checkpoint.save(bare, msd)
    msd2, rest2 = checkpoint.load(bare)
    assert rest2 == {} and set(msd2) == set(sd)


def _run_bench(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=300)


def test_bench_gpus_mismatch_is_an_error_not_a_silent_one_rank_run():
    """`bench.py --gpus N` either runs N ranks or fails: with a torch.distributed environment of another size it
    exits non-zero, and without one it launches the ranks itself - which, on a box with fewer GPUs than ranks
    (this container has none), is refused before anything runs instead of measuring one rank and calling it N."""
    p = _run_bench(["--gpus", "8", "--steps", "1"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and not p.stdout.strip()
    if torch.cuda.device_count() < 2:
        p = _run_bench(["--gpus", "2", "--steps", "1"])
        assert p.returncode != 0 and "HIP device(s) visible" in p.stderr and not p.stdout.strip()


def test_build_staleness_is_by_content(tmp_path, monkeypatch):
    """build.py rebuilds when the sources' content hash differs from the one stored beside the .so (mtimes do not
    matter), and accepts a matching library as is."""
    from unet_lane_detection_amd import build
    digest = build.source_digest()
    assert digest == build.source_digest() and len(digest) == 64
    lib, hf = tmp_path / "libunet_hip.so", tmp_path / "libunet_hip.so.srchash"
    monkeypatch.setattr(build, "LIB", str(lib))
    monkeypatch.setattr(build, "HASHFILE", str(hf))
    assert build.is_stale()                       # no library
    lib.write_bytes(b"x")
    assert build.is_stale()                       # library without a recorded digest
    record = digest + "\n" + build.hipcc_version() + "\n"
    hf.write_text(record)
    assert not build.is_stale()
    os.utime(lib, (1, 1))                         # an old mtime does not make it stale
    assert not build.is_stale()
    hf.write_text("0" * 64 + "\n" + build.hipcc_version() + "\n")
    assert build.is_stale()                       # other sources
    if build.hipcc_path(required=False):
        hf.write_text(digest + "\nHIP version: some other compiler\n")
        assert build.is_stale()                   # same sources, built by another compiler
    monkeypatch.setattr(build, "hipcc_path", lambda required=True: None)
    hf.write_text(digest + "\nHIP version: some other compiler\n")
    assert not build.is_stale()                   # ... which a box without hipcc cannot check: the shipped library is used
    monkeypatch.undo()
    monkeypatch.setattr(build, "LIB", str(lib))
    monkeypatch.setattr(build, "HASHFILE", str(hf))
    monkeypatch.setenv("UNET_HIPCC_FLAGS", "-DUNET_WS_STAMPS=1")
    hf.write_text(record)
    assert build.is_stale()                       # same sources, other flags


def test_bench_profiler_labels_map_to_kernel_instances():
    """bench.py names the dominant kernel by its rocprofv3 name so that profiles/traffic.json and the committed
    kernel-stats summary can be matched to the live hipEvent averages: one label per template instance."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.rocprof_name("conv3x3_ws_f16x3_tw32_e0") == "conv3x3_x3_ws_kernel<32, 0, false>"
    assert bench.rocprof_name("conv3x3_ws_f16x3_tw16_e1_flat") == "conv3x3_x3_ws_kernel<16, 1, true>"
    assert bench.rocprof_name("conv3x3_r512_f16x3_t28_w1_e0_flat") == "conv3x3_x3_r512_kernel<28, 1, 0, true>"
    assert bench.rocprof_name("conv3x3_r512_f16x3_t14_w2_e3") == "conv3x3_x3_r512_kernel<14, 2, 3, false>"
    assert bench.executed_fraction("conv3x3_r512_f16x3_t28_w1_e0") == 3.0
    assert bench.rocprof_name("conv3x3_wino_f32") == "wino_f32_kernel"
    assert bench.executed_fraction("conv3x3_ws_f16x3_tw32_e0_flat") == 3.0
    assert abs(bench.executed_fraction("conv3x3_wino_f32") - 16.0 / 36.0) < 1e-12
    assert bench.mfma_peak("conv3x3_ws_f16x3_tw32_e0") == 2500.0 and bench.mfma_peak("conv3x3_wino_f32") == 157.3
    # the committed PMC file must name the instance the default run's dominant label maps to, or `traffic` goes null
    import json
    with open(os.path.join(os.path.dirname(__file__), "..", "profiles", "traffic.json")) as f:
        tj = json.load(f)
    assert ("conv3x3_x3_ws_kernel<" in tj["kernel"] and tj["kernel"].count(",") == 2) or \
           ("conv3x3_x3_r512_kernel<" in tj["kernel"] and tj["kernel"].count(",") == 3) or \
           ("conv3x3_x3_t448_kernel<" in tj["kernel"] and tj["kernel"].count(",") == 3), tj["kernel"]
    assert bench.rocprof_name("conv3x3_t448_f16x3_t28_c2_e0") == "conv3x3_x3_t448_kernel<28, 2, 0, false>"
    assert bench.rocprof_name("conv3x3_t448_f16x3_t28_c4_e1_flat") == "conv3x3_x3_t448_kernel<28, 4, 1, true>"
    assert bench.rocprof_name("conv3x3_t448_f16x3_t28_c2_e0") in tj["kernel"]   # round 4's dominant instance
    assert bench.executed_fraction("conv3x3_t448_f16x3_t32_c1_e2") == 3.0
