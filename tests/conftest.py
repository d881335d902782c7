import json
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# result of tests/dp_rehearsal.py, run once at session start (see pytest_sessionstart)
DP_REHEARSAL = {"ran": False, "rc": None, "result": None, "log": ""}
# bench.py --gpus 2 (two fresh ranks, gloo, sharing GPU 0), also run at session start
BENCH_2RANK = {"ran": False, "rc": None, "line": None, "log": ""}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_tests_selected(config):
    expr = (config.getoption("markexpr", "") or "").strip()
    return "gpu" in expr and "not gpu" not in expr


def pytest_sessionstart(session):
    """The 2-rank data-parallel rehearsal of UNetTrainer must be started from a process that has not initialised the
    GPU (its ranks are fresh child processes): do it here, before any test runs.  torch.cuda.device_count() does not
    initialise the device.  tests/test_dp_trainer_gpu.py asserts on the stored result."""
    if not _gpu_tests_selected(session.config):
        return
    import torch
    if torch.cuda.device_count() < 1:
        return
    # twice: gradients exchanged after the backward pass (the default), and as two buckets with the tail bucket on a
    # communication stream under the encoder's backward (overlap_allreduce=True)
    DP_REHEARSAL.update(ran=True, rc=0, log="")
    # ... and a third time with widths that are multiples of 64, which puts the step on the f16x3 training kernels
    # (operand planes, dynamic gradient scales, weight-gradient slabs)
    # ... and once more with a failure injected on one rank after the steps (all ranks must raise, none may update), in
    # both exchange modes
    for key, overlap, feats, inject in (("result", 0, "", 0), ("result_overlap", 1, "", 0), ("result_x3", 0, "64,128", 0),
                                        ("result_inject", 0, "", 1), ("result_inject_overlap", 1, "", 1)):
        out = os.path.join(tempfile.mkdtemp(prefix="dp_rehearsal_"), "result.json")
        cmd = [sys.executable, os.path.join(ROOT, "tests", "dp_rehearsal.py"), "--ranks", "2", "--steps", "2" if not inject else "1",
               "--overlap", str(overlap), "--out", out, "--inject", str(inject)]
        if feats:
            cmd += ["--feats", feats]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        DP_REHEARSAL["rc"] = DP_REHEARSAL["rc"] or p.returncode
        DP_REHEARSAL["log"] += p.stdout[-2000:]
        if os.path.exists(out):
            with open(out) as f:
                DP_REHEARSAL[key] = json.loads(f.read())
    # bench.py's own multi-rank path: --gpus 2 starts two fresh rank processes itself (side legs off, one step)
    BENCH_2RANK["ran"] = True
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                        "--warmup", "1", "--batch", "8", "--train-batch", "4", "--train-steps", "1", "--other-tier-steps", "0", "--q8-steps", "0",
                        "--latency-iters", "0", "--bf16-steps", "0", "--int8-steps", "0", "--large-steps", "0",
                        "--no-cpu-baseline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    BENCH_2RANK["rc"] = p.returncode
    BENCH_2RANK["log"] = (p.stdout[-1500:] + p.stderr[-1500:])
    for line in p.stdout.splitlines():
        if line.startswith("{"):
            BENCH_2RANK["line"] = json.loads(line)


@pytest.fixture(scope="session")
def dp_rehearsal():
    return DP_REHEARSAL


@pytest.fixture(scope="session")
def bench_2rank():
    return BENCH_2RANK


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
