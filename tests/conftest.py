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
    for overlap in (0, 1):
        out = os.path.join(tempfile.mkdtemp(prefix="dp_rehearsal_"), "result.json")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dp_rehearsal.py"), "--ranks", "2", "--steps", "2",
                            "--overlap", str(overlap), "--out", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, timeout=600)
        DP_REHEARSAL["rc"] = DP_REHEARSAL["rc"] or p.returncode
        DP_REHEARSAL["log"] += p.stdout[-2000:]
        if os.path.exists(out):
            with open(out) as f:
                DP_REHEARSAL["result" if overlap == 0 else "result_overlap"] = json.loads(f.read())


@pytest.fixture(scope="session")
def dp_rehearsal():
    return DP_REHEARSAL


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
