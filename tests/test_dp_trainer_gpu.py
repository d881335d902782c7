"""UNetTrainer under data parallelism, world_size 2 (BASELINE.json configs[3]; the reference has no distributed code,
SURVEY.md section 2, so the contract is north_star's: batch-sharded DP, gradients averaged by one all-reduce).

The ranks are started by tests/conftest.py at session start from a process that has not touched the GPU
(tests/dp_rehearsal.py: two fresh processes, gloo, sharing GPU 0); this module asserts on what they reported."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["result", "result_overlap"],
                         ids=["exchange_after_backward", "two_buckets_tail_overlapped"])
def test_two_rank_trainer_step_equals_sequential_reference(dp_rehearsal, mode):
    r = dp_rehearsal
    assert r["ran"], "the DP rehearsal did not run (conftest.pytest_sessionstart)"
    assert r["rc"] == 0 and r.get(mode) is not None, r["log"]
    res = r[mode]
    assert res["world"] == 2 and res["steps"] == 2
    assert res["ranks_identical"], res                 # every rank holds the same parameters and moments, bit for bit
    assert res["params_equal_reference"], res          # = one process doing both shards, summing, stepping once
    assert res["moments_equal_reference"], res
    assert res["bn_equal_reference"], res              # rank 0's BatchNorm buffers win
    assert res["loss"][1] < res["loss"][0]
