"""UNetTrainer under data parallelism, world_size 2 (BASELINE.json configs[3]; the reference has no distributed code,
SURVEY.md section 2, so the contract is north_star's: batch-sharded DP, gradients averaged by one all-reduce).

The ranks are started by tests/conftest.py at session start from a process that has not touched the GPU
(tests/dp_rehearsal.py: two fresh processes, gloo, sharing GPU 0); this module asserts on what they reported."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["result", "result_overlap", "result_x3"],
                         ids=["exchange_after_backward", "two_buckets_tail_overlapped", "f16x3_training_kernels"])
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


@pytest.mark.parametrize("mode", ["result_inject", "result_inject_overlap"],
                         ids=["exchange_after_backward", "two_buckets_tail_overlapped"])
def test_one_rank_fails_all_ranks_raise_nobody_updates(dp_rehearsal, mode):
    """A launch failure is local to one rank, its gradients are not: UNetTrainer.step sends every rank's status word
    through the gradient all-reduce, so ALL ranks raise, NONE applies Adam (parameters, moments and step counter
    unchanged), nobody waits for a collective the failed rank never joins (seconds, not the process group's timeout), and
    the next clean step leaves the ranks bit-identical.  Injected through unet_debug_set_error_block on the last rank
    only: word 0 (the entry point refuses to launch) and word 1 (seen by the status kernel behind the backward pass)."""
    r = dp_rehearsal
    assert r["ran"], "the DP rehearsal did not run (conftest.pytest_sessionstart)"
    assert r.get(mode) is not None, r["log"]
    res = r[mode]
    assert res["ranks_identical"] and res["params_equal_reference"], res
    assert [i["word"] for i in res["inject"]] == [0, 1]
    for i in res["inject"]:
        assert i["all_raised"], i
        assert i["none_updated"], i
        assert i["seconds"] < 30.0, i
        assert "ranks reported a failed launch" in i["rank0_message"], i
    assert res["inject_then_identical"], res
    assert res["ok"], res


def test_bench_two_ranks_line(bench_2rank):
    """`bench.py --gpus 2` launches two ranks itself (the driver's multi-GPU path with the ranks it is given is the
    same code after the launch): the line must say so - world size, backend, the gradient exchange of the training
    leg - and count both ranks' frames."""
    b = bench_2rank
    assert b["ran"], "bench.py --gpus 2 did not run (conftest.pytest_sessionstart)"
    assert b["rc"] == 0 and b["line"] is not None, b["log"]
    line = b["line"]
    assert line["n_gpus"] == 2
    assert line["distributed"]["world_size"] == 2 and line["distributed"]["backend"] == "gloo"
    assert line["config"]["global_batch"] == 16
    assert "gloo" in line["train"]["grad_allreduce"] and "world 2" in line["train"]["grad_allreduce"]
    assert line["train"]["grad_allreduce_ms"] is not None and line["train"]["grad_allreduce_ms"] > 0
    assert line["scaling"] == "weak" and line["value"] > 0
