"""Data-parallel path on CPU: world_size 2 over gloo.  The GPU kernels cannot run here, so the CPU oracle
stands in for the per-rank forward/backward; what is under test is the product's DP plumbing
(unet_lane_detection_amd/dp.py: batch sharding, the single flat gradient bucket, SUM all-reduce with the
1/world scale folded into the optimizer, rank-0 buffer broadcast, max-over-ranks timing)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import unet_oracle as O
from unet_lane_detection_amd import dp, state as S

FEATS = [4, 8]
GLOBAL_BATCH = 6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _flat(grads, keys):
    return torch.cat([grads[k].reshape(-1) for k in keys])


def _rank_main(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    sd = O.to_torch_state(S.seeded_state_dict(FEATS, seed=1))
    frames = S.synthetic_frames(GLOBAL_BATCH, 16, 16, seed=0)
    targets = torch.from_numpy(S.synthetic_targets(GLOBAL_BATCH, 16, 16, seed=0))
    lo, hi = dp.shard_range(GLOBAL_BATCH, rank, world)
    loss, grads, new_stats, _ = O.loss_and_grads(sd, O.normalize_u8_nhwc(frames[lo:hi]), targets[lo:hi])
    keys = sorted(grads)
    bucket = _flat(grads, keys).clone()
    work, scale = dp.allreduce_flat_sum(bucket)
    assert work is None and abs(scale - 1.0 / world) < 1e-12
    averaged = bucket * scale
    # BatchNorm buffers: rank 0 wins
    buf = torch.full((8,), float(rank + 1))
    dp.broadcast_buffers(buf)
    t = dp.max_over_ranks(float(rank) + 0.5, torch.device("cpu"))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), averaged=averaged.numpy(), own=_flat(grads, keys).numpy(),
             buf=buf.numpy(), tmax=t, lo=lo, hi=hi, loss=float(loss))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average(tmp_path):
    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    # shards partition the batch
    assert (int(r[0]["lo"]), int(r[0]["hi"]), int(r[1]["lo"]), int(r[1]["hi"])) == (0, 3, 3, 6)
    # every rank holds the same averaged bucket = mean of the per-rank gradients (DDP semantics: per-replica
    # BatchNorm statistics, so this is NOT the gradient of one batch-6 step)
    want = 0.5 * (r[0]["own"] + r[1]["own"])
    for i in range(world):
        np.testing.assert_allclose(r[i]["averaged"], want, rtol=0, atol=1e-7)
    assert np.array_equal(r[0]["averaged"], r[1]["averaged"])
    assert np.all(r[1]["buf"] == 1.0) and np.all(r[0]["buf"] == 1.0)
    assert float(r[0]["tmax"]) == 1.5 and float(r[1]["tmax"]) == 1.5


def test_shard_range_covers_everything():
    for total in (1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
