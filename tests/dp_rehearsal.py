#!/usr/bin/env python3
"""Data-parallel correctness of UNetTrainer itself, world_size 2 (BASELINE.json configs[3] in miniature).

Run from a FRESH process (`python tests/dp_rehearsal.py`, or tests/conftest.py at session start before any test has
touched the GPU): it starts two rank processes (gloo, both on GPU 0 - the test boxes have one GPU; with RCCL the
only difference is the transport of the same SUM all-reduce) and writes one JSON result line.

Checked after every one of `--steps` steps:
  * all ranks hold bit-identical parameters and Adam moments;
  * they equal, bit for bit, a single-process reference that runs the two shards' forward/backward one after the
    other on one trainer (each from rank 0's BatchNorm buffers, as the DDP-style buffer broadcast gives every
    rank), adds the two gradient buffers and takes ONE optimizer step with grad_scale = 1/2;
  * rank 0's BatchNorm running statistics equal the reference's after its shard-0 pass (rank 0's buffers win).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FEATS = [16, 32, 64]          # --feats 64,128: widths that put the step on the f16x3 training kernels (planes mode)
PER_RANK = 3
SIZE = 64


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(args):
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(args.ranks), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()),
                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    procs = []
    for r in range(args.ranks):
        e = dict(env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    deadline = time.time() + args.timeout
    rc = 0
    while any(p.poll() is None for p in procs):
        if time.time() > deadline or any(p.poll() not in (None, 0) for p in procs):
            rc = 1
            break
        time.sleep(0.2)
    for p in procs:
        if p.poll() is None:
            p.kill()
        rc = rc or (p.returncode or 0)
    return rc


def worker(args):
    import torch
    import torch.distributed as dist

    from unet_lane_detection_amd import dp, state as S
    from unet_lane_detection_amd.trainer import UNetTrainer

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(args.backend)
    ndev = torch.cuda.device_count()
    device = rank % ndev
    torch.cuda.set_device(device)
    feats = [int(x) for x in args.feats.split(",")] if args.feats else FEATS
    sd = S.seeded_state_dict(feats, seed=5)
    total = PER_RANK * world
    frames = torch.from_numpy(S.synthetic_frames(total, SIZE, SIZE, seed=11))
    targets = torch.from_numpy(S.synthetic_targets(total, SIZE, SIZE, seed=11))
    lo, hi = dp.shard_range(total, rank, world)
    tr = UNetTrainer(sd, device=device, lr=1e-3, overlap_allreduce=bool(args.overlap))
    ref = UNetTrainer(sd, device=device, lr=1e-3, process_group=dp.LOCAL) if rank == 0 else None
    result = {"world": world, "backend": dist.get_backend(), "steps": args.steps, "features": feats,
              "ranks_identical": True,
              "params_equal_reference": True, "moments_equal_reference": True, "bn_equal_reference": True,
              "max_param_diff_vs_reference": 0.0, "loss": []}
    for step in range(args.steps):
        loss = tr.step(frames[lo:hi], targets[lo:hi])
        torch.cuda.synchronize()
        flat = torch.cat([tr.params, tr.exp_avg, tr.exp_avg_sq]).cpu()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        if rank == 0:
            result["loss"].append(float(loss.item()))
            for g in gathered[1:]:
                result["ranks_identical"] &= bool(torch.equal(g, gathered[0]))
            # the single-process reference of this step
            bn0 = ref.bn.clone()
            acc = None
            bn_after0 = None
            for r in range(world):
                a, b = dp.shard_range(total, r, world)
                ref.bn.copy_(bn0)
                ref.forward_backward(frames[a:b], targets[a:b])
                if r == 0:
                    acc = ref.grads.clone()
                    bn_after0 = ref.bn.clone()
                else:
                    acc += ref.grads
            ref.grads.copy_(acc)
            ref.bn.copy_(bn_after0)
            ref.optimizer_step(1.0 / world)
            torch.cuda.synchronize()
            result["params_equal_reference"] &= bool(torch.equal(ref.params, tr.params))
            result["moments_equal_reference"] &= bool(torch.equal(ref.exp_avg, tr.exp_avg) and
                                                      torch.equal(ref.exp_avg_sq, tr.exp_avg_sq))
            result["bn_equal_reference"] &= bool(torch.equal(ref.bn, tr.bn))
            result["max_param_diff_vs_reference"] = max(result["max_param_diff_vs_reference"],
                                                        float((ref.params - tr.params).abs().max().item()))
    if args.inject:
        # A failure on ONE rank (word 0: a kernel-side failure record, seen by the next entry point, which then launches
        # nothing; word 1: a range record, seen only by the status kernel behind the backward pass): EVERY rank must
        # raise, none may apply the optimizer step, and nobody may be left waiting in a collective.
        result["inject"] = []
        for word in (0, 1):
            before = torch.cat([tr.params, tr.exp_avg, tr.exp_avg_sq]).clone()
            stepc = tr.step_count
            if rank == world - 1:
                assert tr._lib.unet_debug_set_error_block(tr._h, word, 1) == 0
            t0 = time.time()
            raised, msg = False, ""
            try:
                tr.step(frames[lo:hi], targets[lo:hi])
            except Exception as e:   # noqa: BLE001
                raised, msg = True, str(e)
            torch.cuda.synchronize()
            dt = time.time() - t0
            after = torch.cat([tr.params, tr.exp_avg, tr.exp_avg_sq])
            mine = torch.tensor([1.0 if raised else 0.0, 1.0 if torch.equal(before, after) and tr.step_count == stepc else 0.0,
                                 dt], dtype=torch.float64)
            allr = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            if rank == 0:
                result["inject"].append({"word": word, "all_raised": all(bool(a[0]) for a in allr),
                                         "none_updated": all(bool(a[1]) for a in allr),
                                         "seconds": max(float(a[2]) for a in allr), "rank0_message": msg})
        # and the job goes on: a clean step afterwards, ranks still identical
        tr.step(frames[lo:hi], targets[lo:hi])
        torch.cuda.synchronize()
        flat = torch.cat([tr.params, tr.exp_avg, tr.exp_avg_sq]).cpu()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        if rank == 0:
            result["inject_then_identical"] = all(bool(torch.equal(g, gathered[0])) for g in gathered[1:])
    ok = True
    if rank == 0:
        ok = all(result[k] for k in ("ranks_identical", "params_equal_reference", "moments_equal_reference",
                                     "bn_equal_reference"))
        if args.inject:
            ok = ok and result["inject_then_identical"] and all(
                r["all_raised"] and r["none_updated"] and r["seconds"] < 30.0 for r in result["inject"])
        result["ok"] = ok
        line = json.dumps(result)
        print(line, flush=True)
        if args.out:
            with open(args.out, "w") as f:
                f.write(line + "\n")
        ref.release()
    tr.release()
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--overlap", type=int, default=0, help="1: UNetTrainer(overlap_allreduce=True), two buckets, the tail "
                                                           "bucket on a communication stream")
    ap.add_argument("--inject", type=int, default=0, help="1: after the steps, fail one rank's step through the error "
                                                          "block's test hook: all ranks must raise, none may update")
    ap.add_argument("--feats", default="", help="comma-separated feature widths (default 16,32,64)")
    ap.add_argument("--out", default="")
    ap.add_argument("--timeout", type=float, default=420.0)
    args = ap.parse_args()
    if "RANK" not in os.environ:
        sys.exit(launch(args))
    sys.exit(worker(args))


if __name__ == "__main__":
    main()
