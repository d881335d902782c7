#!/usr/bin/env python3
"""Headline benchmark: frames/s of the U-Net forward on synthetic 224x224 RGB frames.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step is one forward pass (uint8 frames already resident in HBM -> logits) over one batch of
`--batch` frames per GPU (BASELINE.json configs[1]: fp32 inference, batch 256, 1 x MI355X).
Frames are independent, so N GPUs run N independent batches with no data-path collective
("weak" scaling); the only collectives are the timing barrier and the max-over-ranks.

One JSON line is printed by rank 0.  `roofline` is for the dominant kernel family (the fp32 MFMA
implicit-GEMM conv): algorithmic FLOPs of its launches / their summed durations, measured with
HIP events on the launch stream inside the timed region.  `cpu_baseline` times the CPU oracle
(oracle/unet_oracle.py, a port of the reference's float model) on the host cores for a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from unet_lane_detection_amd import state as S  # noqa: E402

GFLOP_PER_FRAME_224 = 73.756          # 2*MAC over convs + upconvs, model A @224x224
PEAK_FP32_MATRIX_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_* dense peak (= fp32 vector peak)


def cpu_baseline(batch, seconds_budget=20.0):
    from oracle import unet_oracle as O       # checker / baseline only, never on the product path
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count()
    # a 1-GPU box owns a 16-core share of its host; more threads than that only thrash
    torch.set_num_threads(max(1, min(avail, 16)))
    sd = O.to_torch_state(S.seeded_state_dict(seed=0))
    frames = S.synthetic_frames(batch, seed=0)
    x = O.normalize_u8_nhwc(frames)
    with torch.no_grad():
        O.forward(sd, x[:1])                  # warm-up (thread pool, primitive cache)
        times = []
        t_end = time.perf_counter() + seconds_budget
        while time.perf_counter() < t_end and len(times) < 20:
            t0 = time.perf_counter()
            O.forward(sd, x)
            times.append(time.perf_counter() - t0)
    mean = float(np.mean(times))
    return {"value": batch / mean, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} forward passes of batch {batch} (224x224 fp32, same synthetic frames), "
                      f"torch-CPU restatement of the reference model, mean {mean:.3f} s/pass"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=3,
                    help="also time this many training steps (BCE + Adam, batch --train-batch per GPU, gradients "
                         "all-reduced over RCCL when N > 1); 0 skips the training leg")
    ap.add_argument("--train-batch", type=int, default=64)
    ap.add_argument("--latency-iters", type=int, default=100,
                    help="single-frame latency leg through the container protocol (rank 0 only); 0 skips")
    ap.add_argument("--bf16-steps", type=int, default=2,
                    help="also time this many bf16-tier forward passes (BASELINE.json configs[2]); 0 skips")
    ap.add_argument("--bf16-batch", type=int, default=1024)
    ap.add_argument("--layers", action="store_true", help="print the per-launch table to stderr")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only "
                                                      "for rehearsing the multi-rank path on a 1-GPU box)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a HIP device")
    local_rank = local_rank % ndev       # a rehearsal with more ranks than GPUs shares devices (gloo only)
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from unet_lane_detection_amd.model import UNetHIP
    dev = torch.device("cuda", local_rank)
    model = UNetHIP(S.seeded_state_dict(seed=0), device=local_rank)      # random-init weights of model A
    frames = torch.from_numpy(S.synthetic_frames(args.batch, args.size, args.size, seed=rank)).to(dev)
    model.reserve(args.batch, args.size, args.size)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        model.run_u8(frames)
    sync_all()
    model.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.run_u8(frames)
    sync_all()
    dt = time.perf_counter() - t0
    recs = model.profile_records()
    model.profile(False)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- single-frame latency through the drop-in container, reference protocol (src/unet.py:152-188:
    #      10 warm-up + 100 timed predicts of one 224x224 frame, host numpy in / host numpy out) and the
    #      PCIe-inclusive batch rate (host uint8 frames -> device -> forward -> uint8 mask back on the host) ----
    latency = None
    if rank == 0 and args.latency_iters > 0:
        frame = S.synthetic_frames(1, args.size, args.size, seed=7)
        fdev = torch.empty((1, args.size, args.size, 3), dtype=torch.uint8, device=dev)

        def one_frame():
            fdev.copy_(torch.from_numpy(frame), non_blocking=False)
            _, probs = model.run_u8(fdev, return_probs=True)
            return probs.cpu().numpy()

        for _ in range(10):
            one_frame()
        ts = []
        for _ in range(args.latency_iters):
            t0 = time.perf_counter()
            one_frame()
            ts.append(time.perf_counter() - t0)
        ts = np.asarray(ts)
        host = torch.from_numpy(S.synthetic_frames(args.batch, args.size, args.size, seed=8)).pin_memory()
        dbuf = torch.empty(host.shape, dtype=torch.uint8, device=dev)
        mhost = torch.empty((args.batch, args.size, args.size), dtype=torch.uint8).pin_memory()

        def one_batch():
            dbuf.copy_(host, non_blocking=True)
            _, m = model.run_u8(dbuf, return_mask=True)
            mhost.copy_(m, non_blocking=True)
            torch.cuda.synchronize(dev)

        one_batch()
        t0 = time.perf_counter()
        for _ in range(3):
            one_batch()
        pcie_fps = 3 * args.batch / (time.perf_counter() - t0)
        # camera pipeline of the ROS callback (src/unet_ros_node.py:296-311): 640x480 bgr8 message bytes -> warp to
        # 1055x685 + resize to 224x224 on the GPU -> network -> mask resized back -> mono8 message bytes
        from unet_lane_detection_amd import ros_bridge as RB
        pipe = RB.LanePipelineGPU(model, threshold=0.5)
        cam = np.random.default_rng(9).integers(0, 256, size=(480, 640, 3), dtype=np.uint8)
        msg = RB.ImageMsg(height=480, width=640, encoding="bgr8", data=cam.tobytes())
        for _ in range(5):
            pipe.process(msg)
        tc = []
        for _ in range(max(10, args.latency_iters // 2)):
            t0 = time.perf_counter()
            pipe.process(msg)
            tc.append(time.perf_counter() - t0)
        tc = np.asarray(tc)
        latency = {"protocol": "reference benchmark loop (src/unet.py:152-188): one frame, host numpy -> container "
                               "forward -> host numpy probabilities",
                   "iters": int(args.latency_iters), "mean_ms": float(ts.mean() * 1e3), "std_ms": float(ts.std() * 1e3),
                   "min_ms": float(ts.min() * 1e3), "max_ms": float(ts.max() * 1e3), "fps": float(1.0 / ts.mean()),
                   "camera_pipeline_ms": float(tc.mean() * 1e3),
                   "camera_pipeline_note": "sensor_msgs/Image bytes (640x480 bgr8) -> GPU warpPerspective(1055x685) + "
                                           "resize(224) -> network -> mask resized to 1055x685 -> mono8 bytes "
                                           "(reference on RK3588: 2.1 pre + 8.2 NPU + 1.5 post ms)",
                   "pcie_inclusive_batch_fps": pcie_fps,
                   "pcie_note": f"batch {args.batch}: pinned host uint8 frames -> HBM, forward, uint8 masks -> host"}

    # ---- bf16 tier (BASELINE.json configs[2]): bf16 storage, fp32 accumulate, batch 1024 ----
    bf16 = None
    if args.bf16_steps > 0:
        bframes = torch.from_numpy(S.synthetic_frames(args.bf16_batch, args.size, args.size, seed=1 + rank)).to(dev)
        model.run_u8(bframes, precision="bf16")       # warm-up: packs bf16 weights, allocates the workspace
        sync_all()
        model.profile(True)
        tb0 = time.perf_counter()
        for _ in range(args.bf16_steps):
            model.run_u8(bframes, precision="bf16")
        sync_all()
        bdt = time.perf_counter() - tb0
        brecs = model.profile_records()
        model.profile(False)
        if dist is not None:
            t = torch.tensor([bdt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bdt = float(t.item())
        bconv = [(ms, fl) for (nm, ms, fl, by) in brecs if nm == "conv3x3_igemm_bf16"]
        bf16 = {"frames_per_s": args.bf16_batch * world * args.bf16_steps / bdt,
                "ms_per_step": bdt / args.bf16_steps * 1e3, "batch_per_gpu": args.bf16_batch,
                "dtype": "bf16 storage, fp32 accumulate (first conv fp32)",
                "conv_tflops": sum(f for _, f in bconv) / max(1e-9, sum(m for m, _ in bconv) * 1e-3) / 1e12,
                "peak_tflops_dense_bf16": 2500.0,
                "accuracy_tier": "separate from fp32 parity: see tests/test_bf16_gpu.py (logit error ~1e-2, mask IoU ~0.99)"}
        if args.layers and rank == 0:
            per = len(brecs) // max(args.bf16_steps, 1)
            for (nm, ms, fl, by) in brecs[-per:]:
                print(f"bf16 {nm:24s} {ms:8.3f} ms  {fl / (ms * 1e-3) / 1e12 if ms else 0:7.1f} TF  "
                      f"{by / (ms * 1e-3) / 1e9 if ms else 0:8.1f} GB/s", file=sys.stderr)
        del bframes

    # ---- training leg (BASELINE.json configs[3]): batch 64/GPU, BCE-with-logits + Adam, DP all-reduce ----
    train = None
    if args.train_steps > 0:
        model.release()     # give the activation workspace back before the trainer allocates its own
        model = None
        from unet_lane_detection_amd.trainer import UNetTrainer
        tr = UNetTrainer(S.seeded_state_dict(seed=0), device=local_rank, lr=1e-4)
        tb = args.train_batch
        tframes = torch.from_numpy(S.synthetic_frames(tb, args.size, args.size, seed=100 + rank)).to(dev)
        ttargets = torch.from_numpy(S.synthetic_targets(tb, args.size, args.size, seed=100 + rank)).to(dev)
        tr.step(tframes, ttargets)      # warm-up (allocates the workspace)
        sync_all()
        tr.profile(True)
        t1 = time.perf_counter()
        for _ in range(args.train_steps):
            tr.step(tframes, ttargets)
        sync_all()
        tdt = time.perf_counter() - t1
        trecs = tr.profile_records()
        tr.profile(False)
        final_loss = float(tr.loss.item())
        if dist is not None:
            t = torch.tensor([tdt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tdt = float(t.item())
        agg = {}
        for (nm, ms, fl, by) in trecs:
            a = agg.setdefault(nm, [0.0, 0.0, 0])
            a[0] += ms
            a[1] += fl
            a[2] += 1
        train = {"frames_per_s": tb * world * args.train_steps / tdt, "ms_per_step": tdt / args.train_steps * 1e3,
                 "batch_per_gpu": tb, "steps": args.train_steps, "loss_after": final_loss,
                 "optimizer": "Adam(lr=1e-4)", "loss": "BCEWithLogits(mean)",
                 "grad_allreduce": "none (1 GPU)" if world == 1 else f"RCCL all-reduce, 1 flat fp32 bucket of "
                                                                        f"{tr.params.numel() * 4 / 1e6:.1f} MB",
                 "kernel_ms_per_step": {k: v[0] / args.train_steps for k, v in sorted(agg.items())},
                 "mfma_tflops": sum(v[1] for v in agg.values()) / max(1e-9, sum(v[0] for k, v in agg.items()
                                                                             if v[1] > 0) * 1e-3) / 1e12}
        if args.layers and rank == 0:
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                print(f"train {k:28s} {v[0] / args.train_steps:9.3f} ms/step  x{v[2] // args.train_steps:3d}  "
                      f"{v[1] / (v[0] * 1e-3) / 1e12 if v[0] else 0:6.1f} TF", file=sys.stderr)
        tr.release()

    if rank == 0:
        total_frames = args.batch * world * args.steps
        fps = total_frames / dt
        wino = [(ms, fl) for (nm, ms, fl, by) in recs if nm == "conv3x3_wino_f32"]
        dom = "conv3x3_wino_f32" if wino else "conv3x3_igemm_f32"
        conv = [(ms, fl) for (nm, ms, fl, by) in recs if nm == dom]
        conv_ms = sum(m for m, _ in conv)
        conv_fl = sum(f for _, f in conv)
        all_ms = sum(r[1] for r in recs)
        achieved = conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        if args.layers:
            per = len(recs) // max(args.steps, 1)
            for (nm, ms, fl, by) in recs[-per:]:
                print(f"{nm:24s} {ms:8.3f} ms  {fl / (ms * 1e-3) / 1e12 if ms else 0:7.1f} TF  "
                      f"{by / (ms * 1e-3) / 1e9 if ms else 0:8.1f} GB/s", file=sys.stderr)
            print(f"sum of kernel time {all_ms / args.steps:.3f} ms/step, wall {dt / args.steps * 1e3:.3f} ms/step",
                  file=sys.stderr)
        scale = (args.size / 224.0) ** 2
        # HBM bytes per launch of the dominant kernel come from a committed rocprofv3 PMC run of this same
        # command (tools/gpu_profile.sh): counters cannot be read from inside the process.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.batch == 256 and args.size == 224:
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": f"frames/sec at {args.size}x{args.size} bs={args.batch} (U-Net fp32 inference)",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic uint8 frames (numpy default_rng, seed = rank), seeded random-init weights of the "
                    "reference UNet(features=[64,128,256,512]); frames resident in HBM before the timed region",
            "config": {"workload": f"U-Net fp32 inference, batch {args.batch}/GPU, {args.size}x{args.size}x3 "
                                   f"(BASELINE.json configs[1])",
                       "global_batch": args.batch * world, "parallelism": f"dp{world} (independent batches)"},
            # `achieved` counts the ALGORITHMIC flops of the operator (direct 3x3 convolution, 2*9*Cin*Cout per
            # pixel).  The Winograd F(2x2,3x3) kernel executes 16/36 of them on the MFMA pipe, so
            # `frac` can exceed the share of the pipe that is busy; `mfma_pipe_frac` is that share.
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": sum(by for (nm, ms, fl, by) in recs if nm == dom) / max(len(conv), 1),
                         "kernel": ("wino_f32_kernel (conv3x3+BN+ReLU[+pool], Winograd F(2x2,3x3) on v_mfma_f32_16x16x4_f32)"
                                    if wino else "igemm_f32_kernel<TAPS=9> (conv3x3+BN+ReLU, v_mfma_f32_16x16x4_f32)"),
                         "mfma_pipe_frac": achieved / PEAK_FP32_MATRIX_TFLOPS * (16.0 / 36.0 if wino else 1.0),
                         "launches": len(conv), "kernel_ms_per_step": conv_ms / args.steps,
                         "whole_net_tflops": fps / world * GFLOP_PER_FRAME_224 * scale / 1e3},
        }
        if latency is not None:
            out["latency"] = latency
        if bf16 is not None:
            out["bf16"] = bf16
        if train is not None:
            out["train"] = train
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(8)
        print(json.dumps(out), flush=True)
    if model is not None:
        model.release()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
