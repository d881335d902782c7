#!/usr/bin/env python3
"""Headline benchmark: frames/s of the U-Net forward on synthetic 224x224 RGB frames.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no torch.distributed environment, bench.py launches itself: N fresh child processes (one per
GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set) are started before this process has touched the GPU, and rank 0
prints the JSON line.  Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the
ranks already exist and are used as they are.  Any other combination (WORLD_SIZE != --gpus) exits non-zero.

A step is one forward pass (uint8 frames already resident in HBM -> logits) over one batch of `--batch` frames
per GPU (BASELINE.json configs[1]: fp32 inference, batch 256, 1 x MI355X).  Frames are independent, so N GPUs
run N independent batches with no data-path collective ("weak" scaling); the only collectives are the timing
barrier and the max-over-ranks.

One JSON line is printed by rank 0.  `roofline` is for the dominant kernel of the headline tier: `achieved` =
MFMA flops that kernel EXECUTES per launch / its average launch duration, measured with HIP events on the launch
stream inside the timed region (`frac` <= 1); the algorithmic (direct-convolution) rate is reported beside it.
`cpu_baseline` times the CPU oracle (oracle/unet_oracle.py, a port of the reference's float model) on the host
cores for a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from unet_lane_detection_amd import state as S  # noqa: E402

GFLOP_PER_FRAME_224 = 73.756          # 2*MAC over convs + upconvs, model A @224x224
GFLOP_CONV3X3_224 = 70.465            # of which 3x3 convolutions (35.23 GMAC); the 17 with Cin % 16 == 0: 70.29
GFLOP_FIRST_CONV_224 = 0.1734         # enc1.conv1 (Cin = 3), never Winograd
PEAK_FP32_MATRIX_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_* dense peak (= fp32 vector peak)
PEAK_BF16_MATRIX_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16/f16 MFMA peak
WINO_EXECUTED = 16.0 / 36.0           # F(2x2,3x3) / F(3x3,2x2): 16 products per tile instead of 36 MACs


def executed_fraction(name):
    """MFMA flops a launch executes per algorithmic (direct-convolution) flop: Winograd kernels execute 16/36 of
    them (in fp32), the split-operand kernels three fp16 MFMAs per product."""
    if "f16x3" in name:
        return 3.0
    return WINO_EXECUTED if "wino" in name else 1.0


def rocprof_name(label):
    """Substring of the rocprofv3 kernel name behind a profiler label of libunet_hip.so."""
    if label.startswith("conv3x3_r512_f16x3_t"):
        flat = label.endswith("_flat")
        t, w, e = label[len("conv3x3_r512_f16x3_t"):].replace("_flat", "").replace("_w", " ").replace("_e", " ").split()
        return f"conv3x3_x3_r512_kernel<{t}, {w}, {e}, {'true' if flat else 'false'}>"
    if label.startswith("conv3x3_t448_f16x3_t"):
        flat = label.endswith("_flat")
        t, c, e = label[len("conv3x3_t448_f16x3_t"):].replace("_flat", "").replace("_c", " ").replace("_e", " ").split()
        return f"conv3x3_x3_t448_kernel<{t}, {c}, {e}, {'true' if flat else 'false'}>"
    if label.startswith("conv3x3_ws_f16x3_tw"):
        flat = label.endswith("_flat")
        tw, e = label[len("conv3x3_ws_f16x3_tw"):].replace("_flat", "").split("_e")
        return f"conv3x3_x3_ws_kernel<{tw}, {e}, {'true' if flat else 'false'}>"
    return {"conv3x3_wino_f32": "wino_f32_kernel", "conv3x3_igemm_f32": "igemm_f32_kernel"}.get(label, label)


def mfma_peak(name):
    """Dense MFMA peak (TFLOP/s) of the pipe mode a kernel runs in."""
    return PEAK_BF16_MATRIX_TFLOPS if ("f16x3" in name or "bf16" in name) else PEAK_FP32_MATRIX_TFLOPS


def cpu_count():
    """Host cores this process may really use: the affinity mask, cut down to the cgroup CPU quota when there is one
    (a 1-GPU box shows all 256 host threads in its affinity mask but owns a 16-core share: 256 torch threads on that
    share ran the CPU model 60x slower than 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f1, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                q, per = int(f1.read()), int(f2.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        n = min(n, max(1, int(quota + 0.5)))
    elif n > 64:
        n = 16      # no quota visible on a shared host: the documented per-GPU share
    return n


def cpu_baseline(seconds_budget=24.0):
    """the reference's benchmark protocol (src/unet.py:152-188) on the CPU port of its float
    model, batch 1 and batch 8, on every host core this process may use; bounded to ~`seconds_budget` seconds."""
    from oracle import unet_oracle as O       # checker / baseline only, never on the product path
    torch.set_num_threads(max(1, cpu_count()))
    sd = O.to_torch_state(S.seeded_state_dict(seed=0))
    x8 = O.normalize_u8_nhwc(S.synthetic_frames(8, seed=0))
    res = {}
    with torch.no_grad():
        O.forward(sd, x8[:1])                  # warm-up (thread pool, primitive cache)
        for batch, share, cap in ((1, 0.35, 100), (8, 0.65, 20)):
            x = x8[:batch]
            times = []
            t_end = time.perf_counter() + seconds_budget * share
            while time.perf_counter() < t_end and len(times) < cap:
                t0 = time.perf_counter()
                O.forward(sd, x)
                times.append(time.perf_counter() - t0)
            ts = np.asarray(times)
            res[batch] = {"frames_per_s": batch / float(ts.mean()), "passes": len(times),
                          "mean_s": float(ts.mean()), "std_s": float(ts.std()), "min_s": float(ts.min()),
                          "max_s": float(ts.max())}
    best = max(res, key=lambda b: res[b]["frames_per_s"])
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": res[best]["frames_per_s"], "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu": cpu_model,
            "sample": f"torch-CPU restatement of the reference model (oracle/unet_oracle.py), 224x224 fp32, same "
                      f"synthetic frames: {res[1]['passes']} passes of batch 1 and {res[8]['passes']} passes of batch 8; "
                      f"value = batch {best}",
            "batch1": res[1], "batch8": res[8]}


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """--gpus N without a torch.distributed environment: start N fresh rank processes.  This process has not
    initialised the GPU (device_count() does not) and never will; it only waits for its children."""
    n = args.gpus
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and ndev < n:
        raise SystemExit(f"bench.py --gpus {n}: only {ndev} HIP device(s) visible (RCCL needs one GPU per rank)")
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()),
                "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    procs = []
    for r in range(n):
        e = dict(env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "LOCAL_WORLD_SIZE": str(n)})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:          # one rank failed: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--tier", default="f16x3", choices=["f16x3", "fp32"],
                    help="arithmetic tier of the headline forward, both held to the fp32 parity bar: f16x3 = fp16 hi+lo "
                         "split operands, three MFMAs per product (default); fp32 = exact fp32 MFMA (Winograd)")
    ap.add_argument("--other-tier-steps", type=int, default=3,
                    help="also time this many steps of the other fp32-parity tier; 0 skips")
    ap.add_argument("--q8-steps", type=int, default=3,
                    help="also time this many steps of the f16q8 tier (cross terms on the fp8 matrix pipe); 0 skips")
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
    ap.add_argument("--int8-steps", type=int, default=3,
                    help="also time this many forward passes of the int8 tier of model B (rank 0); 0 skips")
    ap.add_argument("--int8-batch", type=int, default=256)
    ap.add_argument("--large-steps", type=int, default=3,
                    help="also time this many forward passes of the large-input configuration (BASELINE.json configs[4]: "
                         "640x640, --large-batch frames per GPU = its per-GPU share of batch 512 on 8 GPUs); 0 skips")
    ap.add_argument("--large-batch", type=int, default=64)
    ap.add_argument("--large-size", type=int, default=640)
    ap.add_argument("--overlap-allreduce", action="store_true",
                    help="training leg: exchange the gradients as two buckets, the decoder's under the encoder's backward "
                         "pass (UNetTrainer(overlap_allreduce=True)); default: one all-reduce after the backward pass")
    ap.add_argument("--layers", action="store_true", help="print the per-launch table to stderr")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the output check (it launches one single-frame forward, which would dilute the per-kernel "
                         "averages of a rocprofv3 run of this command)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only "
                                                      "for rehearsing the multi-rank path on a 1-GPU box)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args))
        world = 1
    else:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    dist = None
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a HIP device")
    local_rank = local_rank % ndev       # a rehearsal with more ranks than GPUs shares devices (gloo only)
    torch.cuda.set_device(local_rank)
    backend_name = "none"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == world
        backend_name = dist.get_backend()
    coll = {"nccl": "RCCL"}.get(backend_name, backend_name)

    from unet_lane_detection_amd.model import UNetHIP
    dev = torch.device("cuda", local_rank)
    # gloo moves CPU tensors; RCCL moves device tensors
    cdev = dev if backend_name in ("nccl", "none") else torch.device("cpu")
    model = UNetHIP(S.seeded_state_dict(seed=0), device=local_rank)      # random-init weights of model A
    frames_host = S.synthetic_frames(args.batch, args.size, args.size, seed=rank)
    frames = torch.from_numpy(frames_host).to(dev)
    model.reserve(args.batch, args.size, args.size)
    precision = args.tier

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def max_over_ranks(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        model.run_u8(frames, precision=precision)
    sync_all()
    model.profile(True)                   # a hipEvent pair around every launch, on the launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits = model.run_u8(frames, precision=precision)
    sync_all()
    dt = max_over_ranks(time.perf_counter() - t0)
    recs = model.profile_records()
    model.profile(False)
    dev_status = model.device_error()
    if dev_status != 0:
        raise SystemExit(f"kernel-side failure during the timed region (status {dev_status})")
    # the same K steps without the per-launch events
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits = model.run_u8(frames, precision=precision)
    sync_all()
    dt_plain = max_over_ranks(time.perf_counter() - t0)

    # ---- what the timed batch computed: checksums of its first and last frame, the batch-position check, and (when
    #      the batch starts with the frames of tests/golden/modelA_synth2.npz) the distance to the reference's logits ----
    check = None
    if rank == 0 and not args.no_check:
        lg = logits[:, 0]
        n = lg.shape[0]
        check = {"frame0": {"sum": float(lg[0].double().sum().item()), "abs_sum": float(lg[0].double().abs().sum().item())},
                 f"frame{n - 1}": {"sum": float(lg[n - 1].double().sum().item()),
                                   "abs_sum": float(lg[n - 1].double().abs().sum().item())}}
        alone = model.run_u8(frames[n - 1:n].contiguous(), precision=precision)[:, 0]
        check["last_frame_alone_max_abs_diff"] = float((alone[0] - lg[n - 1]).abs().max().item())
        gpath = os.path.join(ROOT, "tests", "golden", "modelA_synth2.npz")
        if args.size == 224 and n >= 2 and os.path.exists(gpath) and np.array_equal(frames_host[:2],
                                                                                  S.synthetic_frames(2, seed=0)):
            ref = torch.from_numpy(np.load(gpath)["logits"]).to(dev)
            check["frames01_max_abs_err_vs_reference_golden"] = float((lg[:2] - ref).abs().max().item())
            check["frames01_mask_mismatches_off_ties"] = int((((lg[:2] > 0) != (ref > 0)) & (ref.abs() > 2e-4)).sum().item())
        del alone

    # ---- the other fp32-parity tier on the same batch ----
    other = None
    if args.other_tier_steps > 0:
        oprec = "fp32" if precision == "f16x3" else "f16x3"
        ol = model.run_u8(frames, precision=oprec)
        sync_all()
        model.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.other_tier_steps):
            ol = model.run_u8(frames, precision=oprec)
        sync_all()
        odt = max_over_ranks(time.perf_counter() - t0)
        orecs = model.profile_records()
        model.profile(False)
        oagg = {}
        for (nm, ms, fl, by) in orecs:
            a = oagg.setdefault(nm, [0.0, 0.0])
            a[0] += ms
            a[1] += fl
        odom = max((k for k in oagg if oagg[k][1] > 0), key=lambda k: oagg[k][0])
        oexe = oagg[odom][1] * executed_fraction(odom) / (oagg[odom][0] * 1e-3) / 1e12
        other = {"tier": oprec, "frames_per_s": args.batch * world * args.other_tier_steps / odt,
                 "ms_per_step": odt / args.other_tier_steps * 1e3, "dominant_kernel": odom,
                 "dominant_executed_tflops": oexe, "dominant_mfma_pipe_frac": oexe / mfma_peak(odom),
                 "max_abs_diff_vs_headline_tier": float((ol - logits).abs().max().item())}
        del ol

    # ---- the f16q8 tier on the same batch (f16x3 with the cross terms of the wide 3x3 convolutions on the fp8 matrix
    #      pipe: its own accuracy tier, north_star's 1e-3 rather than the 2e-4 the headline is tested to) ----
    q8 = None
    if args.q8_steps > 0 and precision == "f16x3" and args.size == 224:
        ql = model.run_u8(frames, precision="f16q8")     # first call: rebuilds the operators with the fp8 fragments
        sync_all()
        model.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.q8_steps):
            ql = model.run_u8(frames, precision="f16q8")
        sync_all()
        qdt = max_over_ranks(time.perf_counter() - t0)
        qrecs = model.profile_records()
        model.profile(False)
        qagg = {}
        for (nm, ms, fl, by) in qrecs:
            qagg[nm] = qagg.get(nm, 0.0) + ms
        q8 = {"tier": "f16q8 (main term on v_mfma_f32_16x16x32_f16, the two cross terms of every product as fp8 e4m3 on "
                      "v_mfma_scale_f32_16x16x128_f8f6f4, in the 3x3 convolutions with Cout % 256 == 0)",
              "frames_per_s": args.batch * world * args.q8_steps / qdt, "ms_per_step": qdt / args.q8_steps * 1e3,
              "steps": args.q8_steps, "device_status": model.device_error(),
              "max_abs_diff_vs_headline_tier": float((ql - logits).abs().max().item()),
              "mask_pixels_differing_from_headline_tier": int(((ql > 0) != (logits > 0)).sum().item()),
              "mask_pixels": int(ql.numel()),
              "kernel_ms_per_step": {k: v / args.q8_steps for k, v in sorted(qagg.items())
                                     if k.startswith("conv3x3_q8") or k == "planes_to_q8"},
              "accuracy_tier": "logits within 1e-3 of the reference's (tests/test_q8_gpu.py); the headline tier is "
                               "tested to 2e-4"}
        if check is not None and "frames01_max_abs_err_vs_reference_golden" in check:
            ref = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "modelA_synth2.npz"))["logits"]).to(dev)
            q8["frames01_max_abs_err_vs_reference_golden"] = float((ql[:2, 0] - ref).abs().max().item())
            q8["frames01_mask_mismatches"] = int(((ql[:2, 0] > 0) != (ref > 0)).sum().item())
        del ql

    # ---- single-frame latency through the drop-in container, reference protocol (src/unet.py:152-188:
    #      10 warm-up + 100 timed predicts of one 224x224 frame, host numpy in / host numpy out) and the
    #      PCIe-inclusive batch rate (host uint8 frames -> device -> forward -> uint8 mask back on the host) ----
    latency = None
    if rank == 0 and args.latency_iters > 0:
        frame = S.synthetic_frames(1, args.size, args.size, seed=7)
        fdev = torch.empty((1, args.size, args.size, 3), dtype=torch.uint8, device=dev)

        def one_frame():
            fdev.copy_(torch.from_numpy(frame), non_blocking=False)
            _, probs = model.run_u8(fdev, return_probs=True, precision=precision)
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
            _, m = model.run_u8(dbuf, return_mask=True, precision=precision)
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
        pipe = RB.LanePipelineGPU(model, threshold=0.5, precision=precision)
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
                   "median_ms": float(np.median(ts) * 1e3),
                   "min_ms": float(ts.min() * 1e3), "max_ms": float(ts.max() * 1e3), "fps": float(1.0 / ts.mean()),
                   "camera_pipeline_ms": float(tc.mean() * 1e3),
                   "camera_pipeline_note": "sensor_msgs/Image bytes (640x480 bgr8) -> GPU warpPerspective(1055x685) + "
                                           "resize(224) -> network -> mask resized to 1055x685 -> mono8 bytes "
                                           "(reference on RK3588: 2.1 pre + 8.2 NPU + 1.5 post ms)",
                   "pcie_inclusive_batch_fps": pcie_fps,
                   "pcie_note": f"batch {args.batch}: pinned host uint8 frames -> HBM, forward, uint8 masks -> host"}
        del host, dbuf, mhost
    del logits

    # ---- large-input configuration (BASELINE.json configs[4]): 640x640, the per-GPU share (64) of batch 512 ----
    large = None
    if args.large_steps > 0:
        ls, lb = args.large_size, args.large_batch
        lframes = torch.from_numpy(S.synthetic_frames(lb, ls, ls, seed=200 + rank)).to(dev)
        ll = model.run_u8(lframes, precision=precision)      # warm-up: grows the workspace
        sync_all()
        model.profile(True)
        tl0 = time.perf_counter()
        for _ in range(args.large_steps):
            ll = model.run_u8(lframes, precision=precision)
        sync_all()
        ldt = max_over_ranks(time.perf_counter() - tl0)
        lrecs = model.profile_records()
        model.profile(False)
        lstatus = model.device_error()
        lagg = {}
        for (nm, ms, fl, by) in lrecs:
            a = lagg.setdefault(nm, [0.0, 0.0])
            a[0] += ms
            a[1] += fl
        lmf = {k: v for k, v in lagg.items() if v[1] > 0}
        lms = sum(v[0] for v in lmf.values())
        large = {"workload": f"U-Net inference at fp32 parity, {ls}x{ls}x3, batch {lb}/GPU (BASELINE.json configs[4]: "
                             f"batch 512 over 8 GPUs)", "tier": precision,
                 "frames_per_s": lb * world * args.large_steps / ldt, "ms_per_step": ldt / args.large_steps * 1e3,
                 "batch_per_gpu": lb, "steps": args.large_steps, "device_status": lstatus,
                 "algorithmic_gflop_per_frame": GFLOP_PER_FRAME_224 * (ls / 224.0) ** 2,
                 "mfma_kernels_ms_per_step": lms / args.large_steps,
                 "mfma_executed_tflops": sum(v[1] * executed_fraction(k) for k, v in lmf.items()) / max(1e-9, lms * 1e-3) / 1e12,
                 "kernel_ms_per_step": {k: v[0] / args.large_steps for k, v in sorted(lagg.items())}}
        if rank == 0 and not args.no_check:   # the batch-position check of the headline leg, at this size
            one = model.run_u8(lframes[:1].contiguous(), precision=precision)
            last = model.run_u8(lframes[lb - 1:lb].contiguous(), precision=precision)
            large["check"] = {"frame0_alone_max_abs_diff": float((one[0] - ll[0]).abs().max().item()),
                              "last_frame_alone_max_abs_diff": float((last[0] - ll[lb - 1]).abs().max().item()),
                              "frame0": {"sum": float(ll[0].double().sum().item()),
                                         "abs_sum": float(ll[0].double().abs().sum().item())}}
            del one, last
        del lframes, ll

    # ---- int8 tier of the deployed network (model B): calibrate on the fp32 tier, quantise, time ----
    int8 = None
    if rank == 0 and args.int8_steps > 0:
        from unet_lane_detection_amd import quant
        from unet_lane_detection_amd.int8 import UNetInt8, calibrate
        sdb = S.seeded_state_dict([32, 64, 128], seed=0)
        fmb = UNetHIP(sdb, device=local_rank)
        ranges = calibrate(fmb, torch.from_numpy(S.synthetic_frames(16, args.size, args.size, seed=50)), batch=8)
        qnet = UNetInt8(quant.quantize_model(sdb, ranges), device=local_rank)
        iframes = frames[:args.int8_batch]
        lq, mq = qnet.run_u8(iframes, return_mask=True)
        lf, mf = fmb.run_u8(iframes, return_mask=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.int8_steps):
            qnet.run_u8(iframes)
        torch.cuda.synchronize(dev)
        idt = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(args.int8_steps):
            fmb.run_u8(iframes)
        torch.cuda.synchronize(dev)
        fdt = time.perf_counter() - t0
        one = iframes[:1].contiguous()
        ts = [float("nan")]
        for i in range(60 if args.latency_iters > 0 else 0):   # single-frame leg, off together with the latency leg
            t0 = time.perf_counter()
            qnet.run_u8(one, return_probs=True)[1].cpu()
            if i >= 10:
                ts.append(time.perf_counter() - t0)
        inter = ((mq > 0) & (mf > 0)).sum().item()
        union = ((mq > 0) | (mf > 0)).sum().item()
        nb = iframes.shape[0]
        int8 = {"model": "model B: UNet(features=[32,64,128]), BatchNorm folded, sigmoid head (the deployed blob's graph)",
                "frames_per_s": nb * args.int8_steps / idt, "ms_per_step": idt / args.int8_steps * 1e3, "batch": nb,
                "dtype": "int8 activations (per tensor) x int8 weights (per output channel), asymmetric, int32 "
                         "accumulate on v_mfma_i32_16x16x64_i8",
                "executed_tops": 14.117e9 * (args.size / 224.0) ** 2 * nb * args.int8_steps / idt / 1e12,
                "fp32_tier_same_model_frames_per_s": nb * args.int8_steps / fdt,
                "single_frame_ms": float(np.nanmean(ts) * 1e3) if len(ts) > 1 else None,
                "reference_published": "RK3588 NPU INT8 8.2 ms / 122 FPS (README.md:4223), single frame",
                "mask_iou_vs_fp32_tier": inter / max(union, 1),
                "logit_abs_err_vs_fp32_tier_mean": float((lq - lf).abs().mean().item()),
                "parity": "bit-exact against oracle/int8_oracle.py (tests/test_int8_gpu.py); against the Rockchip "
                          "runtime: unpinned"}
        qnet.release()
        fmb.release()
        del lq, mq, lf, mf

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
        bdt = max_over_ranks(time.perf_counter() - tb0)
        brecs = model.profile_records()
        model.profile(False)
        bconv = [(ms, fl) for (nm, ms, fl, by) in brecs if nm.startswith("conv3x3") and nm.endswith("bf16")]
        bup = [(ms, fl) for (nm, ms, fl, by) in brecs if nm.startswith("upconv")]
        conv_tf = sum(f for _, f in bconv) / max(1e-9, sum(m for m, _ in bconv) * 1e-3) / 1e12
        bf16 = {"frames_per_s": args.bf16_batch * world * args.bf16_steps / bdt,
                "ms_per_step": bdt / args.bf16_steps * 1e3, "batch_per_gpu": args.bf16_batch,
                "dtype": "bf16 storage, fp32 accumulate (first conv: uint8 normalisation fused, operands split "
                         "into bf16 hi + lo, 3 MFMAs)",
                "conv3x3_executed_tflops": conv_tf, "conv3x3_mfma_pipe_frac": conv_tf / PEAK_BF16_MATRIX_TFLOPS,
                "conv3x3_ms_per_step": sum(m for m, _ in bconv) / args.bf16_steps,
                "upconv_executed_tflops": sum(f for _, f in bup) / max(1e-9, sum(m for m, _ in bup) * 1e-3) / 1e12,
                "kernel_ms_per_step": sum(r[1] for r in brecs) / args.bf16_steps,
                "peak_tflops_dense_bf16": PEAK_BF16_MATRIX_TFLOPS,
                "ceiling_frames_per_s": PEAK_BF16_MATRIX_TFLOPS * 1e3 / (GFLOP_PER_FRAME_224 * (args.size / 224.0) ** 2),
                "accuracy_tier": "separate from fp32 parity: see tests/test_bf16_gpu.py (logit error ~1e-1 max, "
                                 "mask IoU ~0.99 against the fp32 reference)"}
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
        torch.cuda.empty_cache()
        from unet_lane_detection_amd.trainer import UNetTrainer
        tr = UNetTrainer(S.seeded_state_dict(seed=0), device=local_rank, lr=1e-4,
                         overlap_allreduce=args.overlap_allreduce)
        tb = args.train_batch
        tframes = torch.from_numpy(S.synthetic_frames(tb, args.size, args.size, seed=100 + rank)).to(dev)
        ttargets = torch.from_numpy(S.synthetic_targets(tb, args.size, args.size, seed=100 + rank)).to(dev)
        tr.step(tframes, ttargets)      # warm-up (allocates the workspace)
        tr.step(tframes, ttargets)
        sync_all()
        # timed without the per-launch events: with them on, the step keeps its weight gradients in line (the event
        # pairs of overlapping kernels would mean nothing), so the table below comes from a second set of steps
        t1 = time.perf_counter()
        for _ in range(args.train_steps):
            tr.step(tframes, ttargets)
        sync_all()
        tdt = max_over_ranks(time.perf_counter() - t1)
        final_loss = float(tr.loss.item())
        tr.profile(True)
        t1 = time.perf_counter()
        for _ in range(args.train_steps):
            tr.step(tframes, ttargets)
        sync_all()
        tdt_prof = max_over_ranks(time.perf_counter() - t1)
        trecs = tr.profile_records()
        tr.profile(False)
        # the gradient exchange alone, event-timed on the streams it runs on (after the timed steps: same buffers)
        ar_ms = None
        if world > 1:
            sync_all()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ar = []
            for _ in range(3):
                e0.record()
                tr.allreduce_grads()
                e1.record()
                torch.cuda.synchronize(dev)
                ar.append(e0.elapsed_time(e1))
            ar_ms = max_over_ranks(float(np.median(ar)))
        agg = {}
        for (nm, ms, fl, by) in trecs:
            a = agg.setdefault(nm, [0.0, 0.0, 0])
            a[0] += ms
            a[1] += fl
            a[2] += 1
        mf = {k: v for k, v in agg.items() if v[1] > 0}
        mf_ms = sum(v[0] for v in mf.values())
        alg = sum(v[1] for v in mf.values())
        exe = sum(v[1] * executed_fraction(k) for k, v in mf.items())
        train = {"frames_per_s": tb * world * args.train_steps / tdt, "ms_per_step": tdt / args.train_steps * 1e3,
                 "batch_per_gpu": tb, "steps": args.train_steps, "loss_after": final_loss,
                 "weight_gradients": ("in line", "on the handle's side stream, forked when the unit's dZ exists "
                                      "(unet_set_train_side 1; bit-identical to in line)",
                                      "on the handle's side stream, forked behind the unit's input-gradient "
                                      "convolution (unet_set_train_side 2; bit-identical to in line)"
                                      )[tr._lib.unet_set_train_side(-1)],
                 "ms_per_step_in_line_with_launch_events": tdt_prof / args.train_steps * 1e3,
                 "optimizer": "Adam(lr=1e-4)", "loss": "BCEWithLogits(mean)",
                 "grad_allreduce": "none (1 GPU)" if world == 1 else
                                   f"{coll} all-reduce (backend {backend_name}, world {world}) of the flat fp32 gradient "
                                   f"buffer ({tr.params.numel() * 4 / 1e6:.1f} MB) " +
                                   ("as two buckets, the decoder's under the encoder's backward pass"
                                    if args.overlap_allreduce else "after the backward pass"),
                 "grad_allreduce_ms": ar_ms, "overlap_allreduce": bool(args.overlap_allreduce),
                 "kernel_ms_per_step": {k: v[0] / args.train_steps for k, v in sorted(agg.items())},
                 "mfma_kernels_ms_per_step": mf_ms / args.train_steps,
                 "mfma_executed_tflops": exe / max(1e-9, mf_ms * 1e-3) / 1e12,
                 # time-weighted share of its own pipe mode's peak (fp32 MFMA 157.3 TF, fp16 MFMA 2,500 TF) per kernel
                 "mfma_pipe_frac": sum(v[1] * executed_fraction(k) / 1e12 / mfma_peak(k) for k, v in mf.items()) /
                                   max(1e-9, mf_ms * 1e-3),
                 "mfma_algorithmic_tflops": alg / max(1e-9, mf_ms * 1e-3) / 1e12,
                 "per_kernel_pipe_frac": {k: v[1] * executed_fraction(k) / (v[0] * 1e-3) / 1e12 / mfma_peak(k)
                                          for k, v in sorted(mf.items())}}
        if args.layers and rank == 0:
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                print(f"train {k:28s} {v[0] / args.train_steps:9.3f} ms/step  x{v[2] // args.train_steps:3d}  "
                      f"{v[1] / (v[0] * 1e-3) / 1e12 if v[0] else 0:6.1f} TF", file=sys.stderr)
        tr.release()

    if rank == 0:
        total_frames = args.batch * world * args.steps
        fps = total_frames / dt
        by_name = {}
        for (nm, ms, fl, by) in recs:
            a = by_name.setdefault(nm, [0.0, 0.0, 0.0, 0])
            a[0] += ms
            a[1] += fl
            a[2] += by
            a[3] += 1
        dom = max((k for k in by_name if by_name[k][1] > 0), key=lambda k: by_name[k][0])   # most time among MFMA kernels
        d_ms, d_fl, d_by, d_n = by_name[dom]
        all_ms = sum(r[1] for r in recs)
        algorithmic = d_fl / (d_ms * 1e-3) / 1e12 if d_ms > 0 else 0.0
        achieved = algorithmic * executed_fraction(dom)
        peak = mfma_peak(dom)
        if args.layers:
            per = len(recs) // max(args.steps, 1)
            for (nm, ms, fl, by) in recs[-per:]:
                print(f"{nm:24s} {ms:8.3f} ms  {fl / (ms * 1e-3) / 1e12 if ms else 0:7.1f} TF  "
                      f"{by / (ms * 1e-3) / 1e9 if ms else 0:8.1f} GB/s", file=sys.stderr)
            print(f"sum of kernel time {all_ms / args.steps:.3f} ms/step, wall {dt / args.steps * 1e3:.3f} ms/step",
                  file=sys.stderr)
        scale = (args.size / 224.0) ** 2
        # whole-network ceiling of this tier: executed flops per frame / the MFMA peak
        if "wino" in dom:
            exec_gflop = (GFLOP_PER_FRAME_224 - (GFLOP_CONV3X3_224 - GFLOP_FIRST_CONV_224) * (1 - WINO_EXECUTED)) * scale
        else:
            exec_gflop = GFLOP_PER_FRAME_224 * executed_fraction(dom) * scale
        ceiling = peak * 1e3 / exec_gflop
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside the process, so this is
        # the figure of a committed rocprofv3 run of this same command (tools/gpu_profile.sh; FETCH_SIZE doubled as the
        # micro-architecture guide prescribes) - static, named by `traffic_source`; null when no such file applies.
        traffic, traffic_src, pmc_busy = None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.batch == 256 and args.size == 224:
            with open(tpath) as f:
                tj = json.load(f)
            kmatch = rocprof_name(dom)
            if kmatch in tj.get("kernel", ""):
                traffic = tj.get("hbm_bytes_per_launch")
                pmc_busy = tj.get("mfma_busy_frac")
                traffic_src = "profiles/traffic.json (committed rocprofv3 --pmc run of this command, not this run)"
        kernel_names = {
            "conv3x3_wino_f32": "wino_f32_kernel (conv3x3+BN+ReLU[+pool], Winograd F(2x2,3x3) on v_mfma_f32_16x16x4_f32)",
            "conv3x3_igemm_f32": "igemm_f32_kernel<TAPS=9> (conv3x3+BN+ReLU, v_mfma_f32_16x16x4_f32)"}
        if dom.startswith("conv3x3_ws_f16x3"):
            kernel_names[dom] = (rocprof_name(dom) + " (conv3x3+BN+ReLU, epilogue " +
                                 {"e0": "store", "e1": "store + 2x2 max-pool", "e2": "fused 1x1 head",
                                  "e3": "fp32 store"}[dom.replace("_flat", "")[-2:]] +
                                 ("; batch tiled as one tall image" if dom.endswith("_flat") else "") +
                                 "; fp16 hi+lo split operands, 3 x v_mfma_f32_16x16x32_f16 per product, fp32 accumulate)")
        if dom.startswith("conv3x3_r512_f16x3"):
            kernel_names[dom] = (rocprof_name(dom) + " (conv3x3+BN+ReLU; one wave per SIMD with 512 registers, weights "
                                 "straight from L2, 224-pixel tiles" +
                                 ("; batch tiled as one tall image" if dom.endswith("_flat") else "") +
                                 "; fp16 hi+lo split operands, 3 x v_mfma_f32_16x16x32_f16 per product, fp32 accumulate)")
        if dom.startswith("conv3x3_t448_f16x3"):
            kernel_names[dom] = (rocprof_name(dom) + " (conv3x3+BN+ReLU, epilogue " +
                                 {"e0": "store", "e1": "store + 2x2 max-pool", "e2": "fused 1x1 head",
                                  "e3": "fp32 store"}[dom.replace("_flat", "")[-2:]] +
                                 "; one wave per SIMD with 512 registers, weights straight from L2, 16 x 28 / 8 x 28 pixel "
                                 "tiles of 4 x 4-pixel fragments with immediate-only LDS addressing" +
                                 ("; batch tiled as one tall image" if dom.endswith("_flat") else "") +
                                 "; fp16 hi+lo split operands, 3 x v_mfma_f32_16x16x32_f16 per product, fp32 accumulate)")
        dtype = {"fp32": "f32", "f16x3": "f16x3 (every fp32 operand as fp16 hi + lo, three fp16 MFMAs per product, fp32 "
                                         "accumulate; held to the fp32 parity bar)"}[args.tier]
        out = {
            "metric": f"frames/sec at {args.size}x{args.size} bs={args.batch} (U-Net inference at fp32 parity, {args.tier} tier)",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic uint8 frames (numpy default_rng, seed = rank), seeded random-init weights of the "
                    "reference UNet(features=[64,128,256,512]); frames resident in HBM before the timed region",
            "config": {"workload": f"U-Net inference at fp32 parity, batch {args.batch}/GPU, {args.size}x{args.size}x3 "
                                   f"(BASELINE.json configs[1])",
                       "global_batch": args.batch * world, "parallelism": f"dp{world} (independent batches)",
                       "tier": args.tier},
            "distributed": {"world_size": world, "backend": backend_name,
                            "launcher": os.environ.get("TORCHELASTIC_RUN_ID") and "torch.distributed.run" or
                                        ("bench.py self-launch" if world > 1 else "single process")},
            "ms_per_step_without_events": dt_plain / args.steps * 1e3,
            # `achieved` = MFMA flops the dominant kernel EXECUTES (Winograd F(2x2,3x3): 16/36 of the direct-convolution
            # count; split operands: 3x it, on the fp16 pipe) / its launch time; `algorithmic_tflops` = the direct-convolution count / the same time.
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         # share of SHADER cycles the MFMA pipe is busy (SQ_VALU_MFMA_BUSY_CYCLES, same committed run);
                         # frac / this = the clock the chip holds under the kernel relative to the 2.4 GHz of `peak`
                         "pmc_mfma_busy_frac": pmc_busy,
                         "sustained_clock_ghz": (2.4 * achieved / peak / pmc_busy) if pmc_busy else None,
                         "algorithmic_tflops": algorithmic,
                         "algorithmic_bytes_per_launch": d_by / max(d_n, 1),
                         "kernel": kernel_names.get(dom, dom),
                         "launches": d_n, "kernel_ms_per_step": d_ms / args.steps,
                         "avg_launch_ms": d_ms / max(d_n, 1),
                         "whole_net_algorithmic_tflops": fps / world * GFLOP_PER_FRAME_224 * scale / 1e3,
                         "whole_net_executed_tflops": fps / world * exec_gflop / 1e3,
                         "ceiling_frames_per_s": ceiling, "frac_of_ceiling": fps / world / ceiling},
            "check": check,
        }
        if other is not None:
            out["other_parity_tier"] = other
        if q8 is not None:
            out["cross_fp8_tier"] = q8
        if latency is not None:
            out["latency"] = latency
        # counter figures of the side legs' dominant kernels: like `roofline.traffic`, from the committed rocprofv3 PMC run
        lpath = os.path.join(ROOT, "profiles", "traffic_legs.json")
        if os.path.exists(lpath) and args.size == 224:
            with open(lpath) as f:
                lj = json.load(f)
            for leg, blk, ok in (("train", train, args.train_batch == 64), ("bf16", bf16, args.bf16_batch == 1024)):
                if blk is not None and ok and leg in lj:
                    blk["pmc_dominant_kernel"] = dict(lj[leg], note="profiles/traffic_legs.json: a committed rocprofv3 "
                                                                    "--pmc run of this command, not this run")
        if large is not None:
            out["large_input"] = large
        if int8 is not None:
            out["int8_model_b"] = int8
        if bf16 is not None:
            out["bf16"] = bf16
        if train is not None:
            out["train"] = train
        if not args.no_cpu_baseline and world == 1:      # a reported baseline, timed once: at N = 1 only
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if model is not None:
        model.release()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
