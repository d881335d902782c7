"""Training step on the HIP path: train-mode forward, BCE-with-logits, backward, Adam
(reference README.md:2060-2084, :1694-1709, :2173), with an optional data-parallel gradient
all-reduce between backward and the optimizer.  All arithmetic runs in libunet_hip.so; torch
provides the flat device buffers and torch.distributed (RCCL) the collective."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, checkpoint, dp
from .model import infer_features
from .state import INPUT_MEAN, INPUT_STD


class UNetTrainer:
    STATUS_PAD = 4    # floats in front of the gradients in the all-reduce bucket (keeps them 16-byte aligned)

    def __init__(self, state_dict, device=0, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 decoupled=False, process_group=None, in_channels=3, overlap_allreduce=False,
                 check_device_status=True):
        if not torch.cuda.is_available():
            raise RuntimeError("UNetTrainer needs a HIP device; there is no CPU fallback")
        self._lib = _lib.load()
        self.features = infer_features(state_dict)
        self.device = torch.device("cuda", int(device))
        cfg = _lib.UnetConfig()
        cfg.in_channels, cfg.out_channels, cfg.depth = in_channels, 1, len(self.features)
        for i, f in enumerate(self.features):
            cfg.features[i] = f
        cfg.device = int(device)
        for i in range(3):
            cfg.input_mean[i], cfg.input_std[i] = INPUT_MEAN[i], INPUT_STD[i]
        h = C.c_void_p()
        _lib.check(self._lib.unet_create(C.byref(cfg), C.byref(h)), "unet_create")
        self._h = h
        self.lr, self.betas, self.eps = lr, betas, eps
        self.weight_decay, self.decoupled = weight_decay, decoupled
        self.group = process_group
        self.step_count = 0
        self.overlap = bool(overlap_allreduce)
        self.check_device_status = bool(check_device_status)
        self._comm = None
        self._split = 0
        self._fb_rc = 0
        self._defer_fb_error = False

        # flat buffers in unet_param_name order
        n = self._lib.unet_num_params(h)
        self.layout = []   # (name, is_buffer, offset, numel)
        for i in range(n):
            isb, off = C.c_int(), C.c_size_t()
            _lib.check(self._lib.unet_train_layout(h, i, C.byref(isb), C.byref(off)), "unet_train_layout", h)
            self.layout.append((self._lib.unet_param_name(h, i).decode(), bool(isb.value), int(off.value),
                                int(self._lib.unet_param_numel(h, i))))
        P = int(self._lib.unet_train_param_numel(h))
        B = int(self._lib.unet_train_buffer_numel(h))
        host_p = np.empty(P, dtype=np.float32)
        host_b = np.empty(B, dtype=np.float32)
        self._shapes = {}
        for name, isb, off, numel in self.layout:
            v = state_dict[name]
            v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            assert v.size == numel, (name, v.shape, numel)
            self._shapes[name] = tuple(v.shape)
            (host_b if isb else host_p)[off:off + numel] = v.astype(np.float32).reshape(-1)
        self.params = torch.from_numpy(host_p).to(self.device)
        self.bn = torch.from_numpy(host_b).to(self.device)
        # The all-reduce bucket: STATUS_PAD floats in front of the gradients, the first of them the step's status word
        # (0 = every launch of this rank's forward/backward was clean).  It is summed over the ranks by the same
        # all-reduce as the gradients, so all ranks agree on whether to apply the step (see step()).
        self._bucket = torch.zeros(self.STATUS_PAD + P, dtype=torch.float32, device=self.device)
        self.grads = self._bucket[self.STATUS_PAD:]
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.loss_terms = torch.zeros(4, dtype=torch.float32, device=self.device)   # total, bce, dice, -
        self.loss = self.loss_terms[:1]
        self.num_batches_tracked = 0
        self._loss_cfg = None
        self._metric = torch.zeros(4, dtype=torch.float32, device=self.device)
        rc = self._lib.unet_train_attach(h, self._p(self.params), self._p(self.grads), self._p(self.exp_avg),
                                         self._p(self.exp_avg_sq), self._p(self.bn))
        _lib.check(rc, "unet_train_attach", h)
        if self.overlap and dp.world(self.group)[1] > 1:
            self._enable_overlap()

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    def set_loss(self, kind="bce", bce_weight=0.5, dice_weight=0.5, pos_weight=3.0, smooth=1e-6):
        """'bce': BCEWithLogitsLoss (reference README.md:1694-1709, BASELINE.json config);
        'bce_dice': the training script's BCEDiceLoss(0.5, 0.5, pos_weight=3) (README.md:1855-1893, :2169-2170);
        self.loss_terms then holds (total, bce, dice)."""
        if kind == "bce":
            rc = self._lib.unet_train_set_loss(self._h, 0, 1.0, 0.0, 1.0, smooth)
        elif kind == "bce_dice":
            rc = self._lib.unet_train_set_loss(self._h, 1, bce_weight, dice_weight, pos_weight, smooth)
        else:
            raise ValueError(kind)
        _lib.check(rc, "unet_train_set_loss", self._h)
        self._loss_cfg = (kind, bce_weight, dice_weight, pos_weight, smooth)

    def dice_metric(self, logits, targets, threshold=0.5, smooth=1e-6):
        """Dice score of the validation loop (reference README.md:2103-2104, :2115-2120):
        pred = sigmoid(logits) > threshold; returns a 1-element device tensor (no host sync)."""
        from .model import _logit
        logits = logits.to(self.device, torch.float32).contiguous()
        targets = targets.to(self.device, torch.float32).contiguous()
        if logits.numel() != targets.numel():
            raise ValueError("logits and targets differ in size")
        rc = self._lib.unet_dice_metric(self.device.index, self._p(logits), self._p(targets), logits.numel(),
                                        _logit(threshold), smooth, self._p(self._metric), self._stream())
        _lib.check(rc, "unet_dice_metric")
        return self._metric[:1].clone()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- one step, in the reference's order ------------------------------------------------------
    def forward_backward(self, images, targets, return_logits=False):
        """images: (N,H,W,3) uint8 frames or (N,3,H,W) float32 normalised; targets (N,1,H,W) float 0/1.
        Fills self.grads (zero_grad + backward) and self.loss; returns the logits if asked."""
        dp.broadcast_buffers(self.bn, self.group)
        targets = targets.to(self.device, torch.float32).contiguous()
        images = images.to(self.device).contiguous()
        if images.dtype == torch.uint8:
            n, h, w, _ = images.shape
            fn = self._lib.unet_train_forward_backward_u8
        else:
            images = images.to(torch.float32)
            n, _, h, w = images.shape
            fn = self._lib.unet_train_forward_backward_f32
        logits = torch.empty((n, 1, h, w), dtype=torch.float32, device=self.device) if return_logits else None
        rc = fn(self._h, self._p(images), self._p(targets), n, h, w, self._p(self.loss_terms), self._p(logits),
                self._stream())
        self._fb_rc = int(rc)
        if rc != 0 and self._defer_fb_error:    # step() under data parallelism: the ranks have to fail together
            return logits
        _lib.check(rc, "unet_train_forward_backward", self._h)
        self.num_batches_tracked += 1
        return logits

    # overlap_allreduce (off by default): the exchange is 124 MB against a 49 ms step (1-2 % at xGMI rates), while the
    # persistent convolution kernels assume all 256 CUs - a collective kernel that holds some of them while the
    # encoder's backward runs makes every statically-strided kernel wait for its slowest CU.  Until an 8-GPU run has
    # measured both, the gradients are exchanged after the backward pass; the two-bucket overlap stays selectable.
    def _enable_overlap(self):
        """Two gradient buckets (reference: none; torch DDP's bucketing in miniature): the tail of the flat buffer -
        decoder, bottleneck, head: final two thirds into the backward pass - is all-reduced on a communication stream
        that the library releases at that point, under the encoder's backward; the encoder's bucket follows."""
        if self._comm is None:
            self._comm = torch.cuda.Stream(self.device)
            self._split = int(self._lib.unet_train_grad_split(self._h))
            _lib.check(self._lib.unet_train_set_comm_stream(self._h, C.c_void_p(self._comm.cuda_stream)),
                       "unet_train_set_comm_stream", self._h)

    def allreduce_grads(self):
        _, ws = dp.world(self.group)
        if ws == 1:
            return 1.0
        # the bucket (or its last part: the encoder's, exchanged after the backward pass has ended) starts with the
        # status word
        if not self.overlap:
            work, scale = dp.allreduce_flat_sum(self._bucket, self.group)
            return scale
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self._comm):     # already waiting for the mid-backward event
            dp.allreduce_flat_sum(self.grads[self._split:], self.group)
        dp.allreduce_flat_sum(self._bucket[:self.STATUS_PAD + self._split], self.group)
        main.wait_stream(self._comm)            # the optimizer step needs both buckets
        return 1.0 / ws

    def optimizer_step(self, grad_scale=1.0):
        self.step_count += 1
        rc = self._lib.unet_train_adam_step(self._h, self.step_count, self.lr, self.betas[0], self.betas[1], self.eps,
                                            self.weight_decay, 1 if self.decoupled else 0, grad_scale, self._stream())
        _lib.check(rc, "unet_train_adam_step", self._h)

    def device_error(self, current_stream_only=False):
        """Wait for the device (or, with current_stream_only, for torch's current stream of it: everything the trainer
        launches goes there) and return (and clear) the status of every launch on this handle since the last call: 0,
        UNET_ERR_HIP after a kernel-side failure, UNET_ERR_RANGE when an f16x3 activation left the fp16 range
        (include/unet_hip.h)."""
        if current_stream_only:
            return int(self._lib.unet_device_error_on(self._h, self._stream()))
        return int(self._lib.unet_device_error(self._h))

    def step(self, images, targets):
        """forward + backward, gradient exchange, optimizer step.  With check_device_status (default) the step refuses
        to update the parameters from gradients a failed launch produced - on EVERY rank: a failure is local to one rank
        (its kernels, its frames), but its gradients are summed into every rank's buffer, and a rank that raised alone
        would leave the others waiting in the next collective.  So each rank's status word (unet_device_status_to; or the
        status of a forward/backward call that failed at its entry) travels in front of the gradients through the same
        all-reduce; afterwards every rank reads the sum, and either all of them raise or all of them step.  One wait per
        step, for the trainer's stream only."""
        _, ws = dp.world(self.group)
        if not self.check_device_status:
            self.forward_backward(images, targets)
            self.optimizer_step(self.allreduce_grads())
            return self.loss
        if ws == 1:
            self.forward_backward(images, targets)
            rc = self.device_error(current_stream_only=True)
            if rc != 0:
                raise _lib.UnetError(rc, "unet_device_error: the gradients of this step are invalid, no update was applied")
            self.optimizer_step(1.0)
            return self.loss
        self._defer_fb_error = True
        try:
            self.forward_backward(images, targets)
        finally:
            self._defer_fb_error = False
        if self._fb_rc != 0:     # failed at its entry: nothing (or not everything) was launched
            self._bucket[:1].fill_(1.0)
        else:
            _lib.check(self._lib.unet_device_status_to(self._h, self._p(self._bucket), self._stream()),
                       "unet_device_status_to", self._h)
        scale = self.allreduce_grads()
        failed = float(self._bucket[0].item())               # waits for the trainer's stream: backward + exchange
        local = self._fb_rc or self.device_error(current_stream_only=True)   # this rank's own record, cleared
        if failed != 0.0:
            who = "this rank" if local else "another rank"
            raise _lib.UnetError(local or _lib.UNET_ERR_HIP,
                                 f"data-parallel step: {int(round(failed))} of {ws} ranks reported a failed launch ({who}); "
                                 "the gradients of this step are invalid, no rank applied an update",
                                 self._lib.unet_last_error(self._h).decode() if local else "")
        self.optimizer_step(scale)
        return self.loss

    # ---- views in the reference's state_dict naming -----------------------------------------------
    def _view(self, flat_p, flat_b):
        out = {}
        for name, isb, off, numel in self.layout:
            out[name] = (flat_b if isb else flat_p)[off:off + numel].view(self._shapes[name])
        return out

    def state_dict(self):
        sd = {k: v.detach().cpu().clone() for k, v in self._view(self.params, self.bn).items()}
        for name in list(sd):
            if name.endswith("running_var"):
                sd[name.replace("running_var", "num_batches_tracked")] = torch.tensor(self.num_batches_tracked)
        return sd

    # ---- checkpoints in the reference's format (README.md:2208-2213) --------------------------------
    def _param_entries(self):
        return [(n, off, numel, self._shapes[n]) for (n, isb, off, numel) in self.layout if not isb]

    def save_checkpoint(self, path, epoch=0, best_dice=None, with_optimizer=True):
        osd = None
        if with_optimizer:
            osd = checkpoint.optimizer_state_dict(self._param_entries(), self.exp_avg, self.exp_avg_sq,
                                                  self.step_count, self.lr, self.betas, self.eps,
                                                  self.weight_decay, self.decoupled)
        checkpoint.save(path, self.state_dict(), epoch=epoch, optimizer_state=osd, best_dice=best_dice)

    def load_checkpoint(self, path):
        """Restore parameters, BatchNorm buffers and (if present) the Adam moments and step; returns the rest
        of the checkpoint (epoch, best_dice)."""
        msd, rest = checkpoint.load(path)
        for name, isb, off, numel in self.layout:
            (self.bn if isb else self.params)[off:off + numel].copy_(msd[name].reshape(-1).to(torch.float32))
        nbt = [v for k, v in msd.items() if k.endswith("num_batches_tracked")]
        if nbt:
            self.num_batches_tracked = int(nbt[0])
        if "optimizer_state_dict" in rest:
            step, group = checkpoint.load_optimizer_state(rest.pop("optimizer_state_dict"), self._param_entries(),
                                                          self.exp_avg, self.exp_avg_sq)
            self.step_count = step
            self.lr, self.betas, self.eps = group["lr"], tuple(group["betas"]), group["eps"]
            self.weight_decay = group["weight_decay"]
            # torch.optim.AdamW is Adam with decoupled_weight_decay=True (torch 2.10 writes that key for both classes);
            # older AdamW files lack the key, so a caller-selected AdamW is kept when the file does not say
            if "decoupled_weight_decay" in group:
                self.decoupled = bool(group["decoupled_weight_decay"])
        # packed MFMA operands follow the parameters; the TrainState (loss configuration, workspace) stays
        _lib.check(self._lib.unet_train_repack(self._h, self._stream()), "unet_train_repack", self._h)
        return rest

    def grad_dict(self):
        return {k: v for k, v in self._view(self.grads, self.bn).items() if "running_" not in k}

    def profile(self, on=True):
        _lib.check(self._lib.unet_profile_enable(self._h, 1 if on else 0), "unet_profile_enable", self._h)

    def profile_records(self):
        out = []
        name = C.create_string_buffer(64)
        ms, fl, by = C.c_double(), C.c_double(), C.c_double()
        for i in range(self._lib.unet_profile_count(self._h)):
            self._lib.unet_profile_get(self._h, i, name, 64, C.byref(ms), C.byref(fl), C.byref(by))
            out.append((name.value.decode(), ms.value, fl.value, by.value))
        return out

    def release(self):
        if getattr(self, "_h", None) is not None:
            torch.cuda.synchronize(self.device)
            self._lib.unet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass
