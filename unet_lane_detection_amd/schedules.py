"""Learning-rate schedule of the reference training script:
`CosineAnnealingWarmRestarts(optimizer, T_0=10, T_mult=2)`, stepped once per epoch
(reference README.md:2177, :2198).  Pure host arithmetic: the optimizer kernel takes lr per call."""
from __future__ import annotations

import math


class CosineAnnealingWarmRestarts:
    """Same recurrence as torch.optim.lr_scheduler.CosineAnnealingWarmRestarts.step() without an epoch argument:
    lr = eta_min + (base_lr - eta_min) * (1 + cos(pi * T_cur / T_i)) / 2, T_i multiplied by T_mult at each restart."""

    def __init__(self, base_lr, T_0=10, T_mult=2, eta_min=0.0):
        if T_0 <= 0 or T_mult < 1:
            raise ValueError("T_0 must be positive and T_mult >= 1")
        self.base_lr, self.T_0, self.T_mult, self.eta_min = float(base_lr), int(T_0), int(T_mult), float(eta_min)
        self.T_i, self.T_cur, self.last_epoch = self.T_0, 0, 0

    def get_lr(self):
        return self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * self.T_cur / self.T_i)) / 2

    def step(self):
        self.last_epoch += 1
        self.T_cur += 1
        if self.T_cur >= self.T_i:
            self.T_cur -= self.T_i
            self.T_i *= self.T_mult
        return self.get_lr()

    def state_dict(self):
        return dict(self.__dict__)

    def load_state_dict(self, sd):
        self.__dict__.update(sd)
