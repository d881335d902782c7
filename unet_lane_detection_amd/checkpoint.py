"""Checkpoint files in the reference's format (reference README.md:2208-2213, :2225-2231, :2876-2880):

    torch.save({'epoch', 'model_state_dict', 'optimizer_state_dict', 'best_dice'}, path)   # best model
    torch.save({'epoch', 'model_state_dict'}, path)                                         # periodic
    torch.save(model.state_dict(), path)                                                    # bare

`optimizer_state_dict` follows torch.optim.Adam/AdamW.state_dict(): parameters are numbered in
`model.parameters()` order, which for the reference module is the state_dict order without the BatchNorm
buffers - the same order as the flat buffers of the HIP trainer.  Files written here load into the
reference's PyTorch model/optimizer, and the reference's files load here (torch.load(weights_only=True))."""
from __future__ import annotations

import torch


def optimizer_state_dict(param_entries, exp_avg, exp_avg_sq, step, lr, betas, eps, weight_decay, decoupled):
    """param_entries: [(name, offset, numel, shape)] in parameter order; exp_avg / exp_avg_sq: flat tensors."""
    state = {}
    for i, (_, off, numel, shape) in enumerate(param_entries):
        state[i] = {"step": torch.tensor(float(step)),
                    "exp_avg": exp_avg[off:off + numel].detach().cpu().view(shape).clone(),
                    "exp_avg_sq": exp_avg_sq[off:off + numel].detach().cpu().view(shape).clone()}
    group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(param_entries)))}
    if decoupled:
        group["decoupled_weight_decay"] = True
    return {"state": state if step > 0 else {}, "param_groups": [group]}


def load_optimizer_state(osd, param_entries, exp_avg, exp_avg_sq):
    """Fill the flat moment buffers from a torch Adam/AdamW state_dict; returns (step, param_group dict)."""
    step = 0
    for i, (_, off, numel, shape) in enumerate(param_entries):
        st = osd["state"].get(i)
        if st is None:
            exp_avg[off:off + numel].zero_()
            exp_avg_sq[off:off + numel].zero_()
            continue
        exp_avg[off:off + numel].copy_(st["exp_avg"].reshape(-1))
        exp_avg_sq[off:off + numel].copy_(st["exp_avg_sq"].reshape(-1))
        step = max(step, int(float(st["step"])))
    return step, osd["param_groups"][0]


def save(path, model_state_dict, epoch=None, optimizer_state=None, best_dice=None):
    if epoch is None and optimizer_state is None and best_dice is None:
        torch.save(model_state_dict, path)                      # bare state_dict (README.md:2231)
        return
    ck = {"epoch": epoch, "model_state_dict": model_state_dict}
    if optimizer_state is not None:
        ck["optimizer_state_dict"] = optimizer_state
    if best_dice is not None:
        ck["best_dice"] = best_dice
    torch.save(ck, path)


def load(path):
    """Returns (model_state_dict, rest) where rest holds epoch / optimizer_state_dict / best_dice if present."""
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(obj, dict) and "model_state_dict" in obj:       # README.md:2876-2880
        return obj["model_state_dict"], {k: v for k, v in obj.items() if k != "model_state_dict"}
    return obj, {}
