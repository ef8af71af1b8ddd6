"""Learning-rate schedules of `diffusers.optimization.get_scheduler` as plain functions of the step count.

The reference builds `get_scheduler(cfg.optim.lr_scheduler, optimizer, num_warmup_steps=warmup*accum,
num_training_steps=max_train_steps*accum)` (training/coach.py:759-770) and calls `lr_scheduler.step()` once per
micro-iteration (coach.py:217).  The scheduler is passed through `accelerator.prepare` (coach.py:97-99), and
accelerate's AcceleratedScheduler only advances on iterations where the optimizer really stepped, then advances
`num_processes` times ([3p-memory] accelerate/scheduler.py, `split_batches=False`).  So after g optimizer steps
the multiplier in force is lambda(g * world) with the totals above.

diffusers is not installable here: the six schedule shapes are restated from its published definitions
(`get_constant_schedule`, `..._with_warmup`, `get_linear_schedule_with_warmup`, `get_cosine_schedule_with_warmup`
(num_cycles 0.5), `get_cosine_with_hard_restarts_schedule_with_warmup` (num_cycles 1),
`get_polynomial_decay_schedule_with_warmup` (lr_end 1e-7, power 1)) — PARITY UNPINNED like every diffusers piece.

The HIP engine keeps the learning rate in device memory (`TrainStepEngine.hyper[0]`), so the host writes the new
value between graph replays; nothing is re-captured.
"""
from __future__ import annotations

import math

SCHEDULES = ("linear", "cosine", "cosine_with_restarts", "polynomial", "constant", "constant_with_warmup")


def lr_lambda(name: str, step: int, num_warmup_steps: int, num_training_steps: int, lr_init: float = 1.0) -> float:
    """multiplier applied to the base learning rate after `step` scheduler steps"""
    if name not in SCHEDULES:
        raise ValueError(f"unknown optim.lr_scheduler '{name}' (choose from {SCHEDULES})")
    if name == "constant":
        return 1.0
    w, T = num_warmup_steps, num_training_steps
    if name == "constant_with_warmup":
        return float(step) / float(max(1.0, w)) if step < w else 1.0
    if step < w:
        return float(step) / float(max(1, w))
    if name == "linear":
        return max(0.0, float(T - step) / float(max(1, T - w)))
    progress = float(step - w) / float(max(1, T - w))
    if name == "cosine":
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress)))
    if name == "cosine_with_restarts":
        if progress >= 1.0:
            return 0.0
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((1.0 * progress) % 1.0))))
    # polynomial: lr_end = 1e-7, power = 1.0 (absolute end value, hence lr_init)
    lr_end, power = 1e-7, 1.0
    if step > T:
        return lr_end / lr_init
    decay = (lr_init - lr_end) * (1 - (step - w) / (T - w)) ** power + lr_end
    return decay / lr_init


class LRSchedule:
    """lr in force for the optimizer step that follows `g` completed optimizer steps"""

    def __init__(self, name: str, base_lr: float, lr_warmup_steps: int, max_train_steps: int, grad_accum: int,
                 world: int):
        if name not in SCHEDULES:
            raise ValueError(f"unknown optim.lr_scheduler '{name}' (choose from {SCHEDULES})")
        self.name, self.base_lr, self.world = name, base_lr, world
        self.warmup = lr_warmup_steps * grad_accum
        self.total = max_train_steps * grad_accum
        self.constant = name == "constant"

    def lr(self, optimizer_steps_done: int) -> float:
        return self.base_lr * lr_lambda(self.name, optimizer_steps_done * self.world, self.warmup, self.total,
                                        self.base_lr)
