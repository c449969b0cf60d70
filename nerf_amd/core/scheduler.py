"""Learning-rate schedules (reference nerf/core/scheduler.py): host-side scalar
math that writes `param_group["lr"]`; nothing here touches the device."""
import math
from typing import Optional, Tuple

from torch.optim import Optimizer


class Scheduler:
    """Counts optimizer steps and pushes `self.lr` into every param group
    (reference scheduler.py:7-32)."""

    def __init__(self, optim: Optimizer) -> None:
        self.optim = optim
        self.current_step = 0.

    def step(self) -> "Scheduler":
        self.current_step += 1.
        self.update_optim()
        return self

    def update_optim(self) -> None:
        value = self.lr
        for group in self.optim.param_groups:
            group["lr"] = value

    @property
    def lr(self) -> float:
        raise NotImplementedError("Lr not implemented yet!")


class IndendityScheduler(Scheduler):
    """Constant rate.  The (misspelled) name and the unused `epochs` argument are
    the reference's public API (scheduler.py:35-57, core/__init__.py:4)."""

    def __init__(self, optim: Optimizer, epochs: float, lr: Optional[float] = 5e-4) -> None:
        super().__init__(optim)
        self._lr = lr
        self.update_optim()

    @property
    def lr(self) -> float:
        return self._lr


class LogDecayScheduler(Scheduler):
    """lr_min + (lr_max - lr_min) (1 - 10^(-power (1 - t))), t = step / total
    (reference scheduler.py:60-94)."""

    def __init__(self, optim: Optimizer, epochs: float, steps_per_epoch: float,
                 lr_range: Optional[Tuple[float, float]] = (5e-6, 5e-4), power: float = 8) -> None:
        super().__init__(optim)
        self.epochs = epochs
        self.steps_per_epoch = steps_per_epoch
        self.lr_min, self.lr_max = lr_range
        self.power = power
        self.steps = epochs * steps_per_epoch
        self.update_optim()

    @property
    def lr(self) -> float:
        remaining = 1 - self.current_step / self.steps
        return self.lr_min + (self.lr_max - self.lr_min) * (1 - 10 ** (-self.power * remaining))


class MipNeRFScheduler(Scheduler):
    """Log-linear decay from lr_max to lr_min with a sine warm-up over
    `epochs_shift` epochs (reference scheduler.py:97-140)."""

    def __init__(self, optim: Optimizer, epochs: float, epochs_shift: float, steps_per_epoch: float,
                 lr_range: Optional[Tuple[float, float]] = (5e-6, 5e-4), scale: Optional[float] = 1e-2) -> None:
        super().__init__(optim)
        self.epochs = epochs
        self.epochs_shift = epochs_shift
        self.steps_per_epoch = steps_per_epoch
        self.lr_min, self.lr_max = lr_range
        self.scale = scale
        self.steps_shift = epochs_shift * steps_per_epoch
        self.steps = epochs * steps_per_epoch
        self.update_optim()

    @property
    def lr(self) -> float:
        warm = min(max(self.current_step / self.steps_shift, 0.), 1.)
        gain = (1. - self.scale) * math.sin(.5 * math.pi * warm) + self.scale
        t = self.current_step / self.steps
        return gain * math.exp((1. - t) * math.log(self.lr_max) + t * math.log(self.lr_min))
