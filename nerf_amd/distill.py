"""Distillation of a teacher field into a student (reference nerf/distill.py:22-150).

Same names, arguments and order of operations as the reference's `Domain`, `VoxelSampler` and
`distill`; teacher and student run on the HIP MLP kernels (the student with its backward), the
loss of distill.py:123-130 and its gradient are one fused kernel, and the sampler draws on the
device.  Note the reference's loss pairs every density with every segment length
(`sigma.view(-1, 1) * delta` with `delta` of shape (B,)); that is kept.
"""
from typing import Callable, List, NamedTuple, Optional, Tuple

import torch
from torch import Tensor
from torch.nn.utils import clip_grad_norm_
from torch.optim import Optimizer

from . import ops
from .core.model import NeRF
from .core.scheduler import Scheduler
from .utils.history import History
from .utils.pbar import tqdm

GRAD_NORM_CLIP = 0.
LOG_EVERY = 32          # steps between fetches of the device-side loss log


class Domain(NamedTuple):
    """Limits of the sampled volume and of the segment lengths (reference distill.py:22-27)."""
    x: Tuple[float, float]
    y: Tuple[float, float]
    z: Tuple[float, float]
    d: Tuple[float, float]


class VoxelSampler:
    """Batches of query points for distillation (reference distill.py:30-82): `voxels` random cells
    of a regular voxels^3 grid over the domain, `samples` jittered points in each, random
    directions and segment lengths.  Attributes as in the reference (`domain`, `voxels`, `samples`,
    `batch_size`, `voxel_size`, `rx` = the grid's cell centres, len() = their number); `sample()`
    is one launch of the sampling kernel on `device`, drawing from torch's generator of that device."""

    def __init__(self, domain: Domain, voxels: int, samples: int, device=None) -> None:
        self.domain, self.voxels, self.samples = domain, voxels, samples
        self.batch_size = voxels * samples
        self.device = torch.device("cuda" if device is None else device)
        extent = [abs(hi - lo) for lo, hi in (domain.x, domain.y, domain.z)]
        self.voxel_size = torch.tensor(extent, dtype=torch.float32, device=self.device) / voxels
        self._centres = None

    @property
    def rx(self) -> Tensor:
        """(voxels^3, 3) cell centres, x slowest; built on first use (the kernel does not read it)."""
        if self._centres is None:
            axes = [torch.linspace(lo, hi, steps=self.voxels, dtype=torch.float32, device=self.device)
                    for lo, hi in (self.domain.x, self.domain.y, self.domain.z)]
            self._centres = torch.cartesian_prod(*axes)
        return self._centres

    def __len__(self) -> int:
        return self.voxels ** 3

    def sample(self) -> Tuple[Tensor, Tensor, Tensor]:
        """-> rx (B, 3), rd (B, 3), delta (B,) with B = voxels * samples."""
        return ops.voxel_sample(self.domain, self.voxels, self.samples, self.device)


def distill(teacher: NeRF, student: NeRF, sampler: VoxelSampler, optim: Optimizer, scheduler: Scheduler, scaler,
            iterations: Optional[int] = 100, callbacks: Optional[List[Callable[[int, History], None]]] = [],
            verbose: bool = True) -> History:
    """Train `student` to reproduce `teacher`'s opacity and colour on sampled points (reference
    distill.py:85-147).  The per-step loss stays on the device and is fetched every LOG_EVERY
    steps; callbacks therefore see the history up to the last fetch (always complete at the end
    and whenever the number of callbacks is non-zero and a fetch is due)."""
    d = next(teacher.parameters()).device
    H = History()
    log = torch.zeros((iterations,), dtype=torch.float32, device=d)
    fetched = 0

    def fetch(upto: int) -> None:
        nonlocal fetched
        for value in log[fetched:upto].tolist():
            H.train.append((value, None))
        fetched = max(fetched, upto)

    steps = tqdm(range(iterations), desc="[NeRF] Distill Step", disable=not verbose)
    for step in steps:
        rx, rd, delta = sampler.sample()
        rx, rd, delta = rx.to(d), rd.to(d), delta.to(d)

        with torch.autocast("cuda", enabled=scaler.is_enabled()):
            with torch.no_grad():
                t_sigma, t_rgb = teacher(rx, rd)
            s_sigma, s_rgb = student(rx, rd)
            loss = ops.distill_loss(s_sigma, s_rgb, t_sigma, t_rgb, delta)

        scaler.scale(loss).backward()
        scaler.unscale_(optim)
        if GRAD_NORM_CLIP > 0.:
            clip_grad_norm_(student.parameters(), GRAD_NORM_CLIP)
        scaler.step(optim)
        scaler.update()
        optim.zero_grad(set_to_none=True)
        scheduler.step()

        log[step] = loss.detach()
        H.lr.append(scheduler.lr)
        if callbacks or (step + 1) % LOG_EVERY == 0 or step + 1 == iterations:
            fetch(step + 1)
            steps.set_postfix(loss=f"{H.train[-1][0]:.2e}")
        for callback in callbacks:
            callback(step, H)
    return H


NeRF.distill = distill
