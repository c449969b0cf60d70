"""Reptile initialisation (reference nerf/reptile.py:24-232): per batch, a copy of the field takes
`steps` SGD steps on that batch, the difference to the copy becomes the field's "gradient" for
the caller's optimizer, and the copy is reset to the updated field.  Names, arguments, return
values and the order of operations are the reference's; renders and their backward run on the
HIP kernels.  As in the reference, `val_data` / `test_data` are accepted and not used."""
import gc
from copy import deepcopy
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
from torch import device
from torch.nn import Module
from torch.nn.utils import clip_grad_norm_
from torch.optim import SGD, Optimizer
from torch.utils.data import DataLoader, Dataset

from .core.model import NeRF
from .core.renderer import BoundedVolumeRaymarcher as BVR
from .data.utils import loaders
from .utils.history import History
from .utils.pbar import tqdm

GRAD_NORM_CLIP = 0.


def build_meta(nerf: NeRF, optim: Optimizer, scaler) -> Tuple[NeRF, Optimizer, object]:
    """The inner-loop copy of the field, its SGD (momentum 0.9, 1000 x the outer learning rate) and
    its own scaler (reference reptile.py:24-47)."""
    optim_lr = optim.param_groups[0]["lr"]
    meta_nerf = deepcopy(nerf)
    meta_optim = SGD(meta_nerf.parameters(), lr=optim_lr * 1e3, momentum=0.9)
    meta_scaler = deepcopy(scaler)
    return meta_nerf, meta_optim, meta_scaler


class MetaStateHolder:
    """Resets the inner-loop objects after every outer update (reference reptile.py:50-82)."""

    def __init__(self, nerf: NeRF, meta_nerf: NeRF, meta_optim: Optimizer, meta_scaler) -> None:
        self.nerf, self.meta_nerf, self.meta_optim, self.meta_scaler = nerf, meta_nerf, meta_optim, meta_scaler
        self.meta_optim_is = self.meta_optim.state_dict()
        self.meta_scaler_is = self.meta_scaler.state_dict()

    @torch.no_grad()
    def update(self) -> None:
        self.meta_nerf.load_state_dict(self.nerf.state_dict())
        self.meta_optim.load_state_dict(self.meta_optim_is)
        self.meta_scaler.load_state_dict(self.meta_scaler_is)


def meta_step(epoch: int, nerf: NeRF, raymarcher: BVR, optim: Optimizer, criterion: Module, scaler,
              loader: DataLoader, d: device, steps: Optional[int] = 16, perturb: Optional[bool] = False,
              verbose: Optional[bool] = True) -> Tuple[List[float], List[float]]:
    """One pass over `loader` (reference reptile.py:85-172) -> (losses, psnrs) per batch."""
    nerf = nerf.train()
    meta_nerf, meta_optim, meta_scaler = build_meta(nerf, optim, scaler)
    meta_state = MetaStateHolder(nerf, meta_nerf, meta_optim, meta_scaler)

    total_loss, total_psnr = 0., 0.
    steps_loss, steps_psnr = [], []

    batches = tqdm(loader, desc=f"[NeRF] Meta {epoch + 1}", disable=not verbose)
    for C, ro, rd in batches:
        C, ro, rd = C.to(d), ro.to(d), rd.to(d)

        for _ in range(steps):
            with torch.autocast("cuda", enabled=scaler.is_enabled()):
                *_, C_ = raymarcher.render_volume(meta_nerf, meta_nerf, ro, rd, perturb=perturb, train=True)
                meta_loss = criterion(C_, C)

            meta_scaler.scale(meta_loss).backward()
            meta_scaler.unscale_(meta_optim)
            if GRAD_NORM_CLIP > 0.:
                clip_grad_norm_(meta_nerf.parameters(), GRAD_NORM_CLIP)
            meta_scaler.step(meta_optim)
            meta_scaler.update()
            meta_optim.zero_grad(set_to_none=True)

        with torch.no_grad():
            for p, meta_p in zip(nerf.parameters(), meta_nerf.parameters()):
                p.grad = scaler.scale(p - meta_p)

            with torch.autocast("cuda", enabled=scaler.is_enabled()):
                *_, C_ = raymarcher.render_volume(nerf, nerf, ro, rd, perturb=perturb, train=False)
                loss = criterion(C_, C)

            psnr = -10. * torch.log10(loss)

        scaler.step(optim)
        scaler.update()
        optim.zero_grad(set_to_none=True)
        meta_state.update()

        total_loss += loss.item() / len(loader)
        total_psnr += psnr.item() / len(loader)
        steps_loss.append(loss.item())
        steps_psnr.append(psnr.item())
        batches.set_postfix(loss=total_loss, psnr=total_psnr)

    del meta_nerf, meta_optim, meta_scaler
    torch.cuda.empty_cache()
    gc.collect()
    return steps_loss, steps_psnr


def reptile(nerf: NeRF, raymarcher: BVR, optim: Optimizer, criterion: Module, scaler, train_data: Dataset,
            val_data: Optional[Dataset] = None, test_data: Optional[Dataset] = None, epochs: Optional[int] = 100,
            steps: Optional[int] = 16, batch_size: Optional[int] = 32, jobs: Optional[int] = 8,
            perturb: Optional[bool] = False, callbacks: Optional[List[Callable[[int, History], None]]] = [],
            verbose: bool = True) -> History:
    """Reptile initialisation of `nerf` on a ray dataset (reference reptile.py:175-229)."""
    d = next(nerf.parameters()).device
    train, _, _ = loaders(train_data, val_data, test_data, batch_size, jobs, device=d)

    H = History()
    pbar = tqdm(range(epochs), desc="[NeRF] Epoch", disable=not verbose)
    for epoch in pbar:
        mse, psnr = meta_step(epoch, nerf, raymarcher, optim, criterion, scaler, train, d, steps=steps,
                              perturb=perturb, verbose=verbose)
        H.train.append((np.average(mse), np.average(psnr)))
        pbar.set_postfix(mse=np.average(mse), psnr=np.average(psnr))
        for callback in callbacks:
            callback(epoch, H)
    return H


NeRF.reptile = reptile
