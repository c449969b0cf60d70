"""Training loop (reference nerf/train.py:27-180).

`step` and `fit` keep the reference's signatures, return values and the order of
operations of one optimisation step (train.py:75-86):
    autocast -> render_volume(nerf, nerf, train=True) -> criterion
    scaler.scale(loss).backward() -> scaler.unscale_ -> [clip] -> scaler.step
    -> scaler.update -> optim.zero_grad(set_to_none=True) -> scheduler.step
The render and its backward run on the HIP kernels; `autocast` selects the bf16 MFMA
contractions (the loss scale of a GradScaler then only rescales fp32 gradients, which
is harmless).  Optimizer, scaler, criterion and scheduler are the caller's objects.
When torch.distributed is initialised with more than one rank, every rank runs this loop on
its own batches and the gradients are averaged (one flat all-reduce) before `unscale_`.
"""
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
from torch import device
from torch.nn import Module
from torch.nn.utils import clip_grad_norm_
from torch.optim import Optimizer
from torch.utils.data import DataLoader, Dataset

from .core.model import NeRF
from .core.renderer import BoundedVolumeRaymarcher as BVR
from .core.scheduler import Scheduler
from .data.utils import loaders
from .distributed import average_gradients
from .utils.history import History
from .utils.pbar import tqdm

GRAD_NORM_CLIP = 0.


def step(epoch: int, nerf: NeRF, raymarcher: BVR, optim: Optimizer, scheduler: Scheduler, criterion: Module,
         scaler, loader: DataLoader, d: device, split: str, perturb: Optional[bool] = False,
         verbose: Optional[bool] = True) -> Tuple[List[float], List[float], List[float]]:
    """One pass over `loader` (reference train.py:27-101) -> (losses, psnrs, lrs) per step."""
    train = split == "train"
    nerf = nerf.train(train)

    total_loss, total_psnr, total_lr = 0., 0., 0.
    steps_loss, steps_psnr, steps_lr = [], [], []
    batches = tqdm(loader, desc=f"[NeRF] {split.capitalize()} {epoch + 1}", disable=not verbose)

    with torch.set_grad_enabled(train):
        for C, ro, rd in batches:
            C, ro, rd = C.to(d), ro.to(d), rd.to(d)

            with torch.autocast("cuda", enabled=scaler.is_enabled()):
                *_, C_ = raymarcher.render_volume(nerf, nerf, ro, rd, perturb=perturb, train=train)
                loss = criterion(C_, C)

            if train:
                scaler.scale(loss).backward()
                average_gradients(nerf)          # data parallel: no-op in a single process
                scaler.unscale_(optim)
                if GRAD_NORM_CLIP > 0.:
                    clip_grad_norm_(nerf.parameters(), GRAD_NORM_CLIP)
                scaler.step(optim)
                scaler.update()
                optim.zero_grad(set_to_none=True)
                scheduler.step()

                total_lr += scheduler.lr / len(loader)
                steps_lr.append(scheduler.lr)

            with torch.no_grad():
                psnr = -10. * torch.log10(loss)

            total_loss += loss.item() / len(loader)
            total_psnr += psnr.item() / len(loader)
            steps_loss.append(loss.item())
            steps_psnr.append(psnr.item())
            batches.set_postfix(loss=total_loss, psnr=total_psnr, lr=total_lr)

    return steps_loss, steps_psnr, steps_lr


def fit(nerf: NeRF, raymarcher: BVR, optim: Optimizer, scheduler: Scheduler, criterion: Module, scaler,
        train_data: Dataset, val_data: Optional[Dataset] = None, test_data: Optional[Dataset] = None,
        epochs: Optional[int] = 100, batch_size: Optional[int] = 32, jobs: Optional[int] = 8,
        perturb: Optional[bool] = False, callbacks: Optional[List[Callable[[int, History], None]]] = [],
        verbose: bool = True) -> History:
    """Fit `nerf` on a ray dataset (reference train.py:104-177): per epoch a train pass, an
    optional validation pass and the callbacks; a test pass at the end."""
    train, val, test = loaders(train_data, val_data, test_data, batch_size, jobs)

    H = History()
    d = next(nerf.parameters()).device
    args = nerf, raymarcher, optim, scheduler, criterion, scaler

    pbar = tqdm(range(epochs), desc="[NeRF] Epoch", disable=not verbose)
    for epoch in pbar:
        mse, psnr, lr = step(epoch, *args, train, d, split="train", perturb=perturb, verbose=verbose)
        H.train.append((np.average(mse), np.average(psnr)))
        H.lr += lr
        pbar.set_postfix(mse=np.average(mse), psnr=np.average(psnr))

        if val:
            mse, psnr, _ = step(epoch, *args, val, d, split="val", verbose=verbose)
            H.val.append((np.average(mse), np.average(psnr)))
            pbar.set_postfix(mse=np.average(mse), psnr=np.average(psnr))

        for callback in callbacks:
            callback(epoch, H)

    if test:
        mse, psnr, _ = step(epoch, *args, test, d, split="test", verbose=verbose)
        H.test = np.average(mse), np.average(psnr)
        pbar.set_postfix(mse=np.average(mse), psnr=np.average(psnr))

        for callback in callbacks:
            callback(epoch, H)

    return H


NeRF.fit = fit
