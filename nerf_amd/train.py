"""Training loop (reference nerf/train.py:27-180).

`step` and `fit` keep the reference's signatures, return values and the order of
operations of one optimisation step (train.py:75-86):
    autocast -> render_volume(nerf, nerf, train=True) -> criterion
    scaler.scale(loss).backward() -> scaler.unscale_ -> [clip] -> scaler.step
    -> scaler.update -> optim.zero_grad(set_to_none=True) -> scheduler.step
The render and its backward run on the HIP kernels; `autocast` selects the bf16 MFMA
contractions (the loss scale of a GradScaler then only rescales fp32 gradients, which
is harmless).  Optimizer, scaler, criterion and scheduler are the caller's objects; a plain
`MSELoss()` is computed by the fused loss kernel (loss, PSNR and gradient seed in one pass).
Losses and PSNRs are kept on the device and fetched every `LOG_EVERY` steps (and at the end of
the pass), not with four `.item()` calls per step as the reference does (train.py:94-97): the
returned lists are the same, the progress bar refreshes less often, and with
`nerf_amd.optim.Adam` no step waits for the host; behind a GradScaler that optimizer also takes over
the scaler's unscale / inf check / skipped step (nerf_amd.amp), with the same results.
When torch.distributed is initialised with more than one rank, every rank runs this loop on
its own batches and the gradients are averaged (one flat all-reduce) before `unscale_`.
"""
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor, device
from torch.nn import Module
from torch.nn.utils import clip_grad_norm_
from torch.optim import Optimizer
from torch.utils.data import DataLoader, Dataset

from . import amp, ops
from .core.model import NeRF
from .core.renderer import BoundedVolumeRaymarcher as BVR
from .core.scheduler import Scheduler
from .data.utils import loaders
from .distributed import average_gradients, world
from .optim import Adam as FlatAdam
from .utils.history import History
from .utils.pbar import tqdm

GRAD_NORM_CLIP = 0.
LOG_EVERY = 32          # steps between fetches of the device-side loss / PSNR log


def batch_step(nerf: NeRF, raymarcher: BVR, optim: Optimizer, scheduler: Scheduler, criterion: Module, scaler,
               C: Tensor, ro: Tensor, rd: Tensor, perturb: bool, log_row: Tensor) -> None:
    """One optimisation step on one batch that is already on the device (reference train.py:75-86);
    the loss and its PSNR go to `log_row` (2 device floats).

    With a mean `MSELoss` the loss, the PSNR and the seed of the backward come from one kernel and the
    graph starts at the rendered colours (no loss node, no fill, no multiplies).  Behind an enabled
    GradScaler with `nerf_amd.optim.Adam`, in a single process and without gradient clipping, the
    scaler's four calls cost no launches of their own either (see nerf_amd.amp); otherwise they are
    made as the reference makes them."""
    d = ro.device
    fused_mse = type(criterion) is torch.nn.MSELoss and criterion.reduction == "mean"
    with torch.autocast("cuda", enabled=scaler.is_enabled()):
        *_, C_ = raymarcher.render_volume(nerf, nerf, ro, rd, perturb=perturb, train=True)
        if not fused_mse:
            loss = criterion(C_, C)
    scale = None
    if fused_mse and isinstance(optim, FlatAdam) and GRAD_NORM_CLIP <= 0. and world()[1] == 1:
        scale = amp.loss_scale(scaler, d)
    if scale is not None:                                   # scale / backward / unscale_ / step / update, fused
        found_inf = torch.empty(1, dtype=torch.float32, device=d)
        seed = ops.mse_seed(C_, C, log_row, scale, found_inf)
        ops.GRAD_CHECK = found_inf
        try:
            C_.backward(seed)
        finally:
            ops.GRAD_CHECK = None
        optim.step_scaled(scale, found_inf)
        amp.record_step(scaler, optim, found_inf)
        scaler.update()
    else:
        if fused_mse and not scaler.is_enabled():
            C_.backward(ops.mse_seed(C_, C, log_row))
        elif fused_mse:
            scaler.scale(ops.mse_loss(C_, C, log_row)).backward()
        else:
            scaler.scale(loss).backward()
            with torch.no_grad():
                log_row[0] = loss.detach()
                log_row[1] = -10. * torch.log10(loss.detach())
        average_gradients(nerf)          # data parallel: no-op in a single process
        scaler.unscale_(optim)
        if GRAD_NORM_CLIP > 0.:
            clip_grad_norm_(nerf.parameters(), GRAD_NORM_CLIP)
        scaler.step(optim)
        scaler.update()
    optim.zero_grad(set_to_none=True)
    scheduler.step()


def step(epoch: int, nerf: NeRF, raymarcher: BVR, optim: Optimizer, scheduler: Scheduler, criterion: Module,
         scaler, loader: DataLoader, d: device, split: str, perturb: Optional[bool] = False,
         verbose: Optional[bool] = True) -> Tuple[List[float], List[float], List[float]]:
    """One pass over `loader` (reference train.py:27-101) -> (losses, psnrs, lrs) per step."""
    train = split == "train"
    nerf = nerf.train(train)

    total_loss, total_psnr, total_lr = 0., 0., 0.
    steps_loss, steps_psnr, steps_lr = [], [], []
    batches = tqdm(loader, desc=f"[NeRF] {split.capitalize()} {epoch + 1}", disable=not verbose)
    fused_mse = type(criterion) is torch.nn.MSELoss and criterion.reduction == "mean"
    log = torch.zeros((len(loader), 2), dtype=torch.float32, device=d)      # loss, psnr per step
    fetched = 0

    def fetch(upto: int) -> None:
        nonlocal fetched, total_loss, total_psnr
        if upto <= fetched:
            return
        for loss_k, psnr_k in log[fetched:upto].tolist():
            total_loss += loss_k / len(loader)
            total_psnr += psnr_k / len(loader)
            steps_loss.append(loss_k)
            steps_psnr.append(psnr_k)
        fetched = upto
        batches.set_postfix(loss=total_loss, psnr=total_psnr, lr=total_lr)

    with torch.set_grad_enabled(train):
        for k, (C, ro, rd) in enumerate(batches):
            C, ro, rd = C.to(d), ro.to(d), rd.to(d)
            if train:
                batch_step(nerf, raymarcher, optim, scheduler, criterion, scaler, C, ro, rd, perturb, log[k])
                total_lr += scheduler.lr / len(loader)
                steps_lr.append(scheduler.lr)
            else:
                with torch.autocast("cuda", enabled=scaler.is_enabled()):
                    *_, C_ = raymarcher.render_volume(nerf, nerf, ro, rd, perturb=perturb, train=False)
                    if fused_mse:
                        ops.mse_loss(C_, C, log[k])
                    else:
                        loss = criterion(C_, C)
                        log[k, 0] = loss
                        log[k, 1] = -10. * torch.log10(loss)
            if (k + 1) % LOG_EVERY == 0:
                fetch(k + 1)
        fetch(len(loader))

    return steps_loss, steps_psnr, steps_lr


def fit(nerf: NeRF, raymarcher: BVR, optim: Optimizer, scheduler: Scheduler, criterion: Module, scaler,
        train_data: Dataset, val_data: Optional[Dataset] = None, test_data: Optional[Dataset] = None,
        epochs: Optional[int] = 100, batch_size: Optional[int] = 32, jobs: Optional[int] = 8,
        perturb: Optional[bool] = False, callbacks: Optional[List[Callable[[int, History], None]]] = [],
        verbose: bool = True) -> History:
    """Fit `nerf` on a ray dataset (reference train.py:104-177): per epoch a train pass, an
    optional validation pass and the callbacks; a test pass at the end."""
    d = next(nerf.parameters()).device
    data = dict(zip(("train", "val", "test"), loaders(train_data, val_data, test_data, batch_size, jobs, device=d)))
    H = History()
    bar = tqdm(range(epochs), desc="[NeRF] Epoch", disable=not verbose)

    def run(split: str, epoch: int, **options):
        """One pass over a split -> ((mean loss, mean psnr), learning rates of its steps)."""
        mse, psnr, lrs = step(epoch, nerf, raymarcher, optim, scheduler, criterion, scaler, data[split], d,
                              split=split, verbose=verbose, **options)
        summary = np.average(mse), np.average(psnr)
        bar.set_postfix(mse=summary[0], psnr=summary[1])
        return summary, lrs

    def notify(epoch: int) -> None:
        for callback in callbacks:
            callback(epoch, H)

    last = 0
    for last in bar:
        summary, lrs = run("train", last, perturb=perturb)
        H.train.append(summary)
        H.lr.extend(lrs)
        if data["val"]:
            H.val.append(run("val", last)[0])
        notify(last)
    if data["test"]:                                  # after the last epoch, under its number (train.py:169-175)
        H.test = run("test", last)[0]
        notify(last)
    return H


NeRF.fit = fit
