"""A `torch.amp.GradScaler` driven without its extra passes over the gradients.

The reference's step is `scaler.scale(loss).backward(); scaler.unscale_(optim); scaler.step(optim);
scaler.update()` (train.py:80-85).  Through torch alone that is a dozen small launches around the
render: the scale multiply and its backward, a fill for the graph's root, the reciprocal of the
scale, a multi-tensor pass over all gradients for the inf check, and a host wait before a plain
optimizer may step.  Here the same four calls' effects come from kernels that are running anyway:

* `ops.mse_seed` writes d (scale x loss) / d pred straight from the scaler's device float;
* the backward that writes the gradients raises `found_inf` if one is not finite (`ops.GRAD_CHECK`);
* `nerf_amd.optim.Adam.step_scaled` divides by the scale and skips the update on `found_inf`;
* `record_step` leaves in the scaler what `unscale_` + `step` leave there, so that its own
  `update()` grows or backs off the scale exactly as it always does.

The scaler stays the caller's object: `get_scale()`, `state_dict()`, growth and back-off are torch's.
`loss_scale` returns None when a scaler cannot be driven this way; callers then take the plain path.
"""
from typing import Optional

import torch
from torch import Tensor

try:
    from torch.amp.grad_scaler import OptState
except ImportError:                                        # another torch layout: the plain path is used
    OptState = None

_NEEDS = ("_scale", "_lazy_init_scale_growth_tracker", "_per_optimizer_states")


def loss_scale(scaler, device) -> Optional[Tensor]:
    """The loss scale as the device float the scaler keeps (made on first use, as `scaler.scale()`
    makes it), or None: scaler disabled, on another device, or with other internals than expected."""
    if OptState is None or not scaler.is_enabled() or not all(hasattr(scaler, a) for a in _NEEDS):
        return None
    if scaler._scale is None:
        scaler._lazy_init_scale_growth_tracker(torch.device(device))
    scale = scaler._scale
    if scale.device != torch.device(device) or scale.dtype != torch.float32 or scale.numel() != 1:
        return None
    return scale


def record_step(scaler, optim, found_inf: Tensor) -> None:
    """After `optim.step_scaled(scale, found_inf)`: the per-optimizer record `scaler.update()` reads."""
    state = scaler._per_optimizer_states[id(optim)]
    if state["stage"] is not OptState.READY:
        raise RuntimeError("this optimizer was already unscaled or stepped since the scaler's last update()")
    state["found_inf_per_device"] = {found_inf.device: found_inf}
    state["stage"] = OptState.STEPPED
