"""Adam over one flat parameter buffer, one kernel per step.

`torch.optim.Adam` (what the reference's CLI builds, train.py:246) launches a handful of
multi-tensor kernels per step and, behind a GradScaler, makes the host wait for the inf check
before it may launch them.  This optimizer keeps the parameters, both moments and the update
count in flat device buffers and applies the same update in a single launch that also honours
the scaler's `found_inf` on the device, so a training step never waits for the host.  It plugs
into `torch.amp.GradScaler` through the protocol torch's own fused optimizers use
(`_step_supports_amp_scaling`).  Same hyper-parameters and arithmetic as `torch.optim.Adam`
(no amsgrad, no maximize); results agree to fp32 rounding.
"""
from typing import Iterable

import torch
from torch import Tensor

from . import ops
from .distributed import flat_gradient


class Adam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True

    def __init__(self, params: Iterable[Tensor], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        for group in self.param_groups:
            self._flatten(group)

    def _flatten(self, group) -> None:
        """Move the group's parameters into one contiguous buffer; each parameter becomes a view."""
        params = group["params"]
        first = params[0]
        if any(not p.is_cuda or p.dtype != torch.float32 or p.device != first.device for p in params):
            raise RuntimeError("nerf_amd.optim.Adam takes fp32 parameters on one ROCm device")
        sizes = [p.numel() for p in params]
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        for p, v in zip(params, flat.split(sizes)):
            p.data = v.view(p.shape)
        state = self.state[first]                       # the group's state hangs off its first parameter
        state["flat_param"] = flat
        state["sizes"] = sizes
        state["exp_avg"] = torch.zeros_like(flat)
        state["exp_avg_sq"] = torch.zeros_like(flat)
        state["step"] = torch.zeros(2, device=flat.device)          # [applied so far, scratch], swapped per call
        state["which"] = 0
        state["flat_grad"] = torch.zeros_like(flat)                 # used only when the gradients are not flat

    def _ensure_flat(self, group) -> None:
        """Parameters must still be the views made by _flatten (a later `module.to(...)`,
        `load_state_dict(assign=True)` or `p.data = ...` replaces them): if not, adopt the current
        values into the flat buffer again and keep the moments."""
        params = group["params"]
        state = self.state[params[0]]
        flat = state["flat_param"]
        ptr = flat.data_ptr()
        for p, n in zip(params, state["sizes"]):
            if p.data_ptr() != ptr or p.device != flat.device or not p.is_contiguous():
                break
            ptr += 4 * n
        else:
            return
        if any(p.device != flat.device or p.dtype != torch.float32 for p in params):
            raise RuntimeError("nerf_amd.optim.Adam: parameters moved to another device or dtype after construction")
        with torch.no_grad():
            for p, v in zip(params, flat.split(state["sizes"])):
                v.copy_(p.detach().reshape(-1))
                p.data = v.view(p.shape)

    def load_state_dict(self, state_dict) -> None:
        """As torch's, then the flat buffers go back onto the parameters' device as fp32 (a
        checkpoint read with map_location="cpu" would otherwise hand host pointers to the kernel)
        and the parameters are re-adopted as views of the loaded flat buffer's values."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            params = group["params"]
            state = self.state[params[0]]
            dev = params[0].device
            for key in ("flat_param", "exp_avg", "exp_avg_sq", "step", "flat_grad"):
                state[key] = state[key].to(device=dev, dtype=torch.float32).contiguous()
            state["which"] = int(state["which"])
            state["sizes"] = [int(n) for n in state["sizes"]]
            # the live parameters (what load_state_dict of the module set) win over the saved copy
            with torch.no_grad():
                for p, v in zip(params, state["flat_param"].split(state["sizes"])):
                    v.copy_(p.detach().reshape(-1))
                    p.data = v.view(p.shape)

    @property
    def steps_applied(self) -> int:
        """Number of updates actually applied to the first group (synchronises)."""
        state = self.state[self.param_groups[0]["params"][0]]
        return int(state["step"][state["which"]].item())

    @torch.no_grad()
    def step_scaled(self, grad_scale: Tensor, found_inf: Tensor) -> None:
        """`scaler.unscale_(self); scaler.step(self)` (reference train.py:81-83) as one launch: the
        gradients still carry the loss scale `grad_scale` (the scaler's device float), `found_inf` is the
        device float the backward raised if a gradient was not finite (nerf_amd.amp)."""
        self._update(grad_scale, found_inf)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        found_inf = getattr(self, "found_inf", None)        # set by GradScaler.step around this call
        grad_scale = getattr(self, "grad_scale", None)      # None once unscale_ has run
        self._update(grad_scale, found_inf)
        return loss

    def _update(self, grad_scale, found_inf):
        for group in self.param_groups:
            params = group["params"]
            state = self.state[params[0]]
            if all(p.grad is None for p in params):
                continue
            self._ensure_flat(group)
            grad = flat_gradient(params)
            if grad is None:                                # gradients from elsewhere: gather them
                if any(p.grad is None for p in params):
                    # torch.optim.Adam leaves a parameter without a gradient untouched (moments
                    # included); one flat update cannot, so say so instead of moving those weights
                    raise RuntimeError("nerf_amd.optim.Adam: every parameter of a group needs a gradient "
                                       "(put frozen parameters in a group of their own or leave them out)")
                grad = state["flat_grad"]
                for p, v in zip(params, grad.split(state["sizes"])):
                    v.copy_(p.grad.reshape(-1))
            scale = None if grad_scale is None else grad_scale.to(torch.float32).reshape(1)   # no copy when already fp32
            finf = None if found_inf is None else found_inf.to(torch.float32).reshape(1)
            w = state["which"]
            step = state["step"]
            ops.adam_step(state["flat_param"], grad, state["exp_avg"], state["exp_avg_sq"], group["lr"],
                          group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"],
                          None, finf, step[w:w + 1], step[1 - w:2 - w], grad_scale=scale)
            state["which"] = 1 - w
            for p in params:                                # written through raw pointers: tell autograd
                torch.autograd.graph.increment_version(p)   # and the packed-weight cache of ops.ModelHandle
