"""Output activations of the radiance field (reference nerf/core/activations.py).

On the render/train path both are fused into the MLP kernel's epilogue; these
modules exist so that `NeRF` has the reference's submodule layout (and
state_dict keys) and for stand-alone use on any device."""
import torch
from torch import Tensor
from torch.nn import Module


def widened_sigmoid(x: Tensor) -> Tensor:
    """Sigmoid stretched to (-0.001, 1.001) (reference activations.py:18-20)."""
    return 0.5 + 0.501 * torch.tanh(0.5 * x)


def shifted_softplus(x: Tensor) -> Tensor:
    """softplus(x - 1) (reference activations.py:33-35)."""
    return torch.nn.functional.softplus(x - 1.0)


class WidenedSigmoid(Module):
    def forward(self, x: Tensor) -> Tensor:
        return widened_sigmoid(x)


class ShiftedSoftplus(Module):
    def forward(self, x: Tensor) -> Tensor:
        return shifted_softplus(x)
