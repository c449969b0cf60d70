"""The radiance-field module (reference nerf/core/model.py).

Same constructor, attributes, submodule layout and state_dict keys as the
reference's `NeRF`; `forward` does not walk the submodules -- it hands the
weights to the fused HIP kernel (encode -> trunk -> density/colour heads)."""
from typing import Tuple

import torch
from torch import Tensor
from torch.nn import Linear, Module, ReLU, Sequential

from .. import ops
from .activations import ShiftedSoftplus, WidenedSigmoid
from .features import FeatureMapping


class NeRF(Module):
    """NeRF(phi_x, phi_d, width=256, depth=8, activ=ReLU, resid=True).

    `depth` trunk layers split evenly in two halves (`fc_1`, `fc_2`); with `resid`
    the second half sees [phi_x | h]; `sigma` = Linear + shifted softplus;
    `features` = Linear; `rgb` = Linear([phi_d | features]) -> activ -> Linear ->
    widened sigmoid (reference model.py:39-55)."""

    def __init__(self, phi_x: FeatureMapping, phi_d: FeatureMapping, width: int = 256, depth: int = 8,
                 activ: Module = ReLU, resid: bool = True) -> None:
        super().__init__()
        self.phi_x, self.phi_d = phi_x, phi_d
        self.width, self.depth, self.activ, self.resid = width, depth, activ, resid

        def block(n_in: int, n_out: int) -> Sequential:
            return Sequential(Linear(n_in, n_out), activ())

        half = depth // 2
        skip = phi_x.o_dim if resid else 0
        self.fc_1 = Sequential(block(phi_x.o_dim, width), *(block(width, width) for _ in range(half - 1)))
        self.fc_2 = Sequential(block(width + skip, width), *(block(width, width) for _ in range(half - 1)))
        self.sigma = Sequential(Linear(width, 1), ShiftedSoftplus())
        self.features = Linear(width, width)
        self.rgb = Sequential(Linear(phi_d.o_dim + width, width // 2), activ(),
                              Linear(width // 2, 3), WidenedSigmoid())

    def forward(self, rx: Tensor, rd: Tensor) -> Tuple[Tensor, Tensor]:
        """rx, rd (..., 3) -> sigma (M,), rgb (..., 3) with M = prod(leading dims)
        (reference model.py:68-78)."""
        return ops.field_forward(self, rx, rd)

    def requires_grad(self, required: bool = True) -> None:
        """Switch every parameter's requires_grad (reference model.py:80-87)."""
        for p in self.parameters():
            p.requires_grad_(required)
