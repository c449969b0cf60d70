"""File-backed stand-in for the one `kornia` symbol the reference imports.

The reference's `nerf/core/ray.py` does `from kornia import create_meshgrid` at
module top and compiles it into a TorchScript function, so the import fails
without a real source file providing that name.  kornia is not installed in
this container and cannot be fetched.  The stand-in is used ONLY by
`tools/make_golden.py` (never by the product, the tests or the GPU box) and
only reaches `pinhole_ray_directions`, which is off the hot path; nothing the
golden vectors pin flows through it.
"""
import torch
from torch import Tensor


def create_meshgrid(height: int, width: int, normalized_coordinates: bool = True) -> Tensor:
    xs = torch.linspace(0, width - 1, width)
    ys = torch.linspace(0, height - 1, height)
    if normalized_coordinates:
        xs = (xs / (width - 1) - 0.5) * 2
        ys = (ys / (height - 1) - 0.5) * 2
    grid = torch.stack(torch.meshgrid([xs, ys], indexing="ij"), dim=-1)  # (W, H, 2)
    return grid.permute(1, 0, 2).unsqueeze(0)  # (1, H, W, 2), [..., 0] = x
