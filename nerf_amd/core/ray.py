"""Camera rays and ray sampling (the reference's nerf/core/ray.py), backed by the HIP kernels."""
from typing import Tuple

import torch
from torch import Tensor

from .. import ops


def pinhole_ray_directions(H: int, W: int, focal: float, device=None) -> Tensor:
    """Camera-space directions through the pixel grid, (1, H, W, 3) like the reference returns
    (ray.py:10-26): ((i - W/2)/f, -(j - H/2)/f, -1) for column i, row j.  Lives on `device`
    (default: the current ROCm device)."""
    return ops.pinhole_directions(H, W, focal, device)[None]


def phinhole_ray_projection(prd: Tensor, c2w: Tensor) -> Tuple[Tensor, Tensor]:
    """Camera-space directions -> world-space origins and unit directions of the same shape
    (reference ray.py:30-46)."""
    return ops.project_rays(prd, c2w)


def segment_lengths(t: Tensor, rd: Tensor) -> Tensor:
    """delta_i = (t_{i+1} - t_i) |rd|, the last one 1e10 |rd| (reference ray.py:50-66)."""
    return ops.segment_lengths(t, rd)


def uniform_bounded_rays(ro: Tensor, rd: Tensor, tn: float, tf: float, samples: int,
                         perturb: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Stratified samples between tn and tf (reference ray.py:103-135).
    -> rx (B,N,3), rd (B,N,3), t (B,N), delta (B,N)."""
    u = torch.rand((ro.size(0), samples), device=ro.device) if perturb else None
    out = ops.sample_encode(ro, rd, tn, tf, samples, u)
    return out["rx"], rd[:, None, :].expand(-1, samples, -1), out["t"], out["delta"]


def uniform_bounded_z_values(tn: float, tf: float, batch_size: int, samples: int, d: torch.device,
                             perturb: bool) -> Tensor:
    """z values only (reference ray.py:70-99)."""
    zeros = torch.zeros((batch_size, 3), device=d)
    u = torch.rand((batch_size, samples), device=d) if perturb else None
    return ops.sample_encode(zeros, zeros, tn, tf, samples, u, want_rx=False)["t"]


def pdf_rays(ro: Tensor, rd: Tensor, t: Tensor, weights: Tensor, samples: int,
             perturb: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Importance samples from the coarse weights, merged with the coarse samples
    (reference ray.py:192-229).  -> rx (B,N,3), rd (B,N,3), t (B,N), delta (B,N), N = Nc + Nf."""
    u = torch.rand((ro.size(0), samples), device=ro.device) if perturb else None
    t_all = ops.pdf_sample(t, weights, samples, u)
    n = t_all.size(1)
    rx = ro[:, None, :] + rd[:, None, :] * t_all[:, :, None]
    return rx, rd[:, None, :].expand(-1, n, -1), t_all, ops.segment_lengths(t_all, rd)
