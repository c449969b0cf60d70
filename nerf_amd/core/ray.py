"""Ray sampling (the hot-path half of the reference's nerf/core/ray.py), backed by
the HIP sampling kernels.  Camera-ray generation (`pinhole_ray_directions`,
`phinhole_ray_projection`) is not part of this path."""
from typing import Tuple

import torch
from torch import Tensor

from .. import ops


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
