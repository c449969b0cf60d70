"""Camera paths (reference nerf/data/path.py), computed on the device."""
from typing import Tuple

from torch import Tensor

from .. import ops


def turnaround_poses(theta: Tuple[float, float], phi: Tuple[float, float], psy: Tuple[float, float],
                     radius: Tuple[float, float], samples: int, device=None) -> Tensor:
    """(samples, 4, 4) camera-to-world matrices along a turn-around path: step k uses
    `linspace(start, end, samples + 1)[k]` of every range and
    diag(-1,1,1,1) @ R_theta @ R_phi @ R_psy @ T_z(radius) (reference path.py:78-128)."""
    return ops.orbit_poses(theta, phi, psy, radius, samples, device)


def turnaround(theta: float, phi: float, psy: float, radius: float, device=None) -> Tensor:
    """One pose of the above (reference path.py:78-100)."""
    return ops.orbit_poses((theta, theta), (phi, phi), (psy, psy), (radius, radius), 1, device)[0]
