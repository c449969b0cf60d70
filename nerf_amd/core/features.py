"""Feature mappings phi(v): same classes, attributes and buffers as the reference's
nerf/core/features.py; `forward` runs the HIP encoder kernel."""
import torch
from torch import Size, Tensor
from torch.nn import Module

from .. import _lib, ops


def sample_b(size: Size, sigma: float) -> Tensor:
    """b ~ N(0, sigma^2) for Fourier features (reference features.py:8-18)."""
    return torch.randn(size) * sigma


class FeatureMapping(Module):
    """Base class: maps (..., i_dim) to (..., o_dim) (reference features.py:55-79)."""

    def __init__(self, i_dim: int, o_dim: int) -> None:
        super().__init__()
        self.i_dim = i_dim
        self.o_dim = o_dim

    def forward(self, v: Tensor) -> Tensor:
        raise NotImplementedError("Forward pass not implemented yet!")


class PositionalEncoding(FeatureMapping):
    """[v, sin(f_k v), cos(f_k v)]_k with f = 2 ** linspace(1, N_freqs - 1, N_freqs)
    (reference features.py:82-113; note the non-integer exponents and no pi)."""

    def __init__(self, i_dim: int, N_freqs: int = 10) -> None:
        super().__init__(i_dim, 3 + 6 * N_freqs)           # o_dim ignores i_dim, as in features.py:97
        self.N_freqs = N_freqs
        self.register_buffer("freq_bands", 2 ** torch.linspace(1, N_freqs - 1, N_freqs))

    def forward(self, v: Tensor) -> Tensor:
        return ops.encode(v, _lib.ENC_POSITIONAL, self.N_freqs, self.freq_bands)


class FourierFeatures(FeatureMapping):
    """[sin(2 pi v b^T) | cos(2 pi v b^T)], b a fixed Gaussian matrix
    (reference features.py:116-149)."""

    def __init__(self, i_dim: int, features: int = 256, sigma: float = 26.) -> None:
        super().__init__(i_dim, 2 * features)
        self.features = features
        self.sigma = sigma
        self.size = Size((features, i_dim))
        self.register_buffer("b", sample_b(self.size, sigma))

    def forward(self, v: Tensor) -> Tensor:
        return ops.encode(v, _lib.ENC_FOURIER, self.features, self.b)
