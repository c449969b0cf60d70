"""Training loop (reference nerf/train.py:27-180): placeholder until the backward kernels land."""
from .core.model import NeRF


def step(*args, **kwargs):
    raise NotImplementedError("nerf_amd.train.step needs the MLP backward kernels")


def fit(*args, **kwargs):
    raise NotImplementedError("nerf_amd.train.fit needs the MLP backward kernels")


NeRF.fit = fit
