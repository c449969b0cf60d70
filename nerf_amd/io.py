"""Saving and loading radiance fields.

The reference's CLIs keep models as TorchScript files (`jit.save(jit.script(nerf), "*.model.ts")`,
train.py:288-293; read back with `jit.load`, infer.py:106-107).  `nerf_amd.core.NeRF.forward` hands
its weights to the HIP kernels and so cannot be scripted; instead:

  * `save(nerf, path)` / `load(path, device)` keep a model as constructor arguments + state_dict
    in one file that `torch.load(weights_only=True)` can read (nothing in it is executable);
  * `from_torchscript(module_or_path, device)` turns a model the reference saved (a ScriptModule
    with the reference's submodule names) into a `nerf_amd.core.NeRF` holding the same weights.
    A loaded ScriptModule can also be given to `render_volume` / `infer` directly: the kernels
    read parameters and encoder buffers by name (`ops.ModelHandle`).
"""
from typing import Union

import torch
from torch import nn

from .ops import activation_name
from .core.features import FourierFeatures, PositionalEncoding
from .core.model import NeRF

_ACTIVATIONS = {"ReLU": nn.ReLU, "LeakyReLU": nn.LeakyReLU, "SiLU": nn.SiLU}


def _encoder_meta(enc) -> dict:
    bufs = dict(enc.named_buffers())
    if "freq_bands" in bufs:
        return {"kind": "positional", "n": int(bufs["freq_bands"].numel())}
    if "b" in bufs:
        return {"kind": "fourier", "n": int(bufs["b"].shape[0]), "sigma": float(getattr(enc, "sigma", 0.0))}
    raise NotImplementedError("encoder must hold a `freq_bands` or `b` buffer")


def _make_encoder(meta: dict):
    if meta["kind"] == "positional":
        return PositionalEncoding(3, meta["n"])
    return FourierFeatures(3, meta["n"], meta.get("sigma", 0.0))


def describe(module) -> dict:
    """Constructor arguments of a NeRF-shaped module (ours or a scripted reference model), read
    from its parameter names and shapes (reference model.py:39-55)."""
    sd = dict(module.named_parameters())
    half = 0
    while f"fc_1.{half}.0.weight" in sd:
        half += 1
    if half == 0:
        raise ValueError("not a NeRF: no fc_1.0.0.weight")
    width, e_x = sd["fc_1.0.0.weight"].shape
    name = activation_name(module)
    if name not in _ACTIVATIONS:
        raise NotImplementedError(f"hidden activation {name}")
    return {"phi_x": _encoder_meta(module.phi_x), "phi_d": _encoder_meta(module.phi_d), "width": int(width),
            "depth": 2 * half, "activ": name, "resid": bool(sd["fc_2.0.0.weight"].shape[1] == width + e_x)}


def build(meta: dict) -> NeRF:
    return NeRF(_make_encoder(meta["phi_x"]), _make_encoder(meta["phi_d"]), meta["width"], meta["depth"],
                _ACTIVATIONS[meta["activ"]], meta["resid"])


def save(nerf, path: str) -> None:
    """Constructor arguments + state_dict (parameters and encoder buffers), tensors on the CPU."""
    state = {k: v.detach().cpu() for k, v in nerf.state_dict().items()}
    torch.save({"format": "nerf_amd.model/1", "meta": describe(nerf), "state": state}, path)


def load(path: str, device: Union[str, torch.device] = "cuda") -> NeRF:
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(blob, dict) or blob.get("format") != "nerf_amd.model/1":
        raise ValueError(f"{path} is not a nerf_amd model file")
    nerf = build(blob["meta"])
    nerf.load_state_dict(blob["state"])
    return nerf.to(device)


def from_torchscript(module_or_path, device: Union[str, torch.device] = "cuda") -> NeRF:
    """A `nerf_amd.core.NeRF` with the architecture and weights of a scripted reference model."""
    module = torch.jit.load(module_or_path, map_location="cpu") if isinstance(module_or_path, str) else module_or_path
    nerf = build(describe(module))
    nerf.load_state_dict({k: v.detach().cpu() for k, v in module.state_dict().items()})
    return nerf.to(device)
