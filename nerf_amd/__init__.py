"""nerf_amd: the NeRF ray-march -> encode -> MLP -> composite path on MI355X (gfx950).

Same Python surface as the reference's `nerf` package for that path
(`nerf_amd.core`, `nerf_amd.infer`, `nerf_amd.train`); underneath, torch tensors
are handed to hand-written HIP kernels through the C ABI of `libnerfx.so`.
"""
import sys as _sys

from . import _lib          # fails loudly when libnerfx.so is missing
from .ops import set_precision  # noqa: F401

__version__ = "0.1.0"


def install_as(name: str = "nerf") -> None:
    """Alias this package (and its submodules) under another top-level name, so
    that code written as `from nerf.core import ...` picks up the HIP path."""
    from . import core, data, distill, infer, train, utils  # noqa: F401
    from .data import path as _data_path, utils as _data_utils  # noqa: F401
    prefix = __name__ + "."
    for mod_name, mod in list(_sys.modules.items()):
        if mod_name == __name__ or mod_name.startswith(prefix):
            _sys.modules[name + mod_name[len(__name__):]] = mod
