"""Drives training steps of ragged sizes through the range-checked build (GPU box):

    make -C nerf_amd/csrc checked && NERFX_LIB=nerf_amd/libnerfx_checked.so python tools/run_checked.py

In that build every training-tape address (pre-activation tiles, sign words, gradient tiles, encoder
slots) is checked on the device against the tape's tile count and padded sample count; an access
outside prints the offending tile / sample and traps, so a wrong tile index or a store that outlives
its block shows up as a message naming it instead of a memory fault somewhere else.  Sizes are chosen so
that the last workgroup block is partial, there are fewer blocks than workgroups, and more blocks
than workgroups."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import nerf_amd.ops as ops  # noqa: E402
from nerf_amd import _lib  # noqa: E402
from nerf_amd.core import BoundedVolumeRaymarcher, FourierFeatures, NeRF, PositionalEncoding  # noqa: E402

assert "checked" in _lib.LIB_PATH, "run with NERFX_LIB=nerf_amd/libnerfx_checked.so"
dev = torch.device("cuda", 0)
MODELS = {
    "256x4 ReLU PE(3,6)": lambda: NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 256, 4),
    "256x8 SiLU PE(3,10)/PE(3,4)": lambda: NeRF(PositionalEncoding(3, 10), PositionalEncoding(3, 4), 256, 8, torch.nn.SiLU),
    "64x2 LeakyReLU no skip": lambda: NeRF(PositionalEncoding(3, 4), PositionalEncoding(3, 2), 64, 2, torch.nn.LeakyReLU, resid=False),
    "256x4 Fourier(96,8)/(32,4)": lambda: NeRF(FourierFeatures(3, 96, 8.), FourierFeatures(3, 32, 4.), 256, 4),
}
for name, make in MODELS.items():
    for prec in ("bf16", "fp32"):
        ops.set_precision(prec)
        for rays, nc, nf in ((1, 8, 8), (37, 64, 64), (1500, 64, 64), (4096, 16, 24)):
            torch.manual_seed(rays)
            model = make().to(dev)
            bvr = BoundedVolumeRaymarcher(2.0, 6.0, nc, nf)
            ro, rd = torch.rand(rays, 3, device=dev), torch.rand(rays, 3, device=dev)
            _, _, _, rgb_map = bvr.render_volume(model, model, ro, rd, perturb=True, train=True)
            rgb_map.square().mean().backward()
            torch.cuda.synchronize()
            g = torch.cat([p.grad.flatten() for p in model.parameters()])
            assert torch.isfinite(g).all(), (name, prec, rays)
            print(f"ok  {name:32s} {prec}  rays {rays:5d}  {nc}+{nf}", flush=True)
ops.set_precision(None)
print("all tape accesses inside the tape")
