"""Frame rendering (reference nerf/infer.py:13-71)."""
import torch
from torch import Tensor

from .core.model import NeRF
from .core.renderer import BoundedVolumeRaymarcher as BVR
from .utils.pbar import tqdm


@torch.inference_mode()
def infer(coarse: NeRF, fine: NeRF, raymarcher: BVR, ro: Tensor, rd: Tensor, H: int, W: int,
          batch_size: int = 32, verbose: bool = True):
    """Render one H x W frame from its rays.

    -> depth_map, rgb_map: float32 CPU tensors (H, W, 3) in [0, 255], depth first
    (reference infer.py:38-68).  Depth is NaN-cleaned and normalised by the
    frame's own min / max (infer.py:60-63); rgb is clipped to [0, 1] (infer.py:66).
    Rays are rendered `batch_size` at a time like the reference, but results stay
    on the device until the frame is complete: one device-to-host copy per frame."""
    device = next(coarse.parameters()).device
    n = ro.size(0)
    ro, rd = ro.to(device), rd.to(device)
    depth = torch.zeros((H * W,), dtype=torch.float32, device=device)
    rgb = torch.zeros((H * W, 3), dtype=torch.float32, device=device)
    for s in tqdm(range(0, n, batch_size), desc="[NeRF] Rendering", disable=not verbose):
        e = min(s + batch_size, n)
        *_, D, C = raymarcher.render_volume(coarse, fine, ro[s:e], rd[s:e], train=False)
        depth[s:e], rgb[s:e] = D, C
    return postprocess(depth, rgb, H, W)


def postprocess(depth: Tensor, rgb: Tensor, H: int, W: int):
    """Per-ray depth (H*W,) and rgb (H*W, 3) -> the two (H, W, 3) CPU frames in [0, 255]
    (reference infer.py:58-68), computed on the device the inputs live on."""
    depth = depth.view(H, W, 1).repeat((1, 1, 3)).nan_to_num()
    depth = depth - depth.min()
    depth = depth / (depth.max() + 1e-10)
    depth = depth.clip(0, 1) * 255
    rgb = rgb.view(H, W, 3).clip(0, 1) * 255
    return depth.cpu(), rgb.cpu()


NeRF.infer = infer
