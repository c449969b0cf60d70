"""Frame rendering (reference nerf/infer.py:13-71)."""
import torch
from torch import Tensor

from . import ops
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
    (reference infer.py:58-68), computed by the frame kernel on the device the inputs live on."""
    depth_f, rgb_f, _, _ = ops.frame_finish(depth, rgb, as_float=True, as_uint8=False)
    return depth_f.view(H, W, 3).cpu(), rgb_f.view(H, W, 3).cpu()


@torch.inference_mode()
def orbit(coarse: NeRF, fine: NeRF, raymarcher: BVR, poses: Tensor, H: int, W: int, focal: float,
          batch_size: int = 16_384):
    """Render one frame per pose, as the reference's CLI does (infer.py:110-146), yielding
    (depth_map, rgb_map) uint8 numpy arrays (H, W, 3).  Rays are generated on the device per
    batch from the pose (no per-frame ray arrays, no host-to-device copy) and each frame comes
    back as 8-bit pixels in one copy."""
    device = next(coarse.parameters()).device
    poses = poses.to(device=device, dtype=torch.float32)
    n = H * W
    depth = torch.empty((n,), dtype=torch.float32, device=device)
    rgb = torch.empty((n, 3), dtype=torch.float32, device=device)
    for pose in poses:
        for s in range(0, n, batch_size):
            e = min(s + batch_size, n)
            ro, rd = ops.camera_rays(pose, H, W, focal, s, e - s)
            *_, D, C = raymarcher.render_volume(coarse, fine, ro, rd, train=False)
            depth[s:e], rgb[s:e] = D, C
        _, _, depth_u8, rgb_u8 = ops.frame_finish(depth, rgb, as_float=False, as_uint8=True)
        yield depth_u8.view(H, W, 3).cpu().numpy(), rgb_u8.view(H, W, 3).cpu().numpy()


NeRF.infer = infer
