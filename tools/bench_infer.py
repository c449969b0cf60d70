"""Diagnostic (GPU box): ms per 800 x 800 frame through the Python callers, host copies included:
`infer` (float frames back to the host, as the reference returns them) and `orbit` (rays generated
on the device, 8-bit frames back).  One JSON line each."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import nerf_amd.infer as I
import nerf_amd.ops as ops
from nerf_amd.core import BoundedVolumeRaymarcher, NeRF, PositionalEncoding

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 256, 4).to(dev)
bvr = BoundedVolumeRaymarcher(2.0, 6.0, 64, 64)
H = W = 800
focal = .5 * W / np.tan(.5 * 0.6194058656692505)
poses = ops.orbit_poses((-np.pi, np.pi), (-np.pi, np.pi), (-np.pi, np.pi), (4., 4.), 8)
ro, rd = ops.camera_rays(poses[1], H, W, focal)
ops.set_precision("bf16")
for batch in (16384, 65536):
    I.infer(model, model, bvr, ro, rd, H, W, batch, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        depth, rgb = I.infer(model, model, bvr, ro, rd, H, W, batch, False)
    dt = (time.perf_counter() - t0) / 5
    print(json.dumps({"caller": "infer", "batch_size": batch, "ms_per_frame": dt * 1e3, "returns": "2 float32 (H,W,3) host frames"}), flush=True)
    frames = I.orbit(model, model, bvr, poses, H, W, focal, batch)
    next(frames)
    t0 = time.perf_counter()
    n = sum(1 for _ in frames)
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"caller": "orbit", "batch_size": batch, "ms_per_frame": dt * 1e3, "returns": "2 uint8 (H,W,3) host frames, rays generated on the device"}), flush=True)
ops.set_precision(None)
