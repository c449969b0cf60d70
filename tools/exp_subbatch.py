"""Timing experiment (GPU box): does a training step in sub-batches whose tape fits the 256 MB
Infinity Cache beat one 4,096-ray pass?  Prints, per sub-batch size, the time of forward + backward
over 4,096 rays (gradients accumulate in .grad across the sub-batches)."""
import json
import sys
import time

import torch

import nerf_amd.ops as ops
from nerf_amd.core import BoundedVolumeRaymarcher, NeRF, PositionalEncoding

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 256, 4).to(dev)
bvr = BoundedVolumeRaymarcher(2.0, 6.0, 64, 64)
B = 4096
C, ro, rd = (torch.rand(B, 3, device=dev) for _ in range(3))
ops.set_precision("bf16")
log = torch.zeros(2, device=dev)


def step(sub):
    for s in range(0, B, sub):
        *_, C_ = bvr.render_volume(model, model, ro[s:s + sub], rd[s:s + sub], perturb=True, train=True)
        loss = ops.mse_loss(C_, C[s:s + sub], log)
        (loss * 1024.0).backward()


for sub in ([int(a) for a in sys.argv[1:]] or (4096, 2048, 1024, 512, 256, 4096)):
    for _ in range(3):
        step(sub)
        model.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step(sub)
        model.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    print(json.dumps({"sub_batch": sub, "ms_per_4096_rays": (time.perf_counter() - t0) / 10 * 1e3}), flush=True)
