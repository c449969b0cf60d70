"""Timing experiment (GPU box): a 4,096-ray forward + backward as two halves on two HIP streams, so that
one half's weight-gradient kernel (tape reads) can run beside the other half's forward / backward chain
(tape writes).  Prints ms per 4,096 rays for: one pass, two halves on one stream, two halves on two."""
import json
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
side = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]


def part(s, e):
    *_, C_ = bvr.render_volume(model, model, ro[s:e], rd[s:e], perturb=True, train=True)
    return ops.mse_loss(C_, C[s:e], log) * 1024.0


def one():
    part(0, B).backward()


def halves_serial():
    for k in range(2):
        part(k * B // 2, (k + 1) * B // 2).backward()


def halves_streams():
    main = torch.cuda.current_stream()
    losses = []
    for k in range(2):
        side[k].wait_stream(main)
        with torch.cuda.stream(side[k]):
            losses.append(part(k * B // 2, (k + 1) * B // 2))
    for k in range(2):
        with torch.cuda.stream(side[k]):
            losses[k].backward()
    for k in range(2):
        main.wait_stream(side[k])


for name, fn in (("one pass", one), ("two halves, one stream", halves_serial), ("two halves, two streams", halves_streams), ("one pass", one)):
    for _ in range(3):
        fn()
        model.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
        model.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    print(json.dumps({"mode": name, "ms_per_4096_rays": (time.perf_counter() - t0) / 20 * 1e3}), flush=True)
