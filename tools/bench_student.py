"""Diagnostic (GPU box): frames of the distilled student (32 x 2, PE(3,6), no skip connection) through
render_volume in bf16, for kernel traces and counter passes of the whole-ray kernel.

    python tools/bench_student.py [--frames 5] [--nf 64] [--general]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_amd.ops as ops
from nerf_amd import _lib
from nerf_amd.core import BoundedVolumeRaymarcher, NeRF, PositionalEncoding

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--nf", type=int, default=64)
ap.add_argument("--size", type=int, default=800)
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--general", action="store_true", help="keep the model on the general kernels")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 32, 2, resid=False).to(dev)
bvr = BoundedVolumeRaymarcher(2.0, 6.0, 64, args.nf)
n = args.size * args.size
ro, rd = torch.rand(n, 3, device=dev), torch.rand(n, 3, device=dev)
_lib.lib.nerfx_debug_set_narrow(0 if args.general else 1)
ops.set_precision("bf16")


def frame():
    for s in range(0, n, args.batch):
        bvr.render_volume(model, model, ro[s:s + args.batch], rd[s:s + args.batch])


with torch.no_grad():
    frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.frames):
        frame()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.frames
print(json.dumps({"config": f"student 32x2 64+{args.nf}", "general_kernels": args.general, "ms_per_frame": dt * 1e3,
                  "rays_per_s": n / dt, "samples_per_s": n * (128 + args.nf) / dt}))
