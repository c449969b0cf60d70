"""Diagnostic (GPU box): rays/s of `render_volume` for the other model shapes the reference uses
(its Fourier-feature variant, the README's 256 x 8 SiLU model, the 32 x 2 distilled student), same
protocol as bench.py (800 x 800 frame, 65,536-ray calls, inputs resident).  One JSON line each.

    python tools/bench_configs.py [--size 800]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_amd.ops as ops
from nerf_amd.core import BoundedVolumeRaymarcher, FourierFeatures, NeRF, PositionalEncoding

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=800)
ap.add_argument("--batch", type=int, default=65536)
args = ap.parse_args()
dev = torch.device("cuda", 0)
CONFIGS = {
    "base 256x4 ReLU PE(3,6)": lambda: NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 256, 4),
    "fourier 256x4 ReLU FF(256,32)/FF(64,16)": lambda: NeRF(FourierFeatures(3, 256, 32.), FourierFeatures(3, 64, 16.), 256, 4),
    "readme 256x8 SiLU PE(3,10)/PE(3,4)": lambda: NeRF(PositionalEncoding(3, 10), PositionalEncoding(3, 4), 256, 8, torch.nn.SiLU),
    "student 32x2 ReLU no-resid PE(3,6)": lambda: NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 32, 2, resid=False),
}
n = args.size * args.size
torch.manual_seed(0)
ro, rd = torch.rand(n, 3, device=dev), torch.rand(n, 3, device=dev)
bvr = BoundedVolumeRaymarcher(2.0, 6.0, 64, 64)
for name, make in CONFIGS.items():
    torch.manual_seed(0)
    model = make().to(dev)
    for prec in ("bf16", "fp32"):
        ops.set_precision(prec)
        with torch.no_grad():
            def frame():
                for s in range(0, n, args.batch):
                    bvr.render_volume(model, model, ro[s:s + args.batch], rd[s:s + args.batch])
            frame()
            torch.cuda.synchronize()
            steps = 5 if prec == "bf16" else 2
            t0 = time.perf_counter()
            for _ in range(steps):
                frame()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        print(json.dumps({"config": name, "dtype": prec, "ms_per_frame": dt * 1e3, "rays_per_s": n / dt}), flush=True)
ops.set_precision(None)
