"""Standalone timing of the HBM-bound kernels (K1 sampling+encoding, K3 compositing forward /
backward, K4 hierarchical sampling) against their algorithmic bytes, plus the MLP kernels.

    python tools/bench_kernels.py [--rays 262144] [--iters 20]

Prints one JSON object per kernel: achieved GB/s = algorithmic bytes / mean launch time
(HIP events on the launch stream), and the fraction of the 8 TB/s HBM peak.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import nerf_amd.ops as ops  # noqa: E402
from nerf_amd import _lib  # noqa: E402

HBM_PEAK = 8000.0   # GB/s


def timed(fn, iters):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=262144)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    B, it = args.rays, args.iters
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    ro, rd = torch.rand(B, 3, device=dev, generator=g), torch.rand(B, 3, device=dev, generator=g)
    freq = 2.0 ** torch.linspace(1, 5, 6, device=dev)
    rows = []

    def report(name, seconds, bytes_per_ray, note):
        gbps = B * bytes_per_ray / seconds / 1e9
        rows.append({"kernel": name, "rays": B, "ms": seconds * 1e3, "bytes_per_ray": bytes_per_ray,
                     "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK, "note": note})

    # K1, standalone form: t + phi_x (64 x 39) + phi_d (39) written, ro/rd read
    enc = (_lib.ENC_POSITIONAL, 6, freq)
    s = timed(lambda: ops.sample_encode(ro, rd, 2.0, 6.0, 64, None, enc, enc, want_rx=False), it)
    report("K1 sample_encode (Nc=64, encodings materialised)", s, 24 + 256 + 256 + 64 * 39 * 4 + 39 * 4,
           "test/standalone form; the render path never materialises encodings")
    s = timed(lambda: ops.sample_encode(ro, rd, 2.0, 6.0, 64, None, want_rx=False), it)
    report("K1 sample_encode (Nc=64, t + delta only)", s, 24 + 256 + 256, "the form the render path launches")
    # K3 forward / backward at N = 128
    n = 128
    sigma = torch.rand(B, n, device=dev, generator=g)
    rgb = torch.rand(B, n, 3, device=dev, generator=g)
    t = torch.sort(torch.rand(B, n, device=dev, generator=g) * 4 + 2, dim=-1).values
    s = timed(lambda: ops.composite_fwd(sigma, rgb, t, rd), it)
    report("K3 composite_fwd (N=128, colour + depth)", s, 512 + 1536 + 512 + 12 + 12 + 4 + 512, "")
    s = timed(lambda: ops.composite_fwd(sigma, None, t, rd, want_depth=False), it)
    report("K3 composite_fwd (N=128, weights only)", s, 512 + 512 + 12 + 512, "no colour rows: 16-byte slices only")
    sigma64, t64 = sigma[:, :64].contiguous(), t[:, :64].contiguous()
    s = timed(lambda: ops.composite_fwd(sigma64, None, t64, rd, want_depth=False), it)
    report("K3 composite_fwd (N=64, weights only)", s, 256 + 256 + 12 + 256, "coarse pass")
    g_map = torch.rand(B, 3, device=dev, generator=g)
    s = timed(lambda: ops.composite_bwd(sigma, rgb, t, rd, None, g_map, None, None), it)
    report("K3b composite_bwd (N=128)", s, 512 + 1536 + 512 + 12 + 12 + 512 + 1536, "weights recomputed, not read")
    # K4
    t_c = torch.linspace(2, 6, 64, device=dev).expand(B, 64).contiguous()
    w_c = torch.rand(B, 64, device=dev, generator=g)
    s = timed(lambda: ops.pdf_sample(t_c, w_c, 64), it)
    report("K4 pdf_sample (64 + 64)", s, 256 + 256 + 512, "production form: sorted t only")
    s = timed(lambda: ops.coarse_pdf_sample(sigma64, t_c, rd, 64), it)
    report("K3 coarse + K4 in one kernel (64 + 64)", s, 256 + 256 + 12 + 512, "densities in, sorted t out; the weights never reach memory")
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
