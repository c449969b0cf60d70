"""Headline benchmark: rays/s of coarse64+fine64 rendering of 800x800 frames.

    python bench.py --gpus N --steps K --warmup W [--prec bf16|fp32] [--batch RAYS]

A step renders one 800x800 synthetic-ray frame per GPU (640,000 rays, U[0,1)^3 origins and
un-normalised directions like the reference's bench, random-init NeRF(PE(3,6),PE(3,6),256,4),
BVR(2,6,64,64), perturb off) through `BoundedVolumeRaymarcher.render_volume`, inputs and
weights resident in HBM, outputs left on the device; for N > 1 each rank renders its own
frame (the 40-view orbit is frame-sharded: no collective) and rank 0 gathers rgb+depth
(16 B/ray) at the end of the step.  Rank 0 prints ONE JSON line.

roofline: every launch of the fused MLP kernel inside the timed region is bracketed by HIP
events on the launch stream; achieved = algorithmic FLOPs of those launches / their summed
duration (so per-launch FLOPs / average launch duration), against the dense MFMA peak.
cpu_baseline: the CPU oracle (oracle/nerf_oracle.py, torch CPU fp32) renders a 200x200
frame of the same workload on this host's cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_FULL = 641_024          # algorithmic FLOP per sample, full network
FLOP_DENSITY = 433_664       # algorithmic FLOP per sample, trunk + density head
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}


def recorded_traffic(prec):
    """HBM bytes per MLP launch from the committed PMC summary of this same command
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH doubled as gfx950 requires), or None."""
    path = os.path.join(ROOT, "profiles", f"r1_bench_{prec}_pmc_hbm.json")
    try:
        with open(path) as f:
            kernels = json.load(f)["kernels"]
        k = next(v for name, v in kernels.items() if "mlp_fwd_kernel" in name)
        return (k["fetch_corrected_KB"] + k["WRITE_SIZE_KB"]) * 1e3
    except (OSError, KeyError, StopIteration, ValueError):
        return None


def cpu_baseline(model, tn, tf, nc, nf):
    from oracle import nerf_oracle as O
    params = {k: v.detach().cpu() for k, v in model.named_parameters()}
    f = O.positional_freq_bands(6)
    om = O.Model(params, O.make_encoder("pe", f), O.make_encoder("pe", f), "relu", True)
    g = torch.Generator().manual_seed(0)
    H = W = 200
    ro, rd = torch.rand(H * W, 3, generator=g), torch.rand(H * W, 3, generator=g)
    O.infer(om, om, ro[:1000], rd[:1000], 10, 100, tn, tf, nc, nf, batch_size=16384)      # warm-up
    best = float("inf")
    for _ in range(2):
        t0 = time.perf_counter()
        O.infer(om, om, ro, rd, H, W, tn, tf, nc, nf, batch_size=16384)
        best = min(best, time.perf_counter() - t0)
    return {"value": H * W / best, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{H}x{W} frame ({H * W:,} rays) of the same workload via oracle.infer, best of 2, "
                      f"{best:.2f} s"}


FLOP_TRAIN_PER_RAY = 267_517_952      # coarse 64 x density + fine 128 x (forward + wgrad + dgrad)


def bench_train(args, model, bvr, dev, rank, world, dist):
    """A step = one optimisation step on 4,096 rays per GPU exactly as nerf_amd.train.step runs it
    (render train=True, perturb on, fused MSE + PSNR, backward, gradient all-reduce for N > 1,
    unscale_, Adam, scaler update; losses logged on the device and fetched once per LOG_EVERY
    steps), batches resident in HBM.  --optim torch swaps nerf_amd.optim.Adam for torch.optim.Adam."""
    import nerf_amd.ops as ops
    from nerf_amd.distributed import average_gradients, broadcast_parameters
    from nerf_amd.optim import Adam
    from nerf_amd.train import LOG_EVERY
    B = 4096
    broadcast_parameters(model)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    C, ro, rd = (torch.rand(B, 3, device=dev, generator=g) for _ in range(3))
    eps = 1e-4 if args.prec == "bf16" else 1e-8
    optim = (torch.optim.Adam if args.optim == "torch" else Adam)(model.parameters(), lr=5e-4, eps=eps)
    scaler = torch.amp.GradScaler("cuda", enabled=args.prec == "bf16")
    log = torch.zeros((args.warmup + args.steps, 2), device=dev)
    done = [0]

    def step():
        k = done[0]
        *_, C_ = bvr.render_volume(model, model, ro, rd, perturb=True, train=True)
        loss = ops.mse_loss(C_, C, log[k])
        scaler.scale(loss).backward()
        average_gradients(model)
        scaler.unscale_(optim)
        scaler.step(optim)
        scaler.update()
        optim.zero_grad(set_to_none=True)
        done[0] = k + 1
        if (k + 1) % LOG_EVERY == 0:
            return log[k - LOG_EVERY + 1:k + 1].tolist()[-1][0]
        return None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    loss = log[args.warmup + args.steps - 1].tolist()[0]          # the end-of-pass fetch, inside the timed region
    fence()
    dt = time.perf_counter() - t0
    t_max = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.destroy_process_group()
    dt = t_max.item()
    if rank == 0:
        rate = world * B * args.steps / dt
        achieved = rate / world * FLOP_TRAIN_PER_RAY / 1e12
        print(json.dumps({
            "metric": "rays/sec (training step, coarse64+fine64)", "value": rate, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
            "config": {"workload": "4,096-ray optimisation step per GPU, NeRF(PE(3,6),PE(3,6),256,4), BVR(2,6,64c+64f), "
                                   "perturb on, MSELoss, Adam (" + args.optim + "), GradScaler with bf16", "parallelism": f"data-parallel x{world}"},
            "roofline": {"bound": "mfma", "kernel": "whole step", "achieved": achieved, "peak": PEAK_TFLOPS[args.prec],
                         "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS[args.prec], "traffic": None},
            "final_loss": loss}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prec", choices=("bf16", "fp32"), default="bf16")
    ap.add_argument("--batch", type=int, default=65536, help="rays per render_volume call")
    ap.add_argument("--size", type=int, default=800, help="frame is size x size")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--optim", choices=("fused", "torch"), default="fused", help="--mode train: Adam implementation")
    ap.add_argument("--mode", choices=("render", "train"), default="render",
                    help="render: the headline metric; train: 4,096-ray optimisation steps (secondary)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    import nerf_amd.ops as ops
    from nerf_amd.core import BoundedVolumeRaymarcher, NeRF, PositionalEncoding

    tn, tf, nc, nf = 2.0, 6.0, 64, 64
    torch.manual_seed(0)
    model = NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 256, 4).to(dev)
    bvr = BoundedVolumeRaymarcher(tn, tf, nc, nf)
    n_rays = args.size * args.size
    g = torch.Generator(device=dev).manual_seed(rank)
    ro = torch.rand(n_rays, 3, device=dev, generator=g)
    rd = torch.rand(n_rays, 3, device=dev, generator=g)
    out = torch.zeros(n_rays, 4, device=dev)                 # rgb | depth, stays on the device
    gathered = [torch.zeros_like(out) for _ in range(world)] if (world > 1 and rank == 0) else None
    ops.set_precision(args.prec)
    if args.mode == "train":
        return bench_train(args, model, bvr, dev, rank, world, dist)

    def step():
        for s in range(0, n_rays, args.batch):
            e = min(s + args.batch, n_rays)
            *_, D, C = bvr.render_volume(model, model, ro[s:e], rd[s:e])
            out[s:e, :3], out[s:e, 3] = C, D
        if world > 1:
            dist.gather(out, gathered, dst=0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        ops.PROFILE = []
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        log, ops.PROFILE = ops.PROFILE, None

    t_max = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = t_max.item()

    if rank == 0:
        stage_ms, mlp_flop, mlp_launches = {}, 0.0, 0
        for stage, b, n, colour, a, z in log:
            stage_ms[stage] = stage_ms.get(stage, 0.0) + a.elapsed_time(z)
            if stage == "mlp":
                mlp_flop += b * n * (FLOP_FULL if colour else FLOP_DENSITY)
                mlp_launches += 1
        achieved = mlp_flop / (stage_ms["mlp"] * 1e-3) / 1e12
        result = {
            "metric": "rays/sec (coarse64+fine64)", "value": world * n_rays * args.steps / dt, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.prec,
            "data": "synthetic",
            "config": {"workload": f"{args.size}x{args.size} frame per GPU, NeRF(PE(3,6),PE(3,6),256,4) random init, "
                                   f"BVR(2,6,64c+64f), perturb off, {args.batch} rays per render_volume call",
                       "rays_per_step_per_gpu": n_rays, "parallelism": f"frame-sharded x{world}"},
            "roofline": {"bound": "mfma", "kernel": "mlp_fwd_kernel", "achieved": achieved,
                         "peak": PEAK_TFLOPS[args.prec], "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS[args.prec],
                         "traffic": recorded_traffic(args.prec), "launches": mlp_launches,
                         "avg_launch_ms": stage_ms["mlp"] / mlp_launches,
                         "flop_per_launch": mlp_flop / mlp_launches},
            "stage_ms_per_step": {k: v / args.steps for k, v in stage_ms.items()},
        }
        if world == 1 and not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(model, tn, tf, nc, nf)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
