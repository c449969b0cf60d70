"""Headline benchmark: rays/s of coarse64+fine64 rendering of 800x800 frames.

    python bench.py --gpus N --steps K --warmup W [--prec bf16|fp32] [--batch RAYS]

With --gpus N > 1 and no launcher around it, the command starts its N ranks itself: the parent
touches no GPU, spawns N fresh child processes of this same file (rank r on GPU r, rendezvous
on 127.0.0.1) and relays rank 0's JSON line.  Under torchrun (WORLD_SIZE set) it is one rank.

A step renders one 800x800 synthetic-ray frame per GPU (640,000 rays, U[0,1)^3 origins and
un-normalised directions like the reference's bench, random-init NeRF(PE(3,6),PE(3,6),256,4),
BVR(2,6,64,64), perturb off) through `BoundedVolumeRaymarcher.render_volume` -- the fused
`nerfx_render_fwd` entry point, exactly what ships -- with inputs and weights resident in HBM
and outputs left on the device.  For N > 1 each rank renders its own frame (the 40-view orbit is
frame-sharded: no collective in the data path) and rank 0 gathers rgb + depth (16 B/ray) at the
end of the step: `value` = rays of all ranks / max-over-ranks time ("weak" scaling).

Sub-records on the same JSON line (each timed in a region of its own, after `value`):
  * `ray_sharded_frame`: ONE 800x800 frame sharded by rays over the N ranks (ceil(n / N) rays
    each) through `nerf_amd.distributed.render_sharded`, the 16 B/ray gather to rank 0 and the
    frame epilogue on rank 0 included: ms per frame, "strong" scaling;
  * `roofline`: a separate pass of the same step issues the same kernels stage by stage through
    the C ABI with HIP events (on the launch stream) around each: achieved = algorithmic FLOP of
    the fused MLP launches / their summed duration, against the dense MFMA peak;
  * `fp32`: the same step with exact-fp32 MFMA contractions (N = 1 only, 3 steps);
  * `train`: 4,096-ray optimisation steps per GPU (render train=True with perturb, fused MSE,
    backward, one flat gradient all-reduce for N > 1, unscale, Adam), ms per step;
  * `other` (N = 1 only): the reference's other model shapes and its published table's three rows through
    the same path, 3 frames each (see other_leg);
  * `cpu_baseline` (N = 1 only): the CPU oracle renders the reference's own 100x100 case on the
    host cores, executing the colour head in the coarse pass as the reference does.

--backend gloo --dry-run exercises launcher, process group, sharding, gathers and the JSON line
on CPU with a stand-in per-ray function; it measures nothing (`value` is null).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_FULL = 641_024          # algorithmic FLOP per sample, full network as the reference evaluates it (model.py:68-78)
FLOP_DENSITY = 433_664       # algorithmic FLOP per sample, trunk + density head
# The bf16 inference kernel evaluates the colour head with `features` (a Linear without activation)
# folded into rgb.0 when the weights are packed: 2 * 256 * 256 FLOP per sample fewer are EXECUTED for
# the same outputs.  Roofline fractions count the executed network (what the matrix cores did); the
# figures on the reference's count ride along as *_reference_flop.
FLOP_FULL_EXECUTED = {"bf16": 641_024 - 2 * 256 * 256, "fp32": 641_024}
FLOP_TRAIN_PER_RAY = 267_517_952      # coarse 64 x density + fine 128 x (forward + wgrad + dgrad), the reference's network
# ... and as the bf16 training kernels execute it: `features` folded into rgb.0 in the forward, the
# input-gradient chain and the weight gradient (2 * 256 * 256 FLOP per sample fewer in each)
FLOP_TRAIN_PER_RAY_EXECUTED = {"bf16": 267_517_952 - 128 * 3 * 2 * 256 * 256, "fp32": 267_517_952}
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}
TN, TF, NC, NF = 2.0, 6.0, 64, 64
WORKLOAD = "NeRF(PE(3,6),PE(3,6),256,4) random init, BVR(2,6,64c+64f)"


# ---- launcher --------------------------------------------------------------------------------
def launch_ranks(n: int) -> int:
    """Start `n` ranks of this file as child processes and wait for them.  Nothing here touches
    a GPU: every rank initialises its own device in a fresh process."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# ---- legs ------------------------------------------------------------------------------------
def cpu_baseline(model):
    """Config 1 of the benchmark definition: the 100x100 frame the reference's CPU path renders,
    through the oracle with the colour head evaluated in the coarse pass too (renderer.py:91)."""
    import torch
    from oracle import nerf_oracle as O
    params = {k: v.detach().cpu() for k, v in model.named_parameters()}
    f = O.positional_freq_bands(6)
    om = O.Model(params, O.make_encoder("pe", f), O.make_encoder("pe", f), "relu", True)
    g = torch.Generator().manual_seed(0)
    H = W = 100
    ro, rd = torch.rand(H * W, 3, generator=g), torch.rand(H * W, 3, generator=g)
    O.infer(om, om, ro[:1000], rd[:1000], 10, 100, TN, TF, NC, NF, batch_size=16384, as_reference=True)   # warm-up
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        O.infer(om, om, ro, rd, H, W, TN, TF, NC, NF, batch_size=16384, as_reference=True)
        best = min(best, time.perf_counter() - t0)
    return {"value": H * W / best, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{H}x{W} frame ({H * W:,} rays, batch 16,384) of the same workload via oracle.infer with the "
                      f"coarse colour head executed as the reference does, best of 3, {best:.2f} s"}


def staged_pass(step, steps):
    """`steps` steps through the staged path (the kernel sequence of nerfx_render_fwd, one C call and
    one HIP event pair per kernel, on the launch stream), after one staged step that is thrown away:
    the first staged step allocates its per-stage tensors, and an allocation between an event pair
    would be counted as kernel time.  -> {(stage, samples, colour): [ms per launch, ...]}, rays per launch."""
    import nerf_amd.ops as ops
    import torch
    ops.PROFILE = []
    try:
        step()
        torch.cuda.synchronize()
        ops.PROFILE = log = []
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    finally:
        ops.PROFILE = None
    per_class = {}
    for stage, b, n, colour, a, z in log:
        per_class.setdefault((stage, b, n, colour), []).append(a.elapsed_time(z))
    return per_class


def median(xs):
    xs = sorted(xs)
    return 0.5 * (xs[(len(xs) - 1) // 2] + xs[len(xs) // 2])


def roofline_from(per_class, steps, prec, ms_per_step):
    """The fused MLP's achieved FLOP/s from a staged pass: every launch class (coarse density-only /
    fine with colours, full and ragged last batch) enters with the MEDIAN of its launch times, so one
    slow launch cannot move the record.  `stage_ms` = median-based kernel time per step and stage."""
    stage_ms, mlp_flop, ref_flop, mlp_ms, mlp_launches, classes = {}, 0.0, 0.0, 0.0, 0, []
    for (stage, b, n, colour), ms in sorted(per_class.items()):
        per_step = len(ms) / steps
        stage_ms[stage] = stage_ms.get(stage, 0.0) + median(ms) * per_step
        if stage == "mlp":
            flop = b * n * (FLOP_FULL_EXECUTED[prec] if colour else FLOP_DENSITY)
            mlp_flop += flop * per_step
            ref_flop += b * n * (FLOP_FULL if colour else FLOP_DENSITY) * per_step
            mlp_ms += median(ms) * per_step
            mlp_launches += per_step
            classes.append({"rays": b, "samples": n, "colour_head": bool(colour), "launches_per_step": per_step,
                            "ms": {"min": min(ms), "median": median(ms), "max": max(ms)},
                            "tflops": flop / (median(ms) * 1e-3) / 1e12,
                            "frac": flop / (median(ms) * 1e-3) / 1e12 / PEAK_TFLOPS[prec]})
    achieved = mlp_flop / (mlp_ms * 1e-3) / 1e12
    roof = {"bound": "mfma", "kernel": "mlp_fwd16_kernel" if prec == "bf16" else "mlp_fwd_kernel",
            "achieved": achieved, "peak": PEAK_TFLOPS[prec], "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS[prec],
            "launches_per_step": mlp_launches, "avg_launch_ms": mlp_ms / mlp_launches,
            "flop_per_launch": mlp_flop / mlp_launches, "kernel_ms_per_step": mlp_ms,
            "flop_counted": "executed network (colour head with `features` folded into rgb.0 in bf16)",
            "frac_reference_flop": ref_flop / (mlp_ms * 1e-3) / 1e12 / PEAK_TFLOPS[prec],
            "all_stages_ms_per_step": sum(stage_ms.values()),
            "consistent_with_step": sum(stage_ms.values()) <= 1.05 * ms_per_step,
            "classes": classes}
    return roof, stage_ms


def run_roofline(step, prec, steps, ms_per_step, flop_per_step, ref_flop_per_step=None):
    """Staged pass -> roofline record.  The kernels of a step cannot take longer than the step: a pass
    whose summed kernel time exceeds the timed step by more than 5 % is measured again (at most twice)
    and the record says how often and whether the last pass agreed.  `frac_from_step` is the bound that
    holds whatever the staged pass saw: all algorithmic FLOP of a step over the timed step."""
    tries = 0
    while True:
        roof, stage_ms = roofline_from(staged_pass(step, steps), steps, prec, ms_per_step)
        if roof["consistent_with_step"] or tries == 2:
            break
        tries += 1
    roof["remeasured"] = tries
    roof["frac_from_step"] = flop_per_step / (ms_per_step * 1e-3) / 1e12 / PEAK_TFLOPS[prec]
    if ref_flop_per_step is not None:
        roof["frac_from_step_reference_flop"] = ref_flop_per_step / (ms_per_step * 1e-3) / 1e12 / PEAK_TFLOPS[prec]
    roof["measured"] = (f"HIP events on the launch stream around every kernel of {steps} staged steps (one more staged "
                        "step first, discarded), median per launch class")
    return roof, stage_ms


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed counter passes of this same
    command (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in runs of their own, fetch side corrected for
    gfx950 as tools/pmc_summary.py states); None when the file is not in the tree."""
    for name in ("r3_bench_bf16_pmc_hbm.json", "r2_bench_bf16_pmc_hbm.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            for k, v in json.load(open(path))["kernels"].items():
                if kernel_prefix in k:
                    return ((v["fetch_corrected_KB"] + v["WRITE_SIZE_KB"]) * 1024.0,
                            f"profiles/{name}: {k}, fetch_corrected_KB + WRITE_SIZE_KB per launch (counter passes of "
                            "an earlier run of this command, not of this run)")
    return None, "no counter summary in profiles/"


def time_region(fn, steps, fence):
    """Wall clock over `steps` calls between two fences, and a HIP event after every call (recorded,
    never waited for inside the region): -> seconds, [ms of each step on the device]."""
    import torch
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if torch.cuda.is_available() else None
    fence()
    t0 = time.perf_counter()
    if marks:
        marks[0].record()
    for i in range(steps):
        fn()
        if marks:
            marks[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    series = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)] if marks else []
    return dt, series


def spread(series):
    """min / avg / max like the reference's bench table (bench.py:99-109), plus the series itself."""
    return {"min": min(series), "avg": sum(series) / len(series), "max": max(series), "median": median(series),
            "series": [round(x, 3) for x in series]}


def train_leg(args, model, bvr, dev, rank, world, dist, prec, steps, warmup):
    """A step = one optimisation step on 4,096 rays per GPU through nerf_amd.train.batch_step, the body
    of train.step's loop (render train=True with perturb, MSE + PSNR, backward, gradient all-reduce for
    N > 1, the GradScaler's unscale / inf check / step / update, Adam, scheduler; losses stay on the
    device), batches resident in HBM."""
    import torch
    import nerf_amd.ops as ops
    from nerf_amd.core import IndendityScheduler
    from nerf_amd.distributed import broadcast_parameters, seed_ranks
    from nerf_amd.optim import Adam
    from nerf_amd.train import batch_step
    B = 4096
    broadcast_parameters(model)
    seed_ranks(1234)                                         # independent perturb draws per rank
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    C, ro, rd = (torch.rand(B, 3, device=dev, generator=g) for _ in range(3))
    eps = 1e-4 if prec == "bf16" else 1e-8
    optim = (torch.optim.Adam if args.optim == "torch" else Adam)(model.parameters(), lr=5e-4, eps=eps)
    scheduler = IndendityScheduler(optim, 1, 5e-4)
    criterion = torch.nn.MSELoss()
    scaler = torch.amp.GradScaler("cuda", enabled=prec == "bf16")
    log = torch.zeros((warmup + steps, 2), device=dev)
    done = [0]
    ops.set_precision(prec)

    def step():                                              # the body of nerf_amd.train.step's loop
        batch_step(model, bvr, optim, scheduler, criterion, scaler, C, ro, rd, True, log[done[0]])
        done[0] += 1

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    dt, series = time_region(step, steps, fence)
    loss = log[warmup + steps - 1].tolist()[0]
    t_max = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = t_max.item()
    rate = world * B * steps / dt
    achieved = rate / world * FLOP_TRAIN_PER_RAY_EXECUTED[prec] / 1e12
    reference = rate / world * FLOP_TRAIN_PER_RAY / 1e12
    return {"value": rate, "unit": "rays/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "dtype": prec, "rays_per_step_per_gpu": B, "optimizer": args.optim, "step_ms": spread(series),
            "roofline": {"bound": "mfma", "kernel": "whole step", "achieved": achieved, "peak": PEAK_TFLOPS[prec],
                         "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS[prec],
                         "flop_counted": "executed network (`features` folded into rgb.0 in bf16)",
                         "frac_reference_flop": reference / PEAK_TFLOPS[prec]},
            "final_loss": loss}


def flop_per_sample(e_x, e_d, width, depth, resid, density_only=False, executed=False):
    """Algorithmic FLOP (2 per multiply-add, unpadded) of one sample through NeRF.forward (model.py:68-78);
    `executed`: as the bf16 inference kernels evaluate it, i.e. without the `features` layer where they
    fold it into rgb.0 (the 16x16x32 kernel's models: 129..256 wide, skip connection, depth 4 or 8, encoders
    of at most 64 slots; and the whole-ray kernel's narrow models)."""
    hd = depth // 2
    mac = e_x * width + (hd - 1) * width * width + (width + (e_x if resid else 0)) * width + (hd - 1) * width * width + width
    if not density_only:
        folded = executed and ((width > 128 and resid and depth in (4, 8) and e_x <= 64 and e_d <= 64) or
                               (width in (16, 32) and depth == 2 and not resid))
        mac += (0 if folded else width * width) + (e_d + width) * (width // 2) + (width // 2) * 3
    return 2 * mac


def other_leg(args, dev, fence):
    """The reference's other model shapes through the same render path, bf16, 3 timed frames each after
    one warm-up frame (parity cases measured on the side, not the headline): its Fourier-feature variant
    and the distilled student on the 800x800 frame, the three rows of the reference's published table
    (README.md:261-266: 400x400 frames, its CLI's models -- PE(3,10)/PE(3,4), SiLU: Makefile:9-10,
    train.py:212-221) and the headline model at the reference's own batch of 16,384 rays (bench.py:234)."""
    import torch
    import nerf_amd.ops as ops
    from nerf_amd.core import BoundedVolumeRaymarcher, FourierFeatures, NeRF, PositionalEncoding
    PE, FF, SiLU = PositionalEncoding, FourierFeatures, torch.nn.SiLU
    base = lambda: NeRF(PE(3, 6), PE(3, 6), 256, 4)
    student = lambda: NeRF(PE(3, 6), PE(3, 6), 32, 2, resid=False)
    cli_nerf = lambda: NeRF(PE(3, 10), PE(3, 4), 256, 8, SiLU)
    cli_student = lambda: NeRF(PE(3, 10), PE(3, 4), 32, 2, SiLU, resid=False)
    # name -> (coarse, fine | None = same module, frame side, nc, nf, rays per call, (e_x, e_d, width, depth, resid) of coarse / fine)
    cases = {
        "fourier_256x4_800": (lambda: NeRF(FF(3, 256, 32.), FF(3, 64, 16.), 256, 4), None, 800, 64, 64, 65536,
                              (512, 128, 256, 4, True), None),
        "student_32x2_800_64+64": (student, None, 800, 64, 64, 65536, (39, 39, 32, 2, False), None),
        "student_32x2_800_64+128": (student, None, 800, 64, 128, 65536, (39, 39, 32, 2, False), None),
        "readme_nerf+nerf_400": (cli_nerf, None, 400, 64, 64, 16384, (63, 27, 256, 8, True), None),
        "readme_student+nerf_400": (cli_student, cli_nerf, 400, 64, 64, 16384, (63, 27, 32, 2, False), (63, 27, 256, 8, True)),
        "readme_student+student_400": (cli_student, None, 400, 64, 64, 16384, (63, 27, 32, 2, False), None),
        "base_256x4_800_batch16384": (base, None, 800, 64, 64, 16384, (39, 39, 256, 4, True), None),
    }
    out = {}
    ops.set_precision("bf16")
    for name, (make_c, make_f, side, nc, nf, batch, dims_c, dims_f) in cases.items():
        torch.manual_seed(0)
        coarse = make_c().to(dev)
        fine = coarse if make_f is None else make_f().to(dev)
        dims_f = dims_c if dims_f is None else dims_f
        bvr = BoundedVolumeRaymarcher(TN, TF, nc, nf)
        n = side * side
        g = torch.Generator(device=dev).manual_seed(3)
        ro, rd = torch.rand(n, 3, device=dev, generator=g), torch.rand(n, 3, device=dev, generator=g)

        def frame():
            for s in range(0, n, batch):
                bvr.render_volume(coarse, fine, ro[s:s + batch], rd[s:s + batch])

        with torch.no_grad():
            frame()
            dt, series = time_region(frame, 3, fence)
        flop = n * (nc * flop_per_sample(*dims_c, density_only=True) + (nc + nf) * flop_per_sample(*dims_f, executed=True))
        ref = n * (nc * flop_per_sample(*dims_c, density_only=True) + (nc + nf) * flop_per_sample(*dims_f))
        out[name] = {"ms_per_frame": dt / 3 * 1e3, "rays_per_s": 3 * n / dt, "frame": f"{side}x{side}", "samples": f"{nc}+{nf}",
                     "rays_per_call": batch, "tflops": flop / (dt / 3) / 1e12, "mfma_frac": flop / (dt / 3) / 1e12 / PEAK_TFLOPS["bf16"],
                     "mfma_frac_reference_flop": ref / (dt / 3) / 1e12 / PEAK_TFLOPS["bf16"],
                     "step_ms": {k: v for k, v in spread(series).items() if k != "series"}}
        del coarse, fine
    ops.set_precision(None)
    return out


# ---- one rank --------------------------------------------------------------------------------
def dry_run(args, rank, world):
    """Launcher, process group, ray sharding, gathers and the JSON line on CPU tensors, with a
    stand-in per-ray function in place of the kernels.  Measures nothing."""
    import torch
    import torch.distributed as dist
    from nerf_amd.distributed import render_sharded, seed_ranks
    if world > 1:
        dist.init_process_group(args.backend)
    seed_ranks(1234)
    draw = torch.rand(4)                                     # what a perturbed render would draw first
    n_rays = args.size * args.size
    g = torch.Generator().manual_seed(0)
    ro, rd = torch.rand(n_rays, 3, generator=g), torch.rand(n_rays, 3, generator=g)
    fake = lambda o, d: ((o * d).sum(-1), torch.sin(o + 2 * d))
    frame = render_sharded(fake, ro, rd, dst=0)
    ok = True
    if rank == 0:
        depth, rgb = fake(ro, rd)
        ok = torch.equal(frame, torch.cat([rgb, depth[:, None]], dim=1))
    draws = [torch.zeros(4) for _ in range(world)]
    if world > 1:
        dist.all_gather(draws, draw)
        ok = ok and not any(torch.equal(draws[0], d) for d in draws[1:])
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "rays/sec (coarse64+fine64)", "value": None, "unit": "rays/s", "n_gpus": world,
                          "dry_run": True, "backend": args.backend, "sharded_frame_identical": bool(ok)}), flush=True)
    return 0 if ok else 1


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(args.backend, device_id=dev if args.backend == "nccl" else None)

    import nerf_amd.ops as ops
    from nerf_amd.core import BoundedVolumeRaymarcher, NeRF, PositionalEncoding
    from nerf_amd.distributed import render_sharded

    torch.manual_seed(0)
    model = NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 256, 4).to(dev)
    bvr = BoundedVolumeRaymarcher(TN, TF, NC, NF)
    n_rays = args.size * args.size
    g = torch.Generator(device=dev).manual_seed(rank)
    ro = torch.rand(n_rays, 3, device=dev, generator=g)
    rd = torch.rand(n_rays, 3, device=dev, generator=g)
    out = torch.zeros(n_rays, 4, device=dev)                 # rgb | depth, stays on the device
    gathered = [torch.zeros_like(out) for _ in range(world)] if (world > 1 and rank == 0) else None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(dt):
        t_max = torch.tensor([dt], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        return t_max.item()

    if args.only_other:                                      # diagnostic: just the other model shapes
        def fence_():
            torch.cuda.synchronize()
        with torch.no_grad():
            print(json.dumps({"other": other_leg(args, dev, fence_)}), flush=True)
        return 0

    if args.mode == "train":
        rec = train_leg(args, model, bvr, dev, rank, world, dist, args.prec, args.steps, args.warmup)
        if rank == 0:
            print(json.dumps({
                "metric": "rays/sec (training step, coarse64+fine64)", "value": rec["value"], "unit": "rays/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": rec["ms_per_step"],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
                "config": {"workload": "4,096-ray optimisation step per GPU, " + WORKLOAD + ", perturb on, MSELoss, Adam ("
                                       + args.optim + "), GradScaler with bf16", "parallelism": f"data-parallel x{world}"},
                "step_ms": rec["step_ms"],
                "roofline": dict(rec["roofline"], traffic=None), "final_loss": rec["final_loss"]}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 0

    def render_step():
        for s in range(0, n_rays, args.batch):
            e = min(s + args.batch, n_rays)
            *_, D, C = bvr.render_volume(model, model, ro[s:e], rd[s:e])
            out[s:e, :3], out[s:e, 3] = C, D

    def step():
        render_step()
        if world > 1:
            dist.gather(out, gathered, dst=0)

    # the frame every rank shares for the ray-sharded leg (same seed everywhere)
    gs = torch.Generator(device=dev).manual_seed(12345)
    ro_s = torch.rand(n_rays, 3, device=dev, generator=gs)
    rd_s = torch.rand(n_rays, 3, device=dev, generator=gs)

    def render_block(ro_b, rd_b):
        depth = torch.empty((ro_b.shape[0],), device=dev)
        rgb = torch.empty((ro_b.shape[0], 3), device=dev)
        for s in range(0, ro_b.shape[0], args.batch):
            e = min(s + args.batch, ro_b.shape[0])
            *_, D, C = bvr.render_volume(model, model, ro_b[s:e], rd_b[s:e])
            depth[s:e], rgb[s:e] = D, C
        return depth, rgb

    def sharded_step():
        frame = render_sharded(render_block, ro_s, rd_s, dst=0)
        if frame is not None:                                # rank 0: display frames, still on the device
            ops.frame_finish(frame[:, 3], frame[:, :3], as_float=False, as_uint8=True)

    result = None
    with torch.no_grad():
        ops.set_precision(args.prec)
        for _ in range(args.warmup):
            step()
        flop_ref = n_rays * (NC * FLOP_DENSITY + (NC + NF) * FLOP_FULL)
        flop_step = n_rays * (NC * FLOP_DENSITY + (NC + NF) * FLOP_FULL_EXECUTED[args.prec])
        before = None
        if rank == 0 and not args.no_roofline:                # the staged pass once before the timed region ...
            before, _ = roofline_from(staged_pass(render_step, 3), 3, args.prec, float("inf"))
        dt, series = time_region(step, args.steps, fence)
        dt = reduce_max(dt)
        ms_step = dt / args.steps * 1e3
        roof = stage_ms = None
        if rank == 0 and not args.no_roofline:                # ... and right behind it, for the record
            roof, stage_ms = run_roofline(render_step, args.prec, 5, ms_step, flop_step, flop_ref)
            roof["traffic"], roof["traffic_source"] = pmc_traffic("mlp_fwd16_kernel<0, 2, 2>" if args.prec == "bf16" else "mlp_fwd_kernel")
            roof["before_timed_region"] = {k: before[k] for k in ("frac", "avg_launch_ms", "kernel_ms_per_step")}
        for _ in range(max(1, min(args.warmup, 2))):
            sharded_step()
        k_sh = max(3, min(args.steps, 10))
        dt_sh, series_sh = time_region(sharded_step, k_sh, fence)
        dt_sh = reduce_max(dt_sh)
        if rank == 0:
            result = {
                "metric": "rays/sec (coarse64+fine64)", "value": world * n_rays * args.steps / dt, "unit": "rays/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.prec,
                "data": "synthetic",
                "config": {"workload": f"{args.size}x{args.size} frame per GPU, {WORKLOAD}, perturb off, "
                                       f"{args.batch} rays per render_volume call (nerfx_render_fwd)",
                           "rays_per_step_per_gpu": n_rays, "parallelism": f"frame-sharded x{world}"},
                "step_ms": spread(series),
                "roofline": roof, "stage_ms_per_step": stage_ms,
                "ray_sharded_frame": {
                    "metric": f"ms per {args.size}x{args.size} frame", "ms_per_frame": dt_sh / k_sh * 1e3,
                    "rays_per_s": n_rays * k_sh / dt_sh, "n_gpus": world, "steps": k_sh, "scaling": "strong",
                    "rays_per_gpu": -(-n_rays // world), "step_ms": spread(series_sh),
                    "includes": "render of ceil(n/N) rays per rank, 16 B/ray gather to rank 0, frame epilogue on rank 0"},
            }
        if world > 1:
            dist.barrier()
        if world == 1 and not args.no_fp32 and args.prec != "fp32":
            ops.set_precision("fp32")
            render_step()
            dt32, series32 = time_region(render_step, 3, fence)
            roof32 = None
            if not args.no_roofline:
                roof32, _ = run_roofline(render_step, "fp32", 2, dt32 / 3 * 1e3, flop_ref)
                roof32.pop("classes")
            result["fp32"] = {"value": n_rays * 3 / dt32, "unit": "rays/s", "ms_per_step": dt32 / 3 * 1e3, "steps": 3,
                              "warmup": 1, "dtype": "fp32", "step_ms": spread(series32), "roofline": roof32}
        ops.set_precision(args.prec)

    if not args.no_train:
        rec = train_leg(args, model, bvr, dev, rank, world, dist, args.prec, 20, 5)
        if rank == 0:
            result["train"] = rec
    if world == 1 and not args.no_other:
        result["other"] = other_leg(args, dev, fence)
    ops.set_precision(None)
    if rank == 0:
        if world == 1 and not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(model)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prec", choices=("bf16", "fp32"), default="bf16")
    ap.add_argument("--batch", type=int, default=65536, help="rays per render_volume call")
    ap.add_argument("--size", type=int, default=800, help="frame is size x size")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 sub-record")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step sub-record")
    ap.add_argument("--no-roofline", action="store_true", help="skip the staged passes (per-kernel events)")
    ap.add_argument("--no-other", action="store_true", help="skip the other model shapes (sub-record `other`)")
    ap.add_argument("--only-other", action="store_true", help="diagnostic: run only the `other` sub-record")
    ap.add_argument("--optim", choices=("fused", "torch"), default="fused", help="Adam implementation of the training leg")
    ap.add_argument("--mode", choices=("render", "train"), default="render",
                    help="render: the headline metric; train: 4,096-ray optimisation steps as the primary record")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="process-group backend (nccl = RCCL)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / sharding / gather check on CPU; measures nothing")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
