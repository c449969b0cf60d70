"""world_size-2 `gloo` tests of the multi-GPU host logic (ray sharding + gather, data-parallel
gradient averaging, parameter broadcast) on CPU tensors."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_amd import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run(fn, world=2):
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), fn, out), nprocs=world, join=True)
        return dict(out)


def test_shard_bounds_partition():
    for n in (0, 1, 7, 8, 9, 640000):
        for ws in (1, 2, 3, 8):
            blocks = [D.shard_bounds(n, r, ws) for r in range(ws)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert max(e - s for s, e in blocks) == -(-n // ws) or n == 0


def _fake_render(ro, rd):
    # any per-ray function will do: what is under test is the sharding and the gather
    return (ro * rd).sum(-1), torch.sin(ro + 2 * rd)


def _render_job(rank, world):
    g = torch.Generator().manual_seed(0)
    ro, rd = torch.rand(1001, 3, generator=g), torch.rand(1001, 3, generator=g)
    root = D.render_sharded(_fake_render, ro, rd, dst=0)
    everyone = D.render_sharded(_fake_render, ro, rd, dst=None)
    return (None if root is None else root.clone()), everyone.clone()


def test_render_sharded_matches_single_process():
    g = torch.Generator().manual_seed(0)
    ro, rd = torch.rand(1001, 3, generator=g), torch.rand(1001, 3, generator=g)
    depth, rgb = _fake_render(ro, rd)
    want = torch.cat([rgb, depth[:, None]], dim=1)
    out = run(_render_job)
    assert out[1][0] is None                      # only the destination holds the gathered frame
    assert torch.equal(out[0][0], want)           # bit-identical to the unsharded render
    assert torch.equal(out[0][1], want) and torch.equal(out[1][1], want)


def _grad_job(rank, world):
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    if rank == 1:                                  # start from different weights: broadcast must fix it
        for p in net.parameters():
            p.data.add_(1.0)
    D.broadcast_parameters(net, src=0)
    g = torch.Generator().manual_seed(100)
    x, y = torch.rand(8, 5, generator=g), torch.rand(8, 3, generator=g)
    s, e = D.shard_bounds(8, rank, world)
    torch.nn.functional.mse_loss(net(x[s:e]), y[s:e]).backward()
    D.average_gradients(net)
    return [p.grad.clone() for p in net.parameters()], [p.detach().clone() for p in net.parameters()]


def _flat_grad_job(rank, world):
    """As _grad_job, but the gradients are views of ONE buffer, as the HIP backward returns them:
    the all-reduce must then run in place on that buffer."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    g = torch.Generator().manual_seed(100)
    x, y = torch.rand(8, 5, generator=g), torch.rand(8, 3, generator=g)
    s, e = D.shard_bounds(8, rank, world)
    torch.nn.functional.mse_loss(net(x[s:e]), y[s:e]).backward()
    params = list(net.parameters())
    sizes = [p.numel() for p in params]
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    for p, v in zip(params, flat.split(sizes)):
        p.grad = v.view(p.shape)
    assert D.flat_gradient(params) is not None
    D.average_gradients(net)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, flat.split(sizes))), "in place"
    return flat.clone()


def test_gradient_average_equals_full_batch():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    g = torch.Generator().manual_seed(100)
    x, y = torch.rand(8, 5, generator=g), torch.rand(8, 3, generator=g)
    torch.nn.functional.mse_loss(net(x), y).backward()          # mean over the whole batch
    out = run(_grad_job)
    for r in (0, 1):
        grads, params = out[r]
        for got, p in zip(params, net.parameters()):
            assert torch.equal(got, p.detach())
        for got, p in zip(grads, net.parameters()):
            assert torch.allclose(got, p.grad, rtol=1e-6, atol=1e-8)
    for a, b in zip(out[0][0], out[1][0]):
        assert torch.equal(a, b)                   # identical on both ranks (the scaler's inf check relies on it)


def test_flat_gradient_average_in_place():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    g = torch.Generator().manual_seed(100)
    x, y = torch.rand(8, 5, generator=g), torch.rand(8, 3, generator=g)
    torch.nn.functional.mse_loss(net(x), y).backward()
    want = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    out = run(_flat_grad_job)
    assert torch.equal(out[0], out[1])
    assert torch.allclose(out[0], want, rtol=1e-6, atol=1e-8)


def test_single_process_is_a_no_op():
    net = torch.nn.Linear(3, 2)
    net(torch.ones(1, 3)).sum().backward()
    before = [p.grad.clone() for p in net.parameters()]
    D.average_gradients(net)
    assert all(torch.equal(a, p.grad) for a, p in zip(before, net.parameters()))
    assert D.world() == (0, 1)
    x = torch.arange(12.0).view(4, 3)
    assert torch.equal(D.gather_rays(x, 4), x)


def test_flat_gradient_detection():
    """Consecutive views of one buffer are recognised (and shared, not copied); anything else is not."""
    from nerf_amd.distributed import flat_gradient
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((4, 3), (4,), (2, 4))]
    sizes = [p.numel() for p in params]
    flat = torch.arange(float(sum(sizes)))
    for p, v in zip(params, flat.split(sizes)):
        p.grad = v.view(p.shape)
    got = flat_gradient(params)
    assert got is not None and got.data_ptr() == flat.data_ptr() and torch.equal(got, flat)
    got.mul_(2.0)
    assert torch.equal(params[2].grad, (flat.split(sizes)[2]).view(2, 4))          # shared storage
    params[1].grad = torch.zeros(4)                                                # a separate allocation
    assert flat_gradient(params) is None
    params[1].grad = None
    assert flat_gradient(params) is None
    assert flat_gradient([]) is None


# ---- the real callers at world_size 2 -------------------------------------------------------
# `nerf_amd.train.step` and `distributed.infer_sharded` are run as they ship; only the renderer
# behind `raymarcher.render_volume` is a torch stand-in (the kernels need a GPU), so what is under
# test is the control flow around it: sharding, gather, the all-reduce placed before `unscale_`,
# the scaler's rank-consistent inf handling, per-rank seeding.
class _Field(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))

    def forward(self, ro, rd):
        return self.net(torch.cat([ro, rd], dim=-1))


class _Pointwise(torch.nn.Module):
    """Per-ray arithmetic without a contraction: the same bits whatever the batch size."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([0.3, -1.1, 2.0]))

    def forward(self, ro, rd):
        return ro * self.w


class _StubMarcher:
    """render_volume's signature and return tuple; colours come from the module, a uniform draw
    stands for the perturbation, and `poison_call` injects an inf into that call's output."""

    def __init__(self, poison_call=None, squash=True):
        self.calls, self.poison_call, self.draws, self.squash = 0, poison_call, [], squash

    def render_volume(self, coarse, fine, ro, rd, perturb=False, train=False):
        u = torch.rand(1) if perturb else torch.zeros(1)
        self.draws.append(float(u))
        rgb = fine(ro, rd)
        if self.squash:
            rgb = torch.sigmoid(rgb)
        if self.calls == self.poison_call:
            rgb = rgb * float("inf")
        self.calls += 1
        depth = (ro * rd).sum(-1)
        return None, None, depth, rgb


class _PlainMSE(torch.nn.MSELoss):
    """A subclass, so that `step` takes the caller's criterion instead of the fused loss kernel."""


def _train_job(rank, world):
    from nerf_amd import train as T
    from nerf_amd.core import IndendityScheduler
    torch.manual_seed(7)
    net = _Field()
    if rank == 1:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.5)
    D.broadcast_parameters(net, src=0)
    D.seed_ranks(100)
    optim = torch.optim.Adam(net.parameters(), lr=1e-2)
    sched = IndendityScheduler(optim, 1, 1e-2)
    scaler = torch.amp.GradScaler("cpu", enabled=True, init_scale=256.0, growth_interval=1000)
    g = torch.Generator().manual_seed(50 + rank)                     # every rank has its own batches
    batches = [tuple(torch.rand(32, 3, generator=g) for _ in range(3)) for _ in range(4)]
    marcher = _StubMarcher(poison_call=2 if rank == 1 else None)     # rank 1 overflows in its third step
    before_poison = None
    snaps = []

    class Loader(list):
        pass

    def snapshot_hook(_self, closure=None, _orig=optim.step):
        snaps.append(torch.cat([p.detach().reshape(-1).clone() for p in net.parameters()]))
        return _orig(closure)

    optim.step = snapshot_hook.__get__(optim)
    losses, psnrs, lrs = T.step(0, net, marcher, optim, sched, _PlainMSE(), scaler, Loader(batches), torch.device("cpu"),
                                split="train", perturb=True, verbose=False)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    return dict(flat=flat, scale=float(scaler.get_scale()), losses=losses, lrs=lrs, optimizer_calls=len(snaps),
                draws=marcher.draws, grads_none=all(p.grad is None for p in net.parameters()))


def test_train_step_data_parallel_with_overflow_on_one_rank():
    out = run(_train_job)
    a, b = out[0], out[1]
    assert torch.equal(a["flat"], b["flat"]), "ranks diverged"
    assert torch.isfinite(a["flat"]).all()
    # the overflow of rank 1 reached rank 0 through the all-reduce that precedes unscale_:
    # both scalers backed off once (256 -> 128) and both skipped that step's update
    assert a["scale"] == b["scale"] == 128.0
    assert a["optimizer_calls"] == b["optimizer_calls"] == 3          # 4 steps, one skipped on both ranks
    assert len(a["losses"]) == 4 and len(a["lrs"]) == 4
    assert a["grads_none"] and b["grads_none"]
    assert all(x != y for x, y in zip(a["draws"], b["draws"])), "ranks drew identical perturbations"


def _train_reference_job(rank, world):
    """Two ranks on the two halves of every batch == one process on the whole batches."""
    from nerf_amd import train as T
    from nerf_amd.core import IndendityScheduler
    torch.manual_seed(7)
    net = _Field()
    optim = torch.optim.SGD(net.parameters(), lr=0.1)
    sched = IndendityScheduler(optim, 1, 0.1)
    scaler = torch.amp.GradScaler("cpu", enabled=False)
    g = torch.Generator().manual_seed(50)
    whole = [tuple(torch.rand(32, 3, generator=g) for _ in range(3)) for _ in range(3)]
    s, e = D.shard_bounds(32, rank, world)
    mine = [tuple(x[s:e] for x in batch) for batch in whole]
    T.step(0, net, _StubMarcher(), optim, sched, _PlainMSE(), scaler, mine, torch.device("cpu"), split="train",
           verbose=False)
    return torch.cat([p.detach().reshape(-1) for p in net.parameters()])


def test_train_step_data_parallel_equals_whole_batch():
    from nerf_amd import train as T
    from nerf_amd.core import IndendityScheduler
    torch.manual_seed(7)
    net = _Field()
    optim = torch.optim.SGD(net.parameters(), lr=0.1)
    sched = IndendityScheduler(optim, 1, 0.1)
    g = torch.Generator().manual_seed(50)
    whole = [tuple(torch.rand(32, 3, generator=g) for _ in range(3)) for _ in range(3)]
    T.step(0, net, _StubMarcher(), optim, sched, _PlainMSE(), torch.amp.GradScaler("cpu", enabled=False), whole,
           torch.device("cpu"), split="train", verbose=False)
    want = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    out = run(_train_reference_job)
    assert torch.equal(out[0], out[1])
    assert torch.allclose(out[0], want, rtol=1e-5, atol=1e-7)


def _infer_job(rank, world):
    from oracle import nerf_oracle as O
    net = _Pointwise()
    g = torch.Generator().manual_seed(9)
    H, W = 7, 9
    ro, rd = torch.rand(H * W, 3, generator=g), torch.rand(H * W, 3, generator=g)
    marcher = _StubMarcher(squash=False)                     # single roundings only: batch-size invariant bits
    depth, rgb = D.infer_sharded(net, net, marcher, ro, rd, H, W, batch_size=5, finish=O.postprocess_frame)
    return (None if depth is None else depth.clone()), (None if rgb is None else rgb.clone()), marcher.calls


def test_infer_sharded_matches_single_process():
    from oracle import nerf_oracle as O
    net = _Pointwise()
    g = torch.Generator().manual_seed(9)
    H, W = 7, 9
    ro, rd = torch.rand(H * W, 3, generator=g), torch.rand(H * W, 3, generator=g)
    with torch.no_grad():
        *_, depth, rgb = _StubMarcher(squash=False).render_volume(net, net, ro, rd)
    want_d, want_c = O.postprocess_frame(depth, rgb, H, W)
    out = run(_infer_job)
    assert out[1][0] is None and out[1][1] is None           # only rank 0 holds the frame
    assert torch.equal(out[0][0], want_d) and torch.equal(out[0][1], want_c)   # incl. the frame-global depth range
    assert out[0][2] == 7 and out[1][2] == 7                 # 32 and 31 rays in batches of 5


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2 --backend gloo --dry-run`: the parent starts two ranks itself."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run",
                          "--size", "50"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["dry_run"] and line["sharded_frame_identical"]
