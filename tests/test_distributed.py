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
