"""Distillation (reference distill.py) on the HIP path: the fused loss kernel against the oracle's
restatement, and `distill` against the golden run of the reference's own loop.

Tolerances: loss 1e-5 relative, loss gradients 1e-5 relative (1e-9 absolute), student weights
after three Adam steps in fp32 1e-4 relative / 2e-6 absolute.
"""
import pytest
import torch

from conftest import load_golden
from helpers import assert_close, product_model
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    import nerf_amd.ops as ops_
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    ops_.set_precision(None)
    yield ops_
    ops_.set_precision(None)


@pytest.mark.parametrize("n", [1, 48, 257, 1024])
def test_distill_loss_and_gradient(ops, n):
    g = torch.Generator().manual_seed(n)
    t_sigma, s_sigma = torch.rand(n, generator=g) * 5, torch.rand(n, generator=g) * 5
    t_rgb, s_rgb = torch.rand(n, 3, generator=g), torch.rand(n, 3, generator=g)
    delta = 0.002 + torch.rand(n, generator=g) * 0.5
    s_sigma_c, s_rgb_c = s_sigma.clone().requires_grad_(True), s_rgb.clone().requires_grad_(True)
    want = O.distill_loss(s_sigma_c, s_rgb_c, t_sigma, t_rgb, delta)
    (want * 7.0).backward()
    s_sigma_d, s_rgb_d = s_sigma.cuda().requires_grad_(True), s_rgb.cuda().requires_grad_(True)
    got = ops.distill_loss(s_sigma_d, s_rgb_d, t_sigma.cuda(), t_rgb.cuda(), delta.cuda())
    (got * 7.0).backward()
    assert_close(got.detach().cpu(), want.detach(), rtol=1e-5, atol=0, what="loss")
    assert_close(s_sigma_d.grad.cpu(), s_sigma_c.grad, rtol=1e-4, atol=1e-9, what="d loss / d sigma")
    assert_close(s_rgb_d.grad.cpu(), s_rgb_c.grad, rtol=1e-5, atol=1e-9, what="d loss / d rgb")


@pytest.mark.parametrize("fused_adam", [False, True])
def test_distill_golden(ops, fused_adam):
    import nerf_amd.distill as D
    from nerf_amd.core import IndendityScheduler, NeRF
    from nerf_amd.optim import Adam
    meta, g = load_golden("g11_distill")
    teacher = product_model(meta["teacher"], g)
    student = product_model(meta["student"], g)

    class Replay:
        def __init__(self):
            self.k = 0

        def sample(self):
            k = self.k
            self.k += 1
            return g[f"rx{k}"], g[f"rd{k}"], g[f"delta{k}"]          # CPU tensors, like the reference's sampler

    optim = (Adam if fused_adam else torch.optim.Adam)(student.parameters(), lr=meta["lr"], eps=meta["eps"])
    sched = IndendityScheduler(optim, meta["iters"], meta["lr"])
    scaler = torch.amp.GradScaler("cuda", enabled=False)
    ops.set_precision("fp32")
    seen = []
    H = NeRF.distill(teacher, student, Replay(), optim, sched, scaler, iterations=meta["iters"],
                     callbacks=[lambda step, h: seen.append(len(h.train))], verbose=False)
    ops.set_precision(None)
    assert seen == [1, 2, 3] and len(H.lr) == 3 and all(p is None for _, p in H.train)
    assert_close(torch.tensor([l for l, _ in H.train], dtype=torch.float64), g["losses"], rtol=2e-5, atol=0, what="losses")
    for name, p in student.named_parameters():
        assert_close(p.detach().cpu(), g["after." + name], rtol=1e-4, atol=2e-6, what=name)
    assert all(not p.requires_grad or p.grad is None for p in teacher.parameters())


def test_voxel_sampler_on_device(ops):
    import nerf_amd.distill as D
    _, g = load_golden("g11_distill")
    dom = D.Domain(x=(-4, 4), y=(-4, 4), z=(-4, 4), d=(0.00625, 0.00625))
    vs = D.VoxelSampler(dom, voxels=4, samples=8)
    assert len(vs) == 64 and vs.batch_size == 32
    assert torch.equal(vs.rx.cpu(), g["vs_grid"]) and torch.equal(vs.voxel_size.cpu(), g["vs_voxel_size"])
    rx, rd, delta = vs.sample()
    assert rx.is_cuda and rx.shape == (32, 3) and rd.shape == (32, 3) and delta.shape == (32,)
    assert_close(rd.norm(dim=-1).cpu(), torch.ones(32), rtol=1e-6, atol=0, what="unit directions")
    assert torch.equal(delta.cpu(), g["vs_delta"])                  # a degenerate range gives its single value
    # samples stay near their voxel: 8 consecutive rows share one grid point up to the jitter
    spread = (rx.view(4, 8, 3) - rx.view(4, 8, 3).mean(dim=1, keepdim=True)).abs().max()
    assert float(spread) < 6 * 0.5 * float(vs.voxel_size.max())


def test_voxel_sampler_follows_its_uniforms(ops):
    """The sampling kernel against a restatement of distill.py:62-82 on the very uniforms it draws
    (cell choice, Box-Muller jitter of half a cell, normalised directions, lengths), and its
    reproducibility under torch.manual_seed."""
    import math
    import nerf_amd.distill as D
    dom = D.Domain(x=(-2.0, 3.0), y=(0.5, 4.5), z=(-1.0, 1.0), d=(0.01, 0.05))
    V, S = 5, 7
    vs = D.VoxelSampler(dom, voxels=V, samples=S)
    torch.manual_seed(5)
    rx, rd, delta = vs.sample()
    torch.manual_seed(5)
    rng = ops.next_rng(rx.device)
    n = V * S
    u_cell = ops.rng_uniform(rng, 0, V, "cuda").cpu()
    u_jit = ops.rng_uniform(rng, 1, 6 * n, "cuda").cpu().view(n, 3, 2)
    u_dir = ops.rng_uniform(rng, 2, 3 * n, "cuda").cpu().view(n, 3)
    u_len = ops.rng_uniform(rng, 3, n, "cuda").cpu()
    cell = (u_cell * float(V ** 3)).to(torch.int64).clamp_max(V ** 3 - 1)
    centres = vs.rx.cpu()[cell.repeat_interleave(S)]                       # (n, 3): distill.py:73
    normal = torch.sqrt(-2.0 * torch.log(1.0 - u_jit[..., 0])) * torch.cos(2.0 * math.pi * u_jit[..., 1])
    want_rx = centres + normal * (0.5 * vs.voxel_size.cpu()[None, :])       # distill.py:74
    want_rd = u_dir / u_dir.norm(dim=-1, keepdim=True)                      # distill.py:76-77
    want_delta = dom.d[0] + abs(dom.d[1] - dom.d[0]) * u_len               # distill.py:79-80
    assert_close(rx.cpu(), want_rx, rtol=1e-5, atol=1e-5, what="rx")
    assert_close(rd.cpu(), want_rd, rtol=1e-6, atol=1e-7, what="rd")
    assert_close(delta.cpu(), want_delta, rtol=1e-6, atol=0, what="delta")
    torch.manual_seed(5)
    again = vs.sample()
    assert all(torch.equal(a, b) for a, b in zip((rx, rd, delta), again))
    assert not torch.equal(vs.sample()[0], rx)                              # the next call draws afresh
