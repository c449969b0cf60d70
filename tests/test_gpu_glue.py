"""The kernels around the render in a training step (batch gather, fused MSE / PSNR / gradient
seed, flat Adam with the loss scaler's inf check) against torch's own implementations, which
are what the reference's step uses (train.py:72-99).  Through the C ABI.

Tolerances: gather exact; loss and PSNR 1e-6 relative (a different summation order); Adam
1e-6 absolute / 1e-5 relative per step against torch.optim.Adam on the same gradients.
"""
import pytest
import torch

from conftest import load_golden
from helpers import assert_close, product_model

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    import nerf_amd.ops as ops_
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    ops_.set_precision(None)
    yield ops_
    ops_.set_precision(None)


def test_gather_rays(ops):
    g = torch.Generator().manual_seed(3)
    N = 1000
    C, ro, rd = (torch.rand(N, 3, generator=g).cuda() for _ in range(3))
    perm = torch.randperm(N, generator=g).cuda()
    for begin, n in ((0, 256), (900, 100), (17, 1), (0, 0)):
        a, b, c = ops.gather_rays(C, ro, rd, perm, begin, n)
        idx = perm[begin:begin + n]
        assert torch.equal(a, C[idx]) and torch.equal(b, ro[idx]) and torch.equal(c, rd[idx])
        a, b, c = ops.gather_rays(C, ro, rd, None, begin, n)
        assert torch.equal(a, C[begin:begin + n]) and torch.equal(c, rd[begin:begin + n])
    with pytest.raises(RuntimeError):
        ops.gather_rays(C, ro, rd, None, 990, 20)


@pytest.mark.parametrize("n", [1, 5, 4096, 100_003])
def test_mse_loss_psnr_and_gradient(ops, n):
    g = torch.Generator().manual_seed(n)
    pred = torch.rand(n, 3, generator=g).cuda().requires_grad_(True)
    target = torch.rand(n, 3, generator=g).cuda()
    log = torch.zeros(2, device="cuda")
    loss = ops.mse_loss(pred, target, log)
    (loss * 3.0).backward()
    ref_pred = pred.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.mse_loss(ref_pred, target)
    (ref * 3.0).backward()
    assert_close(loss.detach().cpu(), ref.detach().cpu(), rtol=1e-6, atol=0, what="loss")
    assert_close(log.cpu(), torch.stack((ref.detach(), -10 * torch.log10(ref.detach()))).cpu(), rtol=2e-6, atol=1e-6,
                 what="loss / psnr log")
    assert_close(pred.grad.cpu(), ref_pred.grad.cpu(), rtol=1e-6, atol=1e-12, what="gradient seed")
    # no-grad form and run-to-run reproducibility
    with torch.no_grad():
        again = ops.mse_loss(pred, target)
    assert torch.equal(again, loss.detach())


def _param_sets(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(256, 39), (256,), (128, 295), (3, 128), (3,)]
    a = [torch.nn.Parameter((torch.rand(s, generator=g) - 0.5).cuda()) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    return a, b, g


@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
def test_flat_adam_matches_torch_adam(ops, weight_decay):
    from nerf_amd.optim import Adam
    mine, theirs, g = _param_sets(11)
    opt_a = Adam(mine, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
    opt_b = torch.optim.Adam(theirs, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
    base = mine[0].data_ptr()
    assert all(p.data_ptr() == base + 4 * sum(q.numel() for q in mine[:i]) for i, p in enumerate(mine)), "flat views"
    for it in range(6):
        for p, q in zip(mine, theirs):
            grad = (torch.rand(p.shape, generator=g) - 0.5).cuda() * (10.0 ** (it - 3))
            p.grad, q.grad = grad.clone(), grad.clone()
        v0 = mine[0]._version
        opt_a.step()
        opt_b.step()
        assert mine[0]._version > v0, "the update must be visible to version-based caches"
        for k, (p, q) in enumerate(zip(mine, theirs)):
            assert_close(p.detach().cpu(), q.detach().cpu(), rtol=1e-5, atol=1e-6, what=f"step {it} tensor {k}")
    assert opt_a.steps_applied == 6


def test_flat_adam_survives_replaced_parameter_storage(ops):
    """`load_state_dict(assign=True)` / `p.data = ...` after construction: the optimizer adopts the
    new values instead of updating a stale copy."""
    from nerf_amd.optim import Adam
    mine, theirs, g = _param_sets(13)
    opt_a, opt_b = Adam(mine, lr=1e-3), torch.optim.Adam(theirs, lr=1e-3)
    for it in range(3):
        if it == 1:
            for p, q in zip(mine, theirs):
                fresh = (torch.rand(p.shape, generator=g) - 0.5).cuda()
                p.data = fresh.clone()                   # no longer a view of the flat buffer
                q.data.copy_(fresh)
        for p, q in zip(mine, theirs):
            grad = (torch.rand(p.shape, generator=g) - 0.5).cuda()
            p.grad, q.grad = grad.clone(), grad.clone()
        opt_a.step()
        opt_b.step()
        for k, (p, q) in enumerate(zip(mine, theirs)):
            assert_close(p.detach().cpu(), q.detach().cpu(), rtol=1e-5, atol=1e-6, what=f"step {it} tensor {k}")


def test_flat_adam_behind_grad_scaler_skips_on_inf(ops):
    from nerf_amd.optim import Adam
    mine, theirs, g = _param_sets(12)
    opt_a, opt_b = Adam(mine, lr=1e-3), torch.optim.Adam(theirs, lr=1e-3)
    sc_a = torch.amp.GradScaler("cuda", init_scale=1024.0)
    sc_b = torch.amp.GradScaler("cuda", init_scale=1024.0)
    for it in range(4):
        for p, q in zip(mine, theirs):
            grad = (torch.rand(p.shape, generator=g) - 0.5).cuda() * 1024.0      # "scaled" gradients
            if it == 1:
                grad.view(-1)[0] = float("inf")
            p.grad, q.grad = grad.clone(), grad.clone()
        for sc, opt in ((sc_a, opt_a), (sc_b, opt_b)):
            sc._lazy_init_scale_growth_tracker(torch.device("cuda")) if sc._scale is None else None
            if it % 2 == 0:
                sc.unscale_(opt)                     # the reference's order (train.py:82); else let step() unscale
            sc.step(opt)
            sc.update()
        for k, (p, q) in enumerate(zip(mine, theirs)):
            assert_close(p.detach().cpu(), q.detach().cpu(), rtol=1e-5, atol=1e-6, what=f"step {it} tensor {k}")
    assert opt_a.steps_applied == 3, "the step with an inf gradient must be skipped on the device"
    assert sc_a.get_scale() == sc_b.get_scale() == 512.0


def test_train_step_with_fused_glue_matches_torch_glue(ops):
    """nerf_amd.train.step with (DeviceRayLoader, fused MSE, flat Adam) == the same step with
    (DataLoader, torch Adam): same weights after an epoch of 3 steps, same returned lists."""
    import nerf_amd.train as T
    from nerf_amd.core import BoundedVolumeRaymarcher, IndendityScheduler
    from nerf_amd.data.utils import DeviceRayLoader, loaders
    from nerf_amd.distributed import flat_gradient
    from nerf_amd.optim import Adam
    meta, gold = load_golden("g7_train_student")
    g = torch.Generator().manual_seed(21)

    class Rays(torch.utils.data.Dataset):
        def __init__(self):
            self.C, self.ro, self.rd = (torch.rand(96, 3, generator=g) for _ in range(3))

        def __len__(self):
            return self.C.size(0)

        def __getitem__(self, i):
            return self.C[i], self.ro[i], self.rd[i]

    data = Rays()
    bvr = BoundedVolumeRaymarcher(2.0, 6.0, 16, 16)
    ops.set_precision("fp32")
    results = []
    for fused in (True, False):
        model = product_model(meta["model"], gold)
        optim = (Adam if fused else torch.optim.Adam)(model.parameters(), lr=5e-4, eps=1e-8)
        sched = IndendityScheduler(optim, 1, 5e-4)
        scaler = torch.amp.GradScaler("cuda", enabled=False)
        _, val, _ = loaders(None, data, None, 32, 0, device="cuda" if fused else None)    # unshuffled: same batches
        assert isinstance(val, DeviceRayLoader) == fused and len(val) == 3
        mse, psnr, lr = T.step(0, model, bvr, optim, sched, torch.nn.MSELoss(), scaler, val, torch.device("cuda"),
                               split="train", perturb=False, verbose=False)
        assert len(mse) == len(psnr) == len(lr) == 3
        results.append((mse, psnr, [p.detach().cpu() for p in model.parameters()]))
        if fused:                                        # one more backward: gradients come as one flat buffer
            *_, C_ = bvr.render_volume(model, model, data.ro.cuda(), data.rd.cuda(), perturb=False, train=True)
            ops.mse_loss(C_, data.C.cuda()).backward()
            flat = flat_gradient(list(model.parameters()))
            assert flat is not None and flat.numel() == sum(p.numel() for p in model.parameters())
    ops.set_precision(None)
    (mse_a, psnr_a, w_a), (mse_b, psnr_b, w_b) = results
    assert_close(torch.tensor(mse_a), torch.tensor(mse_b), rtol=1e-5, atol=0, what="losses")
    assert_close(torch.tensor(psnr_a), torch.tensor(psnr_b), rtol=1e-5, atol=1e-5, what="psnrs")
    for k, (a, b) in enumerate(zip(w_a, w_b)):
        assert_close(a, b, rtol=1e-4, atol=2e-6, what=f"weights after 3 steps, tensor {k}")


def test_fused_grad_scaler_step_matches_torch_grad_scaler(ops):
    """nerf_amd.train.step behind an enabled GradScaler, with the flat Adam (the scaler's scale /
    unscale_ / inf check / step run inside the step's own kernels, nerf_amd.amp) against the same pass
    with torch's Adam (every scaler call made by torch): five steps of which the third overflows (an
    infinite target colour).  Same losses, same weights, the overflowing step skipped by both, and the
    same loss scale afterwards (halved once)."""
    import nerf_amd.train as T
    from nerf_amd.core import BoundedVolumeRaymarcher, IndendityScheduler
    from nerf_amd.optim import Adam
    meta, gold = load_golden("g7_train_student")
    g = torch.Generator().manual_seed(23)
    batches = [tuple(torch.rand(64, 3, generator=g) for _ in range(3)) for _ in range(5)]
    batches[2][0][7, 1] = float("inf")
    bvr = BoundedVolumeRaymarcher(2.0, 6.0, 16, 16)
    ops.set_precision("fp32")                               # same arithmetic on both sides; autocast only picks bf16
    results = []
    for fused in (True, False):
        model = product_model(meta["model"], gold)
        optim = (Adam if fused else torch.optim.Adam)(model.parameters(), lr=5e-4, eps=1e-4)
        sched = IndendityScheduler(optim, 1, 5e-4)
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=1000, enabled=True)
        mse, psnr, lr = T.step(0, model, bvr, optim, sched, torch.nn.MSELoss(), scaler, batches, torch.device("cuda"),
                               split="train", perturb=False, verbose=False)
        steps = optim.steps_applied if fused else int(optim.state[next(model.parameters())]["step"])
        results.append((mse, [p.detach().cpu() for p in model.parameters()], scaler.get_scale(), steps))
    ops.set_precision(None)
    (mse_a, w_a, scale_a, steps_a), (mse_b, w_b, scale_b, steps_b) = results
    assert scale_a == scale_b == 512.0
    assert steps_a == steps_b == 4
    assert mse_a[2] == float("inf") and mse_b[2] == float("inf")
    keep = [0, 1, 3, 4]
    assert_close(torch.tensor(mse_a)[keep], torch.tensor(mse_b)[keep], rtol=1e-5, atol=0, what="losses")
    for k, (a, b) in enumerate(zip(w_a, w_b)):
        assert bool(torch.isfinite(a).all())
        assert_close(a, b, rtol=1e-4, atol=2e-6, what=f"weights after 5 steps (one skipped), tensor {k}")


@pytest.mark.parametrize("shape", ["256x4", "192x4 Fourier d", "32x2"])
def test_folded_colour_head_and_its_unfolded_gradients(ops, shape):
    """The bf16 kernels evaluate rgb.0(phi_d, features(h)) as one layer (model.py:51-53, :76).  Its weight
    and bias, as the library forms them at a refresh, equal [W_r's encoder columns | W_rc W_f] and
    b_r + W_rc b_f in float64 to fp32 rounding; and the step that turns that layer's gradient into the
    gradients of the two original layers equals autograd through the fold in float64."""
    import ctypes as C
    from nerf_amd import _lib
    from nerf_amd.core import FourierFeatures, NeRF, PositionalEncoding
    torch.manual_seed(5)
    if shape == "256x4":
        model = NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 256, 4)
    elif shape == "192x4 Fourier d":
        model = NeRF(PositionalEncoding(3, 6), FourierFeatures(3, 24, 4.0), 192, 4)
    else:
        model = NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 32, 2, resid=False)
    model = model.cuda()
    with torch.no_grad():                                         # biases that matter
        model.features.bias.normal_()
        dict(model.named_parameters())["rgb.0.bias"].normal_()
    handle = ops.handle_for(model)
    ptr = handle.ready(_lib.PREC_BF16)
    sd = {k: v.detach().double().cpu() for k, v in model.named_parameters()}
    W_r, b_r, W_f, b_f = sd["rgb.0.weight"], sd["rgb.0.bias"], sd["features.weight"], sd["features.bias"]
    half, ld = W_r.shape
    width = W_f.shape[0]
    E = ld - width
    p = lambda t: C.c_void_p(t.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    w, b = torch.empty(half, ld, device="cuda"), torch.empty(half, device="cuda")
    _lib.check(_lib.lib.nerfx_debug_folded_head(ptr, p(w), p(b), stream))
    want_w = torch.cat([W_r[:, :E], W_r[:, E:] @ W_f], dim=1)
    want_b = b_r + W_r[:, E:] @ b_f
    assert_close(w.cpu().double(), want_w, 0.0, 2e-7 * float(want_w.abs().max()), "folded weight")
    assert_close(b.cpu().double(), want_b, 0.0, 2e-7 * float(want_b.abs().max()), "folded bias")
    # gradients: any dL/dW', dL/db' -> the four gradients, against autograd through the fold
    g_w, g_b = torch.randn(half, ld, dtype=torch.float64), torch.randn(half, dtype=torch.float64)
    leaves = [t.clone().requires_grad_(True) for t in (W_r, b_r, W_f, b_f)]
    folded_w = torch.cat([leaves[0][:, :E], leaves[0][:, E:] @ leaves[2]], dim=1)
    folded_b = leaves[1] + leaves[0][:, E:] @ leaves[3]
    ((folded_w * g_w).sum() + (folded_b * g_b).sum()).backward()
    outs = [torch.empty_like(t, dtype=torch.float32, device="cuda") for t in (W_r, b_r, W_f, b_f)]
    gw32, gb32 = g_w.float().cuda(), g_b.float().cuda()
    _lib.check(_lib.lib.nerfx_debug_unfold_grads(ptr, p(gw32), p(gb32), p(outs[0]), p(outs[1]), p(outs[2]), p(outs[3]), stream))
    torch.cuda.synchronize()
    for name, got, leaf in zip(("rgb.0.weight", "rgb.0.bias", "features.weight", "features.bias"), outs, leaves):
        assert_close(got.cpu().double(), leaf.grad, 0.0, 3e-6 * float(leaf.grad.abs().max()), f"unfolded gradient of {name}")
