"""Generate `tests/golden/*.npz` by running the REAL reference on CPU.

Runs only in the build container (needs `/root/reference`); the fixtures it
writes are committed, the reference never is.  Usage:

    python tools/make_golden.py            # rewrites tests/golden/

`kornia` (imported by the reference's ray.py for an off-path camera helper) is
not installed; a file-backed stand-in under `tools/_stubs` provides the one
symbol so that the module imports.  Everything captured below executes
unmodified reference code.

Model weights are not stored: they are drawn by `oracle.nerf_oracle.synth_params`
(numpy PCG64, torch-Linear bounds) from the seed recorded in each fixture and
loaded into the reference module with `load_state_dict`.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "_stubs"))
sys.path.insert(0, "/root/reference")

from nerf.core import BoundedVolumeRaymarcher, FourierFeatures, NeRF, PositionalEncoding  # noqa: E402
from nerf.core import ray as ref_ray, renderer as ref_renderer  # noqa: E402
from nerf.core.activations import shifted_softplus, widened_sigmoid  # noqa: E402
from nerf.core.scheduler import IndendityScheduler, LogDecayScheduler, MipNeRFScheduler  # noqa: E402
import nerf.infer as ref_infer  # noqa: E402
from oracle.nerf_oracle import synth_params  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
ACTS = {"relu": torch.nn.ReLU, "leaky_relu": torch.nn.LeakyReLU, "silu": torch.nn.SiLU}

# name -> (enc_x, enc_d, width, depth, activ, resid, seed); enc = ("pe", L) | ("ff", F, sigma)
MODELS = {
    "base":    (("pe", 6), ("pe", 6), 256, 4, "relu", True, 101),
    "readme":  (("pe", 10), ("pe", 4), 256, 8, "silu", True, 102),
    "student": (("pe", 6), ("pe", 6), 32, 2, "relu", False, 103),
    "tiny16":  (("pe", 6), ("pe", 6), 16, 2, "relu", False, 104),
    "leaky":   (("pe", 4), ("pe", 4), 64, 4, "leaky_relu", True, 105),
    "fourier": (("ff", 256, 32.0), ("ff", 64, 16.0), 256, 4, "relu", True, 106),
}


def build_encoder(spec, seed):
    if spec[0] == "pe":
        return PositionalEncoding(3, spec[1]), None
    torch.manual_seed(seed)
    ff = FourierFeatures(3, spec[1], spec[2])
    return ff, ff.b.clone()


def build_model(name):
    ex, ed, width, depth, activ, resid, seed = MODELS[name]
    phi_x, bx = build_encoder(ex, seed * 7 + 1)
    phi_d, bd = build_encoder(ed, seed * 7 + 2)
    m = NeRF(phi_x, phi_d, width, depth, ACTS[activ], resid)
    sd = synth_params(phi_x.o_dim, phi_d.o_dim, width, depth, resid, seed)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("phi_") for k in missing), (missing, unexpected)
    meta = dict(name=name, enc_x=list(ex), enc_d=list(ed), width=width, depth=depth,
                activ=activ, resid=resid, seed=seed, e_x=phi_x.o_dim, e_d=phi_d.o_dim)
    extra = {}
    if bx is not None:
        extra["b_x"] = bx.numpy()
    if bd is not None:
        extra["b_d"] = bd.numpy()
    return m.eval(), meta, extra


def save(name, meta, **arrays):
    os.makedirs(OUT, exist_ok=True)
    arrays = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
              for k, v in arrays.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print(f"  {name}.npz  {sum(a.nbytes for a in arrays.values()) / 1024:.1f} KiB raw")


def rays(B, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, 3, generator=g), torch.rand(B, 3, generator=g)


@torch.no_grad()
def g1_sampling():
    ro, rd = rays(16, 11)
    out = dict(ro=ro, rd=rd)
    for tag, perturb in (("det", False), ("pert", True)):
        torch.manual_seed(1234)
        rx, rds, t, delta = ref_ray.uniform_bounded_rays(ro, rd, 2.0, 6.0, 64, perturb)
        out.update({f"{tag}_rx": rx, f"{tag}_rds": rds, f"{tag}_t": t, f"{tag}_delta": delta})
    torch.manual_seed(1234)
    out["pert_u"] = torch.rand(16, 64)          # the draw rand_like(t) made above
    # an uneven sample count and other bounds
    rx, rds, t, delta = ref_ray.uniform_bounded_rays(ro, rd, 0.5, 3.25, 48, False)
    out.update(alt_rx=rx, alt_t=t, alt_delta=delta)
    save("g1_sampling", dict(tn=2.0, tf=6.0, n=64, alt=dict(tn=0.5, tf=3.25, n=48), seed=1234), **out)


@torch.no_grad()
def g2_encodings():
    g = torch.Generator().manual_seed(21)
    v_unit = torch.rand(64, 3, generator=g)
    v_far = (torch.rand(64, 3, generator=g) * 2 - 1) * 7.0 * 3 ** 0.5   # |arg| up to ~388 at f=32
    out = dict(v_unit=v_unit, v_far=v_far)
    for L in (4, 6, 10):
        pe = PositionalEncoding(3, L)
        out[f"pe{L}_freq"] = pe.freq_bands
        out[f"pe{L}_unit"] = pe(v_unit)
        out[f"pe{L}_far"] = pe(v_far)
    torch.manual_seed(22)
    ff = FourierFeatures(3, 64, 8.0)
    out.update(ff_b=ff.b, ff_unit=ff(v_unit), ff_far=ff(v_far))
    x = torch.linspace(-12, 12, 97)
    out.update(act_x=x, act_softplus=shifted_softplus(x), act_wsigmoid=widened_sigmoid(x))
    save("g2_encodings", dict(ff=dict(features=64, sigma=8.0)), **out)


@torch.no_grad()
def g3_mlp():
    for name in MODELS:
        m, meta, extra = build_model(name)
        g = torch.Generator().manual_seed(31)
        rx = (torch.rand(96, 3, generator=g) * 2 - 1) * 4.0
        rd = torch.rand(96, 3, generator=g) * 2 - 1
        sigma, rgb = m(rx, rd)
        save(f"g3_mlp_{name}", meta, rx=rx, rd=rd, sigma=sigma, rgb=rgb, **extra)


@torch.no_grad()
def g4_composite():
    g = torch.Generator().manual_seed(41)
    B, N = 12, 128
    sigma = torch.rand(B, N, generator=g) * 3
    rgb = torch.rand(B, N, 3, generator=g)
    t, _ = torch.sort(torch.rand(B, N, generator=g) * 4 + 2, dim=-1)
    rd = torch.rand(B, 3, generator=g)
    # adversarial rows: empty space, a wall, a huge density, tiny densities
    sigma[0] = 0
    sigma[1] = 0; sigma[1, 40] = 1e4
    sigma[2] = 1e6
    sigma[3] = 1e-7
    delta = ref_ray.segment_lengths(t, rd)
    w, rgb_map = ref_renderer.raymarch_volume(sigma, rgb, delta, False)
    depth = torch.sum(w * t, dim=-1)
    w_only, none = ref_renderer.raymarch_volume(sigma, rgb, delta, True)
    assert none is None and torch.equal(w, w_only)
    ecp = ref_renderer.exclusive_cumprod(torch.tensor([[2., 3., 4.]]))
    # odd sample count
    w65, rgb65 = ref_renderer.raymarch_volume(sigma[:, :65], rgb[:, :65], ref_ray.segment_lengths(t[:, :65], rd), False)
    save("g4_composite", {}, sigma=sigma, rgb=rgb, t=t, rd=rd, delta=delta, weights=w,
         rgb_map=rgb_map, depth=depth, ecp=ecp, w65=w65, rgb65=rgb65)


@torch.no_grad()
def g5_pdf():
    g = torch.Generator().manual_seed(51)
    B, Nc, Nf = 16, 64, 64
    ro, rd = rays(B, 52)
    t = torch.linspace(2, 6, Nc).expand(B, Nc).contiguous()
    w = torch.rand(B, Nc, generator=g) ** 4
    w[0] = 0                                 # flat pdf
    w[1] = 0; w[1, 20] = 1                   # one spike: den < 1e-5 elsewhere
    w[2] = 0; w[2, 1] = 1; w[2, 62] = 1      # mass at both interior ends
    w[3] = 1e-9
    tm = 0.5 * (t[:, :-1] + t[:, 1:])
    out = dict(ro=ro, rd=rd, t=t, w=w, bins=tm)
    # pdf_z_values internals restated with the reference's own ops to expose cdf/idx
    wi = w[:, 1:-1] + 1e-5
    pdf = wi / torch.sum(wi, dim=-1, keepdim=True)
    cdf = torch.cat((torch.zeros(B, 1), torch.cumsum(pdf, dim=-1)), dim=-1)
    out["cdf"] = cdf
    for tag, perturb in (("det", False), ("pert", True)):
        torch.manual_seed(777)
        t_pdf = ref_ray.pdf_z_values(tm, w[:, 1:-1], Nf, ro.device, perturb)
        torch.manual_seed(777)
        u = torch.rand(B, Nf) if perturb else torch.linspace(0, 1, Nf).expand(B, Nf).contiguous()
        out[f"{tag}_u"] = u
        out[f"{tag}_idx"] = torch.searchsorted(cdf, u, right=True)
        out[f"{tag}_t_pdf"] = t_pdf
        torch.manual_seed(777)
        rx, rds, ts, delta = ref_ray.pdf_rays(ro, rd, t, w, Nf, perturb)
        out.update({f"{tag}_t_sorted": ts, f"{tag}_delta": delta, f"{tag}_rx": rx})
    # searchsorted corner cases on a hand-made cdf: exact hits, u=0, u=1, duplicates
    cdf2 = torch.tensor([[0.0, 0.25, 0.25, 0.5, 1.0], [0.0, 0.0, 0.0, 1.0, 1.0]])
    u2 = torch.tensor([[0.0, 0.25, 0.2499999, 0.5, 0.75, 1.0, 1.5], [0.0, 1e-30, 0.5, 1.0, 0.99999994, 2.0, -1.0]])
    out.update(ss_cdf=cdf2, ss_u=u2, ss_idx=torch.searchsorted(cdf2, u2, right=True))
    save("g5_pdf", dict(nc=Nc, nf=Nf, seed=777), **out)


@torch.no_grad()
def g6_render():
    ro, rd = rays(24, 61)
    for cname, fname, nc, nf in (("base", "base", 64, 64), ("student", "base", 64, 64),
                                 ("base", None, 64, 0), ("student", "student", 64, 128),
                                 ("fourier", "fourier", 32, 32), ("readme", "readme", 64, 64),
                                 ("leaky", "leaky", 40, 24)):
        coarse, cmeta, cextra = build_model(cname)
        fine, fmeta, fextra = (coarse, cmeta, cextra) if fname in (None, cname) else build_model(fname)
        bvr = BoundedVolumeRaymarcher(2.0, 6.0, nc, nf)
        out = dict(ro=ro, rd=rd)
        for tag, perturb in (("det", False), ("pert", True)):
            torch.manual_seed(4321)
            t, w, depth, rgb = bvr.render_volume(coarse, fine, ro, rd, perturb=perturb, train=False)
            out.update({f"{tag}_t": t, f"{tag}_w": w, f"{tag}_depth": depth, f"{tag}_rgb": rgb})
        torch.manual_seed(4321)
        out["pert_u_c"] = torch.rand(24, nc)
        if nf > 0:
            out["pert_u_f"] = torch.rand(24, nf)
        for k, v in cextra.items():
            out["c_" + k] = v
        for k, v in fextra.items():
            out["f_" + k] = v
        save(f"g6_render_{cname}_{fname or 'none'}_{nc}_{nf}",
             dict(coarse=cmeta, fine=fmeta, tn=2.0, tf=6.0, nc=nc, nf=nf, seed=4321), **out)


def g7_train():
    """train.py:75-86 driven directly (nerf.train itself needs torchvision)."""
    for name, B in (("base", 32), ("student", 64)):
        m, meta, extra = build_model(name)
        m.train()
        ro, rd = rays(B, 71)
        g = torch.Generator().manual_seed(72)
        target = torch.rand(B, 3, generator=g)
        bvr = BoundedVolumeRaymarcher(2.0, 6.0, 64, 64)
        optim = torch.optim.Adam(m.parameters(), lr=5e-4, eps=1e-8)
        out = dict(ro=ro, rd=rd, target=target)
        losses = []
        for it in range(3):
            torch.manual_seed(900 + it)
            out[f"u_c{it}"] = torch.rand(B, 64)
            out[f"u_f{it}"] = torch.rand(B, 64)
            torch.manual_seed(900 + it)
            *_, C = bvr.render_volume(m, m, ro, rd, perturb=True, train=True)
            loss = torch.nn.functional.mse_loss(C, target)
            loss.backward()
            if it == 0:
                for k, p in m.named_parameters():
                    out["grad0." + k] = p.grad.clone()
                    assert p.requires_grad
            optim.step()
            optim.zero_grad(set_to_none=True)
            losses.append(loss.item())
            if it in (0, 2):     # every 17th element keeps the fixture small
                for k, p in m.named_parameters():
                    out[f"after{it + 1}." + k] = p.detach().flatten()[::17].clone()
        out["losses"] = np.array(losses, dtype=np.float64)
        save(f"g7_train_{name}", dict(model=meta, tn=2.0, tf=6.0, nc=64, nf=64, lr=5e-4, eps=1e-8,
                                      steps=3), **out, **extra)


@torch.no_grad()
def g8_infer():
    m, meta, extra = build_model("student")
    H = W = 16
    ro, rd = rays(H * W, 81)
    bvr = BoundedVolumeRaymarcher(2.0, 6.0, 64, 64)
    depth, rgb = ref_infer.infer(m, m, bvr, ro, rd, H, W, batch_size=100, verbose=False)
    # the same through the class attribute (NeRF.infer = infer)
    d2, c2 = m.infer(m, bvr, ro, rd, H, W, 64, False)
    assert torch.equal(depth, d2) and torch.equal(rgb, c2)
    # NaN-injected ray: nan_to_num path of infer.py:60
    ro_nan = ro.clone(); ro_nan[5] = float("nan")
    dn, cn = ref_infer.infer(m, m, bvr, ro_nan, rd, H, W, batch_size=100, verbose=False)
    save("g8_infer", dict(model=meta, H=H, W=W, tn=2.0, tf=6.0, nc=64, nf=64), ro=ro, rd=rd,
         depth=depth, rgb=rgb, nan_depth=dn, nan_rgb=cn)


def g9_schedulers():
    out = {}
    p = [torch.nn.Parameter(torch.zeros(1))]
    for tag, mk in (("identity", lambda o: IndendityScheduler(o, 10, 3e-4)),
                    ("logdecay", lambda o: LogDecayScheduler(o, 4, 25, (5e-6, 5e-4), 8)),
                    ("mipnerf", lambda o: MipNeRFScheduler(o, 4, 1, 25, (5e-6, 5e-4), 1e-2))):
        o = torch.optim.SGD(p, lr=1.0)
        s = mk(o)
        lrs = [o.param_groups[0]["lr"]]
        for _ in range(100):
            s.step()
            lrs.append(o.param_groups[0]["lr"])
        out[tag] = np.array(lrs, dtype=np.float64)
    save("g9_schedulers", {}, **out)


def _load_reference_path_module():
    """nerf/data/path.py by file: `import nerf.data` pulls in torchvision (absent)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_data_path", "/root/reference/nerf/data/path.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_data_path"] = mod
    spec.loader.exec_module(mod)
    return mod


@torch.no_grad()
def g10_camera():
    """Camera rays as the reference's CLI builds them (infer.py:110-121).  The pixel grid inside
    pinhole_ray_directions comes from the kornia stand-in: that one convention is unpinned."""
    ref_path = _load_reference_path_module()
    out = {}
    cam_angle_x = 0.6194058656692505                               # infer.py:85
    ranges = (-np.pi, np.pi), (-np.pi, np.pi), (-np.pi, np.pi), (4., 4.)
    poses = ref_path.turnaround_poses(*ranges, 8)
    out["poses8"] = poses
    out["poses5_alt"] = ref_path.turnaround_poses((0.25, 1.5), (-0.5, 0.75), (0., 0.), (3., 5.), 5)
    sizes = ((6, 10), (9, 7), (16, 16))
    for H, W in sizes:
        focal = .5 * W / np.tan(.5 * cam_angle_x)
        prd = ref_ray.pinhole_ray_directions(H, W, focal)
        out[f"prd_{H}x{W}"] = prd
        for k in (0, 3, 5):
            ro, rd = ref_ray.phinhole_ray_projection(prd, poses[k])
            out[f"ro_{H}x{W}_{k}"] = ro.contiguous()
            out[f"rd_{H}x{W}_{k}"] = rd
    save("g10_camera", dict(cam_angle_x=cam_angle_x, sizes=[list(s) for s in sizes], pose_ids=[0, 3, 5]), **out)


def g11_distill():
    """nerf.distill.distill driven for three iterations with a sampler that replays fixed draws."""
    import nerf.distill as ref_distill
    teacher, tmeta, _ = build_model("base")
    student, smeta, _ = build_model("student")
    student.train()
    g = torch.Generator().manual_seed(111)
    B, iters = 48, 3
    draws = []
    for _ in range(iters):
        rx = (torch.rand(B, 3, generator=g) * 2 - 1) * 4.0
        rd = torch.rand(B, 3, generator=g)
        rd = rd / torch.linalg.norm(rd, dim=-1, keepdim=True)
        delta = 0.002 + 0.05 * torch.rand(B, generator=g)
        draws.append((rx, rd, delta))

    class Replay:
        def __init__(self):
            self.k = 0

        def sample(self):
            out = draws[self.k]
            self.k += 1
            return out

    optim = torch.optim.Adam(student.parameters(), lr=5e-4, eps=1e-8)
    sched = IndendityScheduler(optim, iters, 5e-4)
    scaler = torch.cuda.amp.GradScaler(enabled=False)
    H = ref_distill.distill(teacher, student, Replay(), optim, sched, scaler, iterations=iters, verbose=False)
    out = {}
    for k, (rx, rd, delta) in enumerate(draws):
        out.update({f"rx{k}": rx, f"rd{k}": rd, f"delta{k}": delta})
    out["losses"] = np.array([l for l, _ in H.train], dtype=np.float64)
    for k, p in student.named_parameters():
        out["after." + k] = p.detach().clone()
    # the sampler's own shapes and ranges
    torch.manual_seed(5); np.random.seed(5)
    dom = ref_distill.Domain(x=(-4, 4), y=(-4, 4), z=(-4, 4), d=(0.00625, 0.00625))
    vs = ref_distill.VoxelSampler(dom, voxels=4, samples=8)
    rx, rd, delta = vs.sample()
    out.update(vs_grid=vs.rx, vs_voxel_size=vs.voxel_size, vs_rx=rx, vs_rd=rd, vs_delta=delta)
    save("g11_distill", dict(teacher=tmeta, student=smeta, iters=iters, B=B, lr=5e-4, eps=1e-8), **out)


ALL = (g1_sampling, g2_encodings, g3_mlp, g4_composite, g5_pdf, g6_render, g7_train, g8_infer,
       g9_schedulers, g10_camera, g11_distill)

if __name__ == "__main__":
    torch.set_num_threads(8)
    wanted = sys.argv[1:]                      # e.g. `python tools/make_golden.py g10_camera`
    for fn in ALL:
        if wanted and fn.__name__ not in wanted:
            continue
        print(fn.__name__)
        fn()
