"""CPU oracle for the ray-march -> encode -> MLP -> composite path.

TEST INFRASTRUCTURE ONLY.  This file restates, with plain torch CPU ops in
fp32, the algorithm of the reference's `nerf/core` hot path.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it, and there only as the checker / the timed CPU baseline -- never as part
of the shipped render or training path, which is HIP-only.

Parity status: PINNED.  `tools/make_golden.py` imports the real reference in
the build container and stores its outputs under `tests/golden/`;
`tests/test_oracle_golden.py` checks every function here against them.

Every function cites the reference lines it follows (paths are relative to the
reference checkout).  The functions are stateless: a model is a plain dict of
tensors keyed like the reference module's `state_dict()`.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Tuple

import torch
from torch import Tensor

Params = Dict[str, Tensor]

ACTIVATIONS: Dict[str, Callable[[Tensor], Tensor]] = {
    "relu": torch.relu,
    "leaky_relu": lambda x: torch.nn.functional.leaky_relu(x, 0.01),
    "silu": torch.nn.functional.silu,
}


# --------------------------------------------------------------------------
# feature mappings
# --------------------------------------------------------------------------
def positional_freq_bands(n_freqs: int) -> Tensor:
    """`2 ** linspace(1, L-1, L)` -- features.py:100-101 (non-integer exponents)."""
    return 2.0 ** torch.linspace(1, n_freqs - 1, n_freqs)


def encode_positional(v: Tensor, freq_bands: Tensor) -> Tensor:
    """[v, sin(f0 v), cos(f0 v), ..., sin(fL v), cos(fL v)] -- features.py:32-36."""
    pieces = [v]
    for k in range(freq_bands.numel()):
        arg = freq_bands[k] * v
        pieces.append(torch.sin(arg))
        pieces.append(torch.cos(arg))
    return torch.cat(pieces, dim=-1)


def encode_fourier(v: Tensor, b: Tensor) -> Tensor:
    """[sin(2 pi v b^T) | cos(2 pi v b^T)] -- features.py:50-52."""
    a = 2 * 3.141592653589793 * v @ b.T
    return torch.cat((torch.sin(a), torch.cos(a)), dim=-1)


def make_encoder(kind: str, table: Tensor) -> Callable[[Tensor], Tensor]:
    """kind = "pe" (table = freq_bands) or "ff" (table = b matrix)."""
    if kind == "pe":
        return lambda v: encode_positional(v, table)
    if kind == "ff":
        return lambda v: encode_fourier(v, table)
    raise ValueError(kind)


# --------------------------------------------------------------------------
# output activations
# --------------------------------------------------------------------------
def shifted_softplus(x: Tensor) -> Tensor:
    """softplus(x - 1) in the overflow-safe form of activations.py:33-35."""
    s = x - 1
    return torch.log1p(torch.exp(-s.abs())) + s * (s >= 0)


def widened_sigmoid(x: Tensor) -> Tensor:
    """0.5 (1 + 1.002 tanh(x / 2)) -- activations.py:18-20."""
    return 0.5 * (1.0 + (1.0 + 2.0 * 1e-3) * torch.tanh(0.5 * x))


# --------------------------------------------------------------------------
# radiance-field MLP
# --------------------------------------------------------------------------
def _trunk_depth(params: Params, prefix: str) -> int:
    n = 0
    while f"{prefix}.{n}.0.weight" in params:
        n += 1
    return n


def mlp_forward(
    params: Params,
    enc_x: Callable[[Tensor], Tensor],
    enc_d: Callable[[Tensor], Tensor],
    rx: Tensor,
    rd: Tensor,
    activ: str = "relu",
    resid: bool = True,
    density_only: bool = False,
) -> Tuple[Tensor, Optional[Tensor]]:
    """model.py:68-78.  `density_only` skips the colour head (not a reference
    option; the reference computes and discards it, renderer.py:91-95)."""
    act = ACTIVATIONS[activ]
    lin = torch.nn.functional.linear
    phi_x = enc_x(rx)
    h = phi_x
    for i in range(_trunk_depth(params, "fc_1")):
        h = act(lin(h, params[f"fc_1.{i}.0.weight"], params[f"fc_1.{i}.0.bias"]))
    if resid:
        h = torch.cat((phi_x, h), dim=-1)                          # model.py:72
    for i in range(_trunk_depth(params, "fc_2")):
        h = act(lin(h, params[f"fc_2.{i}.0.weight"], params[f"fc_2.{i}.0.bias"]))
    sigma = shifted_softplus(lin(h, params["sigma.0.weight"], params["sigma.0.bias"])).view(-1)
    if density_only:
        return sigma, None
    feat = lin(h, params["features.weight"], params["features.bias"])
    c = torch.cat((enc_d(rd), feat), dim=-1)                       # model.py:76
    c = act(lin(c, params["rgb.0.weight"], params["rgb.0.bias"]))
    rgb = widened_sigmoid(lin(c, params["rgb.2.weight"], params["rgb.2.bias"]))
    return sigma, rgb


# --------------------------------------------------------------------------
# ray sampling
# --------------------------------------------------------------------------
def segment_lengths(t: Tensor, rd: Tensor) -> Tensor:
    """ray.py:60-66: forward differences, last = 1e10, times |rd| (not unit)."""
    d = t[:, 1:] - t[:, :-1]
    d = torch.cat((d, torch.full((t.size(0), 1), 1e10)), dim=-1)
    return d * torch.norm(rd[:, None, :], dim=-1)


def uniform_z_values(tn: float, tf: float, batch: int, n: int, u: Optional[Tensor]) -> Tensor:
    """ray.py:91-99.  `u` = the uniforms the reference draws with `rand_like(t)`;
    None means perturb=False."""
    t = torch.linspace(tn, tf, n).expand(batch, n).contiguous()
    if u is None:
        return t
    mid = 0.5 * (t[:, :-1] + t[:, 1:])
    hi = torch.cat((mid, t[:, -1:]), dim=-1)
    lo = torch.cat((t[:, :1], mid), dim=-1)
    return lo + (hi - lo) * u


def uniform_rays(ro: Tensor, rd: Tensor, tn: float, tf: float, n: int, u: Optional[Tensor]):
    """ray.py:127-135 -> (rx, rds, t, delta)."""
    t = uniform_z_values(tn, tf, ro.size(0), n, u)
    delta = segment_lengths(t, rd)
    rx = ro[:, None, :] + rd[:, None, :] * t[:, :, None]
    rds = rd[:, None, :].expand(-1, n, -1).contiguous()
    return rx, rds, t, delta


def pdf_cdf(weights: Tensor) -> Tensor:
    """ray.py:161-164: cdf = [0, cumsum((w + 1e-5) / sum(w + 1e-5))]."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    return torch.cat((torch.zeros_like(cdf[:, :1]), cdf), dim=-1)


def searchsorted_right(cdf: Tensor, u: Tensor) -> Tensor:
    """idx = #{k : cdf_k <= u} -- `torch.searchsorted(..., right=True)`, ray.py:174.
    Restated as a count so the oracle does not lean on the op it checks."""
    return (cdf[:, None, :] <= u[:, :, None]).sum(dim=-1)


def inverse_cdf(bins: Tensor, cdf: Tensor, u: Tensor) -> Tuple[Tensor, Tensor]:
    """ray.py:174-186 -> (idx, t)."""
    n = cdf.size(1) - 1
    idx = searchsorted_right(cdf, u)
    below = torch.clamp_min(idx - 1, 0)
    above = torch.clamp_max(idx, n)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    den = cdf_a - cdf_b
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    t = (u - cdf_b) / den
    return idx, bin_b + t * (bin_a - bin_b)


def pdf_z_values(bins: Tensor, weights: Tensor, n: int, u: Optional[Tensor]):
    """ray.py:158-188 -> (cdf, u, idx, t_pdf); `u` None = linspace(0, 1, n)."""
    cdf = pdf_cdf(weights)
    if u is None:
        u = torch.linspace(0, 1, n).expand(weights.size(0), n).contiguous()
    idx, t = inverse_cdf(bins, cdf, u)
    return cdf, u, idx, t


def pdf_rays(ro: Tensor, rd: Tensor, t: Tensor, weights: Tensor, n_fine: int, u: Optional[Tensor]):
    """ray.py:216-229 -> (rx, rds, t_sorted, delta)."""
    mid = 0.5 * (t[:, :-1] + t[:, 1:])
    _, _, _, t_pdf = pdf_z_values(mid, weights[:, 1:-1], n_fine, u)
    t_all, _ = torch.sort(torch.cat((t, t_pdf), dim=-1), dim=-1)
    delta = segment_lengths(t_all, rd)
    rx = ro[:, None, :] + rd[:, None, :] * t_all[:, :, None]
    rds = rd[:, None, :].expand(-1, t_all.size(1), -1).contiguous()
    return rx, rds, t_all, delta


# --------------------------------------------------------------------------
# compositing
# --------------------------------------------------------------------------
def exclusive_cumprod(x: Tensor) -> Tensor:
    """T_i = prod_{j<i} x_j -- renderer.py:23-26 without the in-place write."""
    cp = torch.cumprod(x, dim=-1)
    return torch.cat((torch.ones_like(cp[..., :1]), cp[..., :-1]), dim=-1)


def raymarch_volume(sigma: Tensor, rgb: Optional[Tensor], delta: Tensor):
    """renderer.py:48-57 -> (weights, rgb_map or None)."""
    alpha = 1.0 - torch.exp(-sigma * delta)
    trans = exclusive_cumprod(1.0 - alpha + 1e-10)
    w = alpha * trans
    if rgb is None:
        return w, None
    return w, torch.sum(w[:, :, None] * rgb, dim=-2)


# --------------------------------------------------------------------------
# the boundary: render_volume
# --------------------------------------------------------------------------
class Model:
    """A radiance field for the oracle: tensors + the two encoders + options."""

    def __init__(self, params: Params, enc_x, enc_d, activ: str = "relu", resid: bool = True):
        self.params, self.enc_x, self.enc_d = params, enc_x, enc_d
        self.activ, self.resid = activ, resid

    def __call__(self, rx, rd, density_only=False):
        return mlp_forward(self.params, self.enc_x, self.enc_d, rx, rd,
                           self.activ, self.resid, density_only)


def render_coarse(model: Model, ro, rd, tn, tf, n, u, weights_only: bool, as_reference: bool = False):
    """renderer.py:87-97.  The reference evaluates the colour head even when only the weights are
    used (renderer.py:91); `as_reference` does the same (for timing), otherwise it is skipped."""
    B = ro.size(0)
    rx, rds, t, delta = uniform_rays(ro, rd, tn, tf, n, u)
    sigma, rgb = model(rx, rds, density_only=weights_only and not as_reference)
    if weights_only:
        rgb = None
    w, rgb_map = raymarch_volume(sigma.view(B, n), None if rgb is None else rgb.view(B, n, 3), delta)
    return t, w, rgb_map


def render_volume(coarse: Model, fine: Model, ro: Tensor, rd: Tensor, tn: float, tf: float,
                  n_coarse: int, n_fine: int, u_coarse: Optional[Tensor] = None,
                  u_fine: Optional[Tensor] = None, as_reference: bool = False):
    """renderer.py:198-205 (+ :131-144 for the fine branch) ->
    (t, weights, depth_map, rgb_map).  `u_coarse` / `u_fine` are the uniforms the
    reference draws (in that order) when perturb=True; None = perturb off."""
    B = ro.size(0)
    if n_fine > 0:
        with torch.no_grad():                                     # renderer.py:133-135
            t, w, _ = render_coarse(coarse, ro, rd, tn, tf, n_coarse, u_coarse, True, as_reference)
            rx, rds, t, delta = pdf_rays(ro, rd, t, w, n_fine, u_fine)
        n = n_coarse + n_fine
        sigma, rgb = fine(rx, rds)
        w, rgb_map = raymarch_volume(sigma.view(B, n), rgb.view(B, n, 3), delta)
    else:
        t, w, rgb_map = render_coarse(coarse, ro, rd, tn, tf, n_coarse, u_coarse, False)
    depth = torch.sum(w * t, dim=-1)                               # renderer.py:203
    return t, w, depth, rgb_map


def postprocess_frame(depth: Tensor, rgb: Tensor, H: int, W: int):
    """infer.py:58-66: frame-global depth normalisation, clip, x255."""
    d = depth.view(H, W, 1).repeat((1, 1, 3)).nan_to_num()
    d = d - d.min()
    d = d / (d.max() + 1e-10)
    return d.clip(0, 1) * 255, rgb.view(H, W, 3).clip(0, 1) * 255


@torch.no_grad()
def infer(coarse: Model, fine: Model, ro, rd, H, W, tn, tf, n_coarse, n_fine, batch_size=16384,
          as_reference: bool = False):
    """infer.py:42-68 -> (depth_map, rgb_map), both (H, W, 3) in [0, 255].  `as_reference`: also
    execute the work the reference executes and discards (the coarse pass's colour head)."""
    n = ro.size(0)
    depth, rgb = torch.zeros(H * W), torch.zeros(H * W, 3)
    for s in range(0, n, batch_size):
        e = min(s + batch_size, n)
        *_, D, C = render_volume(coarse, fine, ro[s:e], rd[s:e], tn, tf, n_coarse, n_fine, as_reference=as_reference)
        depth[s:e], rgb[s:e] = D, C
    return postprocess_frame(depth, rgb, H, W)


# ---- camera rays and poses (callers of the path) ---------------------------------------
def pinhole_ray_directions(H: int, W: int, focal: float) -> Tensor:
    """ray.py:10-26 -> (1, H, W, 3).  The pixel grid comes from kornia's create_meshgrid
    (0.5.11, not vendored by the reference: parity unpinned for that one call): x = column
    index 0..W-1, y = row index 0..H-1, un-normalised."""
    j, i = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
    i, j = i[None], j[None]
    rdx = (i - .5 * W) / focal
    rdy = (j - .5 * H) / focal
    rdz = torch.ones_like(i)
    return torch.stack((rdx, -rdy, -rdz), dim=-1)


def pinhole_ray_projection(prd: Tensor, c2w: Tensor) -> Tuple[Tensor, Tensor]:
    """ray.py:30-46 (`phinhole_ray_projection`): rotate, normalise, broadcast the origin."""
    rd = prd @ c2w[:3, :3].T
    rd = rd / torch.norm(rd, dim=-1, keepdim=True)
    return c2w[:3, 3].expand(rd.size()), rd


def turnaround(theta: Tensor, phi: Tensor, psy: Tensor, radius: Tensor) -> Tensor:
    """path.py:9-100: diag(-1,1,1,1) @ R_theta @ R_phi @ R_psy @ T_z, fp32."""
    c, s = torch.cos, torch.sin
    t = torch.tensor([[c(theta), 0., s(theta), 0.], [0., 1., 0., 0.], [-s(theta), 0., c(theta), 0.], [0., 0., 0., 1.]])
    p = torch.tensor([[1., 0., 0., 0.], [0., c(phi), -s(phi), 0.], [0., s(phi), c(phi), 0.], [0., 0., 0., 1.]])
    y = torch.tensor([[c(psy), -s(psy), 0., 0.], [s(psy), c(psy), 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]])
    z = torch.tensor([[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., radius], [0., 0., 0., 1.]])
    flip = torch.tensor([[-1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]])
    return flip @ t @ p @ y @ z


def turnaround_poses(theta, phi, psy, radius, samples: int) -> Tensor:
    """path.py:104-128 -> (samples, 4, 4)."""
    cols = [torch.linspace(lo, hi, samples + 1)[:-1] for lo, hi in (theta, phi, psy, radius)]
    return torch.stack([turnaround(*tpyr) for tpyr in zip(*cols)], dim=0)


def frame_to_uint8(frame: Tensor):
    """infer.py:145-146: `frame.numpy().astype(np.uint8)` (truncation)."""
    return frame.numpy().astype("uint8")


def distill_loss(s_sigma: Tensor, s_rgb: Tensor, t_sigma: Tensor, t_rgb: Tensor, delta: Tensor) -> Tensor:
    """distill.py:123-130, including its (B, 1) x (B,) broadcast of densities against lengths."""
    t_alpha = 1. - torch.exp(-t_sigma.view(-1, 1) * delta)
    s_alpha = 1. - torch.exp(-s_sigma.view(-1, 1) * delta)
    loss_alpha = torch.mean((s_alpha - t_alpha) ** 2, dim=-1)
    loss_rgb = torch.mean((s_rgb - t_rgb) ** 2, dim=-1)
    return torch.mean(loss_alpha + loss_rgb, dim=0)


def distill_step(teacher: "Model", student: "Model", rx: Tensor, rd: Tensor, delta: Tensor,
                 optim: torch.optim.Optimizer):
    """One iteration of distill.py:117-138 without the scaler.  -> (loss, student grads)."""
    with torch.no_grad():
        t_sigma, t_rgb = teacher(rx, rd)
    s_sigma, s_rgb = student(rx, rd)
    loss = distill_loss(s_sigma, s_rgb, t_sigma, t_rgb, delta)
    optim.zero_grad(set_to_none=True)
    loss.backward()
    grads = {k: v.grad.detach().clone() for k, v in student.params.items() if v.grad is not None}
    optim.step()
    return loss.detach(), grads


def train_step(params: Params, enc_x, enc_d, activ, resid, ro, rd, target, tn, tf,
               n_coarse, n_fine, u_coarse, u_fine, optim: torch.optim.Optimizer):
    """train.py:75-86 with the GradScaler disabled: one render(train=True) + MSE
    + backward + optimizer step.  `params` must be leaf tensors with
    requires_grad=True that `optim` owns.  Returns (loss, grads dict)."""
    model = Model(params, enc_x, enc_d, activ, resid)
    *_, rgb_map = render_volume(model, model, ro, rd, tn, tf, n_coarse, n_fine, u_coarse, u_fine)
    loss = torch.nn.functional.mse_loss(rgb_map, target)
    optim.zero_grad(set_to_none=True)
    loss.backward()
    grads = {k: v.grad.detach().clone() for k, v in params.items() if v.grad is not None}
    optim.step()
    return loss.detach(), grads


def psnr(loss: Tensor) -> Tensor:
    """train.py:92."""
    return -10.0 * torch.log10(loss)


# --------------------------------------------------------------------------
# deterministic synthetic weights (shared by fixtures, tests and bench)
# --------------------------------------------------------------------------
def layer_shapes(e_x: int, e_d: int, width: int, depth: int, resid: bool):
    """(key, out, in) for every Linear of model.py:39-55, in state_dict order."""
    hd = depth // 2
    shapes = [("fc_1.0.0", width, e_x)] + [(f"fc_1.{i}.0", width, width) for i in range(1, hd)]
    shapes += [("fc_2.0.0", width, width + (e_x if resid else 0))]
    shapes += [(f"fc_2.{i}.0", width, width) for i in range(1, hd)]
    shapes += [("sigma.0", 1, width), ("features", width, width),
               ("rgb.0", width // 2, e_d + width), ("rgb.2", 3, width // 2)]
    return shapes


def synth_params(e_x: int, e_d: int, width: int, depth: int, resid: bool, seed: int) -> Params:
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) weights and biases (the bound torch's
    Linear uses) from numpy's PCG64, so fixtures need not store weights."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out: Params = {}
    for key, n_out, n_in in layer_shapes(e_x, e_d, width, depth, resid):
        bound = 1.0 / math.sqrt(n_in)
        out[f"{key}.weight"] = torch.from_numpy(
            rng.uniform(-bound, bound, size=(n_out, n_in)).astype(np.float32))
        out[f"{key}.bias"] = torch.from_numpy(
            rng.uniform(-bound, bound, size=(n_out,)).astype(np.float32))
    return out
