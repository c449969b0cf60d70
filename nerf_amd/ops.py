"""Tensor-level wrappers over the C ABI: validate, allocate outputs with torch,
pass raw device pointers and the current HIP stream.  No arithmetic happens here.
"""
import ctypes as C
import weakref
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import lib, check

_PRECISION_OVERRIDE = None       # None | "fp32" | "bf16"


def set_precision(mode: Optional[str]) -> None:
    """Force the MLP arithmetic ("fp32" / "bf16"); None = follow torch autocast."""
    global _PRECISION_OVERRIDE
    assert mode in (None, "fp32", "bf16")
    _PRECISION_OVERRIDE = mode


def precision() -> int:
    """bf16 MFMA when the caller is inside `torch.autocast` (the reference's --amp
    switch, train.py:75, infer.py:132), exact fp32 MFMA otherwise."""
    if _PRECISION_OVERRIDE is not None:
        return _lib.PREC_BF16 if _PRECISION_OVERRIDE == "bf16" else _lib.PREC_FP32
    return _lib.PREC_BF16 if torch.is_autocast_enabled() else _lib.PREC_FP32


def _dev(t: Tensor, name: str, dtype=torch.float32) -> Tensor:
    if not isinstance(t, Tensor):
        raise TypeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"nerf_amd runs on ROCm devices only: `{name}` is on {t.device}")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _p(t: Optional[Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- encoders ----------------------------------------------------------------
def encode(v: Tensor, kind: int, n: int, table: Tensor) -> Tensor:
    v = _dev(v, "v")
    table = _dev(table, "table")
    assert v.shape[-1] == 3, "encoders take 3-vectors"
    width = 3 + 6 * n if kind == _lib.ENC_POSITIONAL else 2 * n
    flat = v.reshape(-1, 3)
    out = torch.empty((flat.shape[0], width), device=v.device, dtype=torch.float32)
    with torch.cuda.device(v.device):
        check(lib.nerfx_encode_fwd(_p(flat), flat.shape[0], kind, n, _p(table), _p(out), _stream()))
    return out.reshape(*v.shape[:-1], width)


def sample_encode(ro: Tensor, rd: Tensor, tn: float, tf: float, n: int, u: Optional[Tensor] = None,
                  enc_x=None, enc_d=None, want_rx: bool = True):
    """K1 standalone.  enc_x / enc_d = (kind, n, table) or None.
    -> dict(t, delta, rx, phi_x, phi_d)."""
    ro, rd = _dev(ro, "ro"), _dev(rd, "rd")
    B = ro.shape[0]
    u = None if u is None else _dev(u, "u")
    dev = ro.device
    out = {"t": torch.empty((B, n), device=dev), "delta": torch.empty((B, n), device=dev),
           "rx": torch.empty((B, n, 3), device=dev) if want_rx else None, "phi_x": None, "phi_d": None}
    kx = nx = kd = nd = 0
    tx = td = None
    if enc_x is not None:
        kx, nx, tx = enc_x[0], enc_x[1], _dev(enc_x[2], "enc_x table")
        out["phi_x"] = torch.empty((B, n, 3 + 6 * nx if kx == 0 else 2 * nx), device=dev)
    if enc_d is not None:
        kd, nd, td = enc_d[0], enc_d[1], _dev(enc_d[2], "enc_d table")
        out["phi_d"] = torch.empty((B, 3 + 6 * nd if kd == 0 else 2 * nd), device=dev)
    with torch.cuda.device(dev):
        check(lib.nerfx_sample_encode_fwd(_p(ro), _p(rd), B, tn, tf, n, _p(u), kx, nx, _p(tx), kd, nd, _p(td),
                                          _p(out["t"]), _p(out["delta"]), _p(out["rx"]), _p(out["phi_x"]),
                                          _p(out["phi_d"]), _stream()))
    return out


# ---- compositing ----------------------------------------------------------------
def composite_fwd(sigma: Tensor, rgb: Optional[Tensor], t: Optional[Tensor], rd: Optional[Tensor],
                  delta: Optional[Tensor] = None, want_depth: bool = True):
    sigma = _dev(sigma, "sigma")
    B, n = sigma.shape
    rgb = None if rgb is None else _dev(rgb, "rgb")
    t = None if t is None else _dev(t, "t")
    rd = None if rd is None else _dev(rd, "rd")
    delta = None if delta is None else _dev(delta, "delta")
    dev = sigma.device
    w = torch.empty((B, n), device=dev)
    rgb_map = torch.empty((B, 3), device=dev) if rgb is not None else None
    depth = torch.empty((B,), device=dev) if (t is not None and want_depth) else None
    with torch.cuda.device(dev):
        check(lib.nerfx_composite_fwd(_p(sigma), _p(rgb), _p(t), _p(rd), _p(delta), B, n, _p(w), _p(rgb_map),
                                      _p(depth), _stream()))
    return w, rgb_map, depth


def composite_bwd(sigma, rgb, t, rd, delta, g_rgb_map, g_depth, g_weights):
    sigma = _dev(sigma, "sigma")
    B, n = sigma.shape
    rgb = None if rgb is None else _dev(rgb, "rgb")
    t = None if t is None else _dev(t, "t")
    rd = None if rd is None else _dev(rd, "rd")
    delta = None if delta is None else _dev(delta, "delta")
    g_rgb_map = None if g_rgb_map is None else _dev(g_rgb_map, "g_rgb_map")
    g_depth = None if g_depth is None else _dev(g_depth, "g_depth")
    g_weights = None if g_weights is None else _dev(g_weights, "g_weights")
    g_sigma = torch.empty_like(sigma)
    g_rgb = torch.empty_like(rgb) if rgb is not None else None
    with torch.cuda.device(sigma.device):
        check(lib.nerfx_composite_bwd(_p(sigma), _p(rgb), _p(t), _p(rd), _p(delta), B, n, _p(g_rgb_map),
                                      _p(g_depth), _p(g_weights), _p(g_sigma), _p(g_rgb), _stream()))
    return g_sigma, g_rgb


# ---- hierarchical sampling --------------------------------------------------------
def pdf_sample(t_c: Tensor, w_c: Tensor, nf: int, u: Optional[Tensor] = None, debug: bool = False):
    """-> t_sorted (B, nc+nf)  [, cdf (B, nc-1), idx int64 (B, nf), t_pdf (B, nf) if debug]."""
    t_c, w_c = _dev(t_c, "t_c"), _dev(w_c, "w_c")
    B, nc = t_c.shape
    u = None if u is None else _dev(u, "u")
    dev = t_c.device
    ts = torch.empty((B, nc + nf), device=dev)
    cdf = idx = tp = None
    if debug:
        cdf = torch.empty((B, nc - 1), device=dev)
        idx = torch.empty((B, nf), device=dev, dtype=torch.int64)
        tp = torch.empty((B, nf), device=dev)
    with torch.cuda.device(dev):
        check(lib.nerfx_pdf_sample(_p(t_c), _p(w_c), B, nc, nf, _p(u), _p(ts), _p(cdf), _p(idx), _p(tp), _stream()))
    return (ts, cdf, idx, tp) if debug else ts


def coarse_pdf_sample(sigma_c: Tensor, t_c: Tensor, rd: Tensor, nf: int, u: Optional[Tensor] = None) -> Tensor:
    """Coarse weights (renderer.py:50-55) + pdf_rays' z values (ray.py:158-188, :220-223) -> t_sorted (B, nc+nf)."""
    sigma_c, t_c, rd = _dev(sigma_c, "sigma_c"), _dev(t_c, "t_c"), _dev(rd, "rd")
    B, nc = t_c.shape
    u = None if u is None else _dev(u, "u")
    ts = torch.empty((B, nc + nf), device=t_c.device)
    scratch = torch.empty((B, nc), device=t_c.device)
    with torch.cuda.device(t_c.device):
        check(lib.nerfx_coarse_pdf_sample(_p(sigma_c), _p(t_c), _p(rd), B, nc, nf, _p(u), _p(scratch), _p(ts), _stream()))
    return ts


def searchsorted_right(cdf: Tensor, u: Tensor) -> Tensor:
    cdf, u = _dev(cdf, "cdf"), _dev(u, "u")
    idx = torch.empty(u.shape, device=u.device, dtype=torch.int64)
    with torch.cuda.device(u.device):
        check(lib.nerfx_searchsorted_right(_p(cdf), _p(u), cdf.shape[0], cdf.shape[1], u.shape[1], _p(idx), _stream()))
    return idx


# ---- model handles ------------------------------------------------------------------
_ACTS = {"ReLU": _lib.ACT_RELU, "LeakyReLU": _lib.ACT_LEAKY_RELU, "SiLU": _lib.ACT_SILU}


def child(module, *names):
    """Submodule by child names; works for nn.Sequential and for its scripted form alike
    (a ScriptModule has no __getitem__)."""
    for name in names:
        module = dict(module.named_children())[name]
    return module


def activation_name(module) -> str:
    act = child(module, "fc_1", "0", "1")
    return getattr(act, "original_name", type(act).__name__)       # ScriptModule or nn.Module


def _activation_code(module) -> int:
    act = child(module, "fc_1", "0", "1")
    name = activation_name(module)
    if name not in _ACTS:
        raise NotImplementedError(f"hidden activation {name} has no HIP kernel (ReLU / LeakyReLU / SiLU)")
    if name == "LeakyReLU" and abs(float(getattr(act, "negative_slope", 0.01)) - 0.01) > 1e-12:
        raise NotImplementedError("LeakyReLU kernels use the default negative_slope=0.01")
    return _ACTS[name]


def _encoder_spec(enc) -> Tuple[int, int, Tensor]:
    bufs = dict(enc.named_buffers())
    if "freq_bands" in bufs:
        return _lib.ENC_POSITIONAL, bufs["freq_bands"].numel(), bufs["freq_bands"]
    if "b" in bufs:
        return _lib.ENC_FOURIER, bufs["b"].shape[0], bufs["b"]
    raise NotImplementedError("encoder must hold a `freq_bands` or `b` buffer")


class ModelHandle:
    """Owns a nerfx_model_t: packed copies of one module's weights on one device.
    Works for `nerf_amd.core.NeRF` and for a `torch.jit.load`ed reference model
    (same state_dict keys)."""

    def __init__(self, module):
        sd = dict(module.named_parameters())
        hd = 0
        while f"fc_1.{hd}.0.weight" in sd:
            hd += 1
        width = sd["fc_1.0.0.weight"].shape[0]
        kx, nx, tx = _encoder_spec(module.phi_x)
        kd, nd, td = _encoder_spec(module.phi_d)
        e_x = 3 + 6 * nx if kx == _lib.ENC_POSITIONAL else 2 * nx
        self.desc = _lib.ModelDesc(kx, nx, kd, nd, width, 2 * hd, _activation_code(module),
                                   int(sd["fc_2.0.0.weight"].shape[1] == width + e_x))
        keys = [f"fc_1.{i}.0" for i in range(hd)] + [f"fc_2.{i}.0" for i in range(hd)]
        keys += ["sigma.0", "features", "rgb.0", "rgb.2"]
        self._sources = [tx, td] + [sd[f"{k}.{w}"] for k in keys for w in ("weight", "bias")]
        self.device = sd["fc_1.0.0.weight"].device
        if self.device.type != "cuda":
            raise RuntimeError(f"nerf_amd runs on ROCm devices only: the model is on {self.device}")
        self._ptr = C.c_void_p()
        self._epoch = 0
        tensors, arr = self._gather()
        with torch.cuda.device(self.device):
            check(lib.nerfx_model_create(C.byref(self.desc), arr, len(tensors), _stream(), C.byref(self._ptr)))
        self._stamp = [self._current_stamp()] * 2                   # per precision: create packs both

    def _gather(self):
        tensors = [_dev(t.detach(), "weight") for t in self._sources]
        arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        return tensors, arr

    def _current_stamp(self):
        return (self._epoch,) + tuple((t.data_ptr(), t._version) for t in self._sources)

    def invalidate(self):
        """Force a re-pack at the next use (see `ops.invalidate`)."""
        self._epoch += 1

    def ready(self, prec: int):
        """The handle, with the packed copies of precision `prec` brought up to date if any
        source tensor was written since they were packed (optimizer step, load_state_dict)."""
        stamp = self._current_stamp()
        if stamp != self._stamp[prec]:
            tensors, arr = self._gather()
            # epoch + the two encoder tables: a change there (or an invalidate) re-packs everything
            full = stamp[:3] != self._stamp[prec][:3]
            with torch.cuda.device(self.device):
                if full:
                    check(lib.nerfx_model_refresh(self._ptr, arr, len(tensors), _stream()))
                    self._stamp = [stamp, stamp]
                else:
                    check(lib.nerfx_model_refresh_precision(self._ptr, arr, len(tensors), prec, _stream()))
            self._stamp[prec] = stamp
        return self._ptr

    @property
    def ptr(self):
        return self._ptr

    def __del__(self):
        try:
            if self._ptr:
                lib.nerfx_model_destroy(self._ptr)
                self._ptr = C.c_void_p()
        except Exception:
            pass


_HANDLES = weakref.WeakKeyDictionary()


def handle_for(module) -> ModelHandle:
    h = _HANDLES.get(module)
    dev = next(module.parameters()).device
    if h is None or h.device != dev:
        h = ModelHandle(module)
        _HANDLES[module] = h
    return h


def invalidate(module) -> None:
    """Tell the kernels that `module`'s weights changed behind autograd's back.  The packed
    weight copies are refreshed whenever a parameter's storage pointer or version counter
    changes, which covers optimizers, `load_state_dict`, `copy_` under no_grad and `.to()`.
    Writes through `p.data` (its version counter is separate from the parameter's) or through
    a raw pointer (another library, a collective on `p.data`) leave no trace: call this after
    them."""
    h = _HANDLES.get(module)
    if h is not None:
        h.invalidate()


# ---- MLP --------------------------------------------------------------------------------
def mlp_points(handle: ModelHandle, x: Tensor, d: Tensor, density_only: bool = False, prec: Optional[int] = None):
    x, d = _dev(x, "rx").reshape(-1, 3), _dev(d, "rd").reshape(-1, 3)
    assert x.shape == d.shape, "positions and directions must pair up"
    M = x.shape[0]
    sigma = torch.empty((M,), device=x.device)
    rgb = None if density_only else torch.empty((M, 3), device=x.device)
    with torch.cuda.device(x.device):
        prec = precision() if prec is None else prec
        check(lib.nerfx_mlp_fwd_points(handle.ready(prec), prec, _p(x), _p(d), M,
                                       _p(sigma), _p(rgb), _stream()))
    return sigma, rgb


def mlp_rays(handle: ModelHandle, ro: Tensor, rd: Tensor, t: Tensor, density_only: bool = False,
             prec: Optional[int] = None):
    ro, rd, t = _dev(ro, "ro"), _dev(rd, "rd"), _dev(t, "t")
    B, n = t.shape
    sigma = torch.empty((B, n), device=t.device)
    rgb = None if density_only else torch.empty((B, n, 3), device=t.device)
    with torch.cuda.device(t.device):
        prec = precision() if prec is None else prec
        check(lib.nerfx_mlp_fwd_rays(handle.ready(prec), prec, _p(ro), _p(rd), _p(t), B, n,
                                     _p(sigma), _p(rgb), _stream()))
    return sigma, rgb


# ---- fused render -------------------------------------------------------------------------
# When set to a list, `render` issues the same kernels stage by stage through the C ABI
# (exactly the sequence nerfx_render_fwd enqueues) with an event pair around each, and
# appends (stage, n_rays, n_samples, colour, start_event, stop_event) records to it.
PROFILE = None


def _render_staged(coarse, fine, ro, rd, tn, tf, nc, nf, u_c, u_f, prec, log):
    def timed(stage, n_samples, colour, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        log.append((stage, ro.shape[0], n_samples, colour, a, b))
        return out

    if nf == 0:
        t = timed("sample", nc, False, lambda: sample_encode(ro, rd, tn, tf, nc, u_c, want_rx=False)["t"])
        sigma, rgb = timed("mlp", nc, True, lambda: mlp_rays(coarse, ro, rd, t, prec=prec))
        w, rgb_map, depth = timed("composite", nc, True, lambda: composite_fwd(sigma, rgb, t, rd))
        return t, w, depth, rgb_map
    t_c = timed("sample", nc, False, lambda: sample_encode(ro, rd, tn, tf, nc, u_c, want_rx=False)["t"])
    sigma_c, _ = timed("mlp", nc, False, lambda: mlp_rays(coarse, ro, rd, t_c, density_only=True, prec=prec))
    t = timed("pdf", nc + nf, False, lambda: coarse_pdf_sample(sigma_c, t_c, rd, nf, u_f))
    sigma, rgb = timed("mlp", nc + nf, True, lambda: mlp_rays(fine, ro, rd, t, prec=prec))
    w, rgb_map, depth = timed("composite", nc + nf, True, lambda: composite_fwd(sigma, rgb, t, rd))
    return t, w, depth, rgb_map


def next_rng(device) -> "_lib.Rng":
    """A fresh (seed, offset) for one render's in-kernel uniforms, taken from torch's generator of
    `device` -- the generator the reference's rand_like / rand draw from (ray.py:99, :167) -- whose
    offset is advanced, so torch.manual_seed makes perturbed renders reproducible and every call
    (and every rank seeded with distributed.seed_ranks) gets its own jitter."""
    dev = torch.device(device)
    gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
    offset = gen.get_offset()
    gen.set_offset(offset + 4)
    return _lib.Rng(gen.initial_seed() & 0xFFFFFFFFFFFFFFFF, offset)


def rng_uniform(rng, which: int, n: int, device) -> Tensor:
    """The uniforms the kernels draw for `rng` (draw 0: coarse stratification, 1: inverse CDF)."""
    out = torch.empty((n,), device=device)
    with torch.cuda.device(out.device):
        check(lib.nerfx_rng_uniform(rng.seed, rng.offset, which, n, _p(out), _stream()))
    return out


def render(coarse: ModelHandle, fine: Optional[ModelHandle], ro: Tensor, rd: Tensor, tn: float, tf: float,
           nc: int, nf: int, u_c: Optional[Tensor] = None, u_f: Optional[Tensor] = None,
           prec: Optional[int] = None, rng=None):
    """render_volume without a graph.  Jitter: `u_c` / `u_f` where given, else drawn in the kernels
    when `rng` (see next_rng) is given, else none."""
    ro, rd = _dev(ro, "ro"), _dev(rd, "rd")
    B, n, dev = ro.shape[0], nc + nf, ro.device
    u_c = None if u_c is None else _dev(u_c, "u_c")
    u_f = None if u_f is None else _dev(u_f, "u_f")
    if PROFILE is not None:
        if rng is not None:                                          # the staged path takes arrays: the same numbers
            u_c = rng_uniform(rng, 0, B * nc, dev).view(B, nc) if u_c is None else u_c
            u_f = rng_uniform(rng, 1, B * nf, dev).view(B, nf) if (u_f is None and nf > 0) else u_f
        return _render_staged(coarse, fine, ro, rd, tn, tf, nc, nf, u_c, u_f,
                              precision() if prec is None else prec, PROFILE)
    t = torch.empty((B, n), device=dev)
    w = torch.empty((B, n), device=dev)
    depth = torch.empty((B,), device=dev)
    rgb_map = torch.empty((B, 3), device=dev)
    ws_bytes = lib.nerfx_render_workspace_bytes(B, nc, nf)
    ws = torch.empty((max(ws_bytes, 1),), device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        prec = precision() if prec is None else prec
        check(lib.nerfx_render_fwd_rng(coarse.ready(prec), None if fine is None else fine.ready(prec),
                                       prec, _p(ro), _p(rd), B, tn, tf, nc, nf, _p(u_c), _p(u_f),
                                       None if rng is None else C.byref(rng), _p(ws), ws_bytes,
                                       _p(t), _p(w), _p(depth), _p(rgb_map), _stream()))
    return t, w, depth, rgb_map


# ---- module-level entry points used by nerf_amd.core ------------------------------------------
def segment_lengths(t: Tensor, rd: Tensor) -> Tensor:
    t, rd = _dev(t, "t"), _dev(rd, "rd")
    delta = torch.empty_like(t)
    with torch.cuda.device(t.device):
        check(lib.nerfx_segment_lengths(_p(t), _p(rd), t.shape[0], t.shape[1], _p(delta), _stream()))
    return delta


def _wants_grad(module) -> bool:
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


class _FieldFunction(torch.autograd.Function):
    """The radiance field with its HIP backward: forward records the tape in a workspace,
    backward turns (d sigma, d rgb) into one gradient per Linear parameter.  Positions,
    directions and t carry no gradient (inputs of the reference's training step)."""

    @staticmethod
    def forward(ctx, handle, prec, p0, p1, t, *params):
        dev = p0.device
        if t is None:
            M, shape_s, shape_c = p0.shape[0], (p0.shape[0],), (p0.shape[0], 3)
        else:
            M, shape_s, shape_c = t.numel(), tuple(t.shape), (*t.shape, 3)
        sigma = torch.empty(shape_s, device=dev)
        rgb = torch.empty(shape_c, device=dev)
        handle.ready(prec)
        ws_bytes = lib.nerfx_mlp_train_workspace_bytes(handle.ptr, prec, M)
        if ws_bytes < 0:
            raise _lib.NerfxError("nerfx_mlp_train_workspace_bytes rejected its arguments")
        ws = torch.empty((max(ws_bytes, 1),), device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            if t is None:
                check(lib.nerfx_mlp_fwd_train_points(handle.ptr, prec, _p(p0), _p(p1), M, _p(sigma), _p(rgb),
                                                     _p(ws), ws_bytes, _stream()))
            else:
                check(lib.nerfx_mlp_fwd_train_rays(handle.ptr, prec, _p(p0), _p(p1), _p(t), t.shape[0], t.shape[1],
                                                   _p(sigma), _p(rgb), _p(ws), ws_bytes, _stream()))
        ctx.handle, ctx.prec, ctx.M, ctx.ws, ctx.ws_bytes = handle, prec, M, ws, ws_bytes
        ctx.param_meta = [(tuple(p.shape), p.requires_grad) for p in params]
        ctx.save_for_backward(sigma, rgb)
        return sigma, rgb

    @staticmethod
    def backward(ctx, g_sigma, g_rgb):
        sigma, rgb = ctx.saved_tensors
        dev = sigma.device
        g_sigma = torch.zeros_like(sigma) if g_sigma is None else _dev(g_sigma, "g_sigma")
        g_rgb = torch.zeros_like(rgb) if g_rgb is None else _dev(g_rgb, "g_rgb")
        # one flat buffer, one view per parameter in module.parameters() order: the optimizer and the
        # data-parallel all-reduce can then work on the whole gradient at once (distributed.flat_gradient)
        sizes = [int(torch.Size(shape).numel()) if need else 0 for shape, need in ctx.param_meta]
        flat = torch.empty((sum(sizes),), device=dev)
        grads = [v.view(shape) if need else None
                 for v, (shape, need) in zip(flat.split(sizes), ctx.param_meta)]
        slots = [None, None] + grads                                 # the two encoder tables come first
        arr = (C.c_void_p * len(slots))(*[None if g is None else g.data_ptr() for g in slots])
        if ctx.M > 0:
            with torch.cuda.device(dev):
                check(lib.nerfx_mlp_bwd_checked(ctx.handle.ptr, ctx.prec, ctx.M, _p(sigma), _p(rgb), _p(g_sigma), _p(g_rgb),
                                                _p(ctx.ws), ctx.ws_bytes, arr, len(slots), _p(GRAD_CHECK), _stream()))
        else:
            grads = [None if g is None else g.zero_() for g in grads]
        ctx.ws = None
        return (None, None, None, None, None, *grads)


class _CompositeFunction(torch.autograd.Function):
    """raymarch_volume + depth (reference renderer.py:50-57, :203) with the K3b backward."""

    @staticmethod
    def forward(ctx, sigma, rgb, t, rd):
        w, rgb_map, depth = composite_fwd(sigma, rgb, t, rd)
        ctx.save_for_backward(sigma, rgb, t, rd)
        return w, rgb_map, depth

    @staticmethod
    def backward(ctx, g_w, g_rgb_map, g_depth):
        sigma, rgb, t, rd = ctx.saved_tensors
        g_sigma, g_rgb = composite_bwd(sigma, rgb, t, rd, None, g_rgb_map, g_depth, g_w)
        return g_sigma, g_rgb, None, None


# While a tensor (one device float, zeroed by mse_seed) is set here, the backward passes below raise it
# to 1 when a parameter gradient they write is not finite: the GradScaler's inf check, made by the
# kernel that writes the gradients instead of by a pass of its own (see nerf_amd.amp).
GRAD_CHECK = None


class _RenderTrainFunction(torch.autograd.Function):
    """render_volume(train=True) as two C calls: nerfx_render_train_fwd records the last pass's tape,
    nerfx_render_bwd turns (d weights, d depth, d rgb_map) into the parameter gradients of the
    model that rendered the last pass.  t, ro, rd carry no gradient (reference renderer.py:133-135)."""

    @staticmethod
    def forward(ctx, hc, hl, prec, ro, rd, tn, tf, nc, nf, u_c, u_f, rng, *params):
        B, n, dev = ro.shape[0], nc + nf, ro.device
        t = torch.empty((B, n), device=dev)
        w = torch.empty((B, n), device=dev)
        depth = torch.empty((B,), device=dev)
        rgb_map = torch.empty((B, 3), device=dev)
        pc, pl = hc.ready(prec), hl.ready(prec)
        ws_bytes = lib.nerfx_render_train_workspace_bytes(pl, prec, B, nc, nf)
        if ws_bytes < 0:
            raise _lib.NerfxError("nerfx_render_train_workspace_bytes rejected its arguments")
        ws = torch.empty((max(ws_bytes, 1),), device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            check(lib.nerfx_render_train_fwd(pc, pl if nf > 0 else None, prec, _p(ro), _p(rd), B, tn, tf, nc, nf, _p(u_c),
                                             _p(u_f), None if rng is None else C.byref(rng), _p(ws), ws_bytes,
                                             _p(t), _p(w), _p(depth), _p(rgb_map), _stream()))
        ctx.handle, ctx.prec, ctx.ws, ctx.ws_bytes, ctx.shape = hl, prec, ws, ws_bytes, (B, nc, nf)
        ctx.param_meta = [(tuple(p.shape), p.requires_grad) for p in params]
        ctx.save_for_backward(rd, t)
        ctx.mark_non_differentiable(t)
        ctx.set_materialize_grads(False)         # outputs the loss does not use arrive as None, not as zero fills
        return t, w, depth, rgb_map

    @staticmethod
    def backward(ctx, g_t, g_w, g_depth, g_rgb_map):
        rd, t = ctx.saved_tensors
        B, nc, nf = ctx.shape
        dev = rd.device
        g_w = None if g_w is None else _dev(g_w, "g_weights")
        g_depth = None if g_depth is None else _dev(g_depth, "g_depth")
        g_rgb_map = None if g_rgb_map is None else _dev(g_rgb_map, "g_rgb_map")
        # one flat buffer, one view per parameter in module.parameters() order (distributed.flat_gradient)
        sizes = [int(torch.Size(shape).numel()) if need else 0 for shape, need in ctx.param_meta]
        flat = torch.empty((sum(sizes),), device=dev)
        grads = [v.view(shape) if need else None for v, (shape, need) in zip(flat.split(sizes), ctx.param_meta)]
        slots = [None, None] + grads                                 # the two encoder tables come first
        arr = (C.c_void_p * len(slots))(*[None if g is None else g.data_ptr() for g in slots])
        if B > 0:
            with torch.cuda.device(dev):
                check(lib.nerfx_render_bwd_checked(ctx.handle.ptr, ctx.prec, _p(rd), B, nc, nf, _p(t), _p(g_rgb_map),
                                                   _p(g_depth), _p(g_w), _p(ctx.ws), ctx.ws_bytes, arr, len(slots),
                                                   _p(GRAD_CHECK), _stream()))
        else:
            grads = [None if g is None else g.zero_() for g in grads]
        ctx.ws = None
        return (None,) * 12 + tuple(grads)


def _field_with_grad(module, p0, p1, t):
    handle = handle_for(module)
    params = [p for p in handle._sources[2:]]
    return _FieldFunction.apply(handle, precision(), p0, p1, t, *params)


def field_forward(module, rx: Tensor, rd: Tensor):
    """NeRF.forward: sigma (M,), rgb (..., 3)."""
    if _wants_grad(module):
        x, d = _dev(rx, "rx").reshape(-1, 3), _dev(rd, "rd").reshape(-1, 3)
        assert x.shape == d.shape, "positions and directions must pair up"
        sigma, rgb = _field_with_grad(module, x, d, None)
    else:
        sigma, rgb = mlp_points(handle_for(module), rx, rd)
    return sigma, rgb.reshape(*rx.shape[:-1], 3)


def render_train(coarse, fine, ro, rd, tn, tf, nc, nf, u_c, u_f, rng=None):
    """render_volume(train=True): the coarse pass and the sample positions are graph-free
    (reference renderer.py:133-135); gradients reach `fine` (the module of the last pass: `coarse`
    itself when nf == 0) through its MLP and the compositor.  Jitter as in `render`."""
    ro, rd = _dev(ro, "ro"), _dev(rd, "rd")
    u_c = None if u_c is None else _dev(u_c, "u_c")
    u_f = None if u_f is None else _dev(u_f, "u_f")
    if nf > 0:
        fine.requires_grad(True)                                    # reference renderer.py:137
    hc, hl = handle_for(coarse), handle_for(fine)
    if _wants_grad(fine):
        params = list(hl._sources[2:])
        return _RenderTrainFunction.apply(hc, hl, precision(), ro, rd, tn, tf, nc, nf, u_c, u_f, rng, *params)
    with torch.no_grad():
        return render(hc, hl if nf > 0 else None, ro, rd, tn, tf, nc, nf, u_c, u_f, rng=rng)


# ---- camera rays in, display frames out ---------------------------------------------
def _device(device) -> torch.device:
    d = torch.device("cuda" if device is None else device)
    if d.type != "cuda":
        raise RuntimeError(f"nerf_amd runs on ROCm devices only: asked for {d}")
    return torch.device("cuda", torch.cuda.current_device()) if d.index is None else d


def orbit_poses(theta, phi, psy, radius, samples: int, device=None) -> Tensor:
    """(samples, 4, 4) camera-to-world turn-around poses; each range is a (start, end) pair."""
    dev = _device(device)
    poses = torch.empty((samples, 4, 4), device=dev)
    with torch.cuda.device(dev):
        check(lib.nerfx_orbit_poses(theta[0], theta[1], phi[0], phi[1], psy[0], psy[1], radius[0], radius[1],
                                    samples, _p(poses), _stream()))
    return poses


def pinhole_directions(H: int, W: int, focal: float, device=None) -> Tensor:
    dev = _device(device)
    prd = torch.empty((H, W, 3), device=dev)
    with torch.cuda.device(dev):
        check(lib.nerfx_pinhole_directions(H, W, focal, _p(prd), _stream()))
    return prd


def project_rays(prd: Tensor, c2w: Tensor) -> Tuple[Tensor, Tensor]:
    prd, c2w = _dev(prd, "prd"), _dev(c2w, "c2w")
    assert prd.shape[-1] == 3 and c2w.shape == (4, 4), "prd (..., 3) and c2w (4, 4)"
    ro, rd = torch.empty_like(prd), torch.empty_like(prd)
    with torch.cuda.device(prd.device):
        check(lib.nerfx_project_rays(_p(prd), prd.numel() // 3, _p(c2w), _p(ro), _p(rd), _stream()))
    return ro, rd


def camera_rays(c2w: Tensor, H: int, W: int, focal: float, begin: int = 0,
                count: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """World-space rays of pixels [begin, begin + count) of an H x W pinhole frame seen from `c2w`."""
    c2w = _dev(c2w, "c2w")
    assert c2w.shape == (4, 4), "c2w (4, 4)"
    count = H * W - begin if count is None else count
    ro = torch.empty((count, 3), device=c2w.device)
    rd = torch.empty((count, 3), device=c2w.device)
    with torch.cuda.device(c2w.device):
        check(lib.nerfx_camera_rays(_p(c2w), H, W, focal, begin, count, _p(ro), _p(rd), _stream()))
    return ro, rd


def frame_finish(depth: Tensor, rgb: Tensor, as_float: bool = True, as_uint8: bool = False):
    """Per-pixel depth (n) and rgb (n, 3) -> display frames (n, 3) in [0, 255]:
    (depth_f32, rgb_f32, depth_u8, rgb_u8), None where not asked for."""
    depth, rgb = _dev(depth, "depth").reshape(-1), _dev(rgb, "rgb").reshape(-1, 3)
    n = depth.numel()
    assert rgb.shape[0] == n, "depth and rgb disagree on the pixel count"
    dev = depth.device
    ws = torch.empty((lib.nerfx_frame_workspace_bytes(),), dtype=torch.uint8, device=dev)
    df = torch.empty((n, 3), device=dev) if as_float else None
    cf = torch.empty((n, 3), device=dev) if as_float else None
    du = torch.empty((n, 3), dtype=torch.uint8, device=dev) if as_uint8 else None
    cu = torch.empty((n, 3), dtype=torch.uint8, device=dev) if as_uint8 else None
    with torch.cuda.device(dev):
        check(lib.nerfx_frame_finish(_p(depth), _p(rgb), n, _p(ws), ws.numel(), _p(df), _p(cf), _p(du), _p(cu),
                                     _stream()))
    return df, cf, du, cu


# ---- around the render in a training step -------------------------------------------------
def gather_rays(C: Tensor, ro: Tensor, rd: Tensor, perm: Optional[Tensor], begin: int, n: int):
    C, ro, rd = _dev(C, "C"), _dev(ro, "ro"), _dev(rd, "rd")
    assert C.shape == ro.shape == rd.shape and C.shape[-1] == 3, "C, ro, rd must all be (N, 3)"
    perm = None if perm is None else _dev(perm, "perm", torch.int64)
    out = [torch.empty((n, 3), device=C.device) for _ in range(3)]
    with torch.cuda.device(C.device):
        check(lib.nerfx_gather_rays(_p(C), _p(ro), _p(rd), _p(perm), begin, n, C.shape[0],
                                    _p(out[0]), _p(out[1]), _p(out[2]), _stream()))
    return out


class _MSEFunction(torch.autograd.Function):
    """mean((pred - target)^2) with the PSNR on the side and d loss / d pred formed in the same pass."""

    @staticmethod
    def forward(ctx, pred, target, loss_psnr):
        seed = torch.empty_like(pred)
        with torch.cuda.device(pred.device):
            check(lib.nerfx_mse_loss(_p(pred), _p(target), pred.numel(), _p(loss_psnr), _p(seed), _stream()))
        ctx.save_for_backward(seed)
        return loss_psnr[0]

    @staticmethod
    def backward(ctx, g):
        (seed,) = ctx.saved_tensors
        return seed * g, None, None


def mse_loss(pred: Tensor, target: Tensor, loss_psnr: Optional[Tensor] = None) -> Tensor:
    """MSELoss (mean) as a 0-dim tensor; `loss_psnr` (2 floats on the device, optional) receives
    the loss and -10 log10(loss)."""
    pred, target = _dev(pred, "pred"), _dev(target.detach(), "target")
    assert pred.shape == target.shape, "pred and target shapes differ"
    if loss_psnr is None:
        loss_psnr = torch.empty((2,), device=pred.device)
    if pred.requires_grad and torch.is_grad_enabled():
        return _MSEFunction.apply(pred, target, loss_psnr)
    with torch.cuda.device(pred.device):
        check(lib.nerfx_mse_loss(_p(pred), _p(target), pred.numel(), _p(loss_psnr), None, _stream()))
    return loss_psnr[0]


def mse_seed(pred: Tensor, target: Tensor, loss_psnr: Tensor, loss_scale: Optional[Tensor] = None,
             found_inf: Optional[Tensor] = None) -> Tensor:
    """d (loss_scale * mse(pred, target)) / d pred, with the unscaled loss and its PSNR written to
    `loss_psnr` (2 device floats) and `found_inf` (one device float) zeroed, all in one kernel: the part
    of a step between the render and the backward (reference train.py:77-80) behind a GradScaler."""
    pred, target = _dev(pred.detach(), "pred"), _dev(target.detach(), "target")
    assert pred.shape == target.shape, "pred and target shapes differ"
    seed = torch.empty_like(pred)
    with torch.cuda.device(pred.device):
        check(lib.nerfx_mse_loss_scaled(_p(pred), _p(target), pred.numel(), _p(loss_psnr), _p(seed), _p(loss_scale),
                                        _p(found_inf), _stream()))
    return seed


def adam_step(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, lr: float, beta1: float,
              beta2: float, eps: float, weight_decay: float, inv_scale: Optional[Tensor],
              found_inf: Optional[Tensor], step_in: Tensor, step_out: Tensor,
              grad_scale: Optional[Tensor] = None) -> None:
    """One Adam update over flat buffers; gradients are multiplied by `inv_scale`, or divided by
    `grad_scale` (the loss scale itself: no reciprocal kernel in front), when one is given."""
    n = param.numel()
    assert inv_scale is None or grad_scale is None, "give the inverse scale or the scale, not both"
    for name, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n):
            raise RuntimeError(f"adam_step: `{name}` must be a contiguous fp32 ROCm tensor of {n} elements")
    for name, t in (("inv_scale", inv_scale), ("grad_scale", grad_scale), ("found_inf", found_inf), ("step_in", step_in),
                    ("step_out", step_out)):
        if t is not None and not (t.is_cuda and t.device == param.device and t.dtype == torch.float32 and t.numel() == 1):
            raise RuntimeError(f"adam_step: `{name}` must be one fp32 element on {param.device}")
    with torch.cuda.device(param.device):
        if grad_scale is not None:
            check(lib.nerfx_adam_step_scaled(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, lr, beta1, beta2, eps,
                                             weight_decay, _p(grad_scale), _p(found_inf), _p(step_in), _p(step_out),
                                             _stream()))
        else:
            check(lib.nerfx_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, lr, beta1, beta2, eps,
                                      weight_decay, _p(inv_scale), _p(found_inf), _p(step_in), _p(step_out), _stream()))


def voxel_sample(domain, voxels: int, samples: int, device, rng=None):
    """One distillation batch drawn by the sampling kernel: rx, rd (voxels * samples, 3), delta (voxels * samples)."""
    dev = _device(device)
    rng = next_rng(dev) if rng is None else rng
    n = voxels * samples
    rx, rd, delta = torch.empty((n, 3), device=dev), torch.empty((n, 3), device=dev), torch.empty((n,), device=dev)
    with torch.cuda.device(dev):
        check(lib.nerfx_voxel_sample(domain.x[0], domain.x[1], domain.y[0], domain.y[1], domain.z[0], domain.z[1],
                                     domain.d[0], domain.d[1], voxels, samples, rng.seed, rng.offset,
                                     _p(rx), _p(rd), _p(delta), _stream()))
    return rx, rd, delta


class _DistillLossFunction(torch.autograd.Function):
    """The distillation loss with its gradient w.r.t. the student's outputs formed in the forward pass."""

    @staticmethod
    def forward(ctx, s_sigma, s_rgb, t_sigma, t_rgb, delta):
        n = s_sigma.numel()
        dev = s_sigma.device
        ws, loss = torch.empty((n,), device=dev), torch.empty((1,), device=dev)
        g_sigma, g_rgb = torch.empty_like(s_sigma), torch.empty_like(s_rgb)
        with torch.cuda.device(dev):
            check(lib.nerfx_distill_loss(_p(t_sigma), _p(t_rgb), _p(s_sigma), _p(s_rgb), _p(delta), n, _p(ws), _p(loss),
                                         _p(g_sigma), _p(g_rgb), _stream()))
        ctx.save_for_backward(g_sigma, g_rgb)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        g_sigma, g_rgb = ctx.saved_tensors
        return g_sigma * g, g_rgb * g, None, None, None


def distill_loss(s_sigma: Tensor, s_rgb: Tensor, t_sigma: Tensor, t_rgb: Tensor, delta: Tensor) -> Tensor:
    """Loss of reference distill.py:123-130 (0-dim tensor; differentiable w.r.t. the student's outputs)."""
    s_sigma, t_sigma, delta = (_dev(x, n).reshape(-1) for x, n in ((s_sigma, "s_sigma"), (t_sigma.detach(), "t_sigma"),
                                                                  (delta.detach(), "delta")))
    s_rgb, t_rgb = _dev(s_rgb, "s_rgb").reshape(-1, 3), _dev(t_rgb.detach(), "t_rgb").reshape(-1, 3)
    n = s_sigma.numel()
    assert t_sigma.numel() == n and delta.numel() == n and s_rgb.shape[0] == n and t_rgb.shape[0] == n, \
        "sigma, delta (n) and rgb (n, 3) must agree on n"
    return _DistillLossFunction.apply(s_sigma, s_rgb, t_sigma, t_rgb, delta)
