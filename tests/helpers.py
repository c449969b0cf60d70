"""Shared test helpers: rebuild a fixture's model for the oracle and for the product."""
import torch

from oracle import nerf_oracle as O


def oracle_encoder(spec, b=None):
    if spec[0] == "pe":
        return O.make_encoder("pe", O.positional_freq_bands(int(spec[1])))
    return O.make_encoder("ff", b)


def oracle_model(meta, arrays, prefix="", requires_grad=False):
    """meta = the `model` dict a fixture stores; arrays may hold `<prefix>b_x` / `<prefix>b_d`."""
    params = O.synth_params(meta["e_x"], meta["e_d"], meta["width"], meta["depth"],
                            meta["resid"], meta["seed"])
    if requires_grad:
        params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    enc_x = oracle_encoder(meta["enc_x"], arrays.get(prefix + "b_x"))
    enc_d = oracle_encoder(meta["enc_d"], arrays.get(prefix + "b_d"))
    return O.Model(params, enc_x, enc_d, meta["activ"], meta["resid"])


def assert_close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    if not torch.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True):
        err = (a.double() - b.double()).abs()
        rel = err / b.double().abs().clamp_min(1e-30)
        i = int(torch.argmax(err))
        raise AssertionError(f"{what}: max abs err {err.max():.3e} (rel {rel.flatten()[i]:.3e}) "
                             f"at flat index {i}: {a.flatten()[i]:.8g} vs {b.flatten()[i]:.8g}; "
                             f"rtol={rtol} atol={atol}")


# ---- product-side helpers (need the HIP library; GPU tests only) -----------------------
def product_model(meta, arrays, prefix="", device="cuda"):
    """The nerf_amd.core.NeRF equal to oracle_model(meta, arrays, prefix)."""
    from nerf_amd.core import FourierFeatures, NeRF, PositionalEncoding
    acts = {"relu": torch.nn.ReLU, "leaky_relu": torch.nn.LeakyReLU, "silu": torch.nn.SiLU}

    def enc(spec, b):
        if spec[0] == "pe":
            return PositionalEncoding(3, int(spec[1]))
        ff = FourierFeatures(3, int(spec[1]), float(spec[2]))
        ff.b.copy_(b)
        return ff

    m = NeRF(enc(meta["enc_x"], arrays.get(prefix + "b_x")), enc(meta["enc_d"], arrays.get(prefix + "b_d")),
             meta["width"], meta["depth"], acts[meta["activ"]], meta["resid"])
    sd = O.synth_params(meta["e_x"], meta["e_d"], meta["width"], meta["depth"], meta["resid"], meta["seed"])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("phi_") for k in missing)
    return m.to(device)


# ---- a scriptable stand-in for a model file written by the reference (test infrastructure) ----
class _ScriptableEncoder(torch.nn.Module):
    def __init__(self, freq_bands):
        super().__init__()
        self.register_buffer("freq_bands", freq_bands)

    def forward(self, v):
        return O.encode_positional(v, self.freq_bands)


class ScriptableField(torch.nn.Module):
    """Submodule names, parameter shapes and forward of the reference's NeRF (model.py:39-78) in plain
    torch ops, so that `torch.jit.script` accepts it: what a `*.model.ts` of the reference holds."""

    def __init__(self, n_x: int, n_d: int, width: int, depth: int, resid: bool):
        super().__init__()
        from torch.nn import Linear, ReLU, Sequential
        self.phi_x = _ScriptableEncoder(O.positional_freq_bands(n_x))
        self.phi_d = _ScriptableEncoder(O.positional_freq_bands(n_d))
        e_x, e_d, half = 3 + 6 * n_x, 3 + 6 * n_d, depth // 2
        self.resid = resid
        blk = lambda i, o: Sequential(Linear(i, o), ReLU())
        self.fc_1 = Sequential(blk(e_x, width), *(blk(width, width) for _ in range(half - 1)))
        self.fc_2 = Sequential(blk(width + (e_x if resid else 0), width), *(blk(width, width) for _ in range(half - 1)))
        self.sigma = Sequential(Linear(width, 1))
        self.features = Linear(width, width)
        self.rgb = Sequential(Linear(e_d + width, width // 2), ReLU(), Linear(width // 2, 3))

    def forward(self, rx, rd):
        px, pd = self.phi_x(rx), self.phi_d(rd)
        x = self.fc_1(px)
        if self.resid:
            x = torch.cat((px, x), dim=-1)
        x = self.fc_2(x)
        sigma = O.shifted_softplus(self.sigma(x)).view(-1)
        rgb = O.widened_sigmoid(self.rgb(torch.cat((pd, self.features(x)), dim=-1)))
        return sigma, rgb


# ---- float64 truth (decides how far fp32 results may sit from a reference fixture) ---------
class float64_default:
    """Context: torch's default dtype is float64, so the oracle's linspace / full / constants are."""

    def __enter__(self):
        self.prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)

    def __exit__(self, *exc):
        torch.set_default_dtype(self.prev)


def oracle_model_f64(meta, arrays, prefix=""):
    """oracle_model with every parameter, encoder table and constant in float64: the same function
    of the same fp32 inputs, evaluated ~1e9 times more finely than either fp32 implementation."""
    params = {k: v.double() for k, v in O.synth_params(meta["e_x"], meta["e_d"], meta["width"], meta["depth"],
                                                       meta["resid"], meta["seed"]).items()}

    def enc(spec, b):
        if spec[0] == "pe":
            with float64_default():
                bands = O.positional_freq_bands(int(spec[1]))
            # the reference's bands are fp32 numbers: keep those, not float64's own 2 ** linspace
            bands = O.positional_freq_bands(int(spec[1])).double()
            return O.make_encoder("pe", bands)
        return O.make_encoder("ff", b.double())

    return O.Model(params, enc(meta["enc_x"], arrays.get(prefix + "b_x")), enc(meta["enc_d"], arrays.get(prefix + "b_d")),
                   meta["activ"], meta["resid"])


def assert_no_farther_than_reference(got, fixture, truth, what, factor=2.0, floor=0.0):
    """max |got - truth| <= factor * max |fixture - truth| (+ floor): the HIP result may sit at
    most `factor` times as far from the float64 evaluation as the reference's own fp32 result.
    Returns (our error, the reference's error)."""
    got, fixture, truth = (torch.as_tensor(x).double() for x in (got, fixture, truth))
    assert got.shape == truth.shape == fixture.shape, f"{what}: shapes {got.shape} {fixture.shape} {truth.shape}"
    ours = float((got - truth).abs().max())
    theirs = float((fixture - truth).abs().max())
    assert ours <= factor * theirs + floor, (f"{what}: max |hip - f64| = {ours:.3e} but the reference's own fp32 result is "
                                            f"only {theirs:.3e} from f64 (allowed {factor:g}x + {floor:g})")
    return ours, theirs
