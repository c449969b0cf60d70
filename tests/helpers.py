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
