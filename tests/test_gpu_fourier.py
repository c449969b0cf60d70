"""The Fourier-feature variant in fp32, judged against a float64 evaluation of the same function.

Fourier angles 2 pi v . b reach ~1e3 rad (sigma 16..32, |v| up to ~7): one fp32 ulp of the angle
is ~6e-5, so ANY fp32 implementation -- the reference on torch CPU included -- sits 1e-4..1e-3
from the exact encoding, and the trunk amplifies that.  The 1e-5 bar of the positional path is
therefore not attainable for this encoder by the reference itself.  What is checked instead:
the HIP result is at most twice as far from the float64 truth as the reference's own fixture is
(max norm), for the encodings, the field and the rendered rays; and the direct HIP-vs-fixture
tolerances used elsewhere for this variant are derived from the same float64 distances.
"""
import pytest
import torch

from conftest import load_golden
from helpers import (assert_close, assert_no_farther_than_reference, float64_default, oracle_model_f64,
                     product_model)
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    import nerf_amd.ops as ops_
    ops_.set_precision(None)
    yield ops_
    ops_.set_precision(None)


def test_fourier_encodings_vs_float64(ops):
    _, g = load_golden("g2_encodings")
    b = g["ff_b"]
    for tag in ("unit", "far"):
        v = g[f"v_{tag}"]
        truth = O.encode_fourier(v.double(), b.double())
        got = ops.encode(v.cuda(), 1, b.shape[0], b.cuda()).cpu()
        ours, theirs = assert_no_farther_than_reference(got, g[f"ff_{tag}"], truth, f"ff {tag}")
        # and directly against the fixture, with the tolerance those two distances allow
        assert_close(got, g[f"ff_{tag}"], rtol=0, atol=3 * theirs, what=f"ff {tag} vs fixture")
        print(f"ff {tag}: |hip - f64| = {ours:.2e}, |reference - f64| = {theirs:.2e}")


def test_fourier_field_vs_float64(ops):
    meta, g = load_golden("g3_mlp_fourier")
    model = product_model(meta, g)
    truth_s, truth_c = oracle_model_f64(meta, g)(g["rx"].double(), g["rd"].double())
    ops.set_precision("fp32")
    with torch.no_grad():
        sigma, rgb = model(g["rx"].cuda(), g["rd"].cuda())
    assert_no_farther_than_reference(sigma.cpu(), g["sigma"], truth_s, "fourier sigma")
    assert_no_farther_than_reference(rgb.cpu(), g["rgb"], truth_c, "fourier rgb")


def test_fourier_render_vs_float64(ops):
    from nerf_amd.core import BoundedVolumeRaymarcher
    name = "g6_render_fourier_fourier_32_32"
    meta, g = load_golden(name)
    model = product_model(meta["coarse"], g, "c_")
    om = oracle_model_f64(meta["coarse"], g, "c_")
    ro, rd = g["ro"], g["rd"]
    nc, nf = meta["nc"], meta["nf"]
    h = ops.handle_for(model)
    for tag, u_c, u_f in (("det", None, None), ("pert", g["pert_u_c"], g["pert_u_f"])):
        with float64_default(), torch.no_grad():
            t64, w64, depth64, rgb64 = O.render_volume(om, om, ro.double(), rd.double(), meta["tn"], meta["tf"], nc, nf,
                                                       None if u_c is None else u_c.double(),
                                                       None if u_f is None else u_f.double())
        t, w, depth, rgb = ops.render(h, h, ro.cuda(), rd.cuda(), meta["tn"], meta["tf"], nc, nf,
                                      None if u_c is None else u_c.cuda(), None if u_f is None else u_f.cuda(), prec=0)
        # (a sample whose cdf bin differs between two evaluations moves continuously, so max norms stay meaningful)
        for key, got, truth in (("t", t, t64), ("depth", depth, depth64), ("rgb", rgb, rgb64), ("w", w, w64)):
            assert_no_farther_than_reference(got.cpu(), g[f"{tag}_{key}"], truth, f"{name} {tag} {key}", floor=1e-6)
