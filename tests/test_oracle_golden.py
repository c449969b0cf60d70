"""Pins the CPU oracle against the golden vectors captured from the real reference
(`tools/make_golden.py`).  Everything here runs on CPU.

Tolerances: the oracle uses the same torch CPU ops as the reference, so nearly
everything is bit-identical; where the restatement re-associates (cat/where
instead of in-place writes) a few-ulp slack is allowed.
"""
import pytest
import torch

from conftest import golden_names, load_golden
from helpers import assert_close, oracle_model
from oracle import nerf_oracle as O


def test_g1_sampling():
    meta, g = load_golden("g1_sampling")
    ro, rd = g["ro"], g["rd"]
    for tag, u in (("det", None), ("pert", g["pert_u"])):
        rx, rds, t, delta = O.uniform_rays(ro, rd, meta["tn"], meta["tf"], meta["n"], u)
        assert torch.equal(t, g[f"{tag}_t"])
        assert torch.equal(delta, g[f"{tag}_delta"])
        assert torch.equal(rx, g[f"{tag}_rx"])
        assert torch.equal(rds, g[f"{tag}_rds"])
    a = meta["alt"]
    rx, _, t, delta = O.uniform_rays(ro, rd, a["tn"], a["tf"], a["n"], None)
    assert torch.equal(t, g["alt_t"]) and torch.equal(delta, g["alt_delta"]) and torch.equal(rx, g["alt_rx"])
    # documented quirks: inclusive end points, last delta = 1e10 * |rd|
    assert t[0, 0] == a["tn"] and t[0, -1] == a["tf"]
    assert_close(g["det_delta"][:, -1], 1e10 * rd.norm(dim=-1), rtol=1e-6)


def test_g2_encodings_and_activations():
    _, g = load_golden("g2_encodings")
    for L in (4, 6, 10):
        f = O.positional_freq_bands(L)
        assert torch.equal(f, g[f"pe{L}_freq"])
        for which in ("unit", "far"):
            out = O.encode_positional(g[f"v_{which}"], f)
            assert out.shape[-1] == 3 + 6 * L
            assert torch.equal(out, g[f"pe{L}_{which}"])
    for which in ("unit", "far"):
        assert torch.equal(O.encode_fourier(g[f"v_{which}"], g["ff_b"]), g[f"ff_{which}"])
    assert torch.equal(O.shifted_softplus(g["act_x"]), g["act_softplus"])
    assert torch.equal(O.widened_sigmoid(g["act_x"]), g["act_wsigmoid"])


@pytest.mark.parametrize("name", golden_names("g3_mlp_"))
def test_g3_mlp(name):
    meta, g = load_golden(name)
    model = oracle_model(meta, g)
    n_params = sum(v.numel() for v in model.params.values())
    if meta["name"] == "base":
        assert n_params == 321_924
    if meta["name"] == "student":
        assert n_params == 4_628
    sigma, rgb = model(g["rx"], g["rd"])
    assert_close(sigma, g["sigma"], rtol=1e-6, atol=1e-7, what="sigma")
    assert_close(rgb, g["rgb"], rtol=1e-6, atol=1e-7, what="rgb")
    s2, none = model(g["rx"], g["rd"], density_only=True)
    assert none is None and torch.equal(s2, sigma)


def test_g4_composite():
    _, g = load_golden("g4_composite")
    assert torch.equal(O.exclusive_cumprod(torch.tensor([[2., 3., 4.]])), g["ecp"])
    delta = O.segment_lengths(g["t"], g["rd"])
    assert torch.equal(delta, g["delta"])
    w, rgb_map = O.raymarch_volume(g["sigma"], g["rgb"], delta)
    assert torch.equal(w, g["weights"]) and torch.equal(rgb_map, g["rgb_map"])
    assert torch.equal(torch.sum(w * g["t"], -1), g["depth"])
    w65, rgb65 = O.raymarch_volume(g["sigma"][:, :65], g["rgb"][:, :65],
                                   O.segment_lengths(g["t"][:, :65], g["rd"]))
    assert torch.equal(w65, g["w65"]) and torch.equal(rgb65, g["rgb65"])
    # the last sample absorbs the remaining transmittance: weights sum to ~1
    # (row 0 has sigma == 0 everywhere, so nothing is absorbed at all)
    assert float(w[0].abs().max()) == 0.0
    assert_close(w[1:].sum(-1), torch.ones(w.size(0) - 1), rtol=0, atol=1e-5)


def test_g5_pdf():
    meta, g = load_golden("g5_pdf")
    cdf = O.pdf_cdf(g["w"][:, 1:-1])
    assert torch.equal(cdf, g["cdf"])
    for tag in ("det", "pert"):
        u = g[f"{tag}_u"]
        idx = O.searchsorted_right(g["cdf"], u)
        assert torch.equal(idx, g[f"{tag}_idx"])                 # bit-exact indices
        assert int(idx.min()) >= 1 and int(idx.max()) <= 63
        c2, u2, idx2, t_pdf = O.pdf_z_values(g["bins"], g["w"][:, 1:-1], meta["nf"],
                                             None if tag == "det" else u)
        assert torch.equal(u2, u) and torch.equal(idx2, idx)
        assert torch.equal(t_pdf, g[f"{tag}_t_pdf"])
        rx, _, ts, delta = O.pdf_rays(g["ro"], g["rd"], g["t"], g["w"], meta["nf"],
                                      None if tag == "det" else u)
        assert torch.equal(ts, g[f"{tag}_t_sorted"])
        assert torch.equal(delta, g[f"{tag}_delta"]) and torch.equal(rx, g[f"{tag}_rx"])
    assert torch.equal(O.searchsorted_right(g["ss_cdf"], g["ss_u"]), g["ss_idx"])


@pytest.mark.parametrize("name", golden_names("g6_render_"))
def test_g6_render(name):
    meta, g = load_golden(name)
    coarse = oracle_model(meta["coarse"], g, "c_")
    fine = oracle_model(meta["fine"], g, "f_")
    for tag in ("det", "pert"):
        u_c = g["pert_u_c"] if tag == "pert" else None
        u_f = g.get("pert_u_f") if tag == "pert" else None
        with torch.no_grad():
            t, w, depth, rgb = O.render_volume(coarse, fine, g["ro"], g["rd"], meta["tn"], meta["tf"],
                                               meta["nc"], meta["nf"], u_c, u_f)
        assert_close(t, g[f"{tag}_t"], rtol=1e-6, atol=1e-6, what="t")
        assert_close(w, g[f"{tag}_w"], rtol=1e-5, atol=1e-7, what="weights")
        assert_close(depth, g[f"{tag}_depth"], rtol=1e-6, atol=1e-6, what="depth")
        assert_close(rgb, g[f"{tag}_rgb"], rtol=1e-6, atol=1e-6, what="rgb")


@pytest.mark.parametrize("name", golden_names("g7_train_"))
def test_g7_train(name):
    meta, g = load_golden(name)
    mm = meta["model"]
    model = oracle_model(mm, g, requires_grad=True)
    params = model.params
    optim = torch.optim.Adam(list(params.values()), lr=meta["lr"], eps=meta["eps"])
    for it in range(meta["steps"]):
        loss, grads = O.train_step(params, model.enc_x, model.enc_d, mm["activ"], mm["resid"],
                                   g["ro"], g["rd"], g["target"], meta["tn"], meta["tf"],
                                   meta["nc"], meta["nf"], g[f"u_c{it}"], g[f"u_f{it}"], optim)
        assert abs(loss.item() - g["losses"][it].item()) <= 1e-6 * abs(g["losses"][it].item()) + 1e-9
        if it == 0:
            assert set(grads) == set(params)                      # all 16 tensors get a gradient
            for k in params:
                scale = g["grad0." + k].abs().max().item()
                assert_close(grads[k], g["grad0." + k], rtol=1e-4, atol=1e-5 * scale + 1e-12,
                             what="grad " + k)
        if it in (0, 2):
            for k in params:
                assert_close(params[k].detach().flatten()[::17], g[f"after{it + 1}." + k],
                             rtol=1e-5, atol=2e-6, what=f"after{it + 1} " + k)


def test_g8_infer():
    meta, g = load_golden("g8_infer")
    model = oracle_model(meta["model"], g)
    depth, rgb = O.infer(model, model, g["ro"], g["rd"], meta["H"], meta["W"], meta["tn"], meta["tf"],
                         meta["nc"], meta["nf"], batch_size=100)
    assert depth.shape == (meta["H"], meta["W"], 3) and rgb.shape == (meta["H"], meta["W"], 3)
    assert_close(depth, g["depth"], rtol=1e-5, atol=1e-3, what="depth")
    assert_close(rgb, g["rgb"], rtol=1e-5, atol=1e-3, what="rgb")
    ro_nan = g["ro"].clone()
    ro_nan[5] = float("nan")
    dn, cn = O.infer(model, model, ro_nan, g["rd"], meta["H"], meta["W"], meta["tn"], meta["tf"],
                     meta["nc"], meta["nf"], batch_size=100)
    assert_close(dn, g["nan_depth"], rtol=1e-5, atol=1e-3, what="nan depth")
    assert_close(cn, g["nan_rgb"], rtol=1e-5, atol=1e-3, what="nan rgb")


def test_g10_camera_rays_and_poses():
    """Camera rays and turn-around poses as the reference's CLI builds them (infer.py:110-121).
    The pixel-grid convention inside pinhole_ray_directions comes from kornia (absent): that
    one call is parity unpinned, everything downstream of it is the reference's own code."""
    import numpy as np
    meta, g = load_golden("g10_camera")
    ranges = (-np.pi, np.pi), (-np.pi, np.pi), (-np.pi, np.pi), (4., 4.)
    poses = O.turnaround_poses(*ranges, 8)
    assert_close(poses, g["poses8"], rtol=0, atol=1e-6, what="poses8")
    alt = O.turnaround_poses((0.25, 1.5), (-0.5, 0.75), (0., 0.), (3., 5.), 5)
    assert_close(alt, g["poses5_alt"], rtol=0, atol=1e-6, what="poses5_alt")
    for H, W in meta["sizes"]:
        focal = .5 * W / np.tan(.5 * meta["cam_angle_x"])
        prd = O.pinhole_ray_directions(H, W, focal)
        assert prd.shape == (1, H, W, 3)
        assert torch.equal(prd, g[f"prd_{H}x{W}"])
        for k in meta["pose_ids"]:
            ro, rd = O.pinhole_ray_projection(prd, g["poses8"][k])
            assert torch.equal(ro, g[f"ro_{H}x{W}_{k}"]) and torch.equal(rd, g[f"rd_{H}x{W}_{k}"])
            assert_close(rd.norm(dim=-1), torch.ones(1, H, W), rtol=1e-6, atol=0, what="unit directions")


def test_frame_to_uint8_truncates():
    _, g = load_golden("g8_infer")
    u8 = O.frame_to_uint8(g["depth"])
    assert u8.dtype.name == "uint8" and u8.max() <= 255
    assert (u8 == g["depth"].floor().to(torch.uint8).numpy()).all()


def test_g11_distill_three_iterations():
    """The oracle's distillation step replays the reference's `distill` (distill.py:85-147)."""
    meta, g = load_golden("g11_distill")
    teacher = oracle_model(meta["teacher"], g)
    student = oracle_model(meta["student"], g, requires_grad=True)
    optim = torch.optim.Adam(list(student.params.values()), lr=meta["lr"], eps=meta["eps"])
    for k in range(meta["iters"]):
        loss, _ = O.distill_step(teacher, student, g[f"rx{k}"], g[f"rd{k}"], g[f"delta{k}"], optim)
        assert_close(loss.double(), g["losses"][k], rtol=1e-5, atol=0, what=f"loss {k}")
    for name, p in student.params.items():
        assert_close(p.detach(), g["after." + name], rtol=1e-4, atol=1e-6, what=name)
    # sampler fixture: grid = cartesian product of linspaces, lengths inside the domain
    assert g["vs_grid"].shape == (64, 3) and g["vs_rx"].shape == (32, 3) and g["vs_delta"].shape == (32,)
    assert_close(g["vs_rd"].norm(dim=-1), torch.ones(32), rtol=1e-6, atol=0, what="unit directions")
