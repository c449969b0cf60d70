"""Diagnostic: per-parameter gradient error of one training step against the golden
fixture and against the CPU oracle evaluated at the kernel's own sample positions."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from conftest import load_golden
from helpers import oracle_model, product_model
from oracle import nerf_oracle as O
import nerf_amd.ops as ops

name = sys.argv[1] if len(sys.argv) > 1 else "g7_train_base"
meta, g = load_golden(name)
ro, rd, target = g["ro"].cuda(), g["rd"].cuda(), g["target"].cuda()
for prec in ("fp32", "bf16"):
    model = product_model(meta["model"], g)
    ops.set_precision(prec)
    t, w, depth, C = ops.render_train(model, model, ro, rd, meta["tn"], meta["tf"], meta["nc"], meta["nf"],
                                      g["u_c0"].cuda(), g["u_f0"].cuda())
    loss = torch.nn.functional.mse_loss(C, target)
    loss.backward()
    # oracle at the same t
    om = oracle_model(meta["model"], g, requires_grad=True)
    tc = t.detach().cpu()
    B, n = tc.shape
    rx = g["ro"][:, None, :] + g["rd"][:, None, :] * tc[:, :, None]
    sig, rgb = om(rx, g["rd"][:, None, :].expand(-1, n, -1))
    delta = O.segment_lengths(tc, g["rd"])
    ww, rgb_map = O.raymarch_volume(sig.view(B, n), rgb.view(B, n, 3), delta)
    lo = torch.nn.functional.mse_loss(rgb_map, g["target"])
    lo.backward()
    # the same in float64: how far apart are two fp32 evaluations of the same function?
    p64 = {k: v.detach().double().requires_grad_(True) for k, v in om.params.items()}
    enc = lambda v: O.encode_positional(v, O.positional_freq_bands(int(meta["model"]["enc_x"][1])).double())
    om64 = O.Model(p64, enc, enc, meta["model"]["activ"], meta["model"]["resid"])
    tc64 = tc.double()
    rx64 = g["ro"].double()[:, None, :] + g["rd"].double()[:, None, :] * tc64[:, :, None]
    s64, c64 = om64(rx64, g["rd"].double()[:, None, :].expand(-1, n, -1))
    w64, m64 = O.raymarch_volume(s64.view(B, n), c64.view(B, n, 3), O.segment_lengths(tc64, g["rd"].double()))
    torch.nn.functional.mse_loss(m64, g["target"].double()).backward()
    print(prec, "loss", loss.item(), "golden", float(g["losses"][0]), "oracle@t", lo.item())
    for k, p in model.named_parameters():
        got = p.grad.cpu().double()
        gold = g["grad0." + k].double()
        orc = om.params[k].grad.double()
        s = gold.abs().max().item()
        print(f"  {k:22s} scale {s:.2e}  vs golden {(got-gold).abs().max().item()/s:.2e}  vs oracle@t {(got-orc).abs().max().item()/s:.2e}  golden-vs-oracle@t {(gold-orc).abs().max().item()/s:.2e}  ours-vs-f64 {(got-p64[k].grad).abs().max().item()/s:.2e}  oracle32-vs-f64 {(orc-p64[k].grad).abs().max().item()/s:.2e}")
ops.set_precision(None)
