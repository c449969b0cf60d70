"""Debug helper (GPU box): sweep model shapes through the fused MLP and compare with the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import oracle_model, product_model
import nerf_amd.ops as ops

g = torch.Generator().manual_seed(1)
x = (torch.rand(300, 3, generator=g) * 2 - 1) * 3
d = torch.rand(300, 3, generator=g) * 2 - 1
for width, depth, resid, L in [(32, 2, False, 6), (32, 2, True, 6), (32, 4, False, 6), (64, 2, False, 6),
                               (256, 2, False, 6), (256, 4, False, 6), (32, 2, False, 10), (128, 4, True, 4)]:
    meta = dict(enc_x=["pe", L], enc_d=["pe", L], width=width, depth=depth, activ="relu", resid=resid,
                seed=7, e_x=3 + 6 * L, e_d=3 + 6 * L)
    om, pm = oracle_model(meta, {}), product_model(meta, {})
    ws, wc = om(x, d)
    h = ops.handle_for(pm)
    for prec in (0, 1):
        s, c = ops.mlp_points(h, x.cuda(), d.cuda(), prec=prec)
        s2, _ = ops.mlp_points(h, x.cuda(), d.cuda(), prec=prec, density_only=True)
        print(f"w{width} d{depth} resid{int(resid)} L{L} prec{prec}: sigma err {(s.cpu()-ws).abs().max():.2e} "
              f"(density-only {(s2.cpu()-ws).abs().max():.2e}) rgb err {(c.cpu()-wc).abs().max():.2e}")
