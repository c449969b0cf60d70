"""Build container only (needs /root/reference): time the REAL reference's `infer` on the 100x100
CPU case next to the oracle's (with the coarse colour head executed, as the reference does), so
that the oracle's rate may be quoted as "the reference's CPU path" where the two agree.

    python tools/time_reference_cpu.py
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "_stubs"))
sys.path.insert(0, "/root/reference")

from nerf.core import BoundedVolumeRaymarcher, NeRF, PositionalEncoding  # noqa: E402
import nerf.infer as ref_infer  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402

torch.manual_seed(0)
model = NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 256, 4)
bvr = BoundedVolumeRaymarcher(2.0, 6.0, 64, 64)
g = torch.Generator().manual_seed(0)
H = W = 100
ro, rd = torch.rand(H * W, 3, generator=g), torch.rand(H * W, 3, generator=g)
params = {k: v.detach() for k, v in model.named_parameters()}
f = O.positional_freq_bands(6)
om = O.Model(params, O.make_encoder("pe", f), O.make_encoder("pe", f), "relu", True)


def best_of(fn, n=3):
    fn()
    best = float("inf")
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


t_ref = best_of(lambda: ref_infer.infer(model, model, bvr, ro, rd, H, W, batch_size=16384, verbose=False))
t_ora = best_of(lambda: O.infer(om, om, ro, rd, H, W, 2.0, 6.0, 64, 64, batch_size=16384, as_reference=True))
t_lean = best_of(lambda: O.infer(om, om, ro, rd, H, W, 2.0, 6.0, 64, 64, batch_size=16384))
print(json.dumps({"threads": torch.get_num_threads(), "rays": H * W,
                  "reference_s": t_ref, "reference_rays_per_s": H * W / t_ref,
                  "oracle_as_reference_s": t_ora, "oracle_as_reference_rays_per_s": H * W / t_ora,
                  "oracle_over_reference_speed": t_ref / t_ora,
                  "oracle_skipping_discarded_head_s": t_lean}))
