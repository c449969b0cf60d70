"""Volume renderer (reference nerf/core/renderer.py): the drop-in boundary.

`BoundedVolumeRaymarcher.render_volume` keeps the reference's signature and
return tuple and drives the HIP kernels: stratified sampling, the fused MLP for
the coarse pass (density only), inverse-CDF resampling + merge, the fused MLP on
all Nc + Nf samples, and the wavefront-scan compositor."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from .. import ops


class BoundedVolumeRaymarcher:
    """BoundedVolumeRaymarcher(tn, tf, samples_c=64, samples_f=64)
    (reference renderer.py:147-171)."""

    def __init__(self, tn: float, tf: float, samples_c: int = 64, samples_f: int = 64) -> None:
        assert tf > tn, "[NeRF] `tf` should always be > `tn`"
        assert samples_c > 0, "[NeRF] `samples_c` should always be > 0"
        # limits of the HIP kernels (the reference has none): one wavefront handles a ray
        if not (samples_c <= 256 and samples_f <= 256 and (samples_f <= 0 or samples_c >= 3)):
            raise ValueError("[NeRF] the HIP kernels take samples_c <= 256, samples_f <= 256 and, with fine "
                             f"samples, samples_c >= 3 (got {samples_c} + {samples_f})")
        self.tn, self.tf = tn, tf
        self.samples_c, self.samples_f = samples_c, samples_f
        self.samples = samples_c + samples_f

    def render_volume(self, coarse, fine, ro: Tensor, rd: Tensor, perturb: Optional[bool] = False,
                      train: Optional[bool] = False) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        """-> t (B,N), weights (B,N), depth_map (B,), rgb_map (B,3)
        (reference renderer.py:173-205).

        `perturb` draws the stratification / inverse-CDF uniforms inside the sampling
        kernels (a Philox stream keyed by the seed and offset of torch's generator of the
        rays' device, which the reference draws from: ray.py:99, :167), so no (B, N)
        arrays of uniforms pass through memory; `torch.manual_seed` reproduces a render.
        With `train`, gradients reach the parameters of `fine` through the fine
        pass only (the coarse pass and the sample positions carry none), and the
        requires_grad flags of both modules are toggled exactly as in
        reference renderer.py:133,137.  Without `train` no graph is recorded, whatever
        autograd's state: every caller in the reference renders that way under
        inference_mode / no_grad (infer.py:13, train.py:71 for the val and test splits,
        bench.py:149); a caller that wants gradients passes train=True."""
        nc, nf = self.samples_c, self.samples_f
        rng = ops.next_rng(ro.device) if (perturb and ro.is_cuda) else None
        if train:
            if nf > 0:
                coarse.requires_grad(False)
            return ops.render_train(coarse, fine if nf > 0 else coarse, ro, rd, self.tn, self.tf, nc, nf, None, None, rng)
        with torch.no_grad():
            hc = ops.handle_for(coarse)
            hf = ops.handle_for(fine) if nf > 0 else None
            return ops.render(hc, hf, ro, rd, self.tn, self.tf, nc, nf, rng=rng)
