"""One process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the
CPU tests).  The path shards by rays, which are independent end to end:

  * rendering: every rank renders a contiguous block of ceil(n / world) rays (or whole frames
    of an orbit) with replicated weights and NO collective in the data path; the per-ray
    results (rgb + depth, 16 B/ray) are gathered once per frame and the frame-global depth
    normalisation of `infer` runs on the gathered frame, so the result is identical to the
    1-GPU render;
  * training: data parallel.  Every rank renders its own ray batch; the reference's loss is a
    mean over rays (train.py:278), so gradients are averaged with ONE all-reduce of a single
    flat fp32 bucket (321,924 elements = 1.29 MB for the 256 x 4 model: latency-bound on xGMI,
    so one collective, not one per tensor), issued before `scaler.unscale_` so that the
    scaler's inf check sees identical values on all ranks.

Nothing here touches a kernel: the functions take tensors on whatever device the process
group's backend supports, which is what lets world_size-2 gloo tests cover them on CPU.
"""
from typing import Callable, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def world(group=None) -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def seed_ranks(seed: int, group=None) -> int:
    """Seed torch's generators with `seed + rank`, so that the stratification / inverse-CDF
    uniforms of `render_volume(perturb=True)` (drawn from the device's global generator, as the
    reference draws them: ray.py:99, :167) differ from rank to rank while staying reproducible.
    Data-parallel ranks that all started from one seed would otherwise perturb their different
    batches with identical uniforms.  Returns the seed this rank used."""
    mine = int(seed) + world(group)[0]
    torch.manual_seed(mine)                          # seeds the CPU and every visible ROCm device
    return mine


def shard_bounds(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [start, end) of `n` rays owned by `rank`: ceil(n / world) rays each,
    the last ranks get what is left (possibly nothing)."""
    per = -(-n // world_size)
    start = min(rank * per, n)
    return start, min(start + per, n)


def gather_rays(local: Tensor, n: int, group=None, dst: Optional[int] = 0) -> Optional[Tensor]:
    """Reassemble per-ray results (n_local, C) sharded with `shard_bounds` into (n, C).
    dst = None: every rank gets the result (all-gather); otherwise only `dst` does."""
    rank, ws = world(group)
    if ws == 1:
        return local
    per = -(-n // ws)
    padded = local.new_zeros((per,) + tuple(local.shape[1:]))      # equal-sized pieces for the collective
    padded[: local.shape[0]] = local
    if dst is None:
        pieces = [torch.empty_like(padded) for _ in range(ws)]
        dist.all_gather(pieces, padded, group=group)
    else:
        pieces = [torch.empty_like(padded) for _ in range(ws)] if rank == dst else None
        dist.gather(padded, pieces, dst=dst, group=group)
        if rank != dst:
            return None
    return torch.cat(pieces, dim=0)[:n]


def render_sharded(render: Callable[[Tensor, Tensor], Tuple[Tensor, Tensor]], ro: Tensor, rd: Tensor,
                   group=None, dst: Optional[int] = 0) -> Optional[Tensor]:
    """Render this rank's block of the rays and gather (n, 4) = rgb | depth.
    `render(ro_block, rd_block) -> (depth (b,), rgb (b, 3))`, e.g. a closure over
    BoundedVolumeRaymarcher.render_volume."""
    rank, ws = world(group)
    n = ro.shape[0]
    s, e = shard_bounds(n, rank, ws)
    out = ro.new_zeros((e - s, 4))
    if e > s:
        depth, rgb = render(ro[s:e], rd[s:e])
        out[:, :3], out[:, 3] = rgb, depth
    return gather_rays(out, n, group, dst)


def infer_sharded(coarse, fine, raymarcher, ro: Tensor, rd: Tensor, H: int, W: int, batch_size: int = 32,
                  group=None, finish: Optional[Callable] = None):
    """`infer` with the frame's rays sharded over the ranks; rank 0 returns what
    `nerf_amd.infer.infer` returns (depth_map, rgb_map: CPU (H, W, 3) in [0, 255]), the
    others (None, None).  `finish(depth (n,), rgb (n, 3), H, W)` turns the gathered frame into
    the two display frames; default: the frame kernel behind `nerf_amd.infer.postprocess`."""
    if finish is None:
        from .infer import postprocess as finish

    device = next(coarse.parameters()).device

    @torch.inference_mode()
    def render(ro_b, rd_b):
        depth = torch.empty((ro_b.shape[0],), device=device)
        rgb = torch.empty((ro_b.shape[0], 3), device=device)
        for s in range(0, ro_b.shape[0], batch_size):
            e = min(s + batch_size, ro_b.shape[0])
            *_, D, C = raymarcher.render_volume(coarse, fine, ro_b[s:e], rd_b[s:e], train=False)
            depth[s:e], rgb[s:e] = D, C
        return depth, rgb

    frame = render_sharded(render, ro.to(device), rd.to(device), group, dst=0)
    if frame is None:
        return None, None
    return finish(frame[:, 3], frame[:, :3], H, W)


def flat_gradient(params) -> Optional[Tensor]:
    """The gradients of `params` as ONE flat tensor if they already are consecutive views of one
    buffer in this order (as the MLP backward above returns them), else None.  No copy."""
    grads = [p.grad for p in params]
    if not grads or any(g is None or g.dtype != torch.float32 or not g.is_contiguous() for g in grads):
        return None
    base, ptr = grads[0], grads[0].data_ptr()
    for g in grads:
        if g.data_ptr() != ptr or g.device != base.device:
            return None
        ptr += 4 * g.numel()
    total = (ptr - base.data_ptr()) // 4
    storage = base.untyped_storage()
    if base.storage_offset() + total > storage.nbytes() // 4:
        return None
    return torch.empty(0, dtype=torch.float32, device=base.device).set_(storage, base.storage_offset(), (total,), (1,))


class GradientBucket:
    """All gradients of a parameter list as one flat fp32 buffer and one all-reduce."""

    def __init__(self, params: Iterable[Tensor]):
        self.params: List[Tensor] = [p for p in params]
        self.sizes = [p.numel() for p in self.params]
        first = self.params[0]
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=first.device)
        self.views = [v.view_as(p) for v, p in zip(self.flat.split(self.sizes), self.params)]

    def average(self, group=None) -> None:
        """grad <- mean over ranks of grad, for every parameter that has one.  A parameter
        without a gradient on this rank contributes zeros (and receives the others' mean)
        only if some rank has one; the set of trained parameters must agree across ranks."""
        rank, ws = world(group)
        if ws == 1:
            return
        flat = flat_gradient(self.params)        # the MLP backward's own flat buffer: no staging copies
        if flat is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            flat.mul_(1.0 / ws)
            return
        have = [p.grad is not None for p in self.params]
        self.flat.zero_()
        srcs = [p.grad for p, h in zip(self.params, have) if h]
        dsts = [v for v, h in zip(self.views, have) if h]
        if srcs:
            torch._foreach_copy_(dsts, srcs)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.mul_(1.0 / ws)
        if srcs:
            torch._foreach_copy_(srcs, dsts)


_BUCKETS = {}


def average_gradients(module, group=None) -> None:
    """Data-parallel gradient averaging for `module` (no-op in a single process)."""
    if world(group)[1] == 1:
        return
    bucket = _BUCKETS.get(id(module))
    params = list(module.parameters())
    if bucket is None or len(bucket.params) != len(params) or any(a is not b for a, b in zip(bucket.params, params)):
        bucket = _BUCKETS[id(module)] = GradientBucket(params)
    bucket.average(group)


def broadcast_parameters(module, src: int = 0, group=None) -> None:
    """Make every rank start from rank `src`'s weights (one flat broadcast)."""
    if world(group)[1] == 1:
        return
    # detach() shares the parameter's version counter (`.data` has its own), so the in-place copy
    # below is seen by everything that caches derived state per version (ops.ModelHandle does)
    params = [p.detach() for p in module.parameters()]
    flat = torch.cat([p.reshape(-1) for p in params])
    dist.broadcast(flat, src=src, group=group)
    torch._foreach_copy_(params, [v.view_as(p) for v, p in zip(flat.split([p.numel() for p in params]), params)])
