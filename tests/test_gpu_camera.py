"""Camera rays in, display frames out: the HIP kernels either side of `render_volume`, called
through the C ABI, against the reference's golden vectors and the CPU oracle.

Tolerances: fp32 ray directions 1e-5 relative (1e-6 absolute near zero); poses 1e-6 absolute
(entries are products of sines and cosines, at most 4 in magnitude); float frames on the
[0, 255] scale 1e-5 relative; 8-bit frames equal the truncated float frames exactly.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import assert_close, product_model
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu

ORBIT = (-np.pi, np.pi), (-np.pi, np.pi), (-np.pi, np.pi), (4., 4.)


@pytest.fixture()
def ops():
    import nerf_amd.ops as ops_
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    ops_.set_precision(None)
    yield ops_
    ops_.set_precision(None)


def test_turnaround_poses_golden(ops):
    from nerf_amd.data.path import turnaround, turnaround_poses
    _, g = load_golden("g10_camera")
    poses = turnaround_poses(*ORBIT, 8)
    assert poses.shape == (8, 4, 4) and poses.is_cuda
    assert_close(poses.cpu(), g["poses8"], rtol=0, atol=1e-6, what="poses8")
    alt = turnaround_poses((0.25, 1.5), (-0.5, 0.75), (0., 0.), (3., 5.), 5)
    assert_close(alt.cpu(), g["poses5_alt"], rtol=0, atol=1e-6, what="poses5_alt")
    one = turnaround(0.25, -0.5, 0., 3.)
    assert_close(one.cpu(), g["poses5_alt"][0], rtol=0, atol=1e-6, what="single pose")
    assert turnaround_poses(*ORBIT, 0).shape == (0, 4, 4)
    # a longer path against the oracle
    many = turnaround_poses((-1., 2.), (0.5, -3.), (3., -3.), (2., 6.), 100)
    assert_close(many.cpu(), O.turnaround_poses((-1., 2.), (0.5, -3.), (3., -3.), (2., 6.), 100),
                 rtol=0, atol=2e-6, what="100 poses")


def test_camera_rays_golden(ops):
    from nerf_amd.core.ray import phinhole_ray_projection, pinhole_ray_directions
    meta, g = load_golden("g10_camera")
    for H, W in meta["sizes"]:
        focal = .5 * W / np.tan(.5 * meta["cam_angle_x"])
        prd = pinhole_ray_directions(H, W, focal)
        assert prd.shape == (1, H, W, 3) and prd.is_cuda
        assert_close(prd.cpu(), g[f"prd_{H}x{W}"], rtol=1e-6, atol=1e-7, what="prd")
        for k in meta["pose_ids"]:
            pose = g["poses8"][k].cuda()
            ro, rd = phinhole_ray_projection(prd, pose)
            assert ro.shape == prd.shape and rd.shape == prd.shape
            assert torch.equal(ro.cpu(), g[f"ro_{H}x{W}_{k}"])
            assert_close(rd.cpu(), g[f"rd_{H}x{W}_{k}"], rtol=1e-5, atol=1e-6, what=f"rd pose {k}")
            # the fused form, whole frame and a ragged pixel range
            ro2, rd2 = ops.camera_rays(pose, H, W, focal)
            assert torch.equal(ro2, ro.view(-1, 3)) and torch.equal(rd2, rd.view(-1, 3))
            b, n = W + 1, H * W - W - 3
            ro3, rd3 = ops.camera_rays(pose, H, W, focal, b, n)
            assert torch.equal(rd3, rd.view(-1, 3)[b:b + n]) and ro3.shape == (n, 3)


def test_camera_rays_full_frame_properties(ops):
    """800 x 800: unit directions, the centre pixel looks down the camera's -z axis, the frame
    is mirror-symmetric about it; range checks raise."""
    H = W = 800
    focal = .5 * W / np.tan(.5 * 0.6194058656692505)
    pose = ops.orbit_poses(*ORBIT, 40)[7]
    ro, rd = ops.camera_rays(pose, H, W, focal)
    assert_close(rd.norm(dim=-1).cpu(), torch.ones(H * W), rtol=1e-6, atol=0, what="unit directions")
    assert torch.equal(ro, pose[:3, 3].expand(H * W, 3))
    centre = rd.view(H, W, 3)[H // 2, W // 2]
    assert_close(centre.cpu(), (-pose[:3, 2]).cpu(), rtol=1e-5, atol=1e-6, what="optical axis")
    ref_ro, ref_rd = O.pinhole_ray_projection(O.pinhole_ray_directions(H, W, focal)[0], pose.cpu())
    assert_close(rd.cpu(), ref_rd.reshape(-1, 3), rtol=1e-5, atol=1e-6, what="rd vs oracle")
    with pytest.raises(RuntimeError):
        ops.camera_rays(pose, H, W, focal, H * W - 5, 6)
    with pytest.raises(RuntimeError):
        ops.camera_rays(pose.cpu(), H, W, focal)


def test_frame_finish_matches_reference_frames(ops):
    """Float frames from the per-ray maps == infer.py:58-66; 8-bit frames == astype(uint8)."""
    g = torch.Generator().manual_seed(5)
    H, W = 37, 53
    depth = torch.rand(H * W, generator=g) * 4 + 2
    rgb = torch.rand(H * W, 3, generator=g) * 1.2 - 0.1           # some values outside [0, 1]
    depth[11] = float("nan")
    depth[12] = float("inf")
    for d in (depth, depth.nan_to_num(posinf=5.5), torch.full((H * W,), 3.0)):
        want_d, want_c = O.postprocess_frame(d, rgb, H, W)
        df, cf, du, cu = ops.frame_finish(d.cuda(), rgb.cuda(), as_float=True, as_uint8=True)
        assert_close(df.view(H, W, 3).cpu(), want_d, rtol=1e-6, atol=0, what="depth frame")
        assert torch.equal(cf.view(H, W, 3).cpu(), want_c)
        assert (du.cpu().numpy() == df.cpu().numpy().astype(np.uint8)).all()
        assert (cu.view(H, W, 3).cpu().numpy() == O.frame_to_uint8(want_c)).all()
    # a full-size frame: extremes land on 0 and 255
    n = 800 * 800
    d = torch.rand(n, generator=g) * 4 + 2
    df, _, du, _ = ops.frame_finish(d.cuda(), torch.zeros(n, 3).cuda(), as_float=True, as_uint8=True)
    want_d, _ = O.postprocess_frame(d, torch.zeros(n, 3), 800, 800)
    assert_close(df.cpu(), want_d.view(-1, 3), rtol=1e-6, atol=0, what="800x800 depth frame")
    assert int(du.min()) == 0 and int(du.max()) == 254 + int(float(df.max()) >= 255.0)


def test_orbit_renders_frames_like_infer(ops):
    """`orbit` (rays generated per batch on the device, 8-bit frames back) == `infer` on rays
    built the reference's way, truncated to 8 bits."""
    import nerf_amd.infer as infer_mod
    from nerf_amd.core import BoundedVolumeRaymarcher, NeRF
    meta, g = load_golden("g8_infer")
    model = product_model(meta["model"], g)
    bvr = BoundedVolumeRaymarcher(2.0, 6.0, 32, 32)
    H, W = 12, 20
    focal = .5 * W / np.tan(.5 * 0.6194058656692505)
    poses = ops.orbit_poses(*ORBIT, 3)
    ops.set_precision("fp32")
    frames = list(infer_mod.orbit(model, model, bvr, poses, H, W, focal, batch_size=64))
    assert len(frames) == 3
    prd = O.pinhole_ray_directions(H, W, focal)[0]
    for k, (d8, c8) in enumerate(frames):
        assert d8.shape == (H, W, 3) and d8.dtype == np.uint8 and c8.dtype == np.uint8
        ro, rd = O.pinhole_ray_projection(prd, poses[k].cpu())
        depth, rgb = NeRF.infer(model, model, bvr, ro.reshape(-1, 3), rd.reshape(-1, 3), H, W, 64, False)
        # one-ulp differences in rd may move a value across an integer boundary: allow off-by-one
        assert np.abs(d8.astype(int) - depth.numpy().astype(np.uint8).astype(int)).max() <= 1
        assert np.abs(c8.astype(int) - rgb.numpy().astype(np.uint8).astype(int)).max() <= 1
    ops.set_precision(None)
