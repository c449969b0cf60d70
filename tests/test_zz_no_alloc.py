"""The allocation-free render entry point against the regular one, through the C ABI: two calls
in a row on one stream (the second with fewer rays, so stale workspace contents of the first are
in reach if anything read them), outputs compared bit for bit (same kernels, same order); the second call is also held against the
CPU oracle directly."""
import ctypes as C

import pytest
import torch

from conftest import load_golden
from helpers import assert_close, oracle_model, product_model
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _outputs(B, n, dev):
    return [torch.full(shape, float("nan"), device=dev) for shape in ((B, n), (B, n), (B,), (B, 3))]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_static_workspace_render_equals_regular_render(prec):
    import nerf_amd.ops as ops
    from nerf_amd import _lib
    lib = _lib.lib
    meta, g = load_golden("g3_mlp_base")
    model = product_model(meta, g)
    handle = ops.handle_for(model)
    code = _lib.PREC_BF16 if prec == "bf16" else _lib.PREC_FP32
    mptr = handle.ready(code)
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nc, nf = 64, 64
    cap = lib.nerfx_render_static_max_rays(nc, nf)
    assert 4096 <= cap <= (16 << 20) // (4 * (3 * nc + 4 * (nc + nf)))
    gen = torch.Generator().manual_seed(21)
    for B, perturb in ((cap, False), (777, True)):          # full workspace first, then a smaller call
        ro, rd = torch.rand(B, 3, generator=gen).to(dev), (torch.rand(B, 3, generator=gen) + 0.05).to(dev)
        u_c = torch.rand(B, nc, generator=gen).to(dev) if perturb else None
        u_f = torch.rand(B, nf, generator=gen).to(dev) if perturb else None
        want = _outputs(B, nc + nf, dev)
        ws_bytes = lib.nerfx_render_workspace_bytes(B, nc, nf)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.nerfx_render_fwd(mptr, mptr, code, _p(ro), _p(rd), B, 2.0, 6.0, nc, nf, _p(u_c), _p(u_f),
                                        _p(ws), ws_bytes, *[_p(o) for o in want], stream))
        got = _outputs(B, nc + nf, dev)
        _lib.check(lib.nerfx_render_fwd_static(mptr, mptr, code, _p(ro), _p(rd), B, 2.0, 6.0, nc, nf, _p(u_c), _p(u_f),
                                               *[_p(o) for o in got], stream))
        torch.cuda.synchronize()
        for name, a, b in zip(("t", "weights", "depth", "rgb"), got, want):
            assert torch.isfinite(b).all(), name
            assert torch.equal(a, b), f"{prec} B={B}: {name} differs from nerfx_render_fwd"
        if B == 777:                                         # and against the CPU oracle, on its own
            om = oracle_model(meta, g)
            with torch.no_grad():
                t_ref, w_ref, d_ref, c_ref = O.render_volume(om, om, ro.cpu(), rd.cpu(), 2.0, 6.0, nc, nf, u_c.cpu(), u_f.cpu())
            tol = dict(rtol=1e-5, atol=1e-5) if prec == "fp32" else dict(rtol=0, atol=1e-2)
            assert_close(got[3].cpu(), c_ref, what=f"{prec} static entry point: rgb_map vs oracle", **tol)
            if prec == "fp32":
                assert_close(got[2].cpu(), d_ref, rtol=1e-5, atol=1e-5, what="static entry point: depth vs oracle")
    # coarse-only mode and the size limit
    B = 100
    ro, rd = torch.rand(B, 3, generator=gen).to(dev), torch.rand(B, 3, generator=gen).to(dev)
    want, got = _outputs(B, nc, dev), _outputs(B, nc, dev)
    ws_bytes = lib.nerfx_render_workspace_bytes(B, nc, 0)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.nerfx_render_fwd(mptr, None, code, _p(ro), _p(rd), B, 2.0, 6.0, nc, 0, None, None, _p(ws), ws_bytes,
                                    *[_p(o) for o in want], stream))
    _lib.check(lib.nerfx_render_fwd_static(mptr, None, code, _p(ro), _p(rd), B, 2.0, 6.0, nc, 0, None, None,
                                           *[_p(o) for o in got], stream))
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    rc = lib.nerfx_render_fwd_static(mptr, mptr, code, _p(ro), _p(rd), cap + 1, 2.0, 6.0, nc, nf, None, None,
                                     *[_p(o) for o in got], stream)
    assert rc == -1 and b"static workspace" in lib.nerfx_last_error()


def test_static_entry_point_under_graph_capture():
    """After one warm call per device (which resolves the workspace symbol and opts the kernels into
    their LDS sizes), the entry point records into a HIP graph: it makes no allocation and no other
    call that capture forbids.  The replayed graph gives the bits of a direct call."""
    import nerf_amd.ops as ops
    from nerf_amd import _lib
    lib = _lib.lib
    meta, g = load_golden("g3_mlp_base")
    model = product_model(meta, g)
    code = _lib.PREC_BF16
    mptr = ops.handle_for(model).ready(code)
    dev = torch.device("cuda", torch.cuda.current_device())
    nc, nf, B = 64, 64, 1500
    gen = torch.Generator().manual_seed(22)
    ro, rd = torch.rand(B, 3, generator=gen).to(dev), (torch.rand(B, 3, generator=gen) + 0.05).to(dev)
    want, got = _outputs(B, nc + nf, dev), _outputs(B, nc + nf, dev)

    def call(outs):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.nerfx_render_fwd_static(mptr, mptr, code, _p(ro), _p(rd), B, 2.0, 6.0, nc, nf, None, None,
                                               *[_p(o) for o in outs], stream))

    call(want)                                               # the warm call, outside capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(got)
    graph.replay()
    torch.cuda.synchronize()
    for name, a, b in zip(("t", "weights", "depth", "rgb"), got, want):
        assert torch.isfinite(b).all(), name
        assert torch.equal(a, b), f"{name}: replayed graph differs from the direct call"
