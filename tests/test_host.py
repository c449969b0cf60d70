"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol that
include/nerfx.h declares, argument validation fails loudly without a GPU, schedulers match
the reference's golden learning-rate curves, and the module layout / state_dict keys match."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nerfx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nerfx_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from nerf_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f"libnerfx.so lacks {n}"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert _lib.lib.nerfx_version() == 100


def test_argument_validation_without_gpu():
    from nerf_amd import _lib
    lib = _lib.lib
    # bad sample count is rejected before any HIP call
    rc = lib.nerfx_composite_fwd(None, None, None, None, None, 4, 0, None, None, None, None)
    assert rc == -1 and b"n_samples" in lib.nerfx_last_error()
    rc = lib.nerfx_pdf_sample(None, None, 4, 2, 8, None, None, None, None, None, None)
    assert rc == -1 and b"nc" in lib.nerfx_last_error()
    rc = lib.nerfx_sample_encode_fwd(None, None, 4, 6.0, 2.0, 8, None, 0, 0, None, 0, 0, None,
                                     None, None, None, None, None, None)
    assert rc == -1 and b"tf" in lib.nerfx_last_error()
    assert lib.nerfx_render_workspace_bytes(1000, 64, 64) >= 1000 * (3 * 64 + 4 * 128) * 4
    assert lib.nerfx_render_workspace_bytes(-1, 64, 64) == -1
    # empty batches are a no-op everywhere
    assert lib.nerfx_composite_fwd(None, None, None, None, None, 0, 8, None, None, None, None) == 0
    assert lib.nerfx_pdf_sample(None, None, 0, 64, 64, None, None, None, None, None, None) == 0
    assert lib.nerfx_mlp_fwd_points(None, 0, None, None, 0, None, None, None) == -1     # NULL model
    assert b"model" in lib.nerfx_last_error()
    with pytest.raises(_lib.NerfxError):
        _lib.check(-1)


def test_product_refuses_cpu_tensors():
    """There is no CPU fallback: CPU tensors are an error, not a slow path."""
    import nerf_amd.ops as ops
    from nerf_amd.core import BoundedVolumeRaymarcher, NeRF, PositionalEncoding
    m = NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 32, 2)
    with pytest.raises(RuntimeError, match="ROCm"):
        with torch.no_grad():
            m(torch.zeros(4, 3), torch.zeros(4, 3))
    with pytest.raises(RuntimeError, match="ROCm"):
        BoundedVolumeRaymarcher(2., 6.).render_volume(m, m, torch.zeros(4, 3), torch.ones(4, 3))
    with pytest.raises(RuntimeError, match="ROCm"):
        ops.composite_fwd(torch.zeros(2, 8), None, torch.zeros(2, 8), torch.ones(2, 3))
    with pytest.raises(RuntimeError, match="ROCm"):
        PositionalEncoding(3, 4)(torch.zeros(2, 3))


def test_module_layout_matches_reference():
    from nerf_amd.core import (BoundedVolumeRaymarcher, FeatureMapping, FourierFeatures, NeRF,
                               PositionalEncoding)
    m = NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 256, 4)
    keys = list(m.state_dict().keys())
    assert keys == ["phi_x.freq_bands", "phi_d.freq_bands",
                    "fc_1.0.0.weight", "fc_1.0.0.bias", "fc_1.1.0.weight", "fc_1.1.0.bias",
                    "fc_2.0.0.weight", "fc_2.0.0.bias", "fc_2.1.0.weight", "fc_2.1.0.bias",
                    "sigma.0.weight", "sigma.0.bias", "features.weight", "features.bias",
                    "rgb.0.weight", "rgb.0.bias", "rgb.2.weight", "rgb.2.bias"]
    assert sum(p.numel() for p in m.parameters()) == 321_924
    assert tuple(m.fc_2[0][0].weight.shape) == (256, 295) and tuple(m.rgb[0].weight.shape) == (128, 295)
    assert (m.width, m.depth, m.resid) == (256, 4, True) and m.activ is torch.nn.ReLU
    small = NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 32, 2, resid=False)
    assert sum(p.numel() for p in small.parameters()) == 4_628
    assert NeRF(PositionalEncoding(3, 10), PositionalEncoding(3, 4)).depth == 8          # ctor default
    pe = PositionalEncoding(3, 6)
    assert (pe.i_dim, pe.o_dim, pe.N_freqs) == (3, 39, 6) and isinstance(pe, FeatureMapping)
    _, g = load_golden("g2_encodings")
    assert torch.equal(pe.freq_bands, g["pe6_freq"])
    ff = FourierFeatures(3, 64, 8.0)
    assert (ff.i_dim, ff.o_dim, ff.features, ff.sigma, tuple(ff.size), tuple(ff.b.shape)) == (3, 128, 64, 8.0, (64, 3), (64, 3))
    m.requires_grad(False)
    assert not any(p.requires_grad for p in m.parameters())
    m.requires_grad()
    assert all(p.requires_grad for p in m.parameters())
    bvr = BoundedVolumeRaymarcher(2., 6.)
    assert (bvr.tn, bvr.tf, bvr.samples_c, bvr.samples_f, bvr.samples) == (2., 6., 64, 64, 128)
    with pytest.raises(AssertionError):
        BoundedVolumeRaymarcher(6., 2.)
    with pytest.raises(AssertionError):
        BoundedVolumeRaymarcher(2., 6., 0, 64)
    assert hasattr(NeRF, "infer") or __import__("nerf_amd.infer") and hasattr(NeRF, "infer")


def test_schedulers_golden():
    from nerf_amd.core import IndendityScheduler, LogDecayScheduler, MipNeRFScheduler, Scheduler
    _, g = load_golden("g9_schedulers")
    p = [torch.nn.Parameter(torch.zeros(1))]
    for tag, mk in (("identity", lambda o: IndendityScheduler(o, 10, 3e-4)),
                    ("logdecay", lambda o: LogDecayScheduler(o, 4, 25, (5e-6, 5e-4), 8)),
                    ("mipnerf", lambda o: MipNeRFScheduler(o, 4, 1, 25, (5e-6, 5e-4), 1e-2))):
        o = torch.optim.SGD(p, lr=1.0)
        s = mk(o)
        assert isinstance(s, Scheduler)
        lrs = [o.param_groups[0]["lr"]]
        for _ in range(100):
            assert s.step() is s
            lrs.append(o.param_groups[0]["lr"])
        want = g[tag].tolist()
        assert len(lrs) == len(want)
        for a, b in zip(lrs, want):
            assert abs(a - b) <= 1e-12 * max(1.0, abs(b)) + 1e-15, (tag, a, b)


def test_install_as_alias():
    import sys
    import nerf_amd
    nerf_amd.install_as("nerf_alias_for_test")
    from nerf_alias_for_test.core import NeRF as A      # noqa
    from nerf_amd.core import NeRF as B
    assert A is B
    for k in [k for k in sys.modules if k.startswith("nerf_alias_for_test")]:
        del sys.modules[k]


def test_model_file_round_trip(tmp_path):
    """save / load keep constructor arguments, parameters and encoder buffers; the file holds
    nothing executable (read with weights_only=True)."""
    import nerf_amd.io as io
    from nerf_amd.core import FourierFeatures, NeRF, PositionalEncoding
    torch.manual_seed(3)
    for model in (NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 4), 32, 2, torch.nn.SiLU, False),
                  NeRF(FourierFeatures(3, 16, 8.0), PositionalEncoding(3, 6), 64, 4)):
        path = str(tmp_path / "m.nerfx")
        io.save(model, path)
        back = io.load(path, device="cpu")
        assert io.describe(back) == io.describe(model)
        assert back.width == model.width and back.depth == model.depth and back.resid == model.resid
        a, b = model.state_dict(), back.state_dict()
        assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    with pytest.raises(Exception):
        torch.save({"format": "other"}, str(tmp_path / "x"))
        io.load(str(tmp_path / "x"), device="cpu")


def test_from_torchscript_reads_a_scripted_field(tmp_path):
    """A scripted module with the reference's layout (what its *.model.ts files hold) becomes a
    nerf_amd NeRF with the same architecture and weights."""
    import nerf_amd.io as io
    from helpers import ScriptableField
    torch.manual_seed(4)
    ref = ScriptableField(6, 4, 32, 4, True)
    path = str(tmp_path / "NeRF_test.model.ts")
    torch.jit.save(torch.jit.script(ref), path)
    nerf = io.from_torchscript(path, device="cpu")
    assert io.describe(nerf) == {"phi_x": {"kind": "positional", "n": 6}, "phi_d": {"kind": "positional", "n": 4},
                                 "width": 32, "depth": 4, "activ": "ReLU", "resid": True}
    want = ref.state_dict()
    got = nerf.state_dict()
    assert set(want) == set(got) and all(torch.equal(want[k], got[k]) for k in want)


def _kernel_assembly(tmp_path, stem):
    """The assembly of nerf_amd/csrc/<stem>.hip as the Makefile built it (kept beside the objects), else
    compiled here with the Makefile's flags."""
    import shutil
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    built = os.path.join(root, "nerf_amd", "csrc", "build", f"{stem}-hip-amdgcn-amd-amdhsa-gfx950.s")
    if os.path.exists(built):
        return built
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine and no assembly kept by the build")
    asm = str(tmp_path / f"{stem}.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "--cuda-device-only", "-S", "-O3", "-std=c++17", "-fPIC",
                    os.path.join(root, "nerf_amd", "csrc", f"{stem}.hip"), "-o", asm], check=True, capture_output=True, timeout=900)
    return asm


def _run_checker(mode, asm, kernel):
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    return subprocess.run([sys.executable, os.path.join(root, "tools", "check_async_loads.py"), mode, asm, kernel],
                          capture_output=True, text=True, timeout=300)


def test_async_tape_loads_keep_their_registers(tmp_path):
    """The bf16 backward chain issues its tape loads from inline asm, which the compiler does not
    track: in the compiled assembly nothing may touch their destination registers before the
    counted wait that precedes their use (tools/check_async_loads.py; the Makefile runs the same
    check on every build)."""
    out = _run_checker("tape", _kernel_assembly(tmp_path, "mlp_bwd_bf16"), "mlp_bwd_kernel")
    assert out.returncode == 0, out.stdout[-2000:]


def test_carried_operands_keep_their_registers(tmp_path):
    """The 16x16x32 forward's run blocks request the first A operands of the block that follows:
    between the two asm statements the compiler must leave those registers alone."""
    out = _run_checker("carry", _kernel_assembly(tmp_path, "mlp_fwd16_bf16"), "mlp_fwd16_kernel")
    assert out.returncode == 0, out.stdout[-2000:]


def test_async_checker_sees_a_violation(tmp_path):
    """The checker itself: a register of an in-flight load touched before the marked wait, an operand
    carried between two asm statements touched in between, and a file with nothing to check."""
    tape = """kern_a:
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v[2:3], off
	;;#ASMEND
	s_waitcnt vmcnt(4)
	v_mov_b32_e32 v9, v5
	;;#ASMSTART
	s_waitcnt vmcnt(1) ; async tape loads landed
	;;#ASMEND
	v_mov_b32_e32 v9, v5
"""
    carry = """kern_b:
	;;#ASMSTART
	v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]
	ds_read_b128 v[8:11], v20 offset:0
	;;#ASMEND
	v_add_u32_e32 v8, 1, v8
	;;#ASMSTART
	v_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]
	;;#ASMEND
"""
    for mode, text, kernel in (("tape", tape, "kern_a"), ("carry", carry, "kern_b")):
        path = tmp_path / f"{mode}.s"
        path.write_text(text)
        out = _run_checker(mode, str(path), kernel)
        assert out.returncode == 1 and "1 violations" in out.stdout, out.stdout
        path.write_text(text.replace("v_mov_b32_e32 v9, v5\n\t;;#ASMSTART\n\ts_waitcnt", "\t;;#ASMSTART\n\ts_waitcnt").replace("\tv_add_u32_e32 v8, 1, v8\n", ""))
        out = _run_checker(mode, str(path), kernel)
        assert out.returncode == 0, out.stdout
    empty = tmp_path / "empty.s"
    empty.write_text("kern_c:\n\tv_mov_b32_e32 v0, v1\n")
    assert _run_checker("tape", str(empty), "kern_c").returncode == 2
