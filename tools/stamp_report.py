"""Diagnostic (GPU box): where do the MLP kernel's wave-cycles go?  Needs the stamps build:
    make -C nerf_amd/csrc stamps && NERFX_LIB=nerf_amd/libnerfx_stamps.so python tools/stamp_report.py
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_amd.ops as ops
from nerf_amd._lib import lib
from nerf_amd.core import NeRF, PositionalEncoding

torch.manual_seed(0)
model = NeRF(PositionalEncoding(3, 6), PositionalEncoding(3, 6), 256, 4).cuda()
h = ops.handle_for(model)
B = 65536
ro, rd = torch.rand(B, 3, device="cuda"), torch.rand(B, 3, device="cuda")
names = ["prologue", "wait for own loads + opening barrier", "mfma runs", "next run set-up", "conversion / refill issue / heads", "encode", "closing barrier", "-"]
# (kernels whose run blocks are not cut at the barriers book both barriers under "mfma runs")
for n, dens in ((64, True), (128, False)):
    t = torch.rand(B, n, device="cuda") * 4 + 2
    for prec in (1,):
        ops.mlp_rays(h, ro, rd, t, density_only=dens, prec=prec)
        buf = torch.zeros(8, dtype=torch.int64, device="cuda")
        lib.nerfx_debug_set_stamps(ctypes.c_void_p(buf.data_ptr()))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.mlp_rays(h, ro, rd, t, density_only=dens, prec=prec); b.record()
        torch.cuda.synchronize()
        lib.nerfx_debug_set_stamps(None)
        v = buf.cpu().double()
        print(f"n={n} density_only={dens} prec={prec}: {a.elapsed_time(b):.3f} ms; wave-cycle shares: " +
              ", ".join(f"{k} {100 * x / v.sum():.1f}%" for k, x in zip(names, v)) + f"; total {v.sum():.3e}")
