#!/bin/bash
# Builds nerf_amd/libnerfx_diag_prev.so: the working tree's objects with the M16 inference kernel of
# the last commit, for same-box A/B timing (NERFX_LIB=nerf_amd/libnerfx_diag_prev.so python bench.py ...).
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/ab_prev && mkdir -p /tmp/ab_prev/nerf_amd /tmp/ab_prev/include
cp -r nerf_amd/csrc /tmp/ab_prev/nerf_amd/csrc && cp include/nerfx.h /tmp/ab_prev/include/
for f in mlp_fwd16_impl.hpp mlp_runs_m16.inc mlp_fwd_impl.hpp mlp.hpp; do git show HEAD:nerf_amd/csrc/$f > /tmp/ab_prev/nerf_amd/csrc/$f; done
(cd /tmp/ab_prev/nerf_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -c mlp_fwd16_bf16.hip -o /tmp/ab_prev/old16.o)
objs=$(ls nerf_amd/csrc/build/*.o | grep -v -e "-hip-" -e "-host-" -e "/mlp_fwd16_bf16.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o nerf_amd/libnerfx_diag_prev.so $objs /tmp/ab_prev/old16.o
