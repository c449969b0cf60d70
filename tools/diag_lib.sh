#!/bin/bash
# Builds nerf_amd/libnerfx_x<name>.so: the working tree's objects with ONE translation unit
# recompiled with extra flags (timing diagnostics; same-box A/B runs through NERFX_LIB=...).
#   tools/diag_lib.sh <name> <unit, e.g. mlp_fwd16_bf16> [-DFLAG ...]
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift; shift
mkdir -p /tmp/diag_$name
(cd nerf_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed "$@" -c $unit.hip -o /tmp/diag_$name/unit.o)
objs=$(ls nerf_amd/csrc/build/*.o | grep -v -e "-hip-" -e "-host-" -e "/$unit.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o nerf_amd/libnerfx_x$name.so $objs /tmp/diag_$name/unit.o
