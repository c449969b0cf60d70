#!/bin/bash
# GPU box: bench.py's `other` sub-record through several builds of the library on ONE box (same-box A/B):
#   tools/ab_other.sh <tag> [lib ...]      (libs relative to nerf_amd/; libnerfx.so runs first and last)
set -u
tag=$1; shift
for lib in libnerfx.so "$@" libnerfx.so; do
  NERFX_LIB=$PWD/nerf_amd/$lib timeout -k 10 200 python3 bench.py --only-other > gpurun_out/abo_${tag}_$lib.json 2> gpurun_out/abo_${tag}_$lib.err || { echo "$lib failed"; tail -3 gpurun_out/abo_${tag}_$lib.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/abo_${tag}_$lib.json').read().strip().splitlines()[-1])['other']
print('$lib', {k: round(v['ms_per_frame'], 3) for k, v in d.items()})"
done
