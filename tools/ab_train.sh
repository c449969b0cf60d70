#!/bin/bash
# GPU box: the training-step bench through several builds of the library on ONE box (same-box A/B):
#   tools/ab_train.sh <tag> [lib ...]      (libs relative to nerf_amd/, default libnerfx.so first)
set -u
tag=$1; shift
for lib in libnerfx.so "$@" libnerfx.so; do
  NERFX_LIB=$PWD/nerf_amd/$lib timeout -k 10 120 python3 bench.py --mode train --steps 100 --warmup 20 > gpurun_out/ab_${tag}_$lib.json 2> gpurun_out/ab_${tag}_$lib.err || { echo "$lib failed"; tail -3 gpurun_out/ab_${tag}_$lib.err; exit 1; }
  python3 -c "
import json,sys; d=json.loads(open('gpurun_out/ab_${tag}_$lib.json').read().strip().splitlines()[-1]); s=d['step_ms']; print('$lib', round(d['ms_per_step'],4), 'min', round(s['min'],4), 'median', round(s['median'],4), 'loss', d['final_loss'])"
done
