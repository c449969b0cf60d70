#!/bin/bash
# GPU box: collects the round's profile evidence into gpurun_out/prof (copy what is judged into profiles/).
# Every step has its own time limit; counters and traces are never collected in one pass.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O
B="python3 bench.py --no-cpu --no-fp32 --no-train --no-other"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16 -o k -- $B --steps 10 --warmup 2 > $O/bench_bf16_under_rocprof.json 2> $O/bf16.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -o t -- $B --steps 1 --warmup 1 > /dev/null 2> $O/trace1.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o m -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_mfma.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_fetch.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B --steps 1 --warmup 1 > /dev/null 2> $O/pmc_write.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp32 -o k -- python3 bench.py --no-cpu --no-fp32 --no-train --no-other --prec fp32 --steps 3 --warmup 1 > $O/bench_fp32_under_rocprof.json 2> $O/fp32.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o k -- python3 bench.py --mode train --steps 20 --warmup 5 > $O/train_bf16_under_rocprof.json 2> $O/train.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/train_fetch -o f -- python3 bench.py --mode train --steps 5 --warmup 2 > /dev/null 2> $O/train_fetch.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/train_write -o w -- python3 bench.py --mode train --steps 5 --warmup 2 > /dev/null 2> $O/train_write.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hbm -o k -- python3 tools/bench_kernels.py > $O/hbm_kernels_under_rocprof.jsonl 2> $O/hbm.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq1 -o a -- python3 tools/bench_kernels.py --iters 2 > /dev/null 2> $O/sq1.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/sq2 -o b -- python3 tools/bench_kernels.py --iters 2 > /dev/null 2> $O/sq2.err || exit 1
timeout -k 10 200 python3 bench.py > $O/bench_full.json 2> $O/bench_full.err || exit 1
timeout -k 10 120 python3 tools/bench_kernels.py > $O/hbm_kernels.jsonl 2>/dev/null || exit 1
timeout -k 10 100 python3 tools/bench_student.py --frames 20 > $O/student.json 2> $O/student.err || exit 1
echo done
