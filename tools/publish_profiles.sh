#!/bin/bash
# Build container: turns what tools/collect_profiles.sh and tools/collect_student_pmc.sh left under
# gpurun_out/ into the committed summaries profiles/<round>_*.   tools/publish_profiles.sh r3
set -e
cd "$(dirname "$0")/.."
R=${1:-r3}; O=gpurun_out/prof; P=profiles
cp $O/bf16/k_kernel_stats.csv $P/${R}_bench_bf16_kernel_stats.csv
cp $O/fp32/k_kernel_stats.csv $P/${R}_bench_fp32_kernel_stats.csv
cp $O/train/k_kernel_stats.csv $P/${R}_train_bf16_kernel_stats.csv
cp $O/hbm/k_kernel_stats.csv $P/${R}_hbm_kernels_kernel_stats.csv
cp $O/hbm_kernels.jsonl $P/${R}_hbm_kernels.jsonl
grep '^{' $O/bench_bf16_under_rocprof.json | tail -1 > $P/${R}_bench_bf16_under_rocprof.json
grep '^{' $O/train_bf16_under_rocprof.json | tail -1 > $P/${R}_train_bf16_under_rocprof.json
grep '^{' $O/bench_full.json | tail -1 > $P/${R}_bench_full.json
python3 tools/pmc_summary.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv > $P/${R}_bench_bf16_pmc_hbm.json
python3 tools/pmc_summary.py $O/train_fetch/f_counter_collection.csv $O/train_write/w_counter_collection.csv \
    | sed 's/python3 bench.py --steps 1 --warmup 1 --no-cpu/python3 bench.py --mode train --steps 5 --warmup 2/' > $P/${R}_train_bf16_pmc_hbm.json
python3 tools/pmc_mfma_summary.py $O/pmc_mfma/m_counter_collection.csv $O/trace1/t_kernel_trace.csv > $P/${R}_bench_bf16_pmc_mfma.json
python3 tools/pmc_sq_summary.py "rocprofv3 --pmc (two passes, tools/collect_profiles.sh) over python3 tools/bench_kernels.py --iters 2: 262,144 rays; per-launch averages" \
    $O/sq1/a_counter_collection.csv $O/sq2/b_counter_collection.csv > $P/${R}_small_kernels_pmc_sq.json
[ -f gpurun_out/student_pmc.json ] && cp gpurun_out/student_pmc.json $P/${R}_student_pmc.json
[ -f $O/student.json ] && grep '^{' $O/student.json | tail -1 > $P/${R}_student_bench.json
echo published
