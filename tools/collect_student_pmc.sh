#!/bin/bash
# GPU box: SQ counters of the student's whole-ray kernel (two passes, each a run of its own) -> gpurun_out/student_pmc.json
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/student_pmc; mkdir -p $O
B="python3 tools/bench_student.py --frames 1"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o a -- $B > /dev/null 2> $O/p1.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/p2 -o b -- $B > /dev/null 2> $O/p2.err || exit 1
python3 tools/pmc_sq_summary.py "rocprofv3 --pmc (two passes, tools/collect_student_pmc.sh) over python3 tools/bench_student.py --frames 1: 800x800 frame of the 32x2 student, bf16, 64+64 samples, 65,536-ray calls; per-launch averages" $(find $O/p1 -name '*counter_collection.csv') $(find $O/p2 -name '*counter_collection.csv') > gpurun_out/student_pmc.json
echo done
