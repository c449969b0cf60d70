#!/bin/bash
# GPU box: SQ counters of the fused-MLP inference kernels under the bench command (three passes, each a run of its own) -> gpurun_out/k2pmc.json
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/k2pmc; mkdir -p $O
B="python3 bench.py --no-cpu --no-fp32 --no-train --no-other --no-roofline --steps 1 --warmup 1"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o a -- $B > /dev/null 2> $O/p1.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p2 -o b -- $B > /dev/null 2> $O/p2.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_MISC --output-format csv -d $O/p3 -o c -- $B > /dev/null 2> $O/p3.err || echo "p3 failed"
python3 tools/pmc_sq_summary.py "k2 counters" $(find $O/p1 -name '*counter_collection.csv') $(find $O/p2 -name '*counter_collection.csv') $(find $O/p3 -name '*counter_collection.csv') > gpurun_out/k2pmc.json
echo done
