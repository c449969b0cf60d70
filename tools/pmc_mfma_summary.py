"""Matrix-pipe counters of the fused MLP kernels of one bench command -> a profiles/*.json summary.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE \\
              --output-format csv -d OUT/pmc_mfma -o m -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-fp32 --no-train
    rocprofv3 --kernel-trace --output-format csv -d OUT/trace -o t -- <the same command>
    python tools/pmc_mfma_summary.py OUT/pmc_mfma/m_counter_collection.csv OUT/trace/t_kernel_trace.csv > profiles/...json

GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (32 per
v_mfma_f32_32x32x16_bf16, 16 per v_mfma_f32_16x16x32_bf16); one MOPS unit is 512 FLOP.  The kernel time comes
from the kernel trace of a separate pass of the same command (counters and traces are never collected together)."""
import collections
import csv
import json
import sys

want = ("mlp_fwd16_kernel", "mlp_fwd_kernel")
acc = collections.defaultdict(float)
launches = 0
for row in csv.DictReader(open(sys.argv[1])):
    if not any(w in row["Kernel_Name"] for w in want):
        continue
    acc[row["Counter_Name"]] += float(row["Counter_Value"])
    launches += row["Counter_Name"] == "GRBM_GUI_ACTIVE"
seconds = 0.0
traced = 0
for row in csv.DictReader(open(sys.argv[2])):
    if any(w in row["Kernel_Name"] for w in want):
        seconds += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
        traced += 1
scale = launches / traced if traced else 0.0          # the two passes may differ in warm-up launches
seconds *= scale
cus, simds = 256, 4
gui = acc["GRBM_GUI_ACTIVE"] / 8.0                     # per-XCD average = shader-clock cycles while the kernels ran
out = {
    "note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE over "
            "`python3 bench.py --steps 1 --warmup 1 --no-cpu --no-fp32 --no-train`, summed over the fused-MLP launches; "
            "kernel seconds from a kernel trace of a separate pass of the same command",
    "launches": launches, "kernel_seconds": seconds, "counters": dict(acc),
    "clock_GHz_during_kernels": gui / seconds / 1e9 if seconds else None,
    "mfma_busy_fraction_of_simd_cycles": acc["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * cus * simds) if gui else None,
    "executed_TFLOP": acc["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / 1e12,
}
print(json.dumps(out, indent=1))
