"""Per-kernel, per-launch averages of every counter in one or more rocprofv3 counter-collection CSVs.

    python tools/pmc_sq_summary.py "<note>" pass1/x_counter_collection.csv [pass2/...csv ...] > profiles/...json

Each pass is a separate `rocprofv3 --pmc ...` run of the same command; counters are summed over the
shader engines / XCDs as rocprofv3 reports them."""
import collections, csv, json, re, sys

kernels = collections.OrderedDict()
for path in sys.argv[2:]:
    acc = collections.OrderedDict()
    for row in csv.DictReader(open(path)):
        if row["Kernel_Name"].startswith("__amd") or "at::" in row["Kernel_Name"]:
            continue
        name = re.sub(r"\(.*$", "", row["Kernel_Name"])
        s = acc.setdefault((name, row["Counter_Name"]), [0, 0.0])
        s[0] += 1
        s[1] += float(row["Counter_Value"])
    for (name, counter), (n, total) in acc.items():
        k = kernels.setdefault(name, collections.OrderedDict())
        k.setdefault("launches", n)
        k[counter] = total / n
for k in kernels.values():
    w = k.get("SQ_WAVES")
    if w:
        for c, short in (("SQ_INSTS_VALU", "valu_per_wave"), ("SQ_INSTS_SALU", "salu_per_wave"), ("SQ_INSTS_LDS", "lds_per_wave")):
            if c in k:
                k[short] = k[c] / w
    if "SQ_ACTIVE_INST_VALU" in k and "SQ_BUSY_CU_CYCLES" in k:
        # SQ_ACTIVE_INST_VALU counts per SIMD (x4 quad-cycles), SQ_BUSY_CU_CYCLES per CU
        k["valu_active_per_busy_cu_cycle"] = k["SQ_ACTIVE_INST_VALU"] / k["SQ_BUSY_CU_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in k and "SQ_LDS_IDX_ACTIVE" in k and k["SQ_LDS_IDX_ACTIVE"]:
        k["lds_bank_conflict_frac"] = k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"]
print(json.dumps({"note": sys.argv[1], "kernels": kernels}, indent=1))
