"""Turns the two counter passes of the bench command into profiles/r1_bench_<prec>_pmc_hbm.json.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d OUT/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu
    python tools/pmc_summary.py OUT/pmc_fetch/f_counter_collection.csv OUT/pmc_write/w_counter_collection.csv > profiles/...json

Counter values are KB per dispatch.  On gfx950 FETCH_SIZE counts the 128-byte requests of wide
streaming reads as 64 bytes, so the fetch side is doubled (`fetch_corrected_KB`)."""
import collections, csv, json, re, sys


def per_kernel(path, counter):
    acc = collections.OrderedDict()
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter or row["Kernel_Name"].startswith("__amd") or "at::" in row["Kernel_Name"]:
            continue
        name = re.sub(r"\(.*$", "", row["Kernel_Name"])
        s = acc.setdefault(name, [0, 0.0])
        s[0] += 1
        s[1] += float(row["Counter_Value"])
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
kernels = {}
for name, (n, total) in fetch.items():
    w = write.get(name, [n, 0.0])
    kernels[name] = {"launches": n, "FETCH_SIZE_KB": total / n, "fetch_corrected_KB": 2 * total / n,
                     "WRITE_SIZE_KB": w[1] / max(w[0], 1)}
print(json.dumps({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 bench.py --steps 1 "
                          "--warmup 1 --no-cpu`; per-launch averages in KB as reported; on gfx950 FETCH_SIZE counts "
                          "128-B requests as 64 B for wide streaming reads, so fetch_corrected_KB = 2 x FETCH_SIZE",
                  "kernels": kernels}, indent=1))
