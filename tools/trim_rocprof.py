#!/usr/bin/env python3
"""Trim rocprofv3 CSV output to the rows of this library's kernels, so that the evidence fits under profiles/.

    python tools/trim_rocprof.py pmc   <dir with *_counter_collection.csv ...>  > profiles/rNN/pmc_counters.csv
    python tools/trim_rocprof.py stats <..._kernel_stats.csv>                    > profiles/rNN/..._kernel_stats.csv

pmc: one row per dispatch and counter of every pgt:: kernel (kernel name shortened, template arguments
kept), plus per-kernel summary rows (avg/min/max) — the inputs of roofline.traffic:
    HBM bytes per launch = 2 x FETCH_SIZE x 1024 (gfx950: FETCH_SIZE counts half of a wide coalesced
    read, MI355X_MICROARCH.md §HBM) + WRITE_SIZE x 1024.
stats: the rocprofv3 --stats summary with torch's kernels dropped except the five largest.
headline <trimmed pmc_counters.csv> <round> <command>: profiles/pmc_headline.json on stdout — the build kernel's average
    FETCH_SIZE / WRITE_SIZE per dispatch, the HBM bytes per launch they give and the SHA-256 of the kernel sources they
    were measured on (bench.py reports roofline.traffic from it only while that hash is the tree's)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("pgt::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    m = re.match(r"void (pgt::)?([A-Za-z0-9_]+(<.*?>)?)\(", name)
    return m.group(2) if m else name[:80]


def pmc(d):
    w = csv.writer(sys.stdout)
    w.writerow(["file", "dispatch_id", "kernel", "grid_size", "workgroup_size", "lds_block_size", "vgpr_count", "counter", "value",
                "duration_ns"])
    summ = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if "pgt::" not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            w.writerow([os.path.basename(f), r["Dispatch_Id"], k, r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"],
                        r["VGPR_Count"], r["Counter_Name"], r["Counter_Value"], dur])
            summ[(k, r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    w.writerow([])
    w.writerow(["summary", "kernel", "grid_size", "counter", "dispatches", "avg", "min", "max"])
    for (k, c, g), v in sorted(summ.items()):
        w.writerow(["summary", k, g, c, len(v), f"{sum(v) / len(v):.3f}", f"{min(v):.3f}", f"{max(v):.3f}"])


def stats(f):
    rows = list(csv.reader(open(f)))
    w = csv.writer(sys.stdout)
    w.writerow(rows[0])
    other = 0
    for r in rows[1:]:
        if "pgt::" in r[0]:
            w.writerow([short(r[0])] + r[1:])
        elif other < 5:
            w.writerow([r[0][:100]] + r[1:])
            other += 1


def headline(trimmed, rnd, command):
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("pgt_kernels.hip", "pgt_device.h"):   # = bench.kernel_source_sha256()
        h.update(open(os.path.join(root, "popgenomicstools_amd", "csrc", f), "rb").read())
    got = {}
    for r in csv.reader(open(trimmed)):
        if len(r) >= 8 and r[0] == "summary" and r[1].startswith("fst_build_kernel"):
            got[r[3]] = (float(r[5]), int(r[4]), r[1])
    fetch, nf, kernel = got["FETCH_SIZE"]
    write, nw, _ = got["WRITE_SIZE"]
    json.dump({"kernel": kernel, "kernel_source_sha256": h.hexdigest(), "collected": rnd, "command": command,
               "dispatches": min(nf, nw), "fetch_kib": round(fetch, 1), "write_kib": round(write, 1),
               "traffic_bytes_per_launch": (2 * round(fetch, 1) + round(write, 1)) * 1024.0,
               "algorithmic_bytes_per_launch": 16.0e9,
               "note": "HBM bytes = 2 x FETCH_SIZE KiB (gfx950 reports half of a wide coalesced read; MI355X_MICROARCH.md, "
                       "HBM / rocprofv3 section) + WRITE_SIZE KiB; separate --pmc passes"}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    if sys.argv[1] == "headline":
        headline(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        {"pmc": pmc, "stats": stats}[sys.argv[1]](sys.argv[2])
