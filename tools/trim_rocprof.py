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
    were measured on (bench.py reports roofline.traffic from it only while that hash is the tree's).
timeline <..._kernel_trace.csv>: the library's kernels in launch order with their durations and the idle gap in front of each
    (end of the previous kernel of the process -> start of this one), then per kernel name the median duration and median gap:
    what a step costs beyond its build kernel (query, tree_up, launch gaps).
traffic <trimmed pmc csv> <sites>: per build kernel HBM bytes (2 x FETCH_SIZE + WRITE_SIZE) against the algorithmic bytes,
    and the SQ wave-cycle split (waiting on memory / issue stalls / executing), as a markdown table."""
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


def timeline(f):
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    prev_end = None
    per = defaultdict(lambda: ([], []))
    print("| # | kernel | grid | start us | duration us | gap before us |\n|---|---|---|---|---|---|")
    t0 = None
    shown = 0
    for r in rows:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"]
        mine = "pgt::" in name
        if mine:
            k = short(name)
            if t0 is None:
                t0 = st
            gap = (st - prev_end) / 1e3 if prev_end is not None else float("nan")
            per[k][0].append((en - st) / 1e3)
            if prev_end is not None:
                per[k][1].append(gap)
            if shown < 400:
                print(f"| {shown} | {k[:60]} | {r.get('Grid_Size', '')} | {(st - t0) / 1e3:.1f} | {(en - st) / 1e3:.1f} | {gap:.1f} |")
                shown += 1
        prev_end = en
    import statistics
    print("\n| kernel | launches | median duration us | median gap before us |\n|---|---|---|---|")
    for k, (d, g) in per.items():
        print(f"| {k[:70]} | {len(d)} | {statistics.median(d):.1f} | {statistics.median(g) if g else float('nan'):.1f} |")


ALG_BYTES_PER_SITE = {"fst_build_kernel": 16, "dxy_build_kernel": 24, "het_build_kernel_w4": 1, "het_build_kernel": 1, "dxy_het_build_kernel": 26,
                      "af_build_kernel<8": 64, "af_build_kernel_w1<8": 64, "af_build_kernel<2": 16, "ext_build_kernel": 8}


def traffic(trimmed, sites):
    n = float(sites)
    d = defaultdict(dict)
    for r in csv.reader(open(trimmed)):
        if len(r) >= 8 and r[0] == "summary" and "build" in r[1]:
            d[(r[1], r[2])][r[3]] = float(r[5])
    print(f"{n:.0e} sites; HBM bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (every byte of every build kernel is read by a 16-byte load, the "
          "width the x2 is calibrated for); SQ columns: shares of SQ_WAVE_CYCLES\n")
    print("| kernel | grid | algorithmic GB | fetched GB | written GB | traffic / algorithmic | waiting (SQ_WAIT_ANY) | issue stalls (SQ_WAIT_INST_ANY) | "
          "executing (SQ_ACTIVE_INST_ANY) | of which VALU | VALU issue share of the SIMDs | L2 hit rate |\n|---|---|---|---|---|---|---|---|---|---|---|---|")
    for (k, g), c in sorted(d.items()):
        base = next((b for key, b in ALG_BYTES_PER_SITE.items() if k.startswith(key)), None)
        if base is None or "FETCH_SIZE" not in c:
            continue
        pairs = 1
        if k.startswith("fst_build_kernel") and int(g) > 200000:  # grid.y = pairs: pmc_kernels.py batches 6
            pairs = round(int(g) / 130304) if n == 1e8 else 6
        alg = base * n * pairs
        fetched, written = 2 * c["FETCH_SIZE"] * 1024, c.get("WRITE_SIZE", 0.0) * 1024
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        def share(x):
            return f"{c[x] / wc * 100:.1f} %" if wc and x in c else "-"
        simd = "-"
        if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]:
            simd = f"{c['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / (c['GRBM_GUI_ACTIVE'] / 8) * 100:.0f} %"
        hit = f"{c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']) * 100:.1f} %" if "TCC_HIT_sum" in c else "-"
        print(f"| {k[:44]} | {g} | {alg / 1e9:.3f} | {fetched / 1e9:.3f} | {written / 1e9:.3f} | **{(fetched + written) / alg:.3f}** | {share('SQ_WAIT_ANY')} | "
              f"{share('SQ_WAIT_INST_ANY')} | {share('SQ_ACTIVE_INST_ANY')} | {share('SQ_ACTIVE_INST_VALU')} | {simd} | {hit} |")


if __name__ == "__main__":
    if sys.argv[1] == "timeline":
        timeline(sys.argv[2])
        sys.exit(0)
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3])
        sys.exit(0)
    if sys.argv[1] == "headline":
        headline(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        {"pmc": pmc, "stats": stats}[sys.argv[1]](sys.argv[2])
