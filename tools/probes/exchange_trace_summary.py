#!/usr/bin/env python3
"""Summary of tools/probes/exchange_trace.py's kernel trace (rocprofv3 --kernel-trace ... csv): python exchange_trace_summary.py <kernel_trace.csv>"""
import csv
import statistics
import sys


def main():
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size", ""), r.get("Workgroup_Size", ""),
           r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", ""))) for r in rows]
    builds = [i for i, k in enumerate(ks) if "fst_build_kernel" in k[0]]
    # halves: split the build launches at the largest gap between consecutive builds (the marker + RowExchange set-up)
    gaps = [(ks[builds[i + 1]][1] - ks[builds[i]][2], i) for i in range(len(builds) - 1)]
    cut = max(gaps)[1] + 1
    med = statistics.median
    print("| half | build launches | build us (median) | query us | period us (build start to build start) | idle before build us | other kernels inside the period |")
    print("|---|---|---|---|---|---|---|")
    for name, idx in (("local", builds[:cut]), ("gather", builds[cut:])):
        idx = idx[5:]  # warm
        bd = [(ks[i][2] - ks[i][1]) / 1e3 for i in idx]
        per = [(ks[idx[j + 1]][1] - ks[idx[j]][1]) / 1e3 for j in range(len(idx) - 1)]
        qd, idle, others = [], [], {}
        for j in range(len(idx) - 1):
            inside = ks[idx[j] + 1: idx[j + 1]]
            for k in inside:
                if "query" in k[0]:
                    qd.append((k[2] - k[1]) / 1e3)
                else:
                    o = others.setdefault(k[0][:80], {"n": 0, "dur": [], "after_build_start": [], "grid": k[3], "wg": k[4], "lds": k[5], "overlap_next_build": []})
                    o["n"] += 1
                    o["dur"].append((k[2] - k[1]) / 1e3)
                    o["after_build_start"].append((k[1] - ks[idx[j]][1]) / 1e3)
                    o["overlap_next_build"].append(max(0.0, (k[2] - ks[idx[j + 1]][1]) / 1e3))
            last_end = max(k[2] for k in ks[idx[j]: idx[j + 1]] if "pgt::" in k[0])
            idle.append((ks[idx[j + 1]][1] - last_end) / 1e3)
        desc = "; ".join(f"{n} x{o['n']}: {med(o['dur']):.1f} us each, starts {med(o['after_build_start']):.1f} us after the build's start, grid {o['grid']} wg {o['wg']} lds {o['lds']}, "
                         f"runs {med(o['overlap_next_build']):.1f} us into the next build" for n, o in others.items()) or "none"
        print(f"| {name} | {len(idx)} | {med(bd):.1f} | {med(qd) if qd else float('nan'):.1f} | {med(per):.1f} | {med(idle):.1f} | {desc} |")


if __name__ == "__main__":
    main()
