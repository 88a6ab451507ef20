#!/usr/bin/env python3
"""Process-to-process spread of the headline build kernel, with telemetry.

Runs `python bench.py --no-cpu --no-exchange-overhead` N times as CHILD processes (this parent never touches the GPU) and
tabulates, per process, the build kernel's fraction of the HBM peak next to what the GPU's own telemetry said during that
process' timed region and sustained leg: shader clock, socket power against the cap, the growth of the power-limiter
(PPT) residency counter, HBM / junction temperature, memory and fabric clocks.  Markdown on stdout.

    python tools/spread_series.py 10 > gpurun_out/spread_series.md
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def med(d, k, i=1):
    v = d.get(k)
    return v[("min", "median", "max")[i]] if isinstance(v, dict) else None


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    extra_args = sys.argv[2:]
    rows = []
    for i in range(n):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-exchange-overhead"] + extra_args,
                           capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print(f"run {i}: exit {r.returncode}: {r.stderr[-500:]}", file=sys.stderr)
            continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        t = d["extra"].get("telemetry", {})
        tr, su = t.get("timed_region", {}), t.get("sustained", {})
        g = lambda s, k: s.get("residency_growth", {}).get(k)  # noqa: E731
        rows.append({
            "run": i, "frac": d["roofline"]["frac"], "kernel_ms": d["roofline"]["kernel_ms"], "ms_per_step": d["ms_per_step"],
            "sust_GBs": d["extra"].get("sustained", {}).get("GB_per_s"),
            "fst1e8": d["extra"].get("fst_1e8", {}).get("roofline_frac"), "af8": d["extra"].get("af8_pairs28_1e8", {}).get("roofline_frac"),
            "t_sclk": med(tr, "current_gfxclks_min_over_units"), "t_sclk_max": med(tr, "current_gfxclks_max_over_units"),
            "t_power": med(tr, "current_socket_power"), "t_ppt": g(tr, "ppt_residency_acc"), "t_acc": g(tr, "accumulation_counter"),
            "t_hbm_c": med(tr, "temperature_mem", 2), "t_hot_c": med(tr, "temperature_hotspot", 2),
            "s_sclk": med(su, "current_gfxclks_min_over_units"), "s_sclk_max": med(su, "current_gfxclks_max_over_units"),
            "s_power": med(su, "current_socket_power"), "s_ppt": g(su, "ppt_residency_acc"), "s_acc": g(su, "accumulation_counter"),
            "s_hbm_c": med(su, "temperature_mem", 2), "s_hot_c": med(su, "temperature_hotspot", 2),
            "uclk": med(su, "current_uclk"), "fclk": med(su, "sysfs_fclk_mhz"),
            "other_thr": sum((g(su, k) or 0) for k in ("prochot_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc")),
            "start_power": t.get("at_start", {}).get("current_socket_power"), "start_hbm_c": t.get("at_start", {}).get("temperature_mem"),
            "pci": t.get("reader", {}).get("pci"),
            "serial": t.get("reader", {}).get("identity", {}).get("asic", {}).get("asic_serial"),
            "vbios": t.get("reader", {}).get("identity", {}).get("vbios", {}).get("version")})
    if not rows:
        sys.exit("no run succeeded")
    print(f"{len(rows)} processes of `python bench.py --no-cpu --no-exchange-overhead {' '.join(extra_args)}` back to back on one box "
          f"(GPU at PCI {rows[0]['pci']}, ASIC serial {rows[0]['serial']}, VBIOS {rows[0]['vbios']}); t_ = during the headline's timed region (0.5 s), s_ = during the sustained leg (2.5 s)\n")
    cols = ["run", "frac", "kernel_ms", "ms_per_step", "sust_GBs", "fst1e8", "af8", "t_sclk", "t_sclk_max", "t_power", "t_ppt", "t_acc", "t_hbm_c",
            "s_sclk", "s_sclk_max", "s_power", "s_ppt", "s_acc", "s_hbm_c", "s_hot_c", "uclk", "fclk", "other_thr", "start_power", "start_hbm_c"]
    print("| " + " | ".join(cols) + " |")
    print("|" + "---|" * len(cols))
    for r in rows:
        print("| " + " | ".join(("%.4f" % r[c] if isinstance(r[c], float) else str(r[c])) for c in cols) + " |")
    # correlation of frac with each numeric field
    import numpy as np
    fr = np.array([r["frac"] for r in rows])
    print("\nPearson correlation of `frac` with each field over the series (n = %d; |r| > 0.63 is p < 0.05 at n = 10):\n" % len(rows))
    for c in cols[2:]:
        xs = [r[c] for r in rows]
        if any(x is None for x in xs):
            continue
        x = np.array(xs, dtype=float)
        if x.std() == 0 or fr.std() == 0:
            print(f"- {c}: constant ({x[0]:g})")
        else:
            print(f"- {c}: r = {np.corrcoef(fr, x)[0, 1]:+.2f}  (range {x.min():g} … {x.max():g})")
    print(f"\nfrac: min {fr.min():.4f}, median {np.median(fr):.4f}, max {fr.max():.4f}")


if __name__ == "__main__":
    main()
