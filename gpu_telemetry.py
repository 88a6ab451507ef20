"""GPU clock / power / temperature / throttle telemetry for bench.py (measurement plumbing, not product).

Read WITHOUT touching HIP and without starting a process: the amdsmi Python binding (libamd_smi reads the driver's
gpu_metrics table through sysfs) with a plain-sysfs fallback (hwmon + pp_dpm_* of the card whose PCI address HIP reports
for the device).  Why it exists: the same binary has read 80 … 87 % of the HBM peak from process to process on this pool
("what's weak" 4) — flat for the life of a process, different in the next — and nothing on record said
whether clocks, the power cap or temperature differ between such processes.

    tel = Telemetry(pci_bus_id)          # e.g. "0000:05:00.0" from torch.cuda.get_device_properties(i).pci_bus_id … or None
    before = tel.snapshot()
    with tel.sampling(period_s=0.05) as s:   # a daemon thread: reads while the GPU works
        ... timed region ...
    during = s.summary()
    after = tel.snapshot()

Every reader is wrapped: a field that cannot be read is simply absent, `source` says what worked, and nothing here can
fail a bench run.
"""
from __future__ import annotations

import glob
import os
import threading
import time


def _read(path):
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return None


def _current_dpm(text):
    """'0: 500Mhz\\n1: 2208Mhz *' -> 2208 (the starred level), None when unreadable."""
    if not text:
        return None
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            try:
                return int(line.split(":")[1].strip().split("M")[0])
            except (IndexError, ValueError):
                return None
    return None


# accumulated throttler residencies of gpu_metrics (counters: the DIFFERENCE over a region says whether that limiter was
# active in it) and the instantaneous fields worth recording
_ACC_FIELDS = ("prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
               "hbm_thm_residency_acc", "accumulation_counter", "gfxclk_lock_status")
_SCALARS = ("temperature_hotspot", "temperature_mem", "temperature_vrsoc", "average_socket_power", "current_socket_power",
            "current_uclk", "average_uclk_frequency", "average_gfx_activity", "average_umc_activity", "throttle_status",
            "indep_throttle_status", "pcie_link_speed", "pcie_link_width", "firmware_timestamp", "system_clock_counter",
            "energy_accumulator")
_LISTS = ("current_gfxclks", "current_socclks", "gfx_busy_inst", "xcp_stats")


def _num(x):
    """amdsmi marks unsupported fields 'N/A' or with all-ones integers."""
    if isinstance(x, bool) or x is None:
        return None
    if isinstance(x, (int, float)):
        if x in (0xFFFF, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF):
            return None
        return x
    return None


class Telemetry:
    def __init__(self, pci_bus_id: str | None = None):
        self.source = []
        self.errors = []
        self.handle = None
        self.smi = None
        self.card = None   # /sys/class/drm/cardN/device of the GPU
        self.hwmon = None
        self._lock = threading.Lock()
        bdf = (pci_bus_id or "").lower()
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            handles = amdsmi.amdsmi_get_processor_handles()
            pick = None
            for h in handles:
                try:
                    if bdf and amdsmi.amdsmi_get_gpu_device_bdf(h).lower() == bdf:
                        pick = h
                except Exception:  # noqa: BLE001
                    pass
            if pick is None and len(handles) == 1:
                pick = handles[0]
            if pick is not None:
                self.smi, self.handle = amdsmi, pick
                self.source.append("amdsmi")
                if not bdf:
                    try:
                        bdf = amdsmi.amdsmi_get_gpu_device_bdf(pick).lower()
                    except Exception:  # noqa: BLE001
                        pass
            else:
                self.errors.append(f"amdsmi: {len(handles)} GPUs visible, none matches {bdf or '(no PCI address given)'}")
        except Exception as e:  # noqa: BLE001 — absent library, no permission, …
            self.errors.append(f"amdsmi: {type(e).__name__}: {e}")
        self.bdf = bdf or None
        if bdf:
            for dev in glob.glob("/sys/class/drm/card*/device"):
                try:
                    if os.path.basename(os.path.realpath(dev)).lower() == bdf:
                        self.card = dev
                        hw = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
                        self.hwmon = hw[0] if hw else None
                        self.source.append("sysfs")
                        break
                except OSError:
                    pass

    # ---- one reading --------------------------------------------------------------------------------------
    def snapshot(self, light: bool = False) -> dict:
        """One reading.  light=True: only what the sampler thread needs (clocks, power, temperatures)."""
        out = {"t": time.time()}
        with self._lock:
            if self.handle is not None:
                try:
                    m = self.smi.amdsmi_get_gpu_metrics_info(self.handle)
                    for k in _SCALARS:
                        v = _num(m.get(k))
                        if v is not None:
                            out[k] = v
                    for k in _ACC_FIELDS:
                        v = _num(m.get(k))
                        if v is not None:
                            out[k] = v
                    for k in ("current_gfxclks", "current_socclks"):
                        v = [_num(x) for x in (m.get(k) or [])]
                        v = [x for x in v if x is not None]
                        if v:
                            out[k] = v
                except Exception as e:  # noqa: BLE001
                    out["amdsmi_error"] = f"{type(e).__name__}: {e}"
                if not light:
                    try:
                        cap = self.smi.amdsmi_get_power_cap_info(self.handle)
                        out["power_cap_w"] = round(float(cap.get("power_cap", 0)) * 1e-6, 1)
                    except Exception:  # noqa: BLE001
                        pass
                    try:
                        out["perf_level"] = str(self.smi.amdsmi_get_gpu_perf_level(self.handle))
                    except Exception:  # noqa: BLE001
                        pass
                    try:  # which limiters are ACTIVE right now (the residency counters above say how long they were)
                        vio = self.smi.amdsmi_get_violation_status(self.handle)
                        out["active_limiters"] = sorted(k[7:] for k, x in vio.items() if k.startswith("active_") and x is True)
                    except Exception:  # noqa: BLE001
                        pass
            if self.card is not None:
                s = {}
                for f in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
                    v = _current_dpm(_read(os.path.join(self.card, f)))
                    if v is not None:
                        s[f[7:] + "_mhz"] = v
                if self.hwmon:
                    for f, key, scale in (("power1_input", "power_w", 1e-6), ("power1_average", "power_avg_w", 1e-6),
                                          ("power1_cap", "power_cap_w", 1e-6), ("temp2_input", "junction_c", 1e-3),
                                          ("temp3_input", "mem_c", 1e-3), ("freq1_input", "sclk_hz_mhz", 1e-6),
                                          ("freq2_input", "mclk_hz_mhz", 1e-6)):
                        v = _read(os.path.join(self.hwmon, f))
                        if v is not None:
                            try:
                                s[key] = round(int(v) * scale, 3)
                            except ValueError:
                                pass
                if not light:
                    for f in ("power_dpm_force_performance_level", "current_link_speed", "current_link_width", "gpu_busy_percent",
                              "mem_busy_percent"):
                        v = _read(os.path.join(self.card, f))
                        if v is not None:
                            s[f] = v
                out["sysfs"] = s
        return out

    # ---- a sampler thread over a region ---------------------------------------------------------------------
    def sampling(self, period_s: float = 0.05):
        return _Sampler(self, period_s)

    def describe(self) -> dict:
        """What was read and from where, plus the card's identity (which physical GPU, its memory vendor, firmware, partition
        modes): box-to-box differences of the same binary are larger than process-to-process ones on this pool, so a line
        should say which card produced it."""
        out = {"source": self.source or ["none"], "pci": self.bdf, "errors": self.errors}
        ident = {}
        if self.handle is not None:
            for key, fn in (("asic", "amdsmi_get_gpu_asic_info"), ("vram", "amdsmi_get_gpu_vram_info"), ("vbios", "amdsmi_get_gpu_vbios_info"),
                            ("board", "amdsmi_get_gpu_board_info"), ("compute_partition", "amdsmi_get_gpu_compute_partition"),
                            ("memory_partition", "amdsmi_get_gpu_memory_partition"), ("driver", "amdsmi_get_gpu_driver_info")):
                try:
                    v = getattr(self.smi, fn)(self.handle)
                    if isinstance(v, dict):
                        v = {k: x for k, x in v.items() if isinstance(x, (int, float, str, bool)) and x not in ("N/A", "")}
                    ident[key] = v if isinstance(v, (dict, int, float, str)) else str(v)
                except Exception:  # noqa: BLE001
                    pass
        if self.card is not None:
            for f in ("unique_id", "vbios_version", "current_compute_partition", "current_memory_partition", "mem_info_vram_total",
                      "mem_info_vram_vendor"):
                v = _read(os.path.join(self.card, f))
                if v is not None:
                    ident["sysfs_" + f] = v
        if ident:
            out["identity"] = ident
        return out


def _stats(xs):
    xs = sorted(xs)
    n = len(xs)
    return {"min": xs[0], "median": xs[n // 2], "max": xs[-1], "n": n}


class _Sampler:
    def __init__(self, tel: Telemetry, period_s: float):
        self.tel, self.period = tel, period_s
        self.samples = []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            self.samples.append(self.tel.snapshot(light=True))
            self._stop.wait(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._th.join(2.0)
        return False

    def summary(self) -> dict:
        """min / median / max of every numeric field over the region + the growth of the throttler residency counters."""
        s = self.samples
        if not s:
            return {"samples": 0}
        out = {"samples": len(s), "seconds": round(s[-1]["t"] - s[0]["t"], 3)}
        series = {}
        for snap in s:
            for k, v in snap.items():
                if k in ("t",) or k in _ACC_FIELDS:
                    continue
                if isinstance(v, (int, float)):
                    series.setdefault(k, []).append(v)
                elif isinstance(v, list) and v and all(isinstance(x, (int, float)) for x in v):
                    series.setdefault(k + "_min_over_units", []).append(min(v))
                    series.setdefault(k + "_max_over_units", []).append(max(v))
                elif k == "sysfs" and isinstance(v, dict):
                    for kk, vv in v.items():
                        if isinstance(vv, (int, float)):
                            series.setdefault("sysfs_" + kk, []).append(vv)
        for k, xs in series.items():
            out[k] = _stats(xs)
        grow = {}
        for k in _ACC_FIELDS:
            if k in s[0] and k in s[-1]:
                grow[k] = s[-1][k] - s[0][k]
        if grow:
            out["residency_growth"] = grow
        return out


def pci_bus_id_of(torch_device_index: int) -> str | None:
    """The PCI address HIP reports for the device, in sysfs spelling (0000:bb:dd.f); None if torch cannot tell."""
    try:
        import torch
        p = torch.cuda.get_device_properties(torch_device_index)
        dom = getattr(p, "pci_domain_id", 0)
        return f"{dom:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:  # noqa: BLE001
        return None


if __name__ == "__main__":
    import json
    import sys
    t = Telemetry(sys.argv[1] if len(sys.argv) > 1 else None)
    print(json.dumps({"describe": t.describe(), "snapshot": t.snapshot()}, indent=1, default=str))
