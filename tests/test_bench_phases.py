"""CPU: bench.py's phase log and watchdog (what makes an unattended N > 1 run fail with a named phase instead of a
driver timeout), and the self-maintaining roofline.traffic figure."""
import json
import os
import subprocess
import sys
import textwrap

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(body, **env):
    code = "import sys, time, json\nsys.argv=['bench.py']\nimport bench\n" + textwrap.dedent(body)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=120,
                          env=dict(os.environ, PYTHONPATH=ROOT, **env))


def test_phase_overrun_exits_124_and_names_rank_and_phase():
    r = _run("""
        ph = bench.Phases(rank=3, world=8, scale=0.02)      # 'roofline': 30 s x 0.02 = 0.6 s
        with ph("roofline"):
            time.sleep(30)
        print("not reached")
    """)
    assert r.returncode == 124 and "not reached" not in r.stdout
    assert "[bench r3/8" in r.stderr and "phase 'roofline' exceeded its deadline" in r.stderr and "exit 124" in r.stderr


def test_degradable_phase_prints_the_secured_line_and_exits_0():
    r = _run("""
        ph = bench.Phases(rank=0, world=2, scale=0.1)       # 'peer timed': 6 s, rank 0 fires 3 s early
        ph.secure(lambda why: json.dumps({"secured": True, "why": why}))
        with ph("peer timed", degradable=True):
            time.sleep(30)
    """)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["secured"] is True and "peer timed" in line["why"]
    r = _run("""
        ph = bench.Phases(rank=1, world=2, scale=0.02)
        with ph("peer timed", degradable=True):
            time.sleep(30)
    """)
    assert r.returncode == 0 and r.stdout.strip() == ""     # other ranks leave quietly, code 0


def test_phases_in_time_are_logged_and_summed_and_fault_hook_fires():
    r = _run("""
        ph = bench.Phases(rank=0, world=1, scale=1.0)
        for _ in range(2):
            with ph("columns"):
                time.sleep(0.05)
        assert 0.09 < ph.taken["columns"] < 1.0
        print("fine")
    """)
    assert r.returncode == 0 and "fine" in r.stdout and r.stderr.count("columns: done in") == 2
    r = _run("""
        ph = bench.Phases(rank=1, world=2, scale=1.0)
        with ph("gather timed"):
            pass
    """, PGT_BENCH_FAULT="1:gather timed:die")
    assert r.returncode == 17 and "fault injected: die in phase 'gather timed'" in r.stderr


def test_every_phase_has_a_deadline_and_the_sum_fits_the_drivers_budget():
    multi = ("init", "columns", "gather timed", "verify", "roofline")
    assert sum(bench.PHASE_DEADLINES_S[p] for p in multi) <= 480   # the driver gives the whole command 600 s
    assert bench.PHASE_DEADLINES_S["peer timed"] + 2 * bench.PHASE_DEADLINES_S["verify"] + sum(
        bench.PHASE_DEADLINES_S[p] for p in ("init", "columns", "gather timed", "roofline")) <= 600


def test_pmc_traffic_follows_the_kernel_source_hash(tmp_path, monkeypatch):
    """roofline.traffic is the committed PMC figure only while the kernel sources hash to what it was measured on."""
    path = os.path.join(ROOT, "profiles", "pmc_headline.json")
    t, why = bench.pmc_traffic(10**8, 1)
    assert t is None and "headline workload only" in why
    if os.path.exists(path):
        rec = json.load(open(path))
        t, why = bench.pmc_traffic(10**9, 1)
        if rec["kernel_source_sha256"] == bench.kernel_source_sha256():
            assert t == rec["traffic_bytes_per_launch"] and 16.0e9 < t < 16.4e9
            assert abs(t - (2 * rec["fetch_kib"] + rec["write_kib"]) * 1024) < 1
        else:
            assert t is None and "other kernel sources" in why
    # a record measured on other sources is never reported
    fake = tmp_path / "profiles"
    fake.mkdir()
    (fake / "pmc_headline.json").write_text(json.dumps({"kernel_source_sha256": "0" * 64, "traffic_bytes_per_launch": 1.0}))
    (tmp_path / "popgenomicstools_amd").symlink_to(os.path.join(ROOT, "popgenomicstools_amd"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    t, why = bench.pmc_traffic(10**9, 1)
    assert t is None and "other kernel sources" in why


def test_rows_check_against_the_reference_tsv(tmp_path):
    """bench.py's live parity check (N = 1): rows of the WHOLE genome against the TSV the reference tool prints for a SAMPLE
    of it — also when the sample cuts a chromosome (the rows the reference prints after the last comparable window belong
    to its truncated tail and are ignored) — and a single differing count, coordinate or sixth digit is reported."""
    import numpy as np
    import oracle_bind
    import synth
    import popgenomicstools_amd as pgt
    from popgenomicstools_amd._lib import FST_ROW_DTYPE

    orc = oracle_bind.load()
    rng = np.random.default_rng(5)
    n, W, S = 60_000, 5_000, 1_000
    chr_ids, pos = synth.chromosomes(rng, n, 4, equal=False)
    a, b = synth.fst_columns(rng, n)
    run_len = np.bincount(chr_ids).astype(np.uint64)
    win = pgt.build_windows_sites(run_len, W, S)
    ref = orc.fst_scan(chr_ids, pos, a, b, W, S)
    assert ref.size == win.size
    table = np.zeros(win.size, dtype=FST_ROW_DTYPE)
    for f in ("start", "end", "mid", "n"):
        table[f] = ref[f]
    table["fst"] = ref["value"]
    binary = oracle_bind.ref_binary("fstWindow")
    for n_sample in (n, int(run_len[0] + run_len[1]), int(run_len[0] + run_len[1] // 2)):  # whole, two chromosomes, a cut one
        text, tsv = str(tmp_path / "in.txt"), str(tmp_path / "out.tsv")
        orc.write_fst_text(text, chr_ids[:n_sample], pos[:n_sample], a[:n_sample], b[:n_sample])
        if binary:
            with open(tsv, "w") as fh:
                subprocess.run([binary, text, str(W), str(S)], stdout=fh, check=True)
        else:
            assert orc.fst_text(text, W, S, tsv) == 0
        res = bench.check_rows_against_tsv(tsv, table, win, n_sample, "test")
        k = int(np.count_nonzero(win["hi"] <= n_sample))
        assert res["equal"] is True and res["windows"] == k and k > 10 and res["reference_rows"] >= k, res
        for field, delta in (("n", 1), ("end", 1), ("fst", 1e-5)):
            bad = table.copy()
            bad[field][k // 2] += type(bad[field][0])(delta) if field != "fst" else abs(bad["fst"][k // 2]) * delta + 1e-7
            res = bench.check_rows_against_tsv(tsv, bad, win, n_sample, "test")
            assert res["equal"] is False and f"row {k // 2}:" in res["mismatch"], (field, res)
        # within 1e-9 relative: still the same printed digits (or a rounding boundary, which is accepted)
        near = table.copy()
        near["fst"] *= 1 + 5e-10
        assert bench.check_rows_against_tsv(tsv, near, win, n_sample, "test")["equal"] is True


def test_rows_check_compares_fst_and_het_rows_with_a_reference_tsv(tmp_path):
    """bench.check_rows_against_tsv (the live parity checks of the headline run and, round 6, of configs[2]): rows of the windows
    that end inside the sample against the reference's TSV — coordinates, midpoint, count, label exact, the statistic as printed
    (`%g`) or within 1e-9 on a rounding boundary; a wrong count, a wrong value and a short TSV are reported, not passed."""
    import numpy as np
    from popgenomicstools_amd._lib import FST_ROW_DTYPE, HET_ROW_DTYPE, WIN_DTYPE
    win = np.zeros(4, dtype=WIN_DTYPE)
    win["lo"], win["hi"], win["label_run"] = [0, 10, 20, 30], [20, 30, 40, 60], [0, 0, 1, 1]
    rows = np.zeros(4, dtype=FST_ROW_DTYPE)
    rows["start"], rows["end"], rows["mid"], rows["n"] = [1, 11, 21, 31], [20, 30, 40, 60], [10, 20, 30, 45], [20, 20, 20, 30]
    rows["fst"] = [0.25, -4e-05, 1.25e6, 0.5]
    tsv = tmp_path / "ref.tsv"
    lines = ["chr1\t1\t20\t10\t0.25\t20", "chr1\t11\t30\t20\t-4e-05\t20", "chr2\t21\t40\t30\t1.25e+06\t20", "chr2\t99\t99\t99\t9\t9"]
    tsv.write_text("\n".join(lines) + "\n")
    res = bench.check_rows_against_tsv(str(tsv), rows.view(np.uint8), win, 40, "a reference")  # windows 0..2 end inside 40 sites
    assert res["equal"] is True and res["windows"] == 3 and res["reference_rows"] == 4 and res["fst_on_a_rounding_boundary"] == 0
    bad = rows.copy()
    bad["n"][1] = 19
    res = bench.check_rows_against_tsv(str(tsv), bad.view(np.uint8), win, 40, "a reference")
    assert res["equal"] is False and "row 1" in res["mismatch"]
    bad = rows.copy()
    bad["fst"][2] = 1.26e6
    assert bench.check_rows_against_tsv(str(tsv), bad.view(np.uint8), win, 40, "a reference")["equal"] is False
    near = rows.copy()
    near["fst"][0] = 0.25 * (1 + 3e-10)  # prints 0.25 too
    assert bench.check_rows_against_tsv(str(tsv), near.view(np.uint8), win, 40, "a reference")["equal"] is True
    tsv.write_text("\n".join(lines[:2]) + "\n")
    res = bench.check_rows_against_tsv(str(tsv), rows.view(np.uint8), win, 40, "a reference")
    assert res["equal"] is False and "2 rows" in res["mismatch"]
    # a TAIL sample: the windows that begin at or behind `first_site` against ALL rows the reference printed for that sample
    tsv.write_text("\n".join(lines[2:3] + ["chr2\t31\t60\t45\t0.5\t30"]) + "\n")
    res = bench.check_rows_against_tsv(str(tsv), rows.view(np.uint8), win, 60, "a reference", first_site=20)
    assert res["equal"] is True and res["windows"] == 2 and res["reference_rows"] == 2
    tsv.write_text(lines[2] + "\n")  # the reference printed one row less than the table has windows there: not a pass
    res = bench.check_rows_against_tsv(str(tsv), rows.view(np.uint8), win, 60, "a reference", first_site=20)
    assert res["equal"] is False and "tail sample" in res["mismatch"]
    # het rows: the value is h, the count column is `nonmissing` (hetWindow.cpp:87)
    h = np.zeros(4, dtype=HET_ROW_DTYPE)
    h["start"], h["end"], h["mid"], h["nonmissing"], h["nhet"] = rows["start"], rows["end"], rows["mid"], [18, 20, 0, 5], [9, 5, 0, 1]
    h["h"] = [0.5, 0.25, 0.0, 0.2]
    tsv.write_text("chr1\t1\t20\t10\t0.5\t18\nchr1\t11\t30\t20\t0.25\t20\nchr2\t21\t40\t30\t0\t0\n")
    res = bench.check_rows_against_tsv(str(tsv), h.view(np.uint8), win, 40, "a reference", row_dtype=HET_ROW_DTYPE, value="h", count="nonmissing")
    assert res["equal"] is True and res["windows"] == 3 and "h_on_a_rounding_boundary" in res


def test_telemetry_reader_degrades_to_nothing_and_summarises_samples():
    """gpu_telemetry: without a driver (this container) every reader is absent and nothing raises; the sampler's summary gives
    min / median / max per field, per-unit lists as min / max over the units, and the GROWTH of the limiter residency counters."""
    import gpu_telemetry as g
    t = g.Telemetry("0000:00:00.0")
    d = t.describe()
    assert d["source"] in (["none"], ["sysfs"]) or "amdsmi" in d["source"]
    assert isinstance(t.snapshot(), dict) and "t" in t.snapshot()
    assert g._current_dpm("0: 500Mhz\n1: 2208Mhz *\n2: 2400Mhz") == 2208 and g._current_dpm("S: 95Mhz *\n0: 500Mhz") == 95
    assert g._current_dpm("") is None and g._current_dpm("0: 2000Mhz") is None
    assert g._num(0xFFFF) is None and g._num("N/A") is None and g._num(47) == 47
    s = g._Sampler(t, 1.0)
    s.samples = [{"t": 10.0, "current_socket_power": 1200, "current_gfxclks": [2300, 2350], "ppt_residency_acc": 100, "accumulation_counter": 1000,
                  "sysfs": {"sclk_mhz": 2340}},
                 {"t": 10.5, "current_socket_power": 1350, "current_gfxclks": [2310, 2390], "ppt_residency_acc": 160, "accumulation_counter": 1500,
                  "sysfs": {"sclk_mhz": 2356}},
                 {"t": 11.0, "current_socket_power": 1300, "current_gfxclks": [2320, 2360], "ppt_residency_acc": 400, "accumulation_counter": 2000,
                  "sysfs": {"sclk_mhz": 2350}}]
    out = s.summary()
    assert out["samples"] == 3 and out["seconds"] == 1.0
    assert out["current_socket_power"] == {"min": 1200, "median": 1300, "max": 1350, "n": 3}
    assert out["current_gfxclks_min_over_units"]["min"] == 2300 and out["current_gfxclks_max_over_units"]["max"] == 2390
    assert out["sysfs_sclk_mhz"]["median"] == 2350
    assert out["residency_growth"] == {"ppt_residency_acc": 300, "accumulation_counter": 1000}
    assert g._Sampler(t, 1.0).summary() == {"samples": 0}


def test_library_banners_do_not_reach_stdout_and_the_secured_line_still_does():
    """gloo and RCCL print banners on stdout with C stdio; bench.py owes stdout ONE line.  stdout_to_stderr() sends file descriptor 1
    to 2 for the duration (C-level writes included) — and a degradable phase that overruns INSIDE it still puts its secured line on
    the process' real stdout."""
    r = _run("""
        import os, ctypes
        libc = ctypes.CDLL(None)
        with bench.stdout_to_stderr():
            libc.puts(b"BANNER from C stdio")           # what RCCL does
            os.write(1, b"BANNER from a raw write\\n")
            print("BANNER from python")
        print("the one line", flush=True)
    """)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().splitlines() == ["the one line"], r.stdout
    assert r.stderr.count("BANNER") == 3
    r = _run("""
        ph = bench.Phases(rank=0, world=1, scale=0.1)       # 'exchange overhead': 120 s x 0.1 = 12 s, rank 0 fires 3 s early
        ph.secure(lambda why: json.dumps({"secured": True, "why": why}))
        with ph("exchange overhead", degradable=True), bench.stdout_to_stderr():
            print("BANNER")
            time.sleep(60)
    """)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and json.loads(lines[0])["secured"] is True, r.stdout
    assert "BANNER" in r.stderr
