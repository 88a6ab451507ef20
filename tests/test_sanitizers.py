"""CPU: the product's host-only window builders + shard plan and the oracle, compiled natively with
AddressSanitizer and UndefinedBehaviorSanitizer and run as a differential fuzzer (SURVEY.md §5:
sanitizers on the CPU build only — GPU ASan is not available on this pool)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_window_builders_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "fuzz_windows")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"),
           "-I" + os.path.join(ROOT, "popgenomicstools_amd", "csrc")]
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
    obj = str(tmp_path / "oracle.o")
    subprocess.run(["gcc", "-std=c11", "-c", os.path.join(ROOT, "oracle", "window_oracle.c"), "-o", obj] + san + inc, check=True)
    subprocess.run(["g++", "-std=c++17", os.path.join(ROOT, "oracle", "fuzz_windows.cpp"),
                    os.path.join(ROOT, "popgenomicstools_amd", "csrc", "pgt_windows.cpp"), obj, "-o", exe, "-lm"] + san + inc,
                   check=True)
    r = subprocess.run([exe, "3000", "2026"], capture_output=True, text=True, timeout=500,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "all equal" in r.stdout


@pytest.mark.timeout(600)
def test_host_ingest_under_asan_ubsan(tmp_path):
    """The hosts' own ingest code (host_common.h: number conversion, the two-pass parallel table parser, the
    binary column cache) under ASan + UBSan: tests/host_parse_check.cpp."""
    exe = str(tmp_path / "host_parse_check")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
    subprocess.run(["g++", "-std=c++17", os.path.join(ROOT, "tests", "host_parse_check.cpp"), "-o", exe,
                    "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "popgenomicstools_amd", "host"),
                    "-lz", "-lpthread"] + san, check=True)
    cache = tmp_path / "cache"
    cache.mkdir()
    r = subprocess.run([exe, "200000", "2026", str(cache)], capture_output=True, text=True, timeout=500,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "all equal" in r.stdout
