"""CPU: the C-ABI library loads without a GPU and exports every symbol include/pgtwin.h declares."""
import os
import re

import pytest

from popgenomicstools_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pgtwin.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/pgtwin.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared


def test_abi_version_and_struct_sizes():
    lib = _lib.load()
    assert lib.pgt_abi_version() == 6 == _lib.PGT_ABI_VERSION
    # struct sizes as the header lays them out
    assert _lib.WIN_DTYPE.itemsize == 32 and _lib.FST_ROW_DTYPE.itemsize == 40
    assert _lib.HET_ROW_DTYPE.itemsize == 32 and _lib.DXY_ROW_DTYPE.itemsize == 24


def test_tree_bytes_monotone_and_small():
    lib = _lib.load()
    for stat, col_bytes in ((_lib.PGT_STAT_FST, 16), (_lib.PGT_STAT_HET, 1), (_lib.PGT_STAT_DXY, 24)):
        prev = 0
        for n in (0, 1, 127, 128, 8192, 8193, 10**6, 10**8, 10**9):
            tb = lib.pgt_tree_bytes(stat, n)
            assert tb >= prev and tb % 256 == 0
            prev = tb
        assert lib.pgt_tree_bytes(stat, 10**9) < 0.02 * col_bytes * 10**9 + (1 << 20)
    assert lib.pgt_tree_bytes(7, 100) == 0


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product must refuse to run, loudly (no CPU path exists)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib.load()
    assert not lib.pgt_open(0)
    assert b"no CPU fallback" in lib.pgt_last_error(None)
    from popgenomicstools_amd import Context
    with pytest.raises(_lib.PgtError):
        Context()


def test_product_never_imports_oracle():
    """Nothing under the package or include/ may reference oracle/ (the checker)."""
    pkg = os.path.join(ROOT, "popgenomicstools_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "liboracle" not in text and "window_oracle" not in text and "oracle_bind" not in text, f


def test_header_is_plain_c(tmp_path):
    """include/pgtwin.h must compile as C99 (no C++-isms): it is what a C / cgo / FFI binding includes.
    A small C program that links the library also proves the symbols have C linkage."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "pgtwin.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    src = tmp_path / "t.c"
    src.write_text('#include "pgtwin.h"\n#include <stdio.h>\n'
                   'int main(void) {\n'
                   '  uint64_t runs[2] = {7, 5}; size_t n = 0;\n'
                   '  if (pgt_abi_version() != PGT_ABI_VERSION) return 2;\n'
                   '  if (pgt_build_windows_sites(runs, 2, 5, 2, NULL, 0, &n) != PGT_OK) return 3;\n'
                   '  if (pgt_build_windows_sites(runs, 2, 2, 5, NULL, 0, &n) != PGT_EARG) return 4;\n'
                   '  printf("%zu %s\\n", n, pgt_last_error(NULL));\n  return 0;\n}\n')
    exe = tmp_path / "t"
    pkg = os.path.join(ROOT, "popgenomicstools_amd")
    subprocess.run(["gcc", "-std=c99", str(src), "-I" + os.path.join(ROOT, "include"), "-L" + pkg, "-lpgtwin",
                    "-Wl,-rpath," + pkg, "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "1 <= step <= window" in r.stdout


def test_only_tests_smoke_and_bench_touch_the_oracle():
    """oracle/ is the checker: nothing outside tests/, __graft_entry__.smoke() and bench.py's
    cpu_baseline leg may import or execute it (tools/ included)."""
    for base in ("tools", "popgenomicstools_amd", "include"):
        for root, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".sh")):
                    text = open(os.path.join(root, f), errors="ignore").read()
                    assert "oracle_bind" not in text and "liboracle" not in text and "window_oracle" not in text, os.path.join(root, f)
    bench_src = open(os.path.join(ROOT, "bench.py")).read()
    # ONE import, inside checker_tools(); that function is called by the CPU-baseline leg and by the configs[2] / configs[4] rows checks only
    assert bench_src.count("import oracle_bind") == 1 and "def cpu_baseline" in bench_src
    tools_at = bench_src.index("def checker_tools")
    assert tools_at < bench_src.index("import oracle_bind") < bench_src.index("def check_rows_against_tsv")
    import re
    callers = [bench_src.rfind("\ndef ", 0, m.start()) for m in re.finditer(r"= checker_tools\(\)", bench_src)]
    names = sorted(re.match(r"\ndef (\w+)", bench_src[c:]).group(1) for c in callers)
    assert names == ["cpu_baseline", "het_rows_check", "pair_rows_check"], names
