"""popgenomicstools_amd — MI355X (gfx950) window-scan engine for PopGenomicsTools'
fstWindow / hetWindow / dxyWindow hot path.

Layout
  csrc/            HIP kernels (pgt_kernels.hip) + the C-ABI (pgt_api.cpp, pgt_windows.cpp)
  host/            the retained C++ hosts: reference argv + TSV, reduction in libpgtwin
  _lib.py          ctypes binding of include/pgtwin.h
  window_scan.py   host-side mirror of the three tools over numpy / torch buffers
  build.py         in-tree hipcc build
"""
from .window_scan import (  # noqa: F401
    Context,
    build_windows_bp,
    build_windows_extreme,
    build_windows_sites,
    dxy_window,
    fst_window,
    het_window,
    ihs_window,
    plan_shards,
    run_lengths,
    xpehh_window,
)

__all__ = ["Context", "build_windows_sites", "build_windows_bp", "fst_window", "het_window",
           "dxy_window", "ihs_window", "xpehh_window", "build_windows_extreme", "plan_shards", "run_lengths"]
