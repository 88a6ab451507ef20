"""Build libpgtwin.so (HIP kernels + C-ABI) and the retained C++ hosts for gfx950, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container as well as on the
GPU box.  Nothing here falls back to a CPU implementation: if the build fails, the package
cannot be used.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
BIN = os.path.join(PKG, "bin")
LIB = os.path.join(PKG, "libpgtwin.so")
FLAGS = os.path.join(PKG, "libpgtwin.flags")

LIB_SOURCES = ["pgt_kernels.hip", "pgt_af_kernels.hip", "pgt_ingest.hip", "pgt_api.cpp", "pgt_windows.cpp"]
HOST_TOOLS = ["fstWindow", "hetWindow", "dxyWindow", "ihsWindow", "xpehhWindow"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libpgtwin needs the ROCm toolchain (no CPU fallback exists)")
    return exe


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]} (exit {r.returncode})")


def _stamp(flags: list[str], deps: list[str]) -> str:
    """Compile flags + content hash of every source and header: what the library was built from."""
    import hashlib
    h = hashlib.sha256()
    for d in deps:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read() + b"\0")
    return " ".join(flags) + "\n" + h.hexdigest() + "\n"


def build_lib(force: bool = False) -> str:
    """(Re)build libpgtwin.so when the sources, headers or compile flags differ from the ones it was
    built from (content hash + flags recorded beside it in libpgtwin.flags; independent of mtimes,
    which a snapshot copy to the GPU box does not preserve).  Concurrent callers (the ranks of one
    job) are serialised by a file lock and the library is replaced atomically.  Without hipcc (a box
    that only received a prebuilt library) an existing library is used as it is."""
    import fcntl
    srcs = [os.path.join(CSRC, s) for s in LIB_SOURCES]
    deps = srcs + [os.path.join(CSRC, "pgt_internal.h"), os.path.join(CSRC, "pgt_device.h"), os.path.join(ROOT, "include", "pgtwin.h")]
    extra = os.environ.get("PGT_EXTRA_HIPCC_FLAGS", "").split()
    if "-DPGT_TUNING_BUILD" in extra:
        # the tuning library (tools/tune_*.py): tools/pgt_kernels_tuning.hip REPLACES csrc/pgt_kernels.hip — it includes the
        # product kernels textually and adds the measured-and-rejected variants; nothing under csrc/ knows about it
        tuning = os.path.join(ROOT, "tools", "pgt_kernels_tuning.hip")
        srcs = [tuning if os.path.basename(s) == "pgt_kernels.hip" else s for s in srcs]
        deps += [tuning, os.path.join(ROOT, "tools", "pgt_build_experiments.inc")]
        extra = extra + ["-I" + os.path.join(ROOT, "tools")]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"] + extra
    stamp = _stamp(flags, deps)

    def stale() -> bool:
        try:
            return not os.path.exists(LIB) or open(FLAGS).read() != stamp
        except OSError:
            return True

    if not force and not stale():
        return LIB
    have_hipcc = bool(shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"))
    if not force and os.path.exists(LIB) and not have_hipcc:
        return LIB
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or stale():  # another rank may have built it while we waited
            tmp = f"{LIB}.{os.getpid()}.tmp"
            _run([_hipcc()] + flags + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", tmp] + srcs)
            os.replace(tmp, LIB)
            with open(FLAGS, "w") as f:
                f.write(stamp)
    return LIB


def build_hosts(force: bool = False) -> list[str]:
    """The retained C++ hosts: same argv and TSV as the reference tools, reduction in libpgtwin."""
    os.makedirs(BIN, exist_ok=True)
    out = []
    common = [os.path.join(HOST, "host_common.h"), os.path.join(HOST, "extreme_common.h"), os.path.join(ROOT, "include", "pgtwin.h"), LIB]
    for tool in HOST_TOOLS:
        src = os.path.join(HOST, tool + "_main.cpp")
        if not os.path.exists(src):
            continue
        exe = os.path.join(BIN, tool)
        if force or _newer(exe, [src] + common):
            _run([_hipcc(), "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + HOST, src,
                  "-o", exe, "-L" + PKG, "-lpgtwin", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN/.."])
        out.append(exe)
    return out


def build_all(force: bool = False) -> None:
    build_lib(force)
    build_hosts(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built", LIB)
