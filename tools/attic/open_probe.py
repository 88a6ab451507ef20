#!/usr/bin/env python3
"""Where HIP start-up goes (the floor of every CLI run): a fresh process without torch times hipInit, the
first hipSetDevice + hipFree(0) (context creation), then pgt_open (device properties, the dynamic-LDS attributes of the
build kernels = loading the library's code objects, three events) and a first tiny launch.  Markdown on stdout."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ms(t0):
    return (time.perf_counter() - t0) * 1e3


def main():
    t_all = time.perf_counter()
    t0 = time.perf_counter()
    hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
    t_dl = ms(t0)
    t0 = time.perf_counter()
    rc = hip.hipInit(0)
    t_init = ms(t0)
    n = C.c_int(0)
    t0 = time.perf_counter()
    hip.hipGetDeviceCount(C.byref(n))
    t_count = ms(t0)
    t0 = time.perf_counter()
    hip.hipSetDevice(0)
    hip.hipFree(None)
    t_ctx = ms(t0)
    t0 = time.perf_counter()
    lib = C.CDLL(os.path.join(ROOT, "popgenomicstools_amd", "libpgtwin.so"))
    t_lib = ms(t0)
    lib.pgt_open.restype = C.c_void_p
    t0 = time.perf_counter()
    ctx = lib.pgt_open(0)
    t_open = ms(t0)
    assert ctx, "pgt_open failed"
    t0 = time.perf_counter()
    ctx2 = lib.pgt_open(0)
    t_open2 = ms(t0)
    p = C.c_void_p()
    lib.pgt_dev_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    t0 = time.perf_counter()
    lib.pgt_dev_alloc(C.c_void_p(ctx), 1 << 20, C.byref(p))
    t_alloc = ms(t0)
    lib.pgt_rowbuf_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p]
    t0 = time.perf_counter()
    lib.pgt_rowbuf_fill(C.c_void_p(ctx), p, 1 << 20, 7, None)
    hip.hipDeviceSynchronize()
    t_launch = ms(t0)
    t0 = time.perf_counter()
    lib.pgt_rowbuf_fill(C.c_void_p(ctx), p, 1 << 20, 7, None)
    hip.hipDeviceSynchronize()
    t_launch2 = ms(t0)
    print(f"rc hipInit {rc}, {n.value} device(s)\n")
    print("| step | ms |\n|---|---|")
    for k, v in (("dlopen libamdhip64", t_dl), ("hipInit", t_init), ("hipGetDeviceCount", t_count), ("hipSetDevice + hipFree(0)", t_ctx),
                 ("dlopen libpgtwin", t_lib), ("pgt_open (first)", t_open), ("pgt_open (second context)", t_open2),
                 ("first hipMalloc through pgt_dev_alloc", t_alloc), ("first launch + sync", t_launch), ("second launch + sync", t_launch2),
                 ("all of the above", ms(t_all))):
        print(f"| {k} | {v:.1f} |")


if __name__ == "__main__":
    main()
