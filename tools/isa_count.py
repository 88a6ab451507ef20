#!/usr/bin/env python3
"""Instruction counts by class of one kernel's assembly (hipcc -S --cuda-device-only): python tools/isa_count.py k.s <substring>."""
import re, sys
from collections import Counter
text = open(sys.argv[1]).read()
for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not all(w in name for w in sys.argv[2:]):
        continue
    ops = re.findall(r"^\s+([a-z]\w+)", body.split(".section")[0], re.M)
    c = Counter()
    for o in ops:
        if o.startswith("v_permlane"): c["permlane_swap"] += 1
        elif o.endswith("_dpp"): c["dpp"] += 1
        elif o.startswith("v_cndmask"): c["cndmask"] += 1
        elif o.startswith(("v_fma_f64", "v_add_f64", "v_mul_f64")): c[o] += 1
        elif o.startswith("v_mov") or o.startswith("v_accvgpr"): c["v_mov/accvgpr"] += 1
        elif o.startswith("v_"): c["other valu"] += 1
        elif o.startswith("s_waitcnt"): c["s_waitcnt"] += 1
        elif o.startswith("s_"): c["salu"] += 1
        elif o.startswith("ds_"): c["lds"] += 1
        elif o.startswith(("global_", "scratch_", "buffer_", "flat_")): c[o.split("_")[0] + "_" + o.split("_")[1]] += 1
    regs = dict(re.findall(r"\.amdhsa_(next_free_vgpr|accum_offset|private_segment_fixed_size) (\d+)", body))
    print(name[:70], "total", len(ops)); print("  ", dict(c)); print("  ", regs)
