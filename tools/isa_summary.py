#!/usr/bin/env python3
"""Instruction mix of the library's kernels from `hipcc -S --cuda-device-only` output (no GPU needed):

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ipopgenomicstools_amd/csrc -S \\
          --cuda-device-only -o /tmp/k.s popgenomicstools_amd/csrc/pgt_kernels.hip
    python tools/isa_summary.py /tmp/k.s [substring of the kernel name ...]

Per kernel: loads / stores by width, cross-lane instructions, VGPRs / SGPRs, scratch.  What the verdicts check by hand
(16-byte nt loads, no mfma, no atomics, no spills)."""
import re
import sys
from collections import Counter


def main():
    text = open(sys.argv[1]).read()
    wanted = sys.argv[2:]
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if wanted and not any(w in name for w in wanted):
            continue
        ops = Counter(re.findall(r"^\s+((?:global|flat|buffer|scratch)_\w+|ds_\w+|v_mov_b32_dpp|v_\w+_dpp|v_mfma\w+|s_waitcnt)\b", body, re.M))
        nt = len(re.findall(r"^\s+global_load_\w+ .* nt\b", body, re.M))
        regs = dict(re.findall(r"\.amdhsa_(next_free_vgpr|next_free_sgpr|private_segment_fixed_size|group_segment_fixed_size) (\d+)", body))
        print(name)
        print("   ", ", ".join(f"{k} x{v}" for k, v in sorted(ops.items())), f"| nt loads {nt}")
        print("   ", regs)


if __name__ == "__main__":
    main()
