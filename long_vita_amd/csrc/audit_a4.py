#!/usr/bin/env python3
"""Build-time check of the kernels that lay the accumulator registers out by hand (attn_fwd_a4.hip, attn_bwd_dkdv_g.hip).

The kernel's inline asm addresses a[0:255] by number, so the compiler must keep out of them: its own code (everything
outside the ASMSTART / ASMEND brackets) may contain no v_accvgpr_* and no scratch access, the kernels must have no
private segment (a value that does not fit the vector half would show there), and the loop must not copy a V fragment
between its LDS read and the MFMA that waits for it.  Usage: audit_a4.py <assembly file> [kernel name, default lv_fwd_a4_kernel]"""
import re
import sys


def main(path, kernel="lv_fwd_a4_kernel"):
    text = open(path).read()
    bad = []
    in_asm = False
    for n, line in enumerate(text.splitlines(), 1):
        if "ASMSTART" in line:
            in_asm = True
        elif "ASMEND" in line:
            in_asm = False
        elif not in_asm and re.search(r"v_accvgpr_|scratch_|Folded Spill|Folded Reload", line):
            bad.append(f"{path}:{n}: compiler-generated {line.strip()}")
    kernels = re.findall(r"\.name:\s+(\S*" + re.escape(kernel) + r"\S*)\s+\.private_segment_fixed_size:\s+(\d+)", text)
    if not kernels:
        bad.append(f"no {kernel} metadata found")
    for name, size in kernels:
        if int(size) != 0:
            bad.append(f"{name}: private segment of {size} bytes (spill)")
    spills = re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)
    if any(int(s) for s in spills):
        bad.append(f"vgpr spills: {spills}")
    if bad:
        print("\n".join(bad))
        return 1
    print(f"audit_a4: {len(kernels)} {kernel} kernels, no compiler use of the accumulator registers, no spill")
    return 0


if __name__ == "__main__":
    sys.exit(main(*sys.argv[1:3]))
