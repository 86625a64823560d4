#!/usr/bin/env python
"""Register and LDS budgets of the shipped kernels, from the code objects' metadata (llvm-readelf --notes):
    python tools/kernel_budgets.py [pattern ...]
VGPR + AGPR per lane decide how many waves a SIMD holds (512 per lane and SIMD on gfx950); static LDS is in the metadata, the dynamic
part (deform_bwd_b3g: 151 872 B, deform_field_fwd_b3: its fragment tables) is set at launch.  DESIGN.md section 3.6 uses this table
for the question "can the HexPlane gather live inside the MLP backward's workgroups?"."""
import re
import subprocess
import sys
import tempfile

import isa_scan


def budgets(patterns):
    rows = []
    for image in isa_scan.code_objects():
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(image)
            f.flush()
            txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name or (patterns and not any(p in name.group(1) for p in patterns)):
                continue
            g = lambda k: int(re.search(rf"\.{k}:\s+(\S+)", blk).group(1))
            short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name.group(1))
            short = re.match(r"[A-Za-z_0-9]+?_kernel(ILb[01]E)?", short)
            regs = g("vgpr_count")          # gfx90a+: the unified count, accumulation registers (.agpr_count) included
            rows.append((short.group(0) if short else name.group(1)[:40], regs, g("sgpr_count"), g("group_segment_fixed_size"),
                         g("private_segment_fixed_size"), 512 // max(regs, 1)))
    return sorted(set(rows))


if __name__ == "__main__":
    print(f"{'kernel':44s} {'registers':>9s} {'sgpr':>5s} {'static LDS':>10s} {'scratch':>8s} {'waves/SIMD by registers':>24s}")
    for r in budgets(sys.argv[1:]):
        print(f"{r[0]:44s} {r[1]:9d} {r[2]:5d} {r[3]:10d} {r[4]:8d} {min(r[5], 8):24d}")
