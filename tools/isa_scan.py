#!/usr/bin/env python
"""Disassemble the gfx950 code objects that SHIP inside lib/libmom4d.so and count instructions per kernel.

    python tools/isa_scan.py                      # per-kernel totals: VALU / MFMA / DS / VMEM / SALU, registers
    python tools/isa_scan.py render_bwd           # mnemonic histogram of the kernels whose name contains the pattern
    python tools/isa_scan.py --packed             # every packed-fp32 instruction (v_pk_*_f32) by kernel

The library holds one clang offload bundle per translation unit (section .hip_fatbin); each bundle's gfx950 entry is an
ELF code object that llvm-objdump disassembles.  tests/test_isa.py uses this to keep packed fp32 arithmetic out of
deform_field.hip's kernels (the packed-fp32 / bf16-MFMA hazard)."""
import collections
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "iclr2025_3d-mom_amd", "lib", "libmom4d.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path=LIB, arch="gfx950"):
    """The device ELF images of every offload bundle in `path` whose target triple names `arch`."""
    data = open(path, "rb").read()
    out, at = [], 0
    while True:
        at = data.find(MAGIC, at)
        if at < 0:
            return out
        n = struct.unpack_from("<Q", data, at + len(MAGIC))[0]
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if arch in triple and size:
                out.append(data[at + off:at + off + size])
        at += len(MAGIC)


def disassemble(image):
    """{kernel symbol: [mnemonic, ...]} of one code object."""
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(image)
        f.flush()
        txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             text=True, check=True).stdout
    kernels, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\b", line)
        if m and cur is not None:
            cur.append(m.group(1))
    return kernels


def classify(mn):
    if mn.startswith("v_mfma") or mn.startswith("v_smfmac"):
        return "mfma"
    if mn.startswith("v_"):
        return "valu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mn.startswith("s_"):
        return "salu"
    return "other"


def all_kernels(path=LIB):
    ks = {}
    for image in code_objects(path):
        ks.update(disassemble(image))
    return ks


def packed_fp32(ks):
    """{kernel: Counter of v_pk_*_f32 mnemonics} for the kernels that have any."""
    out = {}
    for name, insts in ks.items():
        c = collections.Counter(i for i in insts if re.match(r"v_pk_(fma|mul|add)_f32", i))
        if c:
            out[name] = c
    return out


def main():
    ks = all_kernels()
    args = sys.argv[1:]
    if args and args[0] == "--packed":
        for name, c in sorted(packed_fp32(ks).items()):
            print(name, dict(c))
        return
    pat = args[0] if args else None
    for name in sorted(ks):
        if name.endswith(".kd") or (pat and pat not in name):
            continue
        insts = ks[name]
        tot = collections.Counter(classify(i) for i in insts)
        print(f"{name}: {len(insts)} instructions  " + "  ".join(f"{k} {tot[k]}" for k in ("valu", "mfma", "lds", "vmem", "salu", "other")))
        if pat:
            for mn, n in collections.Counter(insts).most_common(60):
                print(f"    {n:6d}  {mn}")


if __name__ == "__main__":
    main()
