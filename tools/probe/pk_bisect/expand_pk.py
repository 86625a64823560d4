"""ISA-level bisection of the packed-fp32 / bf16-MFMA corruption.

deform_field.hip is compiled to gfx950 assembly WITH the SLP vectoriser (the failing build: 1000 wrong launches of 1000); this script
rewrites chosen v_pk_{add,mul,fma}_f32 of deform_field_fwd_b3_kernel into the two scalar instructions they stand for (same operands,
same order of operations, so the same bits when nothing is wrong), leaves every other instruction and the schedule alone, and
assembles the result back into a library:

    python tools/probe/pk_bisect/expand_pk.py build <name> <spec>      ->  iclr2025_3d-mom_amd/lib/var/pk_<name>.so
        spec: "none" | "all" | comma-separated 1-based line ranges "a-b" of the kernel's listing (pk_b3_listing.s) whose packed
              instructions are expanded; a leading "!" expands everything OUTSIDE the ranges
    python tools/probe/pk_bisect/expand_pk.py list                     ->  the packed instructions with their listing lines

v_pk semantics (CDNA3 ISA 6.x "packed math"): operand i feeds the LOW lane of the result from its half op_sel[i] (default 0 = low)
and the HIGH lane from its half op_sel_hi[i] (default 1 = high); neg_lo / neg_hi negate it in the low / high computation."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CS = os.path.join(ROOT, "iclr2025_3d-mom_amd", "csrc")
LL = "/opt/rocm/lib/llvm/bin"
WORK = os.environ.get("PK_WORK", "/tmp/pk")
KERNEL = "deform_field_fwd_b3_kernel"


def sh(cmd):
    r = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        raise SystemExit(f"{cmd}\n{r.stdout[-3000:]}")
    return r.stdout


def device_asm():
    os.makedirs(WORK, exist_ok=True)
    out = os.path.join(WORK, "df_slp.s")
    if not os.path.exists(out):
        sh(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fslp-vectorize -I{ROOT}/include --cuda-device-only -S {CS}/deform_field.hip -o {out}")
    lines = open(out).read().split("\n")
    a = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and KERNEL in l and l.rstrip().endswith(":") or (KERNEL in l and re.match(r"^_ZN\S+:", l)))
    b = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines, a, b


PK = re.compile(r"^\s*v_pk_(add|mul|fma)_f32\s+(.*)$")


def parse(line):
    m = PK.match(line)
    if not m:
        return None
    op, rest = m.group(1), m.group(2)
    mods = dict(re.findall(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([0-9,]+)\]", rest))
    rest = re.sub(r"\s*(op_sel|op_sel_hi|neg_lo|neg_hi):\[[0-9,]+\]", "", rest).split(";")[0].strip()
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    ops.append(cur.strip())
    n = len(ops) - 1
    g = lambda k, d: [int(x) for x in mods[k].split(",")] if k in mods else [d] * n
    return op, ops[0], ops[1:], g("op_sel", 0), g("op_sel_hi", 1), g("neg_lo", 0), g("neg_hi", 0)


def half(src, hi):
    m = re.match(r"^([vs])\[(\d+):(\d+)\]$", src)
    if m:
        return f"{m.group(1)}{int(m.group(2)) + (1 if hi else 0)}"
    return src                                    # inline constant / literal: the same value in both halves


def expand(line):
    """Two scalar instructions for one packed one, or None if the register overlap needs a temporary (left packed, counted)."""
    p = parse(line)
    if p is None:
        return None
    op, dst, srcs, sel, sel_hi, nlo, nhi = p
    d = re.match(r"^v\[(\d+):(\d+)\]$", dst)
    dlo, dhi = f"v{d.group(1)}", f"v{int(d.group(1)) + 1}"
    lo_src = [half(s, sel[i]) for i, s in enumerate(srcs)]
    hi_src = [half(s, sel_hi[i]) for i, s in enumerate(srcs)]
    fmt = lambda regs, neg: ", ".join(("-" if neg[i] else "") + r for i, r in enumerate(regs))
    mnem = {"add": "v_add_f32_e64", "mul": "v_mul_f32_e64", "fma": "v_fma_f32"}[op]
    ilo = f"\t{mnem} {dlo}, {fmt(lo_src, nlo)}"
    ihi = f"\t{mnem} {dhi}, {fmt(hi_src, nhi)}"
    if dlo not in hi_src:
        return [ilo + "\t; pk-lo", ihi + "\t; pk-hi"]
    if dhi not in lo_src:
        return [ihi + "\t; pk-hi", ilo + "\t; pk-lo"]
    return None


def select(spec, n_lines):
    if spec == "none":
        return lambda i: False
    if spec == "all":
        return lambda i: True
    inv = spec.startswith("!")
    rs = [tuple(int(x) for x in r.split("-")) if "-" in r else (int(r), int(r)) for r in spec.lstrip("!").split(",")]
    inside = lambda i: any(a <= i <= b for a, b in rs)
    return (lambda i: not inside(i)) if inv else inside


def build(name, spec):
    lines, a, b = device_asm()
    want = select(spec, b - a)
    out, done, kept = list(lines), 0, 0
    for i in range(a, b):
        if PK.match(lines[i]) and want(i - a + 1):
            e = expand(lines[i])
            if e is None:
                kept += 1
            else:
                out[i] = "\n".join(e)
                done += 1
    left = sum(1 for i in range(a, b) if re.match(r"^\s*v_pk_(add|mul|fma)_f32", out[i]))
    src = os.path.join(WORK, f"pk_{name}.s")
    open(src, "w").write("\n".join(out))
    obj, co, fb, host = (os.path.join(WORK, f"pk_{name}.{e}") for e in ("o", "out", "hipfb", "host.o"))
    sh(f"{LL}/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c {src} -o {obj}")
    sh(f"{LL}/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o {co} {obj}")
    sh(f"{LL}/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 "
       f"-input=/dev/null -input={co} -output={fb}")
    sh(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I{ROOT}/include --cuda-host-only -c {CS}/deform_field.hip "
       f"-Xclang -fcuda-include-gpubinary -Xclang {fb} -o {host}")
    var = os.path.join(ROOT, "iclr2025_3d-mom_amd", "lib", "var")
    os.makedirs(var, exist_ok=True)
    objs = " ".join(os.path.join(ROOT, "iclr2025_3d-mom_amd", "lib", "obj", f) for f in sorted(os.listdir(os.path.join(ROOT, "iclr2025_3d-mom_amd", "lib", "obj")))
                    if f.endswith(".o") and f != "deform_field.o")
    sh(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o {var}/pk_{name}.so {objs} {host}")
    print(f"pk_{name}: spec {spec}: expanded {done}, overlap-kept {kept}, packed fp32 arithmetic left in the kernel {left}")


if __name__ == "__main__":
    if sys.argv[1] == "list":
        lines, a, b = device_asm()
        open(os.path.join(WORK, "pk_b3_listing.s"), "w").write("\n".join(lines[a:b]))
        for i in range(a, b):
            if PK.match(lines[i]):
                print(i - a + 1, lines[i].strip(), "" if expand(lines[i]) else "   <- overlap")
    else:
        build(sys.argv[2], sys.argv[3])
