"""Per-kernel means of the counters in rocprofv3 --pmc counter_collection CSVs (one row per dispatch and counter):
    python tools/pmc_sq.py gpurun_out/<tag>/pass*_counter_collection.csv [--json out.json]
Prints one line per kernel with the mean of every counter over its dispatches (the first 5 dispatches of a kernel are
warm-up and left out), the launch count and, when SQ_BUSY_CYCLES and SQ_INSTS_VALU are both present, the fraction of the
kernel's SIMD-cycles in which a VALU instruction occupied the issue port: 4 * SQ_INSTS_VALU / (SQ_BUSY_CYCLES / 32 * 1024) --
a wave64 VALU instruction holds its SIMD's port for 4 cycles, SQ_BUSY_CYCLES counts shader cycles summed over the 32 shader
engines (it reproduces the kernel's duration at 2.1-2.4 GHz), there are 1024 SIMDs."""
import collections
import csv
import json
import re
import sys

SKIP = 5


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    if out_json:
        args.remove(out_json)
    data = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in args:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                m = re.search(r"(\w+_kernel)", row["Kernel_Name"])
                if m:
                    data[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    table = {}
    for k, ctrs in data.items():
        n = min(len(v) for v in ctrs.values())
        if n <= SKIP:
            continue
        table[k] = {c: sum(v[SKIP:]) / len(v[SKIP:]) for c, v in ctrs.items()}
        table[k]["launches"] = n - SKIP
    order = sorted(table, key=lambda k: -table[k].get("SQ_WAVE_CYCLES", table[k].get("SQ_INSTS_VALU", 0)))
    for k in order:
        t = table[k]
        extra = ""
        if "SQ_BUSY_CYCLES" in t and "SQ_INSTS_VALU" in t and t["SQ_BUSY_CYCLES"] > 0:
            simd_cycles = t["SQ_BUSY_CYCLES"] / 32.0 * 1024.0
            t["valu_issue_frac"] = 4.0 * t["SQ_INSTS_VALU"] / simd_cycles
            extra = f"  valu_issue_frac {t['valu_issue_frac']:.3f}"
        print(f"{k:34s} " + "  ".join(f"{c} {v:.4g}" for c, v in sorted(t.items()) if c not in ("valu_issue_frac",)) + extra)
    if out_json:
        with open(out_json, "w") as fh:
            import importlib
            import os
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            native = importlib.import_module("iclr2025_3d-mom_amd._native")
            json.dump({"skip_first": SKIP, "kernels": table, "lib_version": native.lib().mom_version().decode(),
                       "workload": "200k Gaussians, 60 frames, 960x540, HexPlane on"}, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
