"""Per-kernel means of the counters in rocprofv3 --pmc counter_collection CSVs (one row per dispatch and counter):
    python tools/pmc_sq.py gpurun_out/<tag>/pass*_counter_collection.csv [--json out.json] [--workload "<bench.py config name>"]
Prints one line per kernel with the mean of every counter over its dispatches (the first 5 dispatches of a kernel are
warm-up and left out), the launch count, the mean duration (the rows carry start / end timestamps) and, when GRBM_GUI_ACTIVE
and SQ_INSTS_VALU are both present:
    clock_ghz        = GRBM_GUI_ACTIVE / 8 / duration     (the counter is summed over the 8 XCDs: the guide's effective-clock
                       formula; it reads a little HIGH on dispatches shorter than 0.3 ms, which makes the fraction below read low)
    valu_issue_frac  = 2 * SQ_INSTS_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
the launch's vector instructions against the SIMDs' peak issue rate, one wave64 instruction per 2 cycles (the rate behind the
157 TFLOP/s fp32 vector peak): at most 1 by construction.  (Rounds 1-3 charged 4 cycles per instruction and divided by
SQ_BUSY_CYCLES / 32: that read 1.05 on render_bwd at c2 and reads 1.12 at c3 -- simple VOP2 instructions of different waves do
retire faster than one per 4 cycles, tools/probe/valu_rate.hip.)  Without GRBM_GUI_ACTIVE the old figure is printed as
valu_issue_frac_sq_busy."""
import collections
import csv
import json
import re
import sys

SKIP = 5


def main():
    argv = sys.argv[1:]
    out_json = workload = None
    if "--json" in argv:
        i = argv.index("--json")
        out_json = argv[i + 1]
        del argv[i:i + 2]
    if "--workload" in argv:
        i = argv.index("--workload")
        workload = argv[i + 1]
        del argv[i:i + 2]
    data = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for path in argv:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                m = re.search(r"(\w+_kernel)", row["Kernel_Name"])
                if m:
                    data[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    if row.get("End_Timestamp"):
                        dur[m.group(1)][row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    table = {}
    for k, ctrs in data.items():
        n = min(len(v) for v in ctrs.values())
        if n <= SKIP:
            continue
        table[k] = {c: sum(v[SKIP:]) / len(v[SKIP:]) for c, v in ctrs.items()}
        table[k]["launches"] = n - SKIP
        d = [v for _, v in sorted(dur[k].items(), key=lambda kv: int(kv[0]))][SKIP:]
        if d:
            table[k]["duration_us"] = sum(d) / len(d) / 1e3
    order = sorted(table, key=lambda k: -table[k].get("SQ_WAVE_CYCLES", table[k].get("SQ_INSTS_VALU", 0)))
    for k in order:
        t = table[k]
        extra = ""
        if "GRBM_GUI_ACTIVE" in t and t["GRBM_GUI_ACTIVE"] > 0:
            cyc = t["GRBM_GUI_ACTIVE"] / 8.0                       # cycles of the launch
            if "duration_us" in t:
                t["clock_ghz"] = cyc / (t["duration_us"] * 1e3)
                extra += f"  clock {t['clock_ghz']:.2f} GHz"
            if "SQ_INSTS_VALU" in t:
                t["valu_issue_frac"] = 2.0 * t["SQ_INSTS_VALU"] / (1024.0 * cyc)
                extra += f"  valu_issue_frac {t['valu_issue_frac']:.3f}"
        elif "SQ_BUSY_CYCLES" in t and "SQ_INSTS_VALU" in t and t["SQ_BUSY_CYCLES"] > 0:
            t["valu_issue_frac_sq_busy"] = 4.0 * t["SQ_INSTS_VALU"] / (t["SQ_BUSY_CYCLES"] / 32.0 * 1024.0)
            extra = f"  valu_issue_frac_sq_busy {t['valu_issue_frac_sq_busy']:.3f}"
        skip = ("valu_issue_frac", "valu_issue_frac_sq_busy", "clock_ghz")
        print(f"{k:34s} " + "  ".join(f"{c} {v:.4g}" for c, v in sorted(t.items()) if c not in skip) + extra)
    if out_json:
        with open(out_json, "w") as fh:
            import importlib
            import os
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            native = importlib.import_module("iclr2025_3d-mom_amd._native")
            json.dump({"skip_first": SKIP, "kernels": table, "lib_version": native.lib().mom_version().decode(),
                       "workload": workload or "200k Gaussians, 60 frames, 960x540, HexPlane on"}, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
