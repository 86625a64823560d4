"""Summarise two rocprofv3 --pmc passes into profiles/<prefix>_pmc_traffic.{json,_per_kernel.csv}.

Collect on the GPU box (two SEPARATE passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2), with the program
directly after `--`:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- \
        python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- \
        python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline

then here:

    python tools/pmc_traffic.py gpurun_out/pmc_fetch/*/*_counter_collection.csv \
        gpurun_out/pmc_write/*/*_counter_collection.csv profiles/r01_f

Corrections (MI355X_MICROARCH.md, HBM section): counter values are KiB; on gfx950 FETCH_SIZE reports half of the bytes
of a coalesced streaming read, so it is doubled; WRITE_SIZE is exact.  The script prints the calibration against
adam_kernel, whose byte counts are known exactly (16 B read and 12 B written per optimised float), and refuses to
write a summary when that calibration is off by more than 3 %.  bench.py reads the newest summary whose workload
matches and reports it as roofline.traffic.
"""
import collections
import csv
import json
import re
import os
import sys

# kernel symbol -> the name bench.py / profiling.py use
NAMES = {
    "render_bwd_kernel": "render_bwd", "render_fwd_kernel": "render_fwd", "hexplane_bwd5_gather_kernel": "hexplane_bwd_gather",
    "hexplane_bwd5_scatter_kernel": "hexplane_bwd_scatter", "hexplane_bwd6_gather_kernel": "hexplane_bwd_gather",
    "hexplane_fwd4_kernel": "hexplane_fwd", "adam_kernel": "adam", "l1_kernel": "l1_loss",
    "preprocess_fwd_kernel": "preprocess_fwd", "preprocess_bwd_kernel": "preprocess_bwd", "tile_sort_kernel": "tile_sort",
    "deform_fwd_kernel": "mlp_fwd", "deform_bwd_dx_kernel": "mlp_bwd_dx", "deform_bwd_dw_kernel": "mlp_bwd_dw",
    "plane_reg_kernel": "plane_reg", "deform_field_fwd_b3_kernel": "deform_field_fwd", "deform_field_fwd_kernel": "deform_field_fwd_f32",
    "hexplane_lines_kernel": "hexplane_lines", "tile_hist_kernel": "tile_hist", "tile_scatter_kernel": "tile_scatter",
    "deform_bwd_b3f_kernel": "mlp_bwd", "deform_bwd_b3g_kernel": "mlp_bwd", "deform_bwd_reduce_kernel": "mlp_bwd_reduce", "act_bwd_kernel": "act_bwd",
    "densify_stats_kernel": "densify_stats", "tile_scan_kernel": "tile_scan",
}
SKIP = 5  # warm-up launches left out of the average

# bench.py config c2 (the workload the metric is quoted on) unless a fourth argument names another bench.py config
WORKLOAD = {"workload": "200k Gaussians, 60 frames, 960x540, HexPlane on", "gaussians": 200000, "width": 960, "height": 540}
DEFORM_FLOATS = 2_904_970          # live floats of the deformation field at the default time resolution (50)


def deform_floats(time_res):
    """Live (optimised) floats of the deformation field: the c2 figure with the three space-time planes of both levels resized."""
    base_t, planes = 50, 0
    for mult in (1, 2):
        planes += 3 * 32 * 64 * mult * (time_res - base_t)      # (x,t), (y,t), (z,t) planes: 32 channels x 64*mult x T
    return DEFORM_FLOATS + planes


def load(path):
    per_kernel = collections.defaultdict(list)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            m = re.search(r"(\w+_kernel)", row["Kernel_Name"])
            if m:
                per_kernel[m.group(1)].append(float(row["Counter_Value"]))
    return per_kernel


def main():
    if len(sys.argv) not in (4, 5):
        sys.exit(__doc__)
    fetch, write, prefix = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    n_deform = DEFORM_FLOATS
    if len(sys.argv) == 5:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        c = bench.CONFIGS[sys.argv[4]]
        WORKLOAD.update({"workload": c["name"], "gaussians": c["P"], "width": c["W"], "height": c["H"]})
        n_deform = deform_floats(c["time_res"])
    rows, kernels = [], {}
    for sym, short in NAMES.items():
        fv, wv = fetch.get(sym, [])[SKIP:], write.get(sym, [])[SKIP:]
        if not fv or not wv:
            continue
        f_kib, w_kib = sum(fv) / len(fv), sum(wv) / len(wv)
        if sym == "adam_kernel":
            # the optimizer step is taken in two launches (appearance parameters early, on the second stream; the rest late): the
            # figure is per STEP -- launches per step from the launch counts of adam and the compositing backward
            per_step = max(1, round(len(fetch[sym]) / max(1, len(fetch.get("render_bwd_kernel", [])))))
            f_kib, w_kib = f_kib * per_step, w_kib * per_step
        rd, wr = int(round(f_kib * 1024 * 2)), int(round(w_kib * 1024))
        rows.append([sym, len(fv), round(f_kib, 1), round(w_kib, 1), rd, wr, rd + wr])
        kernels[short] = {"launches_averaged": len(fv), "FETCH_SIZE_KiB_raw": round(f_kib, 1),
                          "WRITE_SIZE_KiB_raw": round(w_kib, 1), "read_bytes": rd, "write_bytes": wr,
                          "traffic_bytes": rd + wr}

    floats = WORKLOAD["gaussians"] * 59 + n_deform
    want_rd, want_wr = floats * 16, floats * 12
    got = kernels.get("adam")
    if not got:
        sys.exit("no adam_kernel launches in the counter files: cannot calibrate")
    err_rd, err_wr = got["read_bytes"] / want_rd - 1, got["write_bytes"] / want_wr - 1
    print(f"calibration on adam_kernel: read {got['read_bytes'] / 1e6:.2f} MB vs {want_rd / 1e6:.2f} MB ({err_rd:+.2%}), "
          f"written {got['write_bytes'] / 1e6:.2f} MB vs {want_wr / 1e6:.2f} MB ({err_wr:+.2%})")
    if abs(err_rd) > 0.03 or abs(err_wr) > 0.03:
        sys.exit("calibration off by more than 3 %: not writing a summary")

    with open(prefix + "_pmc_traffic_per_kernel.csv", "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["kernel", "launches_averaged", "FETCH_SIZE_KiB_raw_avg", "WRITE_SIZE_KiB_raw_avg", "read_bytes_corrected",
                    "write_bytes", "traffic_bytes_per_launch"])
        w.writerows(rows)
    doc = dict(WORKLOAD)
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    native = importlib.import_module("iclr2025_3d-mom_amd._native")
    doc.update({
        # the library the counters were collected on (mom_version() carries a hash of the kernel sources): bench.py reports the
        # figures of this file only while the running library is that build
        "lib_version": native.lib().mom_version().decode(),
        "what": "HBM-side traffic per launch from rocprofv3 PMC counters, one MI355X, bench.py config " + (sys.argv[4] if len(sys.argv) == 5 else "c2"),
        "method": "two separate --pmc passes, per-dispatch values averaged over the launches after the first %d" % SKIP,
        "corrections": {"unit": "counter values are KiB (x1024)", "FETCH_SIZE": "x2 on gfx950", "WRITE_SIZE": "exact"},
        "calibration": {"adam_kernel_read_error": round(err_rd, 4), "adam_kernel_write_error": round(err_wr, 4),
                        "note": "the x2 read correction is calibrated on streaming reads; for the gather kernels "
                                "(render_*, hexplane_*) the read figure is an upper bound"},
        "kernels": kernels,
    })
    with open(prefix + "_pmc_traffic.json", "w") as out:
        json.dump(doc, out, indent=1)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
