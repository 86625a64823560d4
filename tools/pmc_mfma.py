"""Summarise a rocprofv3 --pmc pass into profiles/<prefix>_mfma_util.json: matrix-pipe utilisation of the deformation MLP
kernels (the only MFMA work on the path).

Collect on the GPU box (program directly after `--`, counters in their own run):

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
        SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --steps 20 --warmup 5 \
        --no-cpu-baseline --no-extra

then here:

    python tools/pmc_mfma.py gpurun_out/pmc_mfma/*/*_counter_collection.csv gpurun_out/pmc_mfma/*/*_kernel_trace.csv profiles/r01_i

Utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), with kernel cycles = SQ_BUSY_CYCLES / 32 (that counter
is summed over the 32 shader engines, each busy for the whole launch).  A cross-check that needs no clock: v_mfma_f32_32x32x2_f32 occupies the pipe for 64 cycles, so the busy cycles
must equal 64 x the kernel's MFMA count (FLOP / 4096); the script prints both.
"""
import collections
import csv
import json
import re
import sys

SIMDS = 256 * 4
P = 200_000
# dx: three recomputed head layers + three W1^T + W0^T = seven 64x64 layers per Gaussian (the thin output layers run on the VALU)
FLOP = {"deform_bwd_b3f_kernel": 0, "deform_bwd_b3g_kernel": 0, "deform_fwd_kernel": 34_048 * P, "deform_field_fwd_kernel": 34_048 * P, "deform_field_fwd_b3_kernel": 0, "deform_bwd_dx_kernel": 7 * 2 * 64 * 64 * P, "deform_bwd_dw_kernel": 4 * 2 * 64 * 64 * P}


def main():
    if len(sys.argv) != 4:
        sys.exit(__doc__)
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if m:
            cnt[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[2])):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if m:
            dur[m.group(1)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {}
    for k in ("deform_fwd_kernel", "deform_field_fwd_kernel", "deform_field_fwd_b3_kernel", "deform_bwd_b3f_kernel", "deform_bwd_b3g_kernel", "deform_bwd_dx_kernel",
              "deform_bwd_dw_kernel"):
        if k not in cnt:
            continue
        c = {n: sum(v[5:]) / max(1, len(v[5:])) for n, v in cnt[k].items()}
        d_ns = sum(dur[k][5:]) / max(1, len(dur[k][5:]))
        # SQ_BUSY_CYCLES is summed over the chip's 32 shader engines, each busy for the length of the kernel: it gives the
        # kernel's length in shader cycles (and, with the traced duration, the clock: expect 2.1-2.4 GHz).  Wave residency
        # (SQ_WAVE_CYCLES / waves) does not: waves come and go during a launch.
        kernel_cycles = c["SQ_BUSY_CYCLES"] / 32.0
        clock_ghz = kernel_cycles / d_ns
        util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * kernel_cycles)
        rec = {"duration_us": round(d_ns / 1e3, 1), "mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"],
               "implied_clock_GHz": round(clock_ghz, 2), "mfma_pipe_utilisation": round(util, 3),
               "wait_any": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
               "wait_inst_any": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)}
        if "SQ_ACTIVE_INST_ANY" in c:
            rec["active_inst_any"] = round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
        if k == "deform_field_fwd_b3_kernel":
            # 4 layers x 48 v_mfma_f32_32x32x16_bf16 per tile of 32 Gaussians (six piece products of an exact three-way bf16
            # split per f32 product), 32 cycles each (some runs count 8 passes of 4: compare with the measured busy cycles)
            rec["bf16_mfma_per_launch"] = 192 * ((P + 31) // 32)
            rec["expected_busy_cycles_at_32_per_mfma"] = 32.0 * rec["bf16_mfma_per_launch"]
            rec["equivalent_f32_TFLOPs"] = round(34_048 * P / d_ns / 1e3, 1)
            rec["note"] = "HexPlane gather and MLP in one kernel: the duration covers both; the matrix pipe is the bf16 one"
        if k in ("deform_bwd_b3f_kernel", "deform_bwd_b3g_kernel"):
            # per tile of 32 Gaussians: three head waves x 144 + the trunk wave's 96 v_mfma_f32_32x32x16_bf16 (six piece products per
            # f32 product; eleven 64x64 layer products per Gaussian: three recomputed, three W1^T, W0^T, four weight gradients)
            rec["bf16_mfma_per_launch"] = (3 * 144 + 96) * ((P + 31) // 32)
            rec["expected_busy_cycles_at_32_per_mfma"] = 32.0 * rec["bf16_mfma_per_launch"]
            rec["equivalent_f32_TFLOPs"] = round(11 * 2 * 64 * 64 * P / d_ns / 1e3, 1)
            rec["note"] = ("one wave per SIMD, four roles per workgroup" if k.endswith("b3f_kernel") else "two waves per SIMD, every role cut in two") + "; the head SIMDs carry 144 of a tile's MFMAs each, the trunk SIMD 96"
        if FLOP[k]:
            rec["expected_busy_cycles_from_flop"] = 64.0 * FLOP[k] / 4096.0
            rec["achieved_TFLOPs"] = round(FLOP[k] / d_ns / 1e3, 1)
        out[k] = rec
        print(k, rec)
    doc = {"what": "matrix-pipe utilisation of the deformation MLP kernels, one MI355X, bench.py config c2 (200k Gaussians)",
           "method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... in its own pass; utilisation = busy cycles / (1024 SIMDs x "
                     "SQ_BUSY_CYCLES / 32); v_mfma_f32_32x32x2_f32 holds the pipe for 64 cycles, which the expected_busy_cycles "
                     "cross-check uses", "kernels": out}
    with open(sys.argv[3] + "_mfma_util.json", "w") as fh:
        json.dump(doc, fh, indent=1)


if __name__ == "__main__":
    main()
