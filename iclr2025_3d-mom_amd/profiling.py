"""Roofline bookkeeping for bench.py.

Units: P = Gaussians, R = (Gaussian, tile) instances, N = pixels.  Two instance counts exist and
every figure says which one it is on:
    R_ref     the reference's count -- every tile of every splat's rectangle (duplicateWithKeys, rasterizer_impl.cu:70-111) --
              which SURVEY 8d's algorithmic bytes are defined on, and which a keep_all_tiles launch processes;
    R_binned  what the default binning keeps (the instances that can reach alpha >= 1/255 in their tile): the units the timed
              launches actually PROCESS.
`frac` is always on the processed units; `frac_on_reference_R` is the same time charged with the reference's count.

Bounds:
  * the compositing kernels (render_fwd, render_bwd) are bound by vector-instruction issue, not by HBM: `bound` = "valu",
    achieved = wave64 vector instructions per second = SQ_INSTS_VALU per launch (the committed rocprofv3 --pmc pass of the same
    workload and library build: profiles/*_sq.json) / the live launch duration (HIP events, csrc/profile.hip); peak = 1024 SIMDs x
    clock / 2 cycles per instruction (the fp32 vector peak's issue rate) with the clock MEASURED in that counter pass (GRBM_GUI_ACTIVE / 8 XCDs / launch duration, the
    guide's effective-clock formula).  The HBM figure rides along as `hbm`.
  * the MLP kernels: "mfma" against the fp32 MFMA peak (157 TFLOP/s).
  * everything else: "hbm", algorithmic bytes / duration against 8 TB/s.
"""
import ctypes as C

from . import _native as N

HBM_PEAK_GBS = 8000.0
FP32_VALU_PEAK_TFLOPS = 157.0      # MI355X fp32 vector peak (MI355X_MICROARCH.md)
SIMDS = 1024                       # 256 CUs x 4
CYCLES_PER_VALU = 2.0              # the SIMD's peak: one wave64 vector instruction per 2 cycles (MI355X_MICROARCH.md, v_fma_f32; it is
                                   # what makes 1024 SIMDs x 64 lanes x 2 flop at 2.4 GHz the 157 TFLOP/s fp32 vector peak).  What a real
                                   # mix reaches is lower -- tools/probe/valu_rate.hip prices v_mul / v_add at 2.5 cycles, DPP adds, compares
                                   # and selects at 4.2, v_exp / v_rcp / v_readlane / permlane swaps at 8.2 -- but a fraction of THIS peak is
                                   # at most 1 by construction (with 4 cycles, rounds 1-3's figure, render_bwd read 1.05 at c2 and 1.12 at c3)
# the deformation MLP is the MFMA-bound part of the path: 34 048 FLOP per Gaussian forward (trunk 64x64, three
# head hidden layers 64x64, heads 64x{3,3,4}); the backward (dX and dW, both kernels inside one timed scope) is twice that
FP32_MFMA_PEAK_TFLOPS = 157.0
MFMA_FLOP_PER_GAUSSIAN = {"mlp_fwd": 34_048, "mlp_bwd": 68_096}
VALU_BOUND = ("render_fwd", "render_bwd")

SLOTS = {n: i for i, n in enumerate(
    ["preprocess_fwd", "tile_hist", "tile_scan", "tile_scatter", "tile_sort", "render_fwd", "render_bwd", "preprocess_bwd",
     "hexplane_fwd", "hexplane_bwd", "adam", "l1_loss", "plane_reg", "mlp_fwd", "mlp_bwd"])}


def algorithmic_bytes(kernel, P, R, Npix, deform_floats=2_904_970):
    """Bytes one launch must move if every operand crossed HBM exactly once (R: the instances that launch processes)."""
    return {
        # render fwd: id 4 + record 40 (xy 8, conic+opacity 16, rgb 12, depth 4) per instance; 24 B/pixel out
        "render_fwd": R * 44 + Npix * 24,
        # render bwd: the same 44 B/instance read, 40 B/instance of tile-reduced gradients, 24 B/pixel read
        "render_bwd": R * 84 + Npix * 24,
        # preprocess fwd: 236 B in (means 12, scales 12, rot 16, opacity 4, SH 192) + 48 B record + 24 B cov3D + 8 B
        "preprocess_fwd": P * (236 + 48 + 24 + 8),
        "preprocess_bwd": P * (307 + 256),
        "adam": (P * 59 + deform_floats) * 28,
        "tile_sort": R * 12,
        # binning: the histogram reads the 40-byte rectangle record per Gaussian; the scan 20 B per tile; the scatter writes one
        # 8-byte key per instance
        "tile_hist": P * 40,
        "tile_scan": (Npix // 256) * 20,
        "tile_scatter": P * 40 + R * 8,
        # HexPlane: forward 4620 B gathered + 640 B written per Gaussian; backward (gather + scatter in one
        # scope) 6412 + 768 B per Gaussian in the gather, 768 B in the scatter, and the plane gradients' read-modify-write
        "hexplane_fwd": P * (4620 + 640),
        "hexplane_bwd": P * (6412 + 768 + 768) + 2 * 2_887_680 * 4,
        "l1_loss": Npix * 36,
        "plane_reg": 3 * 2_887_680 * 4,
    }[kernel]


def step_bytes(P, R, npix, lambda_dssim=0.0, deform_floats=2_904_970):
    """algorithmic bytes of one fine-stage step, fp32: B = P*2751 + R*172 + N_pix*84 (+240 with SSIM)
    + B_def, B_def = live deformation floats * 28 (Adam) + 2 * 11.55 MB (plane-gradient RMW) + 11.55 MB (regulariser read)."""
    planes = 2_887_680 * 4
    return P * 2751 + R * 172 + npix * (84 + (240 if lambda_dssim else 0)) + deform_floats * 28 + 3 * planes


def step_roofline(P, R_binned, R_ref, npix, seconds_per_step, lambda_dssim=0.0):
    """The step-level figure: SURVEY 8d's bytes on the instances the step PROCESSED over the step time, against 8 TB/s; the same
    time charged with the reference's instance count beside it."""
    b = step_bytes(P, R_binned, npix, lambda_dssim)
    b_ref = step_bytes(P, R_ref, npix, lambda_dssim)
    gbs = b / seconds_per_step / 1e9
    return {"bound": "hbm", "algorithmic_bytes_per_step": b, "instances": R_binned,
            "instances_are": "processed by the timed steps (the default binning's count)", "achieved": gbs, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "frac_on_reference_R": b_ref / seconds_per_step / 1e9 / HBM_PEAK_GBS, "instances_reference": R_ref,
            "algorithmic_bytes_on_reference_R": b_ref,
            "formula": "P*2751 + R*172 + Npix*84 (+240 with SSIM) + 116 MB (SURVEY 8d); per GPU"}


def enable(kernel, on=True, period=1):
    """period > 1: time every period-th launch (an event pair costs the stream ~13 us of bubbles, tools/gap_stats.py)."""
    N.check(N.lib().mom_profile_enable(SLOTS[kernel], (max(1, int(period)) if on else 0)), "mom_profile_enable")


def read(kernel, reset=True):
    ms, cnt = C.c_double(0), C.c_longlong(0)
    N.check(N.lib().mom_profile_read(SLOTS[kernel], C.byref(ms), C.byref(cnt), 1 if reset else 0), "mom_profile_read")
    return ms.value, cnt.value


def valu_bound(insts_per_launch, clock_ghz, avg_s):
    """achieved / peak / frac of a kernel bound by vector-instruction issue."""
    achieved = insts_per_launch / avg_s / 1e9                     # G wave-instructions / s
    peak = SIMDS * clock_ghz / CYCLES_PER_VALU
    return achieved, peak


def roofline(kernel, P, R_binned, Npix, traffic=None, R_ref=None, sq=None):
    """The live figure for `kernel` from the HIP-event slot.  sq: that kernel's entry of the committed counter summary
    (profiles/*_sq.json: SQ_INSTS_VALU, clock_ghz), or None."""
    ms, cnt = read(kernel)
    if cnt == 0:
        return None
    avg_s = ms / cnt * 1e-3
    if kernel in MFMA_FLOP_PER_GAUSSIAN:
        flop = float(P) * MFMA_FLOP_PER_GAUSSIAN[kernel]
        tf = flop / avg_s / 1e12
        return {"bound": "mfma", "kernel": kernel, "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "avg_launch_us": avg_s * 1e6, "launches": cnt, "launches_are": "the launches timed with HIP events (a sample of the region's launches when bench.py sets a period)",
                "algorithmic_flop_per_launch": flop, "dtype": "f32 (v_mfma_f32_32x32x2_f32)"}
    b = algorithmic_bytes(kernel, P, R_binned, Npix)
    gbs = b / avg_s / 1e9
    hbm = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": b,
           "instances": R_binned, "instances_are": "processed by the timed launches (the default binning's count)"}
    if R_ref is not None:
        b_ref = algorithmic_bytes(kernel, P, R_ref, Npix)
        hbm.update({"frac_on_reference_R": b_ref / avg_s / 1e9 / HBM_PEAK_GBS, "instances_reference": R_ref,
                    "algorithmic_bytes_on_reference_R": b_ref})
    common = {"kernel": kernel, "traffic": traffic, "avg_launch_us": avg_s * 1e6, "launches": cnt,
              "launches_are": "the launches timed with HIP events (a sample of the region's launches when bench.py sets a period)"}
    if kernel in VALU_BOUND:
        out = {"bound": "valu", **common, "hbm": hbm}
        if sq and sq.get("SQ_INSTS_VALU") and sq.get("clock_ghz"):
            achieved, peak = valu_bound(sq["SQ_INSTS_VALU"], sq["clock_ghz"], avg_s)
            out.update({"achieved": achieved, "peak": peak, "unit": "G wave64 vector instructions/s", "frac": achieved / peak,
                        "wave_insts_per_launch": sq["SQ_INSTS_VALU"], "clock_ghz_in_counter_pass": sq["clock_ghz"],
                        "frac_in_counter_pass": sq.get("valu_issue_frac"),
                        "how": "SQ_INSTS_VALU per launch (committed --pmc pass, same workload and library build) / live launch "
                               "duration; peak = 1024 SIMDs x measured clock / 2 cycles per instruction (the issue rate of the 157 TFLOP/s fp32 vector peak)"})
        else:
            out.update({"achieved": None, "peak": None, "unit": "G wave64 vector instructions/s", "frac": None,
                        "note": "no counter summary of this library build is committed: the instruction count is unknown"})
        return out
    return {"bound": "hbm", **common, **hbm}
