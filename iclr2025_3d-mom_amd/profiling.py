"""Roofline bookkeeping for bench.py.

Algorithmic bytes per launch follow SURVEY.md section 8(d) (fp32): P = Gaussians, R = (Gaussian, tile)
instances, N = pixels.  `achieved` = algorithmic bytes / average launch duration measured with HIP events around
every launch of that kernel inside the timed region (csrc/profile.hip); `peak` = 8 TB/s HBM3E (MI355X_MICROARCH.md).
"""
import ctypes as C

from . import _native as N

HBM_PEAK_GBS = 8000.0
FP32_VALU_PEAK_TFLOPS = 157.0      # MI355X fp32 vector peak (MI355X_MICROARCH.md)
# SURVEY 8d, "Raster VALU": FLOP per evaluated (pixel, splat) pair, with one exp each; Q = 256 R pairs per launch
VALU_FLOP_PER_PAIR = {"render_fwd": 20, "render_bwd": 70}
# the deformation MLP is the MFMA-bound part of the path: 34 048 FLOP per Gaussian forward (trunk 64x64, three
# head hidden layers 64x64, heads 64x{3,3,4}); the backward (dX and dW, both kernels inside one timed scope) is twice that
FP32_MFMA_PEAK_TFLOPS = 157.0
MFMA_FLOP_PER_GAUSSIAN = {"mlp_fwd": 34_048, "mlp_bwd": 68_096}

SLOTS = {n: i for i, n in enumerate(
    ["preprocess_fwd", "tile_hist", "tile_scan", "tile_scatter", "tile_sort", "render_fwd", "render_bwd", "preprocess_bwd",
     "hexplane_fwd", "hexplane_bwd", "adam", "l1_loss", "plane_reg", "mlp_fwd", "mlp_bwd"])}


def algorithmic_bytes(kernel, P, R, Npix, deform_floats=2_904_970):
    """Bytes one launch must move if every operand crossed HBM exactly once."""
    return {
        # render fwd: id 4 + record 40 (xy 8, conic+opacity 16, rgb 12, depth 4) per instance; 24 B/pixel out
        "render_fwd": R * 44 + Npix * 24,
        # render bwd: the same 44 B/instance read, 40 B/instance of tile-reduced gradients, 24 B/pixel read
        "render_bwd": R * 84 + Npix * 24,
        # preprocess fwd: 236 B in (means 12, scales 12, rot 16, opacity 4, SH 192) + 48 B record + 24 B cov3D + 8 B
        "preprocess_fwd": P * (236 + 48 + 24 + 8),
        "preprocess_bwd": P * (307 + 256),
        # hexplane: 6144 B of plane texels gathered per Gaussian, 12 B in, 256 B out (fwd); + 256 B in and
        # 6144 B scattered (bwd)
        "hexplane_fwd": P * (12 + 6144 + 256),
        "hexplane_bwd": P * (12 + 6144 + 256 + 6144 + 12),
        "adam": (P * 59 + deform_floats) * 28,
        "tile_sort": R * 12,
    }[kernel]


def enable(kernel, on=True):
    N.check(N.lib().mom_profile_enable(SLOTS[kernel], 1 if on else 0), "mom_profile_enable")


def read(kernel, reset=True):
    ms, cnt = C.c_double(0), C.c_longlong(0)
    N.check(N.lib().mom_profile_read(SLOTS[kernel], C.byref(ms), C.byref(cnt), 1 if reset else 0), "mom_profile_read")
    return ms.value, cnt.value


def roofline(kernel, P, R, Npix, traffic=None):
    ms, cnt = read(kernel)
    if cnt == 0:
        return None
    avg_s = ms / cnt * 1e-3
    if kernel in MFMA_FLOP_PER_GAUSSIAN:
        flop = float(P) * MFMA_FLOP_PER_GAUSSIAN[kernel]
        tf = flop / avg_s / 1e12
        return {"bound": "mfma", "kernel": kernel, "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "avg_launch_us": avg_s * 1e6, "launches": cnt,
                "algorithmic_flop_per_launch": flop, "dtype": "f32 (v_mfma_f32_32x32x2_f32)"}
    b = algorithmic_bytes(kernel, P, R, Npix)
    achieved = b / avg_s / 1e9
    out = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_us": avg_s * 1e6, "launches": cnt,
           "algorithmic_bytes_per_launch": b}
    return out
