"""ctypes front-end of oracle/raster_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It mirrors the call structure of the reference's
CudaRasterizer::Rasterizer::forward / ::backward (rasterizer_impl.cu:198-444) and
RasterizeGaussiansCUDA / ...BackwardCUDA (rasterize_points.cu:35-202) on numpy
arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force:
        subprocess.check_call(["make", "-C", _HERE, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def _lib(fp64: bool):
    key = "f64" if fp64 else "f32"
    if key not in _LIBS:
        path = os.path.join(_HERE, "_build", f"liboracle_{key}.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        lib.oracle_preprocess.restype = C.c_int
        lib.oracle_get_higher_msb.restype = C.c_uint32
        lib.oracle_get_higher_msb.argtypes = [C.c_uint32]
        lib.oracle_real_size.restype = C.c_int
        assert lib.oracle_real_size() == (8 if fp64 else 4)
        _LIBS[key] = lib
    return _LIBS[key]


def set_threads(n: int) -> None:
    for fp64 in (False, True):
        _lib(fp64).oracle_set_threads(C.c_int(n))


def get_higher_msb(n: int) -> int:
    return int(_lib(False).oracle_get_higher_msb(n))


def _p(a: Optional[np.ndarray]):
    if a is None:
        return C.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def _prep(a, dt, allow_empty=True):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a), dtype=dt)
    if allow_empty and a.size == 0:
        return None
    return a


class RasterState(dict):
    """All buffers of one forward call (GeometryState / BinningState / ImageState)."""

    __getattr__ = dict.__getitem__


def forward(means3D, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy, bg,
            shs=None, sh_degree=0, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            scale_modifier=1.0, fp64=False) -> RasterState:
    """CudaRasterizer::Rasterizer::forward (rasterizer_impl.cu:198-339)."""
    lib = _lib(fp64)
    rt = np.float64 if fp64 else np.float32
    real = C.c_double if fp64 else C.c_float
    means3D = _prep(means3D, rt, False)
    P = means3D.shape[0]
    opacities = _prep(opacities, rt, False).reshape(-1)
    shs = _prep(shs, rt)
    colors_precomp = _prep(colors_precomp, rt)
    scales = _prep(scales, rt)
    rotations = _prep(rotations, rt)
    cov3D_precomp = _prep(cov3D_precomp, rt)
    viewmatrix = _prep(viewmatrix, rt, False).reshape(-1)
    projmatrix = _prep(projmatrix, rt, False).reshape(-1)
    campos = _prep(campos, rt, False).reshape(-1)
    bg = _prep(bg, rt, False).reshape(-1)
    M = 0 if shs is None else shs.shape[1]
    if P != 0 and shs is None and colors_precomp is None:
        raise ValueError("need shs or colors_precomp")

    st = RasterState()
    st["P"], st["W"], st["H"], st["M"], st["D"] = P, W, H, M, sh_degree
    st["radii"] = np.zeros(P, np.int32)
    st["means2D"] = np.zeros((P, 2), rt)
    st["depths"] = np.zeros(P, rt)
    st["cov3D"] = np.zeros((P, 6), rt)
    st["rgb"] = np.zeros((P, 3), rt)
    st["conic_opacity"] = np.zeros((P, 4), rt)
    st["tiles_touched"] = np.zeros(P, np.uint32)
    st["point_offsets"] = np.zeros(P, np.uint32)
    st["clamped"] = np.zeros((P, 3), np.uint8)
    st["out_color"] = np.zeros((3, H, W), rt)
    st["out_depth"] = np.zeros((1, H, W), rt)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    st["grid"] = (gx, gy)
    st["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    st["final_T"] = np.zeros(H * W, rt)
    st["n_contrib"] = np.zeros(H * W, np.uint32)
    # entries each tile's BLOCK walks before it exits (forward.cu:305-312): SURVEY 8d's Q = 256 x their sum
    st["tile_walked"] = np.zeros(gx * gy, np.uint32)
    if P == 0:  # rasterize_points.cu:82 short-circuit
        st["num_rendered"] = 0
        st["point_list"] = np.zeros(0, np.uint32)
        st["point_list_keys"] = np.zeros(0, np.uint64)
        st["pairs_evaluated"] = 0
        return st

    R = lib.oracle_preprocess(
        C.c_int(P), C.c_int(sh_degree), C.c_int(M), _p(means3D), _p(scales), real(scale_modifier),
        _p(rotations), _p(opacities), _p(shs), _p(st.clamped), _p(cov3D_precomp), _p(colors_precomp),
        _p(viewmatrix), _p(projmatrix), _p(campos), C.c_int(W), C.c_int(H), real(tanfovx), real(tanfovy),
        _p(st.radii), _p(st.means2D), _p(st.depths), _p(st.cov3D), _p(st.rgb), _p(st.conic_opacity),
        _p(st.tiles_touched), _p(st.point_offsets))
    st["num_rendered"] = int(R)
    depth_bits = np.ascontiguousarray(st.depths.astype(np.float32)).view(np.uint32)
    st["keys_unsorted"] = np.zeros(R, np.uint64)
    st["values_unsorted"] = np.zeros(R, np.uint32)
    st["point_list_keys"] = np.zeros(R, np.uint64)
    st["point_list"] = np.zeros(R, np.uint32)
    lib.oracle_bin(C.c_int(P), C.c_int(W), C.c_int(H), C.c_int(R), _p(st.means2D), _p(depth_bits),
                   _p(st.point_offsets), _p(st.radii), _p(st.keys_unsorted), _p(st.values_unsorted),
                   _p(st.point_list_keys), _p(st.point_list), _p(st.ranges))
    feat = colors_precomp if colors_precomp is not None else st.rgb
    pairs = C.c_uint64(0)
    lib.oracle_render_forward(C.c_int(W), C.c_int(H), _p(st.ranges), _p(st.point_list), _p(st.means2D),
                              _p(feat), _p(st.depths), _p(st.conic_opacity), _p(bg), _p(st.final_T),
                              _p(st.n_contrib), _p(st.out_color), _p(st.out_depth), C.byref(pairs), _p(st.tile_walked))
    st["pairs_evaluated"] = int(pairs.value)
    st["_inputs"] = dict(means3D=means3D, shs=shs, colors_precomp=colors_precomp, scales=scales,
                         rotations=rotations, cov3D_precomp=cov3D_precomp, viewmatrix=viewmatrix,
                         projmatrix=projmatrix, campos=campos, bg=bg, tanfovx=tanfovx, tanfovy=tanfovy,
                         scale_modifier=scale_modifier, fp64=fp64)
    return st


def backward(st: RasterState, dL_dout_color, dL_dout_depth=None) -> dict:
    """CudaRasterizer::Rasterizer::backward (rasterizer_impl.cu:343-444); grads zero-initialised as in
    RasterizeGaussiansBackwardCUDA (rasterize_points.cu:154-163)."""
    i = st["_inputs"]
    fp64 = i["fp64"]
    lib = _lib(fp64)
    rt = np.float64 if fp64 else np.float32
    real = C.c_double if fp64 else C.c_float
    P, W, H, M = st.P, st.W, st.H, st.M
    dpix = _prep(dL_dout_color, rt, False)
    ddep = np.zeros((1, H, W), rt) if dL_dout_depth is None else _prep(dL_dout_depth, rt, False)
    g = dict(
        dL_dmeans2D=np.zeros((P, 3), rt), dL_dconic=np.zeros((P, 4), rt), dL_dopacity=np.zeros((P, 1), rt),
        dL_dcolors=np.zeros((P, 3), rt), dL_ddepths=np.zeros((P, 1), rt), dL_dmeans3D=np.zeros((P, 3), rt),
        dL_dcov3D=np.zeros((P, 6), rt), dL_dsh=np.zeros((P, M, 3), rt), dL_dscales=np.zeros((P, 3), rt),
        dL_drotations=np.zeros((P, 4), rt))
    if P == 0:
        return g
    color_ptr = i["colors_precomp"] if i["colors_precomp"] is not None else st.rgb
    lib.oracle_render_backward(C.c_int(W), C.c_int(H), _p(st.ranges), _p(st.point_list), _p(i["bg"]),
                               _p(st.means2D), _p(st.conic_opacity), _p(color_ptr), _p(st.depths),
                               _p(st.final_T), _p(st.n_contrib), _p(dpix), _p(ddep), _p(g["dL_dmeans2D"]),
                               _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]), _p(g["dL_ddepths"]))
    cov3D_ptr = i["cov3D_precomp"] if i["cov3D_precomp"] is not None else st.cov3D
    lib.oracle_preprocess_backward(
        C.c_int(P), C.c_int(st.D), C.c_int(M), _p(i["means3D"]), _p(st.radii), _p(i["shs"]), _p(st.clamped),
        _p(i["scales"]), _p(i["rotations"]), real(i["scale_modifier"]), _p(cov3D_ptr), _p(i["viewmatrix"]),
        _p(i["projmatrix"]), C.c_int(W), C.c_int(H), real(i["tanfovx"]), real(i["tanfovy"]), _p(i["campos"]),
        _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dmeans3D"]), _p(g["dL_dcolors"]), _p(g["dL_ddepths"]),
        _p(g["dL_dcov3D"]), _p(g["dL_dsh"]), _p(g["dL_dscales"]), _p(g["dL_drotations"]))
    return g


def mark_visible(means3D, viewmatrix, projmatrix):
    lib = _lib(False)
    m = _prep(means3D, np.float32, False)
    out = np.zeros(m.shape[0], np.uint8)
    lib.oracle_mark_visible(C.c_int(m.shape[0]), _p(m), _p(_prep(viewmatrix, np.float32).reshape(-1)),
                            _p(_prep(projmatrix, np.float32).reshape(-1)), _p(out))
    return out.astype(bool)


def knn_mean_dist2(points):
    """simple_knn._C.distCUDA2 (simple-knn/spatial.cu:15-25)."""
    lib = _lib(False)
    p = _prep(points, np.float32, False)
    out = np.zeros(p.shape[0], np.float32)
    if p.shape[0]:
        lib.oracle_knn(C.c_int(p.shape[0]), _p(p), _p(out))
    return out
