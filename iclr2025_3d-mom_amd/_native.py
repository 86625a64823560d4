"""ctypes binding of libmom4d.so (the C ABI declared in include/mom4d.h).

The product path FAILS LOUDLY when the HIP library is missing: there is no CPU or
torch fallback behind these calls.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOM4D_LIB") or os.path.join(_HERE, "lib", "libmom4d.so")
_lib = None

MOM_OK, MOM_EINVAL, MOM_ELAUNCH, MOM_ECAPACITY, MOM_EUNAVAILABLE = 0, -1, -2, -3, -4
_ERR = {MOM_EINVAL: "invalid argument", MOM_ELAUNCH: "HIP launch/runtime failure", MOM_ECAPACITY: "scratch too small",
        MOM_EUNAVAILABLE: "run-time dependency missing (librccl)"}
COMM_F32, COMM_I32, COMM_SUM, COMM_MAX = 0, 1, 0, 1


class MomError(RuntimeError):
    pass


GACC_FLOATS = int(os.environ.get("MOM_GACC_FLOATS", "12"))         # (the override: only for A/B builds with -DMOM_GACC_FLOATS=16)
# MOM_GACC_FLOATS: floats per Gaussian of the backward's accumulator record
ABI_VERSION = 7          # MOM_ABI_VERSION of the include/mom4d.h this mirror was written against


class MomRasterArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint),                          # sizeof(MomRasterArgs): set by __init__, checked by every entry point
                ("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
                ("background", C.c_void_p), ("means3D", C.c_void_p), ("shs", C.c_void_p), ("shs_rest", C.c_void_p),
                ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
                ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("viewmatrix", C.c_void_p),
                ("projmatrix", C.c_void_p), ("campos", C.c_void_p), ("scale_modifier", C.c_float),
                ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("prefiltered", C.c_int), ("debug", C.c_int),
                ("tile_row0", C.c_int), ("tile_row1", C.c_int),     # tile-row shard: 0,0 = every row
                ("forward_only", C.c_int),                          # no backward will follow: skip the state only it reads
                ("overflow_tag", C.c_uint),                         # what an overflow of this call leaves in *status_dev
                ("keep_all_tiles", C.c_int),                        # !=0: bin the whole rectangle like the reference (tests)
                ("l1_target", C.c_void_p), ("l1_grad", C.c_void_p), ("l1_sums", C.c_void_p),   # optional L1 epilogue of the forward
                ("accum_cleared", C.c_int),
                ("l1_partials", C.c_void_p),                        # per-tile sums of the L1 epilogue instead of two contended atomics
                ("status_post", C.c_void_p), ("status_serial", C.c_uint),   # the frame's status bits, posted to pinned host memory
                ("l1_grad_scale", C.c_float)]                       # 0 (= 1) or a factor on the L1 epilogue's gradient (camera-batch shard: 1 / world)

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(type(self))


class MomRasterGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D",
                                          "dL_dsh", "dL_dsh_rest", "dL_dscales", "dL_drotations", "act_rotations_raw",
                                          "dL_dscales_copy", "dL_drotations_copy")]


class MomRasterLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("geom_rec", "geom_cov3D", "geom_clamped", "geom_gacc", "img_ranges",
                                          "img_n_contrib", "img_final_T", "img_tile_counts", "bin_keys",
                                          "bin_point_list", "img_tile_walked")]


class MomHexPlane(C.Structure):
    _fields_ = [("levels", C.c_int), ("channels", C.c_int), ("res", (C.c_int * 4) * 4),
                ("planes", (C.c_void_p * 6) * 4), ("grads", (C.c_void_p * 6) * 4), ("aabb", C.c_float * 6)]


class MomAdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_size_t), ("lr", C.c_float), ("bias_correction1", C.c_float),
                ("bias_correction2_sqrt", C.c_float)]


class MomDeformMLP(C.Structure):
    _fields_ = [("W0", C.c_void_p), ("b0", C.c_void_p), ("W1", C.c_void_p * 3), ("b1", C.c_void_p * 3),
                ("W2", C.c_void_p * 3), ("b2", C.c_void_p * 3), ("dW0", C.c_void_p), ("db0", C.c_void_p),
                ("dW1", C.c_void_p * 3), ("db1", C.c_void_p * 3), ("dW2", C.c_void_p * 3), ("db2", C.c_void_p * 3)]


class MomRegPlane(C.Structure):
    _fields_ = [("plane", C.c_void_p), ("grad", C.c_void_p), ("H", C.c_int), ("W", C.c_int), ("w_smooth", C.c_float),
                ("w_l1", C.c_float), ("grad_scale", C.c_float)]


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into lib/libmom4d.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise MomError("building libmom4d.so failed:\n" + r.stdout[-4000:])
    if verbose:
        print(r.stdout)
    return LIB_PATH


def _sig(lib):
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    if os.environ.get("MOM4D_LIB_LAX") == "1":
        # A/B tooling only (tools/kbench.py with MOM4D_LIB naming an OLDER build of the library): bind what that build has
        # (a variant built from the CURRENT header -- tools/variants.sh -- or an older build: symbols the older one lacks raise on
        # call, and its ABI check is skipped when it predates mom_abi_version)
        class _Missing:
            restype = argtypes = None

            def __init__(self, name):
                self.name = name

            def __call__(self, *a):
                raise MomError(f"{self.name} is not in {LIB_PATH}")

        class _Lax:
            def __init__(self, real):
                object.__setattr__(self, "_real", real)
                object.__setattr__(self, "_missing", {})

            def __getattr__(self, name):
                try:
                    return getattr(self._real, name)
                except AttributeError:
                    # one stub per name: the restype / argtypes assigned below must stick to what later lookups return
                    return self._missing.setdefault(name, _Missing(name))
        lib = _Lax(lib)
    else:
        for name in EXPORTS:
            getattr(lib, name)  # raises AttributeError if the library lacks a declared symbol
    lib.mom_version.restype = C.c_char_p
    lib.mom_abi_version.restype = i32
    lib.mom_abi_version.argtypes = []
    lib.mom_abi_sizeof.restype = sz
    lib.mom_abi_sizeof.argtypes = [i32]
    lib.mom_raster_geom_bytes.restype = sz
    lib.mom_raster_geom_bytes.argtypes = [i32]
    lib.mom_raster_image_bytes.restype = sz
    lib.mom_raster_image_bytes.argtypes = [i32, i32]
    lib.mom_raster_binning_bytes.restype = sz
    lib.mom_raster_binning_bytes.argtypes = [i32, i32, i32, sz]
    lib.mom_raster_layout.argtypes = [i32, i32, i32, sz, C.POINTER(MomRasterLayout)]
    lib.mom_stream_wait_stream.argtypes = [vp, vp]
    lib.mom_stream_mark.argtypes = [i32, vp]
    lib.mom_stream_wait_mark.argtypes = [vp, i32]
    lib.mom_zero_async.argtypes = [vp, sz, vp]
    lib.mom_comm_available.argtypes = []
    lib.mom_comm_last_error.restype = C.c_char_p
    lib.mom_comm_last_error.argtypes = []
    lib.mom_comm_unique_id.argtypes = [vp]
    lib.mom_comm_create.argtypes = [C.POINTER(vp), vp, i32, i32]
    lib.mom_comm_destroy.argtypes = [vp]
    lib.mom_comm_abort.argtypes = [vp]
    lib.mom_comm_world.argtypes = [vp]
    lib.mom_comm_rank.argtypes = [vp]
    lib.mom_comm_group_start.argtypes = []
    lib.mom_comm_group_end.argtypes = []
    lib.mom_comm_all_reduce.argtypes = [vp, vp, sz, i32, i32, vp]
    lib.mom_comm_all_gather.argtypes = [vp, vp, sz, i32, vp]
    lib.mom_comm_reduce_scatter.argtypes = [vp, vp, sz, i32, i32, vp]
    lib.mom_raster_forward_geometry.argtypes = [C.POINTER(MomRasterArgs), vp, vp, vp, vp, vp, vp]
    lib.mom_raster_forward_render.argtypes = [C.POINTER(MomRasterArgs), vp, vp, sz, vp, vp, vp, vp, vp]
    lib.mom_raster_backward.argtypes = [C.POINTER(MomRasterArgs), vp, vp, vp, sz, vp, vp, vp, C.POINTER(MomRasterGrads), vp]
    lib.mom_raster_backward_render.argtypes = [C.POINTER(MomRasterArgs), vp, vp, sz, vp, vp, vp, vp]
    lib.mom_raster_backward_geometry.argtypes = [C.POINTER(MomRasterArgs), vp, vp, C.POINTER(MomRasterGrads), vp]
    lib.mom_mark_visible.argtypes = [i32, vp, vp, vp, vp, vp]
    lib.mom_selftest_wave_sum.argtypes = [vp, vp, i32, vp]
    lib.mom_selftest_row_reduce.argtypes = [vp, vp, i32, i32, vp]
    lib.mom_hexplane_forward.argtypes = [C.POINTER(MomHexPlane), i32, vp, vp, C.c_float, vp, vp, vp]
    lib.mom_hexplane_backward.argtypes = [C.POINTER(MomHexPlane), i32, vp, vp, C.c_float, vp, vp, vp, vp, vp, vp, vp]
    lib.mom_hexplane_backward_lines.argtypes = [C.POINTER(MomHexPlane), i32, vp, C.c_float, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mom_hexplane_backward_scratch_bytes.restype = sz
    lib.mom_hexplane_backward_scratch_bytes.argtypes = [C.POINTER(MomHexPlane), i32]
    lib.mom_hexplane_orders_scratch_bytes.restype = sz
    lib.mom_hexplane_orders_scratch_bytes.argtypes = [i32]
    lib.mom_hexplane_orders.argtypes = [C.POINTER(MomHexPlane), i32, vp, vp, vp, vp, vp]
    lib.mom_morton_order_scratch_bytes.restype = sz
    lib.mom_morton_order_scratch_bytes.argtypes = [i32]
    lib.mom_morton_order.argtypes = [i32, vp, vp, vp, vp]
    lib.mom_adam_step.argtypes = [C.POINTER(MomAdamTensor), i32, C.c_double, C.c_double, C.c_double, vp, vp]
    lib.mom_l1_loss.argtypes = [sz, vp, vp, vp, vp, vp]
    lib.mom_l1_loss_acc.argtypes = [sz, vp, vp, vp, vp, vp]
    lib.mom_plane_regulation.argtypes = [C.POINTER(MomRegPlane), i32, vp, vp]
    lib.mom_plane_regulation_acc.argtypes = [C.POINTER(MomRegPlane), i32, vp, vp]
    lib.mom_plane_regulation_grad.argtypes = [C.POINTER(MomRegPlane), i32, vp, vp, vp]
    lib.mom_deform_forward.argtypes = [C.POINTER(MomDeformMLP), i32, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp]
    lib.mom_deform_forward_activated.argtypes = [C.POINTER(MomDeformMLP), i32, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp, vp,
                                                 vp, vp, vp]
    lib.mom_deform_backward_scratch_bytes.restype = sz
    lib.mom_deform_backward_scratch_bytes.argtypes = [i32]
    lib.mom_deform_backward.argtypes = [C.POINTER(MomDeformMLP), i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mom_deform_backward_split.argtypes = [C.POINTER(MomDeformMLP), i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mom_deform_field_supported.argtypes = [C.POINTER(MomHexPlane)]
    lib.mom_deform_field_scratch_bytes.restype = sz
    lib.mom_deform_field_scratch_bytes.argtypes = [C.POINTER(MomHexPlane), i32]
    lib.mom_deform_field_forward.argtypes = [C.POINTER(MomHexPlane), C.POINTER(MomDeformMLP), i32, vp, C.c_float, vp, vp, vp, vp,
                                             C.c_float, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mom_densify_stats.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp]
    lib.mom_select_scratch_bytes.restype = sz
    lib.mom_select_scratch_bytes.argtypes = [i32]
    lib.mom_select_plan.argtypes = [i32, vp, vp, vp, vp, vp, vp]
    lib.mom_select_apply.argtypes = [i32, vp, C.POINTER(MomRowSelect), i32, vp]
    lib.mom_ssim_forward.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.mom_ssim_forward_slab.argtypes = [i32, i32, i32, sz, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.mom_ssim_backward_slab.argtypes = [i32, i32, i32, sz, vp, vp, vp, vp, C.c_float, vp, vp, vp]
    lib.mom_ssim_backward.argtypes = [i32, i32, i32, vp, vp, vp, vp, C.c_float, vp, vp, vp]
    lib.mom_activations_forward.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp]
    lib.mom_activations_backward.argtypes = [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mom_image_to_rgb8.argtypes = [i32, i32, i32, vp, vp, vp]
    lib.mom_profile_enable.argtypes = [i32, i32]
    lib.mom_profile_read.argtypes = [i32, C.POINTER(C.c_double), C.POINTER(C.c_longlong), i32]
    lib.mom_profile_name.restype = C.c_char_p
    lib.mom_profile_name.argtypes = [i32]
    lib.mom_knn_scratch_bytes.restype = sz
    lib.mom_knn_scratch_bytes.argtypes = [i32]
    lib.mom_knn_mean_dist2.argtypes = [i32, vp, vp, vp, vp]
    return lib


SSIM_SUM_SLOTS = 64      # MOM_SSIM_SUM_SLOTS in include/mom4d.h
SELECT_MAX_TENSORS = 32  # MOM_SELECT_MAX_TENSORS


class MomRowSelect(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("row_bytes", C.c_uint)]


# every symbol include/mom4d.h declares (tests/test_abi.py checks this list against the header)
EXPORTS = [
    "mom_version", "mom_abi_version", "mom_abi_sizeof", "mom_raster_geom_bytes", "mom_raster_image_bytes", "mom_raster_binning_bytes", "mom_raster_layout",
    "mom_raster_forward_geometry", "mom_raster_forward_render", "mom_raster_backward", "mom_mark_visible",
    "mom_selftest_wave_sum", "mom_selftest_row_reduce", "mom_hexplane_forward", "mom_hexplane_backward", "mom_hexplane_backward_lines", "mom_adam_step", "mom_l1_loss",
    "mom_plane_regulation", "mom_knn_scratch_bytes", "mom_knn_mean_dist2",
    "mom_profile_enable", "mom_profile_read", "mom_profile_name",
    "mom_morton_order_scratch_bytes", "mom_morton_order", "mom_activations_forward", "mom_activations_backward", "mom_deform_forward", "mom_deform_forward_activated", "mom_deform_backward_scratch_bytes", "mom_deform_backward", "mom_deform_backward_split",
    "mom_ssim_forward", "mom_ssim_backward", "mom_raster_backward_render", "mom_raster_backward_geometry",
    "mom_densify_stats", "mom_select_scratch_bytes", "mom_select_plan", "mom_select_apply",
    "mom_ssim_forward_slab", "mom_ssim_backward_slab",
    "mom_hexplane_backward_scratch_bytes", "mom_hexplane_orders_scratch_bytes", "mom_hexplane_orders", "mom_image_to_rgb8",
    "mom_l1_loss_acc", "mom_plane_regulation_acc", "mom_plane_regulation_grad",
    "mom_deform_field_supported", "mom_deform_field_scratch_bytes", "mom_deform_field_forward",
    "mom_stream_wait_stream", "mom_stream_mark", "mom_stream_wait_mark", "mom_zero_async",
    "mom_comm_available", "mom_comm_last_error", "mom_comm_unique_id", "mom_comm_create", "mom_comm_destroy", "mom_comm_abort",
    "mom_comm_world", "mom_comm_rank", "mom_comm_group_start", "mom_comm_group_end", "mom_comm_all_reduce", "mom_comm_all_gather",
    "mom_comm_reduce_scatter",
]


# the MOM_STRUCT_* ids of include/mom4d.h, in order, with the ctypes mirror of each struct
ABI_STRUCTS = None


def _abi_structs():
    return [("MOM_STRUCT_RASTER_ARGS", MomRasterArgs), ("MOM_STRUCT_RASTER_GRADS", MomRasterGrads),
            ("MOM_STRUCT_RASTER_LAYOUT", MomRasterLayout), ("MOM_STRUCT_HEXPLANE", MomHexPlane),
            ("MOM_STRUCT_ADAM_TENSOR", MomAdamTensor), ("MOM_STRUCT_ROW_SELECT", MomRowSelect),
            ("MOM_STRUCT_REG_PLANE", MomRegPlane), ("MOM_STRUCT_DEFORM_MLP", MomDeformMLP)]


_CTYPE_NAMES = {C.c_int: "C.c_int", C.c_uint: "C.c_uint", C.c_float: "C.c_float", C.c_void_p: "C.c_void_p",
                C.c_size_t: "C.c_size_t"}


def ctypes_mirror_source(cls, width=112):
    """Python source of the ctypes mirror of one ABI struct, generated from its `_fields_` -- the text INTEGRATION.md section 3
    shows a maintainer; tests/test_abi.py regenerates it and compares, so the document cannot drift from the binding."""
    items = [f'("{n}", {_CTYPE_NAMES[t]})' for n, t in cls._fields_]
    lines, cur = [], "    _fields_ = ["
    for i, it in enumerate(items):
        piece = it + ("," if i + 1 < len(items) else "]")
        if len(cur) + len(piece) + 1 > width:
            lines.append(cur.rstrip())
            cur = " " * 16
        cur += piece + " "
    lines.append(cur.rstrip())
    return (f"class {cls.__name__}(C.Structure):           # include/mom4d.h: {cls.__name__}, ABI version {ABI_VERSION}\n"
            + "\n".join(lines) + "\n\n"
            "    def __init__(self, *args, **kw):\n"
            "        super().__init__(*args, **kw)\n"
            "        self.struct_size = C.sizeof(type(self))   # checked by every entry point: a short struct is MOM_EINVAL\n")


def check_abi(lib):
    """Refuse a library whose ABI is not the one this mirror was written against: the version number, and sizeof of every
    argument struct as the library was compiled against the ctypes mirror's (a short struct would be read past its end)."""
    v = lib.mom_abi_version()
    if v != ABI_VERSION:
        raise MomError(f"{LIB_PATH}: ABI version {v}, this binding was written against {ABI_VERSION} (rebuild the library)")
    for which, (name, cls) in enumerate(_abi_structs()):
        n = lib.mom_abi_sizeof(which)
        if n != C.sizeof(cls):
            raise MomError(f"{LIB_PATH}: sizeof({cls.__name__}) is {n} in the library, {C.sizeof(cls)} in the binding ({name})")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MomError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
        # torch first: it brings its own copy of the HIP runtime (torch/lib/libamdhip64.so), and libmom4d must bind to THAT copy --
        # loaded before torch, libmom4d pulls in /opt/rocm's runtime, the process then holds two, and every launch on one of
        # torch's streams fails (seen as "HIP launch/runtime failure" in smoke() when build() had loaded the library first)
        import torch  # noqa: F401
        l = _sig(C.CDLL(LIB_PATH))
        lax_old = os.environ.get("MOM4D_LIB_LAX") == "1" and not hasattr(l._real, "mom_abi_version")
        if lax_old:
            import warnings
            warnings.warn(f"{LIB_PATH}: no mom_abi_version (a build older than ABI 4) bound in lax mode; struct layouts are NOT checked")
        else:
            check_abi(l)
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != MOM_OK:
        raise MomError(f"{what} failed: {_ERR.get(rc, rc)}")


def ptr(t):
    """Device pointer of a torch tensor (None / empty tensor -> NULL, like the reference's empty tensors)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def current_stream():
    """The current HIP stream's handle.  (torch.cuda.current_stream() builds a Stream object after four layers of device-index
    resolution: 10 us a call, and a training step asks a dozen times.)"""
    global _TC
    if _TC is None:
        import torch
        _TC = torch._C
    return _TC._cuda_getCurrentRawStream(_TC._cuda_getDevice())


_TC = None
