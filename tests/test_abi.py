"""The C ABI library loads on a GPU-less machine and exports every symbol include/mom4d.h declares."""
import importlib
import os
import re

N = importlib.import_module("iclr2025_3d-mom_amd._native")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mom4d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mom_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = N.lib()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mom4d.h but not exported by libmom4d.so"
    assert sorted(N.EXPORTS) == names, "the ctypes binding and the header disagree"
    assert b"gfx950" in lib.mom_version()


def test_sizing_functions_need_no_gpu():
    lib = N.lib()
    assert lib.mom_raster_geom_bytes(1000) >= 1000 * (48 + 24 + 4 + 48)
    assert lib.mom_raster_image_bytes(960, 540) >= 960 * 540 * 8
    assert lib.mom_raster_binning_bytes(1000, 960, 540, 5000) >= 5000 * 12
    assert lib.mom_knn_scratch_bytes(1000) > 0
    lay = N.MomRasterLayout()
    import ctypes as C
    assert lib.mom_raster_layout(1000, 64, 64, 5000, C.byref(lay)) == 0
    assert lay.geom_cov3D >= 1000 * 48 and lay.bin_point_list >= 5000 * 8


def test_invalid_arguments_are_rejected_without_a_gpu():
    import ctypes as C
    lib = N.lib()
    a = N.MomRasterArgs()
    a.P, a.W, a.H = 10, 0, 16          # zero width
    assert lib.mom_raster_forward_geometry(C.byref(a), None, None, None, None, None, None) == N.MOM_EINVAL
    assert lib.mom_mark_visible(-1, None, None, None, None, None) == N.MOM_EINVAL
    assert lib.mom_profile_enable(99, 1) == N.MOM_EINVAL
