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
    assert lay.geom_cov3D >= 1000 * 48 and lay.bin_point_list == 0 and lay.bin_keys >= 5000 * 4      # point_list first: see mom4d.h


def test_invalid_arguments_are_rejected_without_a_gpu():
    import ctypes as C
    lib = N.lib()
    a = N.MomRasterArgs()
    a.P, a.W, a.H = 10, 0, 16          # zero width
    assert lib.mom_raster_forward_geometry(C.byref(a), None, None, None, None, None, None) == N.MOM_EINVAL
    assert lib.mom_mark_visible(-1, None, None, None, None, None) == N.MOM_EINVAL
    assert lib.mom_profile_enable(99, 1) == N.MOM_EINVAL


def test_abi_version_and_struct_sizes_are_checked_at_load():
    """mom_abi_version / mom_abi_sizeof (include/mom4d.h, "ABI versioning"): the binding refuses a library of another ABI."""
    import ctypes as C
    lib = N.lib()
    header = open(os.path.join(ROOT, "include", "mom4d.h")).read()
    assert int(re.search(r"#define MOM_ABI_VERSION (\d+)", header).group(1)) == N.ABI_VERSION == lib.mom_abi_version()
    ids = re.search(r"enum \{(.*?)\};", header, flags=re.S).group(1)
    ids = [t.strip().split("=")[0].strip() for t in ids.replace("\n", " ").split(",") if t.strip()]
    assert ids == [n for n, _ in N._abi_structs()] + ["MOM_STRUCT_COUNT"]
    for which, (_, cls) in enumerate(N._abi_structs()):
        assert lib.mom_abi_sizeof(which) == C.sizeof(cls)
    assert lib.mom_abi_sizeof(len(ids) - 1) == 0

    class Fake:                                   # a library of another ABI version
        def mom_abi_version(self):
            return N.ABI_VERSION - 1
    import pytest
    with pytest.raises(N.MomError, match="ABI version"):
        N.check_abi(Fake())


def test_a_short_or_unsized_raster_args_struct_is_refused():
    """a binder written against an older header passed a struct without the newest field.  struct_size is the
    first field now and every entry point that takes MomRasterArgs compares it with its own sizeof."""
    import ctypes as C
    lib = N.lib()
    a = N.MomRasterArgs()
    assert a.struct_size == C.sizeof(N.MomRasterArgs)
    a.P, a.W, a.H = 0, 16, 16                      # P == 0 is otherwise valid and needs no GPU
    nr = (C.c_uint * 2)()
    for size in (0, C.sizeof(N.MomRasterArgs) - 4, C.sizeof(N.MomRasterArgs) + 8):
        a.struct_size = size
        assert lib.mom_raster_forward_geometry(C.byref(a), None, None, None, nr, None, None) == N.MOM_EINVAL
        assert lib.mom_raster_backward_render(C.byref(a), None, None, 0, None, None, None, None) == N.MOM_EINVAL
        assert lib.mom_raster_backward_geometry(C.byref(a), None, None, None, None) == N.MOM_EINVAL
        assert lib.mom_raster_backward(C.byref(a), None, None, None, 0, None, None, None, None, None) == N.MOM_EINVAL


def test_integration_md_shows_the_binding_as_it_is():
    """INTEGRATION.md section 3's ctypes mirror is generated from _native.py (tools/gen_integration_stub.py); the document's
    copy must be that text, and executing it must give the binding's layout."""
    import ctypes as C
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    begin = "# --- generated from iclr2025_3d-mom_amd/_native.py"
    i = text.index("\n", text.index(begin)) + 1
    j = text.index("# --- end generated ---", i)
    assert text[i:j] == N.ctypes_mirror_source(N.MomRasterArgs), "run tools/gen_integration_stub.py"
    ns = {"C": C}
    exec(text[i:j], ns)
    doc = ns["MomRasterArgs"]
    assert [(n, t) for n, t in doc._fields_] == [(n, t) for n, t in N.MomRasterArgs._fields_]
    assert C.sizeof(doc) == C.sizeof(N.MomRasterArgs) and doc().struct_size == C.sizeof(doc)
    assert f"lib.mom_abi_version() == {N.ABI_VERSION}" in text
    # and the header's struct has the same member names in the same order
    header = open(os.path.join(ROOT, "include", "mom4d.h")).read()
    body = header[header.index("typedef struct MomRasterArgs {"):header.index("} MomRasterArgs;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    members = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if decl:
            head, *rest = decl.split(",")
            members += [re.findall(r"[A-Za-z_0-9]+", head)[-1]] + [re.findall(r"[A-Za-z_0-9]+", r)[-1] for r in rest]
    assert members == [n for n, _ in N.MomRasterArgs._fields_]
