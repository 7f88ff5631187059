"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/istnet_pn2.h declares, and the Python binding table matches the header."""
import ctypes
import re

import pytest
import torch


def test_library_exports_every_declared_symbol():
    from istnet_amd import _native
    names = _native.declared_symbols()
    assert len(names) >= 23 and "istnet_pn2_query_ball_point" in names and "istnet_pw_forward" in names
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), f"{name} declared in istnet_pn2.h but not exported"


def test_binding_table_matches_header():
    from istnet_amd import _native
    text = "".join(open(h).read() for h in _native.HEADER_PATHS)
    decls = dict(re.findall(r"ISTNET_PN2_API\s+int\s+(istnet_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S))
    decls.pop("istnet_pn2_abi_version")
    assert set(decls) == set(_native.SIGNATURES)
    for name, args in decls.items():
        kinds = []
        for a in ([] if args.strip() == "void" else args.split(",")):
            a = a.strip()
            kinds.append(ctypes.c_void_p if "*" in a else ctypes.c_float if a.startswith("float")
                         else ctypes.c_double if a.startswith("double")
                         else ctypes.c_longlong if a.startswith("long long") else ctypes.c_int)
        assert kinds == _native.SIGNATURES[name], name


def test_abi_version_and_target():
    from istnet_amd import _native
    lib = _native.lib()
    assert lib.istnet_pn2_abi_version() == _native.ABI_VERSION == 1
    assert lib.istnet_pn2_target() == b"gfx950"


def test_every_kernel_is_gfx950_code():
    """The shared object embeds a gfx950 code object (no other targets, no fallback)."""
    from istnet_amd import _native
    blob = open(_native.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_"):
        assert b"amdhsa--" + other not in blob


def test_product_ext_rejects_cpu_tensors():
    """Reference behaviour: TORCH_CHECK(false, "CPU not supported") in every host entry."""
    from istnet_amd.pointnet2 import _ext
    x = torch.rand(1, 8, 3)
    f = torch.rand(1, 4, 8)
    i2 = torch.zeros(1, 4, dtype=torch.int32)
    i3 = torch.zeros(1, 4, 2, dtype=torch.int32)
    i33 = torch.zeros(1, 8, 3, dtype=torch.int32)
    calls = [lambda: _ext.furthest_point_sampling(x, 4), lambda: _ext.ball_query(x, x, 0.1, 4),
             lambda: _ext.three_nn(x, x), lambda: _ext.gather_points(f, i2),
             lambda: _ext.gather_points_grad(torch.rand(1, 4, 4), i2, 8), lambda: _ext.group_points(f, i3),
             lambda: _ext.group_points_grad(torch.rand(1, 4, 4, 2), i3, 8),
             lambda: _ext.three_interpolate(f, i33, torch.rand(1, 8, 3)),
             lambda: _ext.three_interpolate_grad(torch.rand(1, 4, 8), i33, torch.rand(1, 8, 3), 8)]
    for c in calls:
        with pytest.raises(RuntimeError, match="CPU not supported"):
            c()


def test_product_does_not_import_oracle():
    """The product package must not import, load or include the oracle anywhere."""
    import os
    import istnet_amd
    root = os.path.dirname(istnet_amd.__file__)
    bad = re.compile(r"^\s*(from|import)\s+oracle|libpn2_oracle|#include.*oracle|import_module\(.oracle", re.M)
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert not bad.search(open(os.path.join(dp, f)).read()), f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No HIP library -> RuntimeError naming the file from the one loader every product entry point goes through
    (istnet_amd._native.lib); there is no torch or CPU fallback to land on."""
    from istnet_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "libistnet_pn2.so"))
    with pytest.raises(RuntimeError, match="libistnet_pn2.so is missing"):
        _native.lib()
    # every module of the product that launches kernels does so through that loader
    import os
    import istnet_amd
    root = os.path.dirname(istnet_amd.__file__)
    for rel in ("pointnet2/_ext.py", "pointnet2/fused_mlp.py", "optim.py", "preprocess.py"):
        text = open(os.path.join(root, rel)).read()
        assert "_native.lib()" in text and "ctypes.CDLL" not in text, rel
