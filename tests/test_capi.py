"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/istnet_pn2.h declares, and the Python binding table matches the header."""
import ctypes
import os
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
    assert lib.istnet_pn2_abi_version() == _native.ABI_VERSION == 2
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


def test_no_kernel_has_a_scratch_segment(tmp_path):
    """A kernel with a scratch (private) segment does not share the chip with kernels of other streams (docs/DESIGN_rounds_1_2.md 5.4:
    a 128-workgroup finalize kernel waited 55 us for the GEMM beside it), and register spills cost bandwidth on their own:
    every kernel of the built library must report private_segment_fixed_size 0 and no spills."""
    import os
    import re
    import subprocess
    from istnet_amd import _native
    tools = "/opt/rocm/lib/llvm/bin"
    objcopy, bundler, readelf = (os.path.join(tools, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))
    if not all(os.path.exists(t) for t in (objcopy, bundler, readelf)):
        pytest.skip("LLVM binary tools of the ROCm install not found")
    _native.lib()
    fat = tmp_path / "fat.bin"
    subprocess.run([objcopy, f"--dump-section=.hip_fatbin={fat}", _native.LIB_PATH, str(tmp_path / "discard.so")], check=True)
    blob = fat.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    assert len(starts) >= 5          # one bundle per .hip source
    kernels = 0
    for i, a in enumerate(starts):
        part = tmp_path / f"bundle{i}.bin"
        part.write_bytes(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = tmp_path / f"dev{i}.co"
        subprocess.run([bundler, "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--output={co}"], check=True)
        notes = subprocess.run([readelf, "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        names = re.findall(r"\.name:\s+(\S+)", notes)
        scratch = [int(v) for v in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
        spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", notes)]
        kernels += len(scratch)
        bad = [n for n, sc, sp in zip([n for n in names if not n.startswith(("hidden_", "by_"))], scratch, spills) if sc or sp]
        assert not any(scratch) and not any(spills), bad
    assert kernels > 100


def test_pmc_traffic_summary_is_tied_to_the_kernel_sources(tmp_path):
    """bench.py reports roofline.traffic from a committed rocprofv3 PMC summary; a summary collected on other kernel
    sources must come back as null with the reason, never as a number."""
    import json
    from istnet_amd import roofline
    rec = {"kernels": {"some_kernel": {"hbm_bytes_per_launch": 123.0}}}
    f = tmp_path / "t.json"
    f.write_text(json.dumps(dict(rec, kernel_source_sha256=roofline.kernel_source_hash())))
    assert roofline.pmc_traffic("some_kernel", str(f)) == (123.0, "t.json")
    assert roofline.pmc_traffic("other_kernel", str(f)) == (None, None)
    f.write_text(json.dumps(dict(rec, kernel_source_sha256="0" * 64)))
    val, why = roofline.pmc_traffic("some_kernel", str(f))
    assert val is None and why.startswith("stale")
    f.write_text(json.dumps(rec))                      # summaries from before the stamp existed
    assert roofline.pmc_traffic("some_kernel", str(f))[0] is None


def test_sq_counter_summary_follows_the_same_staleness_rule(tmp_path):
    """roofline.mfma_busy_frac comes from a committed SQ-counter summary (tools/pmc_sq.sh) under the same rule."""
    import json
    from istnet_amd import roofline
    rec = {"kernels": {"k<2>": {"mfma_busy_frac": 0.4}}}
    f = tmp_path / "sq.json"
    f.write_text(json.dumps(dict(rec, kernel_source_sha256=roofline.kernel_source_hash())))
    got, src = roofline.pmc_sq("k<2>", str(f))
    assert got["mfma_busy_frac"] == 0.4 and src == "sq.json"
    assert roofline.pmc_sq("other", str(f)) == (None, None)
    f.write_text(json.dumps(dict(rec, kernel_source_sha256="0" * 64)))
    got, why = roofline.pmc_sq("k<2>", str(f))
    assert got is None and why.startswith("stale")


def test_committed_pmc_summaries_belong_to_these_kernel_sources():
    """The summaries bench.py names (PMC_TRAFFIC_FILE / PMC_SQ_FILE) were collected on the kernel sources in this tree: a kernel
    edit without a re-collection makes the bench line say `traffic: null` -- and fails here, where it is noticed."""
    import json
    import bench
    from istnet_amd import roofline
    for name in (bench.PMC_TRAFFIC_FILE, bench.PMC_SQ_FILE):
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", name)
        assert os.path.exists(path), name
        assert json.load(open(path))["kernel_source_sha256"] == roofline.kernel_source_hash(), name


def test_entry_points_reject_invalid_arguments_before_touching_the_device():
    """Argument validation is host code ahead of any HIP call: an invalid size or a missing pointer returns
    ISTNET_PN2_EINVAL (100001) on a box without a GPU too -- never a crash, never a launch."""
    from istnet_amd import _native
    lib = _native.lib()
    EINVAL = 100001
    one = 16                       # a non-null, 16-byte aligned "pointer" that is never dereferenced on these paths
    assert lib.istnet_mse_value_grad(0, one, None, one, one, one, None) == EINVAL
    assert lib.istnet_mse_value_grad(8, None, None, one, one, one, None) == EINVAL
    assert lib.istnet_mse_value_grad(8, one + 4, None, one, one, one, None) == EINVAL            # misaligned operand
    assert lib.istnet_crop_resize_normalize(1, 0, 640, one, 0, 1, one, 192, one, one, None, one, None) == EINVAL
    assert lib.istnet_crop_resize_normalize(1, 480, 640, one, 0, 1, one, 192, one, one, None, None, None) == EINVAL
    assert lib.istnet_crop_resize_normalize(-1, 480, 640, one, 0, 1, one, 192, one, one, None, one, None) == EINVAL
    assert lib.istnet_nhwc_bn_act_res_apply(0, 64, 64, one, one, one, one, one, None) == EINVAL
    assert lib.istnet_nhwc_bn_act_res_apply(2, 64, 6, one, one, one, one, one, None) == EINVAL   # channels % 4
    assert lib.istnet_nhwc_bn_act_res_bwd_stats(2, 64, 64, one, one, None, one, one, one, one, one, one, None) == EINVAL
    assert lib.istnet_fc_forward(0, 32, 512, None, None, None, None, None, 1, None) == EINVAL
    assert lib.istnet_ortho6d_forward(0, one, one, None) == EINVAL
    assert lib.istnet_smooth_l1_forward(0, 0.1, one, one, one, one, None) == EINVAL
    assert lib.istnet_backproject_choose(1, 4, 0, 640, one, 0, 0, one, one, 1.0, 1.0, 0.0, 0.0, 1000.0, 192, one, one,
                                         None) == EINVAL
    assert lib.istnet_pw_set_tuning(999, 1) == EINVAL
    assert lib.istnet_mse_parts(1) == 1 and lib.istnet_mse_parts(1 << 30) == 1024
    assert lib.istnet_nhwc_stat_parts(0) == 0 and lib.istnet_nhwc_stat_parts(18432) == 576
