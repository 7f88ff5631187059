"""tuned_gemm: the recorded library-GEMM table ships with the package and is switched on in look-up mode only."""
import os

import torch

import istnet_amd  # noqa: F401
from istnet_amd import tuned_gemm


def test_table_is_packaged_and_names_this_image():
    assert os.path.isfile(tuned_gemm.DEFAULT_TABLE)
    rows = [ln.strip().split(",") for ln in open(tuned_gemm.DEFAULT_TABLE) if ln.strip()]
    validators = {r[1]: r[2] for r in rows if r[0] == "Validator"}
    assert validators.get("GCN_ARCH_NAME", "").startswith("gfx950")
    assert validators.get("PT_VERSION", "").split("+")[0] == torch.__version__.split("+")[0]
    ops = [r for r in rows if r[0] != "Validator"]
    assert len(ops) >= 9 and all(len(r) >= 3 for r in ops)
    # the three decoder stages' weight-gradient products (contraction over the pixels) are what the table is for
    assert any(r[1].startswith("tn_1024_18432_2304") for r in ops)


def test_enable_without_a_table_is_a_no_op(tmp_path):
    assert tuned_gemm.enable(str(tmp_path / "missing.csv")) is None      # returns before touching the device
