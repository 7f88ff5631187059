"""Library GEMMs of the RGB decoder (the channel-mixing products of rgb_branch.PSPUpsample / PSPModule, run by
hipBLASLt) with recorded solutions instead of the library's heuristic pick.

The decoder's weight-gradient products contract over the pixel dimension (18 432 - 294 912 rows) into a small output;
hipBLASLt's default choice for them runs at 35-65 TFLOP/s, its best solution (found by PyTorch's TunableOp search) at
65-130 (tools/exp/decoder_gemm_libs.py: the three stages' nine products 5.13 -> 3.35 ms).  ``enable()`` switches TunableOp
on in look-up mode with the table recorded on an MI355X for this image's library versions
(``tuning/tunableop_gfx950.csv``, validated by TunableOp against the running libraries: a table from other versions is
ignored, and shapes that are not in the table run the default solution).  ``enable(tune=True, path=...)`` searches and
records -- minutes, on a GPU box: ``python bench.py --workload istnet --tune-gemms``."""
import os

DEFAULT_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_gfx950.csv")


def enable(path=None, tune=False):
    """Returns the table's path when TunableOp was switched on, None when there is no table to look up."""
    import torch
    path = path or DEFAULT_TABLE
    if not tune and not os.path.isfile(path):
        return None
    tn = torch.cuda.tunable
    tn.set_filename(path)
    tn.enable(True)
    tn.tuning_enable(bool(tune))
    if tune:
        tn.set_max_tuning_duration(100)      # ms per candidate solution
        tn.set_max_tuning_iterations(20)
    return path


def disable():
    import torch
    torch.cuda.tunable.enable(False)
