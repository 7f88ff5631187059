"""Per-level fp32 error budget of the encoder (GPU): where does the distance to a float64 evaluation come from?

Three evaluations of PointNet2MSG (train-mode BatchNorm) on the same input with the SAME index decisions
(FPS / ball query / three_nn from the bit-exact HIP kernels):

    hip    the product: fused MFMA stacks, fp32
    torch  the reference's composition in fp32 on the same GPU: torch Conv2d / BatchNorm2d / ReLU / max_pool2d over
           the materialised grouped tensors (the fused path disabled) -- what the reference computes, numerically
    f64    the same composition in float64 = the arithmetic truth

For every SA / FP level, in execution order:
    cumulative error   |level output - f64 output| with each evaluation running on its own earlier levels
    local error        |level(f32(f64 inputs)) - f64 output|: the error the level ADDS when fed the exact inputs
                       (the gap between the two is amplification of inherited error by the later levels)

    python tools/error_budget.py [golden|shell32] > profiles/r02_error_budget.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import CAM_RADII, shell_cloud  # noqa: E402
from istnet_amd.modules import PointNet2MSG  # noqa: E402
from istnet_amd.pointnet2 import _ext, fused_mlp, pointnet2_utils  # noqa: E402
from test_golden_gpu import _F64Ext  # noqa: E402

DEV = "cuda:0"


class no_fusion:
    """Force the reference composition (torch ops over grouped tensors) for every SharedMLP."""

    def __enter__(self):
        self.saved = fused_mlp._fusable_shape
        fused_mlp._fusable_shape = lambda *a, **k: False

    def __exit__(self, *exc):
        fused_mlp._fusable_shape = self.saved


def run_levels(enc, pts, capture_inputs=False):
    """Forward with hooks: returns ([(name, output features)], [(name, input args)])."""
    outs, ins, hooks = [], [], []
    for kind, mods in (("SA", enc.SA_modules), ("FP", enc.FP_modules)):
        for i, m in enumerate(mods):
            name = f"{kind}{i + 1}"

            def hook(mod, args, kwargs, out, name=name):
                feat = out[1] if isinstance(out, tuple) else out
                if isinstance(feat, fused_mlp.LazyAct):      # levels 3..1 of the feature propagation hand on raw output + constants
                    feat = feat.materialize()
                outs.append((name, feat.detach()))
                if capture_inputs:
                    ins.append((name, args, kwargs))
            hooks.append(m.register_forward_hook(hook, with_kwargs=True))
    enc(pts)
    for h in hooks:
        h.remove()
    return outs, ins


def stats(a, ref):
    d = (a.double() - ref).abs()
    scale = ref.abs().max().item()
    return d.max().item(), d.pow(2).mean().sqrt().item(), scale


def compute(which="golden"):
    """Rows (level, scale of the output, hip / torch cumulative and local max / rms error vs float64) + the input label."""
    if which == "golden":
        z = np.load(os.path.join(ROOT, "tests", "golden", "encoder_b2.npz"))
        pts = torch.from_numpy(z["pts"]).to(DEV)
        label = "tests/golden/encoder_b2.npz input (B=2, N=1024), seed-0 weights"
    else:
        pts = shell_cloud(32, 1024, 0, DEV)
        label = "shell clouds B=32 N=1024 (the bench batch), seed-0 weights"
    torch.manual_seed(0)
    base = PointNet2MSG([list(r) for r in CAM_RADII]).train()
    state = {k: v.clone() for k, v in base.state_dict().items()}

    def fresh(double=False):
        m = PointNet2MSG([list(r) for r in CAM_RADII]).to(DEV).train()
        m.load_state_dict(state)
        return m.double() if double else m

    saved = pointnet2_utils._ext
    try:
        pointnet2_utils._ext = _F64Ext(_ext)
        out64, in64 = run_levels(fresh(True), pts.double(), capture_inputs=True)
    finally:
        pointnet2_utils._ext = saved
    out_hip, _ = run_levels(fresh(), pts)
    with no_fusion():
        out_torch, _ = run_levels(fresh(), pts)

    def local(fused):
        """Each level alone on the float64 run's inputs rounded to fp32."""
        res = []
        enc = fresh()
        mods = dict([(f"SA{i + 1}", m) for i, m in enumerate(enc.SA_modules)] +
                    [(f"FP{i + 1}", m) for i, m in enumerate(enc.FP_modules)])
        for name, args, kwargs in in64:
            cast = lambda t: t.float().contiguous() if torch.is_tensor(t) and t.is_floating_point() else t
            a = [cast(t) for t in args]
            kw = {}
            if "interp" in kwargs and kwargs["interp"] is not None:
                kw["interp"] = tuple(cast(t) if torch.is_tensor(t) else t for t in kwargs["interp"][:2])
            if "geometry" in kwargs and kwargs["geometry"] is not None:
                g = kwargs["geometry"]
                kw["geometry"] = (cast(g[0]), g[1])
            if fused:
                out = mods[name](*a, **kw)
            else:
                with no_fusion():
                    out = mods[name](*a, **kw)
            res.append((name, (out[1] if isinstance(out, tuple) else out).detach()))
        return res

    loc_hip, loc_torch = local(True), local(False)
    rows = []
    for (name, ref), (_, a), (_, b), (_, la), (_, lb) in zip(out64, out_hip, out_torch, loc_hip, loc_torch):
        ha, hr, sc = stats(a, ref)
        ta, tr, _ = stats(b, ref)
        lha, lhr, _ = stats(la, ref)
        lta, ltr, _ = stats(lb, ref)
        rows.append({"level": name, "scale": sc, "hip_cum_max": ha, "hip_cum_rms": hr, "hip_loc_max": lha, "hip_loc_rms": lhr,
                     "torch_cum_max": ta, "torch_cum_rms": tr, "torch_loc_max": lta, "torch_loc_rms": ltr})
    ref = out64[-1][1]
    final = {}
    for tag, t in (("hip", out_hip[-1][1]), ("torch", out_torch[-1][1])):
        d = (t.double() - ref).abs()
        final[tag] = (d.max().item(), (d > 1e-4 + 1e-4 * ref.abs()).double().mean().item())
    return rows, final, label


def main():
    rows, final, label = compute(sys.argv[1] if len(sys.argv) > 1 else "golden")
    print(f"# fp32 error budget of the encoder vs a float64 evaluation with identical index decisions")
    print(f"# input: {label}")
    print(f"# hip = fused MFMA path (product); torch = Conv2d/BatchNorm2d/ReLU/max_pool2d composition in fp32 on the same GPU")
    print(f"# cumulative: own earlier levels; local: the level alone on the float64 inputs rounded to fp32")
    print(f"{'level':<6}{'|out|max':>10} | {'hip cum max':>12}{'hip cum rms':>12}{'hip loc max':>12}{'hip loc rms':>12} | "
          f"{'torch cum max':>14}{'torch cum rms':>14}{'torch loc max':>14}{'torch loc rms':>14}")
    for r in rows:
        print(f"{r['level']:<6}{r['scale']:>10.3f} | {r['hip_cum_max']:>12.2e}{r['hip_cum_rms']:>12.2e}{r['hip_loc_max']:>12.2e}"
              f"{r['hip_loc_rms']:>12.2e} | {r['torch_cum_max']:>14.2e}{r['torch_cum_rms']:>14.2e}{r['torch_loc_max']:>14.2e}"
              f"{r['torch_loc_rms']:>14.2e}")
    for tag in ("hip", "torch"):
        print(f"# final output, {tag}: max |err| {final[tag][0]:.3e}, fraction outside atol=rtol=1e-4: {final[tag][1]:.2e}")


if __name__ == "__main__":
    main()
