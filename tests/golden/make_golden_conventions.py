"""Index goldens under each arithmetic convention of the index-deciding squared distances.

The reference builds its kernels with plain ``-O3`` (model/pointnet2/setup.py:27): nvcc's default ``-fmad=true`` may contract
``dx*dx + dy*dy + dz*dz`` (ball_query_gpu.cu:36-37, interpolate_gpu.cu:38, sampling_gpu.cu:108-109) into fused multiply-adds.
Convention 0 is the source expression, every operation rounded; 1 is ``fma(dz,dz, fma(dx,dx, dy*dy))`` -- what LLVM's
contraction rule gives for this expression tree (profiles/r05_fma_contraction_llvm.txt) and hence what an nvcc build most
likely computes; 2 is the other possible fusion ``fma(dz,dz, fma(dy,dy, dx*dx))``.

For each convention this script drives the REFERENCE's Python (imported unmodified from /root/reference, as
make_golden.py does) over the CPU oracle set to that convention and stores
  * BASELINE config 1 (U[0,1)^3, B=4 N=1024): FPS indices and ball-query indices;
  * the PointNet2MSG encoder (cam radii, train mode) on four CUBE clouds U(-0.1,0.1)^3 chosen among 32 (seed 0) because
    their index tensors DIFFER between the conventions: every FPS / ball-query / three_nn tensor of the forward pass, the
    three_nn squared distances, and a slice of the output features.
Runs only in the build container.      python tests/golden/make_golden_conventions.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

from oracle import pn2_oracle  # noqa: E402

CAM = [[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]]


def cube_clouds():
    g = torch.Generator().manual_seed(0)
    cube = torch.rand(32, 1024, 3, generator=g) * 0.2 - 0.1
    return (cube - cube.mean(1, keepdim=True)).contiguous()


def level1_signature(xyz):
    """FPS + both level-1 ball queries of every cloud (enough to see which clouds a convention changes)."""
    fps = pn2_oracle.furthest_point_sampling(xyz, 512)
    new_xyz = torch.gather(xyz, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    return [fps] + [pn2_oracle.ball_query(new_xyz, xyz, r, s) for r, s in ((0.01, 16), (0.02, 32))]


def main():
    ref_utils, ref_modules, ref_model_modules, ref_ist, ref_rot = mg.import_reference()
    cube = cube_clouds()
    prev = pn2_oracle.set_convention(0)
    try:
        sig = {}
        for conv in (0, 1, 2):
            pn2_oracle.set_convention(conv)
            sig[conv] = level1_signature(cube)
        differs = [b for b in range(cube.shape[0])
                   if any(not torch.equal(sig[0][k][b], sig[c][k][b]) for c in (1, 2) for k in range(3))]
        print("cube clouds whose level-1 indices depend on the convention:", differs)
        chosen = (differs + [b for b in range(cube.shape[0]) if b not in differs])[:4]
        pts = cube[chosen].contiguous()
        xyz1 = torch.rand(4, 1024, 3, generator=torch.Generator().manual_seed(0))        # config 1
        out = {"pts_cube": mg.npy(pts), "cube_clouds": np.array(chosen), "xyz_config1": mg.npy(xyz1)}
        for conv in (0, 1, 2):
            pn2_oracle.set_convention(conv)
            fps = ref_utils.furthest_point_sample(xyz1, 512)
            new_xyz = ref_utils.gather_operation(xyz1.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
            out[f"c{conv}_config1_fps"] = mg.npy(fps).astype(np.int16)
            out[f"c{conv}_config1_ball"] = mg.npy(ref_utils.ball_query(0.2, 32, xyz1, new_xyz)).astype(np.int16)
            torch.manual_seed(0)
            enc = ref_model_modules.PointNet2MSG(radii_list=[list(r) for r in CAM]).train()
            orig = {n: getattr(pn2_oracle, n) for n in ("furthest_point_sampling", "ball_query", "three_nn")}
            counters = {n: 0 for n in orig}

            def tap(name):
                def fn(*a, **k):
                    res = orig[name](*a, **k)
                    i = counters[name]
                    counters[name] += 1
                    if name == "three_nn":
                        out[f"c{conv}_three_nn_dist2_{i}"] = mg.npy(res[0])
                        out[f"c{conv}_three_nn_idx_{i}"] = mg.npy(res[1]).astype(np.int16)
                    else:
                        out[f"c{conv}_{name}_{i}"] = mg.npy(res).astype(np.int16)
                    return res
                return fn
            for n in orig:
                setattr(pn2_oracle, n, tap(n))
            try:
                with torch.no_grad():
                    feats = enc(pts)
            finally:
                for n, f in orig.items():
                    setattr(pn2_oracle, n, f)
            out[f"c{conv}_out_train"] = mg.npy(feats)[:, :, ::8]
        for conv in (1, 2):
            n_diff = sum(int((out[f"c{conv}_{k[3:]}"] != out[k]).sum()) for k in list(out)
                         if k.startswith("c0_") and out[k].dtype == np.int16)
            print(f"convention {conv}: {n_diff} index entries differ from convention 0 in the stored tensors")
            assert n_diff > 0
    finally:
        pn2_oracle.set_convention(prev)
    np.savez_compressed(os.path.join(HERE, "index_conventions.npz"), **out)
    print("wrote index_conventions.npz:", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
