"""How many index decisions depend on the FMA-contraction convention of the squared distances?

The reference CUDA kernels are built with nvcc defaults (-fmad=true), so a real reference build may compute
dx*dx + dy*dy + dz*dz with fused multiply-adds, while the product, the oracle and the golden vectors use the
un-contracted source order.  This script runs the CPU oracle under the three conventions
(oracle/pn2_oracle.c: 0 un-contracted, 1 fma(dz,dz,fma(dx,dx,dy*dy)), 2 fma(dz,dz,fma(dy,dy,dx*dx))) on the
benchmark clouds and counts the differing entries of every index tensor of the encoder's geometry pass:
FPS picks, ball-query memberships, three_nn neighbour indices.  CPU only:

    python tools/fma_flip_table.py > profiles/r02_fma_convention_flips.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CAM_RADII, shell_cloud  # noqa: E402
from oracle import pn2_oracle as orc  # noqa: E402

NPOINTS = [512, 256, 128, 64]
NSAMPLES = [[16, 32], [16, 32], [16, 32], [16, 32]]     # model/modules.py:249-304
WORLD_RADII = [[0.05, 0.10], [0.10, 0.20], [0.20, 0.30], [0.30, 0.40]]   # ist_net.py:17


def geometry(xyz, radii):
    """Index tensors of one PointNet2MSG geometry pass (4 SA levels + 4 FP levels) under the current convention."""
    out = {}
    levels = [xyz]
    for lv, (npoint, rr, ns) in enumerate(zip(NPOINTS, radii, NSAMPLES)):
        cur = levels[-1]
        fps = orc.furthest_point_sampling(cur, npoint)
        out[f"fps_L{lv + 1}"] = fps
        new = torch.gather(cur, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        for si, (r, s) in enumerate(zip(rr, ns)):
            out[f"ball_L{lv + 1}s{si}"] = orc.ball_query(new, cur, r, s)
        levels.append(new)
    for lv in range(4):
        _, idx = orc.three_nn(levels[lv], levels[lv + 1])
        out[f"three_nn_L{lv}"] = idx
    return out


def clouds():
    g = torch.Generator().manual_seed(0)
    yield "config 1: U[0,1)^3, B=4 N=1024, r=0.2 nsample=32 (one SA layer)", torch.rand(4, 1024, 3, generator=g), None
    yield "config 2 shell: radius 0.1 sigma 0.002, B=32 N=1024, cam radii", shell_cloud(32, 1024, 0), CAM_RADII
    g = torch.Generator().manual_seed(0)
    cube = torch.rand(32, 1024, 3, generator=g) * 0.2 - 0.1
    yield "config 2 cube: U(-0.1,0.1)^3, B=32 N=1024, cam radii", (cube - cube.mean(1, keepdim=True)).contiguous(), CAM_RADII
    g = torch.Generator().manual_seed(5)
    yield ("config 3/5 world cloud: qo ~ U(-0.5,0.5)^3, B=32 N=1024, world radii",
           (torch.rand(32, 1024, 3, generator=g) - 0.5).contiguous(), WORLD_RADII)
    yield "config 5 shell, B=64 N=2048, cam radii", shell_cloud(64, 2048, 5), CAM_RADII


def config1(xyz):
    fps = orc.furthest_point_sampling(xyz, 512)
    new = torch.gather(xyz, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    return {"fps": fps, "ball_r0.2_s32": orc.ball_query(new, xyz, 0.2, 32)}


def main():
    print("# index decisions that change with the FMA-contraction convention of the squared distances")
    print("# conventions: 0 = ((dx*dx + dy*dy) + dz*dz) un-contracted [product / oracle / goldens];")
    print("#              1 = fma(dz,dz,fma(dx,dx,dy*dy));  2 = fma(dz,dz,fma(dy,dy,dx*dx))")
    print("# entries = elements of the index tensor; a later level's count includes the effect of earlier flips")
    print("# (a different FPS pick changes every centroid after it).  CPU oracle, tools/fma_flip_table.py")
    for name, xyz, radii in clouds():
        res = {}
        for conv in (0, 1, 2):
            orc.set_convention(conv)
            res[conv] = config1(xyz) if radii is None else geometry(xyz, radii)
        orc.set_convention(0)
        print(f"\n## {name}")
        print(f"{'tensor':<16}{'entries':>10}{'differ 1 vs 0':>16}{'differ 2 vs 0':>16}{'differ 1 vs 2':>16}")
        tot = [0, 0, 0, 0]
        for key, ref in res[0].items():
            d1 = int((res[1][key] != ref).sum())
            d2 = int((res[2][key] != ref).sum())
            d12 = int((res[1][key] != res[2][key]).sum())
            print(f"{key:<16}{ref.numel():>10}{d1:>16}{d2:>16}{d12:>16}")
            tot = [tot[0] + ref.numel(), tot[1] + d1, tot[2] + d2, tot[3] + d12]
        print(f"{'total':<16}{tot[0]:>10}{tot[1]:>16}{tot[2]:>16}{tot[3]:>16}"
              f"    ({100.0 * tot[1] / tot[0]:.4f} % / {100.0 * tot[2] / tot[0]:.4f} %)")


if __name__ == "__main__":
    main()
