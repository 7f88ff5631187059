"""Point-cloud encoder of IST-Net.

``PointNet2MSG`` mirrors model/modules.py:244-327: four 2-scale MSG set-abstraction levels
(npoint 512/256/128/64, nsample [16, 32]) followed by four feature-propagation levels, returning
per-point features (B, 128, N).  Child names (``SA_modules``, ``FP_modules``) and layer widths are
the reference's so its state dicts load unchanged.
"""
import torch.nn as nn

from .pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleMSG

# (npoint, per-scale output width) of the four SA levels  [ref :249-297]
_SA_LEVELS = ((512, (16, 16, 32)), (256, (32, 32, 64)), (128, (64, 64, 128)), (64, (128, 128, 256)))
_NSAMPLES = (16, 32)


class PointNet2MSG(nn.Module):
    def __init__(self, radii_list, use_xyz=True):
        super().__init__()
        self.SA_modules = nn.ModuleList()
        widths = [0]  # feature width entering each level (level 0 has xyz only)
        for (npoint, hidden), radii in zip(_SA_LEVELS, radii_list):
            c_in = widths[-1]
            self.SA_modules.append(PointnetSAModuleMSG(
                npoint=npoint, radii=list(radii), nsamples=list(_NSAMPLES),
                mlps=[[c_in, *hidden] for _ in _NSAMPLES], use_xyz=use_xyz, bn=True))
            widths.append(hidden[-1] * len(_NSAMPLES))
        c0, c1, c2, c3 = widths[1:]
        # FP_modules[i] refines level i from level i+1  [ref :299-304]
        self.FP_modules = nn.ModuleList([
            PointnetFPModule(mlp=[256, 128, 128], bn=True),
            PointnetFPModule(mlp=[256 + c0, 256, 256], bn=True),
            PointnetFPModule(mlp=[512 + c1, 256, 256], bn=True),
            PointnetFPModule(mlp=[c3 + c2, 512, 512], bn=True),
        ])

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def forward(self, pointcloud):
        """(B, N, 3[+C]) -> (B, 128, N)."""
        xyz, features = self._break_up_pc(pointcloud)
        l_xyz, l_features = [xyz], [features]
        for sa in self.SA_modules:
            nxt_xyz, nxt_feat = sa(l_xyz[-1], l_features[-1])
            l_xyz.append(nxt_xyz)
            l_features.append(nxt_feat)
        for lvl in range(len(self.FP_modules) - 1, -1, -1):  # coarse -> fine  [ref :322-325]
            l_features[lvl] = self.FP_modules[lvl](l_xyz[lvl], l_xyz[lvl + 1], l_features[lvl],
                                                   l_features[lvl + 1])
        return l_features[0]
