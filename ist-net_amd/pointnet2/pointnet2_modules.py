"""PointNet++ set-abstraction (SA) and feature-propagation (FP) modules.

Mirror of model/pointnet2/pointnet2_modules.py with the same class names, constructor
arguments, tensor layouts and child-module names (``groupers``, ``mlps``, ``mlp``), so reference
checkpoints load and reference call sites work unchanged.

SA  [ref :29-73]:  FPS -> gather centroids -> per scale [ball query + group -> SharedMLP ->
                   max over nsample] -> concat scales.
FP  [ref :164-209]: three_nn -> inverse-distance weights -> three_interpolate -> concat skip
                   -> SharedMLP.
"""
import torch
import torch.nn as nn

from . import pointnet2_utils, pytorch_utils
from .fused_mlp import LazyAct, fp_level, sa_level, shared_mlp_maxpool


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def _sample_centroids(self, xyz):
        """(B,N,3) -> (B,npoint,3) by furthest point sampling, or None for GroupAll.  [ref :49-58]"""
        if self.npoint is None:
            return None
        fused = getattr(pointnet2_utils._ext, "furthest_point_sampling_gather", None)
        if fused is not None and xyz.is_cuda and not xyz.requires_grad and xyz.shape[1] <= 4096:
            return fused(xyz.contiguous(), self.npoint)[1]   # sampling + coordinate gather in one kernel
        picked = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        channels_first = xyz.transpose(1, 2).contiguous()
        return pointnet2_utils.gather_operation(channels_first, picked).transpose(1, 2).contiguous()

    def forward(self, xyz, features=None, geometry=None):
        """xyz (B,N,3), features (B,C,N) or None -> new_xyz (B,npoint,3), (B, sum(mlp[-1]), npoint).

        ``geometry`` = (new_xyz, [ball-query idx per scale][, [inverse lists of idx per scale]]) lets a caller that
        already ran the sampling / neighbour search (PointNet2MSG's geometry pre-pass) skip it here."""
        ball_csr = ball_compact = None
        if geometry is None:
            new_xyz, ball_idx = self._sample_centroids(xyz), [None] * len(self.groupers)
        elif len(geometry) == 4:     # + compact-column tables of the ball indices (padded repeats evaluated once)
            new_xyz, ball_idx, ball_csr, ball_compact = geometry
        elif len(geometry) == 3:     # + inverse lists of the ball indices (fused_mlp: atomic-free gradient scatter)
            new_xyz, ball_idx, ball_csr = geometry
        else:
            new_xyz, ball_idx = geometry
        # per scale: grouper -> SharedMLP -> max over nsample -> squeeze, then concat [ref :60-73]; on the GPU the
        # whole level is one fused node (ball-query kernels + gather-fused MFMA stacks writing the concat in place)
        if new_xyz is None:      # GroupAll variant: literal composition
            pooled = [shared_mlp_maxpool(mlp, grouper(xyz, new_xyz, features))
                      for grouper, mlp in zip(self.groupers, self.mlps)]
            return new_xyz, torch.cat(pooled, dim=1)
        return new_xyz, sa_level(list(self.groupers), list(self.mlps), xyz, new_xyz, features, ball_idx, ball_csr,
                                 ball_compact)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Multi-scale-grouping SA layer.  [ref :76-114]

    ``mlps[i][0]`` is the feature width WITHOUT xyz; 3 is added in place when ``use_xyz``
    (the reference mutates the caller's list the same way, :110-111).
    """

    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True):
        super().__init__()
        if not (len(radii) == len(nsamples) == len(mlps)):
            raise AssertionError("radii, nsamples and mlps must have one entry per scale")
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(
                pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                spec[0] += 3
            self.mlps.append(pytorch_utils.SharedMLP(spec, bn=bn))

    def flat_layout_hints(self):
        """The layer-0 weights of the scales, in scale order: adjacent in a flat parameter buffer (optim.FlatAdam) they ARE
        the stacked matrix the level-wide feature-gradient product of backward reads, so no copy is packed per step."""
        w0 = [m[0].conv.weight for m in self.mlps if len(m) > 0 and hasattr(m[0], "conv")]
        return [w0] if len(w0) > 1 and len({tuple(w.shape[1:]) for w in w0}) == 1 else []


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale SA layer.  [ref :117-145]"""

    def __init__(self, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True):
        super().__init__(npoint=npoint, radii=[radius], nsamples=[nsample], mlps=[mlp], bn=bn,
                         use_xyz=use_xyz)


class PointnetFPModule(nn.Module):
    """Feature propagation from a coarse set (known) to a dense one (unknown).  [ref :148-209]"""

    def __init__(self, mlp, bn=True):
        super().__init__()
        self.mlp = pytorch_utils.SharedMLP(mlp, bn=bn)

    @staticmethod
    def interpolation_weights(unknown, known, with_csr=False):
        """three_nn + inverse-distance weights (idx (B,n,3) i32, weight (B,n,3)).  [ref :185-188]
        with_csr: also return the inverse lists of idx used by the backward of three_interpolate."""
        fused = getattr(pointnet2_utils._ext, "three_nn_weights", None)     # absent from a plain reference _ext
        if (fused is not None and unknown.is_cuda and unknown.dtype == torch.float32
                and not unknown.requires_grad and not known.requires_grad and known.size(1) >= 3):
            idx, weight = fused(unknown.contiguous(), known.contiguous())   # one launch instead of six
        else:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            inv = 1.0 / (dist + 1e-8)
            weight = inv / torch.sum(inv, dim=2, keepdim=True)
        if with_csr:
            interp_csr = getattr(pointnet2_utils._ext, "interp_csr", None)   # absent from a plain reference _ext
            return idx, weight, (interp_csr(idx, known.size(1)) if interp_csr is not None and idx.is_cuda else None)
        return idx, weight

    def forward(self, unknown, known, unknow_feats, known_feats, interp=None, lazy_out=False):
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n) or None, known_feats (B,C2,m)
        -> (B, mlp[-1], n).  ``interp`` = precomputed (idx, weight) of interpolation_weights().
        Extensions used by PointNet2MSG between its own levels: ``known_feats`` may be a fused_mlp.LazyAct (the previous
        level's raw output + BatchNorm constants), and ``lazy_out`` asks for one back when the fused node runs."""
        lazy_in = isinstance(known_feats, LazyAct)
        if known is None:
            if lazy_in:
                known_feats = known_feats.materialize()
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        else:
            idx, weight, *csr = interp if interp is not None else self.interpolation_weights(unknown, known)
            fused = fp_level(self.mlp, known_feats, unknow_feats, idx.detach(), weight.detach(),
                             csr[0] if csr else None, lazy_out=lazy_out)
            if fused is not None:        # interpolation + concat + SharedMLP as one node (layer 0 split by linearity)
                return fused
            if lazy_in:
                known_feats = known_feats.materialize()
            if csr and csr[0] is not None:
                interpolated = pointnet2_utils.three_interpolate(known_feats, idx.detach(), weight.detach(), csr[0])
            else:
                interpolated = pointnet2_utils.three_interpolate(known_feats, idx.detach(), weight.detach())

        stacked = interpolated if unknow_feats is None else torch.cat([interpolated, unknow_feats], dim=1)
        return shared_mlp_maxpool(self.mlp, stacked.unsqueeze(-1))   # mlp(...).squeeze(-1)  [ref :205-209]
