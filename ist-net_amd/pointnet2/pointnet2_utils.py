"""Operator wrappers with the reference's public names and signatures.

Mirror of model/pointnet2/pointnet2_utils.py: six ``autograd.Function``s
(``furthest_point_sample``, ``gather_operation``, ``three_nn``, ``three_interpolate``,
``grouping_operation``, ``ball_query``) plus ``QueryAndGroup`` / ``GroupAll``.  Everything above
this file reaches native code only through those names, so callers written against the
reference work unchanged.  The native side is ``_ext`` (this package's drop-in for
``pointnet2._ext``), which drives the gfx950 kernels.

Autograd contract (reference lines in brackets): FPS / ball_query / three_nn produce
non-differentiable outputs [:72,:283,:145]; gather / group / interpolate are differentiable
w.r.t. the feature tensor only and call ``grad_out.contiguous()`` first [:113,:199,:252].
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _ext


class RandomDropout(nn.Module):
    """Feature dropout with a per-call rate drawn from U(0, p) and NO rescaling of the kept values.  [ref :40-48]
    (The reference's forward calls ``pt_utils.feature_dropout_no_scaling``, which its pytorch_utils.py does not
    define -- the class is dead code there; this is the behaviour of the PointNet++ code it was taken from.)"""

    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.p = p
        self.inplace = inplace

    def forward(self, X):
        if not self.training:
            return X
        theta = float(torch.empty(1).uniform_(0, self.p))
        keep = (torch.rand_like(X) >= theta).to(X.dtype)
        return X.mul_(keep) if self.inplace else X * keep


class FurthestPointSampling(Function):
    """xyz (B,N,3) f32, npoint -> (B,npoint) i32 indices (first is 0).  [ref :51-80]"""

    @staticmethod
    def forward(ctx, xyz, npoint):
        picked = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(picked)
        return picked

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    """features (B,C,N), idx (B,npoint) i32 -> (B,C,npoint).  [ref :83-117]"""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.n_src = features.size(2)
        ctx.save_for_backward(idx)
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.gather_points_grad(grad_out.contiguous(), idx, ctx.n_src), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) = sqrt(d2), idx (B,n,3) i32).  [ref :120-149]"""

    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, grad_dist=None, grad_idx=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """features (B,c,m), idx (B,n,3) i32, weight (B,n,3) -> (B,c,n).  [ref :152-206]"""

    @staticmethod
    def forward(ctx, features, idx, weight, csr=None):
        # csr (optional, not in the reference): inverse lists of idx from _ext.interp_csr, built ahead of time
        ctx.m_src = features.size(2)
        ctx.csr = csr
        ctx.save_for_backward(idx, weight)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        if ctx.csr is not None:
            grad_features = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m_src, ctx.csr)
        else:
            grad_features = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m_src)
        return grad_features, None, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """features (B,C,N), idx (B,npoint,nsample) i32 -> (B,C,npoint,nsample).  [ref :209-257]"""

    @staticmethod
    def forward(ctx, features, idx):
        ctx.n_src = features.size(2)
        ctx.save_for_backward(idx)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.group_points_grad(grad_out.contiguous(), idx, ctx.n_src), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """radius, nsample, xyz (B,N,3), new_xyz (B,npoint,3) -> (B,npoint,nsample) i32.  [ref :260-291]

    NB the native call takes (new_xyz, xyz, radius, nsample) [ref :282].
    """

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        inds = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, grad=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball query + grouping around each centroid.  [ref :294-377]

    forward(xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N) or None)
      -> (B, 3+C, npoint, nsample)   [xyz offsets first, then features]
    Options follow the reference constructor: ``use_xyz``, ``ret_grouped_xyz``,
    ``normalize_xyz``, ``sample_uniformly``, ``ret_unique_cnt``.
    """

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        if ret_unique_cnt and not sample_uniformly:
            raise AssertionError("ret_unique_cnt requires sample_uniformly")
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt

    def _resample_uniformly(self, idx):
        # host-side re-draw of the padded slots among the unique hits [ref :337-346]
        unique_cnt = torch.zeros((idx.shape[0], idx.shape[1]))
        for bi in range(idx.shape[0]):
            for ri in range(idx.shape[1]):
                uniq = torch.unique(idx[bi, ri, :])
                k = uniq.shape[0]
                unique_cnt[bi, ri] = k
                draw = torch.randint(0, k, (self.nsample - k,), dtype=torch.long, device=uniq.device)
                idx[bi, ri, :] = torch.cat((uniq, uniq[draw]))
        return unique_cnt

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        unique_cnt = self._resample_uniformly(idx) if self.sample_uniformly else None

        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)  # (B,3,npoint,nsample)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz /= self.radius

        if features is None:
            if not self.use_xyz:
                raise AssertionError("Cannot have not features and not use xyz as a feature!")
            new_features = grouped_xyz
        else:
            grouped_features = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz \
                else grouped_features

        extras = []
        if self.ret_grouped_xyz:
            extras.append(grouped_xyz)
        if self.ret_unique_cnt:
            extras.append(unique_cnt)
        return (new_features, *extras) if extras else new_features


class GroupAll(nn.Module):
    """Single group holding every point.  [ref :380-427]  -> (B, 3+C, 1, N)"""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            new_features = grouped_xyz
        else:
            grouped_features = features.unsqueeze(2)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz \
                else grouped_features
        return (new_features, grouped_xyz) if self.ret_grouped_xyz else new_features
