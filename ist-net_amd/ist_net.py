"""IST head and pose heads of IST-Net (point branch).

Mirror of model/ist_net.py: ``FeatureDeformer`` / ``ImplicitTransformation`` (:114-183),
``WorldSpaceEnhancer`` (:185-200), ``LightEstimator`` (:202-264), ``HeavyEstimator`` (:267-332)
and the point-branch wiring of ``IST_Net.forward`` (:22-76).  Sub-module names and
``nn.Sequential`` indices equal the reference's, so its state dicts load unchanged
(e.g. ``implicit_transform.feature_refine.deform_mlp2.4.weight``).  Device-agnostic: the reference's
``.cuda()`` literal at :38 becomes ``device=pts.device``.

The RGB branch (ResNet-18 + PSPNet, model/modules.py:10-81) is outside the hot path
(SURVEY.md 8f); ``IST_Net`` takes any module that maps rgb (B,3,H,W) -> (B,128,H,W), or
pre-computed per-point RGB features through ``inputs['rgb_local']``.
"""
import os

import torch
import torch.nn as nn

from . import modules as enc_modules
from .modules import PointNet2MSG
from .pointnet2.fused_mlp import pointwise_conv_stack as _run
from .pointnet2.fused_mlp import pointwise_conv_stack_multi as _run_multi
from . import _native, graphed, heads_native
from .rotation_utils import Ortho6d2Mat, ortho6d_to_mat

CAM_RADII = [[0.01, 0.02], [0.02, 0.04], [0.04, 0.08], [0.08, 0.16]]      # ist_net.py:16
WORLD_RADII = [[0.05, 0.10], [0.10, 0.20], [0.20, 0.30], [0.30, 0.40]]    # ist_net.py:189


def _pointwise(widths, final_relu=True):
    """Conv1d(k=1) -> ReLU chain over (B, C, N); indices 0,2,4.. are the convs."""
    layers = []
    for i in range(len(widths) - 1):
        layers.append(nn.Conv1d(widths[i], widths[i + 1], 1))
        if final_relu or i < len(widths) - 2:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


def _fc_head(out_dim):
    return nn.Sequential(nn.Linear(512, 512), nn.ReLU(), nn.Linear(512, 256), nn.ReLU(),
                         nn.Linear(256, out_dim))


def _pooled(seq, feat):
    """``seq`` = [conv, relu, conv, relu, AdaptiveAvgPool1d(1)] (pose_mlp2) on [feat, global mean of feat] (:256-257,
    :324-325): fused convs with the mean as a per-cloud bias, then the mean over N."""
    return _run_multi(seq[:-1], [feat], with_mean=True, pool_mean=True)


class FeatureDeformer(nn.Module):
    """Camera-space features -> world-space features + per-class NOCS coordinates.  [ref :123-183]"""

    def __init__(self, nclass=6):
        super().__init__()
        self.nclass = nclass
        self.pts_mlp1 = _pointwise([3, 32, 64])
        self.deform_mlp1 = _pointwise([64 + 256, 384, 256])
        self.deform_mlp2 = _pointwise([512, 384, 256, 128])
        self.pred_nocs = _pointwise([128, 256, 128, nclass * 3], final_relu=False)

    def forward(self, pts, rgb_local, pts_local, index):
        npoint = pts_local.size(2)
        geo = _run(self.pts_mlp1, pts.transpose(1, 2))
        feat = _run_multi(self.deform_mlp1, [geo, pts_local, rgb_local])          # :167-171 without the concat
        pts_local_w = _run_multi(self.deform_mlp2, [feat], with_mean=True)        # :174-175 without expand + concat
        nocs = _run(self.pred_nocs, pts_local_w).view(-1, 3, npoint).contiguous()   # (B*nclass, 3, N)
        pts_w = torch.index_select(nocs, 0, index).permute(0, 2, 1).contiguous()
        return pts_local_w, pts_w


class ImplicitTransformation(nn.Module):
    def __init__(self, nclass=6):
        super().__init__()
        self.nclass = nclass
        self.feature_refine = FeatureDeformer(nclass)

    def forward(self, rgb_local, pts_local, pts, center, index):
        pts_local_w, pts_w = self.feature_refine(pts, rgb_local, pts_local, index)
        return pts_w, pts_local_w


class _PoseHeads(nn.Module):
    """Shared tail of both estimators: pooled 512-d feature -> R (via 6-D), t, s."""

    def _make_heads(self):
        self.rotation_estimator = _fc_head(6)
        self.translation_estimator = _fc_head(3)
        self.size_estimator = _fc_head(3)

    def _pose(self, pooled):
        fused = heads_native.fc_heads([self.rotation_estimator, self.translation_estimator, self.size_estimator], pooled)
        if fused is not None:       # the three heads layer by layer in three launches, Ortho6d2Mat in one
            r6, t, s = fused
            return ortho6d_to_mat(r6).view(-1, 3, 3), t, s
        r6 = self.rotation_estimator(pooled)
        r = Ortho6d2Mat(r6[:, :3].contiguous(), r6[:, 3:].contiguous()).view(-1, 3, 3)
        return r, self.translation_estimator(pooled), self.size_estimator(pooled)


class LightEstimator(_PoseHeads):
    """Camera-space auxiliary pose head.  [ref :202-264]"""

    def __init__(self):
        super().__init__()
        self.pts_mlp = _pointwise([3, 32, 64])
        self.pose_mlp1 = _pointwise([128 + 64 + 128, 256, 256])
        self.pose_mlp2 = nn.Sequential(*_pointwise([512, 512, 512]), nn.AdaptiveAvgPool1d(1))
        self._make_heads()

    def forward(self, pts, rgb_local, pts_local):
        geo = _run(self.pts_mlp, pts.transpose(1, 2))
        feat = _run_multi(self.pose_mlp1, [rgb_local, geo, pts_local])
        return self._pose(_pooled(self.pose_mlp2, feat))


class HeavyEstimator(_PoseHeads):
    """Main pose head over camera- and world-space cues.  [ref :267-332]"""

    def __init__(self):
        super().__init__()
        self.pts_mlp1 = _pointwise([3, 32, 64])
        self.pts_mlp2 = _pointwise([3, 32, 64])
        self.pose_mlp1 = _pointwise([64 + 64 + 384, 256, 256])
        self.pose_mlp2 = nn.Sequential(*_pointwise([512, 512, 512]), nn.AdaptiveAvgPool1d(1))
        self._make_heads()

    def forward(self, pts, pts_w, rgb_local, pts_local, pts_w_local):
        geo = _run(self.pts_mlp1, pts.transpose(1, 2))
        geo_w = _run(self.pts_mlp2, pts_w.transpose(1, 2))
        feat = _run_multi(self.pose_mlp1, [rgb_local, geo, pts_local, geo_w, pts_w_local])
        return self._pose(_pooled(self.pose_mlp2, feat))


class WorldSpaceEnhancer(nn.Module):
    """Training-only branch on ground-truth NOCS points.  [ref :185-200]"""

    def __init__(self, freeze=False):
        super().__init__()
        self.freeze = freeze
        self.extractor = PointNet2MSG(radii_list=[list(r) for r in WORLD_RADII])
        if not freeze:
            self.pose_estimator = HeavyEstimator()

    def forward(self, pts, pts_w_gt, rgb_local, pts_local, pts_w_local_gt=None):
        """``pts_w_local_gt``: the extractor's output when the caller has already run it (IST_Net.forward does, beside the
        RGB branch: it depends on the ground-truth coordinates only)."""
        if pts_w_local_gt is None:
            pts_w_local_gt = self.extractor(pts_w_gt)
        if self.freeze:
            return None, None, None, pts_w_local_gt
        r, t, s = self.pose_estimator(pts, pts_w_gt, rgb_local.detach(), pts_local.detach(), pts_w_local_gt)
        return r, t, s, pts_w_local_gt


USE_RGB_STREAM = True
USE_GATHER_FIRST = True
_RGB_STREAMS = {}


def _rgb_stream(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _RGB_STREAMS:
        _RGB_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _RGB_STREAMS[key]


class point_branch_side_streams:
    """The point branch's own side streams -- the concurrent MSG scales and the deferred weight-gradient stream of
    pointnet2.fused_mlp -- on or off.  They are worth 15 % for the encoder alone, where nothing else shares the chip;
    inside the full model the RGB branch's convolutions fill it, and the extra streams only let the encoders' backward be
    queued behind the RGB backward instead of beside it (HIP-graph replay of the config-3 step: 33.8 ms with them, 32.3
    without; tools/istnet_step_timeline.py shows the stall).

    ``point_branch_side_streams(False)`` takes effect at once (bench.py --workload istnet and
    examples/train_synthetic.py call it before training the full model); the returned object restores the previous
    process-wide values with ``.restore()`` or as a context manager (``with point_branch_side_streams(False): ...``), so
    a test that flips the switches does not leak into the next one."""

    def __init__(self, enabled):
        from .pointnet2 import fused_mlp
        self._saved = (fused_mlp.USE_SCALE_STREAMS, fused_mlp.USE_DEFERRED_WGRAD, fused_mlp.USE_DEFERRED_WGRAD_HEADS)
        fused_mlp.USE_SCALE_STREAMS = bool(enabled)
        fused_mlp.USE_DEFERRED_WGRAD = bool(enabled)
        if not enabled:
            # the IST / pose heads' per-point stacks keep deferring their weight-gradient GEMMs when the encoders' nodes
            # do not: on the chain they delay the gradient the RGB backward waits for (32.4 -> 32.1 ms)
            fused_mlp.USE_DEFERRED_WGRAD_HEADS = True

    def restore(self):
        from .pointnet2 import fused_mlp
        fused_mlp.USE_SCALE_STREAMS, fused_mlp.USE_DEFERRED_WGRAD, fused_mlp.USE_DEFERRED_WGRAD_HEADS = self._saved

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.restore()
        return False


def _infer_switch_state():
    """Everything process-global that a captured inference pass bakes in (graphed.InferenceGraph keys on it)."""
    from . import rgb_branch
    return (enc_modules._switch_state(), USE_RGB_STREAM, USE_GATHER_FIRST, heads_native.USE_NATIVE_TAIL, rgb_branch.USE_FUSED,
            rgb_branch.USE_NATIVE_TRUNK_CONV, rgb_branch._SPLIT_ON, torch.cuda.tunable.is_enabled())


class IST_Net(nn.Module):
    """IST-Net wiring.  [ref :10-76]  ``rgb_extractor`` maps (B,3,H,W) -> (B,128,H,W)."""

    def __init__(self, nclass=6, freeze_world_enhancer=False, rgb_extractor=None):
        super().__init__()
        self.nclass = nclass
        self.freeze_world_enhancer = freeze_world_enhancer
        if rgb_extractor is not None:
            self.rgb_cam_extractor = rgb_extractor
        self.pts_cam_extractor = PointNet2MSG(radii_list=[list(r) for r in CAM_RADII])
        self.implicit_transform = ImplicitTransformation(nclass)
        self.main_estimator = HeavyEstimator()
        self.cam_enhancer = LightEstimator()
        self.world_enhancer = WorldSpaceEnhancer(freeze=freeze_world_enhancer)

    def _rgb_local(self, inputs, b):
        if "rgb_local" in inputs:
            return inputs["rgb_local"]
        if USE_GATHER_FIRST and getattr(self.rgb_cam_extractor, "gathers_choose", False):
            # last layer(s) on the chosen pixels only (eval: rgb_branch._tail_at; training: _FinalAtChosenFn)
            return self.rgb_cam_extractor(inputs["rgb"], inputs["choose"])
        feat = self.rgb_cam_extractor(inputs["rgb"])
        d = feat.size(1)
        if not feat.is_contiguous() and feat.is_contiguous(memory_format=torch.channels_last):
            # channels-last extractor (MIOpen's faster layout): pick whole d-vectors of the chosen pixels
            rows = feat.permute(0, 2, 3, 1).reshape(b, -1, d)                       # view (B, H*W, d)
            picked = torch.gather(rows, 1, inputs["choose"].unsqueeze(-1).expand(-1, -1, d))
            return picked.transpose(1, 2).contiguous()
        choose = inputs["choose"].unsqueeze(1).repeat(1, d, 1)
        return torch.gather(feat.reshape(b, d, -1), 2, choose).contiguous()   # :41-45

    def forward(self, inputs):
        # An eval-mode, no_grad caller (the reference's test loop: one image per step, its 1-8 instances the batch) is bound
        # by the host issuing ~600 launches; graphed.InferenceGraph replays the whole pass from one HIP graph per batch shape
        pts = inputs["pts"]
        if (not self.training and not torch.is_grad_enabled() and graphed.ENABLED and isinstance(pts, torch.Tensor)
                and pts.is_cuda and _native.TIMING is None and _native.MARKERS is None):
            keys = ("pts", "category_label", "rgb_local") if "rgb_local" in inputs else ("pts", "category_label", "rgb", "choose")
            if all(k in inputs for k in keys):
                return graphed.for_inference(self, IST_Net._plain_forward, _infer_switch_state)(inputs, keys)
        return self._plain_forward(inputs)

    def _plain_forward(self, inputs):
        end_points = {}
        pts = inputs["pts"]
        b = pts.size(0)
        # The RGB branch (dense 2-D convolutions) and the point encoder are independent until the heads: run the
        # RGB branch on its own stream so the encoder's many short kernels fill in around the convolutions
        # (autograd replays each op's backward on the stream its forward ran on, so backward overlaps too).
        side = None
        if "rgb_local" not in inputs and pts.is_cuda and USE_RGB_STREAM:
            main = torch.cuda.current_stream(pts.device)
            side = _rgb_stream(pts.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                rgb_local = self._rgb_local(inputs, b)
        else:
            rgb_local = self._rgb_local(inputs, b)

        # (after the fork: the five small launches below would otherwise delay the start of the RGB branch, which is the longest
        # chain of the step)
        cls = inputs["category_label"].reshape(-1)
        c = torch.mean(pts, 1, keepdim=True)
        pts = pts - c
        index = cls + torch.arange(b, dtype=torch.long, device=pts.device) * self.nclass
        pts_local = self.pts_cam_extractor(pts)
        pts_w_local_gt = None
        if self.training:
            # the world-space encoder of the auxiliary branch reads the ground-truth coordinates only (reference ist_net.py:50-51
            # calls it last): issued here, its sampling chain and its many short kernels run beside the RGB branch instead of
            # heading the serial part of the step; backward follows the same order in reverse.  (On a stream of its own it
            # measured 34.3 vs 32.2 ms and needs care under capture -- tools/exp/world_stream_bisect.py)
            pts_w_local_gt = self.world_enhancer.extractor(inputs["qo"])
        if side is not None:
            main.wait_stream(side)
            rgb_local.record_stream(main)
        if self.training:
            r_cam, t_cam, s_cam = self.cam_enhancer(pts, rgb_local, pts_local)
        pts_w, pts_w_local = self.implicit_transform(rgb_local, pts_local, pts, c, index)
        r, t, s = self.main_estimator(pts, pts_w, rgb_local, pts_local, pts_w_local)
        end_points["pred_qo"] = pts_w
        end_points["pred_rotation"] = r
        end_points["pred_translation"] = t + c.squeeze(1)
        end_points["pred_size"] = s
        if self.training:
            r_w, t_w, s_w, pts_w_local_gt = self.world_enhancer(pts, inputs["qo"], rgb_local, pts_local, pts_w_local_gt)
            end_points["pts_w_local"] = pts_w_local
            end_points["pts_w_local_gt"] = pts_w_local_gt
            end_points["pred_rotation_aux_cam"] = r_cam
            end_points["pred_translation_aux_cam"] = t_cam + c.squeeze(1)
            end_points["pred_size_aux_cam"] = s_cam
            if not self.freeze_world_enhancer:
                end_points["pred_rotation_aux_world"] = r_w
                end_points["pred_translation_aux_world"] = t_w + c.squeeze(1)
                end_points["pred_size_aux_world"] = s_w
        return end_points
