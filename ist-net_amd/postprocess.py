"""Inference post-processing of IST-Net: the tail of ``test_func`` and the pose-error table.

``assemble_pred_RTs``  [ref utils/solver.py:231-241]: network outputs -> the 4x4 similarity transforms and unit
sizes written to the result pickles.
``pose_errors``        [ref utils/evaluation_utils.py:588-688, compute_RT_degree_cm_symmetry / compute_RT_overlaps]:
rotation error in degrees and translation error in centimetres of every (prediction, ground truth) pair, with
the reference's symmetry rules, as ONE batched tensor expression on whatever device the poses live on (the
reference is a Python double loop over numpy 4x4s per image).  float64 throughout, as numpy is there.
"""
import math

import torch

# class-name table of the NOCS REAL275 / CAMERA25 evaluation  [ref evaluation_utils.py:735-760 callers]
SYNSET_NAMES = ("BG", "bottle", "bowl", "camera", "can", "laptop", "mug")
_AXIS_SYMMETRIC = ("bottle", "can", "bowl")     # any rotation about y is equivalent   [ref :633]
_HALF_TURN = ("phone", "eggbox", "glue")        # 180 degrees about y is equivalent    [ref :647]


def assemble_pred_RTs(pred_rotation, pred_translation, pred_size):
    """(B,3,3), (B,3), (B,3) -> pred_RTs (B,4,4) with R*|s| and t, pred_scales (B,3) = s/|s|.  [ref solver.py:231-241]"""
    scale = torch.norm(pred_size, dim=1, keepdim=True)
    rts = torch.eye(4, dtype=torch.float32, device=pred_rotation.device).unsqueeze(0).repeat(pred_rotation.size(0), 1, 1)
    rts[:, :3, 3] = pred_translation
    rts[:, :3, :3] = pred_rotation * scale.unsqueeze(2)
    return rts, pred_size / scale


def pose_errors(pred_RTs, gt_RTs, gt_class_ids, gt_handle_visibility, synset_names=SYNSET_NAMES):
    """pred_RTs (P,4,4), gt_RTs (G,4,4), gt_class_ids (G,), gt_handle_visibility (G,) -> (P,G,2) float64:
    [..., 0] rotation error in degrees, [..., 1] translation error in cm.  [ref evaluation_utils.py:588-688]
    Runs on the device of ``pred_RTs``; the tables are tiny (instances per image), so host tensors are the faster
    choice (64x64: 0.5 ms on the host, 15 ms of launch / sync latency on the GPU, tools/bench_infer_full.py)."""
    pred = torch.as_tensor(pred_RTs).to(torch.float64)
    gt = torch.as_tensor(gt_RTs).to(torch.float64, copy=False).to(pred.device)
    cls = torch.as_tensor(gt_class_ids).to(pred.device).long()
    vis = torch.as_tensor(gt_handle_visibility).to(pred.device)
    if pred.numel() == 0 or gt.numel() == 0:
        return torch.zeros(pred.size(0), gt.size(0), 2, dtype=torch.float64, device=pred.device)
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=torch.float64, device=pred.device)
    if not (torch.equal(pred[:, 3, :], bottom.expand(pred.size(0), 4)) and torch.equal(gt[:, 3, :], bottom.expand(gt.size(0), 4))):
        raise ValueError("pose_errors: last row of every transform must be [0, 0, 0, 1]")   # ref :620-625 exits

    def unit_rotation(rt):      # R / cbrt(det R): strips the isotropic scale  [ref :627,630]
        m = rt[:, :3, :3]        # closed-form 3x3 determinant: torch.linalg.det is a batched LU with a host sync on the GPU
        det = (m[:, 0] * torch.linalg.cross(m[:, 1], m[:, 2], dim=1)).sum(dim=1)
        return rt[:, :3, :3] / (torch.sign(det) * det.abs().pow(1.0 / 3.0)).view(-1, 1, 1)

    r1, r2 = unit_rotation(pred), unit_rotation(gt)                       # (P,3,3), (G,3,3)
    names = list(synset_names)
    member = lambda group: torch.tensor([n in group for n in names], device=pred.device)[cls]
    axis_sym = member(_AXIS_SYMMETRIC) | ((cls == names.index("mug")) & (vis == 0) if "mug" in names else False)
    half_turn = member(_HALF_TURN) & ~axis_sym
    # axis-symmetric classes: angle between the images of the y axis  [ref :634-646]
    y1, y2 = r1[:, :, 1], r2[:, :, 1]
    cos_y = (y1 @ y2.t()) / (y1.norm(dim=1).view(-1, 1) * y2.norm(dim=1).view(1, -1))
    theta_axis = torch.arccos(cos_y)
    # general case: angle of R1 R2^T, clipped  [ref :653-655]
    rel = torch.einsum("pij,gkj->pgik", r1, r2)
    trace = rel.diagonal(dim1=-2, dim2=-1).sum(-1)
    theta_full = torch.arccos(torch.clamp((trace - 1) / 2, -1.0, 1.0))
    theta = torch.where(axis_sym.view(1, -1), theta_axis, theta_full)
    if bool(half_turn.any()):    # min over the half-turn-equivalent pose, unclipped as in the reference  [ref :648-652]
        flip = torch.diag(torch.tensor([-1.0, 1.0, -1.0], dtype=torch.float64, device=pred.device))
        trace_rot = torch.einsum("pij,jk,glk->pgil", r1, flip, r2).diagonal(dim1=-2, dim2=-1).sum(-1)
        both = torch.minimum(torch.arccos((trace - 1) / 2), torch.arccos((trace_rot - 1) / 2))
        theta = torch.where(half_turn.view(1, -1), both, theta)
    shift = (pred[:, None, :3, 3] - gt[None, :, :3, 3]).norm(dim=-1) * 100
    return torch.stack([theta * (180 / math.pi), shift], dim=-1)
