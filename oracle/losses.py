"""Oracle: the losses scripts/main.py assembles inline on the hot path (TEST INFRASTRUCTURE).

Restates
  * ``scripts/main.py:374-386``  bipartite matching (negative DIoU cost, Hungarian on host)
  * ``scripts/main.py:391-415``  DIoU + smooth-L1 projection losses over matched, visible instances
  * ``scripts/main.py:420-431``  cosine-annealed schedules
  * ``scripts/main.py:653-671``  silhouette binary cross-entropy
  * ``scripts/main.py:679-687``  eikonal loss
  * ``scripts/main.py:855`` + ``configs/.../config.json:120-127``  weighted sum
"""
import math

import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

from . import geometry

LOSS_WEIGHTS = dict(silhouette_loss=1.0, l1_projection_loss=1.0, iou_projection_loss=0.1, eikonal_loss=0.01)  # config.json:120-127


def cosine_annealing(x, start, end):
    """main.py:420: (cos(pi x) + 1)/2 * (a - b) + b."""
    return (math.cos(math.pi * x) + 1.0) / 2.0 * (start - end) + end


def schedules(step, num_steps=3000, max_value=1.0, min_value=0.1):
    """main.py:421-431 -> (cosine_ratio, sdf_union_temperature, sdf_std_deviation)."""
    ratio = step / num_steps
    value = cosine_annealing(ratio, max_value, min_value)
    return ratio, value, value


def silhouette_loss(labels, targets, pd_indices=None, gt_indices=None):
    """main.py:653-671: mean BCE(clamp(labels[..., pd], 1e-6, 1-1e-6), targets[..., gt])."""
    if pd_indices is not None:
        labels, targets = labels[..., pd_indices], targets[..., gt_indices]
    return F.binary_cross_entropy(labels.clamp(1.0e-6, 1.0 - 1.0e-6), targets, reduction="none").mean()


def eikonal_loss(gradients):
    """main.py:679-687: mse(||grad||_2, 1)."""
    return ((gradients.norm(dim=-1) - 1.0) ** 2).mean()


def match_instances(pd_boxes_2d, gt_boxes_2d):
    """main.py:374-386: Hungarian on -DIoU of the target view.  [N,2,2] x [M,2,2] -> (pd_idx, gt_idx)."""
    cost = -geometry.distance_box_iou(pd_boxes_2d.flatten(-2, -1), gt_boxes_2d.flatten(-2, -1))
    pd_idx, gt_idx = linear_sum_assignment(cost.detach().cpu().numpy())
    return torch.as_tensor(pd_idx), torch.as_tensor(gt_idx)


def projection_losses(pd_boxes_2d, gt_boxes_2d, visible_masks, pd_idx, gt_idx):
    """main.py:391-415.  pd/gt [V,N,2,2], visible_masks [V,N] (bool, indexed by gt instance)."""
    iou_terms, l1_terms = [], []
    for v in range(pd_boxes_2d.shape[0]):
        keep = visible_masks[v][gt_idx]
        pd = pd_boxes_2d[v][pd_idx[keep]].flatten(-2, -1)
        gt = gt_boxes_2d[v][gt_idx[keep]].flatten(-2, -1)
        iou_terms.append(geometry.distance_box_iou_loss(pd, gt))
        l1_terms.append(F.smooth_l1_loss(pd, gt, reduction="none").flatten())
    return torch.cat(iou_terms).mean(), torch.cat(l1_terms).mean()
