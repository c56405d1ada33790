"""The losses scripts/main.py assembles inline on the hot path, as device-side functions (torch element-wise ops on
small tensors; the heavy lifting -- rendering -- is in the HIP library).

  matching + projection losses    scripts/main.py:374-415   (torchvision.ops.distance_box_iou / _loss restated: the
                                                              reference pins torchvision==0.14.0, environment.yaml:368)
  schedules                       scripts/main.py:420-431
  silhouette / eikonal            scripts/main.py:653-687
  weights                         configs/.../config.json:120-127
"""
import math

import torch
import torch.nn.functional as F

LOSS_WEIGHTS = dict(silhouette_loss=1.0, l1_projection_loss=1.0, iou_projection_loss=0.1, eikonal_loss=0.01, photometric_loss=0.0)


def cosine_annealing(x, start, end):
    return (math.cos(math.pi * x) + 1.0) / 2.0 * (start - end) + end


def schedules(step, num_steps, max_temperature=1.0, min_temperature=0.1, max_std=1.0, min_std=0.1):
    """-> (cosine_ratio, sdf_union_temperature, sdf_std_deviation), main.py:420-431."""
    x = step / num_steps
    return x, cosine_annealing(x, max_temperature, min_temperature), cosine_annealing(x, max_std, min_std)


def _split(boxes):
    boxes = boxes.flatten(-2, -1) if boxes.shape[-1] == 2 else boxes
    return boxes.unbind(-1)


def distance_box_iou(boxes1, boxes2, eps=1e-7):
    """Pairwise DIoU [N,*] x [M,*] -> [N,M] (boxes as [.,4] or [.,2,2])."""
    x1, y1, x2, y2 = (c[:, None] for c in _split(boxes1))
    x1g, y1g, x2g, y2g = (c[None, :] for c in _split(boxes2))
    inter = (torch.min(x2, x2g) - torch.max(x1, x1g)).clamp_min(0) * (torch.min(y2, y2g) - torch.max(y1, y1g)).clamp_min(0)
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    diagonal = (torch.max(x2, x2g) - torch.min(x1, x1g)).clamp_min(0) ** 2 + (torch.max(y2, y2g) - torch.min(y1, y1g)).clamp_min(0) ** 2 + eps
    centres = ((x1 + x2) - (x1g + x2g)) ** 2 / 4 + ((y1 + y2) - (y1g + y2g)) ** 2 / 4
    return inter / union - centres / diagonal


def distance_box_iou_loss(boxes1, boxes2, eps=1e-7):
    """Element-wise DIoU loss (reduction='none') on matched boxes [...,4] / [...,2,2]."""
    x1, y1, x2, y2 = _split(boxes1)
    x1g, y1g, x2g, y2g = _split(boxes2)
    ix1, iy1, ix2, iy2 = torch.max(x1, x1g), torch.max(y1, y1g), torch.min(x2, x2g), torch.min(y2, y2g)
    inter = torch.where((iy2 > iy1) & (ix2 > ix1), (ix2 - ix1) * (iy2 - iy1), torch.zeros_like(x1))
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    diagonal = (torch.max(x2, x2g) - torch.min(x1, x1g)) ** 2 + (torch.max(y2, y2g) - torch.min(y1, y1g)) ** 2 + eps
    centres = ((x1 + x2) / 2 - (x1g + x2g) / 2) ** 2 + ((y1 + y2) / 2 - (y1g + y2g) / 2) ** 2
    return 1 - inter / (union + eps) + centres / diagonal


def linear_sum_assignment(cost):
    """scipy.optimize.linear_sum_assignment(cost) for a device matrix [P,G] (P, G <= 64) without leaving the device
    (vsrd_linear_sum_assignment: the same shortest-augmenting-path algorithm and tie rule, float64 duals)."""
    from . import _lib
    lib = _lib.load()
    cost = cost.detach().to(torch.float32).contiguous()
    P, G = cost.shape
    rows = torch.empty(min(P, G), dtype=torch.int64, device=cost.device)
    cols = torch.empty_like(rows)
    _lib.check(lib.vsrd_linear_sum_assignment(_lib.ptr(cost), P, G, rows.data_ptr(), cols.data_ptr(), _lib.stream()))
    return rows, cols


def match_instances(pd_boxes_2d, gt_boxes_2d):
    """main.py:374-386: Hungarian assignment on -DIoU of the target view.  The reference copies the cost matrix to the host for
    scipy (one synchronisation per step); here both the cost and the assignment stay on the device (vsrd_match_boxes), so the
    optimisation step has no host round trip and can be captured in a hipGraph."""
    from . import _lib
    lib = _lib.load()
    pd = pd_boxes_2d.detach().reshape(-1, 4).to(torch.float32).contiguous()
    gt = gt_boxes_2d.detach().reshape(-1, 4).to(torch.float32).contiguous()
    P, G = pd.shape[0], gt.shape[0]
    pd_idx = torch.empty(min(P, G), dtype=torch.int64, device=pd.device)
    gt_idx = torch.empty_like(pd_idx)
    _lib.check(lib.vsrd_match_boxes(_lib.ptr(pd), _lib.ptr(gt), P, G, pd_idx.data_ptr(), gt_idx.data_ptr(), _lib.stream()))
    return pd_idx, gt_idx


def projection_losses(pd_boxes_2d, gt_boxes_2d, visible_masks, pd_idx, gt_idx):
    """main.py:391-415 for all views at once: pd/gt [V,N,2,2], visible_masks [V,N] (by gt instance) -> (iou_loss, l1_loss).

    The reference concatenates the kept rows of every view and takes the mean; a masked mean over [V,M] is the same number.
    """
    pd = pd_boxes_2d[:, pd_idx].flatten(-2, -1)                      # [V,M,4]
    gt = gt_boxes_2d[:, gt_idx].flatten(-2, -1)
    keep = visible_masks[:, gt_idx].to(pd.dtype)                     # [V,M]
    count = keep.sum().clamp_min(1.0)
    iou = (distance_box_iou_loss(pd, gt) * keep).sum() / count
    l1 = (F.smooth_l1_loss(pd, gt, reduction="none") * keep.unsqueeze(-1)).sum() / (count * 4.0)
    return iou, l1


def silhouette_loss(labels, targets, pd_idx=None, gt_idx=None):
    """main.py:653-671."""
    if pd_idx is not None:
        labels, targets = labels[..., pd_idx], targets[..., gt_idx]
    return F.binary_cross_entropy(labels.clamp(1.0e-6, 1.0 - 1.0e-6), targets, reduction="none").mean()


def eikonal_loss(gradients):
    """main.py:679-687."""
    return ((gradients.norm(dim=-1) - 1.0) ** 2).mean()
