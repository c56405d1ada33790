"""On-disk formats on either side of the hot path: the per-frame checkpoint the loop writes and the
pseudo-label files derived from it.  Host-side I/O only.

  checkpoint dict           scripts/main.py:1109-1121 (`Saver.save` = torch.save of this dict, vsrd/utils.py:191-198);
                            consumer tools/kitti_360/make_predictions.py:61-66 loads models.detector only
  prediction JSON           tools/kitti_360/make_predictions.py:158-169   {boxes_3d, boxes_2d, confidences} per class
  multi-view confidences    tools/kitti_360/make_predictions.py:86-176    mean box IoU over source views + Hungarian (maximise)
  KITTI label line          tools/kitti_360/convert_predictions.py:16-78
"""
import json
import math
import os

import torch


def to_host(value):
    """A checkpoint payload with every tensor copied to host memory (containers rebuilt, everything else as it is)."""
    if isinstance(value, torch.Tensor):
        return value.detach().cpu()
    if isinstance(value, dict):
        return type(value)((k, to_host(v)) for k, v in value.items())
    if isinstance(value, (list, tuple)):
        return type(value)(to_host(v) for v in value)
    return value


def checkpoint_payload(frame_optimizer, step, metrics=None, host=False):
    """``host=True``: the tensors already in host memory -- what a frame optimised next to others must hand to ``torch.save``: serialising
    device tensors copies them with calls that HIP refuses while another frame's thread captures a graph (launcher.main takes the copies
    under optimization.exclusive_device_access())."""
    models = {"detector": frame_optimizer.detector.state_dict(),
              "hyper_distance_field": frame_optimizer.hyper_distance_field.state_dict()}
    # eager and hipGraph mode write the same layout (FrameOptimizer.optimizer_state_dict / scheduler_state_dict)
    payload = dict(step=step, models=models, optimizer=frame_optimizer.optimizer_state_dict(),
                   scheduler=frame_optimizer.scheduler_state_dict(), metrics=metrics or {})
    return to_host(payload) if host else payload


def atomic_torch_save(payload, path):
    """torch.save to ``path`` through a temporary file and os.replace: a process killed mid-write leaves no truncated file that
    the "skip if the final checkpoint exists" restart guard (main.py:134-136) would take for a finished frame."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    temporary = f"{path}.tmp.{os.getpid()}"
    try:
        torch.save(payload, temporary)
        os.replace(temporary, path)
    finally:
        if os.path.exists(temporary):
            os.remove(temporary)


def save_checkpoint(path, frame_optimizer, step, metrics=None):
    atomic_torch_save(checkpoint_payload(frame_optimizer, step, metrics), path)


def prediction_record(boxes_3d, boxes_2d, confidences, class_name="car"):
    """make_predictions.py:158-163: camera-frame corners [N,8,3], 2-D boxes [N,2,2], confidences [N]."""
    return dict(boxes_3d={class_name: boxes_3d.tolist()}, boxes_2d={class_name: boxes_2d.tolist()},
                confidences={class_name: confidences.tolist()})


def save_prediction(path, boxes_3d, boxes_2d, confidences, class_name="car"):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as file:
        json.dump(prediction_record(boxes_3d, boxes_2d, confidences, class_name), file, indent=4, sort_keys=False)


def box_iou(boxes1, boxes2):
    """torchvision.ops.box_iou on [N,4] x [M,4]."""
    area1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    area2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    lt = torch.max(boxes1[:, None, :2], boxes2[None, :, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[None, :, 2:])
    inter = (rb - lt).clamp(min=0).prod(-1)
    return inter / (area1[:, None] + area2[None, :] - inter)


def multi_view_confidences(pd_boxes_2d, gt_boxes_2d, gt_to_target):
    """make_predictions.py:86-176.  Per source view v: pd_boxes_2d[v] [N,2,2] (already clipped), gt_boxes_2d[v] [M_v,2,2],
    gt_to_target[v] [M_v] = index of each source instance among the target's instances (-1 = absent).
    Returns (confidences [K], matched_pd [K], matched_gt [K])."""
    from scipy.optimize import linear_sum_assignment
    num_pd = pd_boxes_2d[0].shape[0]
    num_gt = 1 + max(int(t.max()) for t in gt_to_target)
    iou_sum, count = torch.zeros(num_pd, num_gt), torch.zeros(num_pd, num_gt)
    for pd, gt, index in zip(pd_boxes_2d, gt_boxes_2d, gt_to_target):
        iou = torch.nan_to_num(box_iou(pd.flatten(-2, -1).cpu(), gt.flatten(-2, -1).cpu()))
        keep = index >= 0
        iou_sum[:, index[keep]] += iou[:, keep]
        count[:, index[keep]] += 1
    mean = iou_sum / count
    pd_idx, gt_idx = linear_sum_assignment(mean.numpy(), maximize=True)
    return mean[pd_idx, gt_idx], torch.as_tensor(pd_idx), torch.as_tensor(gt_idx)


def kitti_label_line(class_name, box_3d, box_2d, score):
    """convert_predictions.py:16-78: one KITTI-3D label line from camera-frame corners [8,3], a 2-D box [2,2] and a score."""
    location = box_3d.mean(-2)

    def mean_edge(a, b):
        return (box_3d[a, :] - box_3d[b, :]).norm(dim=-1).mean(-1)
    width, height, length = mean_edge([1, 2, 6, 5], [0, 3, 7, 4]), mean_edge([4, 5, 6, 7], [0, 1, 2, 3]), mean_edge([1, 0, 4, 5], [2, 3, 7, 6])
    forward = (box_3d[[1, 0, 4, 5], :] - box_3d[[2, 3, 7, 6], :]).mean(-2)
    heading = torch.nn.functional.normalize(forward[[2, 0]], dim=-1)
    orientation = torch.atan2(heading[1], heading[0])
    location = location.clone()
    location[1] += height / 2.0
    dimension = torch.stack([height, width, length])
    ray_orientation = torch.atan2(location[0], location[2])
    global_orientation = orientation - math.pi / 2.0
    local_orientation = global_orientation - ray_orientation
    return (f"{class_name.capitalize()} {0.0} {0} {local_orientation} "
            f"{' '.join(map(str, box_2d.flatten().tolist()))} {' '.join(map(str, dimension.tolist()))} "
            f"{' '.join(map(str, location.tolist()))} {global_orientation} {score}\n")


def save_kitti_labels(path, class_names, boxes_3d, boxes_2d, scores):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as file:
        for name, box_3d, box_2d, score in zip(class_names, boxes_3d, boxes_2d, scores):
            file.write(kitti_label_line(name, box_3d, box_2d, score))
