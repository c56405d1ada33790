"""``vsrd.operations.box_3d_iou`` (reference: vsrd/operations/kitti360_operations.py:83-120; caller scripts/main.py:888-905).

Evaluation-time metric on a handful of matched box pairs per logging step: host side, like the reference (which runs it in NumPy
through ``utils.torch_function``).  Same numbers as the reference *including its quirks* (see oracle/geometry.py::box_3d_iou):
the clip's intersection points divide by ``determinant + 0.01``, the overlap area is capped by the smaller footprint, and clockwise
footprints -- what main.py's corner order produces -- are not reordered.  Written array-style: each clip edge classifies all
vertices at once and splices the crossings in with a stable merge.
"""
import numpy as np
import torch


def _to_numpy(value):
    return value.detach().cpu().numpy() if isinstance(value, torch.Tensor) else np.asarray(value)


def _clip_by_edge(polygon, start, end):
    """Keep the part of ``polygon`` [P,2] on the inner side of the directed edge start -> end (strict test, as the reference)."""
    direction = end - start
    side = direction[0] * (polygon[:, 1] - start[1]) > direction[1] * (polygon[:, 0] - start[0])        # vertex is kept
    previous, previous_side = np.roll(polygon, 1, axis=0), np.roll(side, 1)
    crosses = side != previous_side
    # intersection of segment previous -> vertex with the edge line; the + 0.01 is the reference's (kitti360_operations.py:30)
    dc, dp = start - end, previous - polygon
    n1 = start[0] * end[1] - start[1] * end[0]
    n2 = previous[:, 0] * polygon[:, 1] - previous[:, 1] * polygon[:, 0]
    n3 = 1.0 / (dc[0] * dp[:, 1] - dc[1] * dp[:, 0] + 0.01)
    crossing = np.stack([(n1 * dp[:, 0] - n2 * dc[0]) * n3, (n1 * dp[:, 1] - n2 * dc[1]) * n3], axis=1)
    # output order per vertex: its crossing (if the segment crosses) comes before the vertex itself (if kept)
    pieces = np.stack([crossing, polygon], axis=1).reshape(-1, 2)
    keep = np.stack([crosses, side], axis=1).reshape(-1)
    return pieces[keep]


def _cross(a, b):
    return a[0] * b[1] - a[1] * b[0]


def _convex_area(points):
    """Area of the convex hull of ``points`` [P,2] (the reference takes scipy's ConvexHull(...).volume of the clipped polygon)."""
    points = np.unique(points, axis=0)
    if len(points) < 3:
        return 0.0
    points = points[np.lexsort((points[:, 1], points[:, 0]))]

    def chain(sequence):
        hull = []
        for p in sequence:
            while len(hull) >= 2 and _cross(hull[-1] - hull[-2], p - hull[-2]) <= 0:
                hull.pop()
            hull.append(p)
        return hull[:-1]

    hull = np.array(chain(points) + chain(points[::-1]))
    return 0.5 * abs(np.dot(hull[:, 0], np.roll(hull[:, 1], 1)) - np.dot(hull[:, 1], np.roll(hull[:, 0], 1)))


def box_3d_iou(corners1, corners2):
    """corners [8,3] with +Z up and corners 0-3 the upper face -> (3-D IoU, bird's-eye-view IoU), as NumPy float64 scalars."""
    c1, c2 = _to_numpy(corners1).astype(np.float64), _to_numpy(corners2).astype(np.float64)
    foot1, foot2 = c1[[3, 2, 1, 0], :2], c2[[3, 2, 1, 0], :2]

    def shoelace(poly):
        return 0.5 * abs(np.dot(poly[:, 0], np.roll(poly[:, 1], 1)) - np.dot(poly[:, 1], np.roll(poly[:, 0], 1)))

    area1, area2 = shoelace(foot1), shoelace(foot2)
    overlap = foot1
    for k in range(4):
        overlap = _clip_by_edge(overlap, foot2[k - 1], foot2[k])
        if len(overlap) == 0:
            break
    inter_area = min(area1, area2, _convex_area(overlap)) if len(overlap) else 0.0
    iou_bev = inter_area / (area1 + area2 - inter_area)
    height = max(0.0, min(c1[0, 2], c2[0, 2]) - max(c1[4, 2], c2[4, 2]))

    def volume(c):
        return np.linalg.norm(c[0] - c[1]) * np.linalg.norm(c[1] - c[2]) * np.linalg.norm(c[0] - c[4])

    inter_volume = inter_area * height
    return np.float64(inter_volume / (volume(c1) + volume(c2) - inter_volume)), np.float64(iou_bev)
