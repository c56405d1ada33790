"""Oracle: cameras, box parameterisation, multi-view box projection (TEST INFRASTRUCTURE).

Restates
  * ``vsrd/rendering/utils.py:5-18``                       ray_casting
  * ``vsrd/models/detectors/box_parameters.py:5-146``      BoxParameters3D decode / encode
  * ``vsrd/operations/geometric_operations.py:343-389``    clip_lines_to_front, project_box_3d
  * ``scripts/main.py:339-367``                            world -> camera -> 2-D boxes per view
  * torchvision==0.14.0 ``ops.distance_box_iou`` / ``distance_box_iou_loss`` /
    ``clip_boxes_to_image`` (called at ``scripts/main.py:359,375,393``).  torchvision is NOT
    in the build image: these three follow the published DIoU definition
    (IoU - rho^2(centres)/c^2(enclosing diagonal), eps 1e-7) -- PARITY UNPINNED (no torchvision
    output exists to pin them): checked against hand-computed cases, and (round 6) the IoU term and
    the enclosing box against ``transformers``' DETR ``box_iou`` / ``generalized_box_iou`` -- taken
    from torchvision.ops.boxes -- in tests/test_oracle_golden.py.
"""
import torch
import torch.nn.functional as F

# 12 box edges over the 8 corners, scripts/main.py:26-30
BOX_EDGES = ((0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7))
# unit corners, box_parameters.py:78-87
UNIT_CORNERS = ((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1), (-1, 1, 1), (1, 1, 1), (1, 1, -1), (-1, 1, -1))
LOCATION_RANGE = ((-50.0, 1.55 - 1.75 / 2.0 - 5.0, 0.0), (50.0, 1.55 - 1.75 / 2.0 + 5.0, 100.0))  # box_parameters.py:23-26
DIMENSION_RANGE = ((0.75, 0.75, 1.5), (1.0, 1.0, 2.5))                                            # box_parameters.py:27-30


def ray_casting(image_size, intrinsics, extrinsics):
    """utils.py:5-18.  Integer pixel centres (x, y, 1); returns (camera [...,3], dirs [...,H,W,3])."""
    height, width = image_size
    ys, xs = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
    pixels = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1).to(intrinsics)
    inv_e = torch.linalg.inv(extrinsics)
    back = inv_e[..., :3, :3] @ torch.linalg.inv(intrinsics)
    dirs = torch.einsum("...mn,hwn->...hwm", back, pixels)
    return inv_e[..., :3, 3], F.normalize(dirs, dim=-1)


def rotation_matrix_y(cos, sin):
    zero, one = torch.zeros_like(cos), torch.ones_like(cos)
    return torch.stack([
        torch.stack([cos, zero, sin], -1), torch.stack([zero, one, zero], -1), torch.stack([-sin, zero, cos], -1)], -2)


def rotation_matrix_x(angles):
    """geometric_operations.py:30-40."""
    cos, sin = torch.cos(angles), torch.sin(angles)
    zero, one = torch.zeros_like(cos), torch.ones_like(cos)
    return torch.stack([
        torch.stack([one, zero, zero], -1), torch.stack([zero, cos, -sin], -1), torch.stack([zero, sin, cos], -1)], -2)


def expand_to_4x4(matrices):
    """geometric_operations.py:10-15."""
    out = torch.eye(4).to(matrices).repeat(*matrices.shape[:-2], 1, 1)
    out[..., :matrices.shape[-2], :matrices.shape[-1]] = matrices
    return out


def decode_box_parameters(raw_locations, raw_dimensions, raw_orientations):
    """box_parameters.py:60-90,124-146: raw parameters -> (locations, half extents, R_y, corners [...,8,3])."""
    lo, hi = (torch.tensor(r).to(raw_locations) for r in LOCATION_RANGE)
    locations = torch.lerp(lo, hi, torch.sigmoid(raw_locations))
    lo, hi = (torch.tensor(r).to(raw_dimensions) for r in DIMENSION_RANGE)
    dimensions = torch.lerp(lo, hi, torch.sigmoid(raw_dimensions))
    heading = F.normalize(raw_orientations, dim=-1)
    orientations = rotation_matrix_y(heading[..., 0], heading[..., 1])
    return locations, dimensions, orientations, box_corners(locations, dimensions, orientations)


def box_corners(locations, dimensions, orientations):
    corners = torch.tensor(UNIT_CORNERS).to(dimensions) * dimensions.unsqueeze(-2)
    return corners @ orientations.transpose(-2, -1) + locations.unsqueeze(-2)


def encode_box_corners(corners):
    """box_parameters.py:92-122: corners [...,8,3] -> (locations, half extents, R_y)."""
    def edge_mean_norm(a, b):
        return (corners[..., a, :] - corners[..., b, :]).norm(dim=-1).mean(-1)
    locations = corners.mean(-2)
    widths = edge_mean_norm([1, 2, 6, 5], [0, 3, 7, 4])
    heights = edge_mean_norm([4, 5, 6, 7], [0, 1, 2, 3])
    lengths = edge_mean_norm([1, 0, 4, 5], [2, 3, 7, 6])
    forward = (corners[..., [1, 0, 4, 5], :] - corners[..., [2, 3, 7, 6], :]).mean(-2)
    heading = F.normalize(forward[..., [2, 0]], dim=-1)
    return locations, torch.stack([widths, heights, lengths], -1) / 2.0, rotation_matrix_y(heading[..., 0], heading[..., 1])


def clip_edges_to_front(edges, epsilon=1.0e-6):
    """geometric_operations.py:343-365.  edges [...,2,3] -> (clipped [...,2,3], far-end-in-front mask [...])."""
    a, b = edges[..., 0, :], edges[..., 1, :]
    a_deeper = a[..., 2:] > b[..., 2:]
    far, near = torch.where(a_deeper, a, b), torch.where(a_deeper, b, a)
    t = (far[..., 2:] / (far[..., 2:] - near[..., 2:]).clamp_min(epsilon)).clamp_max(1.0)
    near = far + (near - far) * t
    return torch.stack([far, near], dim=-2), far[..., 2] > 0


def project_boxes(camera_corners, intrinsics, epsilon=1.0e-6):
    """geometric_operations.py:368-389, batched: corners [...,8,3] (camera frame), K [3,3] -> [...,2,2].

    Boxes whose every edge lies behind the camera give zeros (``:384-387``).
    """
    idx = torch.tensor(BOX_EDGES, device=camera_corners.device)
    edges, front = clip_edges_to_front(camera_corners[..., idx, :], epsilon)           # [...,12,2,3], [...,12]
    pix = edges @ intrinsics.transpose(-1, -2)
    pix = pix[..., :2] / pix[..., 2:].clamp_min(epsilon)                              # [...,12,2,2]
    keep = front[..., None, None]
    inf = torch.full_like(pix, float("inf"))
    lo = torch.where(keep, pix, inf).flatten(-3, -2).min(-2).values
    hi = torch.where(keep, pix, -inf).flatten(-3, -2).max(-2).values
    any_front = front.any(-1)[..., None, None]
    return torch.where(any_front, torch.stack([lo, hi], dim=-2), torch.zeros_like(torch.stack([lo, hi], dim=-2)))


def project_boxes_multi_view(world_corners, extrinsics, intrinsics, image_size):
    """main.py:339-362: world corners [N,8,3], E [V,4,4], K [V,3,3] -> clipped 2-D boxes [V,N,2,2]."""
    homog = F.pad(world_corners, (0, 1), value=1.0)
    cam = torch.einsum("vmn,ikn->vikm", extrinsics, homog)
    cam = cam[..., :3] / cam[..., 3:]
    boxes = torch.stack([project_boxes(cam[v], intrinsics[v]) for v in range(cam.shape[0])])
    return clip_boxes_to_image(boxes, image_size), cam


def clip_boxes_to_image(boxes, image_size):
    """torchvision.ops.clip_boxes_to_image on [...,2,2] (x in [0,W], y in [0,H])."""
    height, width = image_size
    x = boxes[..., 0].clamp(0, width)
    y = boxes[..., 1].clamp(0, height)
    return torch.stack([x, y], dim=-1)


def _corners(boxes):
    boxes = boxes.reshape(*boxes.shape[:-2], 4) if boxes.shape[-1] == 2 else boxes
    return boxes.unbind(-1)


def distance_box_iou(boxes1, boxes2, eps=1.0e-7):
    """torchvision.ops.distance_box_iou (pairwise [N,4] x [M,4] -> [N,M]).  PARITY UNPINNED."""
    x1, y1, x2, y2 = (c[:, None] for c in _corners(boxes1))
    x1g, y1g, x2g, y2g = (c[None, :] for c in _corners(boxes2))
    inter = (torch.min(x2, x2g) - torch.max(x1, x1g)).clamp_min(0) * (torch.min(y2, y2g) - torch.max(y1, y1g)).clamp_min(0)
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    iou = inter / union
    diag = (torch.max(x2, x2g) - torch.min(x1, x1g)).clamp_min(0) ** 2 + (torch.max(y2, y2g) - torch.min(y1, y1g)).clamp_min(0) ** 2 + eps
    centre = ((x1 + x2) / 2 - (x1g + x2g) / 2) ** 2 + ((y1 + y2) / 2 - (y1g + y2g) / 2) ** 2
    return iou - centre / diag


def distance_box_iou_loss(boxes1, boxes2, eps=1.0e-7):
    """torchvision.ops.distance_box_iou_loss(reduction='none') (element-wise [K,4]).  PARITY UNPINNED."""
    x1, y1, x2, y2 = _corners(boxes1)
    x1g, y1g, x2g, y2g = _corners(boxes2)
    ix1, iy1, ix2, iy2 = torch.max(x1, x1g), torch.max(y1, y1g), torch.min(x2, x2g), torch.min(y2, y2g)
    overlap = (iy2 > iy1) & (ix2 > ix1)
    inter = torch.where(overlap, (ix2 - ix1) * (iy2 - iy1), torch.zeros_like(x1))
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    iou = inter / (union + eps)
    diag = (torch.max(x2, x2g) - torch.min(x1, x1g)) ** 2 + (torch.max(y2, y2g) - torch.min(y1, y1g)) ** 2 + eps
    centre = ((x1 + x2) / 2 - (x1g + x2g) / 2) ** 2 + ((y1 + y2) / 2 - (y1g + y2g) / 2) ** 2
    return 1 - iou + centre / diag


def polygon_distance_map(polygon, image_size):
    """SoftRasterizer.make_distance_map (vsrd/transforms/geometric_transforms.py:265-290): distance of every integer pixel
    centre to the closed polygon [P,2] (x, y)."""
    height, width = image_size
    ys, xs = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
    pixels = torch.stack([xs.flatten(), ys.flatten()], dim=-1).to(torch.float32)         # [HW,2]
    start = polygon.to(torch.float32)
    side = torch.roll(start, shifts=-1, dims=0) - start                                   # [P,2]
    rel = pixels[:, None, :] - start[None, :, :]                                          # [HW,P,2]
    ratio = ((side[None] * rel).sum(-1, keepdim=True) / ((side * side).sum(-1, keepdim=True)[None] + 1.0e-6)).clamp(0.0, 1.0)
    return (rel - side[None] * ratio).norm(dim=-1).min(dim=-1).values.reshape(height, width)


def soft_mask(distance_map, inside, temperature=10.0):
    """geometric_transforms.py:306-307."""
    return torch.sigmoid(torch.where(inside, distance_map, -distance_map) / temperature)


# ---------------------------------------------------------------------------------------------------------------------------
# vsrd.operations.box_3d_iou (vsrd/operations/kitti360_operations.py:8-116; caller scripts/main.py:888-905), numpy float64.
# The quirks are part of the behaviour and are kept: the clip's intersection point divides by (determinant + 0.01) (:30), the
# intersection area is capped by the smaller footprint (:104), and the clip assumes counter-clockwise input without checking it --
# for the corner order main.py feeds it the footprints arrive clockwise, the clip keeps the subject polygon, and the "IoU" becomes
# min(area) / (area1 + area2 - min(area)).
# ---------------------------------------------------------------------------------------------------------------------------

def clip_polygon(subject, clip):
    """Sutherland-Hodgman (:8-55): `subject` clipped by every edge of `clip`; None once nothing is left."""
    import numpy as np
    output = [np.asarray(p, dtype=np.float64) for p in subject]
    previous = np.asarray(clip[-1], dtype=np.float64)
    for vertex in clip:
        current = np.asarray(vertex, dtype=np.float64)
        edge = current - previous

        def inside(p):
            return edge[0] * (p[1] - previous[1]) > edge[1] * (p[0] - previous[0])

        def crossing(s, e):
            dc, dp = previous - current, s - e
            n1 = previous[0] * current[1] - previous[1] * current[0]
            n2 = s[0] * e[1] - s[1] * e[0]
            n3 = 1.0 / (dc[0] * dp[1] - dc[1] * dp[0] + 0.01)
            return np.array([(n1 * dp[0] - n2 * dc[0]) * n3, (n1 * dp[1] - n2 * dc[1]) * n3])

        source, output = output, []
        s = source[-1]
        for e in source:
            if inside(e):
                if not inside(s):
                    output.append(crossing(s, e))
                output.append(e)
            elif inside(s):
                output.append(crossing(s, e))
            s = e
        previous = current
        if not output:
            return None
    return output


def box_3d_iou(corners1, corners2):
    """(:83-116) corners [8,3], up = +Z, corners 0-3 the upper face -> (3-D IoU, bird's-eye-view IoU)."""
    import numpy as np
    from scipy.spatial import ConvexHull
    c1, c2 = np.asarray(corners1, dtype=np.float64), np.asarray(corners2, dtype=np.float64)

    def footprint(c):
        return [(c[i, 0], c[i, 1]) for i in (3, 2, 1, 0)]

    def shoelace(poly):
        x, y = np.array([p[0] for p in poly]), np.array([p[1] for p in poly])
        return 0.5 * abs(np.dot(x, np.roll(y, 1)) - np.dot(y, np.roll(x, 1)))

    def volume(c):
        return np.linalg.norm(c[0] - c[1]) * np.linalg.norm(c[1] - c[2]) * np.linalg.norm(c[0] - c[4])

    f1, f2 = footprint(c1), footprint(c2)
    area1, area2 = shoelace(f1), shoelace(f2)
    overlap = clip_polygon(f1, f2)
    inter_area = ConvexHull(np.array(overlap)).volume if overlap is not None else 0.0
    inter_area = min(min(area1, area2), inter_area)
    iou_bev = inter_area / (area1 + area2 - inter_area)
    height = max(0.0, min(c1[0, 2], c2[0, 2]) - max(c1[4, 2], c2[4, 2]))
    inter_volume = inter_area * height
    return inter_volume / (volume(c1) + volume(c2) - inter_volume), iou_bev
