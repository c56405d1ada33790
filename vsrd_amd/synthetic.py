"""Synthetic KITTI-360-shaped frames: what bench.py, the frame launcher and the full-size tests optimise when no
dataset is mounted.  Seeded, host-side, no GPU work: camera matrices and raw box parameters only."""
import math

import torch


def kitti_intrinsics(height, width):
    """The KITTI-360 perspective intrinsics of the 376 x 1408 images, scaled to the image size."""
    sx, sy = width / 1408.0, height / 376.0
    return torch.tensor([[552.554261 * sx, 0.0, 682.049453 * sx], [0.0, 552.554261 * sy, 238.769549 * sy], [0.0, 0.0, 1.0]])


def synthetic_frame(seed, num_views, height, width, num_instances):
    """KITTI-360 intrinsics, target E = I, sources shifted along z with a small yaw; raw box parameters ~ N(0, 0.5^2) with depth
    forced into 8-60 m.  Returns (K [V,3,3], E [V,4,4], raw locations [1,N,3], raw dimensions [1,N,3], raw orientations [1,N,2])."""
    g = torch.Generator().manual_seed(seed)
    K = kitti_intrinsics(height, width).expand(num_views, 3, 3).contiguous()
    E = torch.eye(4).repeat(num_views, 1, 1)
    half = (num_views - 1) // 2
    offsets = [0] + [k for i in range(1, half + 1) for k in (i, -i)]
    for v, k in enumerate(offsets[:num_views]):
        yaw = math.radians(0.5 * k)
        E[v, :3, :3] = torch.tensor([[math.cos(yaw), 0.0, math.sin(yaw)], [0.0, 1.0, 0.0], [-math.sin(yaw), 0.0, math.cos(yaw)]])
        E[v, 2, 3] = 1.0 * k
    raw_loc = torch.randn(1, num_instances, 3, generator=g) * 0.5
    depth = torch.empty(num_instances).uniform_(8.0, 60.0, generator=g) / 100.0
    raw_loc[0, :, 2] = torch.log(depth / (1.0 - depth))            # sigmoid^-1, decoded z = 100 * sigmoid(raw)
    raw_dim = torch.randn(1, num_instances, 3, generator=g) * 0.5
    raw_ori = torch.nn.functional.normalize(torch.randn(1, num_instances, 2, generator=g), dim=-1)
    return K, E, raw_loc, raw_dim, raw_ori
