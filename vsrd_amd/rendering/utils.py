"""vsrd.rendering.ray_casting on the HIP library (reference: vsrd/rendering/utils.py:5-18)."""
import torch

from .. import _lib


def ray_casting(image_size, intrinsic_matrices, extrinsic_matrices):
    """Returns (camera_positions [...,3], ray_directions [...,H,W,3]); integer pixel centres (x, y, 1).

    The two small matrix inverses (3x3, 4x4 per view) are host-side LAPACK calls exactly as in the
    reference (utils.py:8-10); the per-pixel part runs in ``vsrd_ray_directions``.
    """
    lib = _lib.load()
    device = intrinsic_matrices.device
    height, width = int(image_size[0]), int(image_size[1])
    inv_k = torch.linalg.inv(intrinsic_matrices.detach().cpu().to(torch.float32))
    inv_e = torch.linalg.inv(extrinsic_matrices.detach().cpu().to(torch.float32))
    back = (inv_e[..., :3, :3] @ inv_k)
    lead = back.shape[:-2]
    flat = back.reshape(-1, 9).contiguous().to(device)
    directions = torch.empty(flat.shape[0], height, width, 3, dtype=torch.float32, device=device)
    _lib.check(lib.vsrd_ray_directions(_lib.ptr(flat), flat.shape[0], height, width, _lib.ptr(directions), _lib.stream()))
    camera_positions = inv_e[..., :3, 3].to(device)
    return camera_positions, directions.reshape(*lead, height, width, 3)
