"""The one transform adjacent to the hot path: SoftRasterizer's pixel-to-polygon distance map and soft masks
(reference: vsrd/transforms/geometric_transforms.py:233-317).  Polygon extraction (cv.findContours) and filling (cv.fillPoly)
stay on the host with OpenCV as in the reference; the brute-force [HW, P] distance computation runs in one HIP launch."""
import torch

from . import _lib


def _pack(polygons, device):
    counts = torch.tensor([int(p.shape[0]) for p in polygons], dtype=torch.int32, device=device)
    pmax = int(counts.max())
    packed = torch.zeros(len(polygons), pmax, 2, dtype=torch.float32, device=device)
    for b, p in enumerate(polygons):
        packed[b, :p.shape[0]] = p.to(device=device, dtype=torch.float32)
    return packed.contiguous(), counts


def make_distance_map(polygons, image_size):
    """polygons: [P,2] tensor (x, y) or a list of such -> distance maps [H,W] / [B,H,W]."""
    single = isinstance(polygons, torch.Tensor) and polygons.dim() == 2
    plist = [polygons] if single else list(polygons)
    device = plist[0].device
    lib = _lib.load()
    packed, counts = _pack(plist, device)
    H, W = int(image_size[0]), int(image_size[1])
    out = torch.empty(len(plist), H, W, dtype=torch.float32, device=device)
    _lib.check(lib.vsrd_polygon_soft_masks(_lib.ptr(packed), _lib.iptr(counts), len(plist), packed.shape[1], H, W, None, 1.0,
                                           _lib.ptr(out), None, _lib.stream()))
    return out[0] if single else out


def soft_masks(polygons, binary_masks, temperature=10.0):
    """SoftRasterizer.forward after the OpenCV steps: polygons (list of [P,2]), binary_masks [B,H,W] bool -> soft masks [B,H,W]."""
    lib = _lib.load()
    device = binary_masks.device
    packed, counts = _pack(list(polygons), device)
    B, H, W = binary_masks.shape
    inside = binary_masks.to(torch.uint8).contiguous()
    out = torch.empty(B, H, W, dtype=torch.float32, device=device)
    _lib.check(lib.vsrd_polygon_soft_masks(_lib.ptr(packed), _lib.iptr(counts), B, packed.shape[1], H, W, inside.data_ptr(), float(temperature),
                                           None, _lib.ptr(out), _lib.stream()))
    return out
