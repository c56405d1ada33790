"""vsrd.rendering.sdfs call surface (reference: vsrd/rendering/sdfs.py:9-58).

Same names and argument meaning as the reference; the returned callables are introspectable
objects (vsrd_amd.fields) instead of closures, so the renderer can hand their parameters to the
HIP library.  They pass through whatever the wrapped callable is (scripts/main.py wraps them
around its own ``instance_field`` closure, main.py:533-537).
"""
import torch

from ..fields import BoxSDF, Rotation, Translation, SoftUnion, HardUnion


def norm(inputs, *args, epsilon=1e-6, **kwargs):
    """sdfs.py:5-6: sqrt(sum(x^2) + eps), the smoothed norm the box SDF uses (the kernels inline it: field.h kNormEpsilon)."""
    return torch.sqrt(torch.sum(inputs ** 2.0, *args, **kwargs) + epsilon)


def box(dimension):
    return BoxSDF(dimension)


def translation(sdf, translation_vector):
    return Translation(sdf, translation_vector)


def rotation(sdf, rotation_matrix):
    return Rotation(sdf, rotation_matrix)


def hard_union(sdfs):
    return HardUnion(sdfs)


def soft_union(sdfs):
    # sdfs.py:50-58: softmin without a temperature (T = 1)
    return SoftUnion(sdfs, 1.0)
