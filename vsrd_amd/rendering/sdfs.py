"""vsrd.rendering.sdfs call surface (reference: vsrd/rendering/sdfs.py:9-58).

Same names and argument meaning as the reference; the returned callables are introspectable
objects (vsrd_amd.fields) instead of closures, so the renderer can hand their parameters to the
HIP library.  They pass through whatever the wrapped callable is (scripts/main.py wraps them
around its own ``instance_field`` closure, main.py:533-537).
"""
from ..fields import BoxSDF, Rotation, Translation, SoftUnion, HardUnion


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
