"""Introspectable signed-distance fields and the flat parameter block the HIP library consumes.

The reference passes the renderer an opaque Python callable assembled from nested closures:
``sdfs.box / rotation / translation`` (vsrd/rendering/sdfs.py:9-37) wrapped around the
``instance_field`` / ``residual_composition`` / ``soft_union`` closures that live *inside*
``scripts/main.py:train()`` (``main.py:433-509``).  A fused kernel needs the parameters behind
that callable and a differentiable link to them, so here

  * the combinators are callable *objects* (``BoxSDF``, ``Rotation``, ``Translation``,
    ``InstanceField``, ``ResidualComposition``, ``ResidualField``, ``SoftUnion``, ``HardUnion``)
    that keep their parameter tensors as attributes, and
  * ``flatten(distance_field)`` turns either such an object tree **or the closure tree an
    unchanged scripts/main.py builds around our ``sdfs.*`` objects** (recognised through the
    closures' free variables) into a ``FieldBlock``: packed instances ``[N,16]`` built with
    differentiable torch ops (so autograd reaches locations / orientations / dimensions),
    the union temperature and the optional per-instance MLP weights.

A callable that cannot be flattened is rejected with ``UnsupportedFieldError``: there is no
generic (per-op PyTorch) rendering path in this package.
"""
import functools
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib


class UnsupportedFieldError(TypeError):
    pass


@dataclass
class FieldBlock:
    instances: torch.Tensor                  # [N,16] = location(3) | rotation row-major(9) | half extents(3) | 0
    temperature: float
    mlp_weights: Optional[torch.Tensor]      # [N,1617] or None
    label_indices: Optional[torch.Tensor]    # instance_label of every instance (None = 0..N-1 in order)
    hard: bool = False
    # the rotations were built by rotation_matrix_y(cos, sin) (BoxParameters3D) and are differentiated only through it: the kernels may
    # leave the adjoints of the five constant matrix entries out (VSRD_FLAG_YAW_GRADIENTS).  Set by the code that built the block from a
    # detector (FrameOptimizer, bench.build_union); never inferred from the closure tree of a drop-in call.
    yaw_gradients: bool = False
    capture_key: Optional[tuple] = None      # flatten(): identity + version of every tensor the closure tree captured (BlockHandOver)

    @property
    def num_instances(self):
        return self.instances.shape[0]


def pack_instances(locations, orientations, dimensions):
    """[N,3], [N,3,3], [N,3] -> [N,16] (differentiable)."""
    n = locations.shape[0]
    return torch.cat([locations.reshape(n, 3), orientations.reshape(n, 9), dimensions.reshape(n, 3),
                      locations.new_zeros(n, 1)], dim=-1).to(torch.float32).contiguous()


# -----------------------------------------------------------------------------------------------
# callable combinators (vsrd/rendering/sdfs.py + the closures of scripts/main.py:433-509)
# -----------------------------------------------------------------------------------------------

class _Field:
    """Base: calling a field evaluates it on the device through ``vsrd_field_eval``."""

    def __call__(self, positions):
        from .rendering import evaluate_field  # late import (rendering imports this module)
        return evaluate_field(self, positions)


class BoxSDF(_Field):
    """sdfs.box(dimension): half-extent box at the origin (sdfs.py:9-19)."""

    def __init__(self, dimension):
        self.dimension = dimension


class Rotation(_Field):
    """sdfs.rotation(sdf, R): evaluates ``sdf(positions @ R)`` (sdfs.py:31-37)."""

    def __init__(self, sdf, rotation_matrix):
        self.sdf, self.rotation_matrix = sdf, rotation_matrix


class Translation(_Field):
    """sdfs.translation(sdf, t): evaluates ``sdf(positions - t)`` (sdfs.py:22-28)."""

    def __init__(self, sdf, translation_vector):
        self.sdf, self.translation_vector = sdf, translation_vector


class ResidualField(_Field):
    """main.py:433-449: sigmoid(MLP_w(encode((|x|,y,z)/100)) - 1) with per-instance weights [1617]."""

    def __init__(self, mlp_weights):
        self.mlp_weights = mlp_weights


class ResidualComposition(_Field):
    """main.py:451-458: distance_field + residual_distance_field."""

    def __init__(self, distance_field, residual_distance_field):
        self.distance_field, self.residual_distance_field = distance_field, residual_distance_field


class InstanceField(_Field):
    """main.py:460-475: (distance, one_hot(instance_label)) features."""

    def __init__(self, distance_field, instance_label, num_instances=None):
        self.distance_field, self.instance_label, self.num_instances = distance_field, instance_label, num_instances


class SoftUnion(_Field):
    """main.py:477-492: temperature soft-min over instance fields."""

    def __init__(self, distance_fields, temperature):
        self.distance_fields, self.temperature = list(distance_fields), temperature


class HardUnion(_Field):
    """main.py:494-509 / sdfs.py:40-47: arg-min over instance fields."""

    def __init__(self, distance_fields):
        self.distance_fields = list(distance_fields)


def instance_field(distance_field, instance_label, num_instances=None):
    return InstanceField(distance_field, instance_label, num_instances)


def residual_composition(distance_field, residual_distance_field):
    return ResidualComposition(distance_field, residual_distance_field)


def soft_union(distance_fields, temperature):
    return SoftUnion(distance_fields, temperature)


def hard_union(distance_fields):
    return HardUnion(distance_fields)


# -----------------------------------------------------------------------------------------------
# recogniser
# -----------------------------------------------------------------------------------------------

def _closure_vars(fn):
    """Free variables of a Python closure as a dict (empty for non-closures)."""
    code = getattr(fn, "__code__", None)
    cells = getattr(fn, "__closure__", None)
    if code is None or not cells:
        return {}
    out = {}
    for name, cell in zip(code.co_freevars, cells):
        try:
            out[name] = cell.cell_contents
        except ValueError:  # empty cell
            pass
    return out


def _as_soft_union(field):
    from .utils import Composition
    if isinstance(field, Composition):       # compose(soft_distance_field, itemgetter(0)), main.py:1030: same field, distances only
        return _as_soft_union(field.functions[0])
    if isinstance(field, SoftUnion):
        return field.distance_fields, field.temperature, False
    if isinstance(field, HardUnion):
        return field.distance_fields, 1.0, True
    free = _closure_vars(field)
    if "distance_fields" in free and "temperature" in free:          # main.py:477 soft_union.wrapper
        return list(free["distance_fields"]), free["temperature"], False
    if set(free) == {"distance_fields"}:                              # main.py:494 hard_union.wrapper
        return list(free["distance_fields"]), 1.0, True
    if isinstance(field, (Translation, Rotation, InstanceField, BoxSDF, ResidualComposition)):
        return [field], 1.0, False                                    # a single instance is a 1-element union
    raise UnsupportedFieldError(
        f"cannot flatten {field!r}: expected vsrd_amd.fields.SoftUnion/HardUnion or the soft_union closure of "
        "scripts/main.py:477-492 built over vsrd_amd.rendering.sdfs objects")


def _unwrap_instance(field):
    """One union member -> (location, rotation, dimension, mlp_weights, instance_label, num_labels); ``num_labels`` is the width of
    the member's one-hot feature (main.py:470: ``num_instances`` of the enclosing train()), None when the member has no label."""
    location = rotation = None
    node = field
    while True:
        if isinstance(node, Translation):
            # the kernel evaluates box((x - t) @ R), i.e. translation(rotation(...)) as main.py:533-537 builds
            # it; rotation(translation(...)) is a different map and is rejected rather than re-derived
            if location is not None or rotation is not None:
                raise UnsupportedFieldError("only translation(rotation(box)) compositions are supported")
            location, node = node.translation_vector, node.sdf
        elif isinstance(node, Rotation):
            if rotation is not None:
                raise UnsupportedFieldError("nested rotations are not supported")
            rotation, node = node.rotation_matrix, node.sdf
        else:
            break
    label = num_labels = None
    if isinstance(node, InstanceField):
        label, num_labels, node = node.instance_label, node.num_instances, node.distance_field
    else:
        free = _closure_vars(node)
        if "instance_label" in free and "distance_field" in free:     # main.py:460 instance_field.wrapper
            label, num_labels, node = free["instance_label"], free.get("num_instances"), free["distance_field"]
    mlp = None
    if isinstance(node, ResidualComposition):
        residual, node = node.residual_distance_field, node.distance_field
        mlp = _residual_weights(residual)
    else:
        free = _closure_vars(node)
        if "residual_distance_field" in free and "distance_field" in free:   # main.py:451 residual_composition.wrapper
            mlp, node = _residual_weights(free["residual_distance_field"]), free["distance_field"]
    if not isinstance(node, BoxSDF):
        raise UnsupportedFieldError(f"innermost field must be sdfs.box(dimension), got {node!r}")
    dimension = node.dimension
    if location is None:
        location = torch.zeros_like(dimension)
    if rotation is None:
        rotation = torch.eye(3, dtype=dimension.dtype, device=dimension.device)
    return location, rotation, dimension, mlp, label, num_labels


def member_label(field):
    """(instance_label, num_labels) of a single union member called on its own, as the reference's soft_union does with every
    member (main.py:480-483): the member returns ``(distances, one_hot(instance_label, num_instances))``.  (None, None) for
    anything that is not a labelled single instance."""
    if not isinstance(field, (Translation, Rotation, InstanceField)):
        return None, None
    *_, label, num_labels = _unwrap_instance(field)
    return (label, num_labels) if (label is not None and num_labels) else (None, None)


def _residual_weights(residual):
    if isinstance(residual, ResidualField):
        return residual.mlp_weights
    free = _closure_vars(residual)                                    # main.py:433 residual_distance_field.wrapper
    inner = free.get("distance_field")
    if isinstance(inner, functools.partial) and inner.args:           # partial(hyper.distance_field, weights), main.py:541
        return inner.args[0]
    raise UnsupportedFieldError(f"cannot extract the per-instance MLP weights from {residual!r}")


def flatten(field) -> FieldBlock:
    """Turn a distance-field callable into the parameter block of ``vsrd_field`` (include/vsrd_hip.h)."""
    if isinstance(field, FieldBlock):
        return field
    members, temperature, hard = _as_soft_union(field)
    if not members:
        raise UnsupportedFieldError("empty union")
    if len(members) > _lib.MAX_INSTANCES:
        raise UnsupportedFieldError(f"{len(members)} instances > VSRD_MAX_INSTANCES={_lib.MAX_INSTANCES}")
    parts = [_unwrap_instance(m) for m in members]
    with torch.enable_grad():         # (also under no_grad: the block of main.py's pass 1 is handed to the differentiable pass 2, below)
        block = _build_block(parts, temperature, hard)
    block.capture_key = _key_of(parts, temperature, hard)        # (what BlockHandOver compares: this walk of the tree serves it too)
    return block


class BlockHandOver:
    """The block flattened for pass 1 of main.py's hierarchical_wrapper (main.py:511-523: the SAME closure rendered twice, pass 1 under
    no_grad, pass 2 with pass 1's `sampled_distances` / `sampled_weights`), handed to pass 2 so that the closure tree is walked and the
    block assembled once per step.  The reuse is scoped to exactly that hand-over: the renderer attaches this object to the
    `sampled_distances` tensor it returns from pass 1, and pass 2 takes the block from the tensor it is given back -- only if it is
    that very tensor, the closure is the same object, and every tensor the closure tree captures is the same object at the same
    version.  Nothing is kept in module state (round 3's thread-local "last block" could outlive a parameter update written through
    `.data` or a raw pointer, which no version counter sees: ADVICE r03); the entry dies with the tensor.
    The handed-over block is SINGLE-USE: it carries one autograd graph from the parameters to `instances`, so a caller that renders
    pass 2 twice from the same `sampled_distances` and calls backward on each would walk that graph twice ("backward through the graph
    a second time"); the first take() empties the hand-over and a second pass 2 flattens the closure afresh.
    The tree is walked twice per step: once by flatten() in pass 1 (which leaves its key on the block) and once here in pass 2.
    ``vsrd_amd.fields.BLOCK_HAND_OVER = False`` switches it off."""

    def __init__(self, field, block):
        self.field, self.block = field, block
        self.key = getattr(block, "capture_key", None) or _capture_key(field)

    def take(self, field):
        block, self.block = self.block, None
        if not BLOCK_HAND_OVER or block is None or field is not self.field:
            return None
        try:
            return block if _capture_key(field) == self.key else None
        except UnsupportedFieldError:
            return None


BLOCK_HAND_OVER = True


def _key_of(parts, temperature, hard):
    return (hard, (id(temperature), temperature._version) if isinstance(temperature, torch.Tensor) else float(temperature),
            tuple((id(t), t._version, t.requires_grad) if isinstance(t, torch.Tensor) else t for p in parts for t in p))


def _capture_key(field):
    members, temperature, hard = _as_soft_union(field)
    return _key_of([_unwrap_instance(m) for m in members], temperature, hard)


def _build_block(parts, temperature, hard) -> FieldBlock:
    locations = torch.stack([p[0].reshape(3) for p in parts])
    rotations = torch.stack([p[1].reshape(3, 3) for p in parts])
    dimensions = torch.stack([p[2].reshape(3) for p in parts])
    mlps = [p[3] for p in parts]
    if any(m is not None for m in mlps):
        if not all(m is not None for m in mlps):
            raise UnsupportedFieldError("either every instance or no instance may carry a residual MLP")
        mlp_weights = torch.stack([m.reshape(-1) for m in mlps]).to(torch.float32).contiguous()
        if mlp_weights.shape[-1] != _lib.MLP_WEIGHTS:
            raise UnsupportedFieldError(f"per-instance MLP must have {_lib.MLP_WEIGHTS} weights")
    else:
        mlp_weights = None
    labels = [p[4] for p in parts]
    label_indices = None
    if any(l is not None for l in labels):
        if all(isinstance(l, torch.Tensor) and l.device == locations.device for l in labels):
            # main.py's instance labels are device tensors: one comparison (one synchronisation) instead of one int() per instance
            stacked = torch.stack([l.reshape(()).to(torch.long) for l in labels])
            if not torch.equal(stacked, torch.arange(len(labels), dtype=torch.long, device=locations.device)):
                label_indices = stacked
        else:
            ints = [int(l) if l is not None else i for i, l in enumerate(labels)]
            if ints != list(range(len(ints))):
                label_indices = torch.tensor(ints, dtype=torch.long, device=locations.device)
    temperature = float(temperature.detach()) if isinstance(temperature, torch.Tensor) else float(temperature)
    return FieldBlock(pack_instances(locations, rotations, dimensions), temperature, mlp_weights, label_indices, hard)
