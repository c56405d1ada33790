"""vsrd.rendering renderers on the HIP library (reference: vsrd/rendering/renderers.py:177-270,
scripts/main.py:511-523).

Two ways in:

* ``hierarchical_volumetric_rendering`` -- the reference's signature and return tuple, one call per
  pass exactly as ``scripts/main.py``'s ``hierarchical_wrapper`` drives it.  Random draws come from
  the torch generator in the reference's order (``rand_like`` for the stratified pass, ``rand`` +
  ``sort`` for the importance pass), so a seeded run consumes the same stream as the reference
  would on the same device.
* ``render_hierarchical`` -- both passes in ONE kernel launch (pass 1, importance sampling, merge,
  pass 2), uniforms from in-kernel Philox or supplied, only ``labels`` (and what the backward needs)
  written to HBM.  This is the production / benchmark path.

Gradients reach the field parameters through ``torch.autograd.Function``s whose backward is the
hand-derived adjoint kernel (``vsrd_render_backward``); there is no PyTorch-op fallback.
"""
import os
import threading

import torch

from .. import _lib, profiling
from ..fields import BlockHandOver, FieldBlock, SoftUnion, UnsupportedFieldError, flatten, member_label, _closure_vars
from . import generic

class Workspace:
    """Device scratch of one stream of work: the adjoint launches' per-wave gradient partials / residual jets and seeds
    (``vsrd_workspace_bytes``) and the ray sampler's counters.  One grow-only buffer per device serves box-only and residual
    launches alike (the residual layout starts with the box-only one, include/vsrd_hip.h), so a frame that switches phase does not
    hold two.  Launches that share a Workspace must be ordered on one stream; work that runs concurrently on another stream (a
    second frame being optimised, optimization.py) uses its own.  The buffers die with the object: a ``FrameOptimizer`` owns one,
    so a rank working through its shard of frames returns the memory frame by frame (round 1 kept them in a module-global dict
    keyed by ``id(optimizer)``: ~1.7 GB leaked per frame at N = 16)."""

    def __init__(self, allocate=None):
        """``allocate(nbytes) -> uint8 device tensor``: where the buffers come from instead of ``torch.empty`` -- a frame of a batch keeps
        its scratch in its row of the batch's arena (optimization.FrameArena: the kernels reach frame f's copy of every buffer at a fixed
        stride from frame 0's)."""
        self._allocate = allocate
        self._adjoint = {}      # device -> uint8 tensor
        self._sampler = {}      # device -> uint8 tensor
        self._retired = []      # outgrown buffers that captured graphs may still reference (only kept when asked to)
        self.keep_outgrown = False

    def adjoint(self, device, num_instances, residual=False, step_shape=None, backward_shape=None):
        """``step_shape = (num_samples, num_rays)``: scratch of vsrd_render_residual_step for that launch (seeds of a chunk of rays);
        ``backward_shape = (num_distances, num_rays)``: scratch with which vsrd_render_backward runs its two-kernel form on residual fields."""
        need = _lib.load().vsrd_workspace_bytes(int(num_instances), 1 if residual else 0)
        if residual and step_shape is not None:
            need = max(need, _lib.load().vsrd_residual_step_workspace_bytes(int(num_instances), int(step_shape[0]), int(step_shape[1])))
        if residual and backward_shape is not None:
            need = max(need, _lib.load().vsrd_render_backward_workspace_bytes(int(num_instances), 1, int(backward_shape[0]), int(backward_shape[1])))
        buf = self._adjoint.get(device)
        if buf is None or buf.numel() < need:
            if buf is not None and self.keep_outgrown:
                self._retired.append(buf)
            if buf is not None and self._allocate is not None:
                raise RuntimeError("a Workspace inside a frame arena is reserved once (Workspace.reserve): it cannot grow")
            buf = self._adjoint[device] = torch.empty(need, dtype=torch.uint8, device=device) if self._allocate is None else self._allocate(need)
        return buf

    def reserve(self, device, num_instances, residual=True, step_shape=None):
        """Allocate now what the largest later launch will need (a hipGraph captures the buffer's address)."""
        return self.adjoint(device, num_instances, residual, step_shape)

    def sampler(self, device):
        buf = self._sampler.get(device)
        if buf is None:
            buf = self._sampler[device] = torch.zeros(_lib.load().vsrd_sample_rays_workspace_bytes(), dtype=torch.uint8, device=device)
        return buf

    def sampler_overflowed(self, device):
        """True when some vsrd_sample_rays call on this workspace found more candidate keys than its list holds (4096; the
        surplus is dropped in atomic order, so the draw is then neither complete nor deterministic).  Reads the sticky flag back
        (a host synchronisation): call it outside captured regions, e.g. once per frame."""
        buf = self._sampler.get(device)
        if buf is None:
            return False
        offset = 4 * (4096 + 2)                   # SampleScratch: histogram[4096], threshold_bin, num_candidates, overflow
        return bool(buf[offset:offset + 4].view(torch.int32).item())

    def nbytes(self):
        return sum(b.numel() for b in list(self._adjoint.values()) + list(self._sampler.values()) + self._retired)

    def release(self):
        self._adjoint.clear(); self._sampler.clear(); self._retired.clear()


_default_workspace = Workspace()       # launches outside any scope (tests, the API-faithful entry points, bench.py)
_scope = threading.local()


class workspace_scope:
    """``with workspace_scope(workspace): ...`` -- the launches inside use that ``Workspace``'s scratch buffers."""

    def __init__(self, workspace):
        if not isinstance(workspace, Workspace):
            raise TypeError("workspace_scope expects a rendering.Workspace (the owner of the scratch buffers)")
        self.workspace = workspace

    def __enter__(self):
        self.previous = getattr(_scope, "workspace", None)
        _scope.workspace = self.workspace
        return self.workspace

    def __exit__(self, *exc):
        _scope.workspace = self.previous


def current_workspace():
    return getattr(_scope, "workspace", None) or _default_workspace


# A/B switch for the conservative soft-min instance culling (DESIGN.md "Culling"): False sets VSRD_FLAG_NO_CULLING on
# every launch so that each instance is evaluated at each sample, exactly like the reference's closure loop.
CULLING = True
# A/B switch for the soft-min shift (field.h: union_accumulate): True sets VSRD_FLAG_RUNNING_MINIMUM on every launch.
RUNNING_MINIMUM = os.environ.get("VSRD_RUNNING_MINIMUM", "0") == "1"
# A/B switch for the y-rotation fast path (field.h: box_value<true>): True sets VSRD_FLAG_GENERAL_ROTATIONS on every launch.
GENERAL_ROTATIONS = os.environ.get("VSRD_GENERAL_ROTATIONS", "0") == "1"
# A/B switch for the fused residual step: True sets VSRD_FLAG_RESIDUAL_SINGLE_KERNEL (one kernel, one wave per SIMD) instead of the
# default two kernels per chunk of rays (render_kernels.h: residual_step_front_kernel + residual_mlp_adjoint_kernel).
RESIDUAL_SINGLE_KERNEL = os.environ.get("VSRD_RESIDUAL_SINGLE_KERNEL", "0") == "1"
# A/B switch for the front kernel of residual steps: True sets VSRD_FLAG_RESIDUAL_WAVE_PER_RAY (one wave per ray) where the default splits
# every ray over the two waves of a workgroup (render_kernels.h: residual_step_pair_kernel; S > 64, or <= 2048 rays).
RESIDUAL_WAVE_PER_RAY = os.environ.get("VSRD_RESIDUAL_WAVE_PER_RAY", "0") == "1"
# A/B switch for the box-only fused step: True sets VSRD_FLAG_STEP_WAVE_PER_RAY (render_silhouette_kernel, one wave per ray) where the
# default puts four consecutive rays in a wave for dense launches with S <= 64 and N <= 16 (quad_step.h: render_silhouette_quad_kernel).
STEP_WAVE_PER_RAY = os.environ.get("VSRD_STEP_WAVE_PER_RAY", "0") == "1"
# ... and True here sets VSRD_FLAG_STEP_SPLIT_RAY: every ray split over the two waves of a workgroup (render_silhouette_split_kernel), what
# launches of <= 2048 gathered rays -- the reference's 1000 sampled rays per step -- do by themselves.
STEP_SPLIT_RAY = os.environ.get("VSRD_STEP_SPLIT_RAY", "0") == "1"


# A/B switch for the residual step's front kernel: True sets VSRD_FLAG_MLP_SPLIT_BF16 (the per-instance MLP's products on the bf16 matrix
# instruction with both operands split into two bfloat16 parts; csrc/residual.h) instead of the exact-fp32 matrix instruction.
MLP_SPLIT_BF16 = os.environ.get("VSRD_MLP_SPLIT_BF16", "0") == "1"


def _base_flags():
    return ((0 if CULLING else _lib.FLAG_NO_CULLING) | (_lib.FLAG_MLP_SPLIT_BF16 if MLP_SPLIT_BF16 else 0) | (_lib.FLAG_RUNNING_MINIMUM if RUNNING_MINIMUM else 0)
            | (_lib.FLAG_GENERAL_ROTATIONS if GENERAL_ROTATIONS else 0) | (_lib.FLAG_RESIDUAL_SINGLE_KERNEL if RESIDUAL_SINGLE_KERNEL else 0)
            | (_lib.FLAG_RESIDUAL_WAVE_PER_RAY if RESIDUAL_WAVE_PER_RAY else 0) | (_lib.FLAG_STEP_WAVE_PER_RAY if STEP_WAVE_PER_RAY else 0) | (_lib.FLAG_STEP_SPLIT_RAY if STEP_SPLIT_RAY else 0))


def _mlp_flag(centred_weights):
    return _lib.FLAG_MLP_WEIGHTS_CENTRED if centred_weights is not None else 0


def _workspace(device, num_instances, residual=False, step_shape=None, backward_shape=None):
    return current_workspace().adjoint(device, num_instances, residual, step_shape, backward_shape)


def _prepare_rays(ray_positions, ray_directions):
    lead = ray_directions.shape[:-1]
    directions = ray_directions.reshape(-1, 3).to(torch.float32).contiguous()
    if ray_positions.dim() == 1:
        origins, stride = ray_positions.to(torch.float32).contiguous(), 0
    else:
        origins = ray_positions.expand(*lead, 3).reshape(-1, 3).to(torch.float32).contiguous()
        stride = 3
    return origins, directions, stride, lead


class _RenderAtDistances(torch.autograd.Function):
    """renderers.py:212-270 at given sorted distances (ray-major [R,D])."""

    @staticmethod
    def forward(ctx, instances, mlp_weights, origins, directions, distances, temperature, scalars, origin_stride):
        lib = _lib.load()
        std, ratio, eps, near, far, num_samples, schedule = _unpack(scalars)
        R, D = distances.shape
        N = instances.shape[0]
        instances = instances.detach().contiguous()
        labels = torch.empty(R, N, dtype=torch.float32, device=distances.device)
        gradients = torch.empty(R, D - 1, 3, dtype=torch.float32, device=distances.device)
        weights = torch.empty(R, D - 1, dtype=torch.float32, device=distances.device)
        mlp_weights = None if mlp_weights is None else _centre_mlp(mlp_weights)
        field = _lib.make_field(instances, temperature, mlp_weights)
        config = _lib.make_config(R, num_samples, (near, far), std, ratio, eps, origin_stride, flags=_base_flags() | _mlp_flag(mlp_weights),
                                  schedule=schedule)
        with profiling.timed("vsrd_render_forward"):
            _lib.check(lib.vsrd_render_forward(field, config, _lib.ptr(origins), _lib.ptr(directions), _lib.ptr(distances), D,
                                               _lib.ptr(labels), _lib.ptr(gradients), _lib.ptr(weights), _lib.stream()))
        ctx.residual = mlp_weights is not None
        ctx.save_for_backward(instances, origins, directions, distances, *([mlp_weights] if ctx.residual else []))
        ctx.meta = (temperature, scalars, origin_stride)
        ctx.set_materialize_grads(False)   # unused outputs arrive as None, not as zero tensors
        return labels, gradients, weights

    @staticmethod
    def backward(ctx, grad_labels, grad_gradients, grad_weights):
        instances, origins, directions, distances, *rest = ctx.saved_tensors
        temperature, scalars, origin_stride = ctx.meta
        grad_instances, grad_mlp = _backward(instances, rest[0] if rest else None, origins, directions, distances, temperature, scalars,
                                             origin_stride, grad_labels, grad_gradients, grad_weights)
        return (grad_instances, grad_mlp, None, None, None, None, None, None)


def _centre_mlp(mlp_weights):
    """The four linears that feed a LayerNorm with their columns (bias column included) centred over the 16 output channels:
    LayerNorm makes the field invariant to it and the gradients w.r.t. the centred weights ARE the gradients w.r.t. the originals
    (the adjoint's z_bar has zero channel mean), so the kernels get VSRD_FLAG_MLP_WEIGHTS_CENTRED and skip centring the weight
    operands on each of their ~10^5..10^7 evaluations (residual.h: load_forward_weights)."""
    weights = mlp_weights.detach()
    if weights.is_cuda and weights.dtype == torch.float32 and weights.dim() == 2 and weights.shape[1] == _lib.MLP_WEIGHTS:
        weights = weights.contiguous()
        out = torch.empty_like(weights)            # one launch (csrc/hypernetwork.h: hyper_centre_kernel) instead of ten element-wise ones
        _lib.check(_lib.load().vsrd_centre_mlp_weights(_lib.ptr(weights), weights.shape[0], _lib.ptr(out), _lib.stream()))
        return out
    return _centre_mlp_torch(weights)


def _centre_mlp_torch(mlp_weights):
    """`_centre_mlp` in torch operations: the reference the kernel is tested against (tests/test_hip_step.py)."""
    out = mlp_weights.detach().clone()
    n = out.shape[0]
    first = out[:, :784].view(n, 16, 49)
    first -= first.mean(dim=1, keepdim=True)
    for layer in range(3):
        block = out[:, 784 + 272 * layer:784 + 272 * (layer + 1)].view(n, 16, 17)
        block -= block.mean(dim=1, keepdim=True)
    return out


def _unpack(scalars):
    """(std, ratio, eps, near, far, S[, schedule]) -> 7-tuple; `schedule` is the optional device tensor of _lib.make_config."""
    return tuple(scalars) if len(scalars) == 7 else tuple(scalars) + (None,)


def _offset(stream_offset):
    return stream_offset if isinstance(stream_offset, torch.Tensor) else int(stream_offset)


def _backward(instances, mlp_weights, origins, directions, distances, temperature, scalars, origin_stride,
              grad_labels, grad_gradients, grad_weights):
    lib = _lib.load()
    std, ratio, eps, near, far, num_samples, schedule = _unpack(scalars)
    R, D = distances.shape
    N = instances.shape[0]
    grad_labels = torch.zeros(R, N, dtype=torch.float32, device=distances.device) if grad_labels is None \
        else grad_labels.to(torch.float32).contiguous()
    grad_gradients = None if grad_gradients is None else grad_gradients.to(torch.float32).contiguous()
    grad_weights = None if grad_weights is None else grad_weights.to(torch.float32).contiguous()
    grad_instances = torch.empty_like(instances)
    grad_mlp = None if mlp_weights is None else torch.empty_like(mlp_weights)
    workspace = _workspace(distances.device, N, mlp_weights is not None, backward_shape=(D, R))
    field = _lib.make_field(instances, temperature, mlp_weights)
    config = _lib.make_config(R, num_samples, (near, far), std, ratio, eps, origin_stride, flags=_base_flags() | _mlp_flag(mlp_weights),
                              schedule=schedule)
    with profiling.timed("vsrd_render_backward"):
        _lib.check(lib.vsrd_render_backward(field, config, _lib.ptr(origins), _lib.ptr(directions), _lib.ptr(distances), D,
                                            _lib.ptr(grad_labels), _lib.ptr(grad_gradients), _lib.ptr(grad_weights),
                                            workspace.data_ptr(), workspace.numel(), _lib.ptr(grad_instances), _lib.ptr(grad_mlp),
                                            _lib.stream()))
    return grad_instances, grad_mlp


class _RenderHierarchical(torch.autograd.Function):
    """scripts/main.py:511-523 around renderers.py:177-270, both passes in one launch."""

    @staticmethod
    def forward(ctx, instances, mlp_weights, origins, directions, u_coarse, u_fine, temperature, scalars, origin_stride,
                seed, stream_offset, flags, want_gradients, want_weights, want_uniforms, want_coarse_weights=False):
        lib = _lib.load()
        std, ratio, eps, near, far, S, schedule = _unpack(scalars)
        R = directions.shape[0]
        N = instances.shape[0]
        dev = directions.device
        instances = instances.detach().contiguous()
        labels = torch.empty(R, N, dtype=torch.float32, device=dev)
        distances = torch.empty(R, 2 * S, dtype=torch.float32, device=dev)
        gradients = torch.empty(R, 2 * S - 1, 3, dtype=torch.float32, device=dev) if want_gradients else None
        weights = torch.empty(R, 2 * S - 1, dtype=torch.float32, device=dev) if want_weights else None
        uc_out = torch.empty(R, S, dtype=torch.float32, device=dev) if want_uniforms else None
        uf_out = torch.empty(R, S, dtype=torch.float32, device=dev) if want_uniforms else None
        coarse_weights = torch.empty(R, S - 1, dtype=torch.float32, device=dev) if want_coarse_weights else None
        mlp_weights = None if mlp_weights is None else _centre_mlp(mlp_weights)
        field = _lib.make_field(instances, temperature, mlp_weights)
        ctx.residual = mlp_weights is not None
        ctx.mlp = mlp_weights                                   # the centred copy: what the backward launch is given as well
        config = _lib.make_config(R, S, (near, far), std, ratio, eps, origin_stride, seed, stream_offset, flags | _mlp_flag(mlp_weights),
                                  schedule=schedule)
        with profiling.timed("vsrd_render_hierarchical_forward"):
            _lib.check(lib.vsrd_render_hierarchical_forward(
                field, config, _lib.ptr(origins), _lib.ptr(directions), _lib.ptr(u_coarse), _lib.ptr(u_fine),
                _lib.ptr(labels), _lib.ptr(distances), _lib.ptr(gradients), _lib.ptr(weights), _lib.ptr(coarse_weights),
                _lib.ptr(uc_out), _lib.ptr(uf_out), _lib.stream()))
        ctx.save_for_backward(instances, origins, directions, distances)
        ctx.meta = (temperature, scalars, origin_stride)
        ctx.set_materialize_grads(False)
        outs = (labels, gradients if want_gradients else labels.new_empty(0), weights if want_weights else labels.new_empty(0),
                distances, uc_out if want_uniforms else labels.new_empty(0), uf_out if want_uniforms else labels.new_empty(0),
                coarse_weights if want_coarse_weights else labels.new_empty(0))
        ctx.mark_non_differentiable(outs[3], outs[4], outs[5], outs[6])
        return outs

    @staticmethod
    def backward(ctx, grad_labels, grad_gradients, grad_weights, _gd, _gu1, _gu2, _gcw):
        instances, origins, directions, distances = ctx.saved_tensors
        temperature, scalars, origin_stride = ctx.meta
        if grad_gradients is not None and grad_gradients.numel() == 0:
            grad_gradients = None
        if grad_weights is not None and grad_weights.numel() == 0:
            grad_weights = None
        grad, grad_mlp = _backward(instances, ctx.mlp, origins, directions, distances, temperature, scalars, origin_stride,
                                   grad_labels, grad_gradients, grad_weights)
        return (grad, grad_mlp) + (None,) * 14


def _scatter_labels(labels, block: FieldBlock):
    """One-hot features are indexed by instance_label (main.py:470); identity unless labels were permuted."""
    if block.label_indices is None:
        return labels
    out = torch.zeros_like(labels)
    return out.index_add(-1, block.label_indices, labels)


def render_at_distances(distance_field, ray_positions, ray_directions, distances, sdf_std_deviation,
                        cosine_ratio=1.0, epsilon=1.0e-6):
    """Ray-major core: distances [R,D] sorted -> (labels [R,N], gradients [R,D-1,3], weights [R,D-1])."""
    block = flatten(distance_field)
    origins, directions, stride, _ = _prepare_rays(ray_positions, ray_directions)
    distances = distances.to(torch.float32).contiguous()
    scalars = (float(sdf_std_deviation), float(cosine_ratio), float(epsilon), 0.0, 1.0, max(2, (distances.shape[1] + 1) // 2))
    labels, gradients, weights = _RenderAtDistances.apply(block.instances, block.mlp_weights, origins, directions, distances,
                                                          block.temperature, scalars, stride)
    return _scatter_labels(labels, block), gradients, weights


def render_hierarchical(distance_field, ray_positions, ray_directions, distance_range, num_samples, sdf_std_deviation,
                        cosine_ratio=1.0, epsilon=1.0e-6, u_coarse=None, u_fine=None, seed=0, stream_offset=0,
                        return_gradients=False, return_weights=False, return_uniforms=False, skip_exact_misses=False, schedule=None,
                        return_coarse_weights=False):
    """Fused two-pass render.  Returns a dict: labels [R,N], distances [R,2S], optionally gradients
    [R,2S-1,3], weights [R,2S-1], u_coarse/u_fine [R,S] (the uniforms actually used; u_fine sorted when drawn in the kernel),
    coarse_weights [R,S-1] (pass 1's compositing weights: what pass 1 of main.py:511-523 hands to pass 2).
    ``schedule`` (device float32 [3] = temperature, sdf_std_deviation, cosine_ratio) and a tensor ``stream_offset`` are read on the
    device instead of the scalar arguments (hipGraph replay; see include/vsrd_hip.h)."""
    block = flatten(distance_field)
    origins, directions, stride, _ = _prepare_rays(ray_positions, ray_directions)
    if (u_coarse is None) != (u_fine is None):
        raise ValueError("pass both u_coarse and u_fine, or neither (in-kernel Philox)")
    if u_coarse is not None:
        u_coarse = u_coarse.reshape(-1, num_samples).to(torch.float32).contiguous()
        u_fine = u_fine.reshape(-1, num_samples).to(torch.float32).contiguous()
    flags = (_lib.FLAG_SKIP_EXACT_MISSES if skip_exact_misses else 0) | _base_flags()
    scalars = (float(sdf_std_deviation), float(cosine_ratio), float(epsilon), float(distance_range[0]),
               float(distance_range[1]), int(num_samples), schedule)
    labels, gradients, weights, distances, uc, uf, cw = _RenderHierarchical.apply(
        block.instances, block.mlp_weights, origins, directions, u_coarse, u_fine, block.temperature, scalars, stride,
        int(seed), _offset(stream_offset), flags, bool(return_gradients), bool(return_weights), bool(return_uniforms), bool(return_coarse_weights))
    out = dict(labels=_scatter_labels(labels, block), distances=distances)
    if return_gradients:
        out["gradients"] = gradients
    if return_weights:
        out["weights"] = weights
    if return_uniforms:
        out["u_coarse"], out["u_fine"] = uc, uf
    if return_coarse_weights:
        out["coarse_weights"] = cw
    return out


def hierarchical_volumetric_rendering(
    distance_field,
    ray_positions,
    ray_directions,
    distance_range,
    num_samples,
    sdf_std_deviation,
    cosine_ratio=1.0,
    epsilon=1e-6,
    sampled_distances=None,
    sampled_weights=None,
):
    """Drop-in for vsrd.rendering.hierarchical_volumetric_rendering (renderers.py:177-270).

    Returns ``(labels [...,N], sampled_gradients [S',...,3], sampled_distances [D,...,1],
    sampled_weights [S',...,1])`` with the reference's sample-major shapes (permuted views of the
    ray-major device buffers, just as the reference returns permuted views).
    """
    lib = _lib.load()
    # pass 2 of main.py's hierarchical_wrapper is given back the very `sampled_distances` tensor pass 1 returned: it carries pass 1's block
    hand_over = getattr(sampled_distances, "_vsrd_block_hand_over", None) if sampled_distances is not None else None
    block = hand_over.take(distance_field) if hand_over is not None else None
    if block is None:
        try:
            block = flatten(distance_field)
        except UnsupportedFieldError as reason:
            # a field the kernels do not know: the same algorithm with torch operations on the device (generic.py), not an exception
            generic.warn_once(reason)
            return generic.hierarchical_volumetric_rendering(distance_field, ray_positions, ray_directions, distance_range, num_samples,
                                                             sdf_std_deviation, cosine_ratio, epsilon, sampled_distances, sampled_weights)
    origins, directions, stride, lead = _prepare_rays(ray_positions, ray_directions)
    R = directions.shape[0]
    dev = directions.device
    with torch.no_grad():
        if sampled_distances is None:
            # renderers.py:191-194 + samplers.py:5-8 (rand_like over [..., 1, S])
            u_coarse = torch.rand(*lead, 1, num_samples, device=dev).reshape(R, num_samples)
            distances = torch.empty(R, num_samples, dtype=torch.float32, device=dev)
            config = _lib.make_config(R, num_samples, distance_range, sdf_std_deviation, cosine_ratio, epsilon, stride)
            _lib.check(lib.vsrd_sample_stratified(config, _lib.ptr(u_coarse), _lib.ptr(distances), _lib.stream()))
        else:
            # renderers.py:198-210 + samplers.py:11-36 (rand over [..., 1, S], sorted)
            coarse = sampled_distances.reshape(sampled_distances.shape[0], R).t().to(torch.float32).contiguous()
            cweights = sampled_weights.reshape(sampled_weights.shape[0], R).t().to(torch.float32).contiguous()
            if coarse.shape[1] != num_samples or cweights.shape[1] != num_samples - 1:
                raise NotImplementedError("the importance pass expects num_samples coarse distances and num_samples-1 weights")
            u_fine = torch.sort(torch.rand(*lead, 1, num_samples, device=dev), dim=-1).values.reshape(R, num_samples).contiguous()
            distances = torch.empty(R, 2 * num_samples, dtype=torch.float32, device=dev)
            config = _lib.make_config(R, num_samples, distance_range, sdf_std_deviation, cosine_ratio, epsilon, stride,
                                      flags=_lib.FLAG_FINE_UNIFORMS_SORTED)
            _lib.check(lib.vsrd_sample_importance(config, _lib.ptr(coarse), _lib.ptr(cweights), _lib.ptr(u_fine),
                                                  _lib.ptr(distances), None, _lib.stream()))
    scalars = (float(sdf_std_deviation), float(cosine_ratio), float(epsilon), float(distance_range[0]),
               float(distance_range[1]), int(num_samples))
    labels, gradients, weights = _RenderAtDistances.apply(block.instances, block.mlp_weights, origins, directions, distances,
                                                          block.temperature, scalars, stride)
    labels = _scatter_labels(labels, block)
    D = distances.shape[1]
    distances_out = distances.reshape(*lead, D).movedim(-1, 0).unsqueeze(-1)
    if sampled_distances is None and not isinstance(distance_field, FieldBlock):
        distances_out._vsrd_block_hand_over = BlockHandOver(distance_field, block)      # (fields.BlockHandOver: pass 1 -> pass 2 only)
    return (
        labels.reshape(*lead, -1),
        gradients.reshape(*lead, D - 1, 3).movedim(-2, 0),
        distances_out,
        weights.reshape(*lead, D - 1).movedim(-1, 0).unsqueeze(-1),
    )


class _EvaluateField(torch.autograd.Function):
    """vsrd_field_eval / vsrd_field_eval_backward: the closure call of scripts/main.py:433-509, differentiable w.r.t. the field
    parameters and the positions through the distances and the labels (the analytic normal output is not differentiated)."""

    @staticmethod
    def forward(ctx, instances, mlp_weights, positions, temperature, hard, want_gradients, want_labels):
        lib = _lib.load()
        P, N = positions.shape[0], instances.shape[0]
        dev = positions.device
        instances = instances.detach().contiguous()
        mlp_weights = None if mlp_weights is None else mlp_weights.detach().contiguous()
        distances = torch.empty(P, dtype=torch.float32, device=dev)
        gradients = torch.empty(P, 3, dtype=torch.float32, device=dev) if want_gradients else None
        labels = torch.empty(P, N, dtype=torch.float32, device=dev) if want_labels else None
        field = _lib.make_field(instances, temperature, mlp_weights)
        _lib.check(lib.vsrd_field_eval(field, _lib.ptr(positions), P, _lib.ptr(distances), _lib.ptr(gradients), _lib.ptr(labels),
                                       1 if hard else 0, _lib.stream()))
        ctx.save_for_backward(instances, positions, *([mlp_weights] if mlp_weights is not None else []))
        ctx.meta = (temperature, hard)
        ctx.set_materialize_grads(False)
        outs = (distances, labels if want_labels else distances.new_empty(0), gradients if want_gradients else distances.new_empty(0))
        ctx.mark_non_differentiable(outs[2])
        return outs

    @staticmethod
    def backward(ctx, grad_distances, grad_labels, _grad_normals):
        lib = _lib.load()
        instances, positions, *rest = ctx.saved_tensors
        mlp_weights = rest[0] if rest else None
        temperature, hard = ctx.meta
        if grad_labels is not None and grad_labels.numel() == 0:
            grad_labels = None
        P, N = positions.shape[0], instances.shape[0]
        grad_distances = None if grad_distances is None else grad_distances.to(torch.float32).contiguous()
        grad_labels = None if grad_labels is None else grad_labels.to(torch.float32).contiguous()
        grad_positions = torch.empty_like(positions) if ctx.needs_input_grad[2] else None
        grad_instances = torch.empty_like(instances)
        grad_mlp = None if mlp_weights is None else torch.empty_like(mlp_weights)
        workspace = _workspace(positions.device, N, mlp_weights is not None)
        field = _lib.make_field(instances, temperature, mlp_weights)
        _lib.check(lib.vsrd_field_eval_backward(field, _lib.ptr(positions), P, _lib.ptr(grad_distances), _lib.ptr(grad_labels), 1 if hard else 0,
                                                workspace.data_ptr(), workspace.numel(), _lib.ptr(grad_positions), _lib.ptr(grad_instances),
                                                _lib.ptr(grad_mlp), _lib.stream()))
        return grad_instances, grad_mlp, grad_positions, None, None, None, None


def evaluate_field(distance_field, positions, with_gradients=False, with_labels=None):
    """What calling the reference closure does (main.py:477-509): soft union -> (distances [...,1], labels [...,N]);
    hard union / plain sdfs -> distances [...,1].  ``with_gradients`` adds the analytic normal [...,3].
    Distances and labels are autograd-connected to the field parameters and to ``positions`` (vsrd_field_eval_backward)."""
    block = flatten(distance_field)
    one_hot = None
    if with_labels is None:  # the soft union returns (distances, features); plain sdfs / hard unions distances only
        with_labels = (not block.hard) and (isinstance(distance_field, SoftUnion) or "distance_fields" in _closure_vars(distance_field))
        # a single labelled member (what main.py's soft_union calls N times, main.py:480-483) returns its one-hot feature as well
        label, num_labels = member_label(distance_field)
        if label is not None and not with_gradients:
            one_hot = torch.nn.functional.one_hot(torch.as_tensor(label, dtype=torch.long, device=positions.device), int(num_labels))
    lead = positions.shape[:-1]
    pts = positions.reshape(-1, 3).to(torch.float32).contiguous()
    N = block.num_instances
    distances, labels, gradients = _EvaluateField.apply(block.instances, block.mlp_weights, pts, block.temperature, bool(block.hard),
                                                        bool(with_gradients), bool(with_labels))
    out = [distances.reshape(*lead, 1)]
    if one_hot is not None:
        out.append(one_hot.expand(*lead, -1))
    if with_labels:
        out.append(_scatter_labels(labels, block).reshape(*lead, N))
    if with_gradients:
        out.append(gradients.reshape(*lead, 3))
    return out[0] if len(out) == 1 else tuple(out)


def sphere_tracing(distance_field, ray_positions, ray_directions, num_iterations, convergence_criteria, foreground_masks=None,
                   bounding_radius=None, initialization=True, differentiable=False):
    """Drop-in for vsrd.rendering.sphere_tracing (renderers.py:21-73): returns (ray_positions [...,3], convergence_masks [...,1])."""
    lib = _lib.load()
    block = flatten(distance_field)
    origins, directions, stride, lead = _prepare_rays(ray_positions, ray_directions)
    R = directions.shape[0]
    positions = torch.empty(R, 3, dtype=torch.float32, device=directions.device)
    converged = torch.empty(R, dtype=torch.uint8, device=directions.device)
    fg = None
    if foreground_masks is not None:
        fg = foreground_masks.expand(*lead, 1).reshape(R).to(torch.uint8).contiguous()
    mlp = None if block.mlp_weights is None else block.mlp_weights.detach().contiguous()
    field = _lib.make_field(block.instances.detach().contiguous(), block.temperature, mlp)
    _lib.check(lib.vsrd_sphere_trace(field, _lib.ptr(origins), stride, _lib.ptr(directions), None if fg is None else fg.data_ptr(), R,
                                     int(num_iterations), float(convergence_criteria), float(bounding_radius or 0.0),
                                     1 if initialization else 0, 1 if block.hard else 0, _lib.ptr(positions), converged.data_ptr(),
                                     _lib.stream()))
    positions, converged = positions.reshape(*lead, 3), converged.to(torch.bool).reshape(*lead, 1)
    if differentiable:
        # renderers.py:59-72: one Newton step along the ray at the traced point; the step length -sdf / (grad sdf . r) carries the
        # dependence on the field parameters (the normal enters as a constant, exactly as the reference's autograd.grad without
        # create_graph leaves it)
        out = evaluate_field(distance_field, positions, with_gradients=True, with_labels=False)
        sdf, normals = out[0], out[-1]
        step = -sdf / (normals * ray_directions.to(normals.dtype)).sum(dim=-1, keepdim=True)
        positions = torch.where(converged, positions + ray_directions * step, positions)
    return positions, converged


def sphere_intersection(ray_positions, ray_directions, bounding_radius):
    """vsrd.rendering.sphere_intersection (renderers.py:10-18): entry / exit distances of rays with the origin-centred sphere and
    the hit mask (the entry point is also computed inside vsrd_sphere_trace when ``initialization`` is set)."""
    a = (ray_directions * ray_directions).sum(dim=-1, keepdim=True)
    b = (ray_directions * ray_positions).sum(dim=-1, keepdim=True)
    c = (ray_positions * ray_positions).sum(dim=-1, keepdim=True) - bounding_radius ** 2.0
    discriminant = b ** 2.0 - a * c
    root = torch.sqrt(discriminant)
    return (-b - root) / a, (-b + root) / a, discriminant >= 0.0


def phong_shading(ray_directions, surface_normals, light_directions, light_ambient_colors, light_diffuse_colors, light_specular_colors,
                  material_ambient_colors, material_diffuse_colors, material_specular_colors, material_emission_colors, material_shininesses):
    """vsrd.rendering.phong_shading (renderers.py:116-146), used by the reference's visualisation only: element-wise torch."""
    view = torch.nn.functional.normalize(ray_directions, dim=-1)
    normal = torch.nn.functional.normalize(surface_normals, dim=-1)
    light = torch.nn.functional.normalize(light_directions, dim=-1)
    incidence = (light * normal).sum(dim=-1, keepdim=True)
    reflected = light - 2.0 * normal * incidence
    diffuse = torch.relu(-incidence)
    specular = torch.relu(-(reflected * view).sum(dim=-1, keepdim=True)) ** material_shininesses
    colors = (material_emission_colors + material_ambient_colors * light_ambient_colors
              + material_diffuse_colors * light_diffuse_colors * diffuse + material_specular_colors * light_specular_colors * specular)
    return colors.clamp(0.0, 1.0)


def shadow_rendering(distance_field, surface_positions, surface_normals, light_directions, num_iterations, convergence_criteria,
                     foreground_masks, bounding_radius=None, initialization=False, implicit_differentiation=False):
    """vsrd.rendering.shadow_rendering (renderers.py:149-174): trace from just above the surface towards the light; a point whose
    ray converges on geometry is in shadow."""
    _, convergence_masks = sphere_tracing(distance_field, surface_positions + surface_normals * convergence_criteria, -light_directions,
                                          num_iterations, convergence_criteria, foreground_masks=foreground_masks,
                                          bounding_radius=bounding_radius, initialization=initialization,
                                          differentiable=implicit_differentiation)
    return foreground_masks & convergence_masks


def surface_normal(distance_field, surface_positions, finite_difference_epsilon=None):
    """Drop-in for vsrd.rendering.surface_normal (renderers.py:76-113): unit normals of the field at the given points
    (analytic gradient, or central differences when finite_difference_epsilon is given).  Not differentiable w.r.t. the field."""
    if finite_difference_epsilon:
        eps = float(finite_difference_epsilon)
        offsets = torch.eye(3, device=surface_positions.device, dtype=surface_positions.dtype) * eps
        probes = torch.stack([surface_positions + offsets[k] for k in range(3)] + [surface_positions - offsets[k] for k in range(3)])
        d = evaluate_field(distance_field, probes, with_labels=False)
        normals = torch.cat([d[k] - d[k + 3] for k in range(3)], dim=-1)
    else:
        out = evaluate_field(distance_field, surface_positions, with_gradients=True, with_labels=False)
        normals = out[-1] if isinstance(out, tuple) else out
    return torch.nn.functional.normalize(normals, dim=-1)


class _SilhouetteStep(torch.autograd.Function):
    """Fused render + silhouette BCE + adjoint (vsrd_render_silhouette_step).  The kernel produces the loss and its gradient
    w.r.t. the packed instances in the forward; backward only scales that gradient."""

    @staticmethod
    def forward(ctx, instances, origins, directions, targets, weights, u_coarse, u_fine, temperature, scalars, origin_stride,
                seed, stream_offset, flags, loss_scale, want_labels, want_samples=False):
        lib = _lib.load()
        std, ratio, eps, near, far, S, schedule = _unpack(scalars)
        R, N = directions.shape[0], instances.shape[0]
        dev = directions.device
        instances = instances.detach().contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        grad = torch.empty_like(instances)
        labels = torch.empty(R, N, dtype=torch.float32, device=dev) if want_labels else None
        samples = None
        if want_samples:          # the step's own state between its passes (vsrd_render_config::out_*)
            samples = (torch.empty(R, 2 * S, dtype=torch.float32, device=dev), torch.empty(R, S - 1, dtype=torch.float32, device=dev),
                       torch.empty(R, S, dtype=torch.float32, device=dev), torch.empty(R, S, dtype=torch.float32, device=dev))
        workspace = _workspace(dev, N, False)
        field = _lib.make_field(instances, temperature)
        config = _lib.make_config(R, S, (near, far), std, ratio, eps, origin_stride, seed, stream_offset, flags, schedule=schedule, samples=samples)
        with profiling.timed("vsrd_render_silhouette_step"):
            _lib.check(lib.vsrd_render_silhouette_step(field, config, _lib.ptr(origins), _lib.ptr(directions), _lib.ptr(u_coarse), _lib.ptr(u_fine),
                                                       _lib.ptr(targets), _lib.ptr(weights), float(loss_scale), workspace.data_ptr(),
                                                       workspace.numel(), _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(labels), _lib.stream()))
        ctx.save_for_backward(grad)
        out_labels = labels if want_labels else loss.new_empty(0)
        out_samples = samples if want_samples else tuple(loss.new_empty(0) for _ in range(4))
        ctx.mark_non_differentiable(out_labels, *out_samples)
        return (loss[0], out_labels) + tuple(out_samples)

    @staticmethod
    def backward(ctx, grad_loss, *_unused):
        grad, = ctx.saved_tensors
        return (grad * grad_loss,) + (None,) * 15


class _ResidualStep(torch.autograd.Function):
    """vsrd_render_residual_step: render + silhouette BCE + eikonal term + adjoint of a residual field in one launch.  The kernel
    produces losses = (silhouette, eikonal) and the gradient of  silhouette + eikonal_ratio * eikonal  w.r.t. instances / MLP weights."""

    @staticmethod
    def forward(ctx, instances, mlp_weights, origins, directions, targets, weights, u_coarse, u_fine, temperature, scalars, origin_stride,
                seed, stream_offset, flags, loss_scale, eikonal_ratio, want_labels, want_samples=False):
        lib = _lib.load()
        std, ratio, eps, near, far, S, schedule = _unpack(scalars)
        R, N = directions.shape[0], instances.shape[0]
        dev = directions.device
        instances = instances.detach().contiguous()
        centred = _centre_mlp(mlp_weights)
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        grad, grad_mlp = torch.empty_like(instances), torch.empty_like(centred)
        labels = torch.empty(R, N, dtype=torch.float32, device=dev) if want_labels else None
        samples = None
        if want_samples:          # the step's own state between its passes (vsrd_render_config::out_*; ABI 8: the residual step's one-ray-per-wave form too)
            samples = (torch.empty(R, 2 * S, dtype=torch.float32, device=dev), torch.empty(R, S - 1, dtype=torch.float32, device=dev),
                       torch.empty(R, S, dtype=torch.float32, device=dev), torch.empty(R, S, dtype=torch.float32, device=dev))
        workspace = _workspace(dev, N, True, step_shape=(S, R))
        field = _lib.make_field(instances, temperature, centred)
        config = _lib.make_config(R, S, (near, far), std, ratio, eps, origin_stride, seed, stream_offset, flags | _mlp_flag(centred), schedule=schedule, samples=samples)
        with profiling.timed("vsrd_render_residual_step"):
            _lib.check(lib.vsrd_render_residual_step(field, config, _lib.ptr(origins), _lib.ptr(directions), _lib.ptr(u_coarse), _lib.ptr(u_fine),
                                                     _lib.ptr(targets), _lib.ptr(weights), float(loss_scale), float(eikonal_ratio),
                                                     workspace.data_ptr(), workspace.numel(), _lib.ptr(losses), _lib.ptr(grad), _lib.ptr(grad_mlp),
                                                     _lib.ptr(labels), _lib.stream()))
        ctx.save_for_backward(grad, grad_mlp)
        out_labels = labels if want_labels else losses.new_empty(0)
        out_samples = samples if want_samples else tuple(losses.new_empty(0) for _ in range(4))
        terms = losses.clone()
        ctx.mark_non_differentiable(out_labels, terms, *out_samples)
        return (losses[0] + eikonal_ratio * losses[1], terms, out_labels) + tuple(out_samples)

    @staticmethod
    def backward(ctx, grad_loss, _grad_terms, _grad_labels, *_unused):
        grad, grad_mlp = ctx.saved_tensors
        return (grad * grad_loss, grad_mlp * grad_loss) + (None,) * 16


def silhouette_step(distance_field, ray_positions, ray_directions, targets, distance_range, num_samples, sdf_std_deviation,
                    cosine_ratio=1.0, epsilon=1.0e-6, pd_indices=None, gt_indices=None, u_coarse=None, u_fine=None, seed=0,
                    stream_offset=0, return_labels=False, skip_exact_misses=True, schedule=None, eikonal_ratio=0.0, return_terms=False,
                    return_samples=False, mlp_split_bf16=None):
    """Fused fast path of scripts/main.py:629-687: the two-pass render AND
    ``mean(BCE(clamp(labels[..., pd_indices], 1e-6, 1 - 1e-6), targets[..., gt_indices]))`` in one launch; for residual fields
    (box + per-instance MLP) also ``eikonal_ratio * mean((|sampled_gradients| - 1)^2)`` (main.py:679-687), i.e. the returned loss is
    ``silhouette + eikonal_ratio * eikonal``.  Returns the loss (autograd-connected to the field parameters), then -- when asked --
    the detached terms ``[silhouette, eikonal]``, the labels [R,N], and (``return_samples``) a dict with the step's OWN
    state between its passes: ``distances`` [R,2S] (sorted; NaN in column 0 = ray skipped as an exact miss), ``coarse_weights`` [R,S-1],
    ``u_coarse`` / ``u_fine`` [R,S] -- what its labels, loss and gradients were computed from (the full-size parity tests feed them to the
    oracle)."""
    block = flatten(distance_field)
    residual = block.mlp_weights is not None
    if eikonal_ratio and not residual:
        raise NotImplementedError("the eikonal term is fused for residual fields only (a box-only union is not optimised with it)")
    origins, directions, stride, _ = _prepare_rays(ray_positions, ray_directions)
    R, N = directions.shape[0], block.num_instances
    targets = targets.reshape(R, -1).to(torch.float32)
    if block.label_indices is not None:
        raise NotImplementedError("permuted instance labels are not supported by the fused path")
    if pd_indices is None:
        ordered, weights, kept = targets.contiguous(), None, N
    else:   # labels[..., pd] vs targets[..., gt]  ==  labels[:, n] vs ordered[:, n] for n in pd, weight 0 elsewhere
        ordered = torch.zeros(R, N, dtype=torch.float32, device=targets.device)
        ordered[:, pd_indices] = targets[:, gt_indices]
        weights = torch.zeros(N, dtype=torch.float32, device=targets.device)
        weights.index_fill_(0, pd_indices, 1.0)             # (an indexed scalar assignment would upload the scalar: not capturable)
        kept = int(pd_indices.numel())
    if (u_coarse is None) != (u_fine is None):
        raise ValueError("pass both u_coarse and u_fine, or neither (in-kernel Philox)")
    if u_coarse is not None:
        u_coarse = u_coarse.reshape(-1, num_samples).to(torch.float32).contiguous()
        u_fine = u_fine.reshape(-1, num_samples).to(torch.float32).contiguous()
    flags = (_lib.FLAG_SKIP_EXACT_MISSES if skip_exact_misses else 0) | _base_flags() | (_lib.FLAG_YAW_GRADIENTS if block.yaw_gradients else 0)
    if mlp_split_bf16 is not None:        # (residual fields: the MLP's products on split-bf16 MFMA or on the exact-fp32 one; None: the module switch)
        flags = (flags | _lib.FLAG_MLP_SPLIT_BF16) if mlp_split_bf16 else (flags & ~_lib.FLAG_MLP_SPLIT_BF16)
    scalars = (float(sdf_std_deviation), float(cosine_ratio), float(epsilon), float(distance_range[0]), float(distance_range[1]), int(num_samples), schedule)
    samples = None
    if residual:
        loss, terms, labels, *samples = _ResidualStep.apply(block.instances, block.mlp_weights, origins, directions, ordered, weights, u_coarse, u_fine,
                                                            block.temperature, scalars, stride, int(seed), _offset(stream_offset), flags & ~_lib.FLAG_SKIP_EXACT_MISSES,
                                                            1.0 / (R * max(kept, 1)), float(eikonal_ratio), bool(return_labels), bool(return_samples))
    else:
        loss, labels, *samples = _SilhouetteStep.apply(block.instances, origins, directions, ordered, weights, u_coarse, u_fine, block.temperature, scalars,
                                                       stride, int(seed), _offset(stream_offset), flags, 1.0 / (R * max(kept, 1)), bool(return_labels),
                                                       bool(return_samples))
        terms = torch.stack([loss.detach(), torch.zeros_like(loss.detach())])
    out = (loss,) + ((terms,) if return_terms else ()) + ((labels,) if return_labels else ())
    if return_samples:
        out = out + (dict(zip(("distances", "coarse_weights", "u_coarse", "u_fine"), samples)),)
    return out if len(out) > 1 else loss
