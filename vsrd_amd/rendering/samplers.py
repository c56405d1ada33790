"""vsrd.rendering samplers (reference: vsrd/rendering/samplers.py:5-36) on the HIP library.

The renderer does not call these (sampling is fused into the render kernels); they exist so the
``vsrd.rendering`` call surface is complete and so the samplers can be tested in isolation.
"""
import torch

from .. import _lib


def quadrature_sampler(bins, deterministic=False):
    """Stratified samples inside consecutive bins [..., S+1] -> [..., S].  Bins must be the linspace the
    renderer uses (renderers.py:191-192): the kernel regenerates it from its two end points."""
    lib = _lib.load()
    S = bins.shape[-1] - 1
    flat = bins.reshape(-1, S + 1)
    near, far = float(flat[0, 0]), float(flat[0, -1])
    expected = torch.linspace(near, far, S + 1, device=bins.device)
    if not torch.equal(flat, expected.expand_as(flat)):
        raise NotImplementedError("quadrature_sampler: only the renderer's linspace bins are supported")
    u = torch.full(bins[..., :-1].shape, 0.5, device=bins.device) if deterministic else torch.rand_like(bins[..., :-1])
    u = u.reshape(-1, S).to(torch.float32).contiguous()
    out = torch.empty_like(u)
    config = _lib.make_config(u.shape[0], S, (near, far), 1.0, 1.0, 1.0e-6, 3)
    _lib.check(lib.vsrd_sample_stratified(config, _lib.ptr(u), _lib.ptr(out), _lib.stream()))
    return out.reshape(bins[..., :-1].shape)


def importance_merge(bins, weights, uniforms=None, sorted_uniforms=False):
    """cat(bins, inverse_transform_sampler(bins, weights, S)) sorted (renderers.py:198-210): [...,S] -> [...,2S]."""
    lib = _lib.load()
    S = bins.shape[-1]
    lead = bins.shape[:-1]
    b = bins.reshape(-1, S).to(torch.float32).contiguous()
    w = weights.reshape(-1, S - 1).to(torch.float32).contiguous()
    if uniforms is None:
        uniforms, sorted_uniforms = torch.sort(torch.rand(*lead, S, device=bins.device), dim=-1).values, True
    u = uniforms.reshape(-1, S).to(torch.float32).contiguous()
    out = torch.empty(b.shape[0], 2 * S, dtype=torch.float32, device=bins.device)
    config = _lib.make_config(b.shape[0], S, (0.0, 1.0), 1.0, 1.0, 1.0e-6, 3,
                              flags=_lib.FLAG_FINE_UNIFORMS_SORTED if sorted_uniforms else 0)
    _lib.check(lib.vsrd_sample_importance(config, _lib.ptr(b), _lib.ptr(w), _lib.ptr(u), _lib.ptr(out), None, _lib.stream()))
    return out.reshape(*lead, 2 * S)


def inverse_transform_sampler(bins, weights, num_samples, deterministic=False, uniforms=None):
    """Drop-in for vsrd.rendering.samplers.inverse_transform_sampler (samplers.py:11-36): ``num_samples`` (= bins.shape[-1], the
    only case the renderer uses, renderers.py:203) sorted samples of the piecewise-constant pdf ``weights`` [..., S-1] over the
    points ``bins`` [..., S].  ``deterministic`` takes linspace(0, 1, S) as the uniforms; ``uniforms`` [..., S] (sorted) may be given."""
    lib = _lib.load()
    S = bins.shape[-1]
    if num_samples != S:
        raise NotImplementedError("inverse_transform_sampler: num_samples must equal the number of bins (the renderer's case)")
    lead = bins.shape[:-1]
    b = bins.reshape(-1, S).to(torch.float32).contiguous()
    w = weights.reshape(-1, S - 1).to(torch.float32).contiguous()
    if uniforms is None:
        uniforms = torch.linspace(0.0, 1.0, S, device=bins.device).expand(*lead, S) if deterministic else \
            torch.sort(torch.rand(*lead, S, device=bins.device), dim=-1).values
    u = uniforms.reshape(-1, S).to(torch.float32).contiguous()
    fine = torch.empty_like(b)
    config = _lib.make_config(b.shape[0], S, (0.0, 1.0), 1.0, 1.0, 1.0e-6, 3, flags=_lib.FLAG_FINE_UNIFORMS_SORTED)
    _lib.check(lib.vsrd_sample_importance(config, _lib.ptr(b), _lib.ptr(w), _lib.ptr(u), None, _lib.ptr(fine), _lib.stream()))
    return fine.reshape(*lead, S)


def sample_rays(weights, num_samples, seed=0, stream_offset=0, out=None):
    """scripts/main.py:620-627: ``torch.multinomial(weights, num_samples, replacement=False)`` as a few streaming launches
    (vsrd_sample_rays: ATen's exponential-race algorithm with Philox keyed by (seed, stream_offset; index), no full sort).
    Deterministic in its arguments; ``stream_offset`` may be a device int64 tensor (read on the device: hipGraph replay).
    Returns int64 indices [num_samples], best key first (written into ``out`` when given)."""
    lib = _lib.load()
    weights = weights.detach().reshape(-1).to(torch.float32).contiguous()
    from .renderers import current_workspace
    buf = current_workspace().sampler(weights.device)       # owned by the caller's Workspace (renderers.py), freed with it
    indices = torch.empty(int(num_samples), dtype=torch.int64, device=weights.device) if out is None else out
    if indices.dtype != torch.int64 or indices.numel() != int(num_samples) or not indices.is_contiguous() or indices.device != weights.device:
        raise ValueError("out must be a contiguous int64 tensor of num_samples elements on the weights' device")
    offset_ptr = None
    if isinstance(stream_offset, torch.Tensor):
        offset_ptr, stream_offset = stream_offset.data_ptr(), 0
    _lib.check(lib.vsrd_sample_rays(_lib.ptr(weights), weights.numel(), int(num_samples), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                    int(stream_offset) & 0xFFFFFFFFFFFFFFFF, offset_ptr, buf.data_ptr(), buf.numel(), indices.data_ptr(), _lib.stream()))
    return indices
