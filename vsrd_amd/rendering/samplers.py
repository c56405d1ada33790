"""vsrd.rendering samplers (reference: vsrd/rendering/samplers.py:5-36) on the HIP library.

The renderer does not call these (sampling is fused into the render kernels); they exist so the
``vsrd.rendering`` call surface is complete and so the samplers can be tested in isolation.
"""
import torch

from .. import _lib
from . import generic


def quadrature_sampler(bins, deterministic=False):
    """Stratified samples inside consecutive bins [..., S+1] -> [..., S] (samplers.py:5-8).  The renderer's own bins -- one linspace for
    every ray (renderers.py:191-192), recognised by being an expanded 1-D tensor -- go through vsrd_sample_stratified, which regenerates
    the linspace from its two end points; any other bins take the element-wise torch form on the device (generic.py)."""
    lib = _lib.load()
    S = bins.shape[-1] - 1
    shared = bins.dim() >= 1 and all(stride == 0 or size == 1 for stride, size in zip(bins.stride()[:-1], bins.shape[:-1]))
    first = bins.reshape(-1, S + 1)[0] if shared else None
    near, far = (float(first[0]), float(first[-1])) if shared else (0.0, 0.0)                     # (one host read of two numbers)
    if not shared or S < 1 or S > 256 or not torch.equal(first, torch.linspace(near, far, S + 1, device=bins.device, dtype=bins.dtype)):
        return generic.stratified_samples(bins, deterministic)
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
    """Drop-in for vsrd.rendering.samplers.inverse_transform_sampler (samplers.py:11-36): sorted samples of the piecewise-constant pdf
    ``weights`` [..., S-1] over the points ``bins`` [..., S].  ``num_samples`` = S (the renderer's case, renderers.py:203) runs
    vsrd_sample_importance; any other count takes the torch form on the device.  ``deterministic`` takes linspace(0, 1, num_samples) as
    the uniforms; ``uniforms`` [..., num_samples] (sorted) may be given."""
    lib = _lib.load()
    S = bins.shape[-1]
    if num_samples != S or S > 256:          # not the renderer's case: the torch form on the device (generic.py)
        return generic.inverse_transform_samples(bins, weights, num_samples, deterministic, uniforms)
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


class RayTable:
    """``torch.multinomial(weights, k, replacement=False)`` for weights that stay fixed over many draws (a frame's importance weights,
    scripts/main.py:204-265, 620-627): the table is built once (vsrd_ray_table_build) and every draw is one launch
    (vsrd_sample_rays_table) instead of the five of ``sample_rays``.  Same distribution over ordered samples (successive sampling
    without replacement); the sequences for a given seed differ from ``sample_rays``'s."""

    MAX_PICKS = 32768          # csrc/ray_sampling.h: kTableRounds x 2048 candidates per draw

    def __init__(self, weights, allocate=None):
        """``allocate(nbytes) -> uint8 device tensor``: where the table lives instead of a ``torch.empty`` of its own (a frame of a batch:
        its row of the batch's arena, optimization.FrameArena)."""
        lib = _lib.load()
        self.weights = weights.detach().reshape(-1).to(torch.float32).contiguous()
        if not self.weights.is_cuda:
            raise _lib.VsrdHipError("RayTable needs device weights (there is no CPU path)")
        self.count = int(self.weights.numel())
        nbytes = lib.vsrd_ray_table_bytes(self.count)
        self.table = torch.empty(nbytes, dtype=torch.uint8, device=self.weights.device) if allocate is None else allocate(nbytes)
        _lib.check(lib.vsrd_ray_table_build(_lib.ptr(self.weights), self.count, self.table.data_ptr(), self.table.numel(), _lib.stream()))

    def rebuild(self, weights):
        """The table of OTHER weights of the same count, in place (same buffer, same address: a captured graph that draws from this table
        draws from the new weights).  The build clears the header, the sticky `incomplete` flag included."""
        weights = weights.detach().reshape(-1).to(torch.float32).contiguous()
        if int(weights.numel()) != self.count or weights.device != self.weights.device:
            raise ValueError(f"RayTable.rebuild: {self.count} weights on {self.weights.device} expected")
        self.weights = weights
        _lib.check(_lib.load().vsrd_ray_table_build(_lib.ptr(self.weights), self.count, self.table.data_ptr(), self.table.numel(), _lib.stream()))

    def suits(self, num_samples, margin=4.0):
        """Whether ``num_samples`` distinct picks arrive well inside the kernel's budget: skipping repeats needs about
        num_samples / (share of the weight outside the num_samples - 1 heaviest entries) picks.  One host synchronisation: call it once."""
        positive = self.weights.clamp_min(0).double()
        total = float(positive.sum())
        if total <= 0.0 or num_samples < 1:
            return False
        heaviest = float(torch.topk(positive, min(max(int(num_samples) - 1, 1), self.count)).values.sum()) if num_samples > 1 else 0.0
        outside = max(1.0 - heaviest / total, 0.0)
        return outside > 0.0 and margin * num_samples / outside <= self.MAX_PICKS

    def sample(self, num_samples, seed=0, stream_offset=0, out=None, remap=None):
        """int64 indices [num_samples] in the order they were drawn (``remap[index]`` when a remap table is given); ``stream_offset`` may
        be a device int64 tensor (hipGraph replay).  -1 fills the tail when fewer than num_samples weights are positive."""
        lib = _lib.load()
        device = self.weights.device
        indices = torch.empty(int(num_samples), dtype=torch.int64, device=device) if out is None else out
        if indices.dtype != torch.int64 or indices.numel() != int(num_samples) or not indices.is_contiguous() or indices.device != device:
            raise ValueError("out must be a contiguous int64 tensor of num_samples elements on the weights' device")
        if remap is not None and (remap.dtype != torch.int64 or remap.numel() != self.count or not remap.is_contiguous() or remap.device != device):
            raise ValueError("remap must be a contiguous int64 tensor with one entry per weight on the weights' device")
        offset_ptr = None
        if isinstance(stream_offset, torch.Tensor):
            offset_ptr, stream_offset = stream_offset.data_ptr(), 0
        _lib.check(lib.vsrd_sample_rays_table(self.table.data_ptr(), self.count, int(num_samples), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                              int(stream_offset) & 0xFFFFFFFFFFFFFFFF, offset_ptr, None if remap is None else remap.data_ptr(),
                                              indices.data_ptr(), _lib.stream()))
        return indices

    def incomplete(self):
        """True when some draw ran out of picks and filled its tail with repeats (sticky; a host synchronisation)."""
        return bool(self.table[28:32].view(torch.int32).item() != 0)
