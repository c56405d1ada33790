"""Oracle: sampling, SDF -> opacity conversion and compositing (TEST INFRASTRUCTURE).

Restates, with explicit randomness and ray-major ``[R, S']`` tensors,
  * ``vsrd/rendering/samplers.py:5-8``    stratified ("quadrature") sampling
  * ``vsrd/rendering/samplers.py:11-36``  inverse-transform (importance) sampling
  * ``vsrd/rendering/renderers.py:177-270`` hierarchical_volumetric_rendering
  * ``scripts/main.py:511-523``           the two-pass wrapper (pass 1 without grad)

The reference draws its uniforms from the global torch generator inside the samplers;
here they are arguments (``u_coarse`` ~ ``rand_like(bins[..., :-1])``, ``u_fine`` ~
``rand(R, 1, S)`` *before* the sort of samplers.py:22), which is what makes the path
testable bit-for-bit against recorded draws.
"""
from typing import NamedTuple, Optional

import torch
import torch.nn.functional as F

SAMPLER_EPSILON = 1.0e-6    # samplers.py:33
NORMALIZE_EPSILON = 1.0e-12  # F.normalize default (samplers.py:13, renderers.py:228)


class RenderOutput(NamedTuple):
    labels: torch.Tensor      # [R, N]      accumulated instance features
    gradients: torch.Tensor   # [R, S', 3]  union-SDF gradients at the sample mid-points
    distances: torch.Tensor   # [R, D]      sorted sample distances (D = S or 2S)
    weights: torch.Tensor     # [R, S']     compositing weights (S' = D - 1)


def to_reference_layout(out: RenderOutput):
    """(labels [R,N], gradients [S',R,3], distances [D,R,1], weights [S',R,1]) as renderers.py:265-270."""
    return (out.labels, out.gradients.transpose(0, 1), out.distances.t().unsqueeze(-1), out.weights.t().unsqueeze(-1))


def stratified_distances(distance_range, num_samples, u_coarse):
    """renderers.py:191-194 + samplers.py:5-8.  u_coarse [R,S] -> distances [R,S]."""
    bins = torch.linspace(distance_range[0], distance_range[1], num_samples + 1,
                          dtype=u_coarse.dtype, device=u_coarse.device)
    return torch.lerp(bins[:-1].expand_as(u_coarse), bins[1:].expand_as(u_coarse), u_coarse)


def importance_distances(bins, weights, u_sorted):
    """samplers.py:11-36.  bins [R,S], weights [R,S-1], sorted uniforms [R,M] -> samples [R,M]."""
    total = weights.abs().sum(-1, keepdim=True).clamp_min(NORMALIZE_EPSILON)
    cdf = F.pad(torch.cumsum(weights / total, dim=-1), (1, 0))            # [R,S], cdf[:,0] = 0
    upper = torch.searchsorted(cdf, u_sorted.contiguous(), right=False).clamp(1, cdf.shape[-1] - 1)
    lower = upper - 1
    cdf_lo, cdf_hi = cdf.gather(-1, lower), cdf.gather(-1, upper)
    bin_lo, bin_hi = bins.gather(-1, lower), bins.gather(-1, upper)
    t = (u_sorted - cdf_lo) / (cdf_hi - cdf_lo + SAMPLER_EPSILON)
    return torch.lerp(bin_lo, bin_hi, t)


def merged_distances(coarse_distances, coarse_weights, u_fine):
    """renderers.py:198-210: coarse ∪ importance samples, sorted.  -> [R,2S]."""
    u_sorted = torch.sort(u_fine, dim=-1).values                           # samplers.py:22
    fine = importance_distances(coarse_distances, coarse_weights, u_sorted)
    return torch.sort(torch.cat([coarse_distances, fine], dim=-1), dim=-1).values


def section_opacities(u, gradient, directions, intervals, sdf_std_deviation, cosine_ratio, epsilon):
    """renderers.py:228-248 (NeuS section-point opacities) for u [R,S'], gradient [R,S',3]."""
    normals = F.normalize(gradient, dim=-1)
    cosines = (directions.unsqueeze(1) * normals).sum(-1)
    cosines = -torch.lerp(F.relu(-cosines * 0.5 + 0.5), F.relu(-cosines), cosine_ratio)
    half = cosines * intervals / 2.0
    cdf_prev = torch.sigmoid((u - half) / sdf_std_deviation)
    cdf_next = torch.sigmoid((u + half) / sdf_std_deviation)
    return F.relu((cdf_prev - cdf_next) / (cdf_prev + epsilon))


def composite(opacities):
    """renderers.py:250-258: w_s = alpha_s * prod_{j<s} (1 - alpha_j)."""
    transmittance = torch.cumprod(1.0 - opacities, dim=-1)
    transmittance = torch.cat([torch.ones_like(transmittance[:, :1]), transmittance[:, :-1]], dim=-1)
    return transmittance * opacities


def render_given_distances(field, origins, directions, distances, sdf_std_deviation,
                           cosine_ratio=1.0, epsilon=1.0e-6) -> RenderOutput:
    """renderers.py:212-270 for already-sorted distances [R,D] (ray-major)."""
    if origins.dim() == 1:
        origins = origins.expand_as(directions)
    intervals = distances[:, 1:] - distances[:, :-1]
    midpoints = (distances[:, :-1] + distances[:, 1:]) / 2.0
    positions = origins.unsqueeze(1) + directions.unsqueeze(1) * midpoints.unsqueeze(-1)
    u, labels, gradient = field.evaluate(positions)
    opacities = section_opacities(u, gradient, directions, intervals, sdf_std_deviation, cosine_ratio, epsilon)
    weights = composite(opacities)
    accumulated = (labels * weights.unsqueeze(-1)).sum(1)
    return RenderOutput(accumulated, gradient, distances, weights)


def hierarchical_render(field, origins, directions, distance_range, num_samples, sdf_std_deviation,
                        cosine_ratio, u_coarse, u_fine, epsilon=1.0e-6, return_coarse=False):
    """main.py:511-523 around renderers.py:177-270: coarse pass without grad, fine pass with."""
    with torch.no_grad():
        coarse = render_given_distances(field, origins, directions,
                                        stratified_distances(distance_range, num_samples, u_coarse),
                                        sdf_std_deviation, cosine_ratio, epsilon)
        merged = merged_distances(coarse.distances, coarse.weights, u_fine)
    fine = render_given_distances(field, origins, directions, merged, sdf_std_deviation, cosine_ratio, epsilon)
    return (coarse, fine) if return_coarse else fine
