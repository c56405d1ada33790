"""Oracle: per-instance signed-distance fields and their soft-min union (TEST INFRASTRUCTURE).

Closed-form restatement of
  * ``vsrd/rendering/sdfs.py:5-37``            (norm, box, translation, rotation)
  * ``scripts/main.py:433-492``                (residual field, residual composition,
                                                instance one-hot features, soft union)
  * ``vsrd/models/encoders/sinusoidal_encoder.py:8-19``
  * ``vsrd/models/fields/hyper_distance_field.py:57-73``
  * the ``torch.autograd.grad(sdf, positions)`` normal of
    ``vsrd/rendering/renderers.py:218-228`` -- here written out analytically, so that
    ordinary first-order autograd through this module reproduces the reference's
    double-backward.

The reference builds the field as nested Python closures; here a field is a flat
parameter block (locations ``[N,3]``, orientations ``[N,3,3]``, half-extents ``[N,3]``,
temperature, optional per-instance MLP weights ``[N,1617]``) -- the same block the HIP
library consumes.
"""
import math
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F

NORM_EPSILON = 1.0e-6          # sdfs.py:5  (inside the sqrt)
POSITION_SCALE = 100.0         # max(config.volume_rendering.distance_range), main.py:441
NUM_FREQUENCIES = 8            # configs/.../config.json:157
MLP_WIDTHS = (48, 16, 16, 16, 16, 1)   # config.json:147-154 -> hyper_distance_field.py:18-25
MLP_SPLITS = tuple((MLP_WIDTHS[k] + 1) * MLP_WIDTHS[k + 1] for k in range(5))  # 784,272,272,272,17
LAYER_NORM_EPSILON = 1.0e-5    # F.layer_norm default (hyper_distance_field.py:65)


def box_distance_and_gradient(local, half_extents):
    """Box SDF value (sdfs.py:9-19) and its gradient w.r.t. ``local``.

    local [...,3], half_extents broadcastable to it.  Returns (d [...], grad [...,3]).
    The gradient is what ``autograd.grad`` gives for the reference expression:
    ``sign(p) * (relu(q)/sqrt(sum relu(q)^2 + 1e-6) + [max(q) < 0] * onehot(argmax q))``.
    """
    q = local.abs() - half_extents
    outside = F.relu(q)
    radius = torch.sqrt((outside * outside).sum(-1, keepdim=True) + NORM_EPSILON)
    q_max, q_arg = q.max(dim=-1, keepdim=True)
    distance = radius - F.relu(-q_max)
    interior = F.one_hot(q_arg.squeeze(-1), 3).to(local.dtype) * (q_max < 0).to(local.dtype)
    gradient = torch.sign(local).detach() * (outside / radius + interior)
    return distance.squeeze(-1), gradient


def sinusoidal_features(x, tangents=None, num_frequencies=NUM_FREQUENCIES):
    """sinusoidal_encoder.py:12-18: channels ordered [coord][freq][cos, sin].

    tangents (optional) [...,T,3] are directional derivatives of x; returns their image
    [...,T,6*F] under the encoder Jacobian.
    """
    freqs = (2.0 ** torch.arange(num_frequencies, dtype=x.dtype, device=x.device)) * math.pi
    phase = x.unsqueeze(-1) * freqs                                   # [...,3,F]
    cos, sin = torch.cos(phase), torch.sin(phase)
    feats = torch.stack([cos, sin], dim=-1).flatten(-3, -1)          # [...,3*F*2]
    if tangents is None:
        return feats
    dphase = tangents.unsqueeze(-1) * freqs                           # [...,T,3,F]
    dfeats = torch.stack([-sin.unsqueeze(-3) * dphase, cos.unsqueeze(-3) * dphase], dim=-1).flatten(-3, -1)
    return feats, dfeats


def _layer_norm_gelu(x, dx):
    """F.layer_norm(no affine) then exact GELU, with forward-mode tangents dx [...,T,C]."""
    mean = x.mean(-1, keepdim=True)
    centered = x - mean
    inv_std = torch.rsqrt((centered * centered).mean(-1, keepdim=True) + LAYER_NORM_EPSILON)
    y = centered * inv_std
    cdf = 0.5 * (1.0 + torch.erf(y / math.sqrt(2.0)))
    out = y * cdf
    if dx is None:
        return out, None
    dcentered = dx - dx.mean(-1, keepdim=True)
    yb = y.unsqueeze(-2)
    dy = (dcentered - yb * (yb * dcentered).mean(-1, keepdim=True)) * inv_std.unsqueeze(-2)
    pdf = torch.exp(-0.5 * y * y) / math.sqrt(2.0 * math.pi)
    dout = dy * (cdf + y * pdf).unsqueeze(-2)
    return out, dout


def _instance_linear(block, fan_in, x, dx):
    """[x; 1] -> W [x; 1] for per-instance blocks W [..., out, in + 1] (hyper_distance_field.py:66-70).  When the blocks carry one
    leading instance axis and are broadcast over all the points of x ([N, 1, .., 1, out, in + 1] against [N, .., in]) the product
    is a batched GEMM over that axis; the element-wise form below would materialise an [N, points, out, in] tensor (9 GB per layer
    at BASELINE config 3 sizes on one image row -- 14x slower than the reference itself)."""
    matrix, bias = block[..., :fan_in], block[..., fan_in]
    batched = block.dim() >= 3 and all(d == 1 for d in block.shape[1:-2]) and x.dim() == block.dim() - 1 and x.shape[0] == block.shape[0]
    if block.dim() == 2:                                       # one weight vector for every point: a plain GEMM
        return x @ matrix.t() + bias, (None if dx is None else dx @ matrix.t())
    if batched:
        n, out = block.shape[0], block.shape[-2]
        w_t = matrix.reshape(n, out, fan_in).transpose(1, 2)
        y = torch.bmm(x.reshape(n, -1, fan_in), w_t).reshape(*x.shape[:-1], out) + bias.reshape(n, *([1] * (x.dim() - 2)), out)
        dy = None if dx is None else torch.bmm(dx.reshape(n, -1, fan_in), w_t).reshape(*dx.shape[:-1], out)
        return y, dy
    y = (matrix * x.unsqueeze(-2)).sum(-1) + bias
    dy = None if dx is None else (matrix.unsqueeze(-3) * dx.unsqueeze(-2)).sum(-1)
    return y, dy


def instance_mlp(weights, feats, dfeats=None):
    """hyper_distance_field.py:57-73 for one weight vector per leading index.

    weights [...,1617] broadcast against feats [...,48]; optional tangents dfeats [...,T,48].
    Returns out [...,] (and dout [...,T]).
    """
    x, dx = feats, dfeats
    offset = 0
    for layer, count in enumerate(MLP_SPLITS):
        fan_in, fan_out = MLP_WIDTHS[layer], MLP_WIDTHS[layer + 1]
        block = weights[..., offset:offset + count].unflatten(-1, (fan_out, fan_in + 1))
        offset += count
        if layer:
            x, dx = _layer_norm_gelu(x, dx)
        x, dx = _instance_linear(block, fan_in, x, dx)
    if dx is None:
        return x.squeeze(-1)
    return x.squeeze(-1), dx.squeeze(-1)


def residual_distance_and_gradient(local, mlp_weights):
    """main.py:433-449: sigmoid(MLP(encode((|x|,y,z)/100)) - 1) and its gradient w.r.t. local."""
    fold = torch.ones_like(local)
    fold[..., 0] = torch.sign(local[..., 0]).detach()
    folded = torch.stack([local[..., 0].abs(), local[..., 1], local[..., 2]], dim=-1) / POSITION_SCALE
    eye = torch.eye(3, dtype=local.dtype, device=local.device).expand(*local.shape[:-1], 3, 3)
    feats, dfeats = sinusoidal_features(folded, eye)
    out, dout = instance_mlp(mlp_weights, feats, dfeats)
    residual = torch.sigmoid(out - 1.0)
    gradient = (residual * (1.0 - residual)).unsqueeze(-1) * dout * fold / POSITION_SCALE
    return residual, gradient


@dataclass
class InstanceUnion:
    """Soft-min union of N oriented boxes (+ optional residual MLP), main.py:525-618."""
    locations: torch.Tensor            # [N,3]   translation(...)        sdfs.py:22-28
    orientations: torch.Tensor         # [N,3,3] rotation(...): p @ R    sdfs.py:31-37
    dimensions: torch.Tensor           # [N,3]   half extents            sdfs.py:9-19
    temperature: float                 # soft_union temperature          main.py:477-492
    mlp_weights: Optional[torch.Tensor] = None   # [N,1617]              main.py:527

    @property
    def num_instances(self):
        return self.locations.shape[0]

    def instance_terms(self, positions):
        """Per-instance distance [N,...] and world-space gradient [N,...,3]."""
        lead = (self.num_instances,) + (1,) * (positions.dim() - 1)
        rel = positions.unsqueeze(0) - self.locations.reshape(*lead, 3)
        R = self.orientations.reshape(*lead, 3, 3)
        local = (rel.unsqueeze(-2) @ R).squeeze(-2)                   # row vector times R
        d, g_local = box_distance_and_gradient(local, self.dimensions.reshape(*lead, 3))
        if self.mlp_weights is not None:
            if local[0].numel() > (1 << 16):
                # instance by instance, as the reference's closure loop does (main.py:480-483): the forward-mode tangents make the
                # temporaries 4x the reference's, and [N, points, 3, 16]-sized ones fall out of every cache and into mmap/munmap churn
                parts = [residual_distance_and_gradient(local[i], self.mlp_weights[i]) for i in range(self.num_instances)]
                res, g_res = torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts])
            else:
                res, g_res = residual_distance_and_gradient(local, self.mlp_weights.reshape(*lead, -1))
            d, g_local = d + res, g_local + g_res
        g_world = (g_local.unsqueeze(-2) @ R.transpose(-1, -2)).squeeze(-2)
        return d, g_world

    def evaluate(self, positions):
        """positions [...,3] -> union distance [...], labels [...,N], gradient [...,3].

        u = sum_i w_i d_i with w = softmin(d / T);  labels = w (one-hot features);
        grad u = sum_i w_i (1 - (d_i - u)/T) grad d_i.
        """
        d, g_world = self.instance_terms(positions)
        w = torch.softmax(-d / self.temperature, dim=0)
        u = (w * d).sum(0)
        coeff = w * (1.0 - (d - u) / self.temperature)
        gradient = (coeff.unsqueeze(-1) * g_world).sum(0)
        return u, w.movedim(0, -1), gradient

    def distance(self, positions):
        return self.evaluate(positions)[0]

    def hard_distance(self, positions):
        """main.py:494-509 hard union (arg-min instance)."""
        d, _ = self.instance_terms(positions)
        return d.min(dim=0).values
