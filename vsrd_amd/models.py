"""The optimised parameters of the hot path as torch modules (they stay on PyTorch: a handful of
element-wise ops on [N,3] tensors per step).

  BoxParameters3D      reference: vsrd/models/detectors/box_parameters.py:16-146
  SinusoidalEncoder    reference: vsrd/models/encoders/sinusoidal_encoder.py:8-19
  HyperDistanceField   reference: vsrd/models/fields/hyper_distance_field.py:8-77
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

_UNIT_CORNERS = [[-1, -1, 1], [1, -1, 1], [1, -1, -1], [-1, -1, -1], [-1, 1, 1], [1, 1, 1], [1, 1, -1], [-1, 1, -1]]


def yaw_matrix(cos, sin):
    zero, one = torch.zeros_like(cos), torch.ones_like(cos)
    rows = [torch.stack([cos, zero, sin], -1), torch.stack([zero, one, zero], -1), torch.stack([-sin, zero, cos], -1)]
    return torch.stack(rows, -2)


_corner_cache = {}


def _unit_corners(like):
    """The 8 unit corners on `like`'s device, uploaded once (a host-to-device copy per call would also break graph capture)."""
    key = (like.device, like.dtype)
    if key not in _corner_cache:
        _corner_cache[key] = torch.tensor(_UNIT_CORNERS, dtype=like.dtype, device=like.device)
    return _corner_cache[key]


class BoxParameters3D(nn.Module):
    """Raw per-instance parameters -> boxes.  Same parameter names, ranges and initial values as the reference."""

    def __init__(self, batch_size, num_instances, num_features=256,
                 location_range=((-50.0, 1.55 - 1.75 / 2.0 - 5.0, 0.0), (50.0, 1.55 - 1.75 / 2.0 + 5.0, 100.0)),
                 dimension_range=((0.75, 0.75, 1.5), (1.0, 1.0, 2.5))):
        super().__init__()
        self.locations = nn.Parameter(torch.zeros(batch_size, num_instances, 3))
        self.dimensions = nn.Parameter(torch.zeros(batch_size, num_instances, 3))
        self.orientations = nn.Parameter(torch.tensor([1.0, 0.0]).repeat(batch_size, num_instances, 1))
        self.embeddings = nn.Parameter(torch.rand(num_features).repeat(batch_size, num_instances, 1))
        self.register_buffer("location_range", torch.as_tensor(location_range, dtype=torch.float32))
        self.register_buffer("dimension_range", torch.as_tensor(dimension_range, dtype=torch.float32))

    def decode_location(self, raw):
        return torch.lerp(self.location_range[0], self.location_range[1], torch.sigmoid(raw))

    def decode_dimension(self, raw):
        return torch.lerp(self.dimension_range[0], self.dimension_range[1], torch.sigmoid(raw))

    def decode_orientation(self, raw):
        heading = F.normalize(raw, dim=-1)
        return yaw_matrix(heading[..., 0], heading[..., 1])

    @staticmethod
    def decode_box_3d(locations, dimensions, orientations):
        corners = _unit_corners(dimensions) * dimensions.unsqueeze(-2)
        return corners @ orientations.transpose(-2, -1) + locations.unsqueeze(-2)

    @staticmethod
    def encode_box_3d(boxes_3d):
        def mean_edge(a, b):
            return (boxes_3d[..., a, :] - boxes_3d[..., b, :]).norm(dim=-1).mean(-1)
        locations = boxes_3d.mean(-2)
        half = torch.stack([mean_edge([1, 2, 6, 5], [0, 3, 7, 4]), mean_edge([4, 5, 6, 7], [0, 1, 2, 3]),
                            mean_edge([1, 0, 4, 5], [2, 3, 7, 6])], -1) / 2.0
        forward = (boxes_3d[..., [1, 0, 4, 5], :] - boxes_3d[..., [2, 3, 7, 6], :]).mean(-2)
        heading = F.normalize(forward[..., [2, 0]], dim=-1)
        return locations, half, yaw_matrix(heading[..., 0], heading[..., 1])

    def forward(self):
        locations = self.decode_location(self.locations)
        dimensions = self.decode_dimension(self.dimensions)
        orientations = self.decode_orientation(self.orientations)
        return dict(boxes_3d=self.decode_box_3d(locations, dimensions, orientations), locations=locations,
                    dimensions=dimensions, orientations=orientations, embeddings=self.embeddings)


class SinusoidalEncoder(nn.Module):
    def __init__(self, num_frequencies):
        super().__init__()
        self.register_buffer("frequencies", 2.0 ** torch.arange(num_frequencies) * math.pi)

    def forward(self, inputs):
        phase = self.frequencies * inputs.unsqueeze(-1)
        return torch.stack([torch.cos(phase), torch.sin(phase)], dim=-1).flatten(-3, -1)


class HyperDistanceField(nn.Module):
    """Hypernetwork embeddings [.,256] -> per-instance MLP weights [.,1617].  A plain torch module (eager paths: rocBLAS GEMMs); the graph loop
    of optimization.FrameOptimizer runs its forward, backward and Adam through csrc/hypernetwork.h on this module's own parameter tensors."""

    def __init__(self, in_channels, out_channels_list, hyper_in_channels, hyper_out_channels_list):
        super().__init__()
        fan_in = [in_channels, *out_channels_list]
        fan_out = [*out_channels_list, 1]
        self.num_neurons_list = [(i + 1) * o for i, o in zip(fan_in, fan_out)]
        self.in_channels_list, self.out_channels_list = fan_in, fan_out
        widths = [hyper_in_channels, *hyper_out_channels_list]
        blocks = [nn.Sequential(nn.utils.weight_norm(nn.Linear(a, b)), nn.LayerNorm(b), nn.GELU())
                  for a, b in zip(widths[:-1], widths[1:])]
        blocks.append(nn.Sequential(nn.utils.weight_norm(nn.Linear(widths[-1], sum(self.num_neurons_list)))))
        self.hypernetwork = nn.Sequential(*blocks)

    def distance_field(self, weights, positions):
        """hyper_distance_field.py:57-73: the per-instance MLP on encoded positions, ``weights [...,1617]`` against
        ``positions [...,48]`` -> ``[...,1]``.  Layer l > 0 is preceded by LayerNorm (no affine) and exact GELU; every linear is
        ``W [out, in + 1]`` applied to ``[x; 1]``.

        scripts/main.py:541 binds this method with ``functools.partial(..., weights)`` inside its field closures; the renderer
        never calls it -- ``fields.flatten`` recovers ``weights`` from the partial and the HIP kernels (csrc/residual.h) evaluate
        the MLP together with its input Jacobian.  The method itself is plain torch on whatever device its tensors live on,
        like the rest of this module."""
        features = positions
        blocks = torch.split(weights, self.num_neurons_list, dim=-1)
        for layer, (block, fan_in, fan_out) in enumerate(zip(blocks, self.in_channels_list, self.out_channels_list)):
            if layer:
                features = F.gelu(F.layer_norm(features, [fan_in]))
            block = block.unflatten(-1, (fan_out, fan_in + 1))
            features = (block[..., :fan_in] @ features.unsqueeze(-1)).squeeze(-1) + block[..., fan_in]
        return features

    def forward(self, embeddings):
        return self.hypernetwork(embeddings)
