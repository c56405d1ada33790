#!/usr/bin/env python3
"""cProfile of the host side of one main.py-shaped render + backward through the drop-in call surface (tools/dropin_overhead.py's
iteration, box-only): where the ~1.4 ms of Python per iteration go.  GPU box; experiments only."""
import cProfile
import os
import pstats
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import __graft_entry__
    __graft_entry__.build()
    import vsrd_amd
    vsrd_amd.install_as_vsrd()
    import vsrd
    from test_hip_dropin import train_like_fields
    dev = torch.device("cuda:0")
    N, S, R = 8, 100, 1000
    g = torch.Generator().manual_seed(0)
    loc = torch.stack([torch.empty(N).uniform_(-8, 8, generator=g), torch.empty(N).uniform_(0.5, 1.5, generator=g), torch.empty(N).uniform_(8, 40, generator=g)], -1)
    dim = torch.stack([torch.empty(N).uniform_(0.75, 1.0, generator=g), torch.empty(N).uniform_(0.75, 1.0, generator=g), torch.empty(N).uniform_(1.5, 2.5, generator=g)], -1)
    yaw = torch.empty(N).uniform_(-3, 3, generator=g)
    rot = torch.stack([torch.stack([torch.cos(yaw), torch.zeros(N), torch.sin(yaw)], -1), torch.tensor([0.0, 1.0, 0.0]).expand(N, 3),
                       torch.stack([-torch.sin(yaw), torch.zeros(N), torch.cos(yaw)], -1)], -2)
    target_points = loc[torch.randint(0, N, (R,), generator=g)] + torch.randn(R, 3, generator=g) * 0.5
    directions = torch.nn.functional.normalize(target_points, dim=-1).to(dev)
    origins = torch.zeros(R, 3, device=dev)
    targets = torch.rand(R, N, generator=g).to(dev)
    config = types.SimpleNamespace(volume_rendering=types.SimpleNamespace(distance_range=[0.0, 100.0]))
    models = types.SimpleNamespace(positional_encoder=vsrd.models.SinusoidalEncoder(num_frequencies=8).to(dev),
                                   hyper_distance_field=vsrd.models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).to(dev))
    residual = len(sys.argv) > 1 and sys.argv[1] == "residual"
    leaves = [t.clone().to(dev).requires_grad_(True) for t in (loc, dim, rot)]
    weights = (torch.randn(N, 1617, generator=g) * 0.3).to(dev).requires_grad_(True)
    world = types.SimpleNamespace(locations=leaves[0][None], dimensions=leaves[1][None], orientations=leaves[2][None], distance_field_weights=weights[None])

    def iteration():
        fields, wrapper = train_like_fields(vsrd, config, models, world, N, 0.5, residual)
        labels, gradients = wrapper(vsrd.rendering.hierarchical_volumetric_rendering)(
            distance_field=fields[0], ray_positions=origins, ray_directions=directions, distance_range=(0.0, 100.0), num_samples=S,
            sdf_std_deviation=0.5, cosine_ratio=0.5)
        loss = torch.nn.functional.binary_cross_entropy(labels.clamp(1e-6, 1 - 1e-6), targets)
        loss.backward()
        return loss
    for _ in range(20):
        iteration()
    torch.cuda.synchronize()
    profile = cProfile.Profile()
    profile.enable()
    for _ in range(200):
        iteration()
    profile.disable()
    torch.cuda.synchronize()
    stats = pstats.Stats(profile)
    stats.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
