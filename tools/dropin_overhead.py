#!/usr/bin/env python3
"""Host overhead of the drop-in call surface: one main.py-shaped render per iteration -- the field closures are
rebuilt, flattened, rendered through hierarchical_wrapper(vsrd.rendering.hierarchical_volumetric_rendering) (two passes, four launches) and
back-propagated -- at the reference's own size (1000 rays x 100 samples, N = 8), box-only and residual.  GPU box; experiments only.

  python tools/dropin_overhead.py [--iterations 200]
"""
import argparse
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--iterations", type=int, default=200)
    args = parser.parse_args()
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
    for residual in (False, True):
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
        for _ in range(10):
            iteration()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iterations):
            iteration()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        print(f"{'residual' if residual else 'box-only'}: {total / args.iterations * 1e3:.2f} ms per main.py-shaped render + backward "
              f"(host side alone {host / args.iterations * 1e3:.2f} ms), {R} rays x {S} samples, N = {N}")
        # GPU time of the library calls of one iteration (HIP events around each C-ABI call), and of the fused step on the same rays
        from vsrd_amd import profiling, fields as vfields, rendering as vrendering
        from vsrd_amd.rendering import renderers
        for form in (("two kernels per chunk", False), ("one kernel (round 1)", True)) if residual else (("", False),):
            renderers.RESIDUAL_SINGLE_KERNEL = form[1]
            try:
                with profiling.kernel_timer() as timer:
                    for _ in range(20):
                        iteration()
                    torch.cuda.synchronize()
                calls = timer.summary()
            finally:
                renderers.RESIDUAL_SINGLE_KERNEL = False
            per_iteration = sum(n * ms for n, ms in calls.values()) / 20
            detail = ", ".join(f"{name} {n // 20} x {ms:.3f} ms" for name, (n, ms) in sorted(calls.items()))
            print(f"    GPU time of the library calls per iteration{' [backward: ' + form[0] + ']' if form[0] else ''}: {per_iteration:.3f} ms ({detail})")
        block = vfields.FieldBlock(vfields.pack_instances(leaves[0].detach(), leaves[2].detach(), leaves[1].detach()), 0.5, weights.detach() if residual else None, None)
        with profiling.kernel_timer() as timer:
            for k in range(20):
                vrendering.silhouette_step(block, origins, directions, targets, (0.0, 100.0), S, 0.5, 0.5, seed=1, stream_offset=k,
                                           eikonal_ratio=0.01 if residual else 0.0)
            torch.cuda.synchronize()
        fused = sum(n * ms for n, ms in timer.summary().values()) / 20
        print(f"    fused step on the same rays (vsrd_render_{'residual' if residual else 'silhouette'}_step): {fused:.3f} ms")


if __name__ == "__main__":
    main()
