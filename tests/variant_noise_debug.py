#!/usr/bin/env python3
"""How far two mappings of the fused step (four / two rays per wave against one ray per wave) differ over MANY random scenes, not just the
one a parity test pins -- each mapping runs its own importance sampler, which turns last-bit differences of pass-1 weights into displaced
fine samples on ill-conditioned rays, so the difference has a heavy tail.  Used in round 4 to compare the box norm as s * rsq(s) with
sqrt(s) + rcp (VSRD_HIP_LIBRARY selects the build).  GPU box; experiments only.
    python tests/variant_noise_debug.py [N S R [seeds]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_hip_render import _random_scene  # noqa: E402


def main():
    from vsrd_amd import fields, rendering
    from vsrd_amd.rendering import renderers
    N, S, R = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 128, 37)
    seeds = int(sys.argv[4]) if len(sys.argv) >= 5 else 12
    dev = torch.device("cuda:0")
    T, std, ratio = 0.4, 0.4, 0.4
    rows = []
    for seed in range(seeds):
        sc = _random_scene(1000 + seed, N, R, S, general_rotations=False)
        pd = torch.arange(0, N, 2, device=dev) if N >= 4 else None
        gt = torch.arange(pd.numel() - 1, -1, -1, device=dev) if pd is not None else None
        targets = sc["targets"][:, :pd.numel()].contiguous() if pd is not None else sc["targets"]
        results = {}
        for mode in ("quad", "wave"):
            renderers.STEP_WAVE_PER_RAY = mode == "wave"
            try:
                inst = fields.pack_instances(sc["loc"], sc["rot"], sc["dim"]).to(dev).requires_grad_(True)
                block = fields.FieldBlock(inst, T, None, None)
                loss, labels = rendering.silhouette_step(block, sc["origins"].to(dev), sc["directions"].to(dev), targets.to(dev), (0.0, 100.0), S, std, ratio,
                                                         pd_indices=pd, gt_indices=gt, seed=3, stream_offset=11, return_labels=True,
                                                         u_coarse=sc["u_coarse"].to(dev), u_fine=sc["u_fine"].to(dev))
                results[mode] = (labels.detach(), torch.autograd.grad(loss, inst)[0])
            finally:
                renderers.STEP_WAVE_PER_RAY = False
        label = float((results["quad"][0] - results["wave"][0]).abs().max())
        grad = float((results["quad"][1] - results["wave"][1]).abs().max()) / max(float(results["wave"][1].abs().max()), 1e-6)
        rows.append((label, grad))
        print(f"seed {seed:3d}: labels {label:9.2e}   gradients / largest {grad:9.2e}")
    labels, grads = sorted(r[0] for r in rows), sorted(r[1] for r in rows)
    print(f"{os.environ.get('VSRD_HIP_LIBRARY', 'default library')}: N={N} S={S} R={R}: labels median {labels[len(labels) // 2]:.2e} max {labels[-1]:.2e};  "
          f"gradients median {grads[len(grads) // 2]:.2e} max {grads[-1]:.2e}")


if __name__ == "__main__":
    main()
