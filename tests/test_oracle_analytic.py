"""The hand-derived adjoint (oracle/analytic.py, the blueprint of the HIP backward kernel) against
ordinary autograd through the closed-form oracle, in float64.  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import analytic, fields, rendering


@pytest.mark.parametrize("name,ratio", [("g4_render_n4_s32_mid", 0.5), ("g4_render_n4_s32_late", 1.0),
                                        ("g4_render_n3_s20_mid", 0.0), ("g4_render_n16_s64_mid", 0.3)])
def test_backward_ray_matches_autograd(name, ratio):
    g = load_golden(name)
    rng = np.random.default_rng(0)
    T, std = float(g["temperature"]), float(g["sdf_std_deviation"])
    fine = g["fine_distances"].t().double()
    conditioned = (g["coarse_weights"].sum(0) > 0).numpy()
    rays = [int(i) for i in np.flatnonzero(conditioned)[::7][:12]]
    assert len(rays) >= 3
    for ray in rays:
        loc = g["locations"].double().requires_grad_(True)
        rot = g["orientations"].double().requires_grad_(True)
        dim = g["dimensions"].double().requires_grad_(True)
        union = fields.InstanceUnion(loc, rot, dim, T)
        o, r, dist = g["origins"][ray].double(), g["directions"][ray].double(), fine[ray]
        out = rendering.render_given_distances(union, o[None], r[None], dist[None], std, ratio)
        lam = torch.from_numpy(rng.standard_normal(loc.shape[0]))
        gamma = torch.from_numpy(rng.standard_normal((dist.numel() - 1, 3)) * 0.1)
        omega = torch.from_numpy(rng.standard_normal(dist.numel() - 1) * 0.1)
        loss = (out.labels[0] * lam).sum() + (out.gradients[0] * gamma).sum() + (out.weights[0] * omega).sum()
        gt, gR, gd = torch.autograd.grad(loss, [loc, rot, dim])
        at, aR, ad, f = analytic.backward_ray(o.numpy(), r.numpy(), dist.numpy(), loc.detach().numpy(), rot.detach().numpy(),
                                             dim.detach().numpy(), T, std, ratio, lam.numpy(), gamma.numpy(), omega.numpy())
        np.testing.assert_allclose(f["labels"], out.labels[0].detach().numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(f["wgt"], out.weights[0].detach().numpy(), rtol=1e-9, atol=1e-12)
        for a, b in ((at, gt), (aR, gR), (ad, gd)):
            scale = max(float(b.abs().max()), 1e-9)
            np.testing.assert_allclose(a, b.numpy(), rtol=1e-7, atol=1e-9 * scale + 1e-12)
