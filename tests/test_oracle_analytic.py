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


def test_residual_mlp_jet_adjoint_matches_autograd():
    """oracle/analytic_mlp.py (blueprint of the residual-MLP kernels) against autograd through the closed-form oracle."""
    from oracle import analytic_mlp
    rng = np.random.default_rng(3)
    for trial in range(6):
        p = rng.standard_normal(3) * np.array([2.0, 1.0, 3.0])
        w = rng.standard_normal(1617) * 0.3
        res_bar, gres_bar = rng.standard_normal(), rng.standard_normal(3)
        pt = torch.tensor(p, dtype=torch.float64, requires_grad=True)
        wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
        res, gres = fields.residual_distance_and_gradient(pt[None], wt[None])
        a_res, a_gres, _ = analytic_mlp.forward(p, w)
        np.testing.assert_allclose(a_res, res.item(), rtol=1e-12)
        np.testing.assert_allclose(a_gres, gres[0].detach().numpy(), rtol=1e-10, atol=1e-14)
        loss = res[0] * res_bar + (gres[0] * torch.tensor(gres_bar)).sum()
        gp, gw = torch.autograd.grad(loss, [pt, wt])
        a_p, a_w = analytic_mlp.backward(p, w, res_bar, gres_bar)
        np.testing.assert_allclose(a_p, gp.numpy(), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(a_w, gw.numpy(), rtol=1e-8, atol=1e-12 * max(1.0, float(gw.abs().max())))
        # the single-tangent form of the adjoint (what residual.h runs) and the value + reverse-column form of the forward
        d_p, d_w = analytic_mlp.backward_directional(p, w, res_bar, gres_bar)
        np.testing.assert_allclose(d_p, gp.numpy(), rtol=1e-8, atol=1e-12)
        np.testing.assert_allclose(d_w, gw.numpy(), rtol=1e-8, atol=1e-12 * max(1.0, float(gw.abs().max())))
        r_res, r_gres = analytic_mlp.forward_reverse(p, w)
        np.testing.assert_allclose(r_res, res.item(), rtol=1e-12)
        np.testing.assert_allclose(r_gres, gres[0].detach().numpy(), rtol=1e-9, atol=1e-14)
