"""Parity of the HIP render path (through the C ABI) against the CPU oracle and the committed golden
vectors.  Needs a real MI355X: run with ``-m gpu``."""
import numpy as np
import pytest
import torch

from conftest import load_golden, margin, RENDER_CASES, RESIDUAL_CASES
from oracle import fields as ofields, rendering as orendering, geometry as ogeometry, losses as olosses

pytestmark = pytest.mark.gpu

LABEL_TOL = 1.0e-4   # BASELINE.json north_star: silhouettes within 1e-4 max-abs of the reference
GRAD_TOL = 5.0e-3    # parameter gradients, relative to the largest entry of each tensor


def assert_sampled_distances_close(got, want, num_samples):
    """Importance samples divide by (delta-cdf + 1e-6) (samplers.py:33): where delta-cdf is tiny, fp32 rounding of
    the coarse weights moves a sample by a visible fraction of a bin in ANY two fp32 implementations.  Almost all
    samples must agree to 5e-3 m; the rare ill-conditioned ones must stay within 2 % of a coarse bin."""
    diff = (got - want).abs()
    loose = diff > 5e-3 + 1e-4 * want.abs()
    assert loose.float().mean() <= 2e-3, f"{int(loose.sum())} of {loose.numel()} samples off by more than 5e-3"
    assert diff.max() <= 0.02 * 100.0 / num_samples, f"largest sample displacement {float(diff.max()):.4f} m"


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    import __graft_entry__
    __graft_entry__.build()
    return torch.device("cuda:0")


def hip_union(g, dev, requires_grad=False, temperature=None):
    from vsrd_amd import fields, rendering
    loc = g["locations"].to(dev).requires_grad_(requires_grad)
    dim = g["dimensions"].to(dev).requires_grad_(requires_grad)
    rot = g["orientations"].to(dev).requires_grad_(requires_grad)
    N = loc.shape[0]
    T = float(g["temperature"]) if temperature is None else temperature
    union = fields.soft_union([
        rendering.sdfs.translation(rendering.sdfs.rotation(fields.instance_field(rendering.sdfs.box(dim[i]), i, N), rot[i]), loc[i])
        for i in range(N)], T)
    return union, (loc, dim, rot)


def cpu_union(g, requires_grad=False, dtype=torch.float32):
    loc = g["locations"].to(dtype).clone().requires_grad_(requires_grad)
    dim = g["dimensions"].to(dtype).clone().requires_grad_(requires_grad)
    rot = g["orientations"].to(dtype).clone().requires_grad_(requires_grad)
    return ofields.InstanceUnion(loc, rot, dim, float(g["temperature"])), (loc, dim, rot)


def test_wave_primitives(dev):
    import ctypes
    from vsrd_amd import _lib
    lib = _lib.load()
    fn = lib.vsrd_selftest_wave
    fn.restype, fn.argtypes = ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    v = torch.randn(64, generator=torch.Generator().manual_seed(1))
    out = torch.zeros(576, device=dev)
    _lib.check(fn(_lib.ptr(v.to(dev)), _lib.ptr(out), _lib.stream()))
    out = out.cpu()
    v64 = v.double()
    torch.testing.assert_close(out[0:64], v64.sum().float().expand(64), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out[64:128], torch.cumsum(v64, 0).float(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out[128:192], torch.cumprod(1 + 0.01 * v64, 0).float(), rtol=1e-5, atol=1e-6)
    assert torch.equal(out[192:256], v.max().expand(64))
    assert torch.equal(out[256:320], v.flip(0))
    assert torch.equal(out[320:384], torch.cat([torch.tensor([-7.0]), v[:-1]]))
    u = out[384:448]
    assert torch.all((u >= 0) & (u < 1)) and u.unique().numel() > 60
    assert torch.equal(out[448:512], torch.linspace(0.0, 100.0, 65)[:64])
    j = torch.arange(64) % 16
    torch.testing.assert_close(out[512:576], (v64.sum() * (j + 1) + 64.0 * j).float(), rtol=1e-5, atol=1e-4)


def test_ray_casting_g1(dev):
    from vsrd_amd import rendering
    g = load_golden("g1_ray_casting")
    for tag in ("small", "mid"):
        h, w = (int(v) for v in g[f"{tag}_hw"])
        cam, dirs = rendering.ray_casting((h, w), g[f"{tag}_K"].to(dev), g[f"{tag}_E"].to(dev))
        torch.testing.assert_close(cam.cpu(), g[f"{tag}_camera_positions"], rtol=0, atol=1e-6)
        torch.testing.assert_close(dirs.cpu(), g[f"{tag}_ray_directions"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("temperature,tag", [(1.0, "T1"), (0.1, "T0p1")])
def test_field_eval_g3(dev, temperature, tag):
    from vsrd_amd import rendering
    g = load_golden("g2_g3_sdf_union")
    union, _ = hip_union(g, dev, temperature=temperature)
    u, w, grad = rendering.evaluate_field(union, g["points"].to(dev), with_gradients=True)
    torch.testing.assert_close(u.cpu(), g[f"union_{tag}_distances"], rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(w.cpu(), g[f"union_{tag}_labels"], rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(grad.cpu(), g[f"union_{tag}_gradients"], rtol=1e-4, atol=5e-5)
    # a single translated/rotated box called directly, and the hard union
    from vsrd_amd import fields
    single = union.distance_fields[2]      # a labelled member returns (distances, one-hot feature) like main.py's instance_field
    d, feature = rendering.evaluate_field(single, g["points"].to(dev))
    torch.testing.assert_close(d.cpu(), g["instance_distances"][2], rtol=1e-6, atol=3e-6)
    assert feature.shape == (g["points"].shape[0], 4) and feature.dtype == torch.int64 and bool((feature == torch.tensor([0, 0, 1, 0], device=dev)).all())
    bare = rendering.evaluate_field(single.sdf.sdf.distance_field, g["points"].to(dev))      # sdfs.box alone: distances only
    assert isinstance(bare, torch.Tensor) and bare.shape == (g["points"].shape[0], 1)
    hard = rendering.evaluate_field(fields.hard_union(union.distance_fields), g["points"].to(dev))
    torch.testing.assert_close(hard.cpu(), g["instance_distances"].min(0).values, rtol=1e-6, atol=3e-6)


@pytest.mark.parametrize("name", RENDER_CASES)
def test_render_at_golden_distances(dev, name):
    """Entry (i) of SURVEY §8c: given the reference's own sorted distances -> labels / gradients / weights."""
    from vsrd_amd import rendering
    g = load_golden(name)
    union, _ = hip_union(g, dev)
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    for prefix in ("coarse", "fine"):
        dist = g[f"{prefix}_distances"].t().contiguous().to(dev)
        labels, grads, weights = rendering.render_at_distances(union, g["origins"].to(dev), g["directions"].to(dev), dist, std, ratio)
        assert (labels.cpu() - g[f"{prefix}_labels"]).abs().max() < LABEL_TOL
        assert (weights.cpu() - g[f"{prefix}_weights"].t()).abs().max() < LABEL_TOL
        conditioned = g["coarse_weights"].sum(0) > 0 if prefix == "fine" else torch.ones(dist.shape[0], dtype=torch.bool)
        torch.testing.assert_close(grads.cpu()[conditioned], g[f"{prefix}_gradients"].transpose(0, 1)[conditioned], rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("name", RENDER_CASES)
def test_importance_merge_golden(dev, name):
    from vsrd_amd import rendering
    g = load_golden(name)
    merged = rendering.importance_merge(g["coarse_distances"].t().contiguous().to(dev), g["coarse_weights"].t().contiguous().to(dev),
                                        uniforms=g["u_fine"].to(dev), sorted_uniforms=False).cpu()
    ref = g["fine_distances"].t()
    assert torch.all(merged[:, 1:] >= merged[:, :-1])
    miss = g["coarse_weights"].sum(0) == 0
    assert_sampled_distances_close(merged[~miss], ref[~miss], int(g["num_samples"]))
    torch.testing.assert_close(merged[miss], ref[miss], rtol=1e-5, atol=1e-3)
    # pre-sorted uniforms give the same result
    merged2 = rendering.importance_merge(g["coarse_distances"].t().contiguous().to(dev), g["coarse_weights"].t().contiguous().to(dev),
                                         uniforms=torch.sort(g["u_fine"], -1).values.to(dev), sorted_uniforms=True).cpu()
    assert torch.equal(merged, merged2)
    # a6 by its own name: inverse_transform_sampler returns the fine samples alone; merged = sort(cat(coarse, fine))
    hit = ~miss
    sorted_u = torch.sort(g["u_fine"], -1).values
    fine = rendering.inverse_transform_sampler(g["coarse_distances"].t().contiguous().to(dev), g["coarse_weights"].t().contiguous().to(dev),
                                               int(g["num_samples"]), uniforms=sorted_u.to(dev)).cpu()
    assert torch.equal(torch.sort(torch.cat([g["coarse_distances"].t(), fine], -1), -1).values, merged)
    want = orendering.importance_distances(g["coarse_distances"].t().contiguous(), g["coarse_weights"].t().contiguous(), sorted_u)
    assert_sampled_distances_close(fine[hit], want[hit], int(g["num_samples"]))
    det = rendering.inverse_transform_sampler(g["coarse_distances"].t().contiguous().to(dev), g["coarse_weights"].t().contiguous().to(dev),
                                              int(g["num_samples"]), deterministic=True).cpu()
    assert torch.all(det[:, 1:] >= det[:, :-1]) and torch.isfinite(det[hit]).all()


@pytest.mark.parametrize("name", RENDER_CASES)
def test_fused_hierarchical_golden(dev, name):
    """Fused two-pass kernel with the recorded uniforms against the reference's pass-2 outputs and gradients."""
    from vsrd_amd import rendering
    g = load_golden(name)
    S = int(g["num_samples"])
    union, params = hip_union(g, dev, requires_grad=True)
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    out = rendering.render_hierarchical(union, g["origins"].to(dev), g["directions"].to(dev), (0.0, 100.0), S, std, ratio,
                                        u_coarse=g["u_coarse"].to(dev), u_fine=g["u_fine"].to(dev),
                                        return_gradients=True, return_weights=True)
    labels = out["labels"]
    assert (labels.detach().cpu() - g["fine_labels"]).abs().max() < LABEL_TOL
    assert (out["weights"].detach().cpu() - g["fine_weights"].t()).abs().max() < LABEL_TOL
    miss = g["coarse_weights"].sum(0) == 0
    assert_sampled_distances_close(out["distances"].cpu()[~miss], g["fine_distances"].t()[~miss], S)
    # loss exactly as the golden generator assembled it (BCE + w * eikonal over well-conditioned rays)
    bce = olosses.silhouette_loss(labels, g["targets"].to(dev))
    eik = olosses.eikonal_loss(out["gradients"][(~miss).to(dev)])
    torch.testing.assert_close(bce.detach().cpu(), g["bce"], rtol=1e-4, atol=1e-6)
    # (||g|| - 1)^2 is ~1e-4..1e-2 and dominated by a few samples: one ill-conditioned sample position moves it by ~0.5 %
    torch.testing.assert_close(eik.detach().cpu(), g["eikonal_conditioned"], rtol=1e-2, atol=5e-6)
    loss = bce + float(g["eikonal_weight"]) * eik
    grads = torch.autograd.grad(loss, params)
    for got, key in zip(grads, ("grad_locations", "grad_dimensions", "grad_orientations")):
        scale = max(float(g[key].abs().max()), 1e-6)
        err = (got.cpu() - g[key]).abs().max().item() / scale
        assert err < GRAD_TOL, f"{key}: relative error {err:.3e}"


@pytest.mark.parametrize("name", ["g4_render_n4_s32_mid", "g4_render_n16_s64_mid"])
def test_api_two_pass_wrapper(dev, name):
    """The reference call surface driven exactly like scripts/main.py:511-523 (torch RNG on the device);
    checked against the oracle fed with the same draws."""
    from vsrd_amd import rendering
    g = load_golden(name)
    S = int(g["num_samples"])
    union, params = hip_union(g, dev, requires_grad=True)
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    kwargs = dict(distance_field=union, ray_positions=g["origins"].to(dev), ray_directions=g["directions"].to(dev),
                  distance_range=[0.0, 100.0], num_samples=S, sdf_std_deviation=std, cosine_ratio=ratio)
    torch.manual_seed(1234)
    with torch.no_grad():
        *_, cd, cw = rendering.hierarchical_volumetric_rendering(**kwargs)
    labels, grads, fd, fw = rendering.hierarchical_volumetric_rendering(**kwargs, sampled_distances=cd, sampled_weights=cw)
    R = g["origins"].shape[0]
    assert cd.shape == (S, R, 1) and cw.shape == (S - 1, R, 1)
    assert labels.shape == (R, g["locations"].shape[0]) and grads.shape == (2 * S - 1, R, 3)
    assert fd.shape == (2 * S, R, 1) and fw.shape == (2 * S - 1, R, 1)
    # replay the device RNG stream for the oracle
    torch.manual_seed(1234)
    u_coarse = torch.rand(R, 1, S, device=dev)[:, 0].cpu()
    u_fine = torch.rand(R, 1, S, device=dev)[:, 0].cpu()
    ounion, oparams = cpu_union(g, requires_grad=True)
    fine = orendering.hierarchical_render(ounion, g["origins"], g["directions"], (0.0, 100.0), S, std, ratio, u_coarse, u_fine)
    assert (labels.detach().cpu() - fine.labels.detach()).abs().max() < LABEL_TOL
    loss = olosses.silhouette_loss(labels, g["targets"].to(dev))
    oloss = olosses.silhouette_loss(fine.labels, g["targets"])
    for got, want in zip(torch.autograd.grad(loss, params), torch.autograd.grad(oloss, oparams)):
        err = (got.cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
        assert err < GRAD_TOL


def hip_residual_union(g, dev):
    from vsrd_amd import fields, rendering
    N = g["locations"].shape[0]
    loc, dim, rot, mlp = (g[k].to(dev) for k in ("locations", "dimensions", "orientations", "mlp_weights"))
    return fields.soft_union([
        rendering.sdfs.translation(rendering.sdfs.rotation(fields.instance_field(
            fields.residual_composition(rendering.sdfs.box(dim[i]), fields.ResidualField(mlp[i])), i, N), rot[i]), loc[i])
        for i in range(N)], float(g["temperature"]))


@pytest.mark.parametrize("name", RESIDUAL_CASES)
def test_residual_field_forward_golden(dev, name):
    """Config-3 field (box + per-instance residual MLP): forward parity against the reference's outputs (G10) and G6."""
    from vsrd_amd import rendering
    g = load_golden(name)
    S = int(g["num_samples"])
    union = hip_residual_union(g, dev)
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    # given the reference's own distances
    for prefix in ("coarse", "fine"):
        dist = g[f"{prefix}_distances"].t().contiguous().to(dev)
        labels, grads, weights = rendering.render_at_distances(union, g["origins"].to(dev), g["directions"].to(dev), dist, std, ratio)
        assert (labels.cpu() - g[f"{prefix}_labels"]).abs().max() < LABEL_TOL
        assert (weights.cpu() - g[f"{prefix}_weights"].t()).abs().max() < LABEL_TOL
        conditioned = g["coarse_weights"].sum(0) > 0 if prefix == "fine" else torch.ones(dist.shape[0], dtype=torch.bool)
        torch.testing.assert_close(grads.cpu()[conditioned], g[f"{prefix}_gradients"].transpose(0, 1)[conditioned], rtol=5e-3, atol=5e-4)
    # fused two-pass kernel with the recorded uniforms
    out = rendering.render_hierarchical(union, g["origins"].to(dev), g["directions"].to(dev), (0.0, 100.0), S, std, ratio,
                                        u_coarse=g["u_coarse"].to(dev), u_fine=g["u_fine"].to(dev))
    assert (out["labels"].cpu() - g["fine_labels"]).abs().max() < LABEL_TOL
    # closure call: union distance / labels / analytic normal at the sample points of a few rays, against the CPU oracle
    ounion = ofields.InstanceUnion(g["locations"], g["orientations"], g["dimensions"], float(g["temperature"]), g["mlp_weights"])
    rays = torch.nonzero(g["coarse_weights"].sum(0) > 0)[:8, 0]       # well-conditioned rays (no 1e6 m extrapolation)
    mid = (g["fine_distances"][:-1, rays] + g["fine_distances"][1:, rays]).t() / 2
    pts = g["origins"][rays, None, :] + g["directions"][rays, None, :] * mid[..., None]
    u, w, grad = rendering.evaluate_field(union, pts.to(dev), with_gradients=True)
    ou, ow, og = ounion.evaluate(pts)
    torch.testing.assert_close(u.cpu()[..., 0], ou, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(w.cpu(), ow, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(grad.cpu(), og, rtol=5e-3, atol=5e-4)


@pytest.mark.parametrize("name", RESIDUAL_CASES)
def test_residual_field_backward_golden(dev, name):
    """Config-3 backward: box parameters AND per-instance MLP weights, against the reference's autograd (G10) and against
    float64 autograd through the oracle for random adjoints."""
    from vsrd_amd import rendering, fields
    g = load_golden(name)
    S = int(g["num_samples"])
    N = g["locations"].shape[0]
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    loc, dim, rot, mlp = (g[k].to(dev).requires_grad_(True) for k in ("locations", "dimensions", "orientations", "mlp_weights"))
    union = fields.soft_union([
        rendering.sdfs.translation(rendering.sdfs.rotation(fields.instance_field(
            fields.residual_composition(rendering.sdfs.box(dim[i]), fields.ResidualField(mlp[i])), i, N), rot[i]), loc[i])
        for i in range(N)], float(g["temperature"]))
    out = rendering.render_hierarchical(union, g["origins"].to(dev), g["directions"].to(dev), (0.0, 100.0), S, std, ratio,
                                        u_coarse=g["u_coarse"].to(dev), u_fine=g["u_fine"].to(dev), return_gradients=True)
    miss = g["coarse_weights"].sum(0) == 0
    bce = olosses.silhouette_loss(out["labels"], g["targets"].to(dev))
    eik = olosses.eikonal_loss(out["gradients"][(~miss).to(dev)])
    loss = bce + float(g["eikonal_weight"]) * eik
    grads = torch.autograd.grad(loss, [loc, dim, rot, mlp])
    for got, key in zip(grads, ("grad_locations", "grad_dimensions", "grad_orientations", "grad_mlp_weights")):
        scale = max(float(g[key].abs().max()), 1e-6)
        err = (got.cpu() - g[key]).abs().max().item() / scale
        print(f"[residual golden] {key}: rel err {err:.3e}")
        assert err < 2e-3, f"{key}: relative error {err:.3e}"
    # given distances + random adjoints vs float64 autograd through the oracle (isolates the kernel from the sampler)
    dist = g["fine_distances"].t().contiguous()
    keep = ~miss
    labels, gradients, weights = rendering.render_at_distances(union, g["origins"][keep].to(dev), g["directions"][keep].to(dev),
                                                               dist[keep].to(dev), std, ratio)
    gen = torch.Generator().manual_seed(0)
    lam = torch.randn(labels.shape, generator=gen)
    gam = torch.randn(gradients.shape, generator=gen) * 0.05
    om = torch.randn(weights.shape, generator=gen) * 0.1
    got = torch.autograd.grad([labels, gradients, weights], [loc, dim, rot, mlp], [lam.to(dev), gam.to(dev), om.to(dev)])
    def oracle_vjp(dtype):
        leaves = [g[k].to(dtype).requires_grad_(True) for k in ("locations", "dimensions", "orientations", "mlp_weights")]
        ou = ofields.InstanceUnion(leaves[0], leaves[2], leaves[1], float(g["temperature"]), leaves[3])
        o = orendering.render_given_distances(ou, g["origins"][keep].to(dtype), g["directions"][keep].to(dtype), dist[keep].to(dtype), std, ratio)
        return torch.autograd.grad([o.labels, o.gradients, o.weights], leaves, [lam.to(dtype), gam.to(dtype), om.to(dtype)])

    # Random adjoints of the SDF gradients weigh every sample, including those that sit on a kink of a box SDF (arg max, relu): at a
    # sharp schedule the fp32 evaluation of the reference's own formulas then differs from the exact one by several per cent.  The
    # kernel has to be as close to the float64 answer as fp32 allows: 2e-3, or twice the fp32 oracle's own distance from it.
    want, want32 = oracle_vjp(torch.float64), oracle_vjp(torch.float32)
    for a, b, c, key in zip(got, want, want32, ("locations", "dimensions", "orientations", "mlp_weights")):
        scale = max(b.abs().max().item(), 1e-9)
        err = (a.cpu().double() - b).abs().max().item() / scale
        floor = (c.double() - b).abs().max().item() / scale
        print(f"[residual f64 oracle] {key}: rel err {err:.3e} (fp32 oracle: {floor:.3e})")
        assert err < max(2e-3, 2.0 * floor), f"{key}: relative error {err:.3e} (fp32 oracle {floor:.3e})"


def test_philox_mode_matches_oracle_on_exported_uniforms(dev):
    from vsrd_amd import rendering
    g = load_golden("g4_render_n4_s32_mid")
    S = int(g["num_samples"])
    union, _ = hip_union(g, dev)
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    args = (union, g["origins"].to(dev), g["directions"].to(dev), (0.0, 100.0), S, std, ratio)
    a = rendering.render_hierarchical(*args, seed=7, stream_offset=3, return_uniforms=True)
    b = rendering.render_hierarchical(*args, seed=7, stream_offset=3, return_uniforms=True)
    c = rendering.render_hierarchical(*args, seed=8, stream_offset=3, return_uniforms=True)
    assert torch.equal(a["labels"], b["labels"]) and torch.equal(a["u_fine"], b["u_fine"])      # deterministic
    assert not torch.equal(a["u_coarse"], c["u_coarse"])
    u = torch.cat([a["u_coarse"].flatten(), a["u_fine"].flatten()]).cpu()
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    # the fine uniforms come out sorted (order statistics via exponential spacings): k-th of S has mean k/(S+1)
    uf = a["u_fine"].cpu()
    assert torch.all(uf[:, 1:] >= uf[:, :-1])
    expected = torch.arange(1, S + 1, dtype=torch.float32) / (S + 1)
    assert (uf.mean(0) - expected).abs().max() < 0.02
    assert abs(float(uf[:, S // 2].var()) - (S // 2 + 1) * (S - S // 2) / ((S + 1) ** 2 * (S + 2))) < 2e-3
    ounion, _ = cpu_union(g)
    fine = orendering.hierarchical_render(ounion, g["origins"], g["directions"], (0.0, 100.0), S, std, ratio,
                                          a["u_coarse"].cpu(), a["u_fine"].cpu())
    assert (a["labels"].cpu() - fine.labels).abs().max() < LABEL_TOL


def test_skip_exact_misses_is_exact(dev):
    from vsrd_amd import rendering
    g = load_golden("g4_render_n4_s32_late")
    S = int(g["num_samples"])
    union, params = hip_union(g, dev, requires_grad=True)
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    kw = dict(u_coarse=g["u_coarse"].to(dev), u_fine=g["u_fine"].to(dev))
    args = (union, g["origins"].to(dev), g["directions"].to(dev), (0.0, 100.0), S, std, ratio)
    from vsrd_amd.rendering import renderers
    assert int((g["coarse_weights"].sum(0) == 0).sum()) > 0
    lam = torch.randn(g["fine_labels"].shape, generator=torch.Generator().manual_seed(3)).to(dev)
    for one_ray_per_wave in (True, False):
        renderers.STEP_WAVE_PER_RAY = one_ray_per_wave
        try:
            full = rendering.render_hierarchical(*args, **kw)["labels"]
            fast = rendering.render_hierarchical(*args, skip_exact_misses=True, **kw)["labels"]
            grads = [torch.autograd.grad((out * lam).sum(), params) for out in (full, fast)]
        finally:
            renderers.STEP_WAVE_PER_RAY = False
        missed = (g["coarse_weights"].sum(0) == 0).to(dev)
        assert torch.equal(fast[missed], torch.zeros_like(fast[missed])) and torch.equal(full[missed], fast[missed])
        if one_ray_per_wave:                                     # rays are independent of each other: skipping changes nothing, bit for bit
            assert torch.equal(full, fast) and all(torch.equal(a, b) for a, b in zip(*grads))
        else:
            # several rays per wave (quad_step.h): a wave in which an un-skipped miss sits next to rays that hit takes the running-minimum
            # soft-min for all of them (the miss's fine samples are extrapolated to 1e6 m), so its neighbours agree to rounding, not bit for bit
            assert (full - fast).abs().max() < 2e-6
            for a, b in zip(*grads):
                assert (a - b).abs().max() <= 2e-5 * max(float(b.abs().max()), 1e-6)


# Labels of two modes of the same kernel: the modes round the local position differently (products with exact zeros and ones left out,
# the order of the soft-min sums), and the importance sampler turns a last-bit difference of a pass-1 weight into a displaced fine
# sample wherever its cdf is flat (tests/test_oracle_golden.py: up to 5e-3 m) -- a few labels move by a few 1e-6, never all of them.
LABEL_MODE_TOL = 5e-6


@pytest.mark.parametrize("name", ["g4_render_n16_s64_mid", "g4_render_n4_s32_late", "g4_render_n4_s32_step0"])
def test_culling_is_invisible(dev, name):
    """The two things the kernels do differently from the reference's closure loop must not show: conservative soft-min culling
    (instances whose weight is below exp(-18) on every lane are skipped wave-uniformly: sphere-bound test, then exact test on the
    box distance), the soft-min shift known before the instance loop, and the shortened products for rotations about y.  Baseline: every instance evaluated at every sample,
    running-minimum shift, general rotations (VSRD_FLAG_NO_CULLING | VSRD_FLAG_RUNNING_MINIMUM | VSRD_FLAG_GENERAL_ROTATIONS)."""
    from vsrd_amd import rendering
    from vsrd_amd.rendering import renderers
    g = load_golden(name)
    S = int(g["num_samples"])
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    results = {}
    modes = {"default": (True, False, False), "running": (True, True, False), "general": (True, False, True), "baseline": (False, True, True)}
    for mode, (culling, running, general) in modes.items():
        renderers.CULLING, renderers.RUNNING_MINIMUM, renderers.GENERAL_ROTATIONS = culling, running, general
        try:
            union, params = hip_union(g, dev, requires_grad=True)
            out = rendering.render_hierarchical(union, g["origins"].to(dev), g["directions"].to(dev), (0.0, 100.0), S, std, ratio,
                                                u_coarse=g["u_coarse"].to(dev), u_fine=g["u_fine"].to(dev), return_gradients=True)
            lam = torch.randn(out["labels"].shape, generator=torch.Generator().manual_seed(2)).to(dev)
            gam = (torch.randn(out["gradients"].shape, generator=torch.Generator().manual_seed(3)) * 0.01).to(dev)
            hit = (g["coarse_weights"].sum(0) > 0).to(dev)
            loss = (out["labels"] * lam).sum() + (out["gradients"][hit] * gam[hit]).sum()
            results[mode] = (out["labels"].detach(), torch.autograd.grad(loss, params))
        finally:
            renderers.CULLING, renderers.RUNNING_MINIMUM, renderers.GENERAL_ROTATIONS = True, False, False
    # (the y-rotation fast path only leaves out products with exact zeros and ones; the fixed shift changes the rounding of the sums)
    failures = []
    for mode in ("default", "running", "general"):
        tag = f"test_culling_is_invisible[{name}]"
        label_error = margin(tag, f"labels, {mode}", (results[mode][0] - results["baseline"][0]).abs().max(), LABEL_MODE_TOL)
        moved = ((results[mode][0] - results["baseline"][0]).abs() > 1e-6).float().mean()
        grad_error = max(float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6) for a, b in zip(results[mode][1], results["baseline"][1]))
        margin(tag, f"gradients, {mode}", grad_error, 1e-4)
        margin(tag, f"labels off by > 1e-6, {mode}", moved, 1e-2)
        if not (label_error < LABEL_MODE_TOL and moved <= 1e-2 and grad_error <= 1e-4):
            failures.append((mode, label_error, float(moved), grad_error))
    assert not failures, failures
    assert not torch.equal(results["default"][1][0], results["running"][1][0])      # (the two shifts really are different code paths)


def test_residual_tile_culling_is_invisible(dev):
    """Residual fields are culled per 16-point tile (the MLP is skipped for rows of lanes whose soft-min weight is below exp(-18) on
    the box distance alone): labels, box-parameter and MLP-weight gradients must not change against evaluating everything.
    A street-like scene (8 instances spread over 8-60 m, late schedule) makes most tiles of most rays culled."""
    import bench
    from vsrd_amd import fields, models, rendering
    from vsrd_amd.rendering import renderers
    V, H, W, N, S = 1, 24, 176, 8, 64
    K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(0, V, H, W, N)
    det = models.BoxParameters3D(1, N)
    with torch.no_grad():
        det.locations.copy_(raw_loc); det.dimensions.copy_(raw_dim); det.orientations.copy_(raw_ori)
        boxes = det()
    cam, dirs = rendering.ray_casting((H, W), K.to(dev), E.to(dev))
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
    dirs = dirs.reshape(-1, 3)[H * W // 3:]                   # rows through the objects
    origins = origins[H * W // 3:]
    gen = torch.Generator().manual_seed(5)
    results = {}
    for culling in (True, False):
        renderers.CULLING = culling
        try:
            inst = fields.pack_instances(boxes["locations"][0], boxes["orientations"][0], boxes["dimensions"][0]).to(dev).requires_grad_(True)
            mlp = (torch.randn(N, 1617, generator=torch.Generator().manual_seed(6)) * 0.3).to(dev).requires_grad_(True)
            block = fields.FieldBlock(inst, 0.15, mlp, None)
            out = rendering.render_hierarchical(block, origins, dirs, (0.0, 100.0), S, 0.15, 0.9, seed=3, return_gradients=True)
            lam = torch.randn(out["labels"].shape, generator=torch.Generator().manual_seed(2)).to(dev)
            hit = out["labels"].detach().sum(-1) > 0.5
            loss = (out["labels"] * lam).sum() + 0.01 * ((out["gradients"][hit].norm(dim=-1) - 1.0) ** 2).mean()
            results[culling] = (out["labels"].detach(), torch.autograd.grad(loss, (inst, mlp)), int(hit.sum()))
        finally:
            renderers.CULLING = True
    assert results[True][2] > 100
    assert (results[True][0] - results[False][0]).abs().max() < 1e-6
    for a, b in zip(results[True][1], results[False][1]):
        assert (a - b).abs().max() <= 2e-4 * max(float(b.abs().max()), 1e-6)


def test_single_origin_and_leading_dims(dev):
    """main.py:1011-1026 renders image rows with a [3] camera position and [W,3] directions."""
    from vsrd_amd import rendering
    g = load_golden("g4_render_n4_s32_mid")
    union, _ = hip_union(g, dev)
    dist = g["fine_distances"].t().contiguous().to(dev)
    a = rendering.render_at_distances(union, g["origins"].to(dev), g["directions"].to(dev), dist, 0.5, 0.5)[0]
    b = rendering.render_at_distances(union, g["origins"][0].to(dev), g["directions"].to(dev), dist, 0.5, 0.5)[0]
    assert torch.equal(a, b)
    torch.manual_seed(5)
    out = rendering.hierarchical_volumetric_rendering(union, g["origins"][0].to(dev), g["directions"].reshape(12, 20, 3).to(dev),
                                                      [0.0, 100.0], 32, 0.5, 0.5)
    assert out[0].shape == (12, 20, 4) and out[1].shape == (31, 12, 20, 3) and out[2].shape == (32, 12, 20, 1) and out[3].shape == (31, 12, 20, 1)


def test_backward_linearity_and_determinism(dev):
    from vsrd_amd import rendering
    g = load_golden("g4_render_n16_s64_mid")
    union, params = hip_union(g, dev, requires_grad=True)
    dist = g["fine_distances"].t().contiguous().to(dev)
    labels, grads, weights = rendering.render_at_distances(union, g["origins"].to(dev), g["directions"].to(dev), dist, 0.55, 0.5)
    gen = torch.Generator().manual_seed(0)
    lam = torch.randn(labels.shape, generator=gen).to(dev)
    gam = (torch.randn(grads.shape, generator=gen) * 0.1).to(dev)
    om = (torch.randn(weights.shape, generator=gen) * 0.1).to(dev)
    def vjp(l, g_, o):
        return torch.autograd.grad([labels, grads, weights], params, [l, g_, o], retain_graph=True)
    base = vjp(lam, gam, om)
    again = vjp(lam, gam, om)
    for a, b in zip(base, again):
        assert torch.equal(a, b)                                  # deterministic two-stage reduction
    double = vjp(2 * lam, 2 * gam, 2 * om)
    for a, b in zip(base, double):
        torch.testing.assert_close(2 * a, b, rtol=1e-5, atol=1e-6)
    # against float64 autograd through the oracle
    loc, dim, rot = (g[k].double().requires_grad_(True) for k in ("locations", "dimensions", "orientations"))
    ou = ofields.InstanceUnion(loc, rot, dim, float(g["temperature"]))
    o = orendering.render_given_distances(ou, g["origins"].double(), g["directions"].double(), dist.cpu().double(), 0.55, 0.5)
    want = torch.autograd.grad([o.labels, o.gradients, o.weights], [loc, dim, rot], [lam.cpu().double(), gam.cpu().double(), om.cpu().double()])
    for got, ref in zip(base, want):
        err = (got.cpu().double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
        assert err < 2e-3, err


def test_argument_validation(dev):
    from vsrd_amd import rendering, fields, _lib
    g = load_golden("g4_render_n4_s32_mid")
    union, _ = hip_union(g, dev)
    with pytest.raises(_lib.VsrdHipError):
        rendering.render_at_distances(union, g["origins"], g["directions"].to(dev), g["fine_distances"].t().contiguous().to(dev), 0.5)
    with pytest.raises(fields.UnsupportedFieldError):
        rendering.render_at_distances(lambda p: p.norm(dim=-1, keepdim=True), g["origins"].to(dev), g["directions"].to(dev),
                                      g["fine_distances"].t().contiguous().to(dev), 0.5)
    with pytest.raises(_lib.VsrdHipError):   # std must be positive
        rendering.render_at_distances(union, g["origins"].to(dev), g["directions"].to(dev), g["fine_distances"].t().contiguous().to(dev), 0.0)
    empty = rendering.render_at_distances(union, g["origins"][:0].to(dev), g["directions"][:0].to(dev),
                                          g["fine_distances"].t()[:0].contiguous().to(dev), 0.5)
    assert empty[0].shape == (0, 4)
    # the small entry points reject what they cannot do instead of truncating
    from vsrd_amd import losses
    with pytest.raises(_lib.VsrdHipError):
        losses.linear_sum_assignment(torch.zeros(65, 3, device=dev))                  # more than 64 rows
    with pytest.raises(_lib.VsrdHipError):
        rendering.sample_rays(torch.ones(10000, device=dev), 4096)                    # more than 2048 samples
    with pytest.raises(_lib.VsrdHipError):
        rendering.sample_rays(torch.ones(16), 4)                                      # host tensor
    with pytest.raises(ValueError):
        _lib.make_config(1, 8, (0.0, 1.0), 0.5, 1.0, 1e-6, 3, schedule=torch.ones(2, device=dev))


def test_sphere_tracing_and_surface_normal_g9(dev):
    """N1: vsrd.rendering.sphere_tracing / surface_normal over the soft union, driven like main.py:1028-1041
    (compose(field, itemgetter(0)), shared camera position), against the reference's own outputs."""
    import operator
    from vsrd_amd import rendering, fields, utils
    g = load_golden("g9_sphere_tracing")
    N = g["locations"].shape[0]
    loc, dim, rot = (g[k].to(dev) for k in ("locations", "dimensions", "orientations"))
    union = fields.soft_union([
        rendering.sdfs.translation(rendering.sdfs.rotation(fields.instance_field(rendering.sdfs.box(dim[i]), i, N), rot[i]), loc[i])
        for i in range(N)], float(g["temperature"]))
    field = utils.compose(union, operator.itemgetter(0))
    pos, conv = rendering.sphere_tracing(field, g["origins"].to(dev), g["directions"].to(dev), num_iterations=200,
                                         convergence_criteria=0.01, bounding_radius=100.0)
    assert conv.shape == (96, 1) and conv.dtype == torch.bool
    assert torch.equal(conv.cpu(), g["convergence_masks"])
    hit = g["convergence_masks"][:, 0]
    assert 5 < int(hit.sum()) < 96
    torch.testing.assert_close(pos.cpu()[hit], g["surface_positions"][hit], rtol=1e-5, atol=2e-3)
    normals = rendering.surface_normal(field, pos)
    torch.testing.assert_close(normals.cpu()[hit], g["surface_normals"][hit], rtol=1e-3, atol=2e-3)
    fd = rendering.surface_normal(field, pos, finite_difference_epsilon=1e-3)
    assert (torch.nn.functional.cosine_similarity(fd.cpu()[hit], g["surface_normals"][hit], dim=-1) > 0.97).all()      # finite differences of an fp32 distance: a sanity check only
    # shared origin + image-shaped directions (main.py passes [3] and [H,W,3]); initialization=False as in main.py:1036
    pos2, conv2 = rendering.sphere_tracing(field, g["origins"][0].to(dev), g["directions"].reshape(8, 12, 3).to(dev), 200, 0.01,
                                           bounding_radius=100.0, initialization=False)
    assert pos2.shape == (8, 12, 3) and conv2.shape == (8, 12, 1)
    assert torch.equal(conv2.reshape(-1, 1).cpu(), g["convergence_masks"])


@pytest.fixture(params=["multi_ray", "wave_per_ray", "split_ray"])
def step_mapping(request):
    """The three mappings of vsrd_render_silhouette_step: four / two consecutive rays per wave (the default for these dense launches), one
    ray per wave (VSRD_FLAG_STEP_WAVE_PER_RAY), a ray split over two waves (VSRD_FLAG_STEP_SPLIT_RAY: what small gathered launches take)."""
    from vsrd_amd.rendering import renderers
    renderers.STEP_WAVE_PER_RAY, renderers.STEP_SPLIT_RAY = request.param == "wave_per_ray", request.param == "split_ray"
    yield request.param
    renderers.STEP_WAVE_PER_RAY = renderers.STEP_SPLIT_RAY = False


@pytest.mark.parametrize("name", ["g4_render_n4_s32_mid", "g4_render_n16_s64_mid", "g4_render_n4_s32_late", "g4_render_n3_s20_mid",
                                  "g17_render_n64_s128_mid"])     # the last one: BASELINE config 5's shape
def test_fused_silhouette_step_matches_two_launch_path_and_golden(dev, name, step_mapping):
    """vsrd_render_silhouette_step (render + BCE + adjoint in one launch), in each of its three mappings, against the reference's loss /
    gradients (golden, cases without an eikonal term) and against the two-launch path with torch's BCE, including a Hungarian-style
    column permutation."""
    from vsrd_amd import rendering
    g = load_golden(name)
    S = int(g["num_samples"])
    N = g["locations"].shape[0]
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    rays = (g["origins"].to(dev), g["directions"].to(dev))
    uni = dict(u_coarse=g["u_coarse"].to(dev), u_fine=g["u_fine"].to(dev))
    union, params = hip_union(g, dev, requires_grad=True)
    tag = f"test_fused_silhouette_step_matches_two_launch_path_and_golden[{step_mapping}-{name}]"
    loss, labels = rendering.silhouette_step(union, *rays, g["targets"].to(dev), (0.0, 100.0), S, std, ratio, return_labels=True, **uni)
    assert margin(tag, "labels vs golden", (labels.cpu() - g["fine_labels"]).abs().max(), LABEL_TOL) < LABEL_TOL
    margin(tag, "loss vs golden (rel)", abs(float(loss) - float(g["bce"])) / max(abs(float(g["bce"])), 1e-12), 1e-4)
    torch.testing.assert_close(loss.detach().cpu(), g["bce"], rtol=1e-4, atol=1e-6)
    grads = torch.autograd.grad(loss, params)
    if float(g["eikonal_weight"]) == 0.0:
        for got, key in zip(grads, ("grad_locations", "grad_dimensions", "grad_orientations")):
            scale = max(float(g[key].abs().max()), 1e-6)
            assert margin(tag, "gradients vs golden", (got.cpu() - g[key]).abs().max().item() / scale, GRAD_TOL) <= GRAD_TOL, key
    # two-launch path, same uniforms, torch BCE.  (Two kernels with different orders of summation: 2e-4 of the largest entry -- the
    # N = 64 / S = 128 case sits at 1.2e-4 in some builds.)
    union2, params2 = hip_union(g, dev, requires_grad=True)
    ref_labels = rendering.render_hierarchical(union2, *rays, (0.0, 100.0), S, std, ratio, **uni)["labels"]
    ref_loss = olosses.silhouette_loss(ref_labels, g["targets"].to(dev))
    torch.testing.assert_close(loss.detach(), ref_loss.detach(), rtol=1e-5, atol=1e-7)
    for a, b in zip(grads, torch.autograd.grad(ref_loss, params2)):
        assert margin(tag, "gradients vs two-launch", float((a - b).abs().max()) / max(float(b.abs().max()), 1e-6), 2e-4) <= 2e-4
    # matched-instance form of main.py:653-671 (pd_indices / gt_indices), fewer ground-truth instances than predictions
    if N >= 3:
        pd_idx = torch.tensor([2, 0], device=dev)
        gt_idx = torch.tensor([1, 0], device=dev)
        tg = g["targets"][:, :2].to(dev).contiguous()
        union3, params3 = hip_union(g, dev, requires_grad=True)
        fused = rendering.silhouette_step(union3, *rays, tg, (0.0, 100.0), S, std, ratio, pd_indices=pd_idx, gt_indices=gt_idx, **uni)
        union4, params4 = hip_union(g, dev, requires_grad=True)
        lab = rendering.render_hierarchical(union4, *rays, (0.0, 100.0), S, std, ratio, **uni)["labels"]
        want = olosses.silhouette_loss(lab, tg, pd_idx, gt_idx)
        torch.testing.assert_close(fused.detach(), want.detach(), rtol=1e-5, atol=1e-7)
        for a, b in zip(torch.autograd.grad(fused, params3), torch.autograd.grad(want, params4)):
            assert (a - b).abs().max() <= 1e-4 * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize("residual", [False, True])
def test_field_evaluation_is_differentiable(dev, residual):
    """The closure call of scripts/main.py:433-509 carries autograd in the reference; here vsrd_field_eval_backward does: gradients of a
    functional of (distances, labels) -- and of hard-union distances -- w.r.t. boxes, MLP weights and the positions, against the
    oracle's autograd.  Then sphere_tracing(differentiable=True) (renderers.py:59-72), whose Newton step is the only place the
    reference uses it."""
    from vsrd_amd import fields, rendering
    g = load_golden("g10_render_residual_n3_s16")
    N = g["locations"].shape[0]
    gen = torch.Generator().manual_seed(3)
    centres = g["locations"][torch.randint(0, N, (300,), generator=gen)]
    points = centres + torch.randn(300, 3, generator=gen) * 1.5
    wa, wb = torch.randn(300, generator=gen), torch.randn(300, N, generator=gen)
    T = 0.4

    def parameters(device):
        leaves = [g[k].clone().to(device).requires_grad_(True) for k in ("locations", "dimensions", "orientations")]
        if residual:
            leaves.append(g["mlp_weights"].clone().to(device).requires_grad_(True))
        return leaves, points.clone().to(device).requires_grad_(True)

    def hip_field(leaves, hard):
        loc, dim, rot = leaves[:3]
        def member(i):
            box = rendering.sdfs.box(dim[i])
            if residual:
                box = fields.residual_composition(box, fields.ResidualField(leaves[3][i]))
            return rendering.sdfs.translation(rendering.sdfs.rotation(fields.instance_field(box, i, N), rot[i]), loc[i])
        members = [member(i) for i in range(N)]
        return fields.hard_union(members) if hard else fields.soft_union(members, T)

    for hard in (False, True):
        leaves, pts = parameters(dev)
        if hard:
            loss = (rendering.evaluate_field(hip_field(leaves, True), pts)[..., 0] * wa.to(dev)).sum()
        else:
            u, w = rendering.evaluate_field(hip_field(leaves, False), pts)
            loss = (u[..., 0] * wa.to(dev)).sum() + (w * wb.to(dev)).sum()
        got = torch.autograd.grad(loss, leaves + [pts])
        oleaves, opts = parameters(torch.device("cpu"))
        ounion = ofields.InstanceUnion(oleaves[0], oleaves[2], oleaves[1], T, oleaves[3] if residual else None)
        if hard:
            oloss = (ounion.hard_distance(opts) * wa).sum()
        else:
            ou, ow, _ = ounion.evaluate(opts)
            oloss = (ou * wa).sum() + (ow * wb).sum()
        torch.testing.assert_close(loss.detach().cpu(), oloss.detach(), rtol=1e-4, atol=1e-4)
        for a, b in zip(got, torch.autograd.grad(oloss, oleaves + [opts])):
            assert (a.cpu() - b).abs().max() <= 2e-3 * max(float(b.abs().max()), 1e-6), (hard, a.shape)

    # differentiable sphere tracing: x' = x + r (-sdf(x) / (grad sdf(x) . r)) at converged rays, grad taken as a constant
    leaves, _ = parameters(dev)
    origins = torch.zeros(64, 3)
    directions = torch.nn.functional.normalize(g["locations"][torch.arange(64) % N] + torch.randn(64, 3, generator=gen) * 0.3, dim=-1)
    field = hip_field(leaves, True)
    traced, converged = rendering.sphere_tracing(field, origins.to(dev), directions.to(dev), 64, 1e-3, differentiable=True)
    assert int(converged.sum()) > 16
    plain, _ = rendering.sphere_tracing(field, origins.to(dev), directions.to(dev), 64, 1e-3)
    probe = torch.randn(64, 3, generator=gen)
    got = torch.autograd.grad((traced * probe.to(dev)).sum(), leaves)
    oleaves, _ = parameters(torch.device("cpu"))
    ounion = ofields.InstanceUnion(oleaves[0], oleaves[2], oleaves[1], T, oleaves[3] if residual else None)
    x = plain.detach().cpu().requires_grad_(True)
    sdf = ounion.hard_distance(x)
    normal, = torch.autograd.grad(sdf.sum(), x, retain_graph=True)
    step = -sdf / (normal * directions).sum(-1)
    expected = torch.where(converged.cpu(), x.detach() + directions * step[..., None], x.detach())
    torch.testing.assert_close(traced.detach().cpu(), expected.detach(), rtol=1e-4, atol=1e-4)
    for a, b in zip(got, torch.autograd.grad((expected * probe).sum(), oleaves)):
        assert (a.cpu() - b).abs().max() <= 5e-3 * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize("name", RESIDUAL_CASES)
def test_fused_residual_step_matches_two_launch_path(dev, name):
    """vsrd_render_residual_step (render + silhouette BCE + eikonal + adjoint of a residual field, one launch) against the two-launch
    path (render_hierarchical, torch BCE + eikonal, render_backward) on the same uniforms: loss terms, box gradients, MLP-weight
    gradients -- with and without a Hungarian-style column permutation."""
    from vsrd_amd import fields, rendering
    g = load_golden(name)
    S, N = int(g["num_samples"]), g["locations"].shape[0]
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    # rays that hit something: on the others the fine samples are extrapolated to ~1e6 m (samplers.py:33) and the eikonal adjoint
    # there is fp32 noise in any implementation ("Known ill-conditioning of the reference itself")
    hit = g["coarse_weights"].sum(0).reshape(-1) > 0
    rays = (g["origins"][hit].to(dev), g["directions"][hit].to(dev))
    uni = dict(u_coarse=g["u_coarse"][hit].to(dev), u_fine=g["u_fine"][hit].to(dev))
    targets = g["targets"][hit].to(dev)
    eikonal_ratio = 0.01

    def leaves():
        return [g[k].clone().to(dev).requires_grad_(True) for k in ("locations", "dimensions", "orientations", "mlp_weights")]

    def union(params):
        loc, dim, rot, mlp = params
        return fields.soft_union([
            rendering.sdfs.translation(rendering.sdfs.rotation(fields.instance_field(
                fields.residual_composition(rendering.sdfs.box(dim[i]), fields.ResidualField(mlp[i])), i, N), rot[i]), loc[i])
            for i in range(N)], float(g["temperature"]))

    for pd, gt in ((None, None), (torch.tensor([2, 0], device=dev), torch.tensor([1, 2], device=dev))):
        params = leaves()
        loss, terms, labels = rendering.silhouette_step(union(params), *rays, targets, (0.0, 100.0), S, std, ratio, pd_indices=pd, gt_indices=gt,
                                                        eikonal_ratio=eikonal_ratio, return_terms=True, return_labels=True, **uni)
        grads = torch.autograd.grad(loss, params)
        reference = leaves()
        out = rendering.render_hierarchical(union(reference), *rays, (0.0, 100.0), S, std, ratio, return_gradients=True, **uni)
        silhouette = olosses.silhouette_loss(out["labels"], targets, pd, gt)
        eikonal = olosses.eikonal_loss(out["gradients"])
        assert (labels - out["labels"]).abs().max() < 1e-6
        torch.testing.assert_close(terms[0], silhouette.detach(), rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(terms[1], eikonal.detach(), rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(loss.detach(), (silhouette + eikonal_ratio * eikonal).detach(), rtol=1e-5, atol=1e-7)
        for a, b in zip(grads, torch.autograd.grad(silhouette + eikonal_ratio * eikonal, reference)):
            assert (a - b).abs().max() <= 2e-4 * max(float(b.abs().max()), 1e-6)


@pytest.fixture(params=["fp32_mfma", "split_bf16"])
def mlp_products(request):
    """How the front kernels of vsrd_render_residual_step multiply: the exact-fp32 matrix instruction (default) or the bf16 one with both
    operands split into two bfloat16 parts (VSRD_FLAG_MLP_SPLIT_BF16; csrc/residual.h) -- VERDICT r04 item 2: the same tests, the same
    tolerances, under both settings, with the margins of both in the record."""
    from vsrd_amd.rendering import renderers
    before = renderers.MLP_SPLIT_BF16
    renderers.MLP_SPLIT_BF16 = request.param == "split_bf16"
    yield request.param
    renderers.MLP_SPLIT_BF16 = before


@pytest.mark.parametrize("name", [n for n in RESIDUAL_CASES if n.startswith("g17")])
@pytest.mark.parametrize("form", ["default", "wave_per_ray", "single_kernel"])
def test_fused_residual_step_golden(dev, name, form, mlp_products):
    """vsrd_render_residual_step at the shapes the benchmark times (BASELINE config 3: N = 16, S = 64; the reference's S = 100), in each
    of its three forms -- default: residual_step_pair_kernel<2 / 4> + residual_mlp_adjoint_kernel (launches of <= 2048 rays split a ray
    over two waves); wave_per_ray: residual_step_front_kernel<2 / 4> + residual_mlp_adjoint_kernel, WHAT DENSE LAUNCHES (bench.py's config
    3) RUN; single_kernel: render_residual_step_kernel<2 / 4> -- against the REFERENCE's outputs: silhouettes, loss terms, and the gradients
    w.r.t. box parameters and per-instance MLP weights.  The launch takes the well-conditioned rays (no 1e6 m extrapolation); on
    those the step's loss is  mean BCE over R_c rays + ratio * eikonal, and the golden's two gradient terms (taken separately by
    the generator) combine to  (R / R_c) grad_bce + ratio * grad_eikonal  -- the dropped rays have clamped (zero) labels, whose
    BCE gradient is exactly zero."""
    from vsrd_amd import fields, rendering
    if form == "single_kernel" and mlp_products == "split_bf16":
        pytest.skip("the one-kernel form has no split-bf16 products (the flag is ignored there)")
    tag = f"test_fused_residual_step_golden[{mlp_products}-{form}-{name}]"
    g = load_golden(name)
    S, N = int(g["num_samples"]), g["locations"].shape[0]
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    keep = g["conditioned"].reshape(-1)
    R, Rc = keep.numel(), int(keep.sum())
    assert Rc >= 12 and float(g["fine_labels"][~keep].abs().max() if Rc < R else 0.0) < 1.0e-6
    eikonal_ratio = float(g["eikonal_weight"])
    loc, dim, rot, mlp = (g[k].clone().to(dev).requires_grad_(True) for k in ("locations", "dimensions", "orientations", "mlp_weights"))
    union = fields.soft_union([
        rendering.sdfs.translation(rendering.sdfs.rotation(fields.instance_field(
            fields.residual_composition(rendering.sdfs.box(dim[i]), fields.ResidualField(mlp[i])), i, N), rot[i]), loc[i])
        for i in range(N)], float(g["temperature"]))
    from vsrd_amd.rendering import renderers
    renderers.RESIDUAL_SINGLE_KERNEL, renderers.RESIDUAL_WAVE_PER_RAY = form == "single_kernel", form == "wave_per_ray"
    try:
        loss, terms, labels = rendering.silhouette_step(union, g["origins"][keep].to(dev), g["directions"][keep].to(dev), g["targets"][keep].to(dev),
                                                        (0.0, 100.0), S, std, ratio, u_coarse=g["u_coarse"][keep].to(dev), u_fine=g["u_fine"][keep].to(dev),
                                                        eikonal_ratio=eikonal_ratio, return_terms=True, return_labels=True)
        grads = torch.autograd.grad(loss, [loc, dim, rot, mlp])
    finally:
        renderers.RESIDUAL_SINGLE_KERNEL = renderers.RESIDUAL_WAVE_PER_RAY = False
    assert margin(tag, "labels vs golden", (labels.cpu() - g["fine_labels"][keep]).abs().max(), LABEL_TOL) < LABEL_TOL
    want_bce = torch.nn.functional.binary_cross_entropy(g["fine_labels"][keep].clamp(1.0e-6, 1.0 - 1.0e-6), g["targets"][keep], reduction="none").mean()
    margin(tag, "silhouette loss (rel)", abs(float(terms[0]) - float(want_bce)) / max(abs(float(want_bce)), 1e-12), 1e-4)
    margin(tag, "eikonal loss (rel)", abs(float(terms[1]) - float(g["eikonal_conditioned"])) / max(abs(float(g["eikonal_conditioned"])), 1e-12), 1e-2)
    torch.testing.assert_close(terms[0].cpu(), want_bce, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(terms[1].cpu(), g["eikonal_conditioned"], rtol=1e-2, atol=5e-6)
    for got, key in zip(grads, ("locations", "dimensions", "orientations", "mlp_weights")):
        want = (R / Rc) * g["grad_bce_" + key] + eikonal_ratio * g["grad_eikonal_" + key]
        err = (got.cpu() - want).abs().max().item() / max(float(want.abs().max()), 1e-6)
        print(f"[fused residual step vs reference] {name} {form} {mlp_products} {key}: rel err {err:.3e}")
        assert margin(tag, f"grad {key} vs golden", err, GRAD_TOL) < GRAD_TOL, f"{key}: relative error {err:.3e}"


@pytest.mark.parametrize("name", [n for n in RESIDUAL_CASES if n.startswith("g17")])
def test_residual_step_forms_agree(dev, name):
    """vsrd_render_residual_step has three forms.  Default: two kernels per chunk of rays (front part + MLP adjoint distributed by
    instance), and for launches of <= 2048 rays -- these goldens -- a front kernel that splits every ray over the two waves of a workgroup
    (residual_step_pair_kernel).  VSRD_FLAG_RESIDUAL_WAVE_PER_RAY keeps one wave per ray (what large launches run);
    VSRD_FLAG_RESIDUAL_SINGLE_KERNEL keeps the one-kernel form.  Same arithmetic, different summation trees: losses, labels and every
    gradient must agree to rounding, and the default form must repeat itself bit for bit."""
    from vsrd_amd import fields, rendering
    from vsrd_amd.rendering import renderers
    g = load_golden(name)
    S, N = int(g["num_samples"]), g["locations"].shape[0]
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    keep = g["conditioned"].reshape(-1)
    results = {}
    for form in ("default", "single_kernel", "wave_per_ray", "default"):
        renderers.RESIDUAL_SINGLE_KERNEL, renderers.RESIDUAL_WAVE_PER_RAY = form == "single_kernel", form == "wave_per_ray"
        try:
            inst = fields.pack_instances(g["locations"], g["orientations"], g["dimensions"]).to(dev).requires_grad_(True)
            mlp = g["mlp_weights"].clone().to(dev).requires_grad_(True)
            block = fields.FieldBlock(inst, float(g["temperature"]), mlp, None)
            loss, terms, labels = rendering.silhouette_step(block, g["origins"][keep].to(dev), g["directions"][keep].to(dev), g["targets"][keep].to(dev),
                                                            (0.0, 100.0), S, std, ratio, seed=11, stream_offset=3, eikonal_ratio=0.01,
                                                            return_terms=True, return_labels=True)
            out = (loss.detach(), terms, labels, torch.autograd.grad(loss, (inst, mlp)))
        finally:
            renderers.RESIDUAL_SINGLE_KERNEL = renderers.RESIDUAL_WAVE_PER_RAY = False
        if form in results:                                       # the second default run: bit-identical to the first
            first = results[form]
            assert torch.equal(out[0], first[0]) and torch.equal(out[2], first[2]) and all(torch.equal(a, b) for a, b in zip(out[3], first[3]))
        results[form] = out
    default = results["default"]
    for form in ("single_kernel", "wave_per_ray"):
        other = results[form]
        assert (default[2] - other[2]).abs().max() < 1e-6, form
        torch.testing.assert_close(default[1], other[1], rtol=1e-5, atol=1e-7)
        for a, b in zip(default[3], other[3]):
            assert (a - b).abs().max() <= 2e-4 * max(float(b.abs().max()), 1e-6), form


@pytest.mark.parametrize("name", [n for n in RESIDUAL_CASES if n.startswith("g17")])
def test_residual_backward_forms_agree(dev, name):
    """vsrd_render_backward on residual fields -- what hierarchical_volumetric_rendering(...).backward() of an unchanged main.py runs -- has
    two forms: two kernels per chunk of rays (render_backward_front_kernel + residual_mlp_adjoint_kernel, the default when the workspace
    holds a chunk's seeds) and the one-kernel form of round 1 (VSRD_FLAG_RESIDUAL_SINGLE_KERNEL; itself golden-checked by
    test_residual_field_backward_golden in round 1 and 2).  Random adjoints of labels, SDF gradients and weights: the gradients w.r.t.
    boxes and MLP weights must agree to rounding, and the default form must repeat itself bit for bit."""
    from vsrd_amd import fields, rendering
    from vsrd_amd.rendering import renderers
    g = load_golden(name)
    N = g["locations"].shape[0]
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    keep = g["conditioned"].reshape(-1)
    dist = g["fine_distances"].t().contiguous()[keep].to(dev)
    gen = torch.Generator().manual_seed(3)
    results = {}
    for form in ("default", "single_kernel", "default"):
        renderers.RESIDUAL_SINGLE_KERNEL = form == "single_kernel"
        try:
            inst = fields.pack_instances(g["locations"], g["orientations"], g["dimensions"]).to(dev).requires_grad_(True)
            mlp = g["mlp_weights"].clone().to(dev).requires_grad_(True)
            block = fields.FieldBlock(inst, float(g["temperature"]), mlp, None)
            labels, gradients, weights = rendering.render_at_distances(block, g["origins"][keep].to(dev), g["directions"][keep].to(dev), dist, std, ratio)
            if "adjoints" not in results:
                results["adjoints"] = (torch.randn(labels.shape, generator=gen).to(dev), (torch.randn(gradients.shape, generator=gen) * 0.05).to(dev),
                                       (torch.randn(weights.shape, generator=gen) * 0.1).to(dev))
            out = torch.autograd.grad([labels, gradients, weights], [inst, mlp], list(results["adjoints"]))
        finally:
            renderers.RESIDUAL_SINGLE_KERNEL = False
        if form in results:
            assert all(torch.equal(a, b) for a, b in zip(out, results[form]))
        results[form] = out
    for a, b in zip(results["default"], results["single_kernel"]):
        assert torch.isfinite(a).all() and float(b.abs().max()) > 0
        assert (a - b).abs().max() <= 2e-4 * max(float(b.abs().max()), 1e-6)


def test_shadow_rendering(dev):
    """vsrd.rendering.shadow_rendering (renderers.py:149-174): a point is in shadow when the ray from just above it towards the light
    converges on geometry.  One box hovering over a plane of points, light straight down (+y is down in the camera frame)."""
    from vsrd_amd import fields, rendering
    loc = torch.tensor([[0.0, -3.0, 10.0]], device=dev)
    dim = torch.tensor([[1.0, 0.5, 1.0]], device=dev)
    rot = torch.eye(3, device=dev)[None]
    field = fields.hard_union([rendering.sdfs.translation(rendering.sdfs.rotation(fields.instance_field(rendering.sdfs.box(dim[0]), 0, 1), rot[0]), loc[0])])
    xs = torch.linspace(-4.0, 4.0, 33, device=dev)
    points = torch.stack([xs, torch.zeros_like(xs), torch.full_like(xs, 10.0)], -1)
    normals = torch.tensor([0.0, -1.0, 0.0], device=dev).expand_as(points)
    light = torch.tensor([0.0, 1.0, 0.0], device=dev).expand_as(points)
    shadow = rendering.shadow_rendering(field, points, normals, light, 64, 1e-3, torch.ones(33, 1, dtype=torch.bool, device=dev), bounding_radius=50.0)
    inside = (xs.abs() < 0.95)
    outside = (xs.abs() > 1.05)
    assert bool(shadow[inside, 0].all()) and not bool(shadow[outside, 0].any())


def _random_scene(seed, N, R, S, general_rotations=True, scale=1.0):
    """Boxes in front of a camera at the origin with arbitrary (not about-y) rotations, rays aimed near them."""
    g = torch.Generator().manual_seed(seed)
    loc = torch.stack([torch.empty(N).uniform_(-6, 6, generator=g), torch.empty(N).uniform_(-1.0, 1.5, generator=g),
                       torch.empty(N).uniform_(8, 30, generator=g)], -1)
    dim = torch.stack([torch.empty(N).uniform_(0.75, 1.0, generator=g), torch.empty(N).uniform_(0.75, 1.0, generator=g),
                       torch.empty(N).uniform_(1.5, 2.5, generator=g)], -1)
    if general_rotations:
        q, r = torch.linalg.qr(torch.randn(N, 3, 3, generator=g, dtype=torch.float64))
        q = q * torch.sign(torch.diagonal(r, dim1=-2, dim2=-1)).unsqueeze(-2)
        q[:, :, 0] *= torch.sign(torch.linalg.det(q)).unsqueeze(-1)                # proper rotations
        rot = q.float()
    else:
        yaw = torch.empty(N).uniform_(-3.1, 3.1, generator=g)
        rot = torch.stack([torch.stack([torch.cos(yaw), torch.zeros(N), torch.sin(yaw)], -1), torch.tensor([0.0, 1.0, 0.0]).expand(N, 3),
                           torch.stack([-torch.sin(yaw), torch.zeros(N), torch.cos(yaw)], -1)], -2)
    rot = rot * scale
    target = loc[torch.randint(0, N, (R,), generator=g)] + torch.randn(R, 3, generator=g) * 0.7
    directions = torch.nn.functional.normalize(target, dim=-1)
    return dict(loc=loc, dim=dim, rot=rot, origins=torch.zeros(R, 3), directions=directions, u_coarse=torch.rand(R, S, generator=g),
                u_fine=torch.rand(R, S, generator=g), targets=torch.rand(R, N, generator=g))


@pytest.mark.parametrize("case", ["general_rotations", "scaled_rotations_no_bounds", "tiny_temperature_running_minimum"])
def test_paths_the_kernel_selects_by_itself_match_the_oracle(dev, case):
    """The kernels pick their instance-loop variants from the field: shortened products for rotations about y, culling and the
    soft-min floor only for orthonormal rotations, the floor only while max|dim| / T <= 50.  The goldens (the reference's
    rotation_matrix_y boxes at its schedule's temperatures) exercise one combination; these scenes exercise the others, against
    the CPU oracle on the same uniforms: arbitrary rotations; rotations scaled by 1.01 (not orthonormal to 1e-4: no culling, running
    minimum); T = 0.02 (running minimum with culling)."""
    from vsrd_amd import fields, rendering
    N, S, R = 5, 32, 256
    sc = _random_scene(11, N, R, S, general_rotations=(case != "tiny_temperature_running_minimum"),
                       scale=1.01 if case == "scaled_rotations_no_bounds" else 1.0)
    T, std, ratio = (0.02, 0.3, 0.7) if case == "tiny_temperature_running_minimum" else (0.4, 0.4, 0.4)

    def run(device, hip):
        l, d, r = (sc[k].clone().to(device).requires_grad_(True) for k in ("loc", "dim", "rot"))
        if hip:
            union = fields.soft_union([
                rendering.sdfs.translation(rendering.sdfs.rotation(fields.instance_field(rendering.sdfs.box(d[i]), i, N), r[i]), l[i])
                for i in range(N)], T)
            labels = rendering.render_hierarchical(union, sc["origins"].to(device), sc["directions"].to(device), (0.0, 100.0), S, std, ratio,
                                                   u_coarse=sc["u_coarse"].to(device), u_fine=sc["u_fine"].to(device))["labels"]
        else:
            union = ofields.InstanceUnion(l, r, d, T)
            labels = orendering.hierarchical_render(union, sc["origins"], sc["directions"], (0.0, 100.0), S, std, ratio,
                                                    sc["u_coarse"], sc["u_fine"]).labels
        loss = olosses.silhouette_loss(labels, sc["targets"].to(device))
        return labels.detach().cpu(), [x.cpu() for x in torch.autograd.grad(loss, [l, d, r])]

    hip_labels, hip_grads = run(dev, True)
    ref_labels, ref_grads = run(torch.device("cpu"), False)
    assert (hip_labels - ref_labels).abs().max() < 1e-4
    for a, b in zip(hip_grads, ref_grads):
        assert (a - b).abs().max() <= 5e-3 * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize("N,S,R,case", [(16, 64, 203, "yaw"), (4, 32, 64, "yaw"), (3, 20, 61, "yaw"), (16, 17, 5, "yaw"), (1, 8, 1, "yaw"),
                                        (7, 33, 130, "general"), (5, 64, 97, "tiny_temperature"), (12, 48, 256, "misses"), (16, 64, 64, "philox"),
                                        # more than 16 instances or more than 64 samples: two rays per wave, 32 lanes each (render_silhouette_pair_kernel)
                                        (64, 128, 37, "yaw"), (33, 64, 70, "yaw"), (20, 100, 51, "general"), (17, 20, 9, "misses"), (40, 128, 16, "philox"),
                                        (64, 16, 130, "yaw"), (9, 128, 33, "tiny_temperature")])
def test_quad_step_matches_wave_per_ray(dev, N, S, R, case):
    """vsrd_render_silhouette_step in its mappings: four consecutive rays per wave (quad_step.h, the default for dense launches with
    S <= 64 and N <= 16) or two (N <= 64, S <= 128) against one ray per wave (VSRD_FLAG_STEP_WAVE_PER_RAY; itself checked against the goldens and the oracle).  Ray
    counts that are not multiples of four, sample counts that are not multiples of 16, general rotations, a temperature that forces
    the running minimum, rays that miss everything (rows of a wave that drop out after pass 1), matched-instance weights, and the
    in-kernel Philox stream (same (seed, ray, sample) keys in both mappings)."""
    from vsrd_amd import fields, rendering
    from vsrd_amd.rendering import renderers
    T, std, ratio = (0.02, 0.3, 0.7) if case == "tiny_temperature" else ((0.1, 0.1, 0.9) if case == "misses" else (0.4, 0.4, 0.4))
    # Each mapping runs its OWN importance sampler on its own pass-1 weights (sums taken in a different order), and the sampler's division by
    # cdf differences turns a last-bit difference into a displaced fine sample on ill-conditioned rays; with the BCE's 1 / p label adjoints
    # one such ray can move a summed gradient by percents.  So the difference between two mappings has a heavy tail over scenes
    # (tests/variant_noise_debug.py, 24 scenes of (16, 64, 203): labels median 2.5e-6 / max 7.9e-5, gradients median 1e-4 / max 0.13 -- the same
    # scenes stand out whether the box norm is s * rsq(s) or sqrt(s) and 1 / sqrt(s)), and one pinned scene says little: several scenes,
    # the typical (median) difference held to the tight tolerance and every scene to the loose one.
    # (Philox: the sorted fine uniforms are partial sums of exponential spacings, summed in a different order by the two mappings)
    # (misses: T = std = 0.1 -- the two mappings cull different instance sets (e^-18 terms): on the two worst rays of the first scene the
    #  float32 and the float64 oracle differ by 1.3e-3, either mapping is within 2e-4 of the float32 oracle)
    label_tolerance, gradient_tolerance = (2e-4, 5e-3) if case in ("philox", "misses") else (2e-5, 5e-4)
    if S <= 20:                                            # ~6 m coarse bins: the sampler's cdf differences are small everywhere (observed 2.8e-4)
        gradient_tolerance = 1e-3
    # (round 5, VERDICT r04: no bound of 0.3 on a moving-sample gradient -- that is no bound.  The ADJOINT code of the two mappings is pinned
    #  where it can be: at FIXED samples -- the distances one forward launch saved, the BCE label adjoints of its labels, pushed through
    #  render_backward_{quad,pair}_kernel (the step's own forward sweep / reverse sweep / per-instance phase) and through render_backward_kernel --
    #  within 2e-4 of the largest entry on EVERY scene; the moving-sample gradients keep their median bound and are reported.)
    #  Two sets of label adjoints: seeded standard-normal ones (well conditioned: 2e-4 in EVERY case), and the step's own BCE ones, whose
    #  1 / p reach 1e6 x loss_scale -- there the two mappings' culling (e^-18 terms dropped at different granularity) is visible in the sharp
    #  cases (T = std = 0.1 / T = 0.02: observed 1.2e-3), so those are held to 5e-3 and the ordinary schedules to 2e-4.)
    loose_labels, fixed_gradient_tolerance = 1e-3, 2e-4
    bce_fixed_tolerance = 5e-3 if case in ("misses", "tiny_temperature") else 2e-4
    tag = f"test_quad_step_matches_wave_per_ray[{N}-{S}-{R}-{case}]"
    label_errors, gradient_errors, fixed_errors, bce_fixed_errors = [], [], [], []
    for scene_seed in (31 + N + S, 1000, 1001, 1002, 1003):
        sc = _random_scene(scene_seed, N, R, S, general_rotations=(case == "general"))
        directions = sc["directions"].clone()
        if case == "misses":                                     # every third ray looks away from the scene
            directions[::3] = torch.nn.functional.normalize(torch.tensor([[0.3, -0.9, -0.4]]), dim=-1)
        uni = {} if case == "philox" else dict(u_coarse=sc["u_coarse"].to(dev), u_fine=sc["u_fine"].to(dev))
        pd = torch.arange(0, N, 2, device=dev) if N >= 4 else None
        gt = torch.arange(pd.numel() - 1, -1, -1, device=dev) if pd is not None else None
        targets = sc["targets"][:, :pd.numel()].contiguous() if pd is not None else sc["targets"]
        results = {}
        for mode in ("quad", "wave"):
            renderers.STEP_WAVE_PER_RAY = mode == "wave"
            try:
                inst = fields.pack_instances(sc["loc"], sc["rot"], sc["dim"]).to(dev).requires_grad_(True)
                block = fields.FieldBlock(inst, T, None, None)
                loss, labels = rendering.silhouette_step(block, sc["origins"].to(dev), directions.to(dev), targets.to(dev), (0.0, 100.0), S, std, ratio,
                                                         pd_indices=pd, gt_indices=gt, seed=3, stream_offset=11, return_labels=True, **uni)
                results[mode] = (loss.detach(), labels, torch.autograd.grad(loss, inst)[0])
            finally:
                renderers.STEP_WAVE_PER_RAY = False
        quad, wave = results["quad"], results["wave"]
        assert torch.isfinite(quad[2]).all()
        if case != "misses" and scene_seed == 31 + N + S:
            assert float(wave[1].max()) > 0.05                  # the scene is seen
        label_errors.append(float((quad[1] - wave[1]).abs().max()))
        gradient_errors.append(float((quad[2] - wave[2]).abs().max()) / max(float(wave[2].abs().max()), 1e-6))
        torch.testing.assert_close(quad[0], wave[0], rtol=1e-4 if case != "philox" else 1e-3, atol=1e-7)
        # the adjoint of either mapping at the SAME samples and label adjoints
        inst = fields.pack_instances(sc["loc"], sc["rot"], sc["dim"]).to(dev).requires_grad_(True)
        out = rendering.render_hierarchical(fields.FieldBlock(inst, T, None, None), sc["origins"].to(dev), directions.to(dev), (0.0, 100.0), S, std, ratio,
                                            seed=3, stream_offset=11, skip_exact_misses=True, **uni)
        probabilities = out["labels"].detach().clone().requires_grad_(True)
        chosen = probabilities if pd is None else probabilities[:, pd]
        wanted = targets.to(dev) if gt is None else targets.to(dev)[:, gt]
        bce = torch.nn.functional.binary_cross_entropy(chosen.clamp(1.0e-6, 1.0 - 1.0e-6), wanted)
        lam_bce = torch.autograd.grad(bce, probabilities)[0]
        lam_normal = torch.randn(out["labels"].shape, generator=torch.Generator().manual_seed(scene_seed)).to(dev)
        for lam, errors in ((lam_normal, fixed_errors), (lam_bce, bce_fixed_errors)):
            fixed = {}
            for mode in ("quad", "wave"):
                renderers.STEP_WAVE_PER_RAY = mode == "wave"
                try:
                    fixed[mode] = torch.autograd.grad(out["labels"], inst, grad_outputs=lam, retain_graph=True)[0]
                finally:
                    renderers.STEP_WAVE_PER_RAY = False
            errors.append(float((fixed["quad"] - fixed["wave"]).abs().max()) / max(float(fixed["wave"].abs().max()), 1e-12))
    median = lambda values: sorted(values)[len(values) // 2]
    margin(tag, "labels, median of scenes", median(label_errors), label_tolerance)
    margin(tag, "labels, worst scene", max(label_errors), loose_labels)
    margin(tag, "gradients, median", median(gradient_errors), gradient_tolerance)
    margin(tag, "gradients, worst (informative)", max(gradient_errors), 1.0)
    margin(tag, "gradients, same samples", max(fixed_errors), fixed_gradient_tolerance)
    margin(tag, "gradients, same samples, BCE", max(bce_fixed_errors), bce_fixed_tolerance)
    assert median(label_errors) < label_tolerance and max(label_errors) < loose_labels, label_errors
    assert median(gradient_errors) <= gradient_tolerance, gradient_errors
    assert max(fixed_errors) <= fixed_gradient_tolerance, fixed_errors
    assert max(bce_fixed_errors) <= bce_fixed_tolerance, bce_fixed_errors


@pytest.mark.parametrize("N,S,R", [(16, 64, 203), (12, 48, 130), (40, 128, 37), (20, 100, 51)])
def test_two_launch_rows_kernels_redo_the_groups_the_hot_kernel_marks(dev, N, S, R):
    """Round 6: vsrd_render_hierarchical_forward and vsrd_render_backward on box-only fields are two kernels on one
    grid each, like the fused step -- a hot kernel (rotations about y, fixed soft-min shift; instantiated for a full shape at S = 64 / 128
    with more than half of the instance slots used) and a second one for what it could not serve.  The forward has no scratch: the hot
    kernel marks such a group with NaN in its first label, the second kernel finds the mark and writes the group's real labels.  Here
    every third ray looks away from the scene WITHOUT the exact-miss skip: its fine samples are extrapolated to 1e6 m (samplers.py:33),
    the fixed shift cannot serve its group, and both launches take the redo path for a third of their groups -- labels finite and equal
    to one ray per wave to the mappings' tolerance, the adjoint at the saved distances within 2e-4 of one ray per wave."""
    from vsrd_amd import fields, rendering
    from vsrd_amd.rendering import renderers
    T, std, ratio = 0.4, 0.4, 0.4
    sc = _random_scene(500 + N + S, N, R, S, general_rotations=False)
    directions = sc["directions"].clone()
    directions[::3] = torch.nn.functional.normalize(torch.tensor([[0.3, -0.9, -0.4]]), dim=-1)
    uni = dict(u_coarse=sc["u_coarse"].to(dev), u_fine=sc["u_fine"].to(dev))
    lam = torch.randn(R, N, generator=torch.Generator().manual_seed(N)).to(dev)
    results = {}
    for mode in ("rows", "wave"):
        renderers.STEP_WAVE_PER_RAY = mode == "wave"
        try:
            inst = fields.pack_instances(sc["loc"], sc["rot"], sc["dim"]).to(dev).requires_grad_(True)
            out = rendering.render_hierarchical(fields.FieldBlock(inst, T, None, None), sc["origins"].to(dev), directions.to(dev), (0.0, 100.0), S, std, ratio,
                                                skip_exact_misses=False, **uni)
            results[mode] = (out, inst, torch.autograd.grad(out["labels"], inst, grad_outputs=lam, retain_graph=True)[0])
        finally:
            renderers.STEP_WAVE_PER_RAY = False
    rows, wave = results["rows"], results["wave"]
    assert torch.isfinite(rows[0]["labels"]).all() and torch.isfinite(rows[0]["distances"]).all() and torch.isfinite(rows[2]).all()
    assert float(rows[0]["distances"][::3].max()) > 1.0e4                                   # the rays that look away WERE extrapolated
    assert float(rows[0]["labels"][::3].abs().max()) < 1e-6 and float(wave[0]["labels"].max()) > 0.05
    tag = f"test_two_launch_rows_kernels_redo_the_groups_the_hot_kernel_marks[{N}-{S}-{R}]"
    label_error = float((rows[0]["labels"] - wave[0]["labels"]).abs().max())
    margin(tag, "labels vs one ray per wave", label_error, 1e-3)
    assert label_error < 1e-3
    # the adjoint of either mapping at the SAME saved distances (the rows mapping's), same label adjoints
    fixed = {}
    for mode in ("rows", "wave"):
        renderers.STEP_WAVE_PER_RAY = mode == "wave"
        try:
            fixed[mode] = torch.autograd.grad(rows[0]["labels"], rows[1], grad_outputs=lam, retain_graph=True)[0]
        finally:
            renderers.STEP_WAVE_PER_RAY = False
    error = float((fixed["rows"] - fixed["wave"]).abs().max()) / max(float(fixed["wave"].abs().max()), 1e-12)
    margin(tag, "gradients at the same distances vs one ray per wave", error, 2e-4)
    assert error <= 2e-4


@pytest.mark.parametrize("N,S,R", [(8, 32, 203), (16, 64, 64), (40, 100, 37)])
def test_dense_step_honours_the_target_column_map(dev, N, S, R):
    """include/vsrd_hip.h: vsrd_render_config.target_columns is a column map of the TARGETS, independent of ray_indices.  A dense launch
    (no ray_indices) of a shape the multi-ray kernels take (N <= 64, S <= 128) that sets it read targets[row * N + lane] in round 4 -- the
    multi-ray kernels know no column map -- and returned a wrong loss and gradient without an error.  Such launches now go to
    the one-ray kernels: the C ABI call with the map against the same call with the targets permuted on the host (scripts/main.py:653-671:
    labels[..., pd_indices] vs targets[..., gt_indices]); bit-identical to the one-ray kernel, within the mappings' tolerance of the default."""
    from vsrd_amd import _lib, fields
    from vsrd_amd.rendering import renderers
    lib = _lib.load()
    sc = _random_scene(77 + N, N, R, S, general_rotations=False)
    gen = torch.Generator().manual_seed(N)
    M = max(N - 3, 1)                                                      # ground-truth columns: fewer than predictions
    columns = torch.full((N,), -1, dtype=torch.int32)
    columns[torch.randperm(N, generator=gen)[:M]] = torch.randperm(M, generator=gen).to(torch.int32)
    raw = torch.rand(R, M, generator=gen)
    ordered = torch.zeros(R, N)
    ordered[:, columns >= 0] = raw[:, columns[columns >= 0].long()]
    weights = (columns >= 0).float()
    inst = fields.pack_instances(sc["loc"], sc["rot"], sc["dim"]).to(dev).contiguous()
    origins, directions = sc["origins"].to(dev).contiguous(), sc["directions"].to(dev).contiguous()
    workspace = renderers.current_workspace().adjoint(dev, N)

    def launch(targets, column_map, stride, flags):
        loss, grad = torch.empty(1, device=dev), torch.empty_like(inst)
        labels = torch.empty(R, N, device=dev)
        field = _lib.make_field(inst, 0.4)
        config = _lib.make_config(R, S, (0.0, 100.0), 0.4, 0.4, 1.0e-6, 3, seed=3, stream_offset=11, flags=flags,
                                  gather=None if column_map is None else (None, 0, column_map, stride))
        _lib.check(lib.vsrd_render_silhouette_step(field, config, _lib.ptr(origins), _lib.ptr(directions), None, None, _lib.ptr(targets),
                                                   _lib.ptr(weights.to(dev)), 1.0 / (R * M), workspace.data_ptr(), workspace.numel(),
                                                   _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(labels), _lib.stream()))
        return float(loss), grad, labels

    mapped = launch(raw.to(dev).contiguous(), columns.to(dev), M, 0)                                 # whatever the dispatch picks (one ray per wave, or a ray split over two)
    mapped_one_ray = launch(raw.to(dev).contiguous(), columns.to(dev), M, _lib.FLAG_STEP_WAVE_PER_RAY)
    one_ray = launch(ordered.to(dev).contiguous(), None, 0, _lib.FLAG_STEP_WAVE_PER_RAY)
    default = launch(ordered.to(dev).contiguous(), None, 0, 0)                                       # the multi-ray kernels
    assert mapped_one_ray[0] == one_ray[0] and torch.equal(mapped_one_ray[1], one_ray[1]) and torch.equal(mapped_one_ray[2], one_ray[2])
    assert float(mapped[2].max()) > 0.05 and float(mapped[1].abs().max()) > 0
    for got in (mapped, mapped_one_ray):
        assert abs(got[0] - default[0]) <= 1e-4 * abs(default[0]) and (got[2] - default[2]).abs().max() < 1e-3
        assert (got[1] - default[1]).abs().max() <= 5e-3 * float(default[1].abs().max())


@pytest.mark.parametrize("N,S,R,case", [(8, 100, 250, "yaw"), (16, 64, 203, "yaw"), (4, 32, 64, "yaw"), (3, 20, 61, "general"), (64, 128, 37, "yaw"), (5, 64, 97, "tiny_temperature"),
                                        (12, 48, 256, "misses"), (8, 100, 130, "philox"), (1, 17, 1, "yaw"), (33, 65, 3, "misses")])
def test_split_step_matches_wave_per_ray(dev, N, S, R, case):
    """render_silhouette_split_kernel (VSRD_FLAG_STEP_SPLIT_RAY: a ray's rounds divided over the two waves of a workgroup -- what launches of
    <= 2048 gathered rays, the reference's 1000 sampled rays per step, take by themselves) against one ray per wave, both forced by their
    flags on the same dense launch: two and four rounds, general rotations, the running minimum, exact misses (skipped and not), matched-
    instance weights, the in-kernel Philox stream."""
    from vsrd_amd import fields, rendering
    from vsrd_amd.rendering import renderers
    sc = _random_scene(77 + N + S, N, R, S, general_rotations=(case == "general"))
    T, std, ratio = (0.02, 0.3, 0.7) if case == "tiny_temperature" else ((0.1, 0.1, 0.9) if case == "misses" else (0.4, 0.4, 0.4))
    directions = sc["directions"].clone()
    if case == "misses":
        directions[::3] = torch.nn.functional.normalize(torch.tensor([[0.3, -0.9, -0.4]]), dim=-1)
    uni = {} if case == "philox" else dict(u_coarse=sc["u_coarse"].to(dev), u_fine=sc["u_fine"].to(dev))
    pd = torch.arange(0, N, 2, device=dev) if N >= 4 else None
    gt = torch.arange(pd.numel() - 1, -1, -1, device=dev) if pd is not None else None
    targets = sc["targets"][:, :pd.numel()].contiguous() if pd is not None else sc["targets"]
    results = {}
    for mode in ("split", "wave", "split_no_skip"):
        renderers.STEP_WAVE_PER_RAY, renderers.STEP_SPLIT_RAY = mode == "wave", mode != "wave"
        try:
            inst = fields.pack_instances(sc["loc"], sc["rot"], sc["dim"]).to(dev).requires_grad_(True)
            block = fields.FieldBlock(inst, T, None, None)
            loss, labels = rendering.silhouette_step(block, sc["origins"].to(dev), directions.to(dev), targets.to(dev), (0.0, 100.0), S, std, ratio,
                                                     pd_indices=pd, gt_indices=gt, seed=3, stream_offset=11, return_labels=True,
                                                     skip_exact_misses=(mode != "split_no_skip"), **uni)
            results[mode] = (loss.detach(), labels, torch.autograd.grad(loss, inst)[0])
        finally:
            renderers.STEP_WAVE_PER_RAY = renderers.STEP_SPLIT_RAY = False
    split, wave, unskipped = results["split"], results["wave"], results["split_no_skip"]
    assert torch.isfinite(split[2]).all()
    if case != "misses":
        assert float(wave[1].max()) > 0.05                  # the scene is seen
    label_tolerance, gradient_tolerance = (2e-4, 5e-3) if case in ("philox", "misses") else (2e-5, 2e-4)
    if S <= 20:
        gradient_tolerance = 1e-3
    assert (split[1] - wave[1]).abs().max() < label_tolerance
    torch.testing.assert_close(split[0], wave[0], rtol=1e-5 if case != "philox" else 1e-3, atol=1e-7)
    assert (split[2] - wave[2]).abs().max() <= gradient_tolerance * max(float(wave[2].abs().max()), 1e-6)
    # skipping exact misses is exact within one mapping: same kernel, the skipped rays contribute exact zeros
    assert torch.equal(split[1], unskipped[1]) and torch.equal(split[0], unskipped[0]) and torch.equal(split[2], unskipped[2])


def _shape_sweep():
    """24 seeded random shapes over the whole range of the multi-ray kernels (N <= 64, S <= 128), edge sizes included."""
    g = torch.Generator().manual_seed(2024)
    shapes = [(1, 2, 1), (2, 3, 5), (16, 16, 4), (16, 17, 7), (17, 16, 3), (64, 2, 9), (3, 65, 6), (33, 127, 10), (64, 128, 2)]
    while len(shapes) < 24:
        shapes.append((int(torch.randint(1, 65, (1,), generator=g)), int(torch.randint(2, 129, (1,), generator=g)), int(torch.randint(1, 90, (1,), generator=g))))
    return shapes


@pytest.mark.parametrize("N,S,R", _shape_sweep())
def test_multi_ray_kernels_match_one_ray_kernels_over_shapes(dev, N, S, R):
    """The three entry points that have several-rays-per-wave forms on box-only fields -- vsrd_render_silhouette_step,
    vsrd_render_hierarchical_forward (labels / distances) and vsrd_render_backward (label adjoints) -- against their one-ray-per-wave
    forms (VSRD_FLAG_STEP_WAVE_PER_RAY) on seeded random shapes: instance counts on both sides of 16 and 32, sample counts on both sides of
    16 / 32 / 64 and not multiples of the lanes per ray, ray counts that do not fill the last wave.  The parity tolerance of the path
    (silhouettes 1e-4; gradients 5e-3 of the largest entry): the mappings cull different e^-18 terms and the coarse sample counts of
    some shapes make the importance sampler ill-conditioned."""
    from vsrd_amd import fields, rendering
    from vsrd_amd.rendering import renderers
    sc = _random_scene(1000 + 7 * N + S, N, R, S, general_rotations=(N % 3 == 0))
    T, std, ratio = 0.3 + 0.02 * (S % 7), 0.35, 0.5
    lam = torch.randn(R, N, generator=torch.Generator().manual_seed(N + S)).to(dev)
    results = {}
    for mode in ("rows", "wave"):
        renderers.STEP_WAVE_PER_RAY = mode == "wave"
        try:
            inst = fields.pack_instances(sc["loc"], sc["rot"], sc["dim"]).to(dev).requires_grad_(True)
            block = fields.FieldBlock(inst, T, None, None)
            rays = (sc["origins"].to(dev), sc["directions"].to(dev))
            uni = dict(u_coarse=sc["u_coarse"].to(dev), u_fine=sc["u_fine"].to(dev))
            loss, labels = rendering.silhouette_step(block, *rays, sc["targets"].to(dev), (0.0, 100.0), S, std, ratio, return_labels=True, **uni)
            step_grad = torch.autograd.grad(loss, inst)[0]
            out = rendering.render_hierarchical(block, *rays, (0.0, 100.0), S, std, ratio, **uni)
            api_grad = torch.autograd.grad((out["labels"] * lam).sum(), inst)[0]
            results[mode] = (loss.detach(), labels, step_grad, out["labels"].detach(), out["distances"], api_grad)
        finally:
            renderers.STEP_WAVE_PER_RAY = False
    rows, wave = results["rows"], results["wave"]
    assert all(torch.isfinite(t).all() for t in rows[:4]) and torch.isfinite(rows[5]).all()
    assert (rows[1] - wave[1]).abs().max() < LABEL_TOL and (rows[3] - wave[3]).abs().max() < LABEL_TOL
    assert (rows[1] - rows[3]).abs().max() < 2e-6                  # the step and the forward of one mapping see the same silhouettes
    torch.testing.assert_close(rows[0], wave[0], rtol=2e-4, atol=1e-6)
    assert_sampled_distances_close(rows[4], wave[4], S) if S >= 8 else None
    # (two float32 implementations, each within the parity tolerance of the exact answer -- on shape (36, 72, 61) the fused step of the
    #  one-ray kernel is 2.9e-3 from the float64 oracle, the multi-ray one 1.0e-3, the float32 oracle 1.0e-3 --: twice the tolerance apart)
    for a, b in ((rows[2], wave[2]), (rows[5], wave[5])):
        assert (a - b).abs().max() <= 2.0 * GRAD_TOL * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize("N,S,R", [(64, 128, 200), (33, 40, 130), (2, 65, 3)])
def test_residual_step_forms_agree_at_the_size_limits(dev, N, S, R):
    """The three forms of vsrd_render_residual_step (test_residual_step_forms_agree) away from the golden shapes: the largest field and
    sample count the fused step takes (two waves per ray, 60 KB of LDS per workgroup), an odd instance count with a partial last
    round, and a launch with fewer rays than a workgroup has waves.  The one-kernel form is the reference (itself checked against the
    oracle and the goldens at other shapes)."""
    from vsrd_amd import fields, rendering
    from vsrd_amd.rendering import renderers
    sc = _random_scene(5 + N, N, R, S, general_rotations=False)
    g = torch.Generator().manual_seed(N)
    mlp0 = torch.randn(N, 1617, generator=g) * 0.3
    results = {}
    for form in ("default", "single_kernel", "wave_per_ray"):
        renderers.RESIDUAL_SINGLE_KERNEL, renderers.RESIDUAL_WAVE_PER_RAY = form == "single_kernel", form == "wave_per_ray"
        try:
            inst = fields.pack_instances(sc["loc"], sc["rot"], sc["dim"]).to(dev).requires_grad_(True)
            mlp = mlp0.clone().to(dev).requires_grad_(True)
            block = fields.FieldBlock(inst, 0.3, mlp, None)
            loss, terms, labels = rendering.silhouette_step(block, sc["origins"].to(dev), sc["directions"].to(dev), sc["targets"].to(dev),
                                                            (0.0, 100.0), S, 0.3, 0.6, seed=4, stream_offset=9, eikonal_ratio=0.01,
                                                            return_terms=True, return_labels=True)
            results[form] = (loss.detach(), terms, labels, torch.autograd.grad(loss, (inst, mlp)))
        finally:
            renderers.RESIDUAL_SINGLE_KERNEL = renderers.RESIDUAL_WAVE_PER_RAY = False
    reference = results["single_kernel"]
    assert float(reference[2].max()) > 0.05                      # the scene is seen
    for form in ("default", "wave_per_ray"):
        other = results[form]
        assert (other[2] - reference[2]).abs().max() < 2e-6, form
        torch.testing.assert_close(other[1], reference[1], rtol=2e-5, atol=1e-7)
        for a, b in zip(other[3], reference[3]):
            assert (a - b).abs().max() <= 3e-4 * max(float(b.abs().max()), 1e-6), form


@pytest.mark.parametrize("seed", range(8))
def test_culling_bounds_hold_far_from_the_benchmark_scene(dev, seed):
    """The culling bounds are compared in squares with an error term for the quadratic form along the ray; these scenes stress
    what that term has to cover: camera far from the world origin, directions that are not unit vectors, boxes from 0.2 to 30 m,
    one to 64 instances, temperatures from 0.05 to 2.  Default kernels against every-instance / running-minimum / general-rotation
    kernels (a wrong bound would drop an instance that matters)."""
    from vsrd_amd import fields, rendering
    from vsrd_amd.rendering import renderers
    g = torch.Generator().manual_seed(100 + seed)
    N = [1, 2, 5, 16, 33, 64, 7, 12][seed]
    S, R = [8, 33, 64, 64, 32, 16, 100, 64][seed], 96
    T = [0.05, 0.1, 0.3, 0.55, 1.0, 2.0, 0.2, 0.8][seed]
    offset = torch.tensor([[0.0, 0.0, 0.0], [5.0e3, -2.0e3, 1.0e3], [0.0, 0.0, 0.0], [300.0, 10.0, -40.0],
                           [0.0, 0.0, 0.0], [-1.0e3, 0.0, 0.0], [40.0, 2.0, 7.0], [0.0, 0.0, 0.0]][seed])
    scale = [1.0, 1.0, 3.0, 0.5, 1.0, 2.0, 1.0, 0.1][seed]                 # |direction|
    loc = torch.stack([torch.empty(N).uniform_(-20, 20, generator=g), torch.empty(N).uniform_(-2, 3, generator=g),
                       torch.empty(N).uniform_(4, 90, generator=g)], -1) / scale + offset
    dim = torch.exp(torch.empty(N, 3).uniform_(-1.6, 3.4 if seed % 3 == 0 else 1.0, generator=g)) / scale
    yaw = torch.empty(N).uniform_(-3.1, 3.1, generator=g)
    rot = torch.stack([torch.stack([torch.cos(yaw), torch.zeros(N), torch.sin(yaw)], -1), torch.tensor([0.0, 1.0, 0.0]).expand(N, 3),
                       torch.stack([-torch.sin(yaw), torch.zeros(N), torch.cos(yaw)], -1)], -2)
    aim = loc[torch.randint(0, N, (R,), generator=g)] - offset + torch.randn(R, 3, generator=g) * 2.0 / scale
    directions = torch.nn.functional.normalize(aim, dim=-1) * scale
    origins = offset.expand(R, 3).contiguous()
    u_coarse, u_fine = torch.rand(R, S, generator=g), torch.rand(R, S, generator=g)
    lam = torch.randn(R, N, generator=g)
    results = {}
    modes = {"default": (True, False, False), "culling": (True, True, True), "baseline": (False, True, True)}
    for mode, (culling, running, general) in modes.items():
        renderers.CULLING, renderers.RUNNING_MINIMUM, renderers.GENERAL_ROTATIONS = culling, running, general
        try:
            l, d, r = (t.clone().to(dev).requires_grad_(True) for t in (loc, dim, rot))
            union = fields.soft_union([
                rendering.sdfs.translation(rendering.sdfs.rotation(fields.instance_field(rendering.sdfs.box(d[i]), i, N), r[i]), l[i])
                for i in range(N)], T)
            out = rendering.render_hierarchical(union, origins.to(dev), directions.to(dev), (0.0, 100.0 / scale), S, max(T, 0.1), 0.5,
                                                u_coarse=u_coarse.to(dev), u_fine=u_fine.to(dev))
            results[mode] = (out["labels"].detach(), torch.autograd.grad((out["labels"] * lam.to(dev)).sum(), (l, d, r)))
        finally:
            renderers.CULLING, renderers.RUNNING_MINIMUM, renderers.GENERAL_ROTATIONS = True, False, False
    assert torch.isfinite(results["default"][0]).all()
    # culling alone leaves the arithmetic of the surviving terms untouched: tight.  The soft-min floor and the shortened rotations
    # change roundings, which the importance sampler (a division by cdf differences) amplifies on ill-conditioned rays: the
    # parity tolerance of the path.
    for mode, label_tolerance, gradient_tolerance in (("culling", 2e-6, 2e-4), ("default", 1e-4, 5e-3)):
        assert (results[mode][0] - results["baseline"][0]).abs().max() < label_tolerance, mode
        for a, b in zip(results[mode][1], results["baseline"][1]):
            assert torch.isfinite(a).all()
            assert (a - b).abs().max() <= gradient_tolerance * max(float(b.abs().max()), 1e-6), mode


@pytest.mark.parametrize("N,S,R", [(16, 64, 203), (40, 128, 37)])
def test_yaw_gradients_flag_changes_no_parameter_gradient(dev, N, S, R):
    """VSRD_FLAG_YAW_GRADIENTS (FieldBlock.yaw_gradients: the rotations are rotation_matrix_y(cos, sin) of BoxParameters3D,
    box_parameters.py:5-13, and differentiated only through it): the multi-ray step kernels leave out the adjoints of the five matrix
    entries that function keeps constant.  The gradient of the packed block then has exact zeros there, its other entries and every
    PARAMETER gradient (locations, dimensions, the (cos, sin) pairs) are bit-identical to the run without the flag."""
    import bench
    from vsrd_amd import fields, models, rendering
    torch.manual_seed(5)
    det = models.BoxParameters3D(1, N).to(dev)
    K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(3, 1, 64, 64, N)
    with torch.no_grad():
        det.locations.copy_(raw_loc); det.dimensions.copy_(raw_dim); det.orientations.copy_(raw_ori)
    cam, dirs = rendering.ray_casting((64, 64), K.to(dev), E.to(dev))
    directions = dirs.reshape(-1, 3)[:R].contiguous()
    origins = cam[0].expand(R, 3).contiguous()
    targets = torch.rand(R, N, generator=torch.Generator().manual_seed(1)).to(dev)
    results = {}
    for flag in (False, True):
        out = det()
        inst = fields.pack_instances(out["locations"][0], out["orientations"][0], out["dimensions"][0])
        inst.retain_grad()
        block = fields.FieldBlock(inst, 0.4, None, None, yaw_gradients=flag)
        loss = rendering.silhouette_step(block, origins, directions, targets, (0.0, 100.0), S, 0.4, 0.4, seed=2, stream_offset=7)
        grads = torch.autograd.grad(loss, [inst, det.locations, det.dimensions, det.orientations])
        results[flag] = (loss.detach().clone(), [g.clone() for g in grads])
    (loss_a, grads_a), (loss_b, grads_b) = results[False], results[True]
    assert torch.equal(loss_a, loss_b) and float(grads_a[3].abs().max()) > 0
    constant = [4, 6, 7, 8, 10]                                     # r01, r10, r11, r12, r21 of the row-major matrix at columns 3..11
    assert float(grads_a[0][:, constant].abs().max()) > 0 and float(grads_b[0][:, constant].abs().max()) == 0.0
    kept = [c for c in range(16) if c not in constant]
    assert torch.equal(grads_a[0][:, kept], grads_b[0][:, kept])
    for a, b in zip(grads_a[1:], grads_b[1:]):
        assert torch.equal(a, b)
