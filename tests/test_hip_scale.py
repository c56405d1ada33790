"""Size-independent properties at BASELINE.json's full sizes (config 2: 9 x 376x1408 rays, N=16, S=64; config 5 shapes:
N=64, S=128 on a band of a 752x2816 frame) where the CPU oracle cannot follow.  Needs a GPU."""
import pytest
import torch

from conftest import margin

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import __graft_entry__
    __graft_entry__.build()
    return torch.device("cuda:0")


def scene(dev, N, V, H, W, seed=0):
    import bench
    from vsrd_amd import models, rendering
    K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(seed, V, H, W, N)
    cam, dirs = rendering.ray_casting((H, W), K.to(dev), E.to(dev))
    det = models.BoxParameters3D(1, N).to(dev)
    with torch.no_grad():
        det.locations.copy_(raw_loc); det.dimensions.copy_(raw_dim); det.orientations.copy_(raw_ori)
    return det, cam, dirs


def check_properties(out, N, S):
    labels, dist = out["labels"], out["distances"]
    assert torch.isfinite(labels).all()
    assert labels.min() >= -1e-6 and labels.max() <= 1 + 1e-5
    assert labels.sum(-1).max() <= 1 + 1e-4                        # sum_n labels = sum_s w_s <= 1
    rows = ~torch.isnan(dist[:, 0])                                 # NaN sentinel = ray skipped as an exact miss
    assert torch.all(labels[~rows] == 0)
    d = dist[rows]
    assert torch.all(d[:, 1:] >= d[:, :-1])                         # merged distances are sorted
    assert d.shape[1] == 2 * S and d.min() >= 0.0


def test_config2_full_frame_properties(dev):
    import bench
    from vsrd_amd import rendering
    N, S, V, H, W = 16, 64, 9, 376, 1408
    det, cam, dirs = scene(dev, N, V, H, W)
    directions = dirs.reshape(-1, 3)
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
    union = bench.build_union(det, 0.55)
    a = rendering.render_hierarchical(union, origins, directions, (0.0, 100.0), S, 0.55, 0.5, seed=1, stream_offset=7, skip_exact_misses=True)
    assert a["labels"].shape == (V * H * W, N)
    check_properties(a, N, S)
    b = rendering.render_hierarchical(union, origins, directions, (0.0, 100.0), S, 0.55, 0.5, seed=1, stream_offset=7, skip_exact_misses=True)
    assert torch.equal(a["labels"], b["labels"])                                    # deterministic (Philox keyed by ray)
    # backward: linear in the adjoint, deterministic, finite; gradient of a constant shift of all labels is consistent
    lam = torch.randn(a["labels"].shape, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    params = [det.locations, det.dimensions, det.orientations]
    g1 = torch.autograd.grad((a["labels"] * lam).sum(), params, retain_graph=True)
    g2 = torch.autograd.grad((a["labels"] * (2 * lam)).sum(), params, retain_graph=True)
    g3 = torch.autograd.grad((a["labels"] * lam).sum(), params)
    for x, y, z in zip(g1, g2, g3):
        assert torch.isfinite(x).all() and torch.equal(x, z)
        torch.testing.assert_close(2 * x, y, rtol=1e-4, atol=1e-5 * float(x.abs().max()))
    # the hit fraction is sane for this scene (boxes in view)
    assert 0.01 < float((a["labels"].sum(-1) > 0.5).float().mean()) < 0.9


def test_config5_shapes_band(dev):
    """N = 64 instances, S = 128 samples (4 wave rounds, 81 KB of LDS per workgroup) on 16 rows of a 752x2816 view."""
    import bench
    from vsrd_amd import rendering
    N, S, V, H, W = 64, 128, 1, 752, 2816
    det, cam, dirs = scene(dev, N, V, H, W, seed=2)
    rows = dirs[0, 360:376].reshape(-1, 3).contiguous()
    union = bench.build_union(det, 0.55)
    out = rendering.render_hierarchical(union, cam[0], rows, (0.0, 100.0), S, 0.55, 0.5, seed=3, return_weights=True)
    check_properties(out, N, S)
    assert out["weights"].shape == (rows.shape[0], 2 * S - 1) and out["weights"].min() >= 0
    torch.testing.assert_close(out["weights"].sum(-1), out["labels"].sum(-1), rtol=1e-4, atol=1e-5)
    g = torch.autograd.grad(out["labels"].square().sum(), [det.locations, det.dimensions, det.orientations])
    assert all(torch.isfinite(x).all() for x in g) and float(g[0].abs().max()) > 0
    # culling A/B at this size
    from vsrd_amd.rendering import renderers
    renderers.CULLING = False
    try:
        ref = rendering.render_hierarchical(union, cam[0], rows, (0.0, 100.0), S, 0.55, 0.5, seed=3, return_weights=True)    # (the same kernel)
    finally:
        renderers.CULLING = True
    # (the sampler's conditioning, test_multi_ray_step_culling_is_invisible below: all but a few rays within 2e-6, those few within 1e-3)
    moved = (ref["labels"] - out["labels"]).abs().max(-1).values
    margin("test_config5_shapes_band", "rays with labels > 2e-6", float((moved > 2e-6).float().mean()), 1e-3)
    margin("test_config5_shapes_band", "labels, all rays", float(moved.max()), 1e-3)
    assert float((moved > 2e-6).float().mean()) < 1e-3 and float(moved.max()) < 1e-3
    # labels / distances only: two rays per wave (render_hierarchical_pair_kernel) against the one-ray kernel above (same Philox keys; the
    # sorted fine uniforms are partial sums taken in a different order)
    rows_kernel = rendering.render_hierarchical(union, cam[0], rows, (0.0, 100.0), S, 0.55, 0.5, seed=3)
    assert (rows_kernel["labels"] - out["labels"]).abs().max() < 2e-4
    check_properties(rows_kernel, N, S)


def test_config3_shapes_many_instances(dev):
    """Residual field at the size limits: N = 64 instances and S = 100 samples (the reference's own S: 199 points = 4 wave rounds of the
    MLP adjoint).  Properties only: finite, bounded labels, sorted distances, the fused two-pass kernel agrees with rendering at its own
    saved distances, gradients (boxes and MLP weights) are deterministic, and per-tile culling does not change them."""
    import bench
    from vsrd_amd import fields, rendering
    from vsrd_amd.rendering import renderers
    N, S, V, H, W = 64, 100, 1, 376, 1408
    det, cam, dirs = scene(dev, N, V, H, W, seed=4)
    rows = dirs[0, 200:204].reshape(-1, 3)[::4].contiguous()                      # 1408 rays
    boxes = det()
    mlp0 = (torch.randn(N, 1617, generator=torch.Generator().manual_seed(9)) * 0.3).to(dev)

    def run():
        inst = fields.pack_instances(boxes["locations"][0], boxes["orientations"][0], boxes["dimensions"][0]).detach().requires_grad_(True)
        mlp = mlp0.clone().requires_grad_(True)
        block = fields.FieldBlock(inst, 0.3, mlp, None)
        out = rendering.render_hierarchical(block, cam[0], rows, (0.0, 100.0), S, 0.3, 0.7, seed=3, return_gradients=True)
        hit = out["labels"].detach().sum(-1) > 0.5
        loss = (out["labels"] ** 2).sum() + 0.01 * ((out["gradients"][hit].norm(dim=-1) - 1.0) ** 2).mean()
        return block, out, torch.autograd.grad(loss, (inst, mlp))

    block, out, grads = run()
    check_properties(out, N, S)
    assert all(torch.isfinite(g).all() for g in grads) and float(grads[1].abs().max()) > 0
    labels, _, _ = rendering.render_at_distances(block, cam[0], rows, out["distances"], 0.3, 0.7)
    assert (labels - out["labels"]).abs().max() < 1e-5
    _, _, again = run()
    assert all(torch.equal(a, b) for a, b in zip(grads, again))                   # deterministic two-stage reduction
    renderers.CULLING = False
    try:
        _, ref, ref_grads = run()
    finally:
        renderers.CULLING = True
    # (the sampler's conditioning, test_multi_ray_step_culling_is_invisible below: all but a few rays within 2e-6, those few within 1e-3)
    moved = (ref["labels"] - out["labels"]).abs().max(-1).values
    margin("test_config3_shapes_many_instances", "rays with labels > 2e-6", float((moved > 2e-6).float().mean()), 1e-3)
    margin("test_config3_shapes_many_instances", "labels, all rays", float(moved.max()), 1e-3)
    assert float((moved > 2e-6).float().mean()) < 1e-3 and float(moved.max()) < 1e-3
    for a, b in zip(grads, ref_grads):
        assert (a - b).abs().max() <= 2e-4 * max(float(b.abs().max()), 1e-6)


def test_config1_full_frame_against_the_oracle(dev):
    """BASELINE config 1 (1 target + 2 source views, 4 instances, 128 x 128, 32 samples/ray: the reference's CPU-runnable case) at
    its FULL size, all 49 152 rays: the fused step (render + silhouette BCE + adjoint) against the CPU oracle with identical
    uniforms -- silhouettes within 1e-4 max-abs (the north-star bar), the loss and the raw-parameter gradients."""
    import bench
    from oracle import fields as ofields, rendering as orendering, geometry as ogeometry, losses as olosses
    from vsrd_amd import rendering
    N, S, V, H, W = 4, 32, 3, 128, 128
    det, cam, dirs = scene(dev, N, V, H, W, seed=1)
    directions = dirs.reshape(-1, 3)
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
    R = directions.shape[0]
    gen = torch.Generator().manual_seed(0)
    u_coarse, u_fine = torch.rand(R, S, generator=gen), torch.rand(R, S, generator=gen)
    targets = torch.rand(R, N, generator=gen).round()
    T, std, ratio = 0.55, 0.55, 0.5
    loss, labels = rendering.silhouette_step(bench.build_union(det, T), origins, directions, targets.to(dev), (0.0, 100.0), S, std, ratio,
                                             u_coarse=u_coarse.to(dev), u_fine=u_fine.to(dev), return_labels=True, skip_exact_misses=True)
    raw = [det.locations, det.dimensions, det.orientations]
    grads = torch.autograd.grad(loss, raw)
    # oracle, in chunks of image rows (bounded memory), same decode from the raw parameters
    oraw = [p.detach().cpu()[0].clone().requires_grad_(True) for p in raw]
    loc, dim, rot, _ = ogeometry.decode_box_parameters(*oraw)
    union = ofields.InstanceUnion(loc, rot, dim, T)
    olabels, total = [], 0.0
    ograds = [torch.zeros_like(p) for p in oraw]
    chunk = 4096
    for start in range(0, R, chunk):
        sl = slice(start, start + chunk)
        out = orendering.hierarchical_render(union, origins[sl].cpu(), directions[sl].cpu(), (0.0, 100.0), S, std, ratio, u_coarse[sl], u_fine[sl])
        part = torch.nn.functional.binary_cross_entropy(out.labels.clamp(1e-6, 1 - 1e-6), targets[sl], reduction="none").sum() / (R * N)
        for acc, g in zip(ograds, torch.autograd.grad(part, oraw, retain_graph=False)):
            acc += g
        total += float(part.detach())
        olabels.append(out.labels.detach())
        loc, dim, rot, _ = ogeometry.decode_box_parameters(*oraw)            # fresh graph for the next chunk
        union = ofields.InstanceUnion(loc, rot, dim, T)
    olabels = torch.cat(olabels)
    assert (labels.cpu() - olabels).abs().max() < 1e-4
    assert abs(float(loss) - total) < 1e-5 * max(total, 1.0)
    for got, want in zip(grads, ograds):
        assert (got.cpu()[0] - want).abs().max() <= 5e-3 * float(want.abs().max())


def _fused_step_properties(dev, N, S, V, H, W, residual, seed):
    """One optimisation step of the FUSED kernels (the ones bench.py times) on a full-size frame, twice: finite loss and gradients,
    bit-identical repeats (deterministic two-stage reduction, Philox keyed by ray), labels that are probabilities, and agreement of
    the fused loss with the loss recomputed from the returned labels (a checksum over every ray of the frame)."""
    import bench
    from vsrd_amd import models, rendering
    det, cam, dirs = scene(dev, N, V, H, W, seed=seed)
    directions = dirs.reshape(-1, 3)
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
    R = directions.shape[0]
    with torch.no_grad():
        targets = rendering.render_hierarchical(bench.build_union(det, 0.1), origins, directions, (0.0, 100.0), S, 0.1, 1.0, seed=99,
                                                skip_exact_misses=True)["labels"].clamp(0.0, 1.0).contiguous()
        det.locations.add_(0.02)                                     # so that the loss has a gradient
    hyper = None
    if residual:
        torch.manual_seed(0)
        hyper = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).to(dev)
    results = []
    for _ in range(2):
        union = bench.build_union(det, 0.55)
        if hyper is not None:
            union.mlp_weights = hyper(det.embeddings)[0].contiguous()
        loss, terms, labels = rendering.silhouette_step(union, origins, directions, targets, (0.0, 100.0), S, 0.55, 0.5, seed=5, stream_offset=11,
                                                        skip_exact_misses=not residual, eikonal_ratio=0.01 if residual else 0.0,
                                                        return_terms=True, return_labels=True)
        params = [det.locations, det.dimensions, det.orientations] + ([det.embeddings] if residual else [])
        results.append((loss.detach().clone(), terms.clone(), labels, torch.autograd.grad(loss, params)))
    (loss, terms, labels, grads), (loss2, _, labels2, grads2) = results
    assert labels.shape == (R, N) and torch.isfinite(loss) and torch.equal(loss, loss2) and torch.equal(labels, labels2)
    for a, b in zip(grads, grads2):
        assert torch.isfinite(a).all() and torch.equal(a, b) and float(a.abs().max()) > 0
    assert labels.min() >= -1e-6 and labels.sum(-1).max() <= 1 + 1e-4
    recomputed = torch.nn.functional.binary_cross_entropy(labels.clamp(1.0e-6, 1.0 - 1.0e-6), targets, reduction="none").mean()
    torch.testing.assert_close(terms[0], recomputed, rtol=2e-4, atol=1e-7)
    if residual:
        assert 0.0 <= float(terms[1]) < 1.0e3 and abs(float(loss) - float(terms[0] + 0.01 * terms[1])) <= 1e-5 * max(1.0, abs(float(loss)))
    return float(loss)


def test_config5_full_size_fused_step(dev):
    """BASELINE config 5 at its full size on one GPU: 17 views x 752 x 2816 = 36.0 M rays, 64 instances, 128 samples per ray
    (render_silhouette_pair_kernel<4, true, true>; arithmetic pinned by golden g17_render_n64_s128_mid in test_hip_render.py and, at this
    size, against the oracle by test_full_size_parity_against_the_oracle below)."""
    _fused_step_properties(dev, N=64, S=128, V=17, H=752, W=2816, residual=False, seed=2)


@pytest.mark.parametrize("mapping", ["four_rays_per_wave", "one_ray_per_wave"])
def test_config2_full_size_fused_step(dev, mapping):
    """BASELINE config 2 at its full size through the FUSED step -- what bench.py times: 9 views x 376 x 1408 = 4.76 M rays, 16
    instances, 64 samples per ray, in both mappings of vsrd_render_silhouette_step (render_silhouette_quad_kernel<4>, the default, and
    render_silhouette_kernel<2>; arithmetic pinned by golden g4_render_n16_s64_mid and test_quad_step_matches_wave_per_ray).  The two
    mappings share the Philox keys, so their losses over the 4.76 M rays agree to rounding."""
    from vsrd_amd.rendering import renderers
    renderers.STEP_WAVE_PER_RAY = mapping == "one_ray_per_wave"
    try:
        loss = _fused_step_properties(dev, N=16, S=64, V=9, H=376, W=1408, residual=False, seed=4)
    finally:
        renderers.STEP_WAVE_PER_RAY = False
    _config2_losses[mapping] = loss
    if len(_config2_losses) == 2:
        a, b = _config2_losses.values()
        assert abs(a - b) <= 2e-4 * max(abs(a), 1e-6), _config2_losses


_config2_losses = {}


def test_config3_full_size_fused_step(dev):
    """BASELINE config 3 at its full size: 9 views x 376 x 1408 = 4.76 M rays, 16 instances, 64 samples per ray, residual MLP from
    the hypernetwork + eikonal term (render_residual_step_kernel<2>; arithmetic pinned by golden g17_render_residual_n16_s64_mid)."""
    _fused_step_properties(dev, N=16, S=64, V=9, H=376, W=1408, residual=True, seed=3)


def test_chunked_residual_step_adds_up(dev):
    """A residual step whose MLP-adjoint seeds exceed the 6 GiB budget is cut into chunks of rays that accumulate into the same
    partial rows (api.hip: plan_residual_step).  At the reference's S = 100 (four rounds: residual_step_pair_kernel<4>) and N = 16 a
    chunk is 39 321 rays, so 45 056 rays run as two chunks; the one-kernel form (VSRD_FLAG_RESIDUAL_SINGLE_KERNEL) takes them in
    one piece.  Loss, labels and every gradient of the two must agree."""
    import bench
    from vsrd_amd import models, rendering
    from vsrd_amd.rendering import renderers
    N, S, V, H, W = 16, 100, 1, 176, 256
    det, cam, dirs = scene(dev, N, V, H, W, seed=4)
    directions = dirs.reshape(-1, 3)
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
    with torch.no_grad():
        targets = rendering.render_hierarchical(bench.build_union(det, 0.1), origins, directions, (0.0, 100.0), S, 0.1, 1.0, seed=99,
                                                skip_exact_misses=True)["labels"].clamp(0.0, 1.0).contiguous()
        det.locations.add_(0.02)
    torch.manual_seed(0)
    hyper = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).to(dev)
    results = {}
    for form in ("chunked", "single_kernel"):
        renderers.RESIDUAL_SINGLE_KERNEL = form == "single_kernel"
        try:
            union = bench.build_union(det, 0.55)
            union.mlp_weights = hyper(det.embeddings)[0].contiguous()
            loss, terms, labels = rendering.silhouette_step(union, origins, directions, targets, (0.0, 100.0), S, 0.55, 0.5, seed=5, stream_offset=11,
                                                            eikonal_ratio=0.01, return_terms=True, return_labels=True)
            params = [det.locations, det.dimensions, det.orientations, det.embeddings]
            results[form] = (loss.detach().clone(), terms.clone(), labels, torch.autograd.grad(loss, params))
        finally:
            renderers.RESIDUAL_SINGLE_KERNEL = False
    chunked, single = results["chunked"], results["single_kernel"]
    assert torch.isfinite(chunked[0]) and float(chunked[2].max()) > 0.5
    assert (chunked[2] - single[2]).abs().max() < 2e-6
    torch.testing.assert_close(chunked[1], single[1], rtol=2e-5, atol=1e-7)
    for a, b in zip(chunked[3], single[3]):
        assert (a - b).abs().max() <= 5e-4 * max(float(b.abs().max()), 1e-9)


@pytest.mark.parametrize("shape,schedule", [("quad", "start"), ("quad", "mid"), ("quad", "end"), ("pair", "mid")])
def test_multi_ray_step_culling_is_invisible(dev, shape, schedule):
    """The multi-ray step kernels' OWN culling A/B: one full 376 x 1408 view of the benchmark scene, default flags
    against VSRD_FLAG_NO_CULLING -- which also switches off the two early-outs only these kernels have ("rounds that see nothing",
    rounds without weight / flow; quad_step.h) -- at the start / mid / end of the schedules (scripts/main.py:420-431; 477-492: the soft
    union the culling approximates; 653-671: the loss).  quad: render_silhouette_quad_kernel<4> (N = 16, S = 64, what bench.py times);
    pair: render_silhouette_pair_kernel<4> (N = 64, S = 128).  Same Philox keys in both runs.

    What can be demanded of such an A/B.  Culling changes a coarse weight in its last bits (the culled instances' exp(-18)); the importance
    sampler divides by cdf differences + 1e-6 (samplers.py:33), so a few fine samples move by ~1e-5 m; and the box SDF's normal -- hence
    the opacity -- is DISCONTINUOUS across a box's medial planes, so a sample that sits on one flips with such a move and its ray's label
    changes by up to 2e-4 (tests/culling_ab_debug.py: the float64 oracle maps each mode's distances to that mode's labels within 6e-6 on
    exactly those rays; 124 of 529 408 rays at the start of the schedule, 2 at its middle).  That is the reference algorithm's own
    conditioning, not the culling.  So:
      (a) FUSED STEP: labels within 2e-6 on all but a 1e-3 fraction of the rays, every label within 1e-3;
      (b) FUSED STEP: loss within 1e-5 relative, gradients (moved samples included) within 5e-2 of the largest entry -- the BCE's label
          adjoints reach 1 / p = 1e6, so ONE ray with a moved sample shifts a sum over 5e5 rays by percents (observed 8e-4 with one build
          of the box norm and 1.7e-2 with another on the pair shape; what pins the adjoint code is (c));
      (c) THE SAME SWEEP / ADJOINT CODE AT FIXED SAMPLES (render_backward_{quad,pair}_kernel = the step's forward sweep, reverse sweep and
          per-instance phase on the distances a forward launch saved, same label adjoints): gradients within 2e-4 of the largest entry."""
    import bench
    from vsrd_amd import rendering
    from vsrd_amd.rendering import renderers
    N, S = (16, 64) if shape == "quad" else (64, 128)
    H, W = 376, 1408
    sched = bench.schedule_values(bench.SCHEDULES[schedule])
    T, std, ratio = sched["temperature"], sched["std"], sched["cosine_ratio"]
    det, cam, dirs = scene(dev, N, 1, H, W, seed=0)
    directions = dirs.reshape(-1, 3)
    origins = cam[:, None, None, :].expand(1, H, W, 3).reshape(-1, 3).contiguous()
    with torch.no_grad():
        targets = rendering.render_hierarchical(bench.build_union(det, 0.1), origins, directions, (0.0, 100.0), S, 0.1, 1.0, seed=99,
                                                skip_exact_misses=True)["labels"].clamp(0.0, 1.0).contiguous()
        det.locations.add_(0.02)
    params = [det.locations, det.dimensions, det.orientations]
    step, forward = {}, {}
    for mode in ("default", "no_culling"):
        renderers.CULLING = mode == "default"
        try:
            loss, labels = rendering.silhouette_step(bench.build_union(det, T), origins, directions, targets, (0.0, 100.0), S, std, ratio,
                                                     seed=5, stream_offset=11, return_labels=True)
            step[mode] = (float(loss.detach()), labels, torch.autograd.grad(loss, params))
            if mode == "default":
                forward[mode] = rendering.render_hierarchical(bench.build_union(det, T), origins, directions, (0.0, 100.0), S, std, ratio, seed=5,
                                                              stream_offset=11, skip_exact_misses=True)
        finally:
            renderers.CULLING = True
    tag = f"test_multi_ray_step_culling_is_invisible[{shape}-{schedule}]"
    (loss_a, labels_a, grads_a), (loss_b, labels_b, grads_b) = step["default"], step["no_culling"]
    assert float(labels_b.max()) > 0.5 and all(float(g.abs().max()) > 0 for g in grads_b)
    diff = (labels_a - labels_b).abs().max(-1).values
    outliers = float((diff > 2e-6).float().mean())
    margin(tag, "rays with labels > 2e-6", outliers, 1e-3)
    margin(tag, "labels, all rays", float(diff.max()), 1e-3)
    margin(tag, "loss (relative)", abs(loss_a - loss_b) / max(abs(loss_b), 1e-12), 1e-5)
    step_grad_err = max(float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12) for a, b in zip(grads_a, grads_b))
    margin(tag, "step gradients / largest", step_grad_err, 5e-2)
    assert outliers < 1e-3 and float(diff.max()) < 1e-3
    assert abs(loss_a - loss_b) <= 1e-5 * abs(loss_b) and step_grad_err < 5e-2
    # (c) the adjoint at the default mode's saved samples, with and without culling, for the BCE label adjoints of the default labels
    out = forward["default"]
    probabilities = out["labels"].detach().clamp(1.0e-6, 1.0 - 1.0e-6).requires_grad_(True)
    bce = torch.nn.functional.binary_cross_entropy(probabilities, targets, reduction="none").mean()
    lam = torch.autograd.grad(bce, probabilities)[0] * ((out["labels"].detach() >= 1e-6) & (out["labels"].detach() <= 1 - 1e-6))
    fixed = {}
    for mode in ("default", "no_culling"):
        renderers.CULLING = mode == "default"
        try:
            fixed[mode] = torch.autograd.grad(out["labels"], params, grad_outputs=lam, retain_graph=True)
        finally:
            renderers.CULLING = True
    fixed_err = max(float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12) for a, b in zip(fixed["default"], fixed["no_culling"]))
    margin(tag, "gradients, same samples", fixed_err, 2e-4)
    assert all(float(g.abs().max()) > 0 for g in fixed["no_culling"]) and fixed_err < 2e-4


# ---- round 5: the benched kernels against the ORACLE at full size ------------------------------------------------------------------
def _oracle_union(det, temperature, dtype):
    from oracle import fields as ofields, geometry as ogeometry
    raw = [p.detach().to(dtype).cpu()[0] for p in (det.locations, det.dimensions, det.orientations)]
    loc, dim, rot, _ = ogeometry.decode_box_parameters(*raw)
    return ofields.InstanceUnion(loc, rot, dim, temperature)


def _stable_under_float32_noise(evaluate, inputs, which, reference, bound=1.0e-5, trials=4, ulps=4.0, seed=99):
    """Rays on which the EXACT (float64) evaluation does not move by more than `bound` when input number `which` (sample distances, or ray
    directions) is perturbed by a few float32 units in the last place, `trials` random perturbations: the rays on which float32 ARITHMETIC
    determines the result.  (A sample within an ulp of a box's medial plane takes the other face's normal, a fine uniform within an ulp of a
    plateau of the importance sampler lands in another bin: on such rays any two float32 implementations of renderers.py:212-263 may differ by
    1e-1, the reference's own with a different summation order included.  Agreement of ONE float32 and ONE float64 evaluation -- VERDICT r05's
    definition -- is necessary, not sufficient: at config 5, 64 overlapping boxes, a ray in ten sits that close to a kink somewhere.)"""
    generator = torch.Generator().manual_seed(seed)
    stable = torch.ones(reference.shape[0], dtype=torch.bool)
    for _ in range(trials):
        noisy = list(inputs)
        noise = (torch.rand(noisy[which].shape, generator=generator, dtype=torch.float64) * 2.0 - 1.0) * ulps * 2.0 ** -23
        noisy[which] = noisy[which].double() * (1.0 + noise)
        stable &= (evaluate(*noisy).double() - reference.double()).abs().flatten(1).max(-1).values <= bound
    return stable


def _in_chunks(function, tensors, chunk):
    """function(*rows of every tensor) over chunks of rows (bounded memory on the host), results concatenated."""
    parts = [function(*(t[start:start + chunk] for t in tensors)) for start in range(0, tensors[0].shape[0], chunk)]
    if isinstance(parts[0], tuple):
        return tuple(torch.cat([p[k] for p in parts]) for k in range(len(parts[0])))
    return torch.cat(parts)


@pytest.mark.parametrize("config", ["config2", "config5"])
def test_full_size_parity_against_the_oracle(dev, config):
    """BASELINE configs 2 and 5 at their FULL size on one GPU, the fused step bench.py times (vsrd_render_silhouette_step:
    render_silhouette_quad_kernel<4, true, true> / render_silhouette_pair_kernel<4, true, true>) pinned against the CPU oracle link by
    link (scripts/main.py:511-523, vsrd/rendering/samplers.py:24-36, renderers.py:212-263).

    Rays: a seeded random draw of the frame PLUS the rays on which the step's own culling A/B (default flags against
    VSRD_FLAG_NO_CULLING) moves a label by more than 2e-6 -- the tail test_multi_ray_step_culling_is_invisible describes (the largest
    first when there are more than the cap; config 2: 4096 + up to 4096; config 5, whose oracle costs 16x as much per ray: 1024 + up
    to 1024 -- round 6 halved both and spends the time on the float64 noise trials and the gradients below; VSRD_PARITY_RAYS=<n> overrides
    both numbers for a patient run).

    Round 6.  (a) HARD bounds on the rays where float32 arithmetic determines the result: the float32 and the float64
    oracle agree to 1e-5 AND the float64 oracle does not move by more than 1e-5 under random perturbations of a few float32 ulps of the
    sample distances (pass 2) / the ray directions (end to end) -- `_stable_under_float32_noise`; agreement of one float32 and one float64
    evaluation alone is not enough at config 5 (a determinate-looking ray 4.4e-4 away, first attempt).  Pass 2 at the step's samples: EVERY
    such ray within 2e-5 (observed 1.7e-5 / 9.2e-6 at configs 2 / 5, 99 % / 91 % of the rays); end to end: every such ray within 1e-4 at
    config 2 (observed 2.6e-5, none beyond), at most 0.5 % of them at config 5 (observed none of 55 %, worst 1.9e-5).  (c) Loss and parameter
    gradients at the step's own samples: vsrd_render_forward + vsrd_render_backward on the determinate rays at the exported distances with
    the BCE's label adjoints against the oracle's autograd (double backward through the normal included): loss within 1e-5 relative
    (observed 2.6e-7 / 6e-8), every parameter gradient within 5e-3 of the largest entry of the float64 oracle's, or three times the float32
    oracle's own distance from it (config 5, orientations: the float32 oracle itself is 8e-3 away).

    Every link is checked on the STEP LAUNCH'S OWN state: vsrd_render_config::out_* (ABI 7) makes vsrd_render_silhouette_step write, next
    to its labels, pass 1's weights, the uniforms it drew and the sorted pass-2 distances its labels, loss and gradients were computed at.
    (A separate vsrd_render_hierarchical_forward launch with the same keys is NOT that state: it is another instantiation of the same
    source, rounds differently in places, and the sampler -- see the last item -- turns that into other samples on 2e-4 of the rays.)
      (pass 1)   float32 oracle weights at the kernel's stratified distances (from its u_coarse) vs the kernel's coarse weights
      (sampler)  oracle.importance_distances(kernel coarse distances, KERNEL coarse weights, kernel sorted uniforms), merged and sorted,
                 vs the kernel's pass-2 distances: at most 1e-4 of the samples off by more than 5 mm (observed 9e-6), at most 1e-3 of the
                 rays with a sample off by more than 2 % of a coarse bin
      (pass 2)   float32 oracle.render_given_distances at the KERNEL's distances vs the step's labels
                 Both passes per ray, and both NEXT TO what float32 does to the oracle itself (float32 oracle vs float64 oracle on the same
                 rays): the kernel may be no farther from the float32 oracle than that oracle is from the exact one -- share of rays beyond
                 1e-5 within 1e-3 of the oracle's own share, worst ray within the oracle's own worst (or 2e-4) -- and the MEDIAN ray within
                 2e-6.  Observed, config 2: kernel vs float32 oracle 1.2e-4 of the rays beyond 1e-5, worst 1.7e-5, median 1.6e-7 (VERDICT r04
                 asked for 1e-5 on every ray), where the float32 oracle is 1.8e-4 from the float64 one on its worst ray.  Config 5 (64
                 overlapping boxes): kernel vs float32 oracle 3 % of the rays beyond 1e-5, worst 5.8e-3 -- and float32 oracle vs float64
                 oracle 9 %, worst 3.2e-2: at fixed samples float32 ARITHMETIC does not determine these silhouettes to 1e-4 (a sample on a
                 box's medial plane takes the other normal), for the reference's own float32 code as for this one.
      (end to end, reported + bounded) the whole oracle pipeline on the kernel's uniforms, float32 and float64: the fraction of selected rays
                 with |fused step - float32 oracle| > 1e-4 next to the same fraction for float32 oracle vs float64 oracle -- "the tail is the
                 algorithm's own conditioning (the sampler's / (delta cdf + 1e-6), the box normal's jumps)" as two numbers in the margin
                 table; the HIP fraction may not exceed the oracles' own by more than 1e-3 of the rays (config 5: 2.7 % / 4.5 % of the
                 rays for the kernel against the float32 / float64 oracle, 4.7 % for the float32 oracle against the float64 one).
    Exact misses among the selected rays (NaN sentinel): the float64 oracle's labels there are below 1e-6."""
    import os
    import bench
    from oracle import rendering as orendering
    from vsrd_amd import rendering
    from vsrd_amd.rendering import renderers
    N, S, V, H, W, seed, budget = (16, 64, 9, 376, 1408, 0, 4096) if config == "config2" else (64, 128, 17, 752, 2816, 2, 1024)
    budget = int(os.environ.get("VSRD_PARITY_RAYS", budget))
    sched = bench.schedule_values(bench.SCHEDULES["mid"])
    T, std, ratio = sched["temperature"], sched["std"], sched["cosine_ratio"]
    det, cam, dirs = scene(dev, N, V, H, W, seed=seed)
    directions = dirs.reshape(-1, 3)
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
    R = directions.shape[0]
    tag = f"test_full_size_parity_against_the_oracle[{config}]"
    with torch.no_grad():
        targets = rendering.render_hierarchical(bench.build_union(det, 0.1), origins, directions, (0.0, 100.0), S, 0.1, 1.0, seed=99,
                                                skip_exact_misses=True)["labels"].clamp(0.0, 1.0).contiguous()
        det.locations.add_(0.02)
        keys = dict(seed=5, stream_offset=11)
        # (a) the fused step's culling A/B, then the step itself (default flags: what bench.py times) with its samples
        renderers.CULLING = False
        try:
            _, unculled = rendering.silhouette_step(bench.build_union(det, T), origins, directions, targets, (0.0, 100.0), S, std, ratio,
                                                    return_labels=True, **keys)
        finally:
            renderers.CULLING = True
        _, step_labels, samples = rendering.silhouette_step(bench.build_union(det, T), origins, directions, targets, (0.0, 100.0), S, std, ratio,
                                                            return_labels=True, return_samples=True, **keys)
        all_targets = targets
        del targets
        moved_by = (step_labels - unculled).abs().max(-1).values
        del unculled
        moved = torch.nonzero(moved_by > 2e-6).flatten()
        margin(tag, "rays the culling A/B moves", moved.numel() / R, 1e-3)
        if moved.numel() > budget:
            moved = moved[torch.topk(moved_by[moved], budget).indices]
        # (b) the selection
        draw = torch.randint(0, R, (budget,), generator=torch.Generator().manual_seed(1234)).to(dev)
        selection = torch.unique(torch.cat([draw, moved]))
        selected_targets = all_targets[selection].cpu()
        del all_targets
        hip_labels = step_labels[selection].cpu()
        hip_distances, hip_coarse_weights, u_coarse, u_fine = (samples[k][selection].cpu() for k in ("distances", "coarse_weights", "u_coarse", "u_fine"))
        del samples, step_labels
    o, d = origins[selection].cpu(), directions[selection].cpu()
    missed = torch.isnan(hip_distances[:, 0])
    hit = ~missed
    assert torch.all(hip_labels[missed] == 0) and int(hit.sum()) > selection.numel() // 20
    assert torch.all(u_fine[:, 1:] >= u_fine[:, :-1])                                        # (drawn in the kernel: exported sorted)
    union32, union64 = _oracle_union(det, T, torch.float32), _oracle_union(det, T, torch.float64)
    chunk = 1024 if N <= 16 else 256
    failures = []

    def held(what, got, want, exact, tight=1.0e-5):
        """Per ray: the kernel (`got`) against the float32 oracle (`want`), next to the float32 oracle against the float64 one (`exact`) --
        what float32 arithmetic itself does to this quantity on these rays (a sample on a box's medial plane takes the other normal in the
        other precision: single rays differ by 1e-1 at config 5, where 64 boxes overlap).  Demanded: the kernel is no farther from the
        float32 oracle than that oracle is from the exact one -- the share of rays beyond `tight` within 1e-3 of the oracle's own share,
        the worst ray within the oracle's own worst (or 2e-4)."""
        mine = (got.double() - want.double()).abs().flatten(1).max(-1).values
        own = (want.double() - exact.double()).abs().flatten(1).max(-1).values
        share_mine, share_own = float((mine > tight).float().mean()), float((own > tight).float().mean())
        margin(tag, what + f": rays > {tight:g}", share_mine, share_own + 1.0e-3)
        margin(tag, what + ": worst ray", float(mine.max()), max(2.0e-4, float(own.max())))
        margin(tag, what + ": oracle f32/f64 share", share_own, 1.0)
        margin(tag, what + ": oracle f32/f64 worst", float(own.max()), 1.0)
        if not (share_mine <= share_own + 1.0e-3 and float(mine.max()) <= max(2.0e-4, float(own.max()))):
            failures.append((what, share_mine, share_own, float(mine.max()), float(own.max())))
        return mine

    with torch.no_grad():
        # ---- pass 1: the kernel's coarse weights against the oracle at the kernel's stratified distances ----
        coarse_distances = orendering.stratified_distances((0.0, 100.0), S, u_coarse)
        oracle_coarse32 = _in_chunks(lambda a, b, c: orendering.render_given_distances(union32, a, b, c, std, ratio).weights, (o, d, coarse_distances), chunk)
        oracle_coarse64 = _in_chunks(lambda a, b, c: orendering.render_given_distances(union64, a.double(), b.double(), c.double(), std, ratio).weights,
                                     (o, d, coarse_distances), chunk)
        first = held("pass 1", hip_coarse_weights, oracle_coarse32, oracle_coarse64)
        margin(tag, "pass 1: median ray", float(first.median()), 2e-6)
        assert torch.all(hip_coarse_weights[missed] == 0) and float(first.median()) < 2e-6
        # ---- sampler, fed with the KERNEL's coarse weights ----
        fine = orendering.importance_distances(coarse_distances[hit], hip_coarse_weights[hit], u_fine[hit])
        merged = torch.sort(torch.cat([coarse_distances[hit], fine], dim=-1), dim=-1).values
        # (test_hip_render.assert_sampled_distances_close bounds EVERY sample of a few hundred rays by 2 % of a coarse bin; over the 10^6 samples
        #  here single samples land in another bin -- where the cdf is flat across several bins the search's answer hangs on its last bit;
        #  config 5: one sample 0.59 m away -- so: the share of samples off by more than 5 mm, and of rays with a sample off by more than 2 % of a bin)
        displaced = (hip_distances[hit] - merged).abs()
        off = float((displaced > 5e-3 + 1e-4 * merged.abs()).float().mean())
        rays_off = float((displaced.max(-1).values > 0.02 * 100.0 / S).float().mean())
        margin(tag, "samples off by > 5e-3 m", off, 1e-4)
        margin(tag, "rays with a sample off > 2 % bin", rays_off, 1e-3)
        margin(tag, "largest sample displacement", float(displaced.max()), 100.0)
        assert off <= 1e-4 and rays_off <= 1e-3
        # ---- pass 2 at the kernel's own samples ----
        fixed32 = _in_chunks(lambda a, b, c: orendering.render_given_distances(union32, a, b, c, std, ratio).labels, (o[hit], d[hit], hip_distances[hit]), chunk)
        fixed64 = _in_chunks(lambda a, b, c: orendering.render_given_distances(union64, a.double(), b.double(), c.double(), std, ratio).labels,
                             (o[hit], d[hit], hip_distances[hit]), chunk)
        at_samples = held("pass 2", hip_labels[hit], fixed32, fixed64)
        margin(tag, "pass 2: median ray", float(at_samples.median()), 2e-6)
        assert float(at_samples.median()) < 2e-6
        # a HARD bound where the reference's arithmetic is determinate -- the rays on which the float32 and the float64
        # oracle agree to 1e-5 at these samples: EVERY such ray within 2e-5 of the float32 oracle (= the reference's own arithmetic)
        determinate = (fixed32.double() - fixed64).abs().max(-1).values <= 1.0e-5
        determinate &= _stable_under_float32_noise(
            lambda a, b, c: _in_chunks(lambda x, y, z: orendering.render_given_distances(union64, x.double(), y.double(), z.double(), std, ratio).labels, (a, b, c), chunk),
            (o[hit], d[hit], hip_distances[hit]), 2, fixed64)
        worst_determinate = float(at_samples[determinate].max())
        margin(tag, "pass 2: share of determinate rays", float(determinate.float().mean()), 1.0)
        margin(tag, "pass 2, determinate rays: worst", worst_determinate, 2e-5)
        assert float(determinate.float().mean()) > (0.75 if N <= 16 else 0.5) and worst_determinate <= 2e-5       # (config 5: half of its selection are the culling A/B's outliers)
        # ---- loss and parameter gradients at the step's own samples (scripts/main.py:653-671 through renderers.py:212-263,
        # autograd's double backward through the SDF normal included): vsrd_render_forward + vsrd_render_backward on the selected rays at the
        # step's exported distances, BCE against the frame's own targets, against the float32 oracle's autograd on the same rays and samples
    with torch.enable_grad():
        # (over the DETERMINATE rays: a ray whose labels float32 arithmetic does not determine -- a sample on a medial plane -- has label adjoints
        #  of 1 / p size and a Hessian that jumps; and next to the float32 oracle's own distance from the float64 oracle's gradient, as everywhere)
        from oracle import fields as ofields, geometry as ogeometry
        raw_names = ("locations", "dimensions", "orientations")
        rows_index = torch.nonzero(hit).flatten()[determinate]
        rows = int(rows_index.numel())
        rows_dev = rows_index.to(dev)
        rays_o, rays_d = origins[selection][rows_dev], directions[selection][rows_dev]
        hip_params = [getattr(det, n) for n in raw_names]
        at_labels, _, _ = rendering.render_at_distances(bench.build_union(det, T), rays_o, rays_d, hip_distances[rows_index].to(dev), std, ratio)
        hip_loss = torch.nn.functional.binary_cross_entropy(at_labels.clamp(1.0e-6, 1.0 - 1.0e-6), selected_targets[rows_index].to(dev), reduction="none").mean()
        hip_grads = [g.detach().cpu()[0] for g in torch.autograd.grad(hip_loss, hip_params)]

        def oracle_gradients(dtype):
            raws = [getattr(det, n).detach().cpu()[0].to(dtype).clone().requires_grad_(True) for n in raw_names]
            grads, loss = [torch.zeros_like(r) for r in raws], 0.0
            for start in range(0, rows, chunk):        # (autograd state of a chunk: rays x (2S - 1) points x N instances, twice differentiated)
                index = rows_index[start:start + chunk]
                loc, dim, rot, _ = ogeometry.decode_box_parameters(*raws)
                out = orendering.render_given_distances(ofields.InstanceUnion(loc, rot, dim, T), o[index].to(dtype), d[index].to(dtype), hip_distances[index].to(dtype), std, ratio)
                part = torch.nn.functional.binary_cross_entropy(out.labels.clamp(1.0e-6, 1.0 - 1.0e-6), selected_targets[index].to(dtype), reduction="none").sum() / (rows * N)
                for total, g in zip(grads, torch.autograd.grad(part, raws)):
                    total += g
                loss += float(part.detach())
            return grads, loss

        (grads32, loss32), (grads64, loss64) = oracle_gradients(torch.float32), oracle_gradients(torch.float64)
        loss_error = abs(float(hip_loss) - loss32) / max(abs(loss32), 1e-12)
        margin(tag, "loss at the step's samples (relative)", loss_error, 1e-5)
        margin(tag, "loss at the step's samples: f32 vs f64 oracle", abs(loss32 - loss64) / max(abs(loss64), 1e-12), 1.0)
        assert loss_error <= 1e-5, (float(hip_loss), loss32, loss64)
        for name, got, want, exact in zip(raw_names, hip_grads, grads32, grads64):
            # against the EXACT gradient, next to the float32 oracle's own distance from it (config 5: the float32 oracle's orientation gradient is
            # 8e-3 of the largest entry away from the float64 one -- 64 overlapping boxes, label adjoints of 1 / p size): within 5e-3, or 3 x that
            scale = max(float(exact.abs().max()), 1e-12)
            error, own = float((got.double() - exact).abs().max()) / scale, float((want.double() - exact).abs().max()) / scale
            margin(tag, f"grad {name} at the step's samples vs f64 oracle / largest entry", error, max(5e-3, 3.0 * own))
            margin(tag, f"grad {name}: HIP vs f32 oracle / largest entry", float((got - want).abs().max()) / scale, 1.0)
            margin(tag, f"grad {name}: f32 vs f64 oracle / largest entry", own, 1.0)
            assert scale > 0 and error <= max(5e-3, 3.0 * own), (name, error, own)
    with torch.no_grad():
        # ---- end to end on the kernel's uniforms: the tail, as numbers ----
        whole32 = _in_chunks(lambda a, b, c, e: orendering.hierarchical_render(union32, a, b, (0.0, 100.0), S, std, ratio, c, e).labels, (o, d, u_coarse, u_fine), chunk)
        whole64 = _in_chunks(lambda a, b, c, e: orendering.hierarchical_render(union64, a.double(), b.double(), (0.0, 100.0), S, std, ratio, c.double(), e.double()).labels,
                             (o, d, u_coarse, u_fine), chunk)
    hip_tail = float(((hip_labels - whole32).abs().max(-1).values > 1e-4).float().mean())
    oracle_tail = float(((whole32.double() - whole64).abs().max(-1).values > 1e-4).float().mean())
    hip_tail64 = float(((hip_labels.double() - whole64).abs().max(-1).values > 1e-4).float().mean())
    margin(tag, "rays > 1e-4: HIP vs f32 oracle", hip_tail, max(oracle_tail, 0.0) + 1e-3)
    margin(tag, "rays > 1e-4: f32 vs f64 oracle", oracle_tail, 1.0)
    margin(tag, "rays > 1e-4: HIP vs f64 oracle", hip_tail64, max(oracle_tail, 0.0) + 1e-3)
    margin(tag, "worst ray, HIP vs f32 oracle", float((hip_labels - whole32).abs().max()), 1.0)
    margin(tag, "worst ray, f32 vs f64 oracle", float((whole32.double() - whole64).abs().max()), 1.0)
    margin(tag, "median ray, HIP vs f32 oracle", float((hip_labels - whole32).abs().max(-1).values.median()), 1e-5)
    # VERDICT r05 item 4a, end to end: the rays on which the float32 and the float64 oracle agree to 1e-5 THROUGH THE WHOLE PIPELINE (their fine
    # samples fell on the same side of every plateau, no normal flipped): the north star's "silhouettes within 1e-4" asserted on EVERY such ray
    determinate_whole = (whole32.double() - whole64).abs().max(-1).values <= 1.0e-5
    with torch.no_grad():
        determinate_whole &= _stable_under_float32_noise(
            lambda a, b, c, e: _in_chunks(lambda x, y, z, w: orendering.hierarchical_render(union64, x.double(), y.double(), (0.0, 100.0), S, std, ratio, z.double(), w.double()).labels,
                                          (a, b, c, e), chunk), (o, d, u_coarse, u_fine), 1, whole64, trials=4 if N <= 16 else 8, ulps=4.0 if N <= 16 else 16.0)
    whole_errors = (hip_labels - whole32).abs().max(-1).values
    margin(tag, "end to end: share of determinate rays", float(determinate_whole.float().mean()), 1.0)
    margin(tag, "end to end, determinate rays: worst", float(whole_errors[determinate_whole].max()), 1e-4)
    margin(tag, "end to end, determinate rays beyond 1e-4", float((whole_errors[determinate_whole] > 1e-4).sum()), 0.5)
    assert not failures, failures
    # Config 2 (the metric's workload): EVERY determinate ray within 1e-4.  Config 5 (64 overlapping boxes, 255 points per ray: half of the rays
    # have a sample or a fine uniform within float32 noise of a kink SOMEWHERE along the pipeline, and no finite number of noise trials certifies
    # a ray): the rays that survive eight trials of 16 ulps, at most 0.5 % of them beyond 1e-4 (observed: 3 of 1130 with four trials of 4 ulps).
    beyond = float((whole_errors[determinate_whole] > 1e-4).float().mean())
    margin(tag, "end to end, determinate rays: share beyond 1e-4", beyond, 0.0 if N <= 16 else 5e-3)
    if N <= 16:
        assert float(determinate_whole.float().mean()) > 0.75 and float(whole_errors[determinate_whole].max()) <= 1e-4
    else:
        assert float(determinate_whole.float().mean()) > 0.25 and beyond <= 5e-3
    assert hip_tail <= oracle_tail + 1e-3 and hip_tail64 <= oracle_tail + 1e-3
    assert float((hip_labels - whole32).abs().max(-1).values.median()) < 1e-5
    assert float(whole64[missed].abs().max()) < 1e-6 if bool(missed.any()) else True
    print(f"{tag}: {selection.numel()} rays ({int(moved.numel())} from the culling A/B, {int(missed.sum())} exact misses); > 1e-4 end to end: "
          f"HIP vs f32 oracle {hip_tail:.2e}, f32 vs f64 oracle {oracle_tail:.2e}, HIP vs f64 oracle {hip_tail64:.2e}")


@pytest.mark.parametrize("mlp_products", ["fp32_mfma", "split_bf16"])
def test_config3_full_size_parity_against_the_oracle(dev, mlp_products):
    """BASELINE config 3 at its FULL size (9 x 376 x 1408 = 4.76 M rays, 16 instances, 64 samples, residual MLP from the hypernetwork +
    eikonal term): the fused residual step (vsrd_render_residual_step: residual_step_front_kernel<2> + residual_mlp_adjoint_kernel, and their
    split-bf16 twins of split_front.hip) against the CPU oracle on a seeded draw of the frame's rays (scripts/main.py:433-458, 511-523,
    629-687; vsrd/models/fields/hyper_distance_field.py:57-73; samplers.py:24-36; renderers.py:212-263).

    Round 6: LINK BY LINK on the step's own samples, like test_full_size_parity_against_the_oracle -- the residual step
    exports them too (vsrd_render_config::out_*, ABI 8; residual_step_front_kernel<2, true>): pass 1's weights against the float32 and float64
    oracle at the stratified distances, the importance sampler fed with the KERNEL's coarse weights, pass 2 (per-instance MLPs included) at the
    KERNEL's distances, and the whole pipeline end to end on the same uniforms.  Criteria as there: per link, the kernel no farther from the
    float32 oracle than that oracle is from the float64 one (share of rays beyond 1e-5 within 1e-3 of the oracle's own, worst ray within the
    oracle's own worst or 2e-4), the median ray within 2e-6; end to end the share beyond 1e-4 within the oracles' own + 1e-3 (round 5 allowed
    2.5 rays of 640 on top); and the HARD bounds on the determinate rays (`_stable_under_float32_noise`): every one within 2e-5 at fixed samples
    and within 1e-4 end to end.  Then (item 4c) the loss and the gradients -- boxes and the 16 MLPs' weights -- of vsrd_render_forward +
    vsrd_render_backward at the step's samples against the oracle's autograd on determinate rays.
    1024 rays by default (three quarters of them rays that see something); VSRD_PARITY_RAYS=4096 is the round's patient run
    (profiles/r06/parity_config3_4096_rays.log: 97 % of the rays determinate end to end, none beyond 1.2e-5; gradients 1.8e-6 ... 7.6e-5)."""
    import os
    import bench
    from oracle import fields as ofields, geometry as ogeometry, rendering as orendering
    from vsrd_amd import models, rendering
    N, S, V, H, W, seed = 16, 64, 9, 376, 1408, 3
    # (round 6: 1024 rays by default -- every ray costs the oracle ten passes of 190 points x 16 MLPs with tangents; the round's patient run,
    #  VSRD_PARITY_RAYS=4096, 19 minutes on a 256-core box, is profiles/r06/parity_config3_4096_rays.log: no determinate ray beyond 1.2e-5)
    budget = int(os.environ.get("VSRD_PARITY_RAYS", 1024))
    noise_trials = 2 if budget >= 4096 else 1
    sched = bench.schedule_values(bench.SCHEDULES["mid"])
    T, std, ratio = sched["temperature"], sched["std"], sched["cosine_ratio"]
    torch.manual_seed(0)                 # (the detector's embeddings -- the hypernetwork's input -- are drawn at construction: the same field in every run)
    det, cam, dirs = scene(dev, N, V, H, W, seed=seed)
    directions = dirs.reshape(-1, 3)
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
    R = directions.shape[0]
    tag = f"test_config3_full_size_parity_against_the_oracle[{mlp_products}]"
    hyper = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).to(dev)
    with torch.no_grad():
        targets = rendering.render_hierarchical(bench.build_union(det, 0.1), origins, directions, (0.0, 100.0), S, 0.1, 1.0, seed=99,
                                                skip_exact_misses=True)["labels"].clamp(0.0, 1.0).contiguous()
        det.locations.add_(0.02)
        generator = torch.Generator(device=dev).manual_seed(77)
        u_coarse = torch.rand(R, S, device=dev, generator=generator)
        u_fine = torch.rand(R, S, device=dev, generator=generator)
        union = bench.build_union(det, T)
        weights = hyper(det.embeddings)[0].contiguous()
        union.mlp_weights = weights
        _, labels, samples = rendering.silhouette_step(union, origins, directions, targets, (0.0, 100.0), S, std, ratio, u_coarse=u_coarse, u_fine=u_fine,
                                                       eikonal_ratio=0.01, return_labels=True, return_samples=True, skip_exact_misses=False,
                                                       mlp_split_bf16=mlp_products == "split_bf16")
        all_targets = targets
        del targets
        # three quarters of the selection from the rays that see something (label > 1e-3: a uniform draw of this frame is mostly sky), a quarter from all
        assert torch.isfinite(labels).all() and float(labels.max()) > 0.5
        pick = torch.Generator().manual_seed(4321)
        lit = torch.nonzero(labels.max(-1).values > 1e-3).flatten()
        assert lit.numel() > R // 50
        selection = torch.unique(torch.cat([lit[torch.randint(0, lit.numel(), (3 * budget // 4,), generator=pick).to(dev)],
                                            torch.randint(0, R, (budget // 4,), generator=pick).to(dev)]))
        hip_labels = labels[selection].cpu()
        selected_targets = all_targets[selection].cpu()
        hip_distances, hip_coarse_weights, samples_u_coarse, samples_u_fine = (samples[k][selection].cpu() for k in ("distances", "coarse_weights", "u_coarse", "u_fine"))
        margin(tag, "selected rays that see something", float((hip_labels.max(-1).values > 1e-3).float().mean()), 1.0)
        del labels, samples, all_targets
        o, d, uc, uf = origins[selection].cpu(), directions[selection].cpu(), u_coarse[selection].cpu(), u_fine[selection].cpu()
        del u_coarse, u_fine

        def oracle_union(dtype):
            raw = [p.detach().to(dtype).cpu()[0] for p in (det.locations, det.dimensions, det.orientations)]
            loc, dim, rot, _ = ogeometry.decode_box_parameters(*raw)
            field = ofields.InstanceUnion(loc, rot, dim, T)
            field.mlp_weights = weights.detach().to(dtype).cpu()
            return field

        union32, union64 = oracle_union(torch.float32), oracle_union(torch.float64)
        chunk = 128
        failures = []

        def held(what, got, want, exact, tight=1.0e-5):
            """As in test_full_size_parity_against_the_oracle: the kernel against the float32 oracle next to the float32 oracle against the float64 one."""
            mine = (got.double() - want.double()).abs().flatten(1).max(-1).values
            own = (want.double() - exact.double()).abs().flatten(1).max(-1).values
            share_mine, share_own = float((mine > tight).float().mean()), float((own > tight).float().mean())
            margin(tag, what + f": rays > {tight:g}", share_mine, share_own + 1.0e-3)
            margin(tag, what + ": worst ray", float(mine.max()), max(2.0e-4, float(own.max())))
            margin(tag, what + ": oracle f32/f64 share", share_own, 1.0)
            margin(tag, what + ": oracle f32/f64 worst", float(own.max()), 1.0)
            if not (share_mine <= share_own + 1.0e-3 and float(mine.max()) <= max(2.0e-4, float(own.max()))):
                failures.append((what, share_mine, share_own, float(mine.max()), float(own.max())))
            return mine

        # ---- link by link on the STEP'S OWN samples (vsrd_render_config::out_* of vsrd_render_residual_step, ABI 8) ----
        # (the caller's own uniforms come back as they were given -- raw; the kernel sorts its copy, samplers.py:22)
        sorted_fine = torch.sort(uf, dim=-1).values
        assert torch.equal(samples_u_coarse, uc) and (torch.equal(samples_u_fine, uf) or torch.equal(samples_u_fine, sorted_fine))
        coarse_distances = orendering.stratified_distances((0.0, 100.0), S, uc)
        coarse32 = _in_chunks(lambda a, b, c: orendering.render_given_distances(union32, a, b, c, std, ratio).weights, (o, d, coarse_distances), chunk)
        coarse64 = _in_chunks(lambda a, b, c: orendering.render_given_distances(union64, a.double(), b.double(), c.double(), std, ratio).weights, (o, d, coarse_distances), chunk)
        first = held("pass 1", hip_coarse_weights, coarse32, coarse64)
        margin(tag, "pass 1: median ray", float(first.median()), 2e-6)
        assert float(first.median()) < 2e-6
        fine = orendering.importance_distances(coarse_distances, hip_coarse_weights, sorted_fine)
        merged = torch.sort(torch.cat([coarse_distances, fine], dim=-1), dim=-1).values
        sees = hip_coarse_weights.sum(-1) > 0                                  # (rays without any coarse weight: their fine samples are extrapolated to 1e6 m, samplers.py:33)
        displaced = (hip_distances[sees] - merged[sees]).abs()
        off = float((displaced > 5e-3 + 1e-4 * merged[sees].abs()).float().mean())
        margin(tag, "samples off by > 5e-3 m", off, 1e-4)
        margin(tag, "rays with a sample off > 2 % bin", float((displaced.max(-1).values > 0.02 * 100.0 / S).float().mean()), 1e-3)
        assert off <= 1e-4
        fixed32 = _in_chunks(lambda a, b, c: orendering.render_given_distances(union32, a, b, c, std, ratio).labels, (o, d, hip_distances), chunk)
        fixed64 = _in_chunks(lambda a, b, c: orendering.render_given_distances(union64, a.double(), b.double(), c.double(), std, ratio).labels, (o, d, hip_distances), chunk)
        at_samples = held("pass 2", hip_labels, fixed32, fixed64)
        margin(tag, "pass 2: median ray", float(at_samples.median()), 2e-6)
        determinate = (fixed32.double() - fixed64).abs().max(-1).values <= 1.0e-5
        determinate &= _stable_under_float32_noise(
            lambda a, b, c: _in_chunks(lambda x, y, z: orendering.render_given_distances(union64, x.double(), y.double(), z.double(), std, ratio).labels, (a, b, c), chunk),
            (o, d, hip_distances), 2, fixed64, trials=noise_trials)
        margin(tag, "pass 2: share of determinate rays", float(determinate.float().mean()), 1.0)
        margin(tag, "pass 2, determinate rays: worst", float(at_samples[determinate].max()), 2e-5)
        assert float(at_samples.median()) < 2e-6 and float(determinate.float().mean()) > 0.75 and float(at_samples[determinate].max()) <= 2e-5
        whole32 = _in_chunks(lambda a, b, c, e: orendering.hierarchical_render(union32, a, b, (0.0, 100.0), S, std, ratio, c, e).labels, (o, d, uc, uf), chunk)
        whole64 = _in_chunks(lambda a, b, c, e: orendering.hierarchical_render(union64, a.double(), b.double(), (0.0, 100.0), S, std, ratio, c.double(), e.double()).labels,
                             (o, d, uc, uf), chunk)
    mine32 = (hip_labels - whole32).abs().max(-1).values
    mine64 = (hip_labels.double() - whole64).abs().max(-1).values
    own = (whole32.double() - whole64).abs().max(-1).values
    hip_tail, hip_tail64, oracle_tail = (float((e > 1e-4).float().mean()) for e in (mine32, mine64, own))
    # (a SHARE of 1e-3 needs a few thousand rays to mean anything: with the default 1024 one ray is 1e-3 -- a single non-determinate ray beyond 1e-4
    #  is allowed there; the 4096-ray run holds the plain oracle's-own + 1e-3: 4.9e-4 against 4.9e-4 + 1e-3)
    slack = max(1e-3, 1.5 / selection.numel())
    margin(tag, "rays > 1e-4: HIP vs f32 oracle", hip_tail, oracle_tail + slack)
    margin(tag, "rays > 1e-4: HIP vs f64 oracle", hip_tail64, oracle_tail + slack)
    margin(tag, "rays > 1e-4: f32 vs f64 oracle", oracle_tail, 1.0)
    margin(tag, "rays > 1e-5: HIP vs f32 oracle", float((mine32 > 1e-5).float().mean()), 1.0)
    margin(tag, "rays > 1e-5: f32 vs f64 oracle", float((own > 1e-5).float().mean()), 1.0)
    margin(tag, "worst ray, HIP vs f32 oracle", float(mine32.max()), 1.0)
    margin(tag, "worst ray, f32 vs f64 oracle", float(own.max()), 1.0)
    margin(tag, "median ray, HIP vs f32 oracle", float(mine32.median()), 1e-5)
    margin(tag, "median ray, f32 vs f64 oracle", float(own.median()), 1.0)
    determinate_whole = own <= 1.0e-5
    with torch.no_grad():
        determinate_whole &= _stable_under_float32_noise(
            lambda a, b, c, e: _in_chunks(lambda x, y, z, w: orendering.hierarchical_render(union64, x.double(), y.double(), (0.0, 100.0), S, std, ratio, z.double(), w.double()).labels,
                                          (a, b, c, e), chunk), (o, d, uc, uf), 1, whole64, trials=noise_trials)
    margin(tag, "end to end: share of determinate rays", float(determinate_whole.float().mean()), 1.0)
    margin(tag, "end to end, determinate rays: worst", float(mine32[determinate_whole].max()), 1e-4)
    print(f"{tag}: {selection.numel()} rays; > 1e-4 end to end: HIP vs f32 oracle {hip_tail:.2e}, f32 vs f64 oracle {oracle_tail:.2e}, HIP vs f64 oracle {hip_tail64:.2e}; "
          f"median {float(mine32.median()):.2e}, worst {float(mine32.max()):.2e} (oracle's own {float(own.max()):.2e})")
    assert not failures, failures
    # (round 5 allowed 2.5 rays of 640 here; now: the float32 oracle's own share + 1e-3, or ONE ray where the selection is too small for 1e-3 to be one)
    assert hip_tail <= oracle_tail + slack and hip_tail64 <= oracle_tail + slack
    assert float(mine32.median()) < 1e-5
    assert float(determinate_whole.float().mean()) > 0.75 and float(mine32[determinate_whole].max()) <= 1e-4
    # ---- loss and gradients (boxes AND the MLPs' weights) at the step's own samples, on a subset the oracle's double backward
    # through 16 MLPs affords: vsrd_render_forward + vsrd_render_backward against the float32 oracle's autograd
    # (over rays that are determinate at these samples, as in test_full_size_parity_against_the_oracle: one ray with a sample on a medial plane -- label
    #  adjoints of 1 / p size, a Hessian that jumps -- moved the location gradient of a 384-ray subset by 2.5 % of its largest entry)
    candidates = torch.nonzero(determinate).flatten()
    subset = candidates[::max(candidates.numel() // int(os.environ.get("VSRD_PARITY_GRADIENT_RAYS", 384)), 1)]
    with torch.enable_grad():
        raw_names = ("locations", "dimensions", "orientations")
        hip_weights = weights.detach().clone().requires_grad_(True)
        field = bench.build_union(det, T)
        field.mlp_weights = hip_weights
        sub_dev = selection[subset.to(dev)]
        at_labels, _, _ = rendering.render_at_distances(field, origins[sub_dev], directions[sub_dev], hip_distances[subset].to(dev), std, ratio)
        sub_targets = selected_targets[subset]
        hip_loss = torch.nn.functional.binary_cross_entropy(at_labels.clamp(1.0e-6, 1.0 - 1.0e-6), sub_targets.to(dev), reduction="none").mean()
        hip_grads = [g.detach().cpu() for g in torch.autograd.grad(hip_loss, [*(getattr(det, n) for n in raw_names), hip_weights])]
        hip_grads = [g[0] for g in hip_grads[:3]] + [hip_grads[3]]
        rows = int(subset.numel())

        def oracle_gradients(dtype):
            raws = [getattr(det, n).detach().cpu()[0].to(dtype).clone().requires_grad_(True) for n in raw_names]
            oracle_weights = weights.detach().cpu().to(dtype).clone().requires_grad_(True)
            leaves = [*raws, oracle_weights]
            grads, loss = [torch.zeros_like(t) for t in leaves], 0.0
            for start in range(0, rows, 64):
                index = subset[start:start + 64]
                loc, dim, rot, _ = ogeometry.decode_box_parameters(*raws)
                union = ofields.InstanceUnion(loc, rot, dim, T)
                union.mlp_weights = oracle_weights
                out = orendering.render_given_distances(union, o[index].to(dtype), d[index].to(dtype), hip_distances[index].to(dtype), std, ratio)
                part = torch.nn.functional.binary_cross_entropy(out.labels.clamp(1.0e-6, 1.0 - 1.0e-6), selected_targets[index].to(dtype), reduction="none").sum() / (rows * N)
                for total, g in zip(grads, torch.autograd.grad(part, leaves)):
                    total += g
                loss += float(part.detach())
            return grads, loss

        (grads32, loss32), (grads64, loss64) = oracle_gradients(torch.float32), oracle_gradients(torch.float64)
        loss_error = abs(float(hip_loss) - loss32) / max(abs(loss32), 1e-12)
        margin(tag, "loss at the step's samples (relative)", loss_error, 1e-5)
        assert loss_error <= 1e-5, (float(hip_loss), loss32, loss64)
        for name, got, want, exact in zip((*raw_names, "mlp weights"), hip_grads, grads32, grads64):
            # against the EXACT gradient, next to the float32 oracle's own distance from it (as in test_full_size_parity_against_the_oracle)
            scale = max(float(exact.abs().max()), 1e-12)
            error, own = float((got.double() - exact).abs().max()) / scale, float((want.double() - exact).abs().max()) / scale
            # (a few hundred rays: the BCE's 1 / p label adjoints let single rays carry a gradient entry -- the float32 oracle's own orientation
            #  gradient is 1.5e-3 of its largest entry away from the float64 one on 256 rays, 7e-5 on the 4096-ray run's subset)
            margin(tag, f"grad {name} at the step's samples vs f64 oracle / largest entry", error, max(5e-3, 4.0 * own))
            margin(tag, f"grad {name}: HIP vs f32 oracle / largest entry", float((got - want).abs().max()) / scale, 1.0)
            margin(tag, f"grad {name}: f32 vs f64 oracle / largest entry", own, 1.0)
            assert error <= max(5e-3, 4.0 * own), (name, error, own)

