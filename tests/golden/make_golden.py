#!/usr/bin/env python3
"""Golden-vector generator for the VSRD hot path.

Runs ONLY in the build container, where the read-only reference checkout is
mounted at /root/reference.  It imports the reference's own hot-path modules
(through a namespace stub, so the package __init__ files that need
torchvision / cv2 are skipped), feeds them seeded synthetic inputs and writes
small ``.npz`` fixtures next to this file.  The fixtures are *data* (inputs,
recorded randomness, outputs, parameter gradients); no reference source
travels with them.

The per-instance field closures of ``scripts/main.py:433-523`` live inside
``train()`` and cannot be imported; the helper closures below restate their
composition (instance one-hot features, temperature soft-min union, residual
composition, two-pass wrapper) around the *reference's* ``sdfs.*`` and
``hierarchical_volumetric_rendering`` so that every arithmetic op that exists
in an importable reference module is executed by the reference itself.

Usage:  python tests/golden/make_golden.py   (writes tests/golden/*.npz)
"""
import math
import os
import sys
import types
import importlib

sys.dont_write_bytecode = True  # never drop __pycache__ into /root/reference

import numpy as np
import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("VSRD_REFERENCE_ROOT", "/root/reference")
OUT_DIR = os.path.dirname(os.path.abspath(__file__))


def _stub_package(name, path):
    module = types.ModuleType(name)
    module.__path__ = [path]
    sys.modules[name] = module
    return module


def import_reference():
    """Import reference sub-modules without executing the package __init__s."""
    root = os.path.join(REFERENCE_ROOT, "vsrd")
    _stub_package("vsrd", root)
    for sub in ("rendering", "operations", "models", "models/detectors", "models/fields", "models/encoders"):
        _stub_package("vsrd." + sub.replace("/", "."), os.path.join(root, sub))
    ref = types.SimpleNamespace()
    ref.utils = importlib.import_module("vsrd.utils")
    ref.renderers = importlib.import_module("vsrd.rendering.renderers")
    ref.samplers = importlib.import_module("vsrd.rendering.samplers")
    ref.sdfs = importlib.import_module("vsrd.rendering.sdfs")
    ref.rutils = importlib.import_module("vsrd.rendering.utils")
    ref.geo = importlib.import_module("vsrd.operations.geometric_operations")
    ref.k360 = importlib.import_module("vsrd.operations.kitti360_operations")
    ref.box_parameters = importlib.import_module("vsrd.models.detectors.box_parameters")
    ref.hyper = importlib.import_module("vsrd.models.fields.hyper_distance_field")
    ref.encoder = importlib.import_module("vsrd.models.encoders.sinusoidal_encoder")
    return ref


# ---------------------------------------------------------------------------
# restated composition of scripts/main.py:433-523 (closures, not importable)
# ---------------------------------------------------------------------------

DISTANCE_RANGE = (0.0, 100.0)  # configs/.../config.json:226-229


def make_instance_field(distance_field, instance_index, num_instances):
    # scripts/main.py:460-475
    def field(positions):
        distances = distance_field(positions)
        labels = nn.functional.one_hot(torch.tensor(instance_index, dtype=torch.long), num_instances)
        labels = labels.expand(*distances.shape[:-1], -1)
        return distances, labels
    return field


def make_soft_union(fields, temperature):
    # scripts/main.py:477-492
    def field(positions):
        distances, labels = map(torch.stack, zip(*[f(positions) for f in fields]))
        weights = nn.functional.softmin(distances / temperature, dim=0)
        return torch.sum(distances * weights, dim=0), torch.sum(labels * weights, dim=0)
    return field


def make_residual_field(ref_hyper_module, encoder, mlp_weights):
    # scripts/main.py:433-449 (+ functools.partial at :541-544)
    def field(positions):
        x, y, z = torch.unbind(positions, dim=-1)
        folded = torch.stack([torch.abs(x), y, z], dim=-1) / max(DISTANCE_RANGE)
        return torch.sigmoid(ref_hyper_module.distance_field(mlp_weights, encoder(folded)) - 1.0)
    return field


def make_residual_composition(distance_field, residual_field):
    # scripts/main.py:451-458
    def field(positions):
        return distance_field(positions) + residual_field(positions)
    return field


def two_pass(renderer, **kwargs):
    # scripts/main.py:511-523
    with torch.no_grad():
        *_, coarse_distances, coarse_weights = renderer(**kwargs)
    kwargs.update(sampled_distances=coarse_distances, sampled_weights=coarse_weights)
    outputs = renderer(**kwargs)
    return coarse_distances, coarse_weights, outputs


def rotation_y(angle):
    c, s = np.cos(angle), np.sin(angle)
    return torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=torch.float32)


def kitti_like_intrinsics(height, width):
    # KITTI-360 perspective camera scaled to the image size.
    sx, sy = width / 1408.0, height / 376.0
    return torch.tensor([
        [552.554261 * sx, 0.0, 682.049453 * sx],
        [0.0, 552.554261 * sy, 238.769549 * sy],
        [0.0, 0.0, 1.0],
    ], dtype=torch.float32)


def save(name, **arrays):
    out = {}
    for key, value in arrays.items():
        if isinstance(value, torch.Tensor):
            value = value.detach().cpu().numpy()
        out[key] = np.asarray(value)
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ---------------------------------------------------------------------------
# G1: ray casting
# ---------------------------------------------------------------------------

def golden_ray_casting(ref):
    arrays = {}
    for tag, (h, w) in {"small": (2, 3), "mid": (24, 40)}.items():
        K = kitti_like_intrinsics(376, 1408) if tag == "small" else kitti_like_intrinsics(h, w)
        E = torch.eye(4)
        E[:3, :3] = rotation_y(0.05) @ torch.tensor(
            [[1.0, 0.0, 0.0], [0.0, np.cos(0.02), -np.sin(0.02)], [0.0, np.sin(0.02), np.cos(0.02)]], dtype=torch.float32)
        E[:3, 3] = torch.tensor([0.3, -0.1, 1.5])
        Ks = torch.stack([K, K * torch.tensor([[1.01], [0.99], [1.0]])])
        Es = torch.stack([E, torch.eye(4)])
        cam, dirs = ref.rutils.ray_casting((h, w), Ks, Es)
        arrays.update({f"{tag}_K": Ks, f"{tag}_E": Es, f"{tag}_hw": np.array([h, w]),
                       f"{tag}_camera_positions": cam, f"{tag}_ray_directions": dirs})
    save("g1_ray_casting", **arrays)


# ---------------------------------------------------------------------------
# G2/G3: SDF primitives, normals, soft union
# ---------------------------------------------------------------------------

def scene_instances(num_instances, generator, z_range=(8.0, 40.0)):
    """Boxes inside the reference's decode ranges (box_parameters.py:23-30)."""
    loc = torch.stack([
        torch.empty(num_instances).uniform_(-8.0, 8.0, generator=generator),
        torch.empty(num_instances).uniform_(0.2, 1.2, generator=generator),
        torch.empty(num_instances).uniform_(*z_range, generator=generator),
    ], dim=-1)
    dim = torch.stack([
        torch.empty(num_instances).uniform_(0.75, 1.0, generator=generator),
        torch.empty(num_instances).uniform_(0.75, 1.0, generator=generator),
        torch.empty(num_instances).uniform_(1.5, 2.5, generator=generator),
    ], dim=-1)
    yaw = torch.empty(num_instances).uniform_(-np.pi, np.pi, generator=generator)
    rot = torch.stack([rotation_y(float(a)) for a in yaw])
    return loc, dim, rot


def golden_sdf(ref):
    g = torch.Generator().manual_seed(11)
    # known answers of SURVEY.md §4
    dim = torch.tensor([1.0, 1.0, 2.0])
    pts = torch.tensor([[0, 0, 0], [0.5, 0, 0], [2, 0, 0], [2, 3, 0], [1, 1, 2], [0, 0, 2.5]], dtype=torch.float32)
    known = ref.sdfs.box(dim)(pts)

    loc, dims, rot = scene_instances(4, g)
    # points: random cloud around the boxes + points on faces/edges/inside
    cloud = loc[torch.randint(0, 4, (96,), generator=g)] + torch.randn(96, 3, generator=g) * 2.0
    special = torch.cat([
        loc,                                                         # centres (inside, arg-max ties possible -> offset)
        loc + rot @ torch.tensor([0.3, 0.1, -0.2]),                  # inside, off-centre
        loc + (rot @ (dims * torch.tensor([1.0, 0.0, 0.0])).unsqueeze(-1)).squeeze(-1) * 1.5,  # outside +x face
        loc + (rot @ (dims * torch.tensor([1.2, 1.3, 0.0])).unsqueeze(-1)).squeeze(-1),        # outside an edge
        loc + (rot @ (dims * torch.tensor([1.5, 1.4, 1.3])).unsqueeze(-1)).squeeze(-1),        # outside a corner
    ])
    special[:4] += torch.tensor([0.011, 0.007, 0.003])
    points = torch.cat([cloud, special]).requires_grad_(True)

    per_instance_d, per_instance_g = [], []
    fields = []
    for i in range(4):
        sdf = ref.sdfs.translation(ref.sdfs.rotation(ref.sdfs.box(dims[i]), rot[i]), loc[i])
        d = sdf(points)
        gr, = torch.autograd.grad(d, points, torch.ones_like(d))
        per_instance_d.append(d.detach())
        per_instance_g.append(gr)
        fields.append(make_instance_field(
            ref.sdfs.box(dims[i]), i, 4))
    arrays = dict(known_dim=dim, known_points=pts, known_distances=known,
                  locations=loc, dimensions=dims, orientations=rot, points=points.detach(),
                  instance_distances=torch.stack(per_instance_d), instance_gradients=torch.stack(per_instance_g))

    for temperature in (1.0, 0.1):
        union = make_soft_union([
            ref.sdfs.translation(ref.sdfs.rotation(make_instance_field(ref.sdfs.box(dims[i]), i, 4), rot[i]), loc[i])
            for i in range(4)
        ], temperature)
        u, w = union(points)
        gu, = torch.autograd.grad(u, points, torch.ones_like(u))
        tag = f"T{temperature:g}".replace(".", "p")
        arrays.update({f"union_{tag}_distances": u.detach(), f"union_{tag}_labels": w.detach(), f"union_{tag}_gradients": gu})
    save("g2_g3_sdf_union", **arrays)


# ---------------------------------------------------------------------------
# G4: hierarchical volumetric rendering with recorded randomness
# ---------------------------------------------------------------------------

def build_rays(height, width, rows, cols, extrinsic=None):
    K = kitti_like_intrinsics(height, width)
    E = torch.eye(4) if extrinsic is None else extrinsic
    refmod = sys.modules["vsrd.rendering.utils"]
    cam, dirs = refmod.ray_casting((height, width), K[None], E[None])
    dirs = dirs[0][rows][:, cols].reshape(-1, 3)
    cams = cam[0].expand_as(dirs)
    return cams.contiguous(), dirs.contiguous()


def render_case(ref, name, num_instances, num_samples, temperature, std, cosine_ratio, seed,
                with_residual=False, eikonal_weight=0.0, z_range=(8.0, 40.0), ray_grid=(12, 20), row_range=None, split_gradients=False):
    g = torch.Generator().manual_seed(seed)
    loc, dims, rot = scene_instances(num_instances, g, z_range=z_range)
    loc = loc.clone().requires_grad_(True)
    dims = dims.clone().requires_grad_(True)
    rot = rot.clone().requires_grad_(True)

    H, W = 128, 128
    rows = torch.linspace(*(row_range or (0, H - 1)), ray_grid[0]).long()
    cols = torch.linspace(0, W - 1, ray_grid[1]).long()
    origins, directions = build_rays(H, W, rows, cols)
    num_rays = origins.shape[0]

    mlp_weights = None
    encoder = None
    hyper = None
    if with_residual:
        hyper = ref.hyper.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256])
        encoder = ref.encoder.SinusoidalEncoder(8)
        mlp_weights = (torch.randn(num_instances, 1617, generator=g) * 0.25).requires_grad_(True)

    def make_field():
        fields = []
        for i in range(num_instances):
            base = ref.sdfs.box(dims[i])
            if with_residual:
                base = make_residual_composition(base, make_residual_field(hyper, encoder, mlp_weights[i]))
            fields.append(ref.sdfs.translation(ref.sdfs.rotation(make_instance_field(base, i, num_instances), rot[i]), loc[i]))
        return make_soft_union(fields, temperature)

    field = make_field()
    kwargs = dict(distance_field=field, ray_positions=origins, ray_directions=directions,
                  distance_range=DISTANCE_RANGE, num_samples=num_samples,
                  sdf_std_deviation=std, cosine_ratio=cosine_ratio)

    # ---- pass 1 (no grad), randomness recorded by re-seeding and re-drawing --------
    torch.manual_seed(seed + 1)
    with torch.no_grad():
        c_labels, c_grads, c_dists, c_weights = ref.renderers.hierarchical_volumetric_rendering(**kwargs)
    torch.manual_seed(seed + 1)
    u_coarse = torch.rand(num_rays, 1, num_samples)  # == rand_like(bins[..., :-1]) (samplers.py:6)
    bins = torch.linspace(*DISTANCE_RANGE, num_samples + 1)
    redo = torch.lerp(bins[:-1].expand(num_rays, 1, -1), bins[1:].expand(num_rays, 1, -1), u_coarse)
    assert torch.equal(redo.permute(2, 0, 1), c_dists), "coarse uniforms were not recovered"

    # ---- pass 2 (grad) -----------------------------------------------------------
    torch.manual_seed(seed + 2)
    f_labels, f_grads, f_dists, f_weights = ref.renderers.hierarchical_volumetric_rendering(
        **kwargs, sampled_distances=c_dists, sampled_weights=c_weights)
    torch.manual_seed(seed + 2)
    u_fine = torch.rand(num_rays, 1, num_samples)  # samplers.py:21 (sorted there at :22)

    # silhouette targets: perturbed render at smaller std (deterministic function of labels)
    targets = (torch.rand(num_rays, num_instances, generator=g) < 0.5).float() * 0.9 + 0.05
    bce = nn.functional.binary_cross_entropy(f_labels.clamp(1.0e-6, 1.0 - 1.0e-6), targets, reduction="none").mean()
    eikonal = nn.functional.mse_loss(torch.norm(f_grads, dim=-1), f_grads.new_ones(f_grads.shape[:-1]))
    # Rays with all-zero coarse weights get fine samples extrapolated to ~1e6 m (samplers.py:33), where
    # fp32 cancellation makes the reference's own SDF gradients (and their parameter gradients) rounding
    # noise.  The differentiated loss therefore takes the eikonal term (main.py:679-687) over the
    # well-conditioned rays only; the full-tensor value is kept for a magnitude check.
    conditioned = c_weights.sum(0)[..., 0] > 0
    eikonal_conditioned = nn.functional.mse_loss(
        torch.norm(f_grads[:, conditioned], dim=-1), f_grads.new_ones(f_grads[:, conditioned].shape[:-1]))
    loss = bce + eikonal_weight * eikonal_conditioned
    params = [loc, dims, rot] + ([mlp_weights] if with_residual else [])
    grads = torch.autograd.grad(loss, params, retain_graph=split_gradients)

    arrays = dict(
        locations=loc, dimensions=dims, orientations=rot,
        origins=origins, directions=directions,
        num_samples=np.array(num_samples), temperature=np.array(temperature, dtype=np.float32),
        sdf_std_deviation=np.array(std, dtype=np.float32), cosine_ratio=np.array(cosine_ratio, dtype=np.float32),
        eikonal_weight=np.array(eikonal_weight, dtype=np.float32),
        u_coarse=u_coarse[:, 0, :], u_fine=u_fine[:, 0, :],
        coarse_labels=c_labels, coarse_gradients=c_grads, coarse_distances=c_dists[..., 0], coarse_weights=c_weights[..., 0],
        fine_labels=f_labels, fine_gradients=f_grads, fine_distances=f_dists[..., 0], fine_weights=f_weights[..., 0],
        targets=targets, bce=bce, eikonal=eikonal, eikonal_conditioned=eikonal_conditioned, loss=loss,
        grad_locations=grads[0], grad_dimensions=grads[1], grad_orientations=grads[2],
    )
    if with_residual:
        arrays.update(mlp_weights=mlp_weights, grad_mlp_weights=grads[3])
    if split_gradients:
        # the two terms separately: a fused step over the well-conditioned rays alone weighs them differently (means over fewer rays)
        keys = ["locations", "dimensions", "orientations"] + (["mlp_weights"] if with_residual else [])
        for tag, term in (("bce", bce), ("eikonal", eikonal_conditioned)):
            if term.grad_fn is None:
                continue
            for key, value in zip(keys, torch.autograd.grad(term, params, retain_graph=True, allow_unused=True)):
                arrays[f"grad_{tag}_{key}"] = torch.zeros_like(params[keys.index(key)]) if value is None else value
        arrays["conditioned"] = conditioned
    hit = int((f_labels.sum(-1) > 1e-3).sum())
    miss = int((c_weights.sum(0)[..., 0] == 0).sum())
    print(f"  {name}: rays={num_rays} hit={hit} exact-miss={miss} loss={float(loss):.6f}")
    save(name, **arrays)


def golden_rendering(ref):
    # C1-sized: N=4, S=32; step-0 schedule and a mid/late schedule; hit, graze and miss rays.
    render_case(ref, "g4_render_n4_s32_step0", 4, 32, 1.0, 1.0, 0.0, seed=100)
    render_case(ref, "g4_render_n4_s32_mid", 4, 32, 0.55, 0.55, 0.5, seed=101, eikonal_weight=0.01)
    render_case(ref, "g4_render_n4_s32_late", 4, 32, 0.1, 0.1, 1.0, seed=102, z_range=(6.0, 25.0))
    # non-power-of-two sample count and instance count (ragged wave rounds)
    render_case(ref, "g4_render_n3_s20_mid", 3, 20, 0.4, 0.3, 0.7, seed=103, eikonal_weight=0.01, ray_grid=(6, 10))
    # C2-like sampling density, few rays
    render_case(ref, "g4_render_n16_s64_mid", 16, 64, 0.55, 0.55, 0.5, seed=104, ray_grid=(6, 12), z_range=(8.0, 60.0))
    # one instance (softmin degenerates)
    render_case(ref, "g4_render_n1_s32_late", 1, 32, 0.1, 0.2, 0.9, seed=105, ray_grid=(6, 10), z_range=(6.0, 12.0), eikonal_weight=0.01)
    # G10: residual MLP enabled (C3-like, tiny)
    render_case(ref, "g10_render_residual_n3_s16", 3, 16, 0.4, 0.3, 0.6, seed=106, with_residual=True,
                eikonal_weight=0.01, ray_grid=(5, 8), z_range=(6.0, 20.0))


def golden_rendering_bench_shapes(ref):
    """G17: the shapes bench.py times (VERDICT r01 "parity gaps"): BASELINE config 3 (N=16, S=64, residual MLP) runs
    render_residual_step_kernel<2>, config 5 (N=64, S=128, box-only) runs render_silhouette_kernel<4>, and a residual field with
    S in (64, 128] runs the <4> instantiations of the residual kernels.  A few dozen rays through the objects suffice."""
    render_case(ref, "g17_render_residual_n16_s64_mid", 16, 64, 0.55, 0.55, 0.5, seed=107, with_residual=True, eikonal_weight=0.01,
                ray_grid=(5, 8), row_range=(70, 112), z_range=(8.0, 60.0), split_gradients=True)
    render_case(ref, "g17_render_n64_s128_mid", 64, 128, 0.55, 0.55, 0.5, seed=108, ray_grid=(5, 8), row_range=(70, 112),
                z_range=(8.0, 60.0), split_gradients=True)
    render_case(ref, "g17_render_residual_n4_s100_late", 4, 100, 0.2, 0.2, 0.8, seed=109, with_residual=True, eikonal_weight=0.01,
                ray_grid=(5, 8), row_range=(70, 112), z_range=(6.0, 25.0), split_gradients=True)


# ---------------------------------------------------------------------------
# G5: samplers, deterministic known answers
# ---------------------------------------------------------------------------

def golden_samplers(ref):
    g = torch.Generator().manual_seed(5)
    q = ref.samplers.quadrature_sampler(torch.linspace(0, 4, 5), deterministic=True)
    bins = torch.tensor([0.0, 1.0, 2.0, 3.0])
    it1 = ref.samplers.inverse_transform_sampler(bins, torch.tensor([0.0, 1.0, 1.0]), 5, deterministic=True)
    it0 = ref.samplers.inverse_transform_sampler(bins, torch.tensor([0.0, 0.0, 0.0]), 3, deterministic=True)
    # random case with the uniforms recorded
    rb = torch.sort(torch.rand(7, 1, 12, generator=g) * 50.0, dim=-1).values
    rw = torch.rand(7, 1, 11, generator=g) * (torch.rand(7, 1, 11, generator=g) > 0.4)
    rw[3] = 0.0
    torch.manual_seed(77)
    rs = ref.samplers.inverse_transform_sampler(rb, rw, 9)
    torch.manual_seed(77)
    ru = torch.rand(7, 1, 9)
    save("g5_samplers", quadrature=q, it_bins=bins, it_samples_011=it1, it_samples_000=it0,
         rand_bins=rb, rand_weights=rw, rand_uniforms=ru, rand_samples=rs)


# ---------------------------------------------------------------------------
# G6: positional encoder + per-instance MLP
# ---------------------------------------------------------------------------

def golden_mlp(ref):
    g = torch.Generator().manual_seed(6)
    enc = ref.encoder.SinusoidalEncoder(8)
    known = enc(torch.tensor([0.1, 0.2, 0.3]))
    hyper = ref.hyper.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256])
    weights = torch.randn(4, 1617, generator=g) * 0.25
    x = (torch.rand(4, 33, 3, generator=g) * 2.0 - 1.0) * 0.05
    x.requires_grad_(True)
    feats = enc(x)
    out = hyper.distance_field(weights[:, None, :], feats)
    gx, = torch.autograd.grad(out, x, torch.ones_like(out))
    save("g6_encoder_mlp", encoder_known=known, weights=weights, positions=x.detach(), encoded=feats.detach(),
         outputs=out.detach(), input_gradients=gx,
         num_neurons=np.array(hyper.num_neurons_list))


# ---------------------------------------------------------------------------
# G7/G8: projection, box parameters
# ---------------------------------------------------------------------------

LINE_INDICES = [[0, 1], [1, 2], [2, 3], [3, 0], [4, 5], [5, 6], [6, 7], [7, 4], [0, 4], [1, 5], [2, 6], [3, 7]]  # main.py:26-30


def golden_projection(ref):
    g = torch.Generator().manual_seed(7)
    K = kitti_like_intrinsics(376, 1408)
    bp = ref.box_parameters.BoxParameters3D(1, 6)
    with torch.no_grad():
        bp.locations.copy_(torch.randn(1, 6, 3, generator=g) * 0.6)
        bp.dimensions.copy_(torch.randn(1, 6, 3, generator=g))
        bp.orientations.copy_(torch.randn(1, 6, 2, generator=g))
    out = bp()
    boxes = out["boxes_3d"][0].detach().clone()
    # force cases: [4] straddles z=0, [5] fully behind the camera
    boxes[4] = boxes[4] - boxes[4].mean(0) + torch.tensor([0.5, 0.8, 0.4])
    boxes[5] = boxes[5] - boxes[5].mean(0) + torch.tensor([-1.0, 0.7, -9.0])
    boxes.requires_grad_(True)
    boxes_2d = torch.stack([ref.geo.project_box_3d(b, LINE_INDICES, K) for b in boxes])
    grad_boxes, = torch.autograd.grad(boxes_2d[:5].clamp(-1e4, 1e4).sum(), boxes)
    lines = boxes.detach()[:, LINE_INDICES, :]
    clipped, masks = ref.geo.clip_lines_to_front(lines)
    zero = ref.box_parameters.BoxParameters3D(1, 1)()
    zero_2d = ref.geo.project_box_3d(zero["boxes_3d"][0, 0], LINE_INDICES, K)
    enc_loc, enc_dim, enc_rot = ref.box_parameters.BoxParameters3D.encode_box_3d(out["boxes_3d"])
    save("g7_g8_projection_boxes",
         K=K, raw_locations=bp.locations, raw_dimensions=bp.dimensions, raw_orientations=bp.orientations,
         locations=out["locations"], dimensions=out["dimensions"], orientations=out["orientations"], decoded_boxes_3d=out["boxes_3d"],
         boxes_3d=boxes.detach(), boxes_2d=boxes_2d.detach(), grad_boxes_3d=grad_boxes,
         clipped_lines=clipped, clip_masks=masks,
         zero_location=zero["locations"], zero_dimension=zero["dimensions"], zero_orientation=zero["orientations"],
         zero_box_2d=zero_2d.detach(),
         encoded_locations=enc_loc.detach(), encoded_dimensions=enc_dim.detach(), encoded_orientations=enc_rot.detach(),
         rotation_matrix_x=ref.geo.rotation_matrix_x(torch.tensor([0.0, 0.3, -1.2])),
         expand_to_4x4=ref.geo.expand_to_4x4(torch.arange(18.0).reshape(2, 3, 3)))


def golden_sphere_tracing(ref):
    """N1: sphere tracing + surface normals over soft/hard unions."""
    g = torch.Generator().manual_seed(9)
    loc, dims, rot = scene_instances(3, g, z_range=(8.0, 20.0))
    H, W = 128, 128
    rows = torch.linspace(20, H - 20, 8).long()
    cols = torch.linspace(10, W - 10, 12).long()
    origins, directions = build_rays(H, W, rows, cols)
    union = make_soft_union([
        ref.sdfs.translation(ref.sdfs.rotation(make_instance_field(ref.sdfs.box(dims[i]), i, 3), rot[i]), loc[i])
        for i in range(3)
    ], 0.1)
    field = lambda p: union(p)[0]
    with torch.no_grad():
        pos, conv = ref.renderers.sphere_tracing(field, origins, directions, num_iterations=200,
                                                 convergence_criteria=0.01, bounding_radius=100.0)
    normals = ref.renderers.surface_normal(field, pos.clone())
    save("g9_sphere_tracing", locations=loc, dimensions=dims, orientations=rot, origins=origins, directions=directions,
         temperature=np.array(0.1, dtype=np.float32), surface_positions=pos, convergence_masks=conv, surface_normals=normals.detach())


def golden_soft_rasterizer():
    """N2: SoftRasterizer.make_distance_map + the soft-mask formula (geometric_transforms.py:265-317).
    The module imports torchvision / cv2 / skimage at the top; empty placeholder modules let the import through and are
    never called: only the pure-torch ``make_distance_map`` and the sigmoid formula of ``forward`` are exercised (the OpenCV
    polygon extraction / filling steps need the real libraries and are not part of this fixture)."""
    for name in ("torchvision", "cv2", "skimage"):
        sys.modules.setdefault(name, types.ModuleType(name))
    _stub_package("vsrd.transforms", os.path.join(REFERENCE_ROOT, "vsrd", "transforms"))
    gt = importlib.import_module("vsrd.transforms.geometric_transforms")
    rast = gt.SoftRasterizer(threshold=0.5, temperature=10.0)
    g = torch.Generator().manual_seed(12)
    H, W = 40, 56
    polygons = [
        torch.tensor([[10, 8], [30, 6], [44, 20], [36, 33], [14, 30]], dtype=torch.int32),
        torch.tensor([[3, 3], [20, 4], [20, 18], [3, 17]], dtype=torch.int32),
        torch.randint(0, 40, (9, 2), generator=g).to(torch.int32),
    ]
    maps = [rast.make_distance_map(p, (H, W)) for p in polygons]
    inside = [(torch.rand(H, W, generator=g) > 0.5) for _ in polygons]      # arbitrary binary masks: the formula is per pixel
    soft = [torch.sigmoid(torch.where(b, d, -d) / rast.temperature) for b, d in zip(inside, maps)]
    arrays = {"hw": np.array([H, W]), "temperature": np.array(rast.temperature, dtype=np.float32)}
    for k, (p, d, b, s) in enumerate(zip(polygons, maps, inside, soft)):
        arrays.update({f"polygon_{k}": p, f"distance_{k}": d, f"inside_{k}": b, f"soft_{k}": s})
    save("g11_soft_rasterizer", **arrays)


def golden_formats(ref):
    """N4: KITTI label lines written by the reference's tools/kitti_360/convert_predictions.py::save_prediction,
    and the key/shape layout of the detector state dict the checkpoint consumer (make_predictions.py:61-66) loads.
    convert_predictions imports torchvision / pycocotools at the top: empty placeholders (never called) let it import."""
    import tempfile
    for name in ("torchvision", "pycocotools", "pycocotools.mask"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, os.path.join(REFERENCE_ROOT, "tools", "kitti_360"))
    convert = importlib.import_module("convert_predictions")
    g = torch.Generator().manual_seed(21)
    bp = ref.box_parameters.BoxParameters3D(1, 3)
    with torch.no_grad():
        bp.locations.copy_(torch.randn(1, 3, 3, generator=g) * 0.4)
        bp.dimensions.copy_(torch.randn(1, 3, 3, generator=g))
        bp.orientations.copy_(torch.randn(1, 3, 2, generator=g))
    boxes_3d = bp()["boxes_3d"][0].detach()
    boxes_2d = torch.rand(3, 2, 2, generator=g) * 300
    scores = torch.rand(3, generator=g)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "labels", "0000.txt")
        convert.save_prediction(path, ["car"] * 3, boxes_3d, boxes_2d, scores)
        text = open(path).read()
    state = bp.state_dict()
    save("g12_formats", boxes_3d=boxes_3d, boxes_2d=boxes_2d, scores=scores,
         kitti_text=np.frombuffer(text.encode(), dtype=np.uint8),
         state_keys=np.frombuffer("\n".join(f"{k}:{tuple(v.shape)}" for k, v in state.items()).encode(), dtype=np.uint8))


def golden_loss_library():
    """a22: the vsrd.losses library (never called by main.py) -- outputs of the reference functions on seeded inputs."""
    _stub_package("vsrd.losses", os.path.join(REFERENCE_ROOT, "vsrd", "losses"))
    mods = [importlib.import_module("vsrd.losses." + m) for m in
            ("classification_losses", "photometric_losses", "smoothness_losses", "probabilistic_losses")]
    lib = {}
    for m in mods:
        lib.update({k: v for k, v in vars(m).items() if callable(v)})
    g = torch.Generator().manual_seed(31)
    p, t = torch.rand(2, 3, 8, 9, generator=g), torch.rand(2, 3, 8, 9, generator=g)
    img_a, img_b = torch.rand(2, 3, 12, 10, generator=g), torch.rand(2, 3, 12, 10, generator=g)
    mean, target = torch.randn(5, 7, generator=g), torch.randn(5, 7, generator=g)
    var, shape, scale = torch.rand(5, 7, generator=g) + 0.1, torch.rand(5, 7, generator=g) + 1.5, torch.rand(5, 7, generator=g) + 0.2
    arrays = dict(p=p, t=t, img_a=img_a, img_b=img_b, mean=mean, target=target, var=var, shape=shape, scale=scale)
    for name in ("cross_entropy", "binary_cross_entropy", "kl_divergence", "binary_kl_divergence", "js_divergence",
                 "binary_js_divergence", "focal_loss", "quality_focal_loss", "tversky_loss", "focal_tversky_loss"):
        arrays["out_" + name] = lib[name](p, t, reduction="none")
    arrays["out_cross_entropy_dim1"] = lib["cross_entropy"](p, t, dim=1, reduction="mean")
    arrays["out_ssim_loss"] = lib["ssim_loss"](img_a, img_b, reduction="none")
    arrays["out_photometric_loss"] = lib["photometric_loss"](img_a, img_b, reduction="none")
    arrays["out_smoothness_loss"] = lib["smoothness_loss"](img_a[:, :1], img_b, reduction="none")
    arrays["out_motion_smoothness_loss"] = lib["motion_smoothness_loss"](img_a, reduction="none")
    arrays["out_motion_sparsity_loss"] = lib["motion_sparsity_loss"](img_a - 0.5, reduction="sum")
    arrays["out_gaussian_nll"] = lib["gaussian_nll"](mean, var, target, reduction="none")
    arrays["out_student_nll"] = lib["student_nll"](mean, shape, scale, target, reduction="none")
    save("g13_loss_library", **arrays)


def golden_loss_library_rest():
    """a22, second half (G18): geometric_losses.py and the rest of probabilistic_losses.py (logit-space NLLs, Monte-Carlo energy
    scores).  The energy scores draw from torch's global generator: seeded right before each call, so a consumer that seeds the same
    way and draws in the same order (one rsample([num_samples]) call) replays the reference's samples."""
    _stub_package("vsrd.losses", os.path.join(REFERENCE_ROOT, "vsrd", "losses"))
    geo = importlib.import_module("vsrd.losses.geometric_losses")
    prob = importlib.import_module("vsrd.losses.probabilistic_losses")
    g = torch.Generator().manual_seed(41)

    def pose(n):
        q, _ = torch.linalg.qr(torch.randn(n, 3, 3, generator=g))
        m = torch.eye(4).repeat(n, 1, 1)
        m[:, :3, :3] = q
        m[:, :3, 3] = torch.randn(n, 3, generator=g)
        return m
    source, target = pose(6), pose(6)
    target[0] = torch.linalg.inv(source[0])                      # a consistent pair: both losses vanish
    k1, k2 = torch.rand(6, 11, 2, generator=g) * 100.0, torch.rand(6, 11, 2, generator=g) * 100.0
    fundamental = torch.randn(6, 3, 3, generator=g) * 0.01
    mean, target_v = torch.randn(5, 7, generator=g), torch.randn(5, 7, generator=g)
    var, shape, scale = torch.rand(5, 7, generator=g) + 0.1, torch.rand(5, 7, generator=g) + 1.5, torch.rand(5, 7, generator=g) + 0.2
    unit = torch.rand(5, 7, generator=g) * 0.9 + 0.05
    arrays = dict(source=source, target=target, keypoints_1=k1, keypoints_2=k2, fundamental=fundamental, mean=mean, target_values=target_v,
                  var=var, shape=shape, scale=scale, unit_targets=unit, num_samples=np.array(64), seed=np.array(1234))
    arrays["out_rotation_consistency_loss"] = geo.rotation_consistency_loss(source, target, reduction="none")
    arrays["out_translation_consistency_loss"] = geo.translation_consistency_loss(source, target, reduction="none")
    arrays["out_sampson_epipolar_distance"] = geo.sampson_epipolar_distance(k1, k2, fundamental[:, None], reduction="none")
    arrays["out_logit_gaussian_nll"] = prob.logit_gaussian_nll(mean, var, unit, reduction="none")
    arrays["out_logit_student_nll"] = prob.logit_student_nll(mean, shape, scale, unit, reduction="none")
    for name, args in (("gaussian_energy_score", (mean, var, target_v)), ("student_energy_score", (mean, shape, scale, target_v)),
                       ("logit_gaussian_energy_score", (mean, var, unit)), ("logit_student_energy_score", (mean, shape, scale, unit))):
        torch.manual_seed(1234)
        arrays["out_" + name] = getattr(prob, name)(*args, num_samples=64, reduction="none")
    save("g18_loss_library_rest", **arrays)


def golden_box_3d_iou(ref):
    """G9 of SURVEY.md §8c: vsrd.operations.box_3d_iou on box pairs prepared as scripts/main.py:888-905 does
    (corners @ rotation_matrix_x(-pi/2).T, so that "up" is Z): identical, shifted, yawed, contained, touching and disjoint."""
    g = torch.Generator().manual_seed(11)
    bp = ref.box_parameters.BoxParameters3D(1, 8)
    with torch.no_grad():
        bp.locations.copy_(torch.randn(1, 8, 3, generator=g) * 0.3)
        bp.dimensions.copy_(torch.randn(1, 8, 3, generator=g))
        bp.orientations.copy_(torch.randn(1, 8, 2, generator=g))
    first = bp()["boxes_3d"][0].detach()
    second = first.clone()
    centre = first.mean(1, keepdim=True)
    second[1] = first[1] + torch.tensor([0.4, 0.0, 0.7])                                  # shifted
    yaw = ref.box_parameters.rotation_matrix_y(torch.tensor(0.9659), torch.tensor(0.2588))    # 15 degrees about y
    second[2] = (first[2] - centre[2]) @ yaw.T + centre[2]
    second[3] = (first[3] - centre[3]) * 0.5 + centre[3]                                   # contained
    second[4] = first[4] + torch.tensor([30.0, 0.0, 0.0])                                  # disjoint
    second[5] = (first[5] - centre[5]) @ yaw.T @ yaw.T + centre[5] + torch.tensor([0.3, 0.2, -0.5])
    second[6] = first[6] + torch.tensor([0.0, 0.6, 0.0])                                   # shifted along the up axis only
    second[7] = (first[7] - centre[7]) * 1.7 + centre[7] + torch.tensor([-0.2, 0.0, 0.1])  # containing
    rotation = ref.geo.rotation_matrix_x(torch.tensor(-math.pi / 2.0))
    a, b = first @ rotation.T, second @ rotation.T
    ious = [ref.k360.box_3d_iou(corners1=x, corners2=y) for x, y in zip(a, b)]
    save("g14_box_3d_iou", corners1=a, corners2=b,
         iou_3d=torch.tensor([float(i[0]) for i in ious], dtype=torch.float64), iou_bev=torch.tensor([float(i[1]) for i in ious], dtype=torch.float64))


def golden_hypernetwork(ref):
    """a12: HyperDistanceField.forward (weight-normed Linear + LayerNorm + GELU stack) with a small hyper width, its state dict
    (the checkpoint layout scripts/main.py:1109-1121 saves) and the gradient of a scalar of its output w.r.t. the embeddings."""
    torch.manual_seed(5)
    module = ref.hyper.HyperDistanceField(48, [16, 16, 16, 16], 12, [10, 14])
    embeddings = torch.randn(1, 3, 12).requires_grad_(True)
    weights = module(embeddings)
    probe = torch.randn(weights.shape, generator=torch.Generator().manual_seed(6))
    grad_embeddings, = torch.autograd.grad((weights * probe).sum(), embeddings)
    state = {"state__" + k.replace(".", "__"): v.detach() for k, v in module.state_dict().items()}
    save("g15_hypernetwork", embeddings=embeddings.detach(), weights=weights.detach(), probe=probe, grad_embeddings=grad_embeddings, **state)


def golden_rendering_helpers(ref):
    """The small functions that complete the vsrd.rendering listing: sphere_intersection, phong_shading, sdfs.norm."""
    g = torch.Generator().manual_seed(21)
    positions = torch.randn(12, 3, generator=g) * 3.0
    directions = nn.functional.normalize(torch.randn(12, 3, generator=g), dim=-1)
    near, far, hit = ref.renderers.sphere_intersection(positions, directions, 4.0)
    args = dict(ray_directions=torch.randn(12, 3, generator=g), surface_normals=torch.randn(12, 3, generator=g),
                light_directions=torch.randn(12, 3, generator=g), light_ambient_colors=torch.rand(3, generator=g),
                light_diffuse_colors=torch.rand(3, generator=g), light_specular_colors=torch.rand(3, generator=g),
                material_ambient_colors=torch.rand(12, 3, generator=g), material_diffuse_colors=torch.rand(12, 3, generator=g),
                material_specular_colors=torch.rand(12, 3, generator=g), material_emission_colors=torch.rand(12, 3, generator=g) * 0.2,
                material_shininesses=torch.tensor(8.0))
    colors = ref.renderers.phong_shading(**args)
    vectors = torch.randn(5, 3, generator=g)
    save("g16_rendering_helpers", positions=positions, directions=directions, near=near, far=far, hit=hit, colors=colors,
         vectors=vectors, norms=ref.sdfs.norm(vectors, dim=-1, keepdim=True), **{"phong__" + k: v for k, v in args.items()})


def main():
    torch.set_num_threads(4)
    ref = import_reference()
    if "--bench-shapes" in sys.argv:       # only the G17 files (the others are unchanged since round 1)
        golden_rendering_bench_shapes(ref)
        return
    if "--loss-library-rest" in sys.argv:  # only G18
        golden_loss_library_rest()
        return
    golden_rendering_bench_shapes(ref)
    golden_loss_library_rest()
    golden_rendering_helpers(ref)
    golden_hypernetwork(ref)
    golden_box_3d_iou(ref)
    golden_ray_casting(ref)
    golden_sdf(ref)
    golden_samplers(ref)
    golden_mlp(ref)
    golden_projection(ref)
    golden_sphere_tracing(ref)
    golden_soft_rasterizer()
    golden_formats(ref)
    golden_loss_library()
    golden_rendering(ref)
    leftovers = [p for p, _, _ in os.walk(REFERENCE_ROOT) if p.endswith("__pycache__")]
    assert not leftovers, leftovers


if __name__ == "__main__":
    main()
