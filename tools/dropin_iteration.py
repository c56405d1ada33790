#!/usr/bin/env python3
"""What ONE optimisation step of an unchanged scripts/main.py costs through the drop-in surface.

After ``vsrd_amd.install_as_vsrd()`` the step below walks main.py's own call sequence (scripts/main.py:323-865) at the reference's
sizes -- V = 17 views of 376 x 1408, N = 8 instances, 1000 importance-sampled rays x 100 samples (config.json:16,22-23,236-237):

    detector()                                                       main.py:332
    V x N  vsrd.operations.project_box_3d  + clip to the image       main.py:339-362   (136 calls)
    -DIoU cost on the target view, scipy linear_sum_assignment       main.py:374-386   (device -> host -> device)
    per-view DIoU / smooth-L1 projection losses                      main.py:391-415
    schedules                                                        main.py:420-431
    hypernetwork (residual phase), per-instance field closures       main.py:525-618
    torch.multinomial over V*H*W soft-mask maxima                    main.py:620-627
    hierarchical_wrapper(vsrd.rendering.hierarchical_volumetric_rendering)   main.py:511-523, 629-651
    silhouette BCE (+ eikonal MSE)                                   main.py:653-687
    weighted sum, backward(), Adam step, ExponentialLR step          main.py:855-865

written for this tool from that sequence (the field closures are tests/test_hip_dropin.py's, which have main.py's free variables).
torchvision is not installed here: ``distance_box_iou`` / ``distance_box_iou_loss`` are vsrd_amd.losses' restatements (same
element-wise torch ops as torchvision's), ``clip_boxes_to_image`` is the two clamps it is.

Prints, for the box-only and the residual phase: wall-clock ms per iteration (synchronised over the run), the host's own issue time,
where the host time goes section by section, and the native loop's step on the same frame (FrameOptimizer(graph=True)) beside it.

  python tools/dropin_iteration.py [--iterations 100] [--views 17] [--instances 8]
"""
import argparse
import os
import sys
import time
import types
from collections import defaultdict

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LINE_INDICES = [[0, 1], [1, 2], [2, 3], [3, 0], [4, 5], [5, 6], [6, 7], [7, 4], [0, 4], [1, 5], [2, 6], [3, 7]]


class Sections:
    """Host time per section of the step (perf_counter, no synchronisation: what the host spends issuing the work)."""

    def __init__(self):
        self.totals = defaultdict(float)
        self.last = None

    def start(self):
        self.last = time.perf_counter()

    def mark(self, name):
        now = time.perf_counter()
        self.totals[name] += now - self.last
        self.last = now


def clip_boxes_to_image(boxes, size):
    height, width = size
    x = boxes[..., 0::2].clamp(0, width)
    y = boxes[..., 1::2].clamp(0, height)
    return torch.stack([x, y], dim=-1).flatten(-2, -1)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--iterations", type=int, default=100)
    parser.add_argument("--views", type=int, default=17)
    parser.add_argument("--instances", type=int, default=8)
    parser.add_argument("--rays", type=int, default=1000)
    parser.add_argument("--samples", type=int, default=100)
    args = parser.parse_args()
    import __graft_entry__
    __graft_entry__.build()
    import scipy.optimize
    import bench
    import vsrd_amd
    vsrd_amd.install_as_vsrd()
    import vsrd
    from vsrd_amd import losses as tv            # stands in for torchvision.ops (see the docstring)
    from vsrd_amd import optimization
    from test_hip_dropin import train_like_fields

    dev = torch.device("cuda:0")
    V, H, W, N = args.views, 376, 1408, args.instances
    K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(0, V, H, W, N)
    K, E = K.to(dev), E.to(dev)
    # ---- what main.py:106-316 prepares once per frame: models, optimiser, rays, soft masks, ground-truth 2-D boxes -------------
    torch.manual_seed(0)
    models = types.SimpleNamespace(
        detector=vsrd.models.BoxParameters3D(1, N).to(dev),
        positional_encoder=vsrd.models.SinusoidalEncoder(num_frequencies=8).to(dev),
        hyper_distance_field=vsrd.models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).to(dev))
    with torch.no_grad():
        models.detector.locations.copy_(raw_loc); models.detector.dimensions.copy_(raw_dim); models.detector.orientations.copy_(raw_ori)
        truth = models.detector()
        camera_positions, ray_directions = vsrd.rendering.ray_casting((H, W), K, E)
        block = vsrd_amd.fields.FieldBlock(vsrd_amd.fields.pack_instances(truth["locations"][0], truth["orientations"][0], truth["dimensions"][0]), 0.1, None, None)
        origins = camera_positions[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
        soft_masks = vsrd.rendering.render_hierarchical(block, origins, ray_directions.reshape(-1, 3), (0.0, 100.0), 64, 0.1, 1.0, seed=1,
                                                        skip_exact_misses=True)["labels"].clamp(0, 1).reshape(V, H, W, N).contiguous()
        gt_boxes_2d, _ = vsrd.operations.project_boxes_multi_view(truth["boxes_3d"][0], E, K, (H, W))      # [V,N,2,2]
        models.detector.locations.add_(0.05)
    visible_masks = torch.ones(V, N, dtype=torch.bool, device=dev)
    multi_camera_positions = camera_positions[:, None, None, :].expand(V, H, W, 3)
    flat_positions, flat_directions = multi_camera_positions.flatten(0, -2), ray_directions.flatten(0, -2)
    sampling_weights = soft_masks.max(dim=-1).values.flatten()[None]
    flat_masks = soft_masks.flatten(0, -2)
    config = types.SimpleNamespace(volume_rendering=types.SimpleNamespace(distance_range=[0.0, 100.0]))
    num_steps = 3000
    groups = [dict(params=[p], lr=1.0e-2) for p in (models.detector.locations, models.detector.dimensions, models.detector.orientations)]
    groups.append(dict(params=[models.detector.embeddings], lr=1.0e-3))
    groups.append(dict(params=list(models.hyper_distance_field.parameters()), lr=1.0e-4))
    optimizer = torch.optim.Adam(groups)
    scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=0.01 ** (1.0 / num_steps))
    weights = dict(iou_projection_loss=0.1, l1_projection_loss=1.0, silhouette_loss=1.0, eikonal_loss=0.01)      # config.json:120-127
    sections = Sections()

    def iteration(step, residual):
        sections.start()
        optimizer.zero_grad()
        world = models.detector()
        sections.mark("detector")
        # ---- multi-view projection (main.py:336-367) ------------------------------------------------------------------------------
        world_boxes_3d = nn.functional.pad(world["boxes_3d"], (0, 1), mode="constant", value=1.0)
        boxes_2d = []
        for view in range(V):
            camera_boxes_3d = torch.einsum("bmn,b...n->b...m", E[view][None], world_boxes_3d)
            camera_boxes_3d = camera_boxes_3d[..., :-1] / camera_boxes_3d[..., -1:]
            camera_boxes_2d = torch.stack([
                torch.stack([vsrd.operations.project_box_3d(box_3d=box, line_indices=LINE_INDICES, intrinsic_matrix=intrinsic) for box in boxes], dim=0)
                for boxes, intrinsic in zip(camera_boxes_3d, K[view][None])], dim=0)
            boxes_2d.append(clip_boxes_to_image(camera_boxes_2d.flatten(-2, -1), (H, W)).unflatten(-1, (2, 2)))
        sections.mark(f"projection ({V} x {N} project_box_3d)")
        # ---- matching on the target view (main.py:374-386): device -> host -> device -----------------------------------------------
        cost = -tv.distance_box_iou(boxes_2d[0][0].flatten(-2, -1), gt_boxes_2d[0].flatten(-2, -1))
        rows, cols = scipy.optimize.linear_sum_assignment(cost.detach().cpu().numpy())
        pd_indices, gt_indices = torch.as_tensor(rows, device=dev), torch.as_tensor(cols, device=dev)
        sections.mark("matching (host scipy)")
        # ---- projection losses (main.py:391-415) ----------------------------------------------------------------------------------
        iou_terms, l1_terms = [], []
        for view in range(V):
            keep = visible_masks[view][gt_indices]
            pd = boxes_2d[view][0][pd_indices[keep]].flatten(-2, -1)
            gt = gt_boxes_2d[view][gt_indices[keep]].flatten(-2, -1)
            iou_terms.append(tv.distance_box_iou_loss(pd, gt))
            l1_terms.append(nn.functional.smooth_l1_loss(pd, gt, reduction="none"))
        iou_projection_loss, l1_projection_loss = torch.mean(torch.cat(iou_terms)), torch.mean(torch.cat(l1_terms))
        sections.mark("projection losses")
        # ---- schedules (main.py:420-431) --------------------------------------------------------------------------------------------
        anneal = lambda x, a, b: (np.cos(np.pi * x) + 1.0) / 2.0 * (a - b) + b
        cosine_ratio = step / num_steps
        temperature, std = anneal(step / num_steps, 1.0, 0.1), anneal(step / num_steps, 1.0, 0.1)
        # ---- fields (main.py:525-618) -------------------------------------------------------------------------------------------------
        outputs = types.SimpleNamespace(locations=world["locations"], dimensions=world["dimensions"], orientations=world["orientations"])
        if residual:
            outputs.distance_field_weights = models.hyper_distance_field(world["embeddings"])
        fields, hierarchical_wrapper = train_like_fields(vsrd, config, models, outputs, N, temperature, residual)
        sections.mark("hypernetwork + field closures" if residual else "field closures")
        # ---- rays (main.py:620-627) ---------------------------------------------------------------------------------------------------
        ray_indices = torch.multinomial(sampling_weights, args.rays, replacement=False)[0]
        sections.mark("torch.multinomial")
        # ---- render (main.py:629-651) -------------------------------------------------------------------------------------------------
        labels, gradients = hierarchical_wrapper(vsrd.rendering.hierarchical_volumetric_rendering)(
            distance_field=fields[0], ray_positions=flat_positions[ray_indices], ray_directions=flat_directions[ray_indices],
            distance_range=config.volume_rendering.distance_range, num_samples=args.samples, sdf_std_deviation=std, cosine_ratio=cosine_ratio)
        sections.mark("hierarchical_wrapper(renderer): two passes")
        # ---- losses (main.py:653-687, 855) ----------------------------------------------------------------------------------------------
        silhouette_loss = torch.mean(nn.functional.binary_cross_entropy(labels[..., pd_indices].clamp(1.0e-6, 1.0 - 1.0e-6),
                                                                        flat_masks[ray_indices][..., gt_indices], reduction="none"))
        terms = dict(iou_projection_loss=iou_projection_loss, l1_projection_loss=l1_projection_loss, silhouette_loss=silhouette_loss)
        if residual:
            terms["eikonal_loss"] = nn.functional.mse_loss(torch.norm(gradients, dim=-1), gradients.new_ones(*gradients.shape[:-1]), reduction="mean")
        loss = sum(terms[name] * weights[name] for name in terms)
        sections.mark("silhouette / eikonal losses")
        loss.backward()
        sections.mark("backward()")
        optimizer.step()
        scheduler.step()
        sections.mark("Adam + ExponentialLR")
        return loss

    report = []
    for residual in (False, True):
        first = 1000 if residual else 0
        for k in range(10):
            iteration(first + k, residual)
        torch.cuda.synchronize()
        sections.totals.clear()
        t0 = time.perf_counter()
        for k in range(args.iterations):
            loss = iteration(first + 10 + k, residual)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        phase = "residual phase (step 1000+)" if residual else "box-only phase"
        report.append(f"{phase}: {total / args.iterations * 1e3:.2f} ms per main.py-shaped iteration through the drop-in surface "
                      f"(host issue time {host / args.iterations * 1e3:.2f} ms), V = {V}, N = {N}, {args.rays} rays x {args.samples} samples; final loss {float(loss):.4f}")
        for name, seconds in sections.totals.items():
            report.append(f"    host {seconds / args.iterations * 1e3:7.3f} ms  {name}")
    # ---- project_box_3d alone: host time per call -------------------------------------------------------------------------------------
    with torch.no_grad():
        boxes = models.detector()["boxes_3d"][0]
    boxes = boxes.detach().clone().requires_grad_(True)
    for _ in range(50):
        vsrd.operations.project_box_3d(boxes[0], LINE_INDICES, K[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    calls = 2000
    for i in range(calls):
        vsrd.operations.project_box_3d(boxes[i % N], LINE_INDICES, K[0])
    per_call = (time.perf_counter() - t0) / calls
    torch.cuda.synchronize()
    report.append(f"vsrd.operations.project_box_3d: {per_call * 1e6:.1f} us of host time per call (autograd-recording, one launch)")
    # ---- the native loop on the same frame ----------------------------------------------------------------------------------------------
    inputs = optimization.FrameInputs((H, W), K, E, soft_masks, gt_boxes_2d.reshape(V, N, 2, 2), visible_masks)
    for residual in (False, True):
        loop = optimization.FrameOptimizer(inputs, optimization.OptimizationConfig(num_rays=args.rays, num_samples=args.samples), dev, graph=True)
        if residual:
            loop.step_index = loop.config.warmup_steps
            loop.step_tensor.fill_(loop.step_index)
        for _ in range(20):
            loop.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.run(400, 4)
        torch.cuda.synchronize()
        native = (time.perf_counter() - t0) / 400
        loop.close()
        report.append(f"native loop, {'residual' if residual else 'box-only'} phase (FrameOptimizer(graph=True).run, four steps per hipGraph): {native * 1e3:.4f} ms per step")
    print("\n".join(report))


if __name__ == "__main__":
    main()
