#!/usr/bin/env python3
"""Where do the multi-ray step kernels differ with and without culling?  Per-ray label differences of one 376 x 1408 view of the
benchmark scene, their distribution, and for the worst rays the pass-2 distances of both modes (moved fine samples = the importance
sampler's own ill-conditioning, samplers.py:33: division by cdf differences + 1e-6).
The worst rays are then re-rendered by the float64 CPU oracle AT THE KERNEL'S OWN DISTANCES: which of the two modes is off?
(test infrastructure: imports oracle/)
    python tests/culling_ab_debug.py [start|mid|end] [quad|pair]
The labels are the FUSED STEP's (vsrd_render_silhouette_step: what bench.py times); the distances come from the forward launch with
the same keys (informative: the step does not return its samples).  VSRD_NO_FULL_SHAPE=1 keeps the step on the generic kernels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import fields as ofields, rendering as orendering, geometry as ogeometry
import torch

import bench
from vsrd_amd import rendering
from vsrd_amd.rendering import renderers
from test_hip_scale import scene


def main():
    schedule = sys.argv[1] if len(sys.argv) > 1 else "start"
    shape = sys.argv[2] if len(sys.argv) > 2 else "quad"
    dev = torch.device("cuda:0")
    N, S = (16, 64) if shape == "quad" else (64, 128)
    H, W = 376, 1408
    sched = bench.schedule_values(bench.SCHEDULES[schedule])
    det, cam, dirs = scene(dev, N, 1, H, W, seed=0)
    directions = dirs.reshape(-1, 3)
    origins = cam[:, None, None, :].expand(1, H, W, 3).reshape(-1, 3).contiguous()
    with torch.no_grad():
        det.locations.add_(0.02)
    out = {}
    for mode in ("default", "no_culling"):
        renderers.CULLING = mode == "default"
        with torch.no_grad():
            union = bench.build_union(det, sched["temperature"])
            out[mode] = rendering.render_hierarchical(union, origins, directions, (0.0, 100.0), S, sched["std"], sched["cosine_ratio"], seed=5, stream_offset=11,
                                                      skip_exact_misses=True)
            _, step_labels = rendering.silhouette_step(union, origins, directions, torch.zeros(origins.shape[0], N, device=dev), (0.0, 100.0), S, sched["std"],
                                                       sched["cosine_ratio"], seed=5, stream_offset=11, return_labels=True)
            forward_labels = out[mode]["labels"]
            out[mode] = dict(out[mode], labels=step_labels)
            print(f"  [{mode}] fused step labels vs forward launch labels: max diff {float((step_labels - forward_labels).abs().max()):.3e}")
    renderers.CULLING = True
    a, b = out["default"], out["no_culling"]
    diff = (a["labels"] - b["labels"]).abs().max(-1).values
    print(f"schedule {schedule} ({shape}, N = {N}, S = {S}): rays {diff.numel()}  max label diff {float(diff.max()):.3e}")
    for tol in (1e-7, 1e-6, 2e-6, 1e-5, 1e-4):
        print(f"  rays with diff > {tol:g}: {int((diff > tol).sum())}")
    da, db = a["distances"], b["distances"]
    nan_a, nan_b = torch.isnan(da[:, 0]), torch.isnan(db[:, 0])
    print(f"  exact-miss rays: default {int(nan_a.sum())}  no_culling {int(nan_b.sum())}  differ {int((nan_a != nan_b).sum())}")
    both = ~nan_a & ~nan_b
    moved = ((da - db).abs() > 1e-3).any(-1) & both
    print(f"  rays with a pass-2 distance moved by > 1 mm: {int(moved.sum())};  of the rays with label diff > 2e-6: "
          f"{int((moved & (diff > 2e-6)).sum())} of {int((diff > 2e-6).sum())}")
    still = (diff > 2e-6) & ~moved & both
    print(f"  rays with label diff > 2e-6 whose distances did NOT move: {int(still.sum())}; their worst diff {float(diff[still].max()) if still.any() else 0:.3e}")
    worst = torch.topk(diff, 5).indices
    raw = [p.detach().double().cpu()[0] for p in (det.locations, det.dimensions, det.orientations)]
    loc, dim, rot, _ = ogeometry.decode_box_parameters(*raw)
    ounion = ofields.InstanceUnion(loc, rot, dim, sched["temperature"])
    for r in worst.tolist():
        dd = (da[r] - db[r]).abs()
        print(f"  ray {r} (row {r // W}, col {r % W}): label diff {float(diff[r]):.3e}; labels sum {float(a['labels'][r].sum()):.4f} / {float(b['labels'][r].sum()):.4f}; "
              f"distances moved: {int((dd > 1e-3).sum())} of {dd.numel()}, max move {float(dd.max()):.3e} m")
        for mode, o in (("default", a), ("no_culling", b)):
            ref = orendering.render_given_distances(ounion, origins[r:r + 1].double().cpu(), directions[r:r + 1].double().cpu(), o["distances"][r:r + 1].double().cpu(),
                                                    sched["std"], sched["cosine_ratio"]).labels[0]
            err = (o["labels"][r].double().cpu() - ref).abs()
            print(f"      {mode:10s} vs float64 oracle at its own distances: max label error {float(err.max()):.3e} (instance {int(err.argmax())}); labels {[round(float(x), 6) for x in o['labels'][r] if float(x) > 1e-7]}")
        k = int(dd.argmax())
        print(f"      around the largest move (index {k}): default {da[r, max(k - 2, 0):k + 3].tolist()}\n                                   no_culling {db[r, max(k - 2, 0):k + 3].tolist()}")


if __name__ == "__main__":
    main()
