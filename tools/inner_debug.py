#!/usr/bin/env python3
"""Debug aid (GPU box): config 5's fused step under VSRD_HIP_LIBRARY, its labels and samples saved for the rays of a fixed selection
plus the rays the culling A/B moves most.  python tools/inner_debug.py <out.pt>"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from vsrd_amd import rendering
from vsrd_amd.rendering import renderers
from test_hip_scale import scene

dev = torch.device("cuda:0")
N, S, V, H, W, seed = 64, 128, 17, 752, 2816, 2
sched = bench.schedule_values(bench.SCHEDULES["mid"])
T, std, ratio = sched["temperature"], sched["std"], sched["cosine_ratio"]
det, cam, dirs = scene(dev, N, V, H, W, seed=seed)
directions = dirs.reshape(-1, 3)
origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
with torch.no_grad():
    targets = rendering.render_hierarchical(bench.build_union(det, 0.1), origins, directions, (0.0, 100.0), S, 0.1, 1.0, seed=99,
                                            skip_exact_misses=True)["labels"].clamp(0.0, 1.0).contiguous()
    det.locations.add_(0.02)
    keys = dict(seed=5, stream_offset=11)
    renderers.CULLING = False
    _, unculled = rendering.silhouette_step(bench.build_union(det, T), origins, directions, targets, (0.0, 100.0), S, std, ratio, return_labels=True, **keys)
    renderers.CULLING = True
    _, labels, samples = rendering.silhouette_step(bench.build_union(det, T), origins, directions, targets, (0.0, 100.0), S, std, ratio,
                                                   return_labels=True, return_samples=True, **keys)
    moved_by = (labels - unculled).abs().max(-1).values
    top = torch.topk(moved_by, 4096).indices
    print("rays moved by > 2e-6:", int((moved_by > 2e-6).sum()), " > 1e-4:", int((moved_by > 1e-4).sum()), " worst", float(moved_by.max()))
    out = dict(index=top.cpu(), moved_by=moved_by[top].cpu(), labels=labels[top].cpu(), unculled=unculled[top].cpu(),
               distances=samples["distances"][top].cpu(), origins=origins[top].cpu(), directions=directions[top].cpu(),
               locations=det.locations.detach().cpu(), dimensions=det.dimensions.detach().cpu(), orientations=det.orientations.detach().cpu())
    torch.save(out, sys.argv[1])
