#!/usr/bin/env python3
"""Two FrameOptimizer loops (graph mode) from the same seed in one process: are the parameters bit-identical after `steps` residual steps?
GPU box; experiments only."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__; __graft_entry__.build()
import bench
from vsrd_amd import optimization, rendering, fields, models, operations
dev = torch.device("cuda:0")
V, H, W, N = 17, 376, 1408, 8
K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(0, V, H, W, N)
det = models.BoxParameters3D(1, N).to(dev)
with torch.no_grad():
    det.locations.copy_(raw_loc); det.dimensions.copy_(raw_dim); det.orientations.copy_(raw_ori)
    out = det()
    cam, dirs = rendering.ray_casting((H, W), K.to(dev), E.to(dev))
    block = fields.FieldBlock(fields.pack_instances(out["locations"][0], out["orientations"][0], out["dimensions"][0]), 0.1, None, None)
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
    soft = rendering.render_hierarchical(block, origins, dirs.reshape(-1, 3), (0.0, 100.0), 64, 0.1, 1.0, seed=1, skip_exact_misses=True)["labels"].clamp(0, 1).reshape(V, H, W, N).contiguous()
    gt_boxes, _ = operations.project_boxes_multi_view(out["boxes_3d"][0], E.to(dev), K.to(dev), (H, W))
inputs = optimization.FrameInputs((H, W), K.to(dev), E.to(dev), soft, gt_boxes, torch.ones(V, N, dtype=torch.bool, device=dev))
warmup = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
def run():
    torch.manual_seed(0)
    loop = optimization.FrameOptimizer(inputs, optimization.OptimizationConfig(seed=0, warmup_steps=warmup), dev, graph=True)
    trace = []
    for s in range(warmup + steps):
        losses = loop.step()
        trace.append(float(losses["loss"]))
    torch.cuda.synchronize()
    params = [p.detach().clone() for p in [loop.detector.locations, loop.detector.dimensions, loop.detector.orientations, loop.detector.embeddings, *loop.hyper_distance_field.parameters()]]
    loop.close()
    return trace, params
ta, pa = run()
print("checksum of the first run (compare across processes):", repr(ta[-1]), float(sum(p.double().abs().sum() for p in pa)))
for trial in range(3):
    tb, pb = run()
    first = next((i for i, (x, y) in enumerate(zip(ta, tb)) if x != y), None)
    print("trial", trial, "first step whose loss differs:", first, "| parameters equal:", [bool(torch.equal(x, y)) for x, y in zip(pa, pb)][:6], "...", all(torch.equal(x, y) for x, y in zip(pa, pb)))
