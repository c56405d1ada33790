"""Debug script (GPU box): BASELINE config 3 at full size through the fused residual step with the exact-fp32 and the split-bf16 products of the
per-instance MLP in ONE process on the same uniforms -- losses, per-ray label differences (the share beyond 1e-4 is the conditioning tail the
parity tests allow for) and parameter gradients side by side.  `VIEWS=1 python tools/compare_forms_debug.py` for a quick run."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from test_hip_scale import scene
from vsrd_amd import models, rendering
dev = torch.device("cuda:0")
N, S, V, H, W, seed = 16, 64, int(os.environ.get("VIEWS", 9)), 376, 1408, 3
sched = bench.schedule_values(bench.SCHEDULES["mid"])
T, std, ratio = sched["temperature"], sched["std"], sched["cosine_ratio"]
det, cam, dirs = scene(dev, N, V, H, W, seed=seed)
directions = dirs.reshape(-1, 3)
origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
R = directions.shape[0]
torch.manual_seed(0)
hyper = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).to(dev)
with torch.no_grad():
    targets = rendering.render_hierarchical(bench.build_union(det, 0.1), origins, directions, (0.0, 100.0), S, 0.1, 1.0, seed=99, skip_exact_misses=True)["labels"].clamp(0.0, 1.0).contiguous()
    det.locations.add_(0.02)
    generator = torch.Generator(device=dev).manual_seed(77)
    u_coarse = torch.rand(R, S, device=dev, generator=generator)
    u_fine = torch.rand(R, S, device=dev, generator=generator)
out = {}
for split in (False, True, False):
    union = bench.build_union(det, T)
    union.mlp_weights = hyper(det.embeddings)[0].contiguous()
    params = [det.locations, det.dimensions, det.orientations, det.embeddings]
    loss, terms, labels = rendering.silhouette_step(union, origins, directions, targets, (0.0, 100.0), S, std, ratio, u_coarse=u_coarse, u_fine=u_fine,
                                                    eikonal_ratio=0.01, return_labels=True, return_terms=True, skip_exact_misses=False, mlp_split_bf16=split)
    grads = torch.autograd.grad(loss, params)
    print("split" if split else "fp32 ", "loss", float(loss.detach()), "terms", terms.tolist(), "weights checksum", float(union.mlp_weights.double().sum()))
    out.setdefault(split, (labels.clone(), [g.clone() for g in grads]))
a, b = out[False], out[True]
diff = (a[0] - b[0]).abs().max(-1).values
print("labels fp32 vs split: worst", float(diff.max()), "rays > 1e-4:", int((diff > 1e-4).sum()), "> 1e-5:", int((diff > 1e-5).sum()), "of", R)
for name, x, y in zip(("locations", "dimensions", "orientations", "embeddings"), a[1], b[1]):
    print("grad", name, float((x - y).abs().max()), "of", float(x.abs().max()))
