"""VERDICT r03 item 7, before building it: how many of BASELINE config 5's 64 instances would a per-ray (image-space) candidate mask
keep?  A ray-level test can only use what holds for EVERY sample of the ray: instance i matters somewhere on a ray only where the
nearest box is within rho = 17.5 sigma + max|dim| + 1 (beyond that both logistic cdfs are exactly 1: quad_step.h, "rounds that see
nothing"), and there only if its centre is within  rho + 18 T + |dim_i|  of the sample -- so never, if the ray passes its centre at more
than that distance.  Counts per ray on the benchmark scene, against what the per-round pre-pass keeps today (tests/cull_statistics.py):
    python tests/candidate_statistics.py"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from oracle import geometry as ogeometry
V, H, W, N = 17, 752, 2816, 64
K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(0, V, H, W, N)
loc, dim, rot, _ = ogeometry.decode_box_parameters(raw_loc[0], raw_dim[0], raw_ori[0])
radius = dim.norm(dim=-1)
for name, fraction in (("start", 0.0), ("mid", 0.5), ("end", 1.0)):
    sched = bench.schedule_values(fraction)
    T, sigma = sched["temperature"], sched["std"]
    rho = 17.5 * sigma + float(radius.max()) + 1.0
    kept = []
    for view in (0, 5, 12):
        cam, dirs = ogeometry.ray_casting((H, W), K[view:view + 1], E[view:view + 1])
        d = dirs[0].reshape(-1, 3)[::4999]
        rel = loc[None] - cam[0][None, None]                                   # [1,N,3]
        t = (rel * d[:, None, :]).sum(-1).clamp(0.0, 100.0)                    # closest approach within the sampled range
        closest = (rel - t[..., None] * d[:, None, :]).norm(dim=-1)           # [R,N]
        kept.append((closest <= rho + 18.0 * T + radius[None]).float().sum(-1))
    kept = torch.cat(kept)
    print(f"{name}: T = sigma = {T:.2f}: a ray-level candidate mask keeps {kept.mean():.1f} of {N} instances on average (median {kept.median():.0f}, "
          f"max {kept.max():.0f}); bound radius rho + 18 T = {rho + 18 * T:.1f} m + |dim|")
