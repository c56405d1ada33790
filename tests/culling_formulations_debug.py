#!/usr/bin/env python3
"""Both formulations of the culling bound test of quad_step.h (nested: d2 = a + t (b + c t); partial: e = a + b t against E - c t^2)
emulated in float32 on the device for EVERY (group of rays, round, instance) of one view, at the kernel's own pass-2 distances: where do
the wave-uniform masks differ, and does either drop an instance whose soft-min weight is not negligible?
    python tests/culling_formulations_debug.py [mid] [pair]           (GPU box; test infrastructure)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from vsrd_amd import rendering
from test_hip_scale import scene
from oracle import geometry as ogeometry

schedule = sys.argv[1] if len(sys.argv) > 1 else "mid"; shape = sys.argv[2] if len(sys.argv) > 2 else "pair"
dev = torch.device("cuda:0")
N, S = (16, 64) if shape == "quad" else (64, 128)
L = 16 if shape == "quad" else 32
G = 64 // L
H, W = 376, 1408
sched = bench.schedule_values(bench.SCHEDULES[schedule]); T = sched["temperature"]
det, cam, dirs = scene(dev, N, 1, H, W, seed=0)
directions = dirs.reshape(-1, 3); origins = cam[:, None, None, :].expand(1, H, W, 3).reshape(-1, 3).contiguous()
with torch.no_grad():
    det.locations.add_(0.02)
    out = rendering.render_hierarchical(bench.build_union(det, T), origins, directions, (0.0, 100.0), S, sched["std"], sched["cosine_ratio"], seed=5,
                                        stream_offset=11, skip_exact_misses=True)
    loc, dim, rot, _ = ogeometry.decode_box_parameters(det.locations[0].float(), det.dimensions[0].float(), det.orientations[0].float())
dist = out["distances"]
R = dist.shape[0]
def fma(a, b, c): return (a.double() * b.double() + c.double()).float()
k = 2e-4; margin = 18.0 * T + 2e-3
radius = dim.norm(dim=-1) / (1 - k)
P = 2 * S - 1
rounds = (P + L - 1) // L
chunk = 32768
bad = {"nested": 0, "partial": 0}; differ = 0; total = 0
worst = []
for start in range(0, R, chunk):
    sl = slice(start, min(R, start + chunk))
    o, d_, ds = origins[sl], directions[sl], dist[sl]
    n = o.shape[0]
    e_ = o[:, None, :] - loc[None]
    a = e_[..., 0] * e_[..., 0] + e_[..., 1] * e_[..., 1] + e_[..., 2] * e_[..., 2]                 # [n,N]
    b = 2.0 * (e_[..., 0] * d_[:, None, 0] + e_[..., 1] * d_[:, None, 1] + e_[..., 2] * d_[:, None, 2])
    c2 = (d_ * d_).sum(-1)
    reach0 = torch.sqrt(torch.maximum(a.max(1).values, (o * o).sum(-1)))
    for q in range(rounds):
        idx = torch.arange(q * L, q * L + L, device=dev).clamp_max(P - 1)
        mid = (ds[:, idx] + ds[:, idx + 1]) / 2.0                                                  # [n,L]
        t3 = mid[:, :, None]
        s_ = reach0[:, None] + c2.sqrt()[:, None] * mid.abs()
        err = 2e-6 * s_ * s_
        keep = {}
        for form in ("nested", "partial"):
            if form == "nested":
                val = fma(t3.expand(-1, -1, N), (c2[:, None] * mid)[:, :, None] + b[:, None, :], a[:, None, :].expand(-1, L, -1))
                near2 = val.min(-1).values
                shift = err
            else:
                val = fma(t3.expand(-1, -1, N), b[:, None, :].expand(-1, L, -1), a[:, None, :].expand(-1, L, -1))
                near2 = fma(c2[:, None] * mid, mid, val.min(-1).values)
                shift = err - c2[:, None] * mid * mid
            hi = torch.sqrt(near2.clamp_min(0) + err) * (1 + k)
            limit = (hi + margin) * (1.0 / (1 - k))
            reach = limit[:, :, None] + radius[None, None, :]
            lane_keep = ~(val > fma(reach, reach, shift[:, :, None].expand(-1, -1, N)))           # [n,L,N]
            keep[form] = lane_keep.reshape(n // G, G * L, N).any(1)                                # wave-uniform: [groups,N]
        # exact soft-min weights (float64) of every point
        x = o[:, None, :].double() + d_[:, None, :].double() * mid[:, :, None].double()
        rel = x[:, :, None, :] - loc[None, None].double()
        local = torch.einsum('rpnk,nkj->rpnj', rel, rot.double())
        qq = local.abs() - dim.double()
        dbox = (qq.clamp_min(0).pow(2).sum(-1) + 1e-6).sqrt() - (-qq.max(-1).values).clamp_min(0)
        w = torch.softmax(-dbox / T, dim=-1).reshape(n // G, G * L, N).amax(1)                      # [groups,N] largest weight in the wave's round
        for form in keep:
            lost = (~keep[form]) & (w > 1e-7)
            bad[form] += int(lost.sum())
            if lost.any() and len(worst) < 5:
                g, i = [int(v[0]) for v in torch.nonzero(lost, as_tuple=True)]
                worst.append((form, start // G + g, q, i, float(w[g, i])))
        differ += int((keep["nested"] != keep["partial"]).sum()); total += keep["nested"].numel()
    del e_, a, b
print(f"{shape} {schedule}: (group, round, instance) triples {total}; masks differ in {differ}; dropped with weight > 1e-7: {bad}; examples {worst}")
