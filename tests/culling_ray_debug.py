#!/usr/bin/env python3
"""Follow ONE ray of the culling A/B through the culling pre-pass on the host (float32, the kernels' formulas): which instance loses
weight with culling on, in which round, and what the bound test (quad_step.h: quad_round_bounds / quad_round_mask) says there.
    python tests/culling_ray_debug.py <ray> [mid] [pair]          (GPU box; test infrastructure)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
from vsrd_amd import rendering
from vsrd_amd.rendering import renderers
from test_hip_scale import scene
from oracle import geometry as ogeometry

ray = int(sys.argv[1]); schedule = sys.argv[2] if len(sys.argv) > 2 else "mid"; shape = sys.argv[3] if len(sys.argv) > 3 else "pair"
dev = torch.device("cuda:0")
N, S = (16, 64) if shape == "quad" else (64, 128)
L = 16 if shape == "quad" else 32
H, W = 376, 1408
sched = bench.schedule_values(bench.SCHEDULES[schedule]); T = sched["temperature"]
det, cam, dirs = scene(dev, N, 1, H, W, seed=0)
directions = dirs.reshape(-1, 3); origins = cam[:, None, None, :].expand(1, H, W, 3).reshape(-1, 3).contiguous()
with torch.no_grad():
    det.locations.add_(0.02)
group = ray // (64 // L) * (64 // L)
rays = list(range(group, group + 64 // L))
out = {}
for mode in ("default", "no_culling"):
    renderers.CULLING = mode == "default"
    with torch.no_grad():
        out[mode] = rendering.render_hierarchical(bench.build_union(det, T), origins, directions, (0.0, 100.0), S, sched["std"], sched["cosine_ratio"], seed=5,
                                                  stream_offset=11, skip_exact_misses=True)
renderers.CULLING = True
raw = [p.detach().float().cpu()[0] for p in (det.locations, det.dimensions, det.orientations)]
loc, dim, rot, _ = ogeometry.decode_box_parameters(*raw)
f32 = np.float32
locn, dimn, rotn = loc.numpy().astype(f32), dim.numpy().astype(f32), rot.numpy().astype(f32)
lab_a, lab_b = out["default"]["labels"][ray].cpu().numpy(), out["no_culling"]["labels"][ray].cpu().numpy()
worst = np.argsort(-np.abs(lab_a - lab_b))[:4]
print("ray", ray, "group rays", rays, "label differences (instance: culled, unculled):", [(int(i), float(lab_a[i]), float(lab_b[i])) for i in worst])
k, tau, quad_slack = f32(2e-4), f32(18.0), f32(2e-6)
margin = f32(tau * f32(T) + f32(2e-3))
radius = (np.sqrt((dimn * dimn).sum(-1)) / (f32(1) - k)).astype(f32)
dist = {r: out["default"]["distances"][r].cpu().numpy().astype(f32) for r in rays}
points = 2 * S - 1
for q in range((points + L - 1) // L):
    per_ray = {}
    for r in rays:
        o, d_ = origins[r].cpu().numpy().astype(f32), directions[r].cpu().numpy().astype(f32)
        s_idx = np.minimum(np.arange(q * L, q * L + L), points - 1)
        mid = ((dist[r][s_idx] + dist[r][s_idx + 1]) / f32(2)).astype(f32)
        e_ = (o[None] - locn).astype(f32)                                   # [N,3]
        a = (e_[:, 0] * e_[:, 0] + e_[:, 1] * e_[:, 1] + e_[:, 2] * e_[:, 2]).astype(f32)
        b = (f32(2) * (e_[:, 0] * d_[0] + e_[:, 1] * d_[1] + e_[:, 2] * d_[2])).astype(f32)
        c2 = f32(d_[0] * d_[0] + d_[1] * d_[1] + d_[2] * d_[2])
        reach0 = f32(np.sqrt(max(a.max(), float((o * o).sum()))))
        part = (mid[:, None].astype(np.float64) * b[None].astype(np.float64) + a[None].astype(np.float64)).astype(f32)      # [L,N] e_i = a + b t
        near2 = ((c2 * mid).astype(np.float64) * mid.astype(np.float64) + part.min(1).astype(np.float64)).astype(f32)
        s_ = (reach0 + f32(np.sqrt(c2)) * np.abs(mid)).astype(f32)
        err = (quad_slack * s_ * s_).astype(f32)
        hi = (np.sqrt(np.maximum(near2, 0) + err).astype(f32) * (f32(1) + k)).astype(f32)
        limit = ((hi + margin) * (f32(1) / (f32(1) - k))).astype(f32)
        shift = (err - (c2 * mid * mid).astype(f32)).astype(f32)
        reach = (limit[:, None] + radius[None]).astype(f32)
        keep = ~(part > (reach.astype(np.float64) * reach.astype(np.float64) + shift[:, None].astype(np.float64)).astype(f32))          # [L,N]
        # exact: box distances and soft-min weights (float64)
        x = o[None].astype(np.float64) + d_[None].astype(np.float64) * mid[:, None].astype(np.float64)
        rel = x[:, None, :] - locn[None].astype(np.float64)
        local = np.einsum('pnk,nkj->pnj', rel, rotn.astype(np.float64))
        qq = np.abs(local) - dimn[None].astype(np.float64)
        dbox = np.sqrt((np.maximum(qq, 0) ** 2).sum(-1) + 1e-6) - np.maximum(-qq.max(-1), 0)
        w = np.exp(-(dbox - dbox.min(1, keepdims=True)) / T); w /= w.sum(1, keepdims=True)
        per_ray[r] = (keep, w, dbox, mid)
    wave_keep = np.any(np.concatenate([per_ray[r][0] for r in rays], 0), 0)                   # [N]: the wave-uniform mask of the bound stage
    for i in worst:
        wmax = max(float(per_ray[r][1][:, i].max()) for r in rays)
        if wmax > 1e-6:
            r0 = ray
            gap = float((per_ray[r0][2][:, i] - per_ray[r0][2].min(1)).min())
            print(f"  round {q}: instance {int(i)} max soft-min weight {wmax:.3e}, smallest gap d_i - min d on this ray {gap:.3f} m (18 T = {18 * T:.2f}); "
                  f"bound stage keeps it: {bool(wave_keep[i])}; t range {float(per_ray[r0][3].min()):.2f}..{float(per_ray[r0][3].max()):.2f}")
