"""How many instances survive the culling bounds per round / per 32-, 16-, 8-sample segment / per sample on the benchmark scene
(DESIGN.md sections 2 and 9), bounds against the exact criterion, from the CPU oracle:  python tests/cull_statistics.py [schedule fraction]"""
import sys, math, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from oracle import fields as ofields, rendering as orendering, geometry as ogeometry
torch.manual_seed(0)
H, W, N, S, V = 376, 1408, 16, 64, 9
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
sched = bench.schedule_values(frac)
K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(0, V, H, W, N)
loc, dim, rot, _ = ogeometry.decode_box_parameters(raw_loc[0], raw_dim[0], raw_ori[0])
cam, dirs = ogeometry.ray_casting((H, W), K[:1], E[:1])
d = dirs[0].reshape(-1, 3)[::97]
R = d.shape[0]
union = ofields.InstanceUnion(loc, rot, dim, sched["temperature"])
coarse, fine = orendering.hierarchical_render(union, cam[0], d, (0.0, 100.0), S, sched["std"], sched["cosine_ratio"], torch.rand(R, S), torch.rand(R, S), return_coarse=True)
T = sched["temperature"]; tau = 18.0; k = 2e-4
margin = tau * T + 2e-3
radius = dim.norm(dim=-1)
def stats(dist, name):
    mid = (dist[:, :-1] + dist[:, 1:]) / 2
    pos = cam[0] + d[:, None, :] * mid[..., None]          # [R,P,3]
    di, _, _ = union.instance_terms(pos) if False else (None, None, None)
    # exact per-instance distances
    rel = pos[:, :, None, :] - loc[None, None]              # [R,P,N,3]
    local = torch.einsum('rpnk,nkj->rpnj', rel, rot)
    q = local.abs() - dim
    dist_i = (q.clamp_min(0).pow(2).sum(-1) + 1e-6).sqrt() - (-q.max(-1).values).clamp_min(0)   # [R,P,N]
    centre = rel.norm(dim=-1)
    lb = centre * (1 - k) - radius
    ub = centre.min(-1, keepdim=True).values * (1 + k) + margin
    P = mid.shape[1]
    rounds = (P + 63) // 64
    tot_b = tot_e = tot_any = 0
    for rd in range(rounds):
        sl = slice(rd * 64, min(P, rd * 64 + 64))
        near_b = (lb[:, sl] <= ub[:, sl]).any(1)            # [R,N]
        m = dist_i[:, sl].min(-1, keepdim=True).values
        near_e = ((dist_i[:, sl] - m) <= tau * T).any(1)
        tot_b += near_b.sum().item(); tot_e += (near_e & near_b).sum().item()
    per_sample = (lb <= ub).float().sum(-1).mean().item()
    for seg in (32, 16, 8):
        nseg = (P + seg - 1) // seg
        tot = 0
        for rd in range(nseg):
            sl = slice(rd * seg, min(P, rd * seg + seg))
            tot += (lb[:, sl] <= ub[:, sl]).any(1).sum().item()
        print(f"   segment {seg}: avg active/segment (bounds) = {tot / (R * nseg):.2f}")
    print(f"   per-sample near (bounds) = {per_sample:.2f}")
    print(f"{name}: rounds={rounds} avg active/round bounds={tot_b / (R * rounds):.2f} exact={tot_e / (R * rounds):.2f}")
stats(coarse.distances, "pass1")
stats(fine.distances, "pass2")
