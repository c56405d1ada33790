"""Per round of the four-rays-per-wave kernels (4 consecutive rays x 16 samples) on the benchmark scene: instances past the bound test (candidates),
past the kernel's sequential exact test (survivors: d_i - best <= 18 T on some point, best = running minimum over the survivors so far, starting
from the nearest centre distance), and instances some point of the round really needs (d_i - min_j d_j <= 18 T).  CPU oracle; analysis tool like
cull_statistics.py (profiles/r06/variants.txt):  python tests/survivor_statistics.py [schedule fraction]"""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from oracle import fields as ofields, rendering as orendering, geometry as ogeometry
torch.manual_seed(0)
H, W, N, S, V = 376, 1408, 16, 64, 9
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
sched = bench.schedule_values(frac)
K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(0, V, H, W, N)
loc, dim, rot, _ = ogeometry.decode_box_parameters(raw_loc[0], raw_dim[0], raw_ori[0])
cam, dirs = ogeometry.ray_casting((H, W), K[:1], E[:1])
alld = dirs[0].reshape(-1, 3)
# groups of four consecutive pixels, every 389th group
starts = torch.arange(0, alld.shape[0] - 4, 4 * 389)
idx = (starts[:, None] + torch.arange(4)[None]).reshape(-1)
d = alld[idx]
R = d.shape[0]
union = ofields.InstanceUnion(loc, rot, dim, sched["temperature"])
coarse, fine = orendering.hierarchical_render(union, cam[0], d, (0.0, 100.0), S, sched["std"], sched["cosine_ratio"], torch.rand(R, S), torch.rand(R, S), return_coarse=True)
T = sched["temperature"]; cull = 18.0 * T; k = 2e-4
radius = dim.norm(dim=-1); widest = radius.max()
def stats(dist, name):
    mid = (dist[:, :-1] + dist[:, 1:]) / 2
    pos = cam[0] + d[:, None, :] * mid[..., None]
    rel = pos[:, :, None, :] - loc[None, None]
    local = torch.einsum('rpnk,nkj->rpnj', rel, rot)
    q = local.abs() - dim
    dist_i = (q.clamp_min(0).pow(2).sum(-1) + 1e-6).sqrt() - (-q.max(-1).values).clamp_min(0)   # [R,P,N]
    centre = rel.norm(dim=-1)
    nearest_hi = centre.min(-1).values * (1 + k)
    limit = nearest_hi + cull + 2e-3
    P = mid.shape[1]
    G = R // 4
    rounds = (P + 15) // 16
    tc = ts = tn = tr = 0
    hist = torch.zeros(20)
    for g in range(G):
        for rd in range(rounds):
            sl = slice(rd * 16, min(P, rd * 16 + 16))
            di = dist_i[4 * g:4 * g + 4, sl].reshape(-1, N)        # [64, N]
            ce = centre[4 * g:4 * g + 4, sl].reshape(-1, N)
            lim = limit[4 * g:4 * g + 4, sl].reshape(-1)
            best = nearest_hi[4 * g:4 * g + 4, sl].reshape(-1).clone()
            cand = ((ce * (1 - k) - widest) <= lim[:, None]).any(0)
            m = di.min(-1).values
            need = ((di - m[:, None]) <= cull).any(0)
            surv = 0
            for i in range(N):
                if not cand[i]: continue
                if ((di[:, i] - best) <= cull).any():
                    surv += 1
                    best = torch.minimum(best, di[:, i])
            tc += int(cand.sum()); ts += surv; tn += int((need & cand).sum()); tr += 1
            hist[int(cand.sum())] += 1
    print(f"{name}: rounds {tr}: candidates {tc / tr:.2f}  survivors (sequential test) {ts / tr:.2f}  needed by some point {tn / tr:.2f};  rounds with >= 5 candidates: {hist[5:].sum() / tr:.2f}")
stats(coarse.distances, "pass1")
stats(fine.distances, "pass2")
