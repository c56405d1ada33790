"""How many 16-point MLP tiles a residual step evaluates per (ray, round, instance) on the benchmark scene, one ray per wave (64
consecutive samples per round), under three ways of choosing them from the lanes that NEED the instance (exact culling criterion:
d_i - min_j d_j <= 18 T on the lane):
   rows      the tiles (16-lane rows) that contain a needed lane                       (rounds 2 and 3)
   rotation  the lanes rotated so that the needed run starts at lane 0 (residual.h: tile_plan), where that is fewer
   compact   ceil(#needed lanes / 16): the needed lanes gathered into dense tiles
from the CPU oracle (box distances only):  python tests/tile_statistics.py [schedule fraction]"""
import sys, torch
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
T, tau = sched["temperature"], 18.0


def stats(dist, name):
    mid = (dist[:, :-1] + dist[:, 1:]) / 2
    pos = cam[0] + d[:, None, :] * mid[..., None]
    rel = pos[:, :, None, :] - loc[None, None]
    local = torch.einsum('rpnk,nkj->rpnj', rel, rot)
    q = local.abs() - dim
    dist_i = (q.clamp_min(0).pow(2).sum(-1) + 1e-6).sqrt() - (-q.max(-1).values).clamp_min(0)       # [R,P,N]
    need = (dist_i - dist_i.min(-1, keepdim=True).values) <= tau * T                                  # [R,P,N]
    P = mid.shape[1]
    rows = rotation = compact = pairs = 0
    for rd in range((P + 63) // 64):
        block = need[:, rd * 64:rd * 64 + 64]                                                         # [R,<=64,N]
        width = block.shape[1]
        lanes = torch.arange(width)[None, :, None]
        any_need = block.any(1)                                                                        # [R,N]
        first = torch.where(block, lanes, 64).amin(1)
        last = torch.where(block, lanes, -1).amax(1)
        run = (last - first + 1).clamp_min(0)
        n_rows = sum(block[:, q * 16:q * 16 + 16].any(1).long() for q in range((width + 15) // 16))
        n_rot = torch.minimum(n_rows, (run + 15) // 16)
        n_compact = (block.sum(1) + 15) // 16
        rows += int(n_rows[any_need].sum()); rotation += int(n_rot[any_need].sum()); compact += int(n_compact[any_need].sum())
        pairs += int(any_need.sum())
    print(f"{name}: {pairs / R:.2f} (round, instance) pairs per ray; tiles per ray: rows {rows / R:.2f}, rotation {rotation / R:.2f} "
          f"({100 * (1 - rotation / rows):.1f} % fewer), compact {compact / R:.2f} ({100 * (1 - compact / rows):.1f} % fewer)")


stats(coarse.distances, "pass 1")
stats(fine.distances, "pass 2")
