import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from vsrd_amd import rendering, fields, models
dev = torch.device("cuda:0")
V, H, W, N, S = 9, 376, 1408, 16, 64
K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(0, V, H, W, N)
det = models.BoxParameters3D(1, N).to(dev)
with torch.no_grad():
    det.locations.copy_(raw_loc); det.dimensions.copy_(raw_dim); det.orientations.copy_(raw_ori)
    out = det()
    cam, dirs = rendering.ray_casting((H, W), K.to(dev), E.to(dev))
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
    for name, frac in bench.SCHEDULES.items():
        sc = bench.schedule_values(frac)
        block = fields.FieldBlock(fields.pack_instances(out["locations"][0], out["orientations"][0], out["dimensions"][0]), sc["std"], None, None)
        o = rendering.render_hierarchical(block, origins, dirs.reshape(-1, 3), (0.0, 100.0), S, sc["std"], sc["cosine_ratio"], seed=1, skip_exact_misses=True)
        d = o["distances"]
        miss = torch.isnan(d.reshape(-1, 2 * S)[:, 0]).float().mean().item()
        lab = o["labels"]
        print(name, "exact-miss fraction", round(miss, 4), "rays with any label>1e-6:", round((lab.max(-1).values > 1e-6).float().mean().item(), 4))
