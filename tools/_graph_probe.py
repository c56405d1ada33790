import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from vsrd_amd import rendering, fields, models
dev = torch.device("cuda:0")
N = 8
K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(0, 1, 64, 64, N)
det = models.BoxParameters3D(1, N).to(dev)
with torch.no_grad():
    det.locations.copy_(raw_loc); det.dimensions.copy_(raw_dim); det.orientations.copy_(raw_ori)
cam, dirs = rendering.ray_casting((64, 64), K.to(dev), E.to(dev))
origins = cam[:, None, None, :].expand(1, 64, 64, 3).reshape(-1, 3).contiguous()
dirs = dirs.reshape(-1, 3).contiguous()
targets = torch.rand(64 * 64, N, device=dev)
opt = torch.optim.Adam(det.parameters(), lr=1e-2, capturable=True)
def step():
    opt.zero_grad(set_to_none=False)
    out = det()
    block = fields.FieldBlock(fields.pack_instances(out["locations"][0], out["orientations"][0], out["dimensions"][0]), 0.5, None, None)
    loss = rendering.silhouette_step(block, origins, dirs, targets, (0.0, 100.0), 64, 0.5, 0.5, seed=1, stream_offset=0)
    loss.backward()
    opt.step()
    return loss
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        l = step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager loss", float(l))
g = torch.cuda.CUDAGraph()
for p in det.parameters():
    p.grad = torch.zeros_like(p)
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
for i in range(5):
    g.replay()
torch.cuda.synchronize()
print("replayed loss", float(static_loss))
import time
t0 = time.perf_counter()
for i in range(200):
    g.replay()
torch.cuda.synchronize()
print("graph replay ms/step", (time.perf_counter() - t0) / 200 * 1e3)
t0 = time.perf_counter()
for i in range(200):
    step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 200 * 1e3)
