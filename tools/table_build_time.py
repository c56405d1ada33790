#!/usr/bin/env python3
"""Cost of the per-frame sampling table (vsrd_ray_table_build) and of RayTable.suits at a frame-sized weight vector.  GPU box; experiments only."""
import sys, torch, time
sys.path.insert(0, '.')
import __graft_entry__; __graft_entry__.build()
from vsrd_amd import rendering
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
w = (torch.rand(17*376*1408, generator=g) - 0.62).clamp_min(0).to(dev)
pw = w[w > 0].contiguous()
for _ in range(3): t = rendering.RayTable(pw)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): t = rendering.RayTable(pw)
torch.cuda.synchronize(); print('table build: %.3f ms for %d weights' % ((time.perf_counter() - t0) * 100, pw.numel()))
t0 = time.perf_counter(); ok = t.suits(1000); torch.cuda.synchronize(); print('suits(): %.3f ms' % ((time.perf_counter() - t0) * 1e3), ok)
for _ in range(3):
    t0 = time.perf_counter(); ok = t.suits(1000); torch.cuda.synchronize(); print('suits() again: %.3f ms' % ((time.perf_counter() - t0) * 1e3), ok)
