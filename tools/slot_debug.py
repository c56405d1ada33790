#!/usr/bin/env python3
"""Frame slots against fresh loops on the launcher's synthetic frames, sequentially: final losses per frame (debugging aid, GPU box)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__
__graft_entry__.build()
from vsrd_amd import launcher, optimization

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
config = dict(num_steps=steps, warmup_steps=warm, num_rays=1000, num_samples=100)
frames = {f: launcher.synthetic_frame_inputs(dev, f, 17, 8) for f in range(4)}
torch.manual_seed(0)
slot = optimization.FrameOptimizer(frames[0], optimization.OptimizationConfig(seed=0, **config), dev, graph=True, persistent=True)
t0 = time.perf_counter()
print("graphs", slot.capture_all(), "setup s", time.perf_counter() - t0, flush=True)
for f in (1, 2, 3, 1):
    torch.manual_seed(10 + f)
    t0 = time.perf_counter()
    ok = slot.reset(frames[f])
    t1 = time.perf_counter()
    marks = []
    for upto in (warm, warm + 4, warm + 40, steps):
        out = slot.run(upto - slot.step_index)
        torch.cuda.synchronize()
        marks.append((upto, float(out["loss"])))
    t2 = time.perf_counter()
    print(f"slot  frame {f}: reset {ok} {t1 - t0:.3f} s, run {t2 - t1:.3f} s, losses {marks}", flush=True)
    torch.manual_seed(10 + f)
    t0 = time.perf_counter()
    loop = optimization.FrameOptimizer(frames[f], optimization.OptimizationConfig(seed=0, **config), dev, graph=True)
    marks = []
    for upto in (warm, warm + 4, warm + 40, steps):
        out = loop.run(upto - loop.step_index)
        torch.cuda.synchronize()
        marks.append((upto, float(out["loss"])))
    print(f"fresh frame {f}: {time.perf_counter() - t0:.3f} s, losses {marks}", flush=True)
    loop.close()
