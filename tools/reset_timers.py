#!/usr/bin/env python3
"""Where FrameOptimizer.reset() spends its time at the reference's native size (17 views of 376x1408, N = 8): each part timed with a
device synchronisation around it.  python tools/reset_timers.py   (GPU box)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import __graft_entry__
    __graft_entry__.build()
    from vsrd_amd import launcher, models, optimization, rendering
    dev = torch.device("cuda:0")
    inputs = [launcher.synthetic_frame_inputs(dev, k, 17, 8) for k in range(3)]
    cfg = optimization.OptimizationConfig()
    loop = optimization.FrameOptimizer(inputs[0], cfg, dev, graph=True, persistent=True)
    loop.capture_all()

    def timed(label, fn, repeats=3):
        best = 1e9
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"{label:60s} {best * 1e3:9.3f} ms")
        return out

    for k in (1, 2, 1):
        timed(f"reset(frame {k}) whole", lambda: loop.reset(inputs[k], init_seed=k), repeats=1)
    if "--threads1" in sys.argv:
        torch.set_num_threads(1)
        for k in (1, 2, 1):
            timed(f"torch.set_num_threads(1): reset(frame {k}) whole", lambda: loop.reset(inputs[k], init_seed=k), repeats=1)
    # the pieces in reset's own order, once each, a synchronisation behind each
    def once(label, fn):
        t0 = time.perf_counter()
        out = fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"  in order: {label:50s} host {1e3 * (t1 - t0):8.3f} ms, + device {1e3 * (time.perf_counter() - t1):8.3f} ms")
        return out
    inp = inputs[2]
    once("ray_casting", lambda: rendering.ray_casting(inp.image_size, inp.intrinsic_matrices, inp.extrinsic_matrices))
    once("flat_masks.copy_", lambda: loop.flat_masks.copy_(inp.soft_masks.reshape(-1, 8)))
    once("amax", lambda: torch.amax(loop.flat_masks, dim=-1, out=loop.sampling_weights))
    fresh = once("fresh hypernetwork (host)", lambda: models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]))
    once("parameters -> device", lambda: [p.data.copy_(q) for p, q in zip(loop.hyper_distance_field.parameters(), fresh.parameters())])
    once("positive count", lambda: int((loop.sampling_weights > 0).sum()))
    once("table rebuild", lambda: loop.ray_table.rebuild(loop.sampling_weights))
    once("suits: clamp_min().double()", lambda: loop.sampling_weights.clamp_min(0).double())
    positive = loop.sampling_weights.clamp_min(0).double()
    once("suits: sum", lambda: float(positive.sum()))
    once("suits: topk 999", lambda: float(torch.topk(positive, 999).values.sum()))
    import cProfile
    import pstats
    profile = cProfile.Profile()
    profile.enable()
    loop.reset(inputs[2], init_seed=2)
    profile.disable()
    pstats.Stats(profile).sort_stats("cumulative").print_stats(28)
    inp = inputs[1]
    H, W = inp.image_size
    N = 8
    timed("ray_casting", lambda: rendering.ray_casting((H, W), inp.intrinsic_matrices, inp.extrinsic_matrices))
    timed("flat_masks.copy_", lambda: loop.flat_masks.copy_(inp.soft_masks.reshape(-1, N)))
    timed("amax -> sampling_weights", lambda: torch.amax(loop.flat_masks, dim=-1, out=loop.sampling_weights))
    timed("fresh modules on the host", lambda: (models.BoxParameters3D(1, N), models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256])))
    fresh = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256])
    timed("copy the hypernetwork's parameters to the device", lambda: [p.data.copy_(q) for p, q in zip(loop.hyper_distance_field.parameters(), fresh.parameters())])
    timed("zero Adam's state", lambda: [[s.zero_() for s in (loop.optimizer.state[p]["exp_avg"], loop.optimizer.state[p]["exp_avg_sq"], loop.optimizer.state[p]["step"])]
                                        for g in loop.optimizer.param_groups for p in g["params"]])
    timed("ray_table.rebuild", lambda: loop.ray_table.rebuild(loop.sampling_weights))
    timed("ray_table.suits", lambda: loop.ray_table.suits(cfg.num_rays))
    timed("positive count (.sum -> int)", lambda: int((loop.sampling_weights > 0).sum()))


if __name__ == "__main__":
    main()
