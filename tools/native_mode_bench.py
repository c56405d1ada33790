#!/usr/bin/env python3
"""Steps/s of the whole per-frame loop (vsrd_amd.optimization.FrameOptimizer) in the reference's NATIVE mode:
1 target + 16 source views of 376x1408, 1000 importance-sampled rays per step, 100 samples per ray
(configs/kitti_360/vsrd/*/config.json:16,22-23,236-237), box-only warm-up phase.  The reference's only published
number for this loop is "about 15 minutes per frame on a V100" for 3000 steps = 3.3 steps/s (README.md:128).

  python tools/native_mode_bench.py [--steps 200] [--instances 8]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--steps", type=int, default=200)
    parser.add_argument("--instances", type=int, default=8)
    parser.add_argument("--views", type=int, default=17)
    parser.add_argument("--rays", type=int, default=1000, help="rays per step (the reference: 1000)")
    parser.add_argument("--graph", action="store_true", help="capture the step in a hipGraph and replay it")
    parser.add_argument("--concurrent", type=int, default=1, help="frames optimised at the same time (one host thread and stream each)")
    parser.add_argument("--residual", action="store_true", help="post-warm-up phase: residual MLP + eikonal loss (steps 1000-3000)")
    parser.add_argument("--json", action="store_true", help="also print one JSON line (tools/regimes.py)")
    parser.add_argument("--fp32-mlp", action="store_true", help="OptimizationConfig(mlp_split_bf16=False): the residual MLP on the exact-fp32 matrix "
                        "instruction instead of the split-bf16 products that are the loop's default since round 5")
    parser.add_argument("--steps-per-graph", type=int, default=4, help="graph mode: consecutive steps replayed per hipGraph launch (FrameOptimizer.run); 1 = one launch per step")
    parser.add_argument("--batch", type=int, default=0, help="with --whole-frame: B frames in lock-step through optimization.FrameBatch (one launch of every kernel of a "
                        "step for all of them); the time reported is per frame")
    parser.add_argument("--whole-frame", action="store_true", help="time one whole frame as the reference runs it: steps 0..2999 with the real schedules "
                        "(1000 box-only warm-up steps, then 2000 residual steps), set-up and graph captures included")
    args = parser.parse_args()
    import __graft_entry__
    __graft_entry__.build()
    import bench
    from vsrd_amd import optimization, rendering, fields, models, operations
    dev = torch.device("cuda:0")
    V, H, W, N = args.views, 376, 1408, args.instances
    K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(0, V, H, W, N)
    det = models.BoxParameters3D(1, N).to(dev)
    with torch.no_grad():
        det.locations.copy_(raw_loc); det.dimensions.copy_(raw_dim); det.orientations.copy_(raw_ori)
        out = det()
        cam, dirs = rendering.ray_casting((H, W), K.to(dev), E.to(dev))
        block = fields.FieldBlock(fields.pack_instances(out["locations"][0], out["orientations"][0], out["dimensions"][0]), 0.1, None, None)
        origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
        soft = rendering.render_hierarchical(block, origins, dirs.reshape(-1, 3), (0.0, 100.0), 64, 0.1, 1.0, seed=1,
                                             skip_exact_misses=True)["labels"].clamp(0, 1).reshape(V, H, W, N).contiguous()
        gt_boxes, _ = operations.project_boxes_multi_view(out["boxes_3d"][0], E.to(dev), K.to(dev), (H, W))
    inputs = optimization.FrameInputs((H, W), K.to(dev), E.to(dev), soft, gt_boxes, torch.ones(V, N, dtype=torch.bool, device=dev))
    if args.whole_frame and args.batch > 0:
        cfg = optimization.OptimizationConfig(seed=0, num_rays=args.rays, mlp_split_bf16=not args.fp32_mlp)
        t_setup = time.perf_counter()
        batch = optimization.FrameBatch([inputs] * args.batch, cfg, dev, init_seeds=list(range(args.batch)))
        batch.capture_all(args.steps_per_graph)
        torch.cuda.synchronize()
        setup = time.perf_counter() - t_setup
        times = []
        for turn in range(2):                                # the first turn warms; the second is reported
            for row in range(args.batch):
                batch.reset(row, inputs, init_seed=100 * turn + row)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            batch.run(cfg.warmup_steps, args.steps_per_graph)
            torch.cuda.synchronize(); warm = time.perf_counter() - t0
            batch.run(cfg.num_steps - cfg.warmup_steps, args.steps_per_graph)
            torch.cuda.synchronize(); total = time.perf_counter() - t0
            times.append((warm, total))
        warm, total = times[-1]
        losses = [float(batch.outputs(row)["loss"]) for row in range(args.batch)]
        B = args.batch
        print(f"native mode, whole frames in a batch of {B} ({cfg.num_steps} steps = {cfg.warmup_steps} box-only + {cfg.num_steps - cfg.warmup_steps} residual, real schedules, "
              f"{args.rays} rays x {cfg.num_samples} samples per frame, V={V}, N={N}, hipGraph replay): {total:.2f} s for {B} frames = {total / B:.3f} s per frame "
              f"({warm / B:.3f} s warm-up phase, {(total - warm) / B:.3f} s residual phase = {(total - warm) / B / (cfg.num_steps - cfg.warmup_steps) * 1e3:.3f} ms per frame-step); "
              f"set-up (construction + captures) {setup:.2f} s; final losses {', '.join(f'{x:.4f}' for x in losses)}")
        if args.json:
            import json
            print(json.dumps(dict(mode="native", phase="whole frame", graph=True, frame_batch=B, seconds_per_frame=total / B, seconds_per_batch=total,
                                  warmup_phase_seconds=warm / B, residual_phase_seconds=(total - warm) / B, steps=cfg.num_steps, rays_per_step=args.rays,
                                  samples_per_ray=cfg.num_samples, views=V, instances=N, final_loss=losses[0], final_losses=losses, setup_seconds=setup,
                                  steps_per_graph=args.steps_per_graph, mlp_products="exact fp32 MFMA" if args.fp32_mlp else "split bf16 MFMA")))
        return
    if args.batch > 0:
        # B frames in lock-step, `--steps` steps of one phase at its first step's schedule (what the PMC passes profile: tools/pmc_quick.sh)
        cfg = optimization.OptimizationConfig(seed=0, num_rays=args.rays, mlp_split_bf16=not args.fp32_mlp)
        batch = optimization.FrameBatch([inputs] * args.batch, cfg, dev, init_seeds=list(range(args.batch)))
        start = cfg.warmup_steps if args.residual else 0
        for member in batch.frames:
            member.step_index = start
            member.step_tensor.fill_(start)
        for _ in range(4):
            batch.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        batch.run(args.steps, args.steps_per_graph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"native mode ({'residual' if args.residual else 'box-only'} phase, hipGraph replay, batch of {args.batch} frames): {dt / args.steps * 1e3:.3f} ms per step of the batch = "
              f"{dt / args.steps / args.batch * 1e3:.4f} ms per frame-step ({args.rays} rays x 100 samples per frame, V={V}, N={N})")
        if args.json:
            import json
            print(json.dumps(dict(mode="native", phase="residual" if args.residual else "box-only", graph=True, frame_batch=args.batch, ms_per_step=dt / args.steps * 1e3,
                                  ms_per_frame_step=dt / args.steps / args.batch * 1e3, rays_per_step=args.rays, samples_per_ray=100, views=V, instances=N,
                                  steps_per_graph=args.steps_per_graph, mlp_products="exact fp32 MFMA" if args.fp32_mlp else "split bf16 MFMA")))
        return
    if args.whole_frame:
        def frame(slot):
            loop = optimization.FrameOptimizer(inputs, optimization.OptimizationConfig(seed=slot, num_rays=args.rays, mlp_split_bf16=not args.fp32_mlp), dev, graph=args.graph)
            marks = []
            loop.run(loop.config.warmup_steps, args.steps_per_graph)
            torch.cuda.synchronize(); marks.append(time.perf_counter())
            losses = loop.run(loop.config.num_steps - loop.config.warmup_steps, args.steps_per_graph)
            torch.cuda.synchronize(); marks.append(time.perf_counter())
            return marks, float(losses["loss"]), loop
        frame(0)[2].close()                                  # warm the allocator and the code objects
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks, loss, loop = frame(1)
        warm, total = marks[0] - t0, marks[1] - t0
        cfg = loop.config
        print(f"native mode, whole frame ({cfg.num_steps} steps = {cfg.warmup_steps} box-only + {cfg.num_steps - cfg.warmup_steps} residual, real schedules, "
              f"{args.rays} rays x {cfg.num_samples} samples, V={V}, N={N}{', hipGraph replay' if args.graph else ''}): {total:.2f} s "
              f"({warm:.2f} s warm-up phase, {total - warm:.2f} s residual phase = {(total - warm) / (cfg.num_steps - cfg.warmup_steps) * 1e3:.2f} ms/step); "
              f"reference: about 15 minutes on a V100 (README.md:128); final loss {loss:.4f}")
        if args.json:
            import json
            print(json.dumps(dict(mode="native", phase="whole frame", graph=bool(args.graph), seconds_per_frame=total, warmup_phase_seconds=warm,
                                  residual_phase_seconds=total - warm, steps=cfg.num_steps, rays_per_step=args.rays, samples_per_ray=cfg.num_samples,
                                  views=V, instances=N, final_loss=loss, steps_per_graph=args.steps_per_graph if args.graph else None,
                                  mlp_products="exact fp32 MFMA" if args.fp32_mlp else "split bf16 MFMA")))
        return
    import threading
    # Frames are independent (README.md:128: no exchange): several can be optimised at once, each on its own stream with its own
    # replayed graph, and fill each other's idle SIMDs.  Set-up and capture are serial (stream capture is process-global).
    loops, streams = [], []
    for slot in range(args.concurrent):
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            loop = optimization.FrameOptimizer(inputs, optimization.OptimizationConfig(seed=slot, num_rays=args.rays, mlp_split_bf16=not args.fp32_mlp), dev, graph=args.graph)
            if args.residual:
                loop.step_index = loop.config.warmup_steps
                loop.step_tensor.fill_(loop.step_index)
            for _ in range(20):
                loop.step()
        stream.synchronize()
        loops.append(loop); streams.append(stream)
    results = [None] * args.concurrent
    start_line = threading.Barrier(args.concurrent + 1)

    def worker(slot):
        with torch.cuda.stream(streams[slot]):
            start_line.wait()
            losses = loops[slot].run(args.steps, args.steps_per_graph)
            with optimization.exclusive_device_access():       # (host synchronisations: not next to the other frame's capture)
                streams[slot].synchronize()
                results[slot] = float(losses["loss"])

    threads = [threading.Thread(target=worker, args=(slot,)) for slot in range(args.concurrent)]
    for t in threads:
        t.start()
    start_line.wait()
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    total = args.steps * args.concurrent
    print(f"native mode ({'residual' if args.residual else 'box-only'} phase{', hipGraph replay' if args.graph else ''}"
          f"{', %d frames at once' % args.concurrent if args.concurrent > 1 else ''}): {total / dt:.1f} steps/s ({dt / total * 1e3:.2f} ms/step, "
          f"{args.rays} rays x 100 samples, V={V}, N={N}); 3000-step frame = {3000 * dt / total:.1f} s; reference: ~3.3 steps/s on a V100 "
          f"(README.md:128); final loss {results[0]:.4f}")
    if args.json:
        import json
        print(json.dumps(dict(mode="native", phase="residual" if args.residual else "box-only", graph=bool(args.graph), frames_at_once=args.concurrent,
                              steps_per_s=total / dt, ms_per_step=dt / total * 1e3, seconds_per_3000_step_frame=3000 * dt / total,
                              rays_per_step=args.rays, samples_per_ray=100, views=V, instances=N, final_loss=results[0],
                              steps_per_graph=args.steps_per_graph if args.graph else None,
                              mlp_products="exact fp32 MFMA" if args.fp32_mlp else "split bf16 MFMA")))


if __name__ == "__main__":
    main()
