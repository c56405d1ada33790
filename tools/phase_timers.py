#!/usr/bin/env python3
"""Where a fused step kernel spends its time, phase by phase (GPU box; experiments only).

Builds (or reuses) an instrumented copy of the library (-DVSRD_PHASE_TIMERS: every wave accumulates s_memtime ticks per phase into
a global table), runs a few fused steps of a bench.py-shaped workload through it and prints the share of each phase.

  python tools/phase_timers.py [--residual] [--views 1 --height 188 --width 704] [--instances 16] [--samples 64] [--schedule mid]
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "vsrd_amd", "lib", "libvsrd_hip_timers.so")
PHASES = ["ray setup + sample staging", "pass 1", "importance sampling + merge", "pass 2 sweep (+ labels)", "loss, label mix, reverse sweep",
          "per-instance box adjoint (+ seeds)", "MLP adjoint (batch)", "between rays / other"]


def build():
    import __graft_entry__ as g
    sources = [os.path.join(g.CSRC, f) for f in os.listdir(g.CSRC)]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in sources):
        g.compile_library(LIB, ["-DVSRD_PHASE_TIMERS"])          # (both translation units, as __graft_entry__.build())


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--residual", action="store_true")
    parser.add_argument("--views", type=int, default=9)
    parser.add_argument("--height", type=int, default=376)
    parser.add_argument("--width", type=int, default=1408)
    parser.add_argument("--instances", type=int, default=16)
    parser.add_argument("--samples", type=int, default=64)
    parser.add_argument("--schedule", default="mid")
    parser.add_argument("--steps", type=int, default=3)
    parser.add_argument("--build-only", action="store_true")
    args = parser.parse_args()
    build()
    if args.build_only:
        return
    os.environ["VSRD_HIP_LIBRARY"] = LIB
    import torch
    import bench
    from vsrd_amd import _lib, models, rendering
    lib = _lib.load()
    fn = lib.vsrd_debug_phase_cycles
    fn.restype, fn.argtypes = ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32]
    dev = torch.device("cuda:0")
    V, H, W, N, S = args.views, args.height, args.width, args.instances, args.samples
    sched = bench.schedule_values(bench.SCHEDULES[args.schedule])
    K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(0, V, H, W, N)
    cam, dirs = rendering.ray_casting((H, W), K.to(dev), E.to(dev))
    directions = dirs.reshape(-1, 3).contiguous()
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
    detector = models.BoxParameters3D(1, N).to(dev)
    with torch.no_grad():
        detector.locations.copy_(raw_loc); detector.dimensions.copy_(raw_dim); detector.orientations.copy_(raw_ori)
        targets = rendering.render_hierarchical(bench.build_union(detector, 0.1), origins, directions, (0.0, 100.0), S, 0.1, 1.0, seed=99,
                                                skip_exact_misses=True)["labels"].clamp(0.0, 1.0).contiguous()
    hyper = None
    if args.residual:
        torch.manual_seed(0)
        hyper = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).to(dev)

    def step(index):
        union = bench.build_union(detector, sched["temperature"])
        if hyper is not None:
            union.mlp_weights = hyper(detector.embeddings)[0].contiguous()
        loss = rendering.silhouette_step(union, origins, directions, targets, (0.0, 100.0), S, sched["std"], sched["cosine_ratio"], seed=0, stream_offset=index,
                                         skip_exact_misses=hyper is None, eikonal_ratio=0.01 if hyper is not None else 0.0)
        loss.backward()

    step(0)
    torch.cuda.synchronize()
    fn(None, 1)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for i in range(args.steps):
        step(1 + i)
    stop.record()
    torch.cuda.synchronize()
    ticks = (ctypes.c_ulonglong * 16)()
    _lib.check(fn(ticks, 0))
    total = float(sum(ticks[:8])) or 1.0
    ms = start.elapsed_time(stop) / args.steps
    report = {"workload": f"V{V} {H}x{W} N{N} S{S} {args.schedule} {'residual' if args.residual else 'box'}", "ms_per_step": ms,
              "phases": {name: ticks[i] / total for i, name in enumerate(PHASES)}}
    print(f"{report['workload']}: {ms:.2f} ms/step")
    for i, name in enumerate(PHASES):
        print(f"  {ticks[i] / total * 100:6.2f} %  ~{ticks[i] / total * ms:7.2f} ms  {name}")
    if ticks[8]:
        report["instance_loops"] = {"rounds": ticks[8], "past_the_bound_test_per_round": ticks[9] / ticks[8], "survivors_per_round": ticks[10] / ticks[8]}
        print(f"  instance loops (multi-ray kernels): {ticks[8]} rounds, {ticks[9] / ticks[8]:.2f} instances past the bound test and "
              f"{ticks[10] / ticks[8]:.2f} past the exact test per round")
    if ticks[11]:
        report["pass2_rounds_behind_opaque_surface"] = ticks[12] / ticks[11]
        print(f"  pass-2 rounds that start with a transmittance below 1e-9 / 1e-6 on every ray of the wave: {ticks[12] / ticks[11] * 100:.2f} % / "
              f"{ticks[13] / ticks[11] * 100:.2f} % of {ticks[11]}")
    print(json.dumps(report))


if __name__ == "__main__":
    main()
