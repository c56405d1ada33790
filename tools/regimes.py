#!/usr/bin/env python3
"""One table that brackets the headline: runs on the GPU box and writes gpurun_out/<tag>/regimes.json
(copied to profiles/<round>/regimes.json).

  dense renderer (bench.py, one JSON line each):
    BASELINE config 2 at the start / mid / end of the schedule, culling on and off (worst case: every instance at every sample),
    BASELINE config 3 (residual MLP + eikonal) at full size, BASELINE config 5 (stress sizes) on one GPU,
    the two-launch (API-faithful) path of config 2;
  whole per-frame loop in the reference's native mode (tools/native_mode_bench.py): eager / hipGraph, box-only / residual phase, and one
  whole 3000-step frame with the real schedules.

  python tools/regimes.py [--tag r02] [--quick]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, timeout):
    t0 = time.time()
    out = subprocess.run([sys.executable, *cmd], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    record = json.loads(lines[-1]) if lines else {"error": (out.stderr or out.stdout)[-600:]}
    record["command"] = "python " + " ".join(cmd)
    record["wall_s"] = round(time.time() - t0, 1)
    return record


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--tag", default="r06")
    parser.add_argument("--quick", action="store_true", help="skip the two multi-second-per-step full-size runs (C3, C5)")
    args = parser.parse_args()
    out_dir = os.path.join(ROOT, "gpurun_out", args.tag)
    os.makedirs(out_dir, exist_ok=True)
    base = ["bench.py", "--no-cpu-baseline", "--no-extra-regimes"]
    dense = [("C2 mid (headline)", ["--steps", "10", "--warmup", "2"]),
             ("C2 start", ["--steps", "10", "--warmup", "2", "--schedule", "start"]),
             ("C2 end", ["--steps", "10", "--warmup", "2", "--schedule", "end"]),
             ("C2 mid, no culling", ["--steps", "5", "--warmup", "1", "--no-culling"]),
             ("C2 start, no culling", ["--steps", "5", "--warmup", "1", "--schedule", "start", "--no-culling"]),
             ("C2 end, no culling", ["--steps", "5", "--warmup", "1", "--schedule", "end", "--no-culling"]),
             ("C2 mid, exact misses not skipped", ["--steps", "5", "--warmup", "1", "--no-skip-misses"]),
             ("C2 mid, two-launch (API-faithful) path", ["--steps", "5", "--warmup", "1", "--two-launch"]),
             ("C2 mid, one ray per wave (VSRD_STEP_WAVE_PER_RAY=1: the round-2 mapping)", ["--steps", "10", "--warmup", "2", "--wave-per-ray"]),
             ("C3-shaped frame (1 view 188x704, residual)", ["--steps", "5", "--warmup", "1", "--residual", "--views", "1", "--height", "188", "--width", "704"])]
    if not args.quick:
        dense += [("C3 full size (residual)", ["--steps", "2", "--warmup", "1", "--residual"]),
                  ("C3 full size (residual), MLP on split-bf16 MFMA", ["--steps", "2", "--warmup", "1", "--residual", "--mlp-split-bf16"]),
                  ("C3 full size, start", ["--steps", "1", "--warmup", "1", "--residual", "--schedule", "start"]),
                  ("C3 full size, end", ["--steps", "1", "--warmup", "1", "--residual", "--schedule", "end"]),
                  ("C5 on one GPU", ["--steps", "2", "--warmup", "1", "--views", "17", "--height", "752", "--width", "2816", "--instances", "64", "--samples", "128"])]
    table = {"dense": [], "native": []}
    for name, flags in dense:
        record = run(base + flags, 1800)
        record["regime"] = name
        table["dense"].append(record)
        print(f"{name}: {record.get('value', 0) / 1e6:.2f} Mrays/s, {record.get('ms_per_step', 0):.1f} ms/step {record.get('error', '')}", flush=True)
    native = [("box-only, eager", []), ("box-only, hipGraph", ["--graph"]), ("residual, eager", ["--residual"]), ("residual, hipGraph", ["--residual", "--graph"]),
              ("residual, hipGraph, exact-fp32 MLP", ["--residual", "--graph", "--fp32-mlp"]),
              ("whole frame (3000 steps, real schedules), hipGraph, exact-fp32 MLP", ["--graph", "--whole-frame", "--fp32-mlp"]),
              ("box-only, hipGraph, 2 frames at once", ["--graph", "--concurrent", "2"]), ("residual, hipGraph, 2 frames at once", ["--residual", "--graph", "--concurrent", "2"]),
              ("whole frame (3000 steps, real schedules), hipGraph", ["--graph", "--whole-frame"]),
              # round 6: B frames per launch chain (optimization.FrameBatch); seconds_per_frame is per frame of the batch
              ("whole frames in a batch of 2, hipGraph", ["--graph", "--whole-frame", "--batch", "2"]),
              ("whole frames in a batch of 4, hipGraph", ["--graph", "--whole-frame", "--batch", "4"]),
              ("whole frames in a batch of 8, hipGraph", ["--graph", "--whole-frame", "--batch", "8"]),
              ("whole frames in a batch of 16, hipGraph", ["--graph", "--whole-frame", "--batch", "16"]),
              ("whole frames in a batch of 16, hipGraph, exact-fp32 MLP", ["--graph", "--whole-frame", "--batch", "16", "--fp32-mlp"]),
              ("whole frames in a batch of 32, hipGraph", ["--graph", "--whole-frame", "--batch", "32"])]
    for name, flags in native:
        record = run(["tools/native_mode_bench.py", "--json", "--steps", "300", *flags], 1800)
        record["regime"] = name
        table["native"].append(record)
        print(f"native {name}: {record.get('steps_per_s', 0):.0f} steps/s {record.get('seconds_per_frame', '')} {record.get('error', '')}", flush=True)
    with open(os.path.join(out_dir, "regimes.json"), "w") as f:
        json.dump(table, f, indent=1)


if __name__ == "__main__":
    main()
