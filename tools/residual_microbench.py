#!/usr/bin/env python3
"""Cost of one residual-MLP evaluation (64 points x 1 instance, value + gradient) through vsrd_field_eval, which evaluates
every instance at every point (no culling):  python tools/residual_microbench.py [--points 1048576] [--instances 16]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--points", type=int, default=1 << 20)
    parser.add_argument("--instances", type=int, default=16)
    args = parser.parse_args()
    import __graft_entry__
    __graft_entry__.build()
    from vsrd_amd import fields, rendering
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    N, P = args.instances, args.points
    loc = torch.rand(N, 3, generator=gen) * 20 - 10
    dim = torch.rand(N, 3, generator=gen) + 0.5
    rot = torch.eye(3).repeat(N, 1, 1)
    mlp = torch.randn(N, 1617, generator=gen) * 0.3
    pts = (torch.rand(P, 3, generator=gen) * 30 - 15).to(dev)
    for residual in (False, True):
        block = fields.FieldBlock(fields.pack_instances(loc, rot, dim).to(dev), 0.5, mlp.to(dev) if residual else None, None)
        for _ in range(2):
            rendering.evaluate_field(block, pts, with_gradients=True, with_labels=False)
        torch.cuda.synchronize()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        reps = 5
        for _ in range(reps):
            rendering.evaluate_field(block, pts, with_gradients=True, with_labels=False)
        stop.record()
        torch.cuda.synchronize()
        ms = start.elapsed_time(stop) / reps
        evals = P / 64 * N
        cycles = ms * 1e-3 * 2.4e9 * 1024 / evals          # SIMD-cycles per (64 points x 1 instance)
        print(f"residual={residual}: {ms:.3f} ms per call, {P * N / ms / 1e6:.2f} G point-instances/s, {cycles:.0f} SIMD-cycles per 64-point evaluation")


if __name__ == "__main__":
    main()
