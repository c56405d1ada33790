#!/usr/bin/env python3
"""Is the dense residual step (config 3: chunks of rays, front kernel + MLP adjoint by work items) bit-repeatable?  Runs
rendering.silhouette_step twice on identical inputs and compares the loss and every gradient bit for bit (GPU box; debugging aid).
    python tools/dense_residual_determinism.py [--views 2] [--split]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--views", type=int, default=2)
    parser.add_argument("--split", action="store_true")
    parser.add_argument("--repeats", type=int, default=3)
    args = parser.parse_args()
    import torch
    import bench
    from vsrd_amd import fields, models, rendering
    dev = torch.device("cuda:0")
    V, H, W, N, S = args.views, 376, 1408, 16, 64
    sched = bench.schedule_values(bench.SCHEDULES["mid"])
    K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(0, V, H, W, N)
    cam, dirs = rendering.ray_casting((H, W), K.to(dev), E.to(dev))
    directions = dirs.reshape(-1, 3).contiguous()
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
    detector = models.BoxParameters3D(1, N).to(dev)
    with torch.no_grad():
        detector.locations.copy_(raw_loc); detector.dimensions.copy_(raw_dim); detector.orientations.copy_(raw_ori)
    targets = (torch.rand(origins.shape[0], N, generator=torch.Generator().manual_seed(5)) > 0.7).float().to(dev)
    mlp0 = (torch.randn(N, 1617, generator=torch.Generator().manual_seed(9)) * 0.3).to(dev)

    def run():
        out = detector()
        inst = fields.pack_instances(out["locations"][0], out["orientations"][0], out["dimensions"][0]).detach().requires_grad_(True)
        mlp = mlp0.clone().requires_grad_(True)
        block = fields.FieldBlock(inst, float(sched["temperature"]), mlp, None, yaw_gradients=True)
        value = rendering.silhouette_step(block, origins, directions, targets, (0.0, 100.0), S, sched["std"], sched["cosine_ratio"],
                                          seed=0, stream_offset=0, skip_exact_misses=False, eikonal_ratio=0.01, mlp_split_bf16=args.split)
        grads = torch.autograd.grad(value, (inst, mlp))
        torch.cuda.synchronize()
        return value.detach().clone(), grads[0].clone(), grads[1].clone()

    first = run()
    for r in range(args.repeats):
        again = run()
        same = [torch.equal(a, b) for a, b in zip(first, again)]
        worst = [float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)) for a, b in zip(first, again)]
        print(f"repeat {r}: loss/instances/mlp identical {same}; relative differences {worst}")


if __name__ == "__main__":
    main()
