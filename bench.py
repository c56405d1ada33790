#!/usr/bin/env python3
"""Headline benchmark: rendered rays/s (forward + backward) on BASELINE.json config 2
(KITTI-360 376x1408, 16 instances, 64 samples/ray, 8 source views), synthetic data.

One step = one pass of the hot path over one dense frame: fused two-pass render of all V*H*W rays
(``vsrd_render_hierarchical_forward``), silhouette BCE against synthetic soft masks, backward
(``vsrd_render_backward``) through the box decode to the raw box parameters, Adam update.
Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--schedule start|mid|end]

N > 1: launched by ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...``; every rank
renders its own frame (target frames are independent optimisation problems -- no data-path collective),
barrier + synchronize on both sides of the timed region, max over ranks.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VALU_PEAK_TF = 157.3   # MI355X_MICROARCH.md: peak FP32 vector

SCHEDULES = {"start": 0.0, "mid": 0.5, "end": 1.0}   # fraction of the 3000 optimisation steps


def schedule_values(fraction):
    """scripts/main.py:420-431: cosine annealing 1.0 -> 0.1 of T and sigma; cosine_ratio = step/num_steps."""
    value = (math.cos(math.pi * fraction) + 1.0) / 2.0 * (1.0 - 0.1) + 0.1
    return dict(temperature=value, std=value, cosine_ratio=fraction)


def kitti_intrinsics(height, width):
    sx, sy = width / 1408.0, height / 376.0
    return torch.tensor([[552.554261 * sx, 0.0, 682.049453 * sx], [0.0, 552.554261 * sy, 238.769549 * sy], [0.0, 0.0, 1.0]])


def synthetic_frame(seed, num_views, height, width, num_instances):
    """KITTI-360 intrinsics, target E = I, sources shifted along z with a small yaw; raw box
    parameters ~ N(0, 0.5^2) with depth forced into 8-60 m."""
    g = torch.Generator().manual_seed(seed)
    K = kitti_intrinsics(height, width).expand(num_views, 3, 3).contiguous()
    E = torch.eye(4).repeat(num_views, 1, 1)
    half = (num_views - 1) // 2
    offsets = [0] + [k for i in range(1, half + 1) for k in (i, -i)]
    for v, k in enumerate(offsets[:num_views]):
        yaw = math.radians(0.5 * k)
        E[v, :3, :3] = torch.tensor([[math.cos(yaw), 0.0, math.sin(yaw)], [0.0, 1.0, 0.0], [-math.sin(yaw), 0.0, math.cos(yaw)]])
        E[v, 2, 3] = 1.0 * k
    raw_loc = torch.randn(1, num_instances, 3, generator=g) * 0.5
    depth = torch.empty(num_instances).uniform_(8.0, 60.0, generator=g) / 100.0
    raw_loc[0, :, 2] = torch.log(depth / (1.0 - depth))            # sigmoid^-1, decoded z = 100 * sigmoid(raw)
    raw_dim = torch.randn(1, num_instances, 3, generator=g) * 0.5
    raw_ori = torch.nn.functional.normalize(torch.randn(1, num_instances, 2, generator=g), dim=-1)
    return K, E, raw_loc, raw_dim, raw_ori


def build_union(detector, temperature):
    """The soft-min union of the current boxes as the flat parameter block (what fields.flatten() produces
    from the sdfs.translation(sdfs.rotation(instance_field(sdfs.box(...)))) tree, built here in one cat)."""
    from vsrd_amd import fields
    out = detector()
    return fields.FieldBlock(fields.pack_instances(out["locations"][0], out["orientations"][0], out["dimensions"][0]),
                             float(temperature), None, None)


def cpu_baseline(args, sched, frame, cores):
    """The oracle (CPU PyTorch restatement, kind 'port') on a bounded sample of the same workload:
    ``rows`` image rows of W rays each, issued row by row (the reference's dense idiom, main.py:1011-1023)."""
    from oracle import fields as ofields, rendering as orendering, geometry as ogeometry, losses as olosses
    torch.set_num_threads(cores)
    K, E, raw_loc, raw_dim, raw_ori = frame
    H, W, N, S = args.height, args.width, args.instances, args.samples
    cam, dirs = ogeometry.ray_casting((H, W), K[:1], E[:1])
    g = torch.Generator().manual_seed(1)
    rows = torch.linspace(H * 0.45, H * 0.8, args.cpu_rows).long()
    raws = [t[0].clone().requires_grad_(True) for t in (raw_loc, raw_dim, raw_ori)]

    def one_pass():
        total = 0.0
        for r in rows:
            loc, dim, rot, _ = ogeometry.decode_box_parameters(*raws)
            union = ofields.InstanceUnion(loc, rot, dim, sched["temperature"])
            d = dirs[0, r]
            out = orendering.hierarchical_render(union, cam[0], d, (0.0, 100.0), S, sched["std"], sched["cosine_ratio"],
                                                 torch.rand(W, S, generator=g), torch.rand(W, S, generator=g))
            loss = olosses.silhouette_loss(out.labels, torch.rand(W, N, generator=g))
            loss.backward()
            total += float(loss.detach())
        return total

    one_pass()  # warm-up
    best = float("inf")
    for _ in range(2):
        t0 = time.perf_counter()
        one_pass()
        best = min(best, time.perf_counter() - t0)
    rays = args.cpu_rows * W
    return dict(value=rays / best, unit="rays/s", cores=cores, kind="port",
                sample=f"{args.cpu_rows} image rows x {W} rays (fwd+bwd, N={N}, S={S}), oracle/ on host CPU, best of 2")


def measured_traffic(kernel_symbol):
    """HBM bytes per launch of `kernel_symbol` from the newest committed rocprofv3 PMC summary (profiles/rNN/traffic.json,
    written by tools/summarize_profile.py from separate FETCH_SIZE / WRITE_SIZE passes of this same command), or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            kernels = json.load(open(path))["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        for name, entry in kernels.items():
            if kernel_symbol in name and "FETCH_SIZE_bytes" in entry and "WRITE_SIZE_bytes" in entry:
                return entry["FETCH_SIZE_bytes"] + entry["WRITE_SIZE_bytes"], os.path.relpath(path, ROOT)
    return None, None


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=10)
    parser.add_argument("--warmup", type=int, default=2)
    parser.add_argument("--views", type=int, default=9)         # 1 target + 8 source views
    parser.add_argument("--height", type=int, default=376)
    parser.add_argument("--width", type=int, default=1408)
    parser.add_argument("--instances", type=int, default=16)
    parser.add_argument("--samples", type=int, default=64)
    parser.add_argument("--schedule", choices=sorted(SCHEDULES), default="mid")
    parser.add_argument("--cpu-rows", type=int, default=24)
    parser.add_argument("--cpu-threads", type=int, default=16)   # best of {8,16,32,64} on the 2x64-core EPYC 9575F GPU host
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-skip-misses", action="store_true")
    parser.add_argument("--residual", action="store_true", help="BASELINE config 3: per-instance residual MLP + eikonal loss")
    parser.add_argument("--two-launch", action="store_true",
                        help="render forward, torch BCE, render backward as separate launches instead of the fused step kernel")
    args = parser.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)     # RCCL; used for barrier / max-reduce only

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if distributed:
        dist.barrier()
    from vsrd_amd import models, rendering, profiling

    sched = schedule_values(SCHEDULES[args.schedule])
    V, H, W, N, S = args.views, args.height, args.width, args.instances, args.samples
    frame = synthetic_frame(seed=rank, num_views=V, height=H, width=W, num_instances=N)   # one frame per rank
    K, E, raw_loc, raw_dim, raw_ori = frame

    # ---- resident inputs (untimed) -----------------------------------------------------------------
    cam, dirs = rendering.ray_casting((H, W), K.to(dev), E.to(dev))                 # [V,3], [V,H,W,3]
    directions = dirs.reshape(-1, 3).contiguous()
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()   # per-pixel, as main.py:289-296
    R = directions.shape[0]
    detector = models.BoxParameters3D(1, N).to(dev)
    with torch.no_grad():
        detector.locations.copy_(raw_loc); detector.dimensions.copy_(raw_dim); detector.orientations.copy_(raw_ori)
        # targets: soft silhouettes of a perturbed copy of the boxes at the final (sharp) schedule
        perturbed = models.BoxParameters3D(1, N).to(dev)
        g = torch.Generator().manual_seed(1000 + rank)
        perturbed.locations.copy_(raw_loc + torch.randn(raw_loc.shape, generator=g) * 0.05)
        perturbed.dimensions.copy_(raw_dim + torch.randn(raw_dim.shape, generator=g) * 0.2)
        perturbed.orientations.copy_(raw_ori + torch.randn(raw_ori.shape, generator=g) * 0.1)
        targets = rendering.render_hierarchical(build_union(perturbed, 0.1), origins, directions, (0.0, 100.0), S, 0.1, 1.0,
                                                seed=99, skip_exact_misses=True)["labels"].clamp(0.0, 1.0).contiguous()
    params = [detector.locations, detector.dimensions, detector.orientations]
    hyper = None
    if args.residual:
        torch.manual_seed(rank)
        hyper = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).to(dev)
        params += [detector.embeddings, *hyper.parameters()]
    optimizer = torch.optim.Adam(params, lr=1e-2)
    skip = not args.no_skip_misses and not args.residual     # eikonal needs every ray's gradients
    fused = not args.two_launch                               # one launch: render + silhouette BCE (+ eikonal) + adjoint

    def step(index):
        optimizer.zero_grad(set_to_none=True)
        union = build_union(detector, sched["temperature"])
        if hyper is not None:
            union.mlp_weights = hyper(detector.embeddings)[0].contiguous()
        if fused:
            loss = rendering.silhouette_step(union, origins, directions, targets, (0.0, 100.0), S, sched["std"], sched["cosine_ratio"],
                                             seed=rank, stream_offset=index, skip_exact_misses=skip, eikonal_ratio=0.01 if hyper is not None else 0.0)
            loss.backward()
            optimizer.step()
            return loss
        out = rendering.render_hierarchical(union, origins, directions, (0.0, 100.0), S, sched["std"], sched["cosine_ratio"],
                                            seed=rank, stream_offset=index, skip_exact_misses=skip, return_gradients=hyper is not None)
        loss = torch.nn.functional.binary_cross_entropy(out["labels"].clamp(1.0e-6, 1.0 - 1.0e-6), targets, reduction="none").mean()
        if hyper is not None:
            loss = loss + 0.01 * ((out["gradients"].norm(dim=-1) - 1.0) ** 2).mean()
        loss.backward()
        optimizer.step()
        return loss

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    with profiling.kernel_timer() as timer:
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = step(args.warmup + i)
        fence()
        elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernels = timer.summary()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * R * args.steps / elapsed
        # Algorithmic bytes per ray (B_api = 24 + 12 N for fwd+bwd with per-ray origins excluded):
        #   forward launch : direction 12 + labels out 4N ; backward launch: direction re-read 12 + grad_labels in 4N
        #   fused step launch: direction 12 + targets 4N (labels, label adjoints and saved distances never leave the chip)
        if fused:
            entry = "vsrd_render_residual_step" if args.residual else "vsrd_render_silhouette_step"
            fwd_n, fwd_ms = kernels[entry]
            bwd_n, bwd_ms = 0, 0.0
            dominant, dom_ms, dom_bytes, symbol = entry, fwd_ms, 12 + 4 * N, "render_residual_step_kernel" if args.residual else "render_silhouette_kernel"
        else:
            fwd_n, fwd_ms = kernels["vsrd_render_hierarchical_forward"]
            bwd_n, bwd_ms = kernels["vsrd_render_backward"]
            dominant, dom_ms, dom_bytes, symbol = ("vsrd_render_backward", bwd_ms, 12 + 4 * N, "render_backward_kernel") if bwd_ms >= fwd_ms else \
                                                  ("vsrd_render_hierarchical_forward", fwd_ms, 12 + 4 * N, "render_hierarchical_kernel")
        traffic, traffic_source = measured_traffic(symbol)
        achieved_gbs = R * dom_bytes / (dom_ms * 1e-3) / 1e9
        flop_per_ray = 3.5 * (3 * S - 2) * (63 * N + 45)           # SURVEY.md §8d box-only model, fwd+bwd
        valu_tf = R * flop_per_ray / ((fwd_ms + bwd_ms) * 1e-3) / 1e12
        with torch.no_grad():
            miss = float((targets.sum(-1) == 0).float().mean())
        result = {
            "metric": "rendered rays/sec (fwd+bwd), KITTI-360 376x1408, 16 instances",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config 2: dense frame, {V} views x {H}x{W} = {R} rays/step/GPU, {N} box instances, "
                                   f"{S} samples/ray (pass 1: {S - 1}, pass 2: {2 * S - 1} points), " + ("box + residual-MLP field, eikonal loss" if args.residual else "box-only field"),
                       "schedule": f"{args.schedule}: T=std={sched['std']:.3f}, cosine_ratio={sched['cosine_ratio']:.2f}",
                       "skip_exact_misses": skip, "rng": "in-kernel Philox4x32-10", "rays_per_gpu": R,
                       "loss": ("silhouette BCE" + (" + 0.01 eikonal" if args.residual else "") + " fused into the render kernel" if fused
                                else "silhouette BCE" + (" + 0.01 eikonal" if args.residual else "") + " (torch elementwise)") + " + Adam",
                       "launches_per_step": "1 fused (render + loss + adjoint) + partial reductions" if fused else "forward, torch loss, backward",
                       "final_loss": float(loss.detach()), "target_empty_fraction": miss},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_ray": dom_bytes, "launch_ms": dom_ms,
                         "note": "the fused path is fp32-VALU/transcendental bound, not HBM bound (SURVEY.md §8d); see roofline_valu"},
            "roofline_valu": {"bound": "fp32-valu", "achieved": valu_tf, "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s",
                              "frac": valu_tf / FP32_VALU_PEAK_TF, "model_flop_per_ray": flop_per_ray,
                              "forward_ms": fwd_ms, "backward_ms": bwd_ms, "launches": [fwd_n, bwd_n]},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, sched, frame, min(args.cpu_threads, os.cpu_count() or 1))
        print(json.dumps(result))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
