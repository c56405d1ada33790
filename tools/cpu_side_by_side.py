#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs the read-only reference checkout at /root/reference; nothing here travels to the GPU box).

Times the REFERENCE's own hot path and the CPU oracle (oracle/, bench.py's `cpu_baseline` of kind "port") side by side on the same
rays, the same boxes and the same host cores, so that the oracle's rays/s on the GPU host can be read as "the reference would be
about <ratio> x that" (VERDICT r01 item 3).

  reference leg : vsrd.rendering.hierarchical_volumetric_rendering (renderers.py:177-270) driven by the two-pass wrapper of
                  scripts/main.py:511-523 over the closure tree of main.py:433-509 built around the reference's sdfs.* -- the N-way
                  Python closure loop, the [N,S',R,N] one-hot products and the double backward through autograd.grad(create_graph)
                  are all executed by the reference's code (the closures are the restatement tests/golden/make_golden.py uses,
                  because main.py's are nested in train() and not importable);
  oracle leg    : oracle.rendering.hierarchical_render over oracle.fields.InstanceUnion (closed-form normals, ray-major).

Both legs: rows of W rays issued one call per row (main.py:1011-1023; a quarter row per call for the residual case), silhouette BCE (+ 0.01 eikonal with --residual),
backward to the box parameters (and MLP weights); forward and backward timed separately; 1 warm-up + 3 repeats, fastest repeat.

  python tools/cpu_side_by_side.py [--out profiles/r02/cpu_side_by_side.json]
"""
import argparse
import importlib.util
import json
import os
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

import bench
from oracle import fields as ofields, rendering as orendering, geometry as ogeometry, losses as olosses


def load_generator():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def timed_legs(case, ref, gen, threads):
    N, S, H, W, num_rays, residual = case["N"], case["S"], case["H"], case["W"], case["rays"], case["residual"]
    torch.set_num_threads(threads)
    sched = bench.schedule_values(0.5)
    K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(0, 1, H, W, N)
    cam, dirs = ogeometry.ray_casting((H, W), K[:1], E[:1])
    flat = dirs[0].reshape(-1, 3)
    first = max(0, min(int(H * 0.55) * W, H * W - num_rays))
    chunk = case.get("chunk", W)
    chunks = [(a, min(a + chunk, first + num_rays)) for a in range(first, first + num_rays, chunk)]
    g = torch.Generator().manual_seed(1)
    targets = [torch.rand(b - a, N, generator=g) for a, b in chunks]
    uniforms = [(torch.rand(b - a, S, generator=g), torch.rand(b - a, S, generator=g)) for a, b in chunks]
    mlp0 = torch.randn(N, 1617, generator=g) * 0.25 if residual else None

    def leaves():
        loc, dim, rot, _ = ogeometry.decode_box_parameters(raw_loc[0], raw_dim[0], raw_ori[0])
        out = [t.detach().clone().requires_grad_(True) for t in (loc, dim, rot)]
        if residual:
            out.append(mlp0.clone().requires_grad_(True))
        return out

    hyper = ref.hyper.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]) if residual else None
    encoder = ref.encoder.SinusoidalEncoder(8) if residual else None

    def reference_pass():
        params = leaves()
        loc, dim, rot = params[:3]
        members = []
        for i in range(N):
            base = ref.sdfs.box(dim[i])
            if residual:
                base = gen.make_residual_composition(base, gen.make_residual_field(hyper, encoder, params[3][i]))
            members.append(ref.sdfs.translation(ref.sdfs.rotation(gen.make_instance_field(base, i, N), rot[i]), loc[i]))
        field = gen.make_soft_union(members, sched["temperature"])
        forward = backward = 0.0
        for (a, b), target in zip(chunks, targets):
            t0 = time.perf_counter()
            _, _, (labels, gradients, _, _) = gen.two_pass(ref.renderers.hierarchical_volumetric_rendering, distance_field=field,
                                                           ray_positions=cam[0], ray_directions=flat[a:b], distance_range=(0.0, 100.0),
                                                           num_samples=S, sdf_std_deviation=sched["std"], cosine_ratio=sched["cosine_ratio"])
            loss = torch.nn.functional.binary_cross_entropy(labels.clamp(1.0e-6, 1.0 - 1.0e-6), target, reduction="none").mean()
            if residual:
                loss = loss + 0.01 * torch.nn.functional.mse_loss(torch.norm(gradients, dim=-1), gradients.new_ones(gradients.shape[:-1]))
            t1 = time.perf_counter()
            loss.backward()
            t2 = time.perf_counter()
            forward, backward = forward + t1 - t0, backward + t2 - t1
        return forward, backward

    def oracle_pass():
        params = leaves()
        loc, dim, rot = params[:3]
        forward = backward = 0.0
        for (a, b), target, (uc, uf) in zip(chunks, targets, uniforms):
            t0 = time.perf_counter()
            union = ofields.InstanceUnion(loc, rot, dim, sched["temperature"], params[3] if residual else None)
            out = orendering.hierarchical_render(union, cam[0], flat[a:b], (0.0, 100.0), S, sched["std"], sched["cosine_ratio"], uc, uf)
            loss = olosses.silhouette_loss(out.labels, target)
            if residual:
                loss = loss + 0.01 * olosses.eikonal_loss(out.gradients)
            t1 = time.perf_counter()
            loss.backward()
            t2 = time.perf_counter()
            forward, backward = forward + t1 - t0, backward + t2 - t1
        return forward, backward

    result = dict(case, threads=threads, calls=len(chunks))
    for name, one_pass in (("reference", reference_pass), ("oracle", oracle_pass)):
        one_pass()
        forward, backward = min((one_pass() for _ in range(3)), key=sum)
        result[name] = dict(forward_s=forward, backward_s=backward, rays_per_s=num_rays / (forward + backward))
    result["oracle_over_reference"] = result["oracle"]["rays_per_s"] / result["reference"]["rays_per_s"]
    return result


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02", "cpu_side_by_side.json"))
    parser.add_argument("--threads", type=int, default=os.cpu_count())
    parser.add_argument("--only", default="", help="run only the cases whose name contains this text; the other rows of --out are kept")
    args = parser.parse_args()
    gen = load_generator()
    ref = gen.import_reference()
    cases = [
        dict(name="BASELINE config 1 sizes (N=4, S=32, 128x128)", N=4, S=32, H=128, W=128, rays=16384, residual=False),
        dict(name="BASELINE config 2 sizes (N=16, S=64, 376x1408)", N=16, S=64, H=376, W=1408, rays=16384, residual=False),
        dict(name="BASELINE config 3 sizes (N=16, S=64, residual MLP + eikonal)", N=16, S=64, H=376, W=1408, rays=1408, residual=True, chunk=352),
    ]
    rows = []
    kept = {}
    if args.only and os.path.exists(args.out):
        kept = {row["name"]: row for row in json.load(open(args.out))["cases"]}
    for case in cases:
        if args.only and args.only not in case["name"]:
            if case["name"] in kept:
                rows.append(kept[case["name"]])
            continue
        row = timed_legs(case, ref, gen, args.threads)
        rows.append(row)
        print(f"{row['name']}: reference {row['reference']['rays_per_s']:.0f} rays/s (fwd {row['reference']['forward_s']:.2f} s, bwd "
              f"{row['reference']['backward_s']:.2f} s); oracle {row['oracle']['rays_per_s']:.0f} rays/s (fwd {row['oracle']['forward_s']:.2f} s, "
              f"bwd {row['oracle']['backward_s']:.2f} s); oracle / reference = {row['oracle_over_reference']:.2f}", flush=True)
    cpu = bench.cpu_model_name()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(dict(host=f"{cpu}, {os.cpu_count()} logical cores (build container)", torch=torch.__version__, threads=args.threads,
                       protocol="rows of W rays per call, fwd and bwd separately, 1 warm-up + 3 repeats, fastest repeat; same boxes, rays and targets for both legs",
                       cases=rows), f, indent=1)
    leftovers = [p for p, _, _ in os.walk("/root/reference") if p.endswith("__pycache__")]
    assert not leftovers, leftovers


if __name__ == "__main__":
    main()
