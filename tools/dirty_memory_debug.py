"""Debug script (GPU box): does the residual step leave any output it returns unwritten?  The caching allocator's blocks are filled with a
pattern before each of two runs of the same step (0.123 / 0.777); outputs that differ between the runs were read from memory the kernels never wrote."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bench
    from test_hip_scale import scene
    from vsrd_amd import models, rendering
    dev = torch.device("cuda:0")
    N, S, V, H, W, seed = 16, 64, int(os.environ.get("VIEWS", 9)), 376, 1408, 3
    sched = bench.schedule_values(bench.SCHEDULES["mid"])
    T, std, ratio = sched["temperature"], sched["std"], sched["cosine_ratio"]
    det, cam, dirs = scene(dev, N, V, H, W, seed=seed)
    directions = dirs.reshape(-1, 3)
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
    R = directions.shape[0]
    torch.manual_seed(0)
    hyper = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).to(dev)
    with torch.no_grad():
        targets = rendering.render_hierarchical(bench.build_union(det, 0.1), origins, directions, (0.0, 100.0), S, 0.1, 1.0, seed=99,
                                                skip_exact_misses=True)["labels"].clamp(0.0, 1.0).contiguous()
        det.locations.add_(0.02)
        generator = torch.Generator(device=dev).manual_seed(77)
        u_coarse = torch.rand(R, S, device=dev, generator=generator)
        u_fine = torch.rand(R, S, device=dev, generator=generator)
    results = []
    for pattern in (0.123, 0.777, float("nan")):
        junk = [torch.full((1 << 28,), pattern, device=dev) for _ in range(24)]          # 24 GiB of pattern, back to the allocator's cache
        del junk
        union = bench.build_union(det, T)
        union.mlp_weights = hyper(det.embeddings)[0].contiguous()
        params = [det.locations, det.dimensions, det.orientations, det.embeddings]
        loss, terms, labels = rendering.silhouette_step(union, origins, directions, targets, (0.0, 100.0), S, std, ratio, u_coarse=u_coarse, u_fine=u_fine,
                                                        eikonal_ratio=0.01, return_labels=True, return_terms=True, skip_exact_misses=False,
                                                        mlp_split_bf16=os.environ.get("SPLIT", "0") == "1")
        grads = torch.autograd.grad(loss, params)
        results.append((float(loss), terms.clone(), labels.clone(), [g.clone() for g in grads]))
        del labels, loss, grads
    base = results[0]
    for k, other in enumerate(results[1:], start=1):
        moved = torch.nonzero(((base[2] != other[2]) & ~(torch.isnan(base[2]) & torch.isnan(other[2]))).any(-1)).flatten()
        print(f"run {k} vs run 0: loss {base[0]!r} vs {other[0]!r}; terms {base[1].tolist()} vs {other[1].tolist()}; rays whose labels differ: {moved.numel()} of {R}"
              + (f" (first: {moved[:8].tolist()}; labels there {base[2][moved[0]].tolist()[:6]} vs {other[2][moved[0]].tolist()[:6]})" if moved.numel() else ""))
        for name, a, b in zip(("locations", "dimensions", "orientations", "embeddings"), base[3], other[3]):
            print(f"    grad {name}: max |difference| {float((a - b).abs().max()):.3e} of {float(a.abs().max()):.3e}, nan {bool(torch.isnan(b).any())}")


if __name__ == "__main__":
    main()
