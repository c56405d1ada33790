"""Debug script (asserts nothing; `python tests/mode_noise_debug.py [steps]` on a GPU box): the eager loop against the graph loop of
test_hip_step.py::test_graph_mode_replays_the_same_steps over MANY residual steps, parameters and Adam moments re-synchronised before every
step -- how often does a step's raw box gradient differ by more than 1e-3 of its largest entry, and by how much?  The two loops get MLP
weights that differ in the last bit (torch hypernetwork against csrc/hypernetwork.h); what that does to a step is the renderer's own
conditioning (importance sampler, box-normal flips) and should not depend on the library build: compare builds with VSRD_HIP_LIBRARY."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_hip_step import c1_frame          # noqa: E402


def main():
    from vsrd_amd import fields, optimization, rendering
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device("cuda:0")
    V, H, W, N, S, R = 3, 128, 128, 4, 32, 256
    K, E, (loc, dim, rot), gt_boxes, visible = c1_frame()
    cam, dirs = rendering.ray_casting((H, W), K.to(dev), E.to(dev))
    block = fields.FieldBlock(fields.pack_instances(loc.to(dev), rot.to(dev), dim.to(dev)), 0.1, None, None)
    origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
    soft = rendering.render_hierarchical(block, origins, dirs.reshape(-1, 3), (0.0, 100.0), S, 0.1, 1.0, seed=5)["labels"].clamp(0, 1)
    soft = (soft.reshape(V, H, W, N) * visible.to(dev)[:, None, None, :]).contiguous()
    inputs = optimization.FrameInputs((H, W), K.to(dev), E.to(dev), soft, gt_boxes.to(dev), visible.to(dev))
    g = torch.Generator().manual_seed(4)
    start = [torch.randn(N, 3, generator=g) * 0.2, torch.randn(N, 3, generator=g) * 0.2,
             torch.nn.functional.normalize(torch.tensor([1.0, 0.0]) + torch.randn(N, 2, generator=g) * 0.2, dim=-1)]
    start[0][:, 2] -= 1.5
    for split in (True, False):
        config = optimization.OptimizationConfig(num_samples=S, num_rays=R, warmup_steps=6, num_steps=3000, mlp_split_bf16=split)
        loops = []
        for graph in (False, True):
            torch.manual_seed(0)
            loop = optimization.FrameOptimizer(inputs, config, dev, graph=graph, fused_glue=graph)
            with torch.no_grad():
                for p, v in zip((loop.detector.locations, loop.detector.dimensions, loop.detector.orientations), start):
                    p.copy_(v[None].to(dev))
            loops.append(loop)
        loops[1].hyper_distance_field.load_state_dict(loops[0].hyper_distance_field.state_dict())
        with torch.no_grad():
            loops[1].detector.embeddings.copy_(loops[0].detector.embeddings)
        weights = soft.reshape(-1, N).max(-1).values

        def tensors(loop):
            params = list(loop.detector.parameters()) + list(loop.hyper_distance_field.parameters())
            return params, [[loop.optimizer.state[p][k] for k in ("exp_avg", "exp_avg_sq")] if loop.optimizer.state.get(p) else None for p in params]

        errors = []
        torch.manual_seed(1)
        for step in range(6 + steps):
            idx = torch.multinomial((weights > 0.5).float(), R, replacement=False)
            with torch.no_grad():
                (pe, me), (pg, mg) = tensors(loops[0]), tensors(loops[1])
                for a, b in zip(pe, pg):
                    b.copy_(a)
                for a, b in zip(me, mg):
                    if a is not None and b is not None:
                        b[0].copy_(a[0]), b[1].copy_(a[1])
            eager, replayed = loops[0].step(idx), loops[1].step(idx)
            if step >= 6:
                errors.append(max(float((a - b).abs().max()) / max(float(a.abs().max()), 1e-6) for a, b in zip(eager["raw_gradients"], replayed["raw_gradients"])))
        errors_sorted = sorted(errors)
        print(f"{'split bf16' if split else 'exact fp32'}: {steps} residual steps, raw box gradients eager vs graph loop: median {errors_sorted[len(errors) // 2]:.2e}, "
              f"steps beyond 1e-4: {sum(e > 1e-4 for e in errors)}, beyond 1e-3: {sum(e > 1e-3 for e in errors)}, worst {errors_sorted[-1]:.2e}; "
              f"first ten: {' '.join(f'{e:.1e}' for e in errors[:10])}")
        for loop in loops:
            loop.close()


if __name__ == "__main__":
    main()
