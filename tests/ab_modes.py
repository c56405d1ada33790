"""A/B of the kernels' run-time switches (culling, soft-min shift, y-rotation fast path) on the golden scenes: labels and gradients
of every combination against the all-off baseline and against the float64 oracle.  GPU only:  python tests/ab_modes.py"""
import sys, itertools, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_hip_render import load_golden, hip_union
from vsrd_amd import rendering
from vsrd_amd.rendering import renderers
dev = torch.device("cuda:0")
for name in ["g4_render_n16_s64_mid", "g4_render_n4_s32_late", "g4_render_n4_s32_step0"]:
    g = load_golden(name)
    S = int(g["num_samples"]); std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    results = {}
    for culling, running, general in itertools.product((True, False), repeat=3):
        renderers.CULLING, renderers.RUNNING_MINIMUM, renderers.GENERAL_ROTATIONS = culling, running, general
        union, params = hip_union(g, dev, requires_grad=True)
        out = rendering.render_hierarchical(union, g["origins"].to(dev), g["directions"].to(dev), (0.0, 100.0), S, std, ratio,
                                            u_coarse=g["u_coarse"].to(dev), u_fine=g["u_fine"].to(dev), return_gradients=True)
        lam = torch.randn(out["labels"].shape, generator=torch.Generator().manual_seed(2)).to(dev)
        gam = (torch.randn(out["gradients"].shape, generator=torch.Generator().manual_seed(3)) * 0.01).to(dev)
        hit = (g["coarse_weights"].sum(0) > 0).to(dev)
        loss = (out["labels"] * lam).sum() + (out["gradients"][hit] * gam[hit]).sum()
        results[(culling, running, general)] = (out["labels"].detach(), torch.autograd.grad(loss, params))
    base = results[(False, True, True)]
    for key, (lab, grads) in results.items():
        errs = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(grads, base[1])]
        print(name, "culling=%d running=%d general=%d" % key, "labels %.2e" % float((lab - base[0]).abs().max()), "grads", " ".join("%.2e" % e for e in errs))
    # float64 oracle of the same loss
    from oracle import fields as ofields, rendering as orendering
    from test_hip_render import cpu_union
    union64, params64 = cpu_union(g, requires_grad=True, dtype=torch.float64)
    out = orendering.hierarchical_render(union64, g["origins"].double(), g["directions"].double(), (0.0, 100.0), S, std, ratio,
                                         g["u_coarse"].double(), g["u_fine"].double())
    lam = torch.randn(out.labels.shape, generator=torch.Generator().manual_seed(2)).double()
    gam = (torch.randn(out.gradients.shape, generator=torch.Generator().manual_seed(3)) * 0.01).double()
    hit = (g["coarse_weights"].sum(0) > 0)
    loss = (out.labels * lam).sum() + (out.gradients[hit] * gam[hit]).sum()
    ref = torch.autograd.grad(loss, params64)
    for key, (lab, grads) in results.items():
        errs = [float((a.double().cpu() - b).abs().max() / b.abs().max()) for a, b in zip(grads, ref)]
        print(name, "vs float64 oracle: culling=%d running=%d general=%d" % key, "labels %.2e" % float((lab.double().cpu() - out.labels).abs().max()), "grads", " ".join("%.2e" % e for e in errs))
