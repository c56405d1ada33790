#!/usr/bin/env python3
"""What would the residual MLP on split-bf16 matrix products cost in accuracy?  (round 5; test infrastructure:
imports oracle/.)  CPU emulation, no GPU: the oracle's per-instance linear layers (oracle/fields.py::_instance_linear, i.e.
hyper_distance_field.py:66-70) are run with BOTH operands replaced by a sum of bfloat16 parts --

    parts = 1:  x ~ bf16(x)                                    (plain bf16: 2^-9 relative per operand)
    parts = 2:  x ~ h + l,  h = bf16(x), l = bf16(x - h)       (2^-18: what two v_mfma_f32_16x16x32_bf16 per 16 x 16 layer give: the
                                                                 K = 32 slots of a lane hold [h(4 channels) | l(4 channels)], the weight
                                                                 operand is [w_h | w_h] then [w_l | w_l]: all four products)
    parts = 3:  x ~ h + m + l by truncation                    (exact: 8 + 8 + 8 significand bits; six of the nine products in three MFMAs)

-- with exact accumulation (float64), against the float32 oracle and the reference's golden outputs, on the residual goldens the GPU
tests use.  Printed: the worst label difference and the gradient differences relative to the largest entry, next to the tolerances
of tests/test_hip_render.py (labels 1e-4, gradients 5e-3).

    python tests/split_bf16_emulation.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, RESIDUAL_CASES                     # noqa: E402
from oracle import fields, rendering, losses                         # noqa: E402
from test_oracle_golden import union_from                            # noqa: E402


def split(x, parts):
    if parts == 0:
        return x
    if parts == 3:                                                   # truncation: top 16 bits of the float32 pattern, three times
        def top(v):
            return (v.contiguous().view(torch.int32) & -65536).view(torch.float32)
        h = top(x); m = top(x - h); low = top(x - h - m)
        return (h.double() + m.double() + low.double()).to(x.dtype) if x.dtype != torch.float64 else h + m + low
    total = torch.zeros_like(x, dtype=torch.float64)
    rest = x.double()
    for _ in range(parts):
        part = rest.float().to(torch.bfloat16).double()
        total = total + part
        rest = rest - part
    return total


class _Quantised(torch.autograd.Function):
    """y = q(x) in the forward, identity in the backward (the kernels differentiate the exact network; the emulation only perturbs values)."""

    @staticmethod
    def forward(ctx, x, parts):
        return split(x.detach().float(), parts).to(x.dtype)

    @staticmethod
    def backward(ctx, grad):
        return grad, None


def patched_linear(parts, original):
    def linear(block, fan_in, x, dx):
        q = lambda t: None if t is None else _Quantised.apply(t, parts)        # noqa: E731
        matrix = q(block[..., :fan_in])
        block_q = torch.cat([matrix, block[..., fan_in:]], dim=-1)
        return original(block_q, fan_in, q(x), q(dx))
    return linear


def run_case(name, parts):
    g = load_golden(name)
    S = int(g["num_samples"])
    union = union_from(g, requires_grad=True)
    std, ratio = float(g["sdf_std_deviation"]), float(g["cosine_ratio"])
    original = fields._instance_linear
    fields._instance_linear = patched_linear(parts, original) if parts else original
    try:
        fine = rendering.hierarchical_render(union, g["origins"], g["directions"], (0.0, 100.0), S, std, ratio, g["u_coarse"], g["u_fine"])
        miss = g["coarse_weights"].sum(0) == 0
        bce = losses.silhouette_loss(fine.labels, g["targets"])
        eik = losses.eikonal_loss(fine.gradients[~miss])
        loss = bce + float(g["eikonal_weight"]) * eik
        params = [union.locations, union.dimensions, union.orientations, union.mlp_weights]
        grads = torch.autograd.grad(loss, params)
    finally:
        fields._instance_linear = original
    return fine.labels.detach(), [x.detach() for x in grads], g


def main():
    names = ["grad_locations", "grad_dimensions", "grad_orientations", "grad_mlp_weights"]
    for case in RESIDUAL_CASES:
        base_labels, base_grads, g = run_case(case, 0)
        print(f"{case}:  float32 oracle vs golden: labels {float((base_labels - g['fine_labels']).abs().max()):.2e}; "
              + "  ".join(f"{n[5:]} {float((a - g[n]).abs().max() / g[n].abs().max()):.2e}" for a, n in zip(base_grads, names)))
        for parts in (1, 2, 3):
            labels, grads, _ = run_case(case, parts)
            print(f"    {parts} bf16 part(s):  vs float32 oracle: labels {float((labels - base_labels).abs().max()):.2e}; "
                  + "  ".join(f"{n[5:]} {float((a - b).abs().max() / b.abs().max()):.2e}" for a, b, n in zip(grads, base_grads, names))
                  + f"   | vs golden: labels {float((labels - g['fine_labels']).abs().max()):.2e} (tolerance 1e-4); "
                  + "  ".join(f"{n[5:]} {float((a - g[n]).abs().max() / g[n].abs().max()):.2e}" for a, n in zip(grads, names)) + " (5e-3)")


if __name__ == "__main__":
    main()
