#!/usr/bin/env python3
"""Where frame_prologue_kernel (csrc/frame_step.h, one workgroup) spends its time: the 100 MHz clock at each of its barriers, from a
-DVSRD_PHASE_TIMERS build, for the native-mode frame of tools/native_mode_bench.py (17 views, N = 8).  GPU box; experiments only."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
MARKS = ["decode + corners", "projection of every (view, box)", "DIoU cost matrix", "Hungarian matching (wave 0)", "projection losses + their 2-D gradients",
         "adjoint of the projection", "sum over the views", "corners -> raw parameters (end)"]


def main():
    import phase_timers
    phase_timers.build()
    os.environ["VSRD_HIP_LIBRARY"] = phase_timers.LIB
    import torch
    import bench
    from vsrd_amd import _lib, optimization, rendering, fields, models, operations
    lib = _lib.load()
    fn = lib.vsrd_debug_phase_cycles
    fn.restype, fn.argtypes = ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32]
    dev = torch.device("cuda:0")
    V, H, W, N = 17, 376, 1408, 8
    K, E, raw_loc, raw_dim, raw_ori = bench.synthetic_frame(0, V, H, W, N)
    det = models.BoxParameters3D(1, N).to(dev)
    with torch.no_grad():
        det.locations.copy_(raw_loc); det.dimensions.copy_(raw_dim); det.orientations.copy_(raw_ori)
        out = det()
        cam, dirs = rendering.ray_casting((H, W), K.to(dev), E.to(dev))
        block = fields.FieldBlock(fields.pack_instances(out["locations"][0], out["orientations"][0], out["dimensions"][0]), 0.1, None, None)
        origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()
        soft = rendering.render_hierarchical(block, origins, dirs.reshape(-1, 3), (0.0, 100.0), 64, 0.1, 1.0, seed=1,
                                             skip_exact_misses=True)["labels"].clamp(0, 1).reshape(V, H, W, N).contiguous()
        gt_boxes, _ = operations.project_boxes_multi_view(out["boxes_3d"][0], E.to(dev), K.to(dev), (H, W))
    inputs = optimization.FrameInputs((H, W), K.to(dev), E.to(dev), soft, gt_boxes, torch.ones(V, N, dtype=torch.bool, device=dev))
    loop = optimization.FrameOptimizer(inputs, optimization.OptimizationConfig(seed=0), dev, graph=True)
    for _ in range(30):
        loop.step()
    torch.cuda.synchronize()
    ticks = (ctypes.c_ulonglong * 16)()
    fn(ctypes.cast(ticks, ctypes.c_void_p), 0)
    last = 0
    for name, t in zip(MARKS, list(ticks)[8:]):
        print(f"{t / 100:7.2f} us  (+{(t - last) / 100:5.2f})  {name}")
        last = t


if __name__ == "__main__":
    main()
