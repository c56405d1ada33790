#!/usr/bin/env python3
"""Clock (100 MHz) of one worker wave of hyper_hidden_forward_kernel at its barriers, from a -DVSRD_PHASE_TIMERS build.  Experiments only."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import phase_timers
phase_timers.build()
os.environ["VSRD_HIP_LIBRARY"] = phase_timers.LIB
import torch
from vsrd_amd import _lib, models, optimization
lib = _lib.load()
fn = lib.vsrd_debug_phase_cycles
fn.restype, fn.argtypes = ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32]
dev = torch.device("cuda:0")
N = 8
net = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).to(dev)
emb = torch.nn.Parameter(torch.randn(1, N, 256, device=dev))
lr = lambda v: torch.tensor(v, dtype=torch.float32, device=dev)
opt = torch.optim.Adam([dict(params=[emb], lr=lr(1e-3)), dict(params=list(net.parameters()), lr=lr(1e-4))], lr=lr(1e-3), capturable=True)
tensors = optimization.hypernetwork_tensors(net, emb, opt, 0.99)
ws = torch.empty(lib.vsrd_hypernetwork_workspace_bytes(N), dtype=torch.uint8, device=dev)
out, centred = torch.zeros(N, _lib.MLP_WEIGHTS, device=dev), torch.zeros(N, _lib.MLP_WEIGHTS, device=dev)
grad = torch.randn(N, _lib.MLP_WEIGHTS, device=dev) * 0.01
for _ in range(5):
    _lib.check(lib.vsrd_hypernetwork_forward(tensors, ws.data_ptr(), ws.numel(), _lib.ptr(out), _lib.ptr(centred), _lib.stream()))
    _lib.check(lib.vsrd_hypernetwork_backward_step(tensors, ws.data_ptr(), ws.numel(), _lib.ptr(grad), 1.0, _lib.stream()))
_lib.check(lib.vsrd_hypernetwork_forward(tensors, ws.data_ptr(), ws.numel(), _lib.ptr(out), _lib.ptr(centred), _lib.stream()))
torch.cuda.synchronize()
ticks = (ctypes.c_ulonglong * 16)()
fn(ctypes.cast(ticks, ctypes.c_void_p), 0)
names = ["wave 1: layer 0 barrier A passed", "wave 1: 16 dots reduced", "wave 1: out written, next rows requested", "wave 0: barrier B passed",
         "wave 0: LayerNorm + GELU done"]
for name, t in zip(names, list(ticks)[8:13]):
    print(f"{t / 100:7.2f} us  {name}")
