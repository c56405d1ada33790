#!/usr/bin/env python3
"""Bitwise determinism of csrc/hypernetwork.h: twenty forward / backward steps from the same state, six times over -- every generated
weight block and every parameter must be identical run to run.  GPU box; experiments only."""
import sys, torch
sys.path.insert(0, '.')
import __graft_entry__; __graft_entry__.build()
from vsrd_amd import _lib, models, optimization
lib = _lib.load()
dev = torch.device('cuda:0')
N = 8
def make():
    torch.manual_seed(0)
    net = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).to(dev)
    emb = torch.nn.Parameter(torch.randn(1, N, 256, device=dev))
    lr = lambda v: torch.tensor(v, dtype=torch.float32, device=dev)
    opt = torch.optim.Adam([dict(params=[emb], lr=lr(1e-3)), dict(params=list(net.parameters()), lr=lr(1e-4))], lr=lr(1e-3), capturable=True)
    return net, emb, opt
torch.manual_seed(1)
grads = [torch.randn(N, _lib.MLP_WEIGHTS, device=dev) * 0.01 for _ in range(20)]
def run():
    net, emb, opt = make()
    tensors = optimization.hypernetwork_tensors(net, emb, opt, 0.99)
    ws = torch.empty(lib.vsrd_hypernetwork_workspace_bytes(N), dtype=torch.uint8, device=dev)
    out, centred = torch.zeros(N, _lib.MLP_WEIGHTS, device=dev), torch.zeros(N, _lib.MLP_WEIGHTS, device=dev)
    outs = []
    for g in grads:
        _lib.check(lib.vsrd_hypernetwork_forward(tensors, ws.data_ptr(), ws.numel(), _lib.ptr(out), _lib.ptr(centred), _lib.stream()))
        outs.append(out.clone())
        _lib.check(lib.vsrd_hypernetwork_backward_step(tensors, ws.data_ptr(), ws.numel(), _lib.ptr(g), 1.0, _lib.stream()))
    torch.cuda.synchronize()
    return outs, [p.detach().clone() for p in [emb, *net.parameters()]]
a, pa = run()
for trial in range(5):
    b, pb = run()
    first = next((i for i, (x, y) in enumerate(zip(a, b)) if not torch.equal(x, y)), None)
    print('trial', trial, 'first differing forward:', first, 'params equal:', all(torch.equal(x, y) for x, y in zip(pa, pb)))
