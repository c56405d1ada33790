#!/usr/bin/env python3
"""Phase clock of sample_table_kernel (csrc/ray_sampling.h) at a frame-sized table: ticks of the 100 MHz s_memrealtime clock at the end
of its set-up, its picks, its hash inserts and the whole draw, plus the back-to-back time per draw.  The ticks need a library built
with  VSRD_HIPCC_EXTRA="-DVSRD_TABLE_TIMERS" python -c "import __graft_entry__ as g; g.build()"  (they sit in the table header's
padding; an ordinary build prints zeros).  GPU box; experiments only."""
import sys, torch
sys.path.insert(0, '.')
import __graft_entry__; __graft_entry__.build()
from vsrd_amd import rendering
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
w = (torch.rand(17*376*1408, generator=g) - 0.62).clamp_min(0).to(dev)
pos = torch.nonzero(w > 0).flatten(); pw = w[pos].contiguous()
t = rendering.RayTable(pw)
print('count', pw.numel(), 'suits', t.suits(1000))
for i in range(5):
    idx = t.sample(1000, seed=1, stream_offset=i, remap=pos)
torch.cuda.synchronize()
pad = t.table[32:64].view(torch.int64).cpu().tolist()
print('ticks (100 MHz): init %d, picks %d, insert %d, end %d' % tuple(pad))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for i in range(200): t.sample(1000, seed=1, stream_offset=i, remap=pos)
e.record(); torch.cuda.synchronize()
print('us per draw (back to back):', s.elapsed_time(e) * 1000 / 200)
