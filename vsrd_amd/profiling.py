"""Per-launch HIP-event timing of the library calls (used by bench.py for the roofline object).

The library launches on torch's current stream, so ``torch.cuda.Event`` pairs recorded on that stream
bracket exactly one C-ABI call (= the kernel(s) it launches).  No synchronisation happens while
recording; durations are read after the caller synchronises.
"""
import contextlib
from collections import defaultdict

import torch

_active = None


class KernelTimer:
    def __init__(self):
        self.events = defaultdict(list)

    def record(self, name):
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.events[name].append((start, stop))
        return start, stop

    def summary(self):
        """name -> (launches, mean milliseconds); call after torch.cuda.synchronize()."""
        return {name: (len(pairs), sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)) for name, pairs in self.events.items()}


@contextlib.contextmanager
def kernel_timer():
    global _active
    timer = KernelTimer()
    previous, _active = _active, timer
    try:
        yield timer
    finally:
        _active = previous


@contextlib.contextmanager
def timed(name):
    """Bracket one library call; free when no timer is active."""
    if _active is None:
        yield
        return
    start, stop = _active.record(name)
    start.record()
    try:
        yield
    finally:
        stop.record()
