"""Frame-sharded multi-GPU launcher: one process per GPU, frames split across ranks, no data-path collective.

Reference behaviour (scripts/main.py:45-57, vsrd/distributed/loader.py:4-9, README.md:128): ``torch.distributed`` is
initialised, ranks print in order between barriers, a ``DistributedSampler`` hands every rank its share of the target
frames, and each rank optimises its frames alone -- gradients are never averaged.  The only traffic is the start-up
barrier; here additionally rank 0 broadcasts the run manifest so every rank agrees on the frame list.  On ROCm the
``nccl`` backend is RCCL (xGMI carries latency-only messages; nothing bandwidth-bound exists on this path).
"""
import os
import random
from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def init_process_group(backend=None):
    """env:// rendezvous as with torchrun (main.py:49).  Returns (rank, world_size, device)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local % max(torch.cuda.device_count(), 1)) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world)
    return rank, world, device


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def ordered(action: Callable[[int], None]):
    """Run ``action(rank)`` rank by rank between barriers (the ordered start-up print of main.py:53-57)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    for turn in range(world):
        if turn == rank:
            action(rank)
        barrier()


def broadcast_manifest(manifest=None):
    """Rank 0's manifest (any picklable object) to every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return manifest
    box = [manifest if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def shard_frames(frames: Sequence, rank: int, world_size: int, seed: int = 0) -> List:
    """Seeded permutation, then frame j of the permutation goes to rank j mod world_size.

    Unlike ``DistributedSampler`` (loader.py:8) no frame is duplicated to pad the shards: the reference relies on its
    "skip if the final checkpoint exists" guard (main.py:134-136) to make the padded duplicates harmless; here they never exist.
    """
    order = list(range(len(frames)))
    random.Random(seed).shuffle(order)
    return [frames[j] for position, j in enumerate(order) if position % world_size == rank]


def run_frames(frames: Sequence, optimise: Callable, checkpoint_path: Callable[[object], str] = None, frames_in_flight: int = 1):
    """Optimise this rank's frames; a frame whose final checkpoint exists is skipped (main.py:134-136).

    ``frames_in_flight`` > 1 runs that many frames at the same time on this rank's GPU, one host thread and one stream each: at the
    reference's 1000 rays per step the launch-bound box-only phase runs twice as fast with two frames, the compute-bound residual
    phase about 10 % faster; more than two host threads lose to the GIL.  ``optimise`` is then called from worker threads, inside ``torch.cuda.stream(<own stream>)``; a
    ``FrameOptimizer(graph=True)`` built there is safe (own scratch, own capture stream, captures serialised).
    The returned list keeps the order of ``frames``."""
    pending = []
    for frame in frames:
        path = checkpoint_path(frame) if checkpoint_path else None
        if not (path and os.path.exists(path)):
            pending.append((frame, path))

    def one(item):
        frame, path = item
        result = optimise(frame)
        if path:
            from .formats import atomic_torch_save
            atomic_torch_save(result, path)   # utils.Saver.save == torch.save(dict) (vsrd/utils.py:191-198), written atomically
        return frame

    if frames_in_flight <= 1:
        return [one(item) for item in pending]

    from concurrent.futures import ThreadPoolExecutor

    def on_own_stream(item):
        if torch.cuda.is_available():
            with torch.cuda.stream(torch.cuda.Stream()):
                frame = one(item)
                torch.cuda.current_stream().synchronize()
                return frame
        return one(item)

    with ThreadPoolExecutor(max_workers=frames_in_flight) as pool:
        return list(pool.map(on_own_stream, pending))
