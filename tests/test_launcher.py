"""Frame sharding and the launcher's collectives with 2 gloo ranks on CPU (the N > 1 path of bench.py / the launcher)."""
import os
import socket
import tempfile

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from vsrd_amd import launcher
    r, w, device = launcher.init_process_group(backend="gloo")
    assert (r, w) == (rank, world) and device.type == "cpu"
    manifest = launcher.broadcast_manifest({"frames": list(range(11)), "seed": 3} if rank == 0 else None)
    assert manifest == {"frames": list(range(11)), "seed": 3}
    mine = launcher.shard_frames(manifest["frames"], rank, world, seed=manifest["seed"])
    order = []
    launcher.ordered(lambda k: order.append(k))
    assert order == [rank]
    # frame 4's checkpoint already exists -> skipped by whichever rank owns it (idempotent restart)
    path = lambda f: os.path.join(tmp, f"frame_{f}", "step_final.pt")
    if rank == 0:
        os.makedirs(os.path.dirname(path(4)), exist_ok=True)
        torch.save({"step": -1}, path(4))
    launcher.barrier()
    done = launcher.run_frames(mine, lambda f: {"step": 2999, "frame": f, "rank": rank}, path)
    assert 4 not in done
    # weak-scaling timing reduction used by bench.py: MAX over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    assert t.item() == float(world)
    gathered = [None] * world
    torch.distributed.all_gather_object(gathered, mine)
    if rank == 0:
        flat = sorted(f for part in gathered for f in part)
        assert flat == list(range(11)), flat                       # every frame exactly once, no padding duplicates
        assert abs(len(gathered[0]) - len(gathered[1])) <= 1
        assert all(os.path.exists(path(f)) for f in range(11))
    launcher.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_frame_sharding_gloo():
    from vsrd_amd import launcher
    assert launcher.shard_frames(list("abcdefg"), 0, 1) and sorted(launcher.shard_frames(list("abcdefg"), 0, 1)) == list("abcdefg")
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)


def test_frames_in_flight_runs_every_frame_once():
    """run_frames(frames_in_flight=2): worker threads (one stream each on a GPU; plain threads here), order kept, restart guard kept."""
    import threading
    from vsrd_amd import launcher
    seen, lock = [], threading.Lock()

    def optimise(frame):
        with lock:
            seen.append((frame, threading.current_thread().name))
        return {"frame": frame}

    with tempfile.TemporaryDirectory() as tmp:
        path = lambda f: os.path.join(tmp, f"frame_{f}", "step_final.pt")
        os.makedirs(os.path.dirname(path(2)), exist_ok=True)
        torch.save({"step": -1}, path(2))
        done = launcher.run_frames(list(range(7)), optimise, path, frames_in_flight=2)
        assert done == [0, 1, 3, 4, 5, 6]
        assert sorted(f for f, _ in seen) == done and all(os.path.exists(path(f)) for f in range(7))
        assert torch.load(path(5))["frame"] == 5
