"""Frame sharding and the launcher's collectives with 2 gloo ranks on CPU (the N > 1 path of bench.py / the launcher)."""
import math
import os
import socket
import tempfile

import pytest

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def test_bench_spawns_its_own_ranks_and_reports_what_ran():
    """`python bench.py --gpus 2` without torchrun spawns the two ranks itself; the line reports the ranks that actually passed the
    barriers (not the flag), the slowest rank's barrier-to-barrier time, and every rank's own time.  On this CPU-only container the
    render step is replaced by a sleep (--launcher-selftest, gloo) and the line says that it is not a measurement; without that
    flag and without a GPU the bench refuses to run instead of measuring something else."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--launcher-selftest", "--steps", "4", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                            # rank 0 prints ONE JSON line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert "NOT a measurement" in line["metric"] and line["value"] == 0.0
    own = line["per_rank_ms_per_step"]
    assert len(own) == 2 and own[1] > own[0] > 1.5                    # rank r sleeps 2 (1 + r) ms per step
    assert line["ms_per_step"] >= max(own) - 0.2                      # the reported time is the slowest rank's
    # under a launcher (the driver's torchrun form) the same script is a rank: WORLD_SIZE in the env, no second spawn
    single = subprocess.run([sys.executable, bench, "--gpus", "1", "--launcher-selftest", "--steps", "2", "--warmup", "0"],
                            capture_output=True, text=True, timeout=300, env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"))
    assert single.returncode == 0 and json.loads([l for l in single.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
    if not torch.cuda.is_available():
        refused = subprocess.run([sys.executable, bench, "--steps", "1"], capture_output=True, text=True, timeout=300, env=env)
        assert refused.returncode != 0 and "no CPU fallback" in refused.stderr
        assert not [l for l in refused.stdout.splitlines() if l.startswith("{")]


def test_bench_under_the_drivers_torchrun_command():
    """The driver's own launch line for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- with the render step replaced by a sleep (--launcher-selftest, gloo, CPU): the
    script is a rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), nothing is spawned twice, rank 0 prints ONE line with
    `n_gpus` = N ranks through the barriers, K steps, and the slowest rank's time."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--launcher-selftest"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and len(line["per_rank_ms_per_step"]) == 2
    assert line["ms_per_step"] >= max(line["per_rank_ms_per_step"]) - 0.2 and "NOT a measurement" in line["metric"]


def test_frame_launcher_under_torchrun(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 4 bench.py --native --gpus 2 --procs-per-gpu 2 --selftest ...`: under a launcher the
    frames/s entry point is a rank (no supervisor of its own: torchrun's --max-restarts plays that part), --nproc-per-node = GPUs x processes
    per GPU, local ranks 0, 1 take device 0 and 2, 3 device 1, and rank 0 prints the one line."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "bench.py"), "--native", "--gpus", "2", "--procs-per-gpu", "2", "--selftest", "--frames", "9", "--selftest-seconds", "0.05",
                          "--out", str(tmp_path)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["ranks"] == 4 and line["rank_devices"] == [0, 0, 1, 1] and line["frames"] == 9 and line["restarts"] == 0
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("frame_")) == [f"frame_{k:06d}" for k in range(9)]


@pytest.mark.gpu
def test_bench_two_ranks_with_the_real_step_on_one_gpu():
    """The multi-rank path of bench.py with the REAL render step: two ranks spawned by bench.py itself, both on
    cuda:0 (--ranks-share-gpu: gloo rendezvous, since RCCL refuses two ranks on one device), build on rank 0 behind a barrier, W warm-up
    + K timed steps between barriers, all_gather of the per-rank times, rank 0's single JSON line.  What the driver's 8-GPU run
    exercises except for the transport: there the line must show n_gpus = 8 and a per_rank_ms_per_step spread of a few per cent."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--ranks-share-gpu", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--views", "2", "--height", "94", "--width", "352"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and "NOT a scaling measurement" in line["metric"]
    rays = 2 * 94 * 352
    assert len(line["per_rank_ms_per_step"]) == 2 and min(line["per_rank_ms_per_step"]) > 0.0
    assert abs(line["value"] - 2 * rays / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]     # whole-job rays/s = both ranks' rays / slowest time
    assert line["ms_per_step"] >= max(line["per_rank_ms_per_step"]) - 0.05
    assert line["config"]["rays_per_gpu"] == rays and math.isfinite(line["config"]["final_loss"])


@pytest.mark.gpu
def test_native_frames_entry_point_two_ranks_on_one_gpu(tmp_path):
    """The frames/s entry point: `python bench.py --native --gpus 2` = vsrd_amd.launcher.main -- two ranks spawned
    by the launcher itself (both on cuda:0: --ranks-share-gpu, gloo), manifest broadcast, ordered start-up barrier, every rank
    optimising ITS frames with FrameOptimizer(graph=True) two at a time, atomic checkpoints in the reference's layout, gather of the
    per-rank report, ONE JSON line from rank 0; a second run over the same directory finds every frame done and optimises nothing
    (main.py:134-136).  The first 8-GPU node runs the same command without --ranks-share-gpu."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    command = [sys.executable, bench, "--native", "--gpus", "2", "--ranks-share-gpu", "--frames", "5", "--views", "3", "--instances", "4", "--height", "128",
               "--width", "128", "--rays", "256", "--samples", "32", "--num-steps", "40", "--warmup-steps", "12", "--frames-in-flight", "2",
               "--procs-per-gpu", "1", "--queue", "static", "--out", str(tmp_path)]
    out = subprocess.run(command, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["unit"] == "frames/s" and line["n_gpus"] == 2 and line["frames"] == 5 and sorted(line["per_rank_frames"]) == [2, 3]
    assert "NOT a scaling measurement" in line["metric"]
    assert abs(line["value"] - 5 / line["seconds"]) <= 1e-9 * line["value"] and line["seconds"] >= max(line["per_rank_seconds"]) - 1e-3
    assert all(math.isfinite(x) for x in line["mean_final_loss"])
    files = sorted(p for p in os.listdir(tmp_path))
    assert files == [f"frame_{k:06d}" for k in range(5)]
    payload = torch.load(os.path.join(tmp_path, "frame_000003", "step_39.pt"), weights_only=False)
    assert payload["step"] == 39 and set(payload["models"]) == {"detector", "hyper_distance_field"} and "optimizer" in payload and "scheduler" in payload
    again = subprocess.run(command, capture_output=True, text=True, timeout=900, env=env)
    assert again.returncode == 0, again.stderr[-3000:]
    second = json.loads([l for l in again.stdout.splitlines() if l.startswith("{")][0])
    assert second["frames"] == 0 and second["frames_skipped_as_done"] == 5


@pytest.mark.gpu
def test_native_frames_with_two_rank_processes_per_gpu(tmp_path):
    """`--gpus 1 --procs-per-gpu 2` on the GPU (the launcher's default layout): two rank processes on cuda:0 (gloo control plane: RCCL refuses
    two ranks on one device), one frame in flight each, frames sharded over both; the line counts ONE GPU and two ranks, every frame has its
    checkpoint, and the final losses are those of the same frames optimised by ONE process -- to 1e-5: a frame starts from parameters drawn for
    (seed, frame) (OptimizationConfig.init_seed), its samples come from Philox streams keyed the same way, and a slot's next frame walks a
    fresh loop's trajectory, so who runs a frame does not matter (the reference seeds once per rank: scripts/main.py:67-74).  The same holds for
    two frames in flight in ONE process (round 5 found the seed-and-draw of two frames' threads interleaving: optimization._initial_draw)."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    small = ["--frames", "4", "--views", "3", "--instances", "4", "--height", "128", "--width", "128", "--rays", "256", "--samples", "32", "--num-steps", "40",
             "--warmup-steps", "12"]
    lines = {}
    for procs in (2, 1, "threads"):          # two rank processes; one process; one process with two frames in flight (threads, streams, slots)
        layout = ["--procs-per-gpu", "1", "--frames-in-flight", "2"] if procs == "threads" else ["--procs-per-gpu", str(procs), "--frame-batch", "1", "--queue", "static"]
        out = subprocess.run([sys.executable, bench, "--native", "--gpus", "1", *small, *layout, "--out", str(tmp_path / str(procs))],
                             capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines[procs] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    line = lines[2]
    assert line["n_gpus"] == 1 and line["ranks"] == 2 and line["procs_per_gpu"] == 2 and line["rank_devices"] == [0, 0] and line["control_plane"] == "gloo"
    assert line["frames"] == 4 and line["per_rank_frames"] == [2, 2] and line["restarts"] == 0 and "NOT a scaling" not in line["metric"]
    assert abs(line["frames_per_s_per_gpu"] - line["value"]) <= 1e-12 and line["seconds"] >= max(line["per_rank_seconds"]) - 1e-3
    assert lines[1]["ranks"] == 1 and lines[1]["control_plane"] == "RCCL"
    assert sorted(os.listdir(tmp_path / "2")) == [f"frame_{k:06d}" for k in range(4)]
    assert set(line["final_loss_per_frame"]) == {"0", "1", "2", "3"}
    for frame, loss in line["final_loss_per_frame"].items():
        for other in (1, "threads"):
            theirs = lines[other]["final_loss_per_frame"][frame]
            assert math.isfinite(loss) and abs(loss - theirs) <= 1e-5 * max(abs(loss), 1.0), (other, frame, loss, theirs)
    assert line["frames_outside_slots"] == [] and lines[1]["frames_outside_slots"] == []


@pytest.mark.gpu
def test_supervisor_restarts_gpu_ranks_after_a_kill(tmp_path):
    """The restart path on the real stack: two rank processes on the GPU (the default layout), one of them is SIGKILLed from outside as soon
    as the job's first checkpoint exists -- a rank dying inside the runtime looks like this to everybody else.  The supervisor ends the
    attempt, starts both ranks again as fresh processes (which initialise the GPU next to whatever the dead process left behind), the second
    attempt skips the finished frames: exit code 0, `restarts: 1`, every frame has exactly one checkpoint, done + skipped = all."""
    import json
    import signal
    import subprocess
    import sys
    import time
    import psutil
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    frames = 12
    command = [sys.executable, "-m", "vsrd_amd.launcher", "--gpus", "1", "--procs-per-gpu", "2", "--frames", str(frames), "--views", "3", "--instances", "4", "--height", "128",
               "--width", "128", "--rays", "256", "--samples", "32", "--num-steps", "900", "--warmup-steps", "300", "--max-restarts", "2", "--out", str(tmp_path)]
    job = subprocess.Popen(command, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    killed = None
    deadline = time.time() + 600
    while time.time() < deadline and job.poll() is None and killed is None:
        done = [d for d in os.listdir(tmp_path) if d.startswith("frame_") and any(f.endswith(".pt") for f in os.listdir(tmp_path / d))]
        if done:
            ranks = [p for p in psutil.Process(job.pid).children(recursive=False) if p.is_running()]
            if len(ranks) == 2:
                killed = ranks[1].pid
                os.kill(killed, signal.SIGKILL)
                break
        time.sleep(0.02)
    out, err = job.communicate(timeout=900)
    assert killed is not None, "the job finished before a rank could be killed: " + err[-1000:]
    assert job.returncode == 0, err[-3000:]
    assert "exited with code" in err and "starting the ranks again" in err
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["restarts"] == 1 and line["ranks"] == 2 and line["frames"] + line["frames_skipped_as_done"] == frames and line["frames_skipped_as_done"] >= 1
    assert sorted(d for d in os.listdir(tmp_path) if d.startswith("frame_")) == [f"frame_{k:06d}" for k in range(frames)]
    assert all(os.listdir(tmp_path / f"frame_{k:06d}") == ["step_899.pt"] for k in range(frames))          # no temporary file of the killed rank left behind
    assert all(math.isfinite(x) for x in line["final_loss_per_frame"].values())


def test_supervisor_restarts_a_dead_rank_and_every_frame_is_done_once(tmp_path):
    """VERDICT r04 item 4 (README.md:146-155: the reference leans on `torchrun --max_restarts`; main.py:134-136: skip-if-done): a gloo rank
    is killed mid-job -- os._exit from inside its second frame's slot, no clean-up, its peer left in the final barrier.  The supervisor
    (launcher._supervise: a GPU-free parent polling its ranks) ends the attempt, starts BOTH ranks again as fresh processes on a new port,
    and the second attempt skips what has a checkpoint: the job returns 0, every frame has exactly one checkpoint, the frame the dying rank
    finished before it died is not optimised again, and the report counts done + skipped = all frames."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    command = [sys.executable, "-m", "vsrd_amd.launcher", "--selftest", "--gpus", "2", "--procs-per-gpu", "1", "--frames", "9", "--selftest-seconds", "0.2",
               "--selftest-fail", "1:1", "--max-restarts", "2", "--queue", "static", "--out", str(tmp_path)]
    out = subprocess.run(command, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "rank 1 exited with code 23" in out.stderr and "starting the ranks again" in out.stderr
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                              # the failed attempt printed nothing: its rank 0 never got to the report
    line = lines[0]
    assert line["restarts"] == 1 and line["n_gpus"] == 2 and line["frames"] + line["frames_skipped_as_done"] == 9 and line["frames_skipped_as_done"] >= 1
    frames = sorted(os.listdir(tmp_path))
    assert [f for f in frames if f.startswith("frame_")] == [f"frame_{k:06d}" for k in range(9)]
    payloads = [torch.load(os.path.join(tmp_path, f"frame_{k:06d}", "step_final.pt"), weights_only=False) for k in range(9)]
    assert [p["frame"] for p in payloads] == list(range(9))
    assert all(len(os.listdir(os.path.join(tmp_path, f"frame_{k:06d}"))) == 1 for k in range(9))     # no temporary files, no second checkpoint
    completed = [tuple(int(v) for v in row.split()) for row in open(os.path.join(tmp_path, "completed.log")).read().splitlines()]
    from vsrd_amd import launcher
    first_of_dying_rank = launcher.shard_frames(list(range(9)), 1, 2, seed=0)[0]
    assert [row for row in completed if row[0] == first_of_dying_rank] == [(first_of_dying_rank, 0, 1)]      # done before the rank died: not repeated
    assert {row[0] for row in completed} == set(range(9))
    assert all(sum(1 for row in completed if row[0] == k) <= 2 for k in range(9))                      # (a frame in flight when the attempt ended runs again)
    assert any(p["attempt"] == 1 for p in payloads) and payloads[first_of_dying_rank]["attempt"] == 0
    # no restarts allowed: the job fails with the dead rank's code instead of hanging in the survivor's barrier
    failing = subprocess.run([*command[:-1], str(tmp_path / "second"), "--max-restarts", "0"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert failing.returncode == 23 and "no restarts left" in failing.stderr


def test_procs_per_gpu_maps_consecutive_local_ranks_to_one_device(tmp_path):
    """`--gpus 2 --procs-per-gpu 2` (round 5: two rank processes per GPU overlap their kernels better than the streams of one process): the
    supervisor starts four ranks, local ranks 0, 1 report device 0 and 2, 3 device 1, the frames are sharded over all four, the line counts
    two GPUs and four ranks.  (Selftest: sleeps on gloo, no GPU; the device is what init_process_group(ranks_per_device=2) would select.)"""
    import json
    import subprocess
    import sys
    from vsrd_amd import launcher
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-m", "vsrd_amd.launcher", "--selftest", "--gpus", "2", "--procs-per-gpu", "2", "--frames", "10", "--selftest-seconds", "0.1",
                          "--queue", "static", "--out", str(tmp_path)], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")][-1]
    assert line["n_gpus"] == 2 and line["ranks"] == 4 and line["procs_per_gpu"] == 2 and line["rank_devices"] == [0, 0, 1, 1]
    assert line["frames"] == 10 and line["per_rank_frames"] == [len(launcher.shard_frames(list(range(10)), r, 4, seed=0)) for r in range(4)]
    completed = [tuple(int(v) for v in row.split()) for row in open(tmp_path / "completed.log").read().splitlines()]
    assert sorted(row[0] for row in completed) == list(range(10))
    os.environ.pop("LOCAL_RANK", None)
    # the device a GPU rank would take
    for local, expected in ((0, 0), (1, 0), (2, 1), (3, 1)):
        assert local // 2 == expected


def test_supervisor_ends_a_hung_attempt(tmp_path):
    """`--stall-timeout`: a rank that HANGS (inside the runtime, on a dead GPU) exits with nothing and its peers wait in their barrier for ever;
    the supervisor watches the checkpoint directory instead -- no new checkpoint for that long, and the attempt is ended and restarted like after
    a dead rank.  Here rank 1 of two gloo ranks goes to sleep for ever after its first frame; the job still finishes every frame."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    command = [sys.executable, "-m", "vsrd_amd.launcher", "--selftest", "--gpus", "2", "--procs-per-gpu", "1", "--frames", "8", "--selftest-seconds", "0.1",
               "--selftest-hang", "1:1", "--stall-timeout", "4", "--max-restarts", "1", "--out", str(tmp_path)]
    out = subprocess.run(command, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "no checkpoint for 4 s" in out.stderr and "starting the ranks again" in out.stderr
    line = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")][-1]
    assert line["restarts"] == 1 and line["frames"] + line["frames_skipped_as_done"] == 8 and line["frames_skipped_as_done"] >= 4
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("frame_")) == [f"frame_{k:06d}" for k in range(8)]


@pytest.mark.gpu
def test_native_frames_in_a_batch_match_the_same_frames_alone(tmp_path):
    """The launcher's default layout since round 6 -- ONE rank process per GPU (RCCL control plane) stepping `--frame-batch` frames together
    (optimization.FrameBatch) -- against frame slots with one frame per launch chain: ten frames (two full groups of four and a last group of
    two, captured at start-up because the static queue knows its tail), every frame with its checkpoint, and the final loss of EVERY frame equal
    to the last digit to the loss of the same frame optimised alone with the same work-item size of the MLP adjoint (the one number of the launch
    geometry its sums depend on: vsrd_render_config::adjoint_slots_per_item).  Also: the line says what ran (frame_batch, control plane, queue,
    MLP products, where the time went), and a second run skips everything."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    small = ["--frames", "10", "--views", "3", "--instances", "4", "--height", "128", "--width", "128", "--rays", "256", "--samples", "40", "--num-steps", "40",
             "--warmup-steps", "12", "--adjoint-item-slots", "8"]
    lines = {}
    for batch in (4, 1):
        command = [sys.executable, bench, "--native", "--gpus", "1", *small, "--frame-batch", str(batch), "--out", str(tmp_path / str(batch))]
        out = subprocess.run(command, capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines[batch] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    line = lines[4]
    assert line["n_gpus"] == 1 and line["ranks"] == 1 and line["procs_per_gpu"] == 1 and line["control_plane"] == "RCCL" and line["frame_batch"] == 4
    assert line["frames"] == 10 and line["queue"].startswith("static") and line["mlp_products"].startswith("split bf16") and line["restarts"] == 0
    assert line["graphs_per_slot"] == [8] and line["frames_outside_slots"] == [] and line["frames_with_unhealthy_draws"] == []      # both phases x {1, 4 steps} x {4, 2 frames}
    phases = line["phase_seconds"][0]
    assert phases["steps"] > 0 and phases["reset"] > 0 and abs(sum(phases.values()) - line["per_rank_seconds"][0]) < 0.5 * line["per_rank_seconds"][0] + 0.5
    assert sorted(os.listdir(tmp_path / "4")) == [f"frame_{k:06d}" for k in range(10)]
    assert set(line["final_loss_per_frame"]) == {str(k) for k in range(10)} == set(lines[1]["final_loss_per_frame"])
    for frame, loss in line["final_loss_per_frame"].items():
        assert math.isfinite(loss) and loss == lines[1]["final_loss_per_frame"][frame], (frame, loss, lines[1]["final_loss_per_frame"][frame])
    payload = torch.load(os.path.join(tmp_path, "4", "frame_000007", "step_39.pt"), weights_only=False)
    alone = torch.load(os.path.join(tmp_path, "1", "frame_000007", "step_39.pt"), weights_only=False)
    for name, value in payload["models"]["detector"].items():
        assert torch.equal(value, alone["models"]["detector"][name]), name
    for name, value in payload["models"]["hyper_distance_field"].items():
        assert torch.equal(value, alone["models"]["hyper_distance_field"][name]), name
    again = subprocess.run([sys.executable, bench, "--native", "--gpus", "1", *small, "--frame-batch", "4", "--out", str(tmp_path / "4")],
                           capture_output=True, text=True, timeout=900, env=env)
    assert again.returncode == 0, again.stderr[-3000:]
    second = json.loads([l for l in again.stdout.splitlines() if l.startswith("{")][-1])
    assert second["frames"] == 0 and second["frames_skipped_as_done"] == 10


def test_dynamic_queue_bounds_the_tail_of_eight_ranks_with_unequal_frames(tmp_path):
    """VERDICT r05 item 8 (vsrd/distributed/loader.py:4-9 hands every rank a fixed share; scripts/main.py:134-136 is the only guard): real
    frames differ in instance count and cost, and with the static split `frame j -> rank j mod world` the job ends when the unluckiest rank
    does.  `--queue dynamic`: the ranks take their next frame from ONE queue -- an atomic TCPStore counter on rank 0, no collective -- so no
    rank idles for more than one frame.  Eight gloo ranks, 48 frames that cost 0.05 s x (1 + 4 u_j): the ranks' own finishing times spread
    by less than the most expensive frame (+ scheduling slack) with the queue, every frame is done exactly once, and the static split of the
    same frames -- measured in the same test -- spreads by more."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    lines = {}
    for queue in ("dynamic", "static"):
        out = subprocess.run([sys.executable, "-m", "vsrd_amd.launcher", "--selftest", "--gpus", "8", "--frames", "48", "--selftest-seconds", "0.05", "--selftest-spread", "4",
                              "--queue", queue, "--max-restarts", "0", "--out", str(tmp_path / queue)], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        lines[queue] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        completed = [tuple(int(v) for v in row.split()) for row in open(tmp_path / queue / "completed.log").read().splitlines()]
        assert sorted(row[0] for row in completed) == list(range(48))                                # every frame exactly once
    line = lines["dynamic"]
    assert line["ranks"] == 8 and line["n_gpus"] == 8 and line["frames"] == 48 and line["queue"] == "dynamic" and sum(line["per_rank_frames"]) == 48
    most_expensive = 0.05 * (1 + 4)
    assert line["rank_finish_spread_seconds"] <= most_expensive + 0.15, line["per_rank_seconds"]      # tail <= one frame (+ process scheduling slack on 8 cores)
    # the static split of the same frames: its spread is what the shards' sums differ by (known in advance: the costs are a hash of the frame number)
    import random
    from vsrd_amd import launcher
    cost = lambda frame: 0.05 * (1.0 + 4.0 * random.Random(1000 + frame).random())
    shard_sums = [sum(cost(f) for f in launcher.shard_frames(list(range(48)), r, 8, seed=0)) for r in range(8)]
    assert lines["static"]["queue"] == "static" and lines["static"]["per_rank_frames"] == [6] * 8
    assert lines["static"]["rank_finish_spread_seconds"] >= 0.6 * (max(shard_sums) - min(shard_sums))
    assert max(shard_sums) - min(shard_sums) > most_expensive, "the example must be one in which the static split loses"
    assert max(lines["dynamic"]["per_rank_seconds"]) < max(lines["static"]["per_rank_seconds"])


def test_dynamic_queue_survives_a_dead_rank(tmp_path):
    """The queue and the supervisor together: a rank dies mid-job (exit code 23) on attempt 0, the supervisor starts all ranks again, the new
    attempt's queue starts over (its counter is keyed by the attempt) and the skip-if-done guard drops what has a checkpoint: every frame ends
    with exactly one checkpoint and no frame that was finished is optimised twice."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    out = subprocess.run([sys.executable, "-m", "vsrd_amd.launcher", "--selftest", "--gpus", "3", "--frames", "12", "--selftest-seconds", "0.15", "--queue", "dynamic",
                          "--selftest-fail", "2:1", "--max-restarts", "1", "--out", str(tmp_path)], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["restarts"] == 1 and line["queue"] == "dynamic" and line["frames"] + line["frames_skipped_as_done"] >= 12 and line["frames_skipped_as_done"] >= 1
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("frame_")) == [f"frame_{k:06d}" for k in range(12)]
    completed = [tuple(int(v) for v in row.split()) for row in open(tmp_path / "completed.log").read().splitlines()]
    assert {row[0] for row in completed} == set(range(12))
    finished_first = {row[0] for row in completed if row[1] == 0}
    assert not finished_first & {row[0] for row in completed if row[1] == 1}                          # what attempt 0 finished, attempt 1 left alone


def test_frame_queue_hands_every_frame_out_once():
    """launcher.FrameQueue: the static split is shard_frames; the dynamic one takes groups from a shared counter (here: a stand-in store)."""
    from vsrd_amd import launcher

    class Counter:
        def __init__(self):
            self.values = {}

        def add(self, key, amount):
            self.values[key] = self.values.get(key, 0) + amount
            return self.values[key]

    frames = list(range(11))
    for rank in range(3):
        queue = launcher.FrameQueue(frames, rank, 3, seed=5)
        taken = []
        while True:
            group = queue.take(2)
            if not group:
                break
            taken += group
        assert taken == launcher.shard_frames(frames, rank, 3, seed=5) == queue.taken
    store = Counter()
    queues = [launcher.FrameQueue(frames, rank, 3, seed=5, store=store) for rank in range(3)]
    seen, turn = [], 0
    while True:
        group = queues[turn % 3].take(4)
        turn += 1
        if not group:
            break
        assert len(group) <= 4
        seen += group
    assert sorted(seen) == frames and seen == queues[0].order and all(q.take(4) == [] for q in queues)


def test_frame_batch_never_exceeds_a_ranks_share():
    """Groups are handed out whole: a batch larger than total / world would leave ranks without work (32 frames, 8 ranks, batches of 16)."""
    from vsrd_amd import launcher
    assert launcher.effective_frame_batch(16, 32, 1) == 16
    assert launcher.effective_frame_batch(16, 32, 8) == 4
    assert launcher.effective_frame_batch(16, 33, 8) == 5
    assert launcher.effective_frame_batch(16, 3, 8) == 1
    assert launcher.effective_frame_batch(1, 1000, 8) == 1
    assert launcher.effective_frame_batch(16, 10000, 8) == 16
    for total in (1, 7, 32, 100):
        for world in (1, 2, 8):
            batch = launcher.effective_frame_batch(16, total, world)
            groups = -(-total // batch)
            assert groups >= min(world, total)                      # every rank (or every frame) gets a group
