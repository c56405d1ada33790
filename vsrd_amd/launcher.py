"""Frame-sharded multi-GPU launcher: one rank process per GPU on RCCL (the default since round 6: a rank optimises `--frame-batch` frames in
lock-step, optimization.FrameBatch, which fills the GPU from ONE process; `--procs-per-gpu P` > 1 -- round 5's answer, P processes per GPU
with a gloo control plane -- is still there), frames handed to the ranks from a shared queue (or split statically), no data-path collective;
persistent frame slots / batches inside a rank; a supervisor that restarts dead ranks.

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


def init_process_group(backend=None, ranks_per_device=1):
    """env:// rendezvous as with torchrun (main.py:49).  Returns (rank, world_size, device).  `ranks_per_device` consecutive local ranks
    share a device (the launcher's --procs-per-gpu)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) // max(int(ranks_per_device), 1)
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


class FrameQueue:
    """Where a rank's next frames come from.

    ``static``: the rank's own shard (``shard_frames``: frame j of the seeded permutation to rank j mod world), fixed before the clock starts.
    ``dynamic`` (SURVEY section 8e "tail imbalance"): ONE queue for the job -- the seeded permutation of the manifest's
    frames -- and a rank that has nothing left to do takes the next ``n`` frames of it with one atomic ``TCPStore.add`` on a counter that rank 0's
    store serves.  No collective, nobody waits for anybody: real frames differ in instance count and cost, and with a static split the job
    ends when the unluckiest rank does; with the queue the ranks finish within one group of frames of each other.  The skip-if-done guard
    (main.py:134-136) makes a restarted job idempotent either way: the queue of a new attempt starts over and finished frames are skipped."""

    def __init__(self, frames, rank=0, world=1, seed=0, store=None, key="vsrd_next_frame"):
        self.dynamic = store is not None
        self.store, self.key = store, key
        order = list(range(len(frames)))
        random.Random(seed).shuffle(order)
        self.order = [frames[j] for j in order]
        self.mine = [f for position, f in enumerate(self.order) if position % world == rank]       # (= shard_frames: the static split)
        self.cursor = 0
        self.taken = []

    def take(self, n=1):
        """The next (at most) ``n`` frames for this rank; [] when the job has none left."""
        n = max(int(n), 1)
        if self.dynamic:
            end = int(self.store.add(self.key, n))
            group = self.order[max(end - n, 0):min(end, len(self.order))] if end - n < len(self.order) else []
        else:
            group = self.mine[self.cursor:self.cursor + n]
            self.cursor += len(group)
        self.taken += group
        return group


def pending_frames(frames: Sequence, checkpoint_path: Callable[[object], str] = None):
    """[(frame, path)] of the frames whose final checkpoint does not exist yet (main.py:134-136), with what a rank that was killed inside
    formats.atomic_torch_save left behind in their folders removed (the frame is this rank's now: nobody else writes there)."""
    pending = []
    for frame in frames:
        path = checkpoint_path(frame) if checkpoint_path else None
        if not (path and os.path.exists(path)):
            pending.append((frame, path))
            folder, stem = (os.path.dirname(path), os.path.basename(path) + ".tmp.") if path else (None, None)
            if folder and os.path.isdir(folder):
                for name in os.listdir(folder):
                    if name.startswith(stem):
                        os.remove(os.path.join(folder, name))
    return pending


def run_frames(frames, optimise: Callable, checkpoint_path: Callable[[object], str] = None, frames_in_flight: int = 1):
    """Optimise this rank's frames -- a sequence, or a ``FrameQueue`` the frames are taken from one by one -- skipping every frame whose final
    checkpoint exists (main.py:134-136).

    ``frames_in_flight`` > 1 runs that many frames at the same time on this rank's GPU, one host thread and one stream each: at the
    reference's 1000 rays per step the launch-bound box-only phase runs twice as fast with two frames, the compute-bound residual
    phase about 10 % faster.  ``optimise`` is then called from worker threads, inside ``torch.cuda.stream(<own stream>)``; a
    ``FrameOptimizer(graph=True)`` built there is safe because nothing it calls into HIP -- construction, eager warm-up steps, replay
    launches, graph destruction, host synchronisations -- runs during ANOTHER frame's capture, and captures (with a graph's first launch)
    run alone: optimization._CaptureGate (round 4, one box: 0.749 frames/s with one frame in flight, 0.91 with three; without the gate a
    dead rank about every tenth run of 36 frames).
    Returns the frames optimised, in the order of ``frames`` (a queue: in the order they were taken)."""
    import threading

    def source():
        if hasattr(frames, "take"):
            while True:
                group = frames.take(1)
                if not group:
                    return
                yield group[0]
        else:
            yield from frames

    numbered = enumerate(source())
    take_lock = threading.Lock()

    def next_pending():
        """(index, frame, path) of the next frame that still has to be optimised, or None."""
        while True:
            with take_lock:
                item = next(numbered, None)
            if item is None:
                return None
            found = pending_frames([item[1]], checkpoint_path)
            if found:
                return item[0], found[0][0], found[0][1]

    def work():
        finished = []
        while True:
            item = next_pending()
            if item is None:
                return finished
            index, frame, path = item
            result = optimise(frame)
            if path:
                from .formats import atomic_torch_save
                atomic_torch_save(result, path)   # utils.Saver.save == torch.save(dict) (vsrd/utils.py:191-198), written atomically
            finished.append((index, frame))

    if frames_in_flight <= 1:
        return [frame for _, frame in work()]

    from concurrent.futures import ThreadPoolExecutor
    # a worker thread starts with device 0 current: it takes over this thread's device (the rank's GPU) before it makes its stream
    device = torch.cuda.current_device() if torch.cuda.is_available() else None

    def on_own_stream():
        if torch.cuda.is_available():
            torch.cuda.set_device(device)
            from . import optimization
            with torch.cuda.stream(torch.cuda.Stream(device=device)):
                finished = work()
                with optimization.exclusive_device_access():      # (a host synchronisation: not next to another frame's capture)
                    torch.cuda.current_stream().synchronize()
                return finished
        return work()

    with ThreadPoolExecutor(max_workers=frames_in_flight) as pool:
        futures = [pool.submit(on_own_stream) for _ in range(frames_in_flight)]
        finished = [item for future in futures for item in future.result()]
    return [frame for _, frame in sorted(finished, key=lambda item: item[0])]


# ---------------------------------------------------------------------------------------------------------------------------------
# frames/s: the unit the reference shards (README.md:128 "about 15 minutes per frame"; main.py:106-136 one optimisation per frame)
#   python -m vsrd_amd.launcher --gpus N --frames K      (or under torchrun; `python bench.py --native --gpus N ...` forwards here)
# ---------------------------------------------------------------------------------------------------------------------------------

def synthetic_frame_inputs(device, frame, views, instances, height=376, width=1408):
    """FrameInputs of synthetic frame number `frame` (synthetic.synthetic_frame(seed = frame): KITTI-360 intrinsics, boxes in the
    reference's ranges): soft masks rendered from the true boxes at the sharp end of the schedule, ground-truth 2-D boxes projected
    from them, every instance visible in every view."""
    from . import fields, models, operations, optimization, rendering, synthetic
    K, E, raw_loc, raw_dim, raw_ori = synthetic.synthetic_frame(int(frame), views, height, width, instances)
    K, E = K.to(device), E.to(device)
    det = models.BoxParameters3D(1, instances).to(device)
    with torch.no_grad():
        det.locations.copy_(raw_loc); det.dimensions.copy_(raw_dim); det.orientations.copy_(raw_ori)
        out = det()
        cam, dirs = rendering.ray_casting((height, width), K, E)
        block = fields.FieldBlock(fields.pack_instances(out["locations"][0], out["orientations"][0], out["dimensions"][0]), 0.1, None, None)
        origins = cam[:, None, None, :].expand(views, height, width, 3).reshape(-1, 3).contiguous()
        soft = rendering.render_hierarchical(block, origins, dirs.reshape(-1, 3), (0.0, 100.0), 64, 0.1, 1.0, seed=1,
                                             skip_exact_misses=True)["labels"].clamp(0, 1).reshape(views, height, width, instances).contiguous()
        gt_boxes, _ = operations.project_boxes_multi_view(out["boxes_3d"][0], E, K, (height, width))
    return optimization.FrameInputs((height, width), K, E, soft, gt_boxes, torch.ones(views, instances, dtype=torch.bool, device=device))


def _newest_change(folder):
    """Modification time of the newest entry of the checkpoint directory: a frame's folder changes when its checkpoint is renamed into it."""
    try:
        return max([os.stat(folder).st_mtime, *(entry.stat().st_mtime for entry in os.scandir(folder))])      # (the folder itself: rank 0 touches it when its set-up is done)
    except OSError:
        return 0.0


def _wait_for_ranks(children, grace_seconds=10.0, progress_folder=None, stall_seconds=0.0):
    """Poll the ranks until all have exited.  The first rank that exits non-zero takes the others with it (SIGTERM, SIGKILL after
    `grace_seconds`): a rank whose peer died would otherwise sit in its final barrier until the collective's watchdog fires, tens of
    minutes later.  `stall_seconds` > 0: an attempt during which NO checkpoint appears in `progress_folder` for that long (counted from the
    attempt's start) ends the same way -- a rank that hangs inside the runtime exits with nothing, and its peers wait for it for ever.
    Returns None when every rank exited 0, else (rank, exit code) of the first failure ((-1, 124) for a stall)."""
    import time
    pending = dict(enumerate(children))
    failed, deadline = None, None
    started = time.time()
    while pending:
        for rank, child in list(pending.items()):
            code = child.poll()
            if code is None:
                continue
            del pending[rank]
            if code != 0 and failed is None:
                failed, deadline = (rank, code), time.monotonic() + grace_seconds
                for other in pending.values():
                    other.terminate()
        if failed is None and pending and stall_seconds > 0 and time.time() - max(started, _newest_change(progress_folder)) > stall_seconds:
            failed, deadline = (-1, 124), time.monotonic() + grace_seconds
            for other in pending.values():
                other.terminate()
        if failed is not None and pending and time.monotonic() > deadline:
            for other in pending.values():
                other.kill()
            deadline = float("inf")
        if pending:
            time.sleep(0.05)
    return failed


def _supervise(argv, args):
    """No launcher around us: this process stays GPU-free and supervises `--gpus` rank processes (README.md:146-155: the reference leans on
    `torchrun --max_restarts` for the same thing, and on its skip-if-done guard, main.py:134-136, for what a restart repeats).

    A rank that dies -- a segmentation fault inside the runtime, an out-of-memory kill -- strands its shard and leaves its peers in a
    barrier.  The supervisor ends the attempt (see _wait_for_ranks) and starts ALL ranks again as FRESH child processes, up to
    `--max-restarts` times: nothing that has initialised a GPU is ever re-executed, the rendezvous gets a new port, and every frame whose
    final checkpoint exists is skipped, so a restart costs the frames that were in flight.  The checkpoint directory therefore has to
    outlive an attempt: the supervisor fixes it (a temporary directory unless --out names one) and hands it to the ranks."""
    import socket
    import subprocess
    import sys
    import tempfile
    # (counting devices does not initialise the runtime on this image; a selftest has no devices to count)
    if not args.ranks_share_gpu and not args.selftest and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"vsrd_amd.launcher --gpus {args.gpus}: this node has {torch.cuda.device_count()} visible GPU(s)")
    ranks = args.gpus * args.procs_per_gpu
    out_dir = args.out or tempfile.mkdtemp(prefix="vsrd_frames_")
    if not args.out:
        argv = list(argv) + ["--out", out_dir]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    failed = None
    for attempt in range(args.max_restarts + 1):
        with socket.socket() as s:                 # (a free port now; should somebody take it before rank 0 binds it, the attempt fails and the next one asks again)
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        children = []
        for rank in range(ranks):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(ranks), LOCAL_WORLD_SIZE=str(ranks), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            children.append(subprocess.Popen([sys.executable, "-m", "vsrd_amd.launcher", *argv, "--attempt", str(attempt)], env=env, cwd=root))
        failed = _wait_for_ranks(children, progress_folder=out_dir, stall_seconds=args.stall_timeout)
        if failed is None:
            if not args.out:                   # (the supervisor's own temporary directory -- full per-frame checkpoints -- goes with the job)
                import shutil
                shutil.rmtree(out_dir, ignore_errors=True)
            return 0
        again = attempt < args.max_restarts
        print(f"[vsrd_amd.launcher] attempt {attempt}: " + (f"no checkpoint for {args.stall_timeout:g} s (--stall-timeout): the ranks were ended; " if failed[0] < 0 else
                                                           f"rank {failed[0]} exited with code {failed[1]}; ")
              + ("starting the ranks again as fresh processes (frames with a final checkpoint are skipped)" if again else "no restarts left"),
              file=sys.stderr, flush=True)
    return abs(failed[1]) or 1


class _RenderWork:
    """What ONE PROCESS does with the frames it takes from the queue on its GPU: frame slots / the frame batch (and, with a static queue,
    the inputs) before the clock, then the frames."""

    def __init__(self, args, manifest, device, queue):
        self.args, self.manifest, self.device, self.queue = args, manifest, device, queue
        self.config = dict(num_steps=args.num_steps, warmup_steps=args.warmup_steps, num_rays=args.rays, num_samples=args.samples,
                           mlp_split_bf16=not args.fp32_mlp, mlp_adjoint_item_slots=args.adjoint_item_slots)

    def path_of(self, frame):
        return os.path.join(self.manifest["out"], f"frame_{int(frame):06d}", f"step_{self.args.num_steps - 1}.pt")

    def init_seed(self, frame):
        # a frame starts from parameters drawn for (seed, frame): the same optimisation whichever rank, process, slot or batch row runs it
        # (the reference seeds once per rank, scripts/main.py:67-74)
        return (int(self.manifest["seed"]) * 1000003 + int(frame)) & 0x7FFFFFFF

    def inputs_of(self, frame):
        """A static queue's inputs are resident before the clock (what main.py:106-316 prepares per frame is the dataset's work, not the
        loop's); a dynamic queue's are built when the frame is taken -- inside the clock, like the reference's loader -- and dropped after."""
        if frame in self.inputs:
            return self.inputs[frame]
        args = self.args
        return synthetic_frame_inputs(self.device, frame, args.views, args.instances, args.height, args.width)

    def optimization_config(self, **more):
        from . import optimization
        return optimization.OptimizationConfig(seed=int(self.manifest["seed"]), **self.config, **more)

    def prepare(self):
        """Before the clock: the frame slots (round 5) or the frame batch (round 6) -- every persistent loop constructed and ALL its hipGraphs
        captured here, before any worker thread exists -- and, with a static queue, this rank's inputs.  A frame is then a reset (copies and
        fills), graph replays and a checkpoint: no construction, no eager steps, no capture.  (--fresh-loops: the round-4 form, for A/B.)"""
        import queue
        import time
        from . import optimization
        args = self.args
        mine = [] if self.queue.dynamic else [frame for frame in self.queue.mine if not os.path.exists(self.path_of(frame))]
        self.inputs = {frame: synthetic_frame_inputs(self.device, frame, args.views, args.instances, args.height, args.width) for frame in mine}
        self.slots, self.batch = queue.Queue(), None
        self.setup_seconds, self.graphs_per_slot, self.unsuitable_founders = 0.0, 0, []
        # the frames a slot or batch may be founded on: this rank's own (static), or the head of the job's queue (dynamic: built here, once)
        candidates = mine if not self.queue.dynamic else [f for f in self.queue.order if not os.path.exists(self.path_of(f))][:max(args.frame_batch, 1) + 2]
        if candidates and not args.fresh_loops:
            t_setup = time.perf_counter()
            if args.frame_batch > 1:
                founders = [self.inputs_of(f) for f in candidates[:args.frame_batch]]
                founders += [founders[-1]] * (args.frame_batch - len(founders))           # (fewer frames than rows: the shape is what matters)
                try:
                    self.batch = optimization.FrameBatch(founders, self.optimization_config(), self.device)
                    self.unsuitable_founders += [int(candidates[min(f, len(candidates) - 1)]) for f in self.batch.unsuitable_founders]
                    tail = len(mine) % args.frame_batch if mine else 0                   # a static queue's last group is known now: capture it too
                    self.graphs_per_slot = self.batch.capture_all(actives=sorted({args.frame_batch, tail} - {0}, reverse=True))
                except optimization.UnsuitableFrameError:                                 # (not one of the founders suits the table sampler)
                    self.batch = None                                                     # every frame then runs in a loop of its own
                    self.unsuitable_founders += [int(f) for f in candidates[:args.frame_batch]]
            else:
                for _ in range(min(args.frames_in_flight, len(candidates))):
                    for founder in candidates:      # (a slot is founded on a frame that suits the table sampler its graphs will draw from)
                        try:
                            loop = optimization.FrameOptimizer(self.inputs_of(founder), self.optimization_config(), self.device, graph=True, persistent=True)
                        except optimization.UnsuitableFrameError:
                            self.unsuitable_founders.append(int(founder))
                            continue
                        self.graphs_per_slot = loop.capture_all()
                        self.slots.put(loop)
                        break
            torch.cuda.synchronize()
            self.setup_seconds = time.perf_counter() - t_setup
        self.captures_before = optimization.exclusive_device_access().capture_seconds

    def _own_loop(self, frame):
        """A frame outside the slots / the batch (--fresh-loops, or importance weights that do not suit the table sampler): a loop of its own."""
        from . import formats, optimization
        loop = optimization.FrameOptimizer(self.inputs_of(frame), self.optimization_config(init_seed=self.init_seed(frame)), self.device, graph=True)
        record = loop.run(self.args.num_steps)
        with optimization.exclusive_device_access():
            torch.cuda.current_stream().synchronize()
            loss = float(record["loss"])
            payload = formats.checkpoint_payload(loop, step=self.args.num_steps - 1, metrics={}, host=True)
        loop.close()
        return loss, payload

    def run(self):
        from . import formats, optimization
        args, losses, fallbacks, unhealthy = self.args, {}, [], []

        def health(frame, loop):
            # the device sampler's sticky flags, read at the END of a frame (a slot's next reset() rebuilds the table and clears them)
            if loop.ray_table is not None and loop.ray_table.incomplete() or loop.workspace.sampler_overflowed(self.device):
                unhealthy.append(int(frame))
                import warnings
                warnings.warn(f"frame {int(frame)}: some ray draw of this frame ran out of picks or overflowed its candidate list: those draws repeat "
                              "rays / are not reproducible (csrc/ray_sampling.h)", RuntimeWarning)

        def optimise(frame):
            slot = self.slots.get() if self._have_slots else None      # (--fresh-loops, or no frame could found a slot)
            try:
                if slot is None or not slot.reset(self.inputs_of(frame), init_seed=self.init_seed(frame)):
                    fallbacks.append(int(frame))
                    losses[frame], payload = self._own_loop(frame)
                    return payload
                record = slot.run(args.num_steps)
                with optimization.exclusive_device_access():       # (host synchronisations and copies: refused now and then while another frame's thread captures)
                    torch.cuda.current_stream().synchronize()
                    losses[frame] = float(record["loss"])
                    health(frame, slot)
                    return formats.checkpoint_payload(slot, step=args.num_steps - 1, metrics={}, host=True)
            finally:
                if slot is not None:
                    self.slots.put(slot)

        self._have_slots = not args.fresh_loops and not self.slots.empty()
        if self.batch is not None:
            done = self._run_batches(losses, fallbacks, health)
        else:
            done = run_frames(self.queue, optimise, self.path_of, frames_in_flight=args.frames_in_flight)
        torch.cuda.synchronize()
        taken = len(self.queue.taken)
        return dict(frames=len(done), skipped=taken - len(done),
                    gate_capture_seconds=optimization.exclusive_device_access().capture_seconds - self.captures_before,
                    slot_setup_seconds=self.setup_seconds, graphs_per_slot=self.graphs_per_slot, final_losses={int(f): losses[f] for f in sorted(losses)},
                    frames_outside_slots=sorted(fallbacks), unsuitable_founders=self.unsuitable_founders, frames_with_unhealthy_draws=sorted(unhealthy),
                    phase_seconds=getattr(self, "phase_seconds", None))

    def _run_batches(self, losses, fallbacks, health):
        """Groups of --frame-batch frames through the FrameBatch: reset row by row, ONE graph replay per (four) step(s) for the whole group,
        then the group's checkpoints.  Frames the batch cannot take (their weights do not suit its table sampler) run alone afterwards."""
        import time
        from . import formats, optimization
        args, batch, done = self.args, self.batch, []
        clock = self.phase_seconds = dict(inputs=0.0, reset=0.0, steps=0.0, checkpoint_copy=0.0, checkpoint_write=0.0)
        while True:
            group = pending_frames(self.queue.take(args.frame_batch), self.path_of)
            if not group:
                if self.queue.dynamic and len(self.queue.taken) < len(self.queue.order) or not self.queue.dynamic and self.queue.cursor < len(self.queue.mine):
                    continue                                        # (a group whose frames were all finished by an earlier attempt)
                break
            rows, alone = [], []
            for frame, path in group:
                t0 = time.perf_counter()
                inputs = self.inputs_of(frame)
                t1 = time.perf_counter()
                (rows if batch.reset(len(rows), inputs, init_seed=self.init_seed(frame)) else alone).append((frame, path))
                clock["inputs"] += t1 - t0
                clock["reset"] += time.perf_counter() - t1
            if rows:
                t0 = time.perf_counter()
                batch.run(args.num_steps, active=len(rows))
                with optimization.exclusive_device_access():
                    torch.cuda.current_stream().synchronize()
                    t1 = time.perf_counter()
                    payloads = []
                    for row, (frame, path) in enumerate(rows):
                        losses[frame] = float(batch.outputs(row)["loss"])
                        health(frame, batch.frames[row])
                        payloads.append(formats.checkpoint_payload(batch.frames[row], step=args.num_steps - 1, metrics={}, host=True))
                t2 = time.perf_counter()
                for (frame, path), payload in zip(rows, payloads):
                    formats.atomic_torch_save(payload, path)
                    done.append(frame)
                clock["steps"] += t1 - t0
                clock["checkpoint_copy"] += t2 - t1
                clock["checkpoint_write"] += time.perf_counter() - t2
            for frame, path in alone:
                fallbacks.append(int(frame))
                losses[frame], payload = self._own_loop(frame)
                formats.atomic_torch_save(payload, path)
                done.append(frame)
        return done


class _SleepWork:
    """`--selftest`: a frame is a sleep and a small checkpoint.  `--selftest-fail RANK:FRAMES` makes that rank die (exit code 23, no
    clean-up, no goodbye to the process group) on attempt 0 once it has finished FRAMES frames.  `--selftest-spread X`: frame j costs
    --selftest-seconds x (1 + X u_j), u_j in [0, 1) a hash of j -- frames of unequal cost, what a static split balances badly."""

    def __init__(self, args, manifest, rank, queue):
        self.args, self.manifest, self.rank, self.queue = args, manifest, rank, queue
        self.fail_rank, self.fail_after = (int(v) for v in args.selftest_fail.split(":")) if args.selftest_fail else (-1, 0)
        self.hang_rank, self.hang_after = (int(v) for v in args.selftest_hang.split(":")) if args.selftest_hang else (-1, 0)

    def path_of(self, frame):
        return os.path.join(self.manifest["out"], f"frame_{int(frame):06d}", "step_final.pt")

    def prepare(self):
        os.makedirs(self.manifest["out"], exist_ok=True)

    def cost(self, frame):
        return self.args.selftest_seconds * (1.0 + self.args.selftest_spread * random.Random(1000 + int(frame)).random())

    def run(self):
        import time
        args, finished = self.args, []

        def optimise(frame):
            if args.attempt == 0 and self.rank == self.fail_rank and len(finished) >= self.fail_after:
                os._exit(23)
            if args.attempt == 0 and self.rank == self.hang_rank and len(finished) >= self.hang_after:
                time.sleep(1.0e6)
            time.sleep(self.cost(frame))
            finished.append(frame)
            with open(os.path.join(self.manifest["out"], "completed.log"), "a") as log:      # (O_APPEND: one short line per frame, whole)
                log.write(f"{int(frame)} {args.attempt} {self.rank}\n")
            return dict(frame=int(frame), attempt=args.attempt, rank=self.rank)

        done = run_frames(self.queue, optimise, self.path_of, frames_in_flight=1)
        return dict(frames=len(done), skipped=len(self.queue.taken) - len(done), gate_capture_seconds=0.0, slot_setup_seconds=0.0, graphs_per_slot=0, final_losses={},
                    frames_outside_slots=[], unsuitable_founders=[], frames_with_unhealthy_draws=[])


DEFAULT_FRAME_BATCH = 16


def effective_frame_batch(requested: int, total_frames: int, world_size: int) -> int:
    """Frames one rank steps together: what was asked for, but never more than a rank's share of the job -- groups are handed out whole
    (FrameQueue.take), so 32 frames in batches of 16 would occupy two of eight ranks.  Every rank computes the same number."""
    if requested <= 1:
        return 1
    share = -(-max(int(total_frames), 1) // max(int(world_size), 1))
    return max(1, min(int(requested), share))


def main(argv=None):
    """Optimise K synthetic frames, sharded over the ranks, each rank keeping `--frames-in-flight` frames on its GPU at a time with
    ``FrameOptimizer(graph=True)``; checkpoints through formats.checkpoint_payload (atomic, restartable: a frame whose final checkpoint
    exists is skipped, main.py:134-136).  Rank 0 broadcasts the manifest, the ranks start behind an ordered barrier, and rank 0 prints
    ONE JSON line: whole-job frames/s (K over the slowest rank's barrier-to-barrier time), frames/s per GPU, every rank's own time."""
    import argparse
    import json
    import sys
    import tempfile
    import time
    parser = argparse.ArgumentParser(prog="python -m vsrd_amd.launcher")
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--frames", type=int, default=0, help="frames of the whole job (default: two rounds of --frames-in-flight per process)")
    parser.add_argument("--frames-in-flight", type=int, default=1,
                        help="frames optimised at the same time by ONE process (one host thread, one stream and one persistent frame slot each; "
                             "1 = no threads at all).  Round 5, one box, split-bf16 MLP: 0.82 / 0.86 / 0.89 frames/s with 1 / 3 / 5 in one process")
    parser.add_argument("--frame-batch", type=int, default=None,
                        help="frames ONE process optimises in lock-step, every launch of a step covering all of them (optimization.FrameBatch; "
                             "include/vsrd_hip.h 'frame batches'): what fills the GPU from one RCCL rank.  1: one frame per launch chain (frame slots, "
                             "--frames-in-flight / --procs-per-gpu: the round-5 layouts).  Default: %d, or 1 when one of those layouts is asked for "
                             "(--frames-in-flight > 1, --procs-per-gpu > 1, --fresh-loops) or for a selftest" % DEFAULT_FRAME_BATCH)
    parser.add_argument("--procs-per-gpu", type=int, default=None,
                        help="rank processes per GPU (local ranks P g .. P g + P - 1 use GPU g); default 1 -- one RCCL rank per GPU -- or, under "
                             "torchrun, LOCAL_WORLD_SIZE / visible GPUs.  Round 5's layout was P = 2 with --frame-batch 1: the kernels of two processes "
                             "overlap better than the streams of one (1.02-1.11 frames/s against 0.82, profiles/r05/frames_per_s.txt); RCCL refuses two "
                             "ranks on one device, so with P > 1 the control plane -- barriers, manifest broadcast, gather of the report; nothing else "
                             "crosses ranks -- is gloo")
    parser.add_argument("--queue", choices=("static", "dynamic", "auto"), default="auto",
                        help="static: frame j of the seeded permutation to rank j mod world, inputs resident before the clock.  dynamic: the ranks take "
                             "their next group of frames from ONE queue (an atomic counter on rank 0's TCPStore): frames of unequal cost leave no rank idle "
                             "for more than one group; inputs are built when a frame is taken.  auto: dynamic when the job has more than one rank")
    parser.add_argument("--adjoint-item-slots", type=int, default=0,
                        help="OptimizationConfig.mlp_adjoint_item_slots (vsrd_render_config::adjoint_slots_per_item): 0 = planned (4 for one frame of 1000 rays; "
                             "a batch picks FrameBatch.item_slots).  A frame's trajectory is bit-identical alone and in a batch for equal numbers")
    parser.add_argument("--fp32-mlp", action="store_true", help="the residual MLP's products on the exact-fp32 matrix instruction instead of split bf16")
    parser.add_argument("--views", type=int, default=17)
    parser.add_argument("--instances", type=int, default=8)
    parser.add_argument("--rays", type=int, default=1000)
    parser.add_argument("--samples", type=int, default=100)
    parser.add_argument("--num-steps", type=int, default=3000)
    parser.add_argument("--warmup-steps", type=int, default=1000)
    parser.add_argument("--height", type=int, default=376)
    parser.add_argument("--width", type=int, default=1408)
    parser.add_argument("--out", default="", help="checkpoint directory (default: a temporary one)")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--ranks-share-gpu", action="store_true",
                        help="TEST ONLY: every rank on cuda:0, gloo rendezvous (RCCL refuses two ranks on one device); the line says so")
    parser.add_argument("--fresh-loops", action="store_true",
                        help="A/B: build a new FrameOptimizer (construction, eager warm-up steps, ~10 captures) for every frame, as round 4 did, "
                             "instead of the persistent frame slots whose graphs are captured once at start-up")
    parser.add_argument("--max-restarts", type=int, default=2,
                        help="without torchrun around it, the launcher supervises its ranks: when one dies, all are started again as fresh "
                             "processes, this many times at most; finished frames are skipped (0 with --gpus 1: no supervisor process)")
    parser.add_argument("--stall-timeout", type=float, default=0.0,
                        help="supervisor: seconds without anything new in the checkpoint directory after which the ranks are taken for hung, ended and "
                             "restarted like after a dead rank; 0 = never (the default).  The clock starts with the attempt and starts over when rank 0 "
                             "has finished its set-up (it touches the directory) and with every checkpoint, so the value must exceed both the start-up (library "
                             "build, inputs, graph capture: tens of seconds) and the time of one group of --frame-batch frames")
    parser.add_argument("--attempt", type=int, default=0, help=argparse.SUPPRESS)          # set by the supervisor: restarts so far
    parser.add_argument("--selftest", action="store_true",
                        help="no rendering, no GPU: a frame is a sleep of --selftest-seconds and a small checkpoint (gloo, CPU) -- the supervisor, "
                             "the sharding, the rank -> device map, the skip-if-done guard and the report, for tests/test_launcher.py; the line says so")
    parser.add_argument("--selftest-seconds", type=float, default=0.05)
    parser.add_argument("--selftest-spread", type=float, default=0.0, help="frame j costs --selftest-seconds x (1 + spread x u_j), u_j in [0, 1) fixed per frame")
    parser.add_argument("--selftest-fail", default="", help="RANK:FRAMES -- on attempt 0 that rank dies (exit code 23) once it has finished FRAMES frames")
    parser.add_argument("--selftest-hang", default="", help="RANK:FRAMES -- on attempt 0 that rank hangs (sleeps for ever) once it has finished FRAMES frames")
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parser.parse_args(argv)
    if not 1 <= args.frames_in_flight <= (4 if args.fresh_loops else 8):
        # (round 4, a new loop per frame: five and more frames on one device ended in a memory fault or a segmentation fault inside torch / HIP
        #  within seconds -- construction and eager warm-up steps of that many threads next to a capture.  Frame slots capture nothing once
        #  the workers run; eight is what was tried.)
        raise SystemExit("--frames-in-flight must be 1..8 (1..4 with --fresh-loops)")
    under_launcher = "RANK" in os.environ and "--attempt" not in argv          # (torchrun around us: the supervisor's own ranks carry --attempt)
    if args.procs_per_gpu is None:
        # (a `torchrun --nproc-per-node <GPUs>` command must mean one rank per GPU, whatever this launcher's own default is)
        args.procs_per_gpu = 1
        if under_launcher and not args.selftest and not args.ranks_share_gpu:
            local_world, devices = int(os.environ.get("LOCAL_WORLD_SIZE", "1")), max(torch.cuda.device_count(), 1)
            args.procs_per_gpu = max(local_world // devices, 1)
    if under_launcher and not args.selftest and not args.ranks_share_gpu:
        local_world, devices = int(os.environ.get("LOCAL_WORLD_SIZE", "1")), max(torch.cuda.device_count(), 1)
        if local_world % args.procs_per_gpu or local_world // args.procs_per_gpu > devices:
            raise SystemExit(f"{local_world} local ranks do not map onto {devices} visible GPU(s) with --procs-per-gpu {args.procs_per_gpu}: "
                             "--nproc-per-node must be GPUs x procs-per-gpu")
    if under_launcher and not args.out and int(os.environ.get("TORCHELASTIC_MAX_RESTARTS", "0") or 0) > 0:
        # (a restarted attempt must find the checkpoints of the one before: rank 0's temporary directory would be a new one every time)
        raise SystemExit("torchrun --max-restarts needs --out: the skip-if-done guard works on a checkpoint directory that outlives an attempt")
    if not (1 <= args.procs_per_gpu <= 4 and (args.selftest or args.procs_per_gpu * args.frames_in_flight <= 6 or args.procs_per_gpu == 1)):
        # (four processes with two frames each took 190 s for 32 frames where 4 x 1 and 2 x 3 take 16 and 26: profiles/r05/frames_per_s.txt)
        raise SystemExit("--procs-per-gpu must be 1..4 with at most 6 frames in flight per GPU between them")
    if args.frame_batch is None:
        args.frame_batch = 1 if (args.frames_in_flight > 1 or args.procs_per_gpu > 1 or args.fresh_loops or args.selftest) else DEFAULT_FRAME_BATCH
    if not 1 <= args.frame_batch <= 64:
        raise SystemExit("--frame-batch must be 1..64")
    if args.frame_batch > 1 and (args.frames_in_flight > 1 or args.fresh_loops):
        raise SystemExit("--frame-batch B > 1 is ONE thread stepping B frames together: use it with --frames-in-flight 1 and without --fresh-loops")
    if args.ranks_share_gpu:
        args.procs_per_gpu = 1               # (the test flag puts every rank on cuda:0 by itself)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # __graft_entry__.build lives at the repo root
    if root not in sys.path:
        sys.path.insert(0, root)
    if "RANK" not in os.environ and (args.gpus * args.procs_per_gpu > 1 or args.max_restarts > 0):
        return _supervise(argv, args)
    return _rank_main(args)


def _rank_main(args):
    import json
    import sys
    import tempfile
    import time
    use_gpu = not args.selftest
    procs = args.procs_per_gpu
    # RCCL only when every rank has a device of its own
    share = args.ranks_share_gpu or procs > 1
    if args.ranks_share_gpu:
        os.environ["LOCAL_RANK"] = "0"
    local_device = int(os.environ.get("LOCAL_RANK", "0")) // procs
    if use_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("vsrd_amd.launcher optimises frames on HIP devices: no GPU visible, and vsrd_amd has no CPU fallback")
        rank, world, device = init_process_group(backend="gloo" if share else None, ranks_per_device=procs)
        import __graft_entry__
        if rank == 0:
            __graft_entry__.build()
        barrier()
    else:
        rank, world, device = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), None
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    total = args.frames or (2 * max(args.frames_in_flight, args.frame_batch) * world if use_gpu else 4 * world)
    args.frame_batch = effective_frame_batch(args.frame_batch, total, world)
    out_dir = args.out or (tempfile.mkdtemp(prefix="vsrd_frames_") if rank == 0 else None)
    dynamic = args.queue == "dynamic" or (args.queue == "auto" and world > 1)
    store = None
    store_port = 0
    if dynamic:                              # rank 0 serves the queue's counter; the port travels in the manifest
        host = os.environ.get("MASTER_ADDR", "127.0.0.1")
        if rank == 0:
            store = dist.TCPStore(host, 0, world, is_master=True, wait_for_workers=False)
            store_port = store.port
    manifest = broadcast_manifest(dict(frames=list(range(total)), seed=args.seed, out=out_dir, store_port=store_port, attempt=args.attempt) if rank == 0 else None)
    if not manifest["out"]:
        raise SystemExit("no checkpoint directory (--out; the supervisor supplies one)")
    if dynamic and rank != 0:
        store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(manifest["store_port"]), world, is_master=False)
    queue = FrameQueue(manifest["frames"], rank, world, seed=manifest["seed"], store=store, key=f"vsrd_next_frame_{manifest['attempt']}")
    ordered(lambda r: print(f"[rank {r}/{world}] " + (f"selftest, device {local_device}" if args.selftest else str(device)) +
                            (": frames from the job's queue" if dynamic else f": frames {queue.mine}"), file=sys.stderr, flush=True))
    work = _SleepWork(args, manifest, rank, queue) if args.selftest else _RenderWork(args, manifest, device, queue)
    work.prepare()
    if rank == 0:                            # the supervisor's stall clock starts over at the first fence: the directory's own time stamp
        os.makedirs(manifest["out"], exist_ok=True)
        os.utime(manifest["out"], None)

    def fence():
        barrier()
        if use_gpu:
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    report = work.run()
    own = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    losses = report["final_losses"]
    report.update(rank=rank, device=local_device, own_seconds=own, elapsed_seconds=elapsed, mean_final_loss=(sum(losses.values()) / len(losses)) if losses else None)
    gathered = [report]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, report)
    if rank == 0:
        slowest = max(r["elapsed_seconds"] for r in gathered)
        frames_done = sum(r["frames"] for r in gathered)
        n_gpus = len(gathered) if args.ranks_share_gpu else max(len(gathered) // procs, 1)       # (--ranks-share-gpu: the test's line counts ranks, as before)
        if args.selftest:
            line = {"metric": "launcher selftest: no rendering, NOT a measurement", "value": 0.0, "unit": "none", "n_gpus": n_gpus, "ranks": len(gathered),
                    "procs_per_gpu": procs, "rank_devices": [r["device"] for r in gathered], "frames": frames_done,
                    "frames_skipped_as_done": sum(r["skipped"] for r in gathered), "per_rank_frames": [r["frames"] for r in gathered],
                    "restarts": args.attempt, "max_restarts": args.max_restarts, "queue": "dynamic" if dynamic else "static",
                    "per_rank_seconds": [r["own_seconds"] for r in gathered],
                    "rank_finish_spread_seconds": max(r["own_seconds"] for r in gathered) - min(r["own_seconds"] for r in gathered),
                    "config": {"workload": f"sleep({args.selftest_seconds} s x (1 + {args.selftest_spread} u)) per frame on gloo / CPU", "checkpoints": manifest["out"]}}
        else:
            line = {
                "metric": "optimised target frames/s, whole job (reference: about 15 minutes per frame on a V100, README.md:128)",
                "value": frames_done / slowest, "unit": "frames/s", "n_gpus": n_gpus, "higher_is_better": True, "scaling": "weak",
                "frames": frames_done, "frames_skipped_as_done": sum(r["skipped"] for r in gathered), "seconds": slowest,
                "restarts": args.attempt, "max_restarts": args.max_restarts,     # (restarts > 0: `seconds` and `frames` are the last attempt's)
                "frames_per_s_per_gpu": frames_done / slowest / n_gpus,
                "ranks": len(gathered), "procs_per_gpu": procs, "frames_in_flight_per_process": args.frames_in_flight, "rank_devices": [r["device"] for r in gathered],
                "frame_batch": args.frame_batch,        # frames one process steps together, one launch of every kernel for all of them (optimization.FrameBatch)
                "control_plane": "gloo" if share else "RCCL",
                "queue": "dynamic (TCPStore counter on rank 0; inputs built when a frame is taken, inside the clock)" if dynamic else
                         "static (frame j of the seeded permutation to rank j mod world; inputs resident before the clock)",
                # the spread of the ranks' own finishing times: what a static split of unequal frames costs the job, what the queue bounds by one group
                "rank_finish_spread_seconds": max(r["own_seconds"] for r in gathered) - min(r["own_seconds"] for r in gathered),
                "mlp_products": "exact fp32 (v_mfma_f32_16x16x4_f32)" if args.fp32_mlp else "split bf16 (v_mfma_f32_16x16x32_bf16 on hi/lo parts, fp32 accumulation)",
                "per_rank_seconds": [r["own_seconds"] for r in gathered], "per_rank_frames": [r["frames"] for r in gathered],
                "seconds_per_frame_per_rank": [r["own_seconds"] / r["frames"] if r["frames"] else None for r in gathered],
                "mean_final_loss": [r["mean_final_loss"] for r in gathered], "data": "synthetic", "dtype": "f32",
                "final_loss_per_frame": {str(f): v for r in gathered for f, v in r["final_losses"].items()},
                # host time per frame spent capturing hipGraphs (optimization._CaptureGate): what the other frames in flight wait for
                "capture_seconds_per_frame": [r["gate_capture_seconds"] / r["frames"] if r["frames"] else None for r in gathered],
                # persistent frame slots: built and captured once per process BEFORE the clock (a job of thousands of frames pays it once)
                "frame_slots": not args.fresh_loops, "slot_setup_seconds": [r["slot_setup_seconds"] for r in gathered],
                "graphs_per_slot": [r["graphs_per_slot"] for r in gathered],
                # frames a slot could not take (their importance weights do not suit the sampling table its graphs draw from: a loop of their own, race sampler)
                # where a batched rank's time went, in seconds over its frames: building inputs (dynamic queue only), resetting rows, the steps, checkpoints
                "phase_seconds": [r.get("phase_seconds") for r in gathered],
                "frames_outside_slots": sorted(f for r in gathered for f in r["frames_outside_slots"]),
                "unsuitable_founders": sorted(f for r in gathered for f in r["unsuitable_founders"]),
                # frames in which some ray draw ran out of picks / overflowed (read at the end of each frame; RuntimeWarning on stderr)
                "frames_with_unhealthy_draws": sorted(f for r in gathered for f in r["frames_with_unhealthy_draws"]),
                "config": {"workload": f"{args.num_steps} optimisation steps per frame ({args.warmup_steps} box-only + {args.num_steps - args.warmup_steps} with the "
                                       f"residual MLP), {args.rays} rays x {args.samples} samples per step, {args.views} views of {args.height}x{args.width}, "
                                       f"{args.instances} instances, " +
                                       (f"FrameBatch: {args.frame_batch} frames per launch chain, {procs} rank process(es) per GPU (graphs captured once at start-up)"
                                        if args.frame_batch > 1 else
                                        f"FrameOptimizer(graph=True), {procs} rank process(es) per GPU x {args.frames_in_flight} frame(s) in flight each"
                                        + (" (a new loop per frame)" if args.fresh_loops else " (persistent frame slots: graphs captured once at start-up)")),
                           "parallelism": f"frames sharded over {len(gathered)} rank(s) on {n_gpus} GPU(s), no data-path collective; "
                                          + ("gloo (RCCL refuses two ranks on one device)" if share else "RCCL") + ": barriers, manifest broadcast, gather of the report",
                           "checkpoints": manifest["out"]},
            }
            if args.ranks_share_gpu:
                line["metric"] = "ranks share ONE GPU (--ranks-share-gpu): a test of the multi-rank path, NOT a scaling measurement; " + line["metric"]
        print(json.dumps(line), flush=True)
    barrier()
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
