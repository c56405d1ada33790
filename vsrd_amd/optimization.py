"""One target frame's optimisation loop: the hot path of scripts/main.py:323-865 assembled from the device pieces.

    detector() -> multi-view projection (HIP) -> Hungarian matching (HIP, on the device) -> projection losses
    -> schedules -> field block -> ray sampling -> fused two-pass render (HIP) -> silhouette (+ eikonal) loss
    -> backward (HIP adjoint kernels + torch for the tiny decode) -> Adam -> ExponentialLR

With ``graph=True`` the whole step is captured once per phase in a hipGraph (torch.cuda.CUDAGraph) and replayed: at 1000
rays the step is ~360 small launches and host-bound, and nothing in it needs the host any more -- the matching runs on the
device, and everything that changes from step to step (schedules, Philox counter, Adam step, learning rates) lives in
device memory that the kernels read (vsrd_render_config::device_schedule / device_stream_offset).

Frames are independent problems (README.md:128): multi-GPU runs shard *frames* over ranks (vsrd_amd/launcher.py) and
never exchange gradients.
"""
from dataclasses import dataclass, field as dataclass_field
from typing import Optional, Sequence

import math
import threading
import time

import torch

from . import _lib, fields, losses, models, operations, rendering


class _CaptureGate:
    """Stream capture against everything else, over the frames that one rank optimises concurrently (launcher.run_frames).  Capture is a
    process-wide mode, and every collision seen in round 4 -- each one a dead rank -- had a capture on one side: torch refuses a replay
    during another thread's capture ("Cannot prepare for replay during capturing stage": every graph registers the process-wide default
    generator); ROCm's ~CUDAGraph synchronises the device, which HIP refuses during a capture; HIP refuses, now and then, even a STREAM
    synchronisation (`.item()`, `.cpu()`, `stream.synchronize()`, `torch.nonzero`, a module's `.to(device)`) of one thread while another
    captures -- or lets it through and breaks the capture instead ("capturing stream has unjoined work"; a graph from such a capture is
    the likely cause of the one segmentation fault inside hipGraphLaunch).  So:
      ``with gate.capture():``  a capture -- alone: nothing else of any frame calls into HIP meanwhile (a waiting capture goes first);
      ``with gate:`` / ``acquire`` / ``release`` / ``with gate.replaying():``  everything else that calls into HIP from a frame's thread --
      construction, eager warm-up steps, replay launches, graph destruction, host synchronisations and copies: any number at once, never
      during a capture.  Nestable, also inside the capturing thread's own capture.
    ``acquire(blocking=False)``: the garbage collector's path; it never waits, and inside the capturing thread it is refused.
    (Two stricter gates came first: with EVERY call taking turns, replay launches included, the gain of several frames in flight was
    gone, 0.85 -> 0.75 frames/s -- a launch blocks for milliseconds behind the previous launch of the same graph; with only the launches
    shared, a frame still held the gate exclusively for 0.22-0.25 s of its 1.33 s.)"""

    def __init__(self):
        # (a re-entrant lock under the condition: the collector may run a FrameOptimizer's __del__ in a thread that is inside one of these
        #  few-line critical sections itself)
        self._state = threading.Condition(threading.RLock())
        self._capturer, self._capture_depth, self._waiting_captures, self._others = None, 0, 0, 0
        self._local = threading.local()
        self.capture_seconds, self._since = 0.0, 0.0            # total time spent capturing (launcher.main reports it per frame)

    def _depth(self):
        return getattr(self._local, "depth", 0)

    def acquire(self, blocking=True):
        me = threading.get_ident()
        with self._state:
            if self._capturer == me:
                if not blocking:             # the collector inside the capturing thread: not now
                    return False
                self._local.depth = self._depth() + 1           # (inside its own capture the capturing thread passes)
                return True
            if self._depth() > 0:            # nested: a capture that waits for THIS thread to leave must not stop it on the way out
                self._local.depth += 1
                return True
            while self._capturer is not None or self._waiting_captures:
                if not blocking:
                    return False
                self._state.wait()
            self._others += 1
            self._local.depth = 1
            return True

    def release(self):
        with self._state:
            self._local.depth -= 1
            if self._local.depth == 0 and self._capturer != threading.get_ident():
                self._others -= 1
                if not self._others:
                    self._state.notify_all()

    def __enter__(self):
        self.acquire()
        return self

    def __exit__(self, *exc):
        self.release()

    def replaying(self):
        return self

    def capture(self):
        return _CaptureTurn(self)


class _CaptureTurn:
    def __init__(self, gate):
        self.gate = gate

    def __enter__(self):
        gate, me = self.gate, threading.get_ident()
        with gate._state:
            if gate._capturer == me:
                gate._capture_depth += 1
                return
            if gate._depth() > 0:
                raise RuntimeError("a capture cannot start inside a section that holds the capture gate (it would wait for itself)")
            gate._waiting_captures += 1
            try:
                while gate._capturer is not None or gate._others:
                    gate._state.wait()
            finally:
                gate._waiting_captures -= 1
            gate._capturer, gate._capture_depth = me, 1
            gate._since = time.perf_counter()

    def __exit__(self, *exc):
        gate = self.gate
        with gate._state:
            gate._capture_depth -= 1
            if gate._capture_depth == 0:
                gate.capture_seconds += time.perf_counter() - gate._since
                gate._capturer = None
                gate._state.notify_all()


_capture_lock = _CaptureGate()


class _InitialDraw:
    """Seeding torch's host generator and drawing a frame's initial parameters from it: one frame at a time, and on ONE host thread -- the
    draw is a handful of tiny CPU operations (the largest: weight_norm over [1617, 256]), and on a 256-core box torch's intra-op pool takes
    100-200 ms to run them on all cores where one thread takes 2 ms (tools/reset_timers.py: a frame slot's reset() 100 ms -> 4.4 ms; the
    launcher lost 0.11 s per frame to it).  The values drawn do not depend on the thread count."""

    def __init__(self):
        self._lock = threading.Lock()

    def __enter__(self):
        self._lock.acquire()
        self._threads = torch.get_num_threads()
        if self._threads > 1:
            torch.set_num_threads(1)
        return self

    def __exit__(self, *exc):
        if self._threads > 1:
            torch.set_num_threads(self._threads)
        self._lock.release()


_initial_draw = _InitialDraw()


def exclusive_device_access():
    """``with exclusive_device_access(): ...`` -- no frame of this process captures meanwhile: for the host synchronisations and
    device-to-host copies (`.item()`, `stream.synchronize()`, checkpoints) of a frame that is optimised next to others."""
    return _capture_lock


_graveyard = []          # graphs whose owner went away while somebody captured: destroyed by the next holder of the lock


def _destroy_graphs(graphs, wait=True):
    """Destroy captured graphs where nobody captures.  On ROCm ``torch.cuda.CUDAGraph``'s destructor synchronises the DEVICE, which HIP
    refuses while any thread of the process is capturing ("operation not permitted when stream is capturing": the frames/s launcher
    lost a rank about once in five runs when one frame ended during the other frame's capture).  ``wait=False`` (garbage collection:
    the collector may run inside the capturing thread itself) parks the graphs instead of blocking."""
    if not graphs and not _graveyard:
        return
    if _capture_lock.acquire(blocking=wait):
        try:
            graphs.clear()
            _graveyard.clear()
        finally:
            _capture_lock.release()
    else:
        _graveyard.append(list(graphs.values()))
        graphs.clear()


class UnsuitableFrameError(ValueError):
    """A frame that cannot be served by a persistent frame slot (its importance weights do not suit the slot's table sampler)."""


@dataclass
class FrameInputs:
    """What scripts/main.py:106-316 prepares for one target frame; view 0 is the target view."""
    image_size: Sequence[int]               # (H, W)
    intrinsic_matrices: torch.Tensor        # [V,3,3]
    extrinsic_matrices: torch.Tensor        # [V,4,4] (target-camera frame -> view camera frame)
    soft_masks: torch.Tensor                # [V,H,W,N] re-indexed to the target's instance order (main.py:204-265)
    boxes_2d: torch.Tensor                  # [V,N,2,2] ground-truth 2-D boxes, zeros where invisible
    visible_masks: torch.Tensor             # [V,N] bool


@dataclass
class OptimizationConfig:
    """configs/kitti_360/vsrd/*/config.json:120-127,166-238."""
    num_steps: int = 3000
    warmup_steps: int = 1000
    num_rays: int = 1000
    num_samples: int = 100
    distance_range: Sequence[float] = (0.0, 100.0)
    max_sdf_union_temperature: float = 1.0
    min_sdf_union_temperature: float = 0.1
    max_sdf_std_deviation: float = 1.0
    min_sdf_std_deviation: float = 0.1
    learning_rate: float = 1.0e-2
    embedding_learning_rate: float = 1.0e-3       # config.json:193-196
    hypernetwork_learning_rate: float = 1.0e-4    # config.json:197-200
    lr_gamma: float = 0.01 ** (1.0 / 3000.0)
    loss_weights: dict = dataclass_field(default_factory=lambda: dict(losses.LOSS_WEIGHTS))
    seed: int = 0
    skip_exact_misses: bool = True
    # the residual phase's per-instance MLP on split-bf16 matrix products (VSRD_FLAG_MLP_SPLIT_BF16; csrc/residual.h): the same tests within the
    # same tolerances as the exact-fp32 products (against the reference's goldens: labels within 1.1e-6 (tolerance 1e-4), gradients within 4.2e-4 of the largest entry (tolerance 5e-3; exact fp32: 1.5e-4)), the
    # residual step 8 % faster.  False: the exact-fp32 matrix instruction (the C ABI's own default).  The launcher's line names the form (`mlp_products`).
    mlp_split_bf16: bool = True
    # vsrd_render_config::adjoint_slots_per_item of the residual step (0: the library plans it from one frame's launch: 4 slots at 1000 rays).
    # A FrameBatch whose config leaves this at 0 picks the number for its B-fold item count (FrameBatch.item_slots: 16 for 8 x 1000 rays);
    # a frame walks bit-identical trajectories alone and in a batch when both use the same number.
    mlp_adjoint_item_slots: int = 0
    # The initial parameters (a detector's embeddings, the hypernetwork) are drawn from torch's global generator when a loop is built or reset.
    # None: whatever state that generator is in (the reference seeds it once per rank, scripts/main.py:67-74, so a frame's start depends on the
    # frames its rank optimised before).  An integer: the generator is seeded with it right before the draw -- a frame's result then depends on
    # the frame alone, not on the rank, process or slot that runs it (vsrd_amd.launcher passes seed and frame number).
    init_seed: Optional[int] = None


class FrameArena:
    """Device memory of a BATCH of frames (include/vsrd_hip.h, "frame batches"): one row of ``row_bytes`` per frame, and every buffer a
    frame's kernels touch sits at the SAME offset of its frame's row -- frame f's copy of a buffer is ``stride`` bytes x f behind frame 0's,
    which is all a batched launch needs to know.  The rows are filled by the frames' own constructors (``FrameRow.new``): same calls in the
    same order give the same offsets, which ``FrameBatch`` verifies."""

    ALIGNMENT = 256

    def __init__(self, num_frames, row_bytes, device):
        self.stride = (int(row_bytes) + self.ALIGNMENT - 1) // self.ALIGNMENT * self.ALIGNMENT
        self.buffer = torch.empty(int(num_frames), self.stride, dtype=torch.uint8, device=device)
        self.rows = [FrameRow(self, f) for f in range(int(num_frames))]


class FrameRow:
    """Bump allocator over one frame's row of a ``FrameArena``."""

    def __init__(self, arena, index):
        self.arena, self.index = arena, index
        self.cursor = 0
        self.layout = []                 # (offset, nbytes) of every buffer, in order: equal across the rows of an arena

    def bytes(self, nbytes):
        nbytes = int(nbytes)
        offset = self.cursor
        end = offset + nbytes
        if end > self.arena.stride:
            raise RuntimeError(f"frame arena row of {self.arena.stride} bytes exhausted ({end} needed): FrameBatch._row_bytes is too small")
        self.cursor = (end + FrameArena.ALIGNMENT - 1) // FrameArena.ALIGNMENT * FrameArena.ALIGNMENT
        self.layout.append((offset, nbytes))
        return self.arena.buffer[self.index, offset:end]

    def new(self, shape, dtype=torch.float32, fill=None):
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)))
        count = 1
        for x in shape:
            count *= x
        view = self.bytes(count * torch.empty((), dtype=dtype).element_size()).view(dtype).view(shape)
        if fill is not None:
            view.fill_(fill)
        return view

    def adopt(self, tensor):
        """A row-resident copy of `tensor` (same shape, dtype and values)."""
        view = self.new(tensor.shape, tensor.dtype)
        view.copy_(tensor)
        return view


def adam_state_tensors(optimizer, parameter, group, row=None):
    """`parameter`'s torch.optim.Adam(capturable=True) state as the C ABI's pointer block, created now if the optimiser has not
    stepped yet (torch creates it lazily): the kernels update the moments, the counter and the parameter in place.  `row`: the frame's
    row of a batch's arena (FrameRow) -- the state is created there."""
    state = optimizer.state[parameter]
    if not state:
        if row is not None:
            state["step"] = row.new((), torch.float32, fill=0.0)
            state["exp_avg"], state["exp_avg_sq"] = row.new(parameter.shape, torch.float32, fill=0.0), row.new(parameter.shape, torch.float32, fill=0.0)
        else:
            state["step"] = torch.zeros((), dtype=torch.float32, device=parameter.device)
            state["exp_avg"], state["exp_avg_sq"] = torch.zeros_like(parameter), torch.zeros_like(parameter)
    return _lib.AdamTensors(parameter.data_ptr(), state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(), state["step"].data_ptr(),
                            group["lr"].data_ptr())


def hypernetwork_tensors(hyper_distance_field, embeddings, optimizer, lr_gamma, row=None):
    """include/vsrd_hip.h::vsrd_hypernetwork over the torch module's own parameters and the optimiser's own state.  `optimizer`'s
    groups: the one holding `embeddings` and the one holding the hypernetwork, learning rates as device tensors."""
    def group_of(p):
        return next(g for g in optimizer.param_groups if any(q is p for q in g["params"]))
    net = _lib.Hypernetwork()
    net.num_instances, net.num_outputs = int(embeddings.shape[-2]), _lib.MLP_WEIGHTS
    betas = group_of(embeddings)["betas"]
    net.beta1, net.beta2, net.adam_epsilon, net.lr_gamma = float(betas[0]), float(betas[1]), float(group_of(embeddings)["eps"]), float(lr_gamma)
    net.embeddings = adam_state_tensors(optimizer, embeddings, group_of(embeddings), row)
    blocks = list(hyper_distance_field.hypernetwork)
    if len(blocks) != _lib.HYPER_LAYERS or embeddings.shape[-1] != 256 or not all(p.is_contiguous() for p in hyper_distance_field.parameters()):
        raise ValueError("csrc/hypernetwork.h is built for config.json:143-156: 256-d embeddings, four hidden blocks of 256")
    for l, block in enumerate(blocks):
        linear = block[0]
        net.weight_v[l], net.weight_g[l], net.bias[l] = (adam_state_tensors(optimizer, p, group_of(p), row) for p in (linear.weight_v, linear.weight_g, linear.bias))
        if l + 1 < len(blocks):
            net.norm_weight[l] = adam_state_tensors(optimizer, block[1].weight, group_of(block[1].weight), row)
            net.norm_bias[l] = adam_state_tensors(optimizer, block[1].bias, group_of(block[1].bias), row)
    return net


class FrameOptimizer:
    """``graph=True``: the step is captured in a hipGraph and replayed.  ``fused_glue`` (default: on in graph mode): everything around
    the render launch that concerns the boxes -- decode, projection, matching, projection losses and their gradients, schedules,
    the chain rule through the decode, Adam and the learning-rate decay -- runs as two single-workgroup HIP kernels
    (csrc/frame_step.h) instead of ~330 torch element-wise launches, and the render kernel gathers its rays from the
    frame-resident tensors by index; in the residual phase the hypernetwork, its backward and its Adam are csrc/hypernetwork.h
    (``fused_hypernetwork = False`` keeps them with torch: the A/B reference)."""

    def __init__(self, inputs: FrameInputs, config: OptimizationConfig, device, graph=False, fused_glue=None, persistent=False, row=None):
        """``persistent``: the loop will be given other frames of the same shape through ``reset()`` (a frame SLOT, launcher.py): it then
        owns every buffer a captured graph holds the address of -- in particular a copy of the soft masks instead of the caller's tensor.
        ``row`` (a ``FrameRow``): the loop is one frame of a ``FrameBatch`` and keeps EVERY device buffer its kernels touch -- parameters,
        Adam's state, rates, counters, rays, masks, sampling table, scratch -- in that row of the batch's arena."""
        self._graphs = {}
        self.persistent = bool(persistent)
        self._row = row
        if row is not None and not (graph and persistent and fused_glue in (None, True)):
            raise ValueError("a frame of a batch is a persistent graph-mode loop with the fused glue")
        # the whole construction stays out of other frames' captures: it copies modules to the device, reads ranges back (.item()), runs
        # nonzero -- host synchronisations that, next to ANOTHER frame's capture, fail now and then or break that capture ("capturing
        # stream has unjoined work" in the other thread: one 36-frame run in fourteen)
        with _capture_lock:
            self._construct(inputs, config, device, graph, fused_glue)

    def _construct(self, inputs, config, device, graph, fused_glue):
        self.inputs, self.config, self.device = inputs, config, torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:      # the scratch buffers are keyed by device: one spelling of it
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.graph = bool(graph)
        self.fused_glue = self.graph if fused_glue is None else bool(fused_glue)
        if self.fused_glue and not self.graph:
            raise ValueError("fused_glue keeps the learning rates and Adam's counters on the device: it needs graph=True")
        V, H, W, N = inputs.soft_masks.shape
        self.num_views, self.num_instances = V, N
        with _initial_draw:
            if config.init_seed is not None:
                torch.default_generator.manual_seed(int(config.init_seed))      # (the CPU generator: the modules are built on the host; the device generators, which captured graphs register, stay untouched)
            detector = models.BoxParameters3D(1, N)
            # config.json:143-156: per-instance residual MLP 48->16->16->16->16->1 generated from 256-d embeddings
            field = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256])
        self.detector, self.hyper_distance_field = detector.to(self.device), field.to(self.device)
        if self._row is not None:          # the parameters move into the frame's row (module order: the same in every frame of the batch)
            for p in [*self.detector.parameters(), *self.hyper_distance_field.parameters()]:
                p.data = self._row.adopt(p.data)
        def rate(value):   # graph mode: learning rates are device tensors decayed in place inside the captured step
            if self._row is not None:
                return self._row.new((), torch.float32, fill=value)
            return torch.tensor(value, dtype=torch.float32, device=self.device) if self.graph else value
        groups = [dict(params=[p], lr=rate(config.learning_rate)) for p in (self.detector.locations, self.detector.dimensions, self.detector.orientations)]
        groups.append(dict(params=[self.detector.embeddings], lr=rate(config.embedding_learning_rate)))
        groups.append(dict(params=list(self.hyper_distance_field.parameters()), lr=rate(config.hypernetwork_learning_rate)))
        self.optimizer = torch.optim.Adam(groups, lr=rate(config.learning_rate), capturable=self.graph)
        self.scheduler = None if self.graph else torch.optim.lr_scheduler.ExponentialLR(self.optimizer, gamma=config.lr_gamma)
        # per-step scalars on the device (graph mode): step index = Philox counter, (temperature, std, cosine_ratio)
        self.step_tensor = self._new(1, torch.int64, fill=0)
        self.schedule = self._new(3, torch.float32, fill=1.0)
        # this frame's own scratch (frames may be optimised concurrently on other streams); freed with the optimizer.  Graph mode
        # sizes it for the residual phase up front: a captured graph holds the buffer's address
        self.workspace = rendering.Workspace(allocate=None if self._row is None else self._row.bytes)
        if self.graph:
            self.workspace.keep_outgrown = True          # a captured graph holds the buffer's address: an outgrown buffer must outlive it
            self.workspace.reserve(self.device, N, residual=True, step_shape=(config.num_samples, config.num_rays))
        self._graphs = {}
        self._capture_stream = None
        self._eager_graph_steps = {}
        self._prepare_rays(inputs, config, H, W, N)
        self.pixels_per_view = H * W
        self.step_index = 0
        if self.fused_glue:
            self._init_fused_glue()

    def _new(self, shape, dtype=torch.float32, fill=None):
        """A device buffer of this loop: in the frame's row of the batch's arena when there is one."""
        if self._row is not None:
            return self._row.new(shape, dtype, fill)
        if fill is None:
            return torch.empty(shape, dtype=dtype, device=self.device)
        return torch.full(shape if isinstance(shape, (tuple, list)) else (shape,), fill, dtype=dtype, device=self.device)

    def _own(self, tensor):
        """`tensor` as a buffer of this loop: a copy in the frame's row of the batch's arena, or the tensor itself."""
        return tensor if self._row is None else self._row.adopt(tensor)

    def _prepare_rays(self, inputs, config, H, W, N):
        # rays of every view, once per frame (main.py:267-296)
        cam, dirs = rendering.ray_casting((H, W), inputs.intrinsic_matrices, inputs.extrinsic_matrices)
        self.camera_positions = self._own(cam)                              # [V,3]
        self.ray_directions = self._own(dirs.reshape(-1, 3).contiguous())   # [V*H*W,3]
        self.flat_masks = inputs.soft_masks.reshape(-1, N)
        if self._row is not None:
            self.flat_masks = self._row.adopt(self.flat_masks.to(torch.float32))
        elif self.persistent:                                               # (a slot's graphs read the masks through this address for every later frame)
            self.flat_masks = self.flat_masks.to(torch.float32).clone(memory_format=torch.contiguous_format)
        self.sampling_weights = self._own(self.flat_masks.max(dim=-1).values.to(torch.float32).contiguous())          # main.py:620-624
        self.ray_table = self.ray_remap = None
        if not self._prepare_sampler(config):
            raise UnsuitableFrameError("this frame's importance weights do not suit the table sampler (too much of the weight in fewer pixels "
                                       "than a draw needs; rendering.RayTable.suits): it cannot found a frame slot -- optimise it in a loop of "
                                       "its own (persistent=False), which keeps the per-step race sampler")

    def _prepare_sampler(self, config):
        """What depends on the frame's importance weights (fixed for the frame).  Graph mode draws the rays on the device: the sampler's
        table is built here, once per frame, over ALL V*H*W weights -- entries without weight have zero width and are never drawn, and the
        table's size and entry count depend on the frame's SHAPE only, so a slot's graphs (which hold the table's address and count) serve
        every frame of that shape; a step's draw is one launch (csrc/ray_sampling.h).  Frames whose weight sits in fewer entries than a
        draw needs keep the per-step exponential race over the positive pixels (`ray_table` None).  False: such a frame cannot take over a
        slot whose graphs were captured with the table."""
        positive = int((self.sampling_weights > 0).sum())
        # torch.multinomial(replacement=False) raises when fewer categories than samples have weight; the device sampler would pad
        # its output with -1 instead (ray_sampling.h), silently: the weights are fixed for the frame, so check once here
        if positive < config.num_rays:
            raise ValueError(f"only {positive} pixels have a positive soft mask, fewer than num_rays = {config.num_rays} "
                             "(torch.multinomial without replacement raises in the reference as well)")
        if not (self.graph and self.device.type == "cuda"):
            return True
        had_table = self.ray_table is not None
        if had_table:
            self.ray_table.rebuild(self.sampling_weights)
        table = self.ray_table if had_table else rendering.RayTable(self.sampling_weights, allocate=None if self._row is None else self._row.bytes)
        if table.suits(config.num_rays):
            self.ray_table = table
            return True
        if had_table or self.persistent:
            # a slot's graphs are captured with ONE sampler: a frame that does not suit the table cannot take over a slot (nor found one:
            # __init__ raises UnsuitableFrameError) -- the caller optimises it in a loop of its own, with the race sampler
            return False
        # the race sampler only visits the pixels that can be drawn at all
        self.positive_pixels = torch.nonzero(self.sampling_weights > 0).flatten()
        self.positive_weights = self.sampling_weights[self.positive_pixels].contiguous()
        return True

    # ---- fused glue (csrc/frame_step.h) ---------------------------------------------------------------------------------
    def _init_fused_glue(self):
        cfg, inp, dev = self.config, self.inputs, self.device
        V, N = self.num_views, self.num_instances
        lib = _lib.load()
        det = self.detector
        w = cfg.loss_weights
        frame = _lib.FrameConfig()
        frame.num_boxes, frame.num_views = N, V
        frame.height, frame.width, frame.epsilon = float(inp.image_size[0]), float(inp.image_size[1]), 1.0e-6
        for j in range(3):
            frame.location_lo[j], frame.location_hi[j] = float(det.location_range[0, j]), float(det.location_range[1, j])
            frame.dimension_lo[j], frame.dimension_hi[j] = float(det.dimension_range[0, j]), float(det.dimension_range[1, j])
        frame.num_steps = cfg.num_steps
        frame.max_temperature, frame.min_temperature = cfg.max_sdf_union_temperature, cfg.min_sdf_union_temperature
        frame.max_std, frame.min_std = cfg.max_sdf_std_deviation, cfg.min_sdf_std_deviation
        frame.weight_iou, frame.weight_l1, frame.weight_silhouette = w["iou_projection_loss"], w["l1_projection_loss"], w["silhouette_loss"]
        betas, eps = self.optimizer.param_groups[0]["betas"], self.optimizer.param_groups[0]["eps"]      # the box tensors' Adam constants
        frame.beta1, frame.beta2, frame.adam_epsilon, frame.lr_gamma = float(betas[0]), float(betas[1]), float(eps), cfg.lr_gamma
        self._frame = frame
        f32 = dict(dtype=torch.float32, device=dev)
        b = self._glue = {}
        b["extrinsics"] = inp.extrinsic_matrices.to(**f32).reshape(V, 16).contiguous()
        b["intrinsics"] = inp.intrinsic_matrices.to(**f32).reshape(V, 9).contiguous()
        b["gt_boxes"] = inp.boxes_2d.to(**f32).reshape(V, N, 4).contiguous()
        b["visible"] = inp.visible_masks.to(device=dev, dtype=torch.uint8).contiguous()
        if self.persistent:       # a slot OWNS what reset() overwrites: float32 contiguous inputs would otherwise be these very tensors -- the
            for name in ("extrinsics", "intrinsics", "gt_boxes", "visible"):      # caller's, and every other slot's built from the same frame
                b[name] = b[name].clone() if self._row is None else self._row.adopt(b[name])
        new = self._new            # (a frame of a batch: everything below lives in the frame's row of the arena)
        b["scratch"] = new(lib.vsrd_frame_scratch_bytes(V, N), torch.uint8)
        b["instances"] = new((N, 16), fill=0.0)
        b["pd_indices"], b["gt_indices"] = new(N, torch.int64, fill=0), new(N, torch.int64, fill=0)
        b["target_columns"] = new(N, torch.int32, fill=0)
        b["instance_weights"] = new(N, fill=1.0)
        b["projection_losses"], b["render_losses"] = new(2, fill=0.0), new(2, fill=0.0)
        b["grad_raw"], b["raw_gradients"] = new((N, 8), fill=0.0), new((N, 8), fill=0.0)
        b["grad_instances"], b["grad_mlp"] = new((N, 16), fill=0.0), new((N, _lib.MLP_WEIGHTS), fill=0.0)
        b["record"] = new(5, fill=0.0)
        b["masks"] = self.flat_masks.to(**f32).contiguous()        # (persistent: _prepare_rays made this the slot's own copy)
        self.fused_hypernetwork = True
        self.rebind()
        b["hyper_workspace"] = new(lib.vsrd_hypernetwork_workspace_bytes(N), torch.uint8)
        # the three things a step needs before its render launch -- its rays, the boxes' side (prologue) and the generated MLP weights --
        # do not depend on each other: they run as three branches (two side streams, forked from and joined to the step's stream), and
        # so do the two things after it (epilogue, hypernetwork backward).  Captured, they become parallel branches of the hipGraph
        self._branches = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        b["picks"] = new(cfg.num_rays, torch.int64, fill=0)
        b["ray_indices"] = new(cfg.num_rays, torch.int64, fill=0)
        b["mlp_weights"], b["mlp_centred"] = new((N, _lib.MLP_WEIGHTS), fill=0.0), new((N, _lib.MLP_WEIGHTS), fill=0.0)

    # ---- the tensors whose addresses the kernels and the captured graphs hold ----------------------------------------------------
    def _bound_tensors(self):
        """Every tensor of the MODULES and the OPTIMISER whose address is handed to a kernel as a raw pointer (adam_state_tensors,
        hypernetwork_tensors, the prologue's parameter pointers): parameters, Adam's moments and counters, the learning rates.  (The
        frame's own buffers -- self._glue, step_tensor, schedule -- are created once and never replaced.)"""
        tensors = []
        for group, params in zip(self.optimizer.param_groups, self._module_parameters()):
            tensors.append(group["lr"])
            for p in params:
                state = self.optimizer.state.get(p, {})
                tensors += [p, state.get("exp_avg"), state.get("exp_avg_sq"), state.get("step")]
        return tensors

    def _module_parameters(self):
        """The modules' CURRENT parameters in the order of the optimiser's groups (__init__)."""
        det = self.detector
        return [[det.locations], [det.dimensions], [det.orientations], [det.embeddings], list(self.hyper_distance_field.parameters())]

    def rebind(self):
        """(Re)build the pointer blocks the kernels read -- Adam's state for the box tensors, the hypernetwork's parameter / moment /
        counter / rate pointers -- from the CURRENT tensors of the modules and the optimiser, and drop the captured graphs (they hold
        the old addresses).  Called at construction and, through `_check_bindings`, by itself whenever a tensor was replaced behind
        the loop's back: `optimizer.load_state_dict(...)` (new moment tensors; learning rates possibly plain floats again),
        `module.to(...)`, `load_state_dict(assign=True)`.  Adam's state is created as torch.optim.Adam(capturable=True) lays it out
        (torch creates it lazily at a parameter's first step): the kernels update these tensors in place, so checkpoints and
        optimizer.state_dict() see them like any other state."""
        cfg, det = self.config, self.detector
        for group, params in zip(self.optimizer.param_groups, self._module_parameters()):
            # load_state_dict(assign=True) swaps the parameter OBJECTS of a module: the optimiser follows them, state included
            for old, new in zip(group["params"], params):
                if old is not new and old in self.optimizer.state:
                    self.optimizer.state[new] = self.optimizer.state.pop(old)
            group["params"] = list(params)
        for group in self.optimizer.param_groups:        # a loaded state dict may carry float rates: graph mode decays device tensors in place
            lr = group["lr"]
            if not (isinstance(lr, torch.Tensor) and lr.device == self.device and lr.dtype == torch.float32):
                group["lr"] = torch.as_tensor(float(lr), dtype=torch.float32, device=self.device).clone()
            for p in group["params"]:
                state = self.optimizer.state.get(p)
                if state and not (isinstance(state.get("step"), torch.Tensor) and state["step"].device == self.device and state["step"].dtype == torch.float32):
                    state["step"] = torch.as_tensor(float(state["step"]), dtype=torch.float32, device=self.device).clone()
        self._adam = [adam_state_tensors(self.optimizer, p, group, self._row)
                      for group, p in zip(self.optimizer.param_groups[:3], (det.locations, det.dimensions, det.orientations))]
        # the hypernetwork and the embeddings (residual phase) run through csrc/hypernetwork.h on the same footing: torch's module
        # owns the parameters, torch.optim.Adam owns the moments and counters, the kernels update both in place
        self._hypernetwork = hypernetwork_tensors(self.hyper_distance_field, det.embeddings, self.optimizer, cfg.lr_gamma, self._row)
        _destroy_graphs(self._graphs)
        # references keep every bound tensor alive until the next rebind: a replaced tensor's memory is not handed to somebody else
        # while a graph that writes through its address may still be replayed
        self._bound = self._bound_tensors()
        self._bound_signature = tuple(t.data_ptr() for t in self._bound)

    def _check_bindings(self):
        """Before anything is launched or replayed: are the tensors the pointer blocks were built from still the ones the modules and
        the optimiser hold?  (A tuple compare of ~130 addresses, ~10 us.)  If not, rebind -- the loop continues on the new tensors."""
        if not self.fused_glue:
            return
        current = self._bound_tensors()
        # (a loaded state dict may hold plain floats where the loop keeps device tensors, or lack a parameter's state: both rebind)
        if (len(current) != len(self._bound) or not all(isinstance(t, torch.Tensor) for t in current)
                or tuple(t.data_ptr() for t in current) != self._bound_signature):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("a tensor bound to the captured step was replaced during stream capture")
            if self._row is not None:
                raise RuntimeError("a parameter, moment, counter or rate of a batch frame was REPLACED (load_state_dict, module.to, assign=True): "
                                   "the frames of a FrameBatch keep them at fixed offsets of the batch's arena -- copy_ into the existing tensors")
            with _capture_lock:                          # (a device-wide synchronisation is refused while another frame's thread captures)
                torch.cuda.synchronize(self.device)      # replays that still write through the old addresses finish first
            self.rebind()

    def _fused_step(self, ray_indices, count=True, joins=(True, True), frames=None):
        """One step with the box-side glue in frame_step.h.  Same arithmetic as `_step_in_scope` (the eager torch path is its
        parity reference: tests/test_hip_step.py).  ``frames = (B, stride)``: this loop is frame 0 of a batch (``FrameBatch``) and
        every launch of the step covers the B frames whose buffers lie ``stride`` bytes apart (include/vsrd_hip.h, "frame batches")."""
        cfg, b, lib, frame = self.config, self._glue, _lib.load(), self._frame
        batch = (1, 0) if frames is None else (int(frames[0]), int(frames[1]))
        frame.num_frames, frame.frame_stride = batch
        self._hypernetwork.num_frames, self._hypernetwork.frame_stride = batch
        det = self.detector
        N = self.num_instances
        residual = self.step_index >= cfg.warmup_steps
        main = torch.cuda.current_stream(self.device)
        stream = _lib.stream()
        rays_branch, net_branch = self._branches
        fused_net = residual and self.fused_hypernetwork
        sampled = ray_indices is None
        # the draw rides in the prologue's launch (a second workgroup): no branch, no join
        with_prologue = sampled and self.ray_table is not None and getattr(self, "_prologue_draws", True)
        if batch[0] > 1 and not (with_prologue and (fused_net or not residual)):
            raise RuntimeError("a batch of frames draws its rays from the frames' sampling tables inside the prologue launch and runs the "
                               "hypernetwork through csrc/hypernetwork.h")
        if sampled and not with_prologue:   # branch 1: this step's rays (device sampler keyed by the step counter, ray_sampling.h)
            rays_branch.wait_stream(main)
            with torch.cuda.stream(rays_branch):
                self._draw_rays(b["ray_indices"], b["picks"])
        if sampled:
            ray_indices = b["ray_indices"]
        if fused_net:                   # branch 2: embeddings -> MLP weights
            hyper_ws = b["hyper_workspace"]
            if joins[0]:                # (inside a several-step graph the branch simply continues behind the hypernetwork backward of the step
                net_branch.wait_stream(main)     # before: nothing on the main stream concerns it until this step's render adjoint)
            with torch.cuda.stream(net_branch):
                _lib.check(lib.vsrd_hypernetwork_forward(self._hypernetwork, hyper_ws.data_ptr(), hyper_ws.numel(), _lib.ptr(b["mlp_weights"]),
                                                         _lib.ptr(b["mlp_centred"]), _lib.stream()))
        prologue_args = (frame, _lib.ptr(det.locations.data), _lib.ptr(det.dimensions.data), _lib.ptr(det.orientations.data),
                         _lib.ptr(b["extrinsics"]), _lib.ptr(b["intrinsics"]), _lib.ptr(b["gt_boxes"]), b["visible"].data_ptr(),
                         self.step_tensor.data_ptr(), b["scratch"].data_ptr(), b["scratch"].numel(), _lib.ptr(b["instances"]),
                         b["pd_indices"].data_ptr(), b["gt_indices"].data_ptr(), b["target_columns"].data_ptr(),
                         _lib.ptr(b["instance_weights"]), _lib.ptr(self.schedule), _lib.ptr(b["projection_losses"]), _lib.ptr(b["grad_raw"]))
        if with_prologue:
            table = self.ray_table
            code = lib.vsrd_frame_prologue_sample(*prologue_args, table.table.data_ptr(), table.count, cfg.num_rays, (cfg.seed + 1) & 0xFFFFFFFFFFFFFFFF,
                                                  None, b["ray_indices"].data_ptr(), stream)          # (the table covers every pixel: no remap)
            if code == _lib.E_UNSUPPORTED and batch[0] == 1 and not torch.cuda.is_current_stream_capturing():
                # the combined launch needs ~125 KB of LDS in one workgroup (cost matrix + the sampler's hash table): where that opt-in is
                # refused (and only then: a failed launch is an error), the frame keeps the bit-identical two-launch form
                # (tests: test_frame_prologue_matches_the_torch_path)
                import warnings
                warnings.warn("vsrd_frame_prologue_sample: the LDS opt-in of the combined launch was refused; this frame draws its rays "
                              "in a launch of their own (vsrd_frame_prologue + vsrd_sample_rays_table)", RuntimeWarning)
                self._prologue_draws = False
                _lib.check(lib.vsrd_frame_prologue(*prologue_args, stream))
                self._draw_rays(b["ray_indices"], b["picks"])
            else:
                _lib.check(code)
        else:
            _lib.check(lib.vsrd_frame_prologue(*prologue_args, stream))
        if sampled and not with_prologue:
            main.wait_stream(rays_branch)
        if fused_net:
            main.wait_stream(net_branch)
        ray_indices = ray_indices.contiguous()
        R = int(ray_indices.numel())
        weights = cfg.loss_weights
        eikonal_ratio = weights["eikonal_loss"] / weights["silhouette_loss"] if residual else 0.0
        from .rendering import renderers
        flags = (renderers._base_flags() & ~_lib.FLAG_MLP_SPLIT_BF16) | _lib.FLAG_YAW_GRADIENTS      # (the prologue decodes rotation_matrix_y; the epilogue reads r00, r02, r20, r22 only)
        if residual and cfg.mlp_split_bf16:
            flags |= _lib.FLAG_MLP_SPLIT_BF16
        mlp_weights = centred = None
        if fused_net:
            centred = b["mlp_centred"]
            flags |= _lib.FLAG_MLP_WEIGHTS_CENTRED
        elif residual:            # A/B: the hypernetwork through torch (rocBLAS GEMMs, autograd, torch.optim.Adam)
            self.optimizer.zero_grad(set_to_none=True)
            mlp_weights = self.hyper_distance_field(det.embeddings)[0].contiguous()          # [N,1617]
            centred = renderers._centre_mlp(mlp_weights)
            flags |= _lib.FLAG_MLP_WEIGHTS_CENTRED
        elif cfg.skip_exact_misses:
            flags |= _lib.FLAG_SKIP_EXACT_MISSES
        workspace = self.workspace.adjoint(self.device, N, residual, step_shape=(cfg.num_samples, R))
        field = _lib.make_field(b["instances"], 1.0, centred)          # (the temperature comes from the device schedule)
        config = _lib.make_config(R, cfg.num_samples, cfg.distance_range, 1.0, 1.0, 1.0e-6, 3, seed=cfg.seed, stream_offset=self.step_tensor, flags=flags,
                                  schedule=self.schedule, gather=(ray_indices, self.pixels_per_view, b["target_columns"], N), frames=frames,
                                  adjoint_slots_per_item=cfg.mlp_adjoint_item_slots if residual else 0)
        loss_scale = 1.0 / (R * N)
        if residual:
            _lib.check(lib.vsrd_render_residual_step(field, config, _lib.ptr(self.camera_positions), _lib.ptr(self.ray_directions), None, None,
                                                     _lib.ptr(b["masks"]), _lib.ptr(b["instance_weights"]), loss_scale, float(eikonal_ratio),
                                                     workspace.data_ptr(), workspace.numel(), _lib.ptr(b["render_losses"]), _lib.ptr(b["grad_instances"]),
                                                     _lib.ptr(b["grad_mlp"]), None, stream))
        else:
            _lib.check(lib.vsrd_render_silhouette_step(field, config, _lib.ptr(self.camera_positions), _lib.ptr(self.ray_directions), None, None,
                                                       _lib.ptr(b["masks"]), _lib.ptr(b["instance_weights"]), loss_scale,
                                                       workspace.data_ptr(), workspace.numel(), _lib.ptr(b["render_losses"]), _lib.ptr(b["grad_instances"]),
                                                       None, stream))
        if fused_net:     # branch: backward through the hypernetwork, Adam on it and on the embeddings, both rates decayed (next to the epilogue)
            net_branch.wait_stream(main)
            with torch.cuda.stream(net_branch):
                _lib.check(lib.vsrd_hypernetwork_backward_step(self._hypernetwork, hyper_ws.data_ptr(), hyper_ws.numel(), _lib.ptr(b["grad_mlp"]),
                                                               float(weights["silhouette_loss"]), _lib.stream()))
        groups = self.optimizer.param_groups
        # ExponentialLR decays every group's rate AFTER the optimiser step: the epilogue does it for the groups it steps itself (and, in
        # the box-only phase, for the two that have nothing to step); in the residual phase those two are decayed after torch's step
        others = (None, None) if residual else (_lib.ptr(groups[3]["lr"]), _lib.ptr(groups[4]["lr"]))
        _lib.check(lib.vsrd_frame_epilogue(frame, _lib.ptr(b["grad_instances"]), _lib.ptr(b["grad_raw"]), _lib.ptr(b["projection_losses"]),
                                           _lib.ptr(b["render_losses"]), float(eikonal_ratio), self._adam[0], self._adam[1], self._adam[2],
                                           others[0], others[1], self.step_tensor.data_ptr(),
                                           _lib.ptr(b["record"]), _lib.ptr(b["raw_gradients"]), stream))
        if residual and not fused_net:    # autograd + torch.optim.Adam (the box tensors have no .grad: skipped)
            mlp_weights.backward(b["grad_mlp"] * weights["silhouette_loss"])
            self.optimizer.step()
            groups[3]["lr"].mul_(cfg.lr_gamma)
            groups[4]["lr"].mul_(cfg.lr_gamma)
        if fused_net and joins[1]:
            main.wait_stream(net_branch)
        if count:
            self.step_index += 1
        record, raw = b["record"], b["raw_gradients"]
        result = dict(iou_projection_loss=record[0], l1_projection_loss=record[1], silhouette_loss=record[2], loss=record[4],
                      raw_gradients=[raw[:, 0:3].unsqueeze(0), raw[:, 3:6].unsqueeze(0), raw[:, 6:8].unsqueeze(0)])
        if residual:
            result["eikonal_loss"] = record[3]
        return result

    # ------------------------------------------------------------------------------------------------
    def sample_rays(self):
        """main.py:620-627: importance-sample rays by the strongest soft mask (torch.multinomial, no replacement).  Graph mode
        uses the library's own sampler (same algorithm, Philox keyed by the device-side step counter): ATen's captured multinomial
        faults on replay with this torch build, and it sorts all V*H*W keys every step."""
        if self.graph:
            return self._draw_rays(torch.empty(self.config.num_rays, dtype=torch.int64, device=self.device))
        return torch.multinomial(self.sampling_weights, self.config.num_rays, replacement=False)

    def _draw_rays(self, out, picks=None):
        """Graph mode's draw of this step's rays into `out` (pixel indices over all views), keyed by the device-side step counter."""
        cfg = self.config
        if self.ray_table is not None:
            return self.ray_table.sample(cfg.num_rays, seed=cfg.seed + 1, stream_offset=self.step_tensor, out=out, remap=self.ray_remap)
        picks = rendering.sample_rays(self.positive_weights, cfg.num_rays, seed=cfg.seed + 1, stream_offset=self.step_tensor, out=picks)
        return torch.index_select(self.positive_pixels, 0, picks, out=out)

    def field_block(self, outputs, temperature, mlp_weights=None):
        return fields.FieldBlock(fields.pack_instances(outputs["locations"][0], outputs["orientations"][0], outputs["dimensions"][0]),
                                 float(temperature), mlp_weights, None, yaw_gradients=True)      # (BoxParameters3D: rotation_matrix_y)

    def step(self, ray_indices: Optional[torch.Tensor] = None, u_coarse=None, u_fine=None):
        """One optimisation step.  Steps < warmup_steps optimise the boxes only; later steps add the per-instance residual MLP
        (hypernetwork on the embeddings, main.py:525-578) and the eikonal loss (main.py:679-687).
        ray_indices / uniforms may be supplied for reproducible parity runs."""
        if self.graph:
            if u_coarse is not None or u_fine is not None:
                raise ValueError("graph mode draws its uniforms in the kernels (Philox keyed by the device-side step counter)")
            return self._graph_step(ray_indices)
        return self._step(ray_indices, u_coarse, u_fine)

    # ---- hipGraph mode ---------------------------------------------------------------------------------
    def _device_schedule(self):
        """scripts/main.py:420-431 evaluated on the device from the device-side step counter."""
        cfg = self.config
        x = self.step_tensor.to(torch.float32) / cfg.num_steps
        anneal = (torch.cos(math.pi * x) + 1.0) / 2.0
        self.schedule.copy_(torch.cat([anneal * (cfg.max_sdf_union_temperature - cfg.min_sdf_union_temperature) + cfg.min_sdf_union_temperature,
                                       anneal * (cfg.max_sdf_std_deviation - cfg.min_sdf_std_deviation) + cfg.min_sdf_std_deviation, x]))

    def _graph_step(self, ray_indices):
        """Three eager steps per (phase, ray source) on a side stream warm the allocator and the lazy initialisations, then the step is
        captured once and replayed.  The eager steps run the same device-side code, so they are ordinary optimisation steps."""
        if ray_indices is not None and int(ray_indices.numel()) > self.config.num_rays:
            # the captured graphs hold the address of scratch sized for config.num_rays rays; a larger launch would outgrow it
            raise ValueError(f"graph mode replays steps of at most config.num_rays = {self.config.num_rays} rays, got {int(ray_indices.numel())}")
        self._check_bindings()
        residual = self.step_index >= self.config.warmup_steps
        key = (residual, ray_indices is not None)
        if key in self._graphs:
            graph, static_rays, outputs = self._graphs[key]
            with _capture_lock.replaying():      # (the copy too: nothing of a frame calls into HIP during another frame's capture)
                if static_rays is not None:
                    static_rays.copy_(ray_indices)
                graph.replay()
            self.step_index += 1
            return outputs
        done = self._eager_graph_steps.get(key, 0)
        if done < 3:
            with _capture_lock:          # (a phase's three eager steps too: nothing of a frame calls into HIP next to another frame's capture)
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    outputs = self._step(ray_indices, None, None)
                torch.cuda.current_stream(self.device).wait_stream(side)
            self._eager_graph_steps[key] = done + 1
            return outputs
        static_rays = ray_indices.clone() if ray_indices is not None else None
        graph = torch.cuda.CUDAGraph()
        # an own capture stream per loop: torch keys the rocBLAS / hipBLASLt workspaces by stream, and its default capture stream is
        # shared by every capture -- two loops replayed at the same time would then run their GEMMs in one workspace
        if self._capture_stream is None:
            self._capture_stream = torch.cuda.Stream(device=self.device)
        # stream capture is a process-wide mode: one capture at a time, and thread-local error checking so that another frame's
        # thread (launcher.run_frames(frames_in_flight=2)) may keep replaying its own graph meanwhile
        # (no device-wide synchronisation out here: while ANOTHER frame's thread captures, HIP refuses it -- "operation not permitted when
        #  stream is capturing"; torch.cuda.graph synchronises by itself once this thread holds the lock and nobody captures)
        with _capture_lock.capture():
            with torch.cuda.graph(graph, stream=self._capture_stream, capture_error_mode="thread_local"):
                outputs = self._step(static_rays, None, None, count=False)
            self._graphs[key] = (graph, static_rays, outputs)
            # capture does not execute: this replay IS the step.  A graph's FIRST launch stays inside the capture's turn: next to another
            # thread's launch it died inside hipGraphLaunch (one 36-frame run in twelve; later launches of an uploaded graph have not)
            graph.replay()
        self.step_index += 1
        return outputs

    def run(self, num_steps, steps_per_graph=4):
        """``num_steps`` optimisation steps, the last step's outputs returned.  In graph mode up to ``steps_per_graph`` consecutive
        steps of one phase are captured in ONE hipGraph and replayed together: between two graph launches the GPU idles for ~20 us
        (profiles/r03/native_step_timeline_residual.txt: the gap behind the last kernel of a replay), a fifth of a box-only step.  The
        steps themselves are the same launches in the same order as ``step()``'s, so the trajectory is bit-identical
        (tests/test_hip_step.py::test_run_replays_several_steps_per_graph)."""
        outputs = None
        remaining = int(num_steps)
        while remaining > 0:
            k = 1
            if self.graph and self.fused_glue and steps_per_graph > 1:
                residual = self.step_index >= self.config.warmup_steps
                to_boundary = remaining if residual else self.config.warmup_steps - self.step_index
                # whole groups of `steps_per_graph` only: a tail shorter than that replays the one-step graph (every distinct k would
                # otherwise cost a capture of its own -- a synchronisation -- for one use)
                k = int(steps_per_graph) if min(remaining, to_boundary) >= int(steps_per_graph) else 1
                if self._eager_graph_steps.get((residual, False), 0) < 3 or (residual, False) not in self._graphs:
                    k = 1            # the phase's eager warm-up steps and its one-step graph come first (step() owns that protocol)
            outputs = self._graph_replay_many(k) if k > 1 else self.step()
            remaining -= k
        return outputs

    def _graph_replay_many(self, k):
        self._check_bindings()
        residual = self.step_index >= self.config.warmup_steps
        key = (residual, False, k)
        if key not in self._graphs:
            graph = torch.cuda.CUDAGraph()
            with _capture_lock.capture():
                with torch.cuda.graph(graph, stream=self._capture_stream, capture_error_mode="thread_local"):
                    for j in range(k):  # the hypernetwork's branch is joined to the step's stream only at the ends of the graph: between two
                        # steps the next prologue (which needs the epilogue only) overlaps the hypernetwork's backward and forward
                        outputs = self._step(None, None, None, count=False, joins=(j == 0, j == k - 1))
                self._graphs[key] = (graph, None, outputs)
                graph.replay()          # (the first launch inside the capture's turn: _graph_step)
            self.step_index += k
            return outputs
        graph, _, outputs = self._graphs[key]
        with _capture_lock.replaying():
            graph.replay()
        self.step_index += k
        return outputs

    def _step(self, ray_indices, u_coarse, u_fine, count=True, joins=(True, True)):
        with rendering.workspace_scope(self.workspace):
            if self.fused_glue:
                if not torch.cuda.is_current_stream_capturing():
                    self._check_bindings()
                return self._fused_step(ray_indices, count, joins)
            return self._step_in_scope(ray_indices, u_coarse, u_fine, count)

    def _step_in_scope(self, ray_indices, u_coarse, u_fine, count):
        cfg, inp = self.config, self.inputs
        step = self.step_index
        residual = step >= cfg.warmup_steps
        self.optimizer.zero_grad(set_to_none=True)
        outputs = self.detector()
        # ---- multi-view projection, matching, projection losses (main.py:339-415) --------------------
        pd_boxes_2d, _ = operations.project_boxes_multi_view(outputs["boxes_3d"][0], inp.extrinsic_matrices, inp.intrinsic_matrices, inp.image_size)
        pd_idx, gt_idx = losses.match_instances(pd_boxes_2d[0], inp.boxes_2d[0])
        iou_loss, l1_loss = losses.projection_losses(pd_boxes_2d, inp.boxes_2d, inp.visible_masks, pd_idx, gt_idx)
        # ---- instance loss (main.py:420-671) ---------------------------------------------------------
        ratio, temperature, std = losses.schedules(step, cfg.num_steps, cfg.max_sdf_union_temperature, cfg.min_sdf_union_temperature,
                                                   cfg.max_sdf_std_deviation, cfg.min_sdf_std_deviation)
        schedule = offset = None
        if self.graph:       # the kernels read (temperature, std, ratio) and the Philox counter from device memory instead
            self._device_schedule()
            schedule, offset = self.schedule, self.step_tensor
        mlp_weights = self.hyper_distance_field(outputs["embeddings"])[0].contiguous() if residual else None     # [N,1617]
        block = self.field_block(outputs, temperature, mlp_weights)
        if ray_indices is None:
            ray_indices = self.sample_rays()
        origins = self.camera_positions[ray_indices // self.pixels_per_view]
        directions = self.ray_directions[ray_indices]
        weights = cfg.loss_weights
        if residual:    # render + silhouette BCE + eikonal term + adjoint (boxes and MLP weights) in one launch
            eikonal_ratio = weights["eikonal_loss"] / weights["silhouette_loss"]
            rendered, parts = rendering.silhouette_step(block, origins, directions, self.flat_masks[ray_indices], cfg.distance_range,
                                                        cfg.num_samples, std, ratio, pd_indices=pd_idx, gt_indices=gt_idx,
                                                        u_coarse=u_coarse, u_fine=u_fine, seed=cfg.seed, stream_offset=step if offset is None else offset,
                                                        skip_exact_misses=False, schedule=schedule, eikonal_ratio=eikonal_ratio, return_terms=True,
                                                        mlp_split_bf16=cfg.mlp_split_bf16)
            silhouette, eikonal = parts[0], parts[1]
        else:           # box-only phase: render + silhouette BCE + adjoint in one launch
            rendered = rendering.silhouette_step(block, origins, directions, self.flat_masks[ray_indices], cfg.distance_range,
                                                 cfg.num_samples, std, ratio, pd_indices=pd_idx, gt_indices=gt_idx,
                                                 u_coarse=u_coarse, u_fine=u_fine, seed=cfg.seed, stream_offset=step if offset is None else offset,
                                                 skip_exact_misses=cfg.skip_exact_misses, schedule=schedule)
            silhouette = rendered.detach()
        terms = dict(iou_projection_loss=iou_loss, l1_projection_loss=l1_loss, silhouette_loss=silhouette)
        if residual:
            terms["eikonal_loss"] = eikonal
        # main.py:855: sum of weighted terms; `rendered` already is silhouette (+ eikonal_ratio * eikonal)
        total = weights["iou_projection_loss"] * iou_loss + weights["l1_projection_loss"] * l1_loss + weights["silhouette_loss"] * rendered
        total.backward()
        raw_gradients = [p.grad.detach().clone() for p in (self.detector.locations, self.detector.dimensions, self.detector.orientations)]
        self.optimizer.step()
        if self.graph:      # ExponentialLR and the step counter, in place on the device
            for group in self.optimizer.param_groups:
                group["lr"].mul_(cfg.lr_gamma)
            self.step_tensor.add_(1)
        else:
            self.scheduler.step()
        if count:
            self.step_index += 1
        terms["loss"] = total
        result = {name: value.detach() for name, value in terms.items()}
        result["raw_gradients"] = raw_gradients
        return result

    # ---- a frame SLOT: the loop and its captured graphs, reused for frame after frame (round 5) -----------------------------------------
    def capture_all(self, steps_per_graph=4):
        """Run, NOW, every eager warm-up step and every capture a frame's ``run()`` would meet on its way -- box-only and residual phase, the
        one-step and the ``steps_per_graph``-step graph of each -- on the current frame's data, then ``reset()`` the frame.  A launcher
        does this once per slot at start-up, one slot after the other and before any worker thread exists: afterwards a frame is copies,
        fills and graph replays only, and stream capture -- a process-wide mode that every collision of round 4 had on one side -- never
        happens next to another frame's work again.  Returns the number of graphs held."""
        if not (self.graph and self.fused_glue and self.persistent):
            raise ValueError("capture_all() belongs to a persistent graph-mode loop (FrameOptimizer(..., graph=True, persistent=True))")
        cfg = self.config
        k = int(steps_per_graph)
        for phase_start in sorted({0, min(cfg.warmup_steps, cfg.num_steps)}):
            phase_steps = (cfg.warmup_steps if phase_start < cfg.warmup_steps else cfg.num_steps) - phase_start
            if phase_steps <= 0:
                continue
            with _capture_lock:
                self.step_index = phase_start
                self.step_tensor.fill_(phase_start)
            for _ in range(min(4, phase_steps)):     # three eager steps, then the capture (and first replay) of the one-step graph
                self.step()
            if k > 1 and phase_steps >= 4 + k:       # (run() takes the k-step graph only for whole groups of k inside a phase)
                self._graph_replay_many(k)
        with _capture_lock:
            torch.cuda.current_stream(self.device).synchronize()
        self.reset(self.inputs)
        return len(self._graphs)

    def reset(self, inputs: FrameInputs, init_seed=None):
        """Start ANOTHER frame of the same shape in this loop, in place: the new frame's matrices, boxes and soft masks are copied into
        the buffers the captured graphs read; parameters are re-initialised as a fresh ``BoxParameters3D`` / ``HyperDistanceField`` would
        be (scripts/main.py:106-199 builds new models per frame); Adam's moments, counters and learning rates, the step counter and the
        sampling table start over.  No allocation that a graph would have to learn about, no capture.  Returns False -- and leaves the
        loop unusable for that frame -- when the frame's importance weights do not suit the table sampler the graphs were captured with
        (the caller then optimises it in a loop of its own)."""
        if not (self.graph and self.fused_glue and self.persistent):
            raise ValueError("reset() belongs to a persistent graph-mode loop (FrameOptimizer(..., graph=True, persistent=True))")
        cfg, dev, b = self.config, self.device, self._glue
        V, N = self.num_views, self.num_instances
        if tuple(inputs.soft_masks.shape) != (V, *[int(x) for x in self.inputs.image_size], N) or tuple(inputs.image_size) != tuple(self.inputs.image_size):
            raise ValueError(f"a slot serves frames of ONE shape: {tuple(self.inputs.soft_masks.shape)}, got {tuple(inputs.soft_masks.shape)}")
        with _capture_lock, torch.no_grad():
            main = torch.cuda.current_stream(dev)
            main.synchronize()                                      # (replays of the previous frame have finished: nothing reads the buffers)
            # ---- inputs ----
            H, W = (int(x) for x in inputs.image_size)
            cam, dirs = rendering.ray_casting((H, W), inputs.intrinsic_matrices, inputs.extrinsic_matrices)
            self.camera_positions.copy_(cam)
            self.ray_directions.copy_(dirs.reshape(-1, 3))
            self.flat_masks.copy_(inputs.soft_masks.reshape(-1, N))
            if b["masks"].data_ptr() != self.flat_masks.data_ptr():
                b["masks"].copy_(self.flat_masks)
            torch.amax(self.flat_masks, dim=-1, out=self.sampling_weights)
            b["extrinsics"].copy_(inputs.extrinsic_matrices.reshape(V, 16))
            b["intrinsics"].copy_(inputs.intrinsic_matrices.reshape(V, 9))
            b["gt_boxes"].copy_(inputs.boxes_2d.reshape(V, N, 4))
            b["visible"].copy_(inputs.visible_masks)
            self.inputs = inputs
            # ---- parameters, optimiser, counters ----
            with _initial_draw:                                   # (OptimizationConfig.init_seed, for THIS frame: seed and draw as one step -- the generator is the process's,
                if init_seed is not None:                         #  and the capture gate lets several frames' threads in here at once)
                    torch.default_generator.manual_seed(int(init_seed))
                fresh_detector = models.BoxParameters3D(1, N)
                fresh_field = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256])
            for module, fresh in ((self.detector, fresh_detector), (self.hyper_distance_field, fresh_field)):
                for p, q in zip(module.parameters(), fresh.parameters()):
                    p.copy_(q)
            initial = [cfg.learning_rate] * 3 + [cfg.embedding_learning_rate, cfg.hypernetwork_learning_rate]
            for group, rate in zip(self.optimizer.param_groups, initial):
                group["lr"].fill_(rate)
                for p in group["params"]:
                    state = self.optimizer.state.get(p)
                    if state:
                        state["exp_avg"].zero_(); state["exp_avg_sq"].zero_(); state["step"].zero_()
            self.step_tensor.zero_()
            self.schedule.fill_(1.0)
            self.step_index = 0
            for name in ("record", "raw_gradients", "grad_raw", "grad_instances", "grad_mlp", "projection_losses", "render_losses"):
                b[name].zero_()
            suits = self._prepare_sampler(cfg)
            main.synchronize()
        return suits

    def boxes(self):
        with torch.no_grad():
            return self.detector()

    def close(self):
        """Drop the captured graphs and the scratch buffers (also happens when the optimizer is garbage-collected).  Once per frame
        this is also where the device ray sampler's sticky overflow flag is read back (a host synchronisation, so not per step): a
        draw whose threshold bin held more candidate keys than the sampler lists was incomplete and not reproducible."""
        with _capture_lock:
            incomplete = bool(self.graph and self.ray_table is not None and self.ray_table.incomplete())
            overflowed = bool(self.graph and self.workspace.sampler_overflowed(self.device))
        if incomplete:
            import warnings
            warnings.warn("vsrd_sample_rays_table ran out of picks in some step of this frame: those draws repeat some rays "
                          "(see csrc/ray_sampling.h and RayTable.suits)", RuntimeWarning)
        if overflowed:
            import warnings
            warnings.warn("vsrd_sample_rays overflowed its candidate list in some step of this frame: those draws were incomplete "
                          "(many equal importance weights in one histogram bin); see csrc/ray_sampling.h", RuntimeWarning)
        _destroy_graphs(self._graphs)
        self.ray_table = None                  # (43 MB per frame at the reference's size: the table and its guide)
        self.workspace.release()

    def __del__(self):
        try:
            _destroy_graphs(self._graphs, wait=False)
        except Exception:                      # (interpreter shutdown: module globals may be gone)
            pass

    # ---- checkpoint views (scripts/main.py:1109-1121 saves optimizer.state_dict() and scheduler.state_dict()) ----------------
    def optimizer_state_dict(self):
        """``optimizer.state_dict()`` in the layout the reference's ``torch.optim.Adam`` loads: float learning rates (graph mode
        keeps them as device tensors decayed in place), ``initial_lr`` per group as ExponentialLR records it, step counters as
        host float tensors (capturable Adam keeps them on the device)."""
        cfg = self.config
        state = self.optimizer.state_dict()
        initial = [cfg.learning_rate] * 3 + [cfg.embedding_learning_rate, cfg.hypernetwork_learning_rate]
        groups = []
        for group, base in zip(state["param_groups"], initial):
            group = dict(group)
            group["lr"] = float(group["lr"])
            group.setdefault("initial_lr", base)
            group["capturable"] = False
            groups.append(group)
        per_param = {}
        for index, entry in state["state"].items():
            entry = dict(entry)
            if isinstance(entry.get("step"), torch.Tensor):
                entry["step"] = entry["step"].detach().to("cpu", torch.float32)
            per_param[index] = entry
        return dict(state=per_param, param_groups=groups)

    def scheduler_state_dict(self):
        """``ExponentialLR.state_dict()`` after ``step_index`` steps.  Graph mode has no scheduler object (the rates are decayed on
        the device inside the captured step): the same dictionary is assembled from the step counter and the device-side rates."""
        if self.scheduler is not None:
            return self.scheduler.state_dict()
        cfg = self.config
        return {"gamma": cfg.lr_gamma,
                "base_lrs": [cfg.learning_rate] * 3 + [cfg.embedding_learning_rate, cfg.hypernetwork_learning_rate],
                "last_epoch": self.step_index, "_step_count": self.step_index + 1, "_is_initial": False,
                "_get_lr_called_within_step": False, "_last_lr": [float(group["lr"]) for group in self.optimizer.param_groups]}


class FrameBatch:
    """B frames of ONE shape optimised in lock-step: every launch of a step -- prologue + ray draw, hypernetwork forward, render +
    losses + adjoint, MLP adjoint, reductions, hypernetwork backward + Adam, epilogue -- covers all B frames at once (include/vsrd_hip.h,
    "frame batches").  The reference's loop carries the same batch dimension (scripts/main.py:525-651: lists over the batch of distance
    fields, camera positions, ray directions, soft masks; BoxParameters3D(batch_size, num_instances), box_parameters.py:34-49); its
    configs use batch_size = 1 because one V100 was full with one frame -- one MI355X is not (1000 rays = one wave per SIMD).

    The frames stay INDEPENDENT problems: each has its own detector, hypernetwork, Adam state, learning rates, step counter (which keys its
    Philox streams), sampling table and scratch, in its own row of ``self.arena``; nothing is summed across frames.  A frame of a batch
    therefore walks, bit for bit, the trajectory it walks alone in a ``FrameOptimizer(graph=True, persistent=True)``
    (tests/test_hip_step.py::test_frame_batch_walks_each_frames_own_trajectory).  Each member is such a loop whose buffers live in the
    arena; the batch launches member 0's step with ``frames = (B, stride)``.

    Like a frame slot, a batch is built and captured once (``capture_all``) and then serves group after group of frames through
    ``reset(f, inputs, init_seed)``; a last group of fewer than B frames runs with ``active < B`` (the first ``active`` rows)."""

    def __init__(self, inputs, config, device, init_seeds=None):
        inputs = list(inputs)
        if not inputs:
            raise ValueError("FrameBatch needs at least one frame")
        shape = tuple(inputs[0].soft_masks.shape)
        if any(tuple(i.soft_masks.shape) != shape or tuple(i.image_size) != tuple(inputs[0].image_size) for i in inputs):
            raise ValueError("the frames of a batch share ONE shape (views, height, width, instances)")
        self.size = len(inputs)
        import dataclasses
        if config.mlp_adjoint_item_slots == 0:
            config = dataclasses.replace(config, mlp_adjoint_item_slots=self.item_slots(self.size, config, shape[-1]))
        self.config, self.device = config, torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        with _capture_lock:
            self.arena = FrameArena(self.size, self._row_bytes(shape, config), self.device)
        # A row is FOUNDED on a frame that suits the table sampler its graphs will draw from (the frames a row later serves come through reset(),
        # which says so when one does not suit).  A founder that does not suit is replaced by the first one that does -- the shape is what a
        # row needs from its founder -- and listed in `unsuitable_founders`; no suitable founder at all: UnsuitableFrameError.
        self.frames, self.unsuitable_founders = [], []
        stand_in = None
        for f in range(self.size):
            cfg = config if init_seeds is None else dataclasses.replace(config, init_seed=init_seeds[f])
            row = self.arena.rows[f]
            candidates = [inputs[f]] + ([stand_in] if stand_in is not None else inputs[f + 1:])
            for position, frame_inputs in enumerate(candidates):
                row.cursor, row.layout = 0, []                           # (a founder that failed had begun to fill the row)
                try:
                    member = FrameOptimizer(frame_inputs, cfg, self.device, graph=True, persistent=True, row=row)
                except UnsuitableFrameError:
                    if position == 0:
                        self.unsuitable_founders.append(f)
                    if position + 1 == len(candidates):
                        raise
                    continue
                if stand_in is None:
                    stand_in = frame_inputs
                self.frames.append(member)
                break
        lead = self.arena.rows[0].layout
        for row in self.arena.rows[1:]:
            if row.layout != lead:
                raise RuntimeError("the frames of a batch laid their buffers out differently: frame batches need equal offsets in every row")
        self._graphs = {}
        self._eager = {}
        self._capture_stream = None

    @staticmethod
    def item_slots(num_frames, config, num_instances):
        """Slots per work item of the MLP adjoint for a batch of `num_frames` frames (vsrd_render_config::adjoint_slots_per_item): the library
        plans about 16384 items per LAUNCH from one frame's size; a batch has B times the items, so its items can be B times as large --
        fewer partial rows to write and to sum (measured, 1000 rays x 100 samples, N = 8: 8 frames 0.392 -> 0.337 ms per frame-step with 16
        slots, 4 frames 0.423 -> 0.379; profiles/r06_native).  The next power of two, 4..32."""
        rounds = 2 if config.num_samples <= 64 else 4
        wanted = num_frames * config.num_rays * rounds * num_instances / 16384.0
        slots = 4
        while slots < wanted and slots < 32:
            slots *= 2
        return slots

    @staticmethod
    def _row_bytes(shape, config):
        """Bytes of one frame's row: every device buffer a FrameOptimizer(row=...) creates (FrameRow.new raises if this is short)."""
        V, H, W, N = (int(x) for x in shape)
        lib = _lib.load()
        pixels = V * H * W
        with torch.device("meta"):               # (no memory, no random numbers drawn)
            parameters = sum(p.numel() for p in models.BoxParameters3D(1, N).parameters()) + \
                sum(p.numel() for p in models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).parameters())
        workspace = max(lib.vsrd_workspace_bytes(N, 1), lib.vsrd_residual_step_workspace_bytes(N, int(config.num_samples), int(config.num_rays)))
        total = 4 * 3 * parameters                                   # parameters, exp_avg, exp_avg_sq
        total += pixels * (12 + 4 * N + 4) + 12 * V                  # ray directions, soft masks, importance weights, camera positions
        total += lib.vsrd_ray_table_bytes(pixels)
        total += workspace + lib.vsrd_hypernetwork_workspace_bytes(N) + lib.vsrd_frame_scratch_bytes(V, N)
        total += 4 * (3 * N * _lib.MLP_WEIGHTS + 64 * N + 128 * V) + 16 * int(config.num_rays)          # the glue's tables
        return total + (4 << 20)                                     # 256-byte alignment of ~400 buffers, step counters, rates, slack

    # ---- one step of the first `active` frames -----------------------------------------------------------------------------------
    def _members(self, active):
        active = self.size if active is None else int(active)
        if not 1 <= active <= self.size:
            raise ValueError(f"active frames: 1..{self.size}, got {active}")
        return active, self.frames[:active]

    def _step(self, active, count=True, joins=(True, True)):
        active, members = self._members(active)
        lead = members[0]
        if any(m.step_index != lead.step_index for m in members):
            raise RuntimeError("the frames of a batch step together: reset() every active frame before running the batch")
        with rendering.workspace_scope(lead.workspace):
            if not torch.cuda.is_current_stream_capturing():
                for m in members:
                    m._check_bindings()
            lead._fused_step(None, count=False, joins=joins, frames=(active, self.arena.stride))
        if count:
            for m in members:
                m.step_index += 1

    def outputs(self, f):
        """Frame f's record of its last step: the dictionary ``FrameOptimizer.step`` returns (device tensors, rewritten by every step)."""
        m = self.frames[f]
        record, raw = m._glue["record"], m._glue["raw_gradients"]
        result = dict(iou_projection_loss=record[0], l1_projection_loss=record[1], silhouette_loss=record[2], loss=record[4],
                      raw_gradients=[raw[:, 0:3].unsqueeze(0), raw[:, 3:6].unsqueeze(0), raw[:, 6:8].unsqueeze(0)])
        if m.step_index > self.config.warmup_steps:
            result["eikonal_loss"] = record[3]
        return result

    def _phase(self, active):
        _, members = self._members(active)
        return members[0].step_index >= self.config.warmup_steps

    def step(self, active=None):
        """One optimisation step of the first ``active`` frames: three eager steps per (phase, active) warm the lazy initialisations, then the
        step is captured once and replayed (the protocol of ``FrameOptimizer._graph_step``)."""
        active, members = self._members(active)
        key = (self._phase(active), active, 1)
        if key in self._graphs:
            with _capture_lock.replaying():
                self._graphs[key].replay()
            for m in members:
                m.step_index += 1
            return
        done = self._eager.get(key, 0)
        if done < 3:
            with _capture_lock:
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    self._step(active)
                torch.cuda.current_stream(self.device).wait_stream(side)
            self._eager[key] = done + 1
            return
        self._capture(key, active, 1)

    def _capture(self, key, active, k):
        _, members = self._members(active)
        graph = torch.cuda.CUDAGraph()
        if self._capture_stream is None:
            self._capture_stream = torch.cuda.Stream(device=self.device)
        with _capture_lock.capture():
            with torch.cuda.graph(graph, stream=self._capture_stream, capture_error_mode="thread_local"):
                for j in range(k):      # (the hypernetwork's branch joins the step's stream at the ends of the graph only: FrameOptimizer._graph_replay_many)
                    self._step(active, count=False, joins=(j == 0, j == k - 1))
            self._graphs[key] = graph
            graph.replay()              # capture does not execute: this replay IS the step(s); a graph's first launch stays inside the capture's turn
        for m in members:
            m.step_index += k

    def run(self, num_steps, steps_per_graph=4, active=None):
        """``num_steps`` steps of the first ``active`` frames; whole groups of ``steps_per_graph`` steps inside a phase replay ONE hipGraph
        (``FrameOptimizer.run``)."""
        active, members = self._members(active)
        remaining = int(num_steps)
        k_many = int(steps_per_graph)
        while remaining > 0:
            residual = self._phase(active)
            to_boundary = remaining if residual else self.config.warmup_steps - members[0].step_index
            k = k_many if k_many > 1 and min(remaining, to_boundary) >= k_many else 1
            if (residual, active, 1) not in self._graphs:
                k = 1                   # the phase's eager steps and its one-step graph come first
            if k == 1:
                self.step(active)
            else:
                key = (residual, active, k)
                if key in self._graphs:
                    with _capture_lock.replaying():
                        self._graphs[key].replay()
                    for m in members:
                        m.step_index += k
                else:
                    self._capture(key, active, k)
            remaining -= k

    def capture_all(self, steps_per_graph=4, actives=None):
        """Every eager warm-up step and every capture ``run()`` would meet -- both phases, the one-step and the ``steps_per_graph``-step graph,
        for every frame count in ``actives`` (default: the full batch) -- now, then every frame is reset.  Returns the number of graphs."""
        cfg = self.config
        k = int(steps_per_graph)
        for active in (actives or [self.size]):
            _, members = self._members(active)
            for phase_start in sorted({0, min(cfg.warmup_steps, cfg.num_steps)}):
                phase_steps = (cfg.warmup_steps if phase_start < cfg.warmup_steps else cfg.num_steps) - phase_start
                if phase_steps <= 0:
                    continue
                with _capture_lock:
                    for m in members:
                        m.step_index = phase_start
                        m.step_tensor.fill_(phase_start)
                for _ in range(min(4, phase_steps)):
                    self.step(active)
                if k > 1 and phase_steps >= 4 + k:
                    self._capture((self._phase(active), active, k), active, k)
        with _capture_lock:
            torch.cuda.current_stream(self.device).synchronize()
        for m in self.frames:
            m.reset(m.inputs)
        return len(self._graphs)

    def reset(self, f, inputs, init_seed=None):
        """Row f starts another frame (``FrameOptimizer.reset``).  False: the frame does not suit the table sampler -- optimise it in a loop of its own."""
        return self.frames[f].reset(inputs, init_seed=init_seed)

    def close(self):
        _destroy_graphs(self._graphs)
        for m in self.frames:
            m.close()

    def __del__(self):
        try:
            _destroy_graphs(self._graphs, wait=False)
        except Exception:
            pass
