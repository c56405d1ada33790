#!/usr/bin/env python3
"""Headline benchmark: rendered rays/s (forward + backward), synthetic KITTI-360-shaped frames.

One step = one pass of the hot path over one dense frame: the fused two-pass render of all V*H*W rays + silhouette BCE
(+ eikonal term for residual fields) + adjoint in one launch (``vsrd_render_silhouette_step`` / ``vsrd_render_residual_step``),
backward through the box decode (and the hypernetwork) to the raw parameters, Adam update.  Inputs are resident in HBM
before the timed region.  Defaults = BASELINE.json config 2 (9 views x 376x1408, 16 instances, 64 samples/ray).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--residual] [--schedule start|mid|end] [--no-culling] ...

Multi-GPU: target frames are independent optimisation problems (README.md:128) -- every rank renders its own frame, there is no
data-path collective ("scaling": "weak"); RCCL carries the barriers and the max-reduce of the timing.  Either the driver starts
the ranks (``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N``: RANK / WORLD_SIZE come from the env), or
``python bench.py --gpus N`` alone spawns its N ranks as child processes itself -- before anything in the parent touches a GPU.
``n_gpus`` in the JSON line is the number of ranks that passed the barrier, not the flag.  Rank 0 prints ONE JSON line.

``--launcher-selftest`` runs the same launch / rendezvous / barrier / max-over-ranks / report path with a sleep instead of the
render step, on gloo without a GPU (tests/test_launcher.py); its line says so and is not a measurement.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TF = 157.3        # MI355X_MICROARCH.md: peak FP32 vector = dense FP32 MFMA peak (64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz)
PEAK_CLOCK_HZ = 2.4e9
NUM_SIMDS = 1024
MFMA_16X16X4_FLOP = 2048    # one v_mfma_f32_16x16x4_f32 wave instruction: 16 x 16 x 4 multiply-adds
MFMA_16X16X32_BF16_FLOP = 16384   # one v_mfma_f32_16x16x32_bf16 wave instruction (the vsrd_split:: kernels of split_front.hip: every MFMA they issue)
BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)

SCHEDULES = {"start": 0.0, "mid": 0.5, "end": 1.0}   # fraction of the 3000 optimisation steps
BASELINE_METRIC = "rendered rays/sec (fwd+bwd) per GPU, KITTI-360 376×1408, 16 instances"   # BASELINE.json:metric


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=10)
    parser.add_argument("--warmup", type=int, default=2)
    parser.add_argument("--views", type=int, default=9)         # 1 target + 8 source views
    parser.add_argument("--height", type=int, default=376)
    parser.add_argument("--width", type=int, default=1408)
    parser.add_argument("--instances", type=int, default=16)
    parser.add_argument("--samples", type=int, default=64)
    parser.add_argument("--schedule", choices=sorted(SCHEDULES), default="mid")
    parser.add_argument("--residual", action="store_true", help="BASELINE config 3: per-instance residual MLP + eikonal loss")
    parser.add_argument("--mlp-split-bf16", action="store_true",
                        help="with --residual: VSRD_FLAG_MLP_SPLIT_BF16 -- the per-instance MLP's products on v_mfma_f32_16x16x32_bf16 with both operands "
                             "split into two bfloat16 parts instead of the exact-fp32 matrix instruction (csrc/residual.h)")
    parser.add_argument("--two-launch", action="store_true",
                        help="render forward, torch loss, render backward as separate launches instead of the fused step kernel")
    parser.add_argument("--no-culling", action="store_true", help="VSRD_FLAG_NO_CULLING: every instance at every sample (worst case)")
    parser.add_argument("--no-skip-misses", action="store_true")
    parser.add_argument("--wave-per-ray", action="store_true",
                        help="VSRD_FLAG_STEP_WAVE_PER_RAY: one ray per wave for the fused box-only step (A/B against four rays per wave)")
    parser.add_argument("--cpu-rays", type=int, default=0, help="rays of the CPU-baseline sample (default 16384; 1408 with --residual)")
    parser.add_argument("--cpu-chunk", type=int, default=0, help="rays per oracle call (default: one image row; 352 with --residual)")
    parser.add_argument("--cpu-threads", type=int, default=16)   # best of {8,16,32,64} on the 2x64-core EPYC 9575F GPU host (r01)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-extra-regimes", action="store_true",
                        help="default run only: skip the `extra_regimes` object (config 3 / config 5 at full size and the reference's native mode, measured after the headline)")
    parser.add_argument("--launcher-selftest", action="store_true", help="no rendering: exercise the N-rank launch/report path (gloo, CPU)")
    parser.add_argument("--ranks-share-gpu", action="store_true",
                        help="TEST ONLY: all ranks render on cuda:0 and rendezvous over gloo -- runs the real step under the multi-rank "
                             "build / barrier / gather / report path on a one-GPU box; the line says that it is not a scaling measurement")
    parser.add_argument("--master-port", type=int, default=0)
    parser.add_argument("--native", action="store_true",
                        help="frames/s instead of rays/s: every rank optimises its shard of synthetic frames in the reference's native mode; "
                             "every other argument goes to vsrd_amd.launcher.main (--gpus, --frames, --frames-in-flight, --num-steps, --views, "
                             "... -- see python -m vsrd_amd.launcher -h)")
    return parser.parse_args(argv)


def schedule_values(fraction):
    """scripts/main.py:420-431: cosine annealing 1.0 -> 0.1 of T and sigma; cosine_ratio = step/num_steps."""
    value = (math.cos(math.pi * fraction) + 1.0) / 2.0 * (1.0 - 0.1) + 0.1
    return dict(temperature=value, std=value, cosine_ratio=fraction)


def workload_key(args):
    """Identifies the per-launch work: a committed rocprof summary is only quoted for the workload it was measured on."""
    return (f"V{args.views}_H{args.height}_W{args.width}_N{args.instances}_S{args.samples}_{args.schedule}_"
            f"{'residual' if args.residual else 'box'}_{'twolaunch' if args.two_launch else 'fused'}"
            f"{'_nocull' if args.no_culling else ''}{'_noskip' if args.no_skip_misses else ''}{'_waveperray' if args.wave_per_ray else ''}"
            f"{'_splitbf16' if args.residual and args.mlp_split_bf16 else ''}")


def describe_workload(args):
    """(metric string, workload description) from the ACTUAL arguments; the BASELINE names only where the sizes are BASELINE's."""
    V, H, W, N, S = args.views, args.height, args.width, args.instances, args.samples
    c2 = (V, H, W, N, S) == (9, 376, 1408, 16, 64)
    c5 = (V, H, W, N, S) == (17, 752, 2816, 64, 128)
    c1 = (V, H, W, N, S) == (3, 128, 128, 4, 32)
    if c2:
        name = "BASELINE config 3" if args.residual else "BASELINE config 2"
    elif c5 and not args.residual:
        name = "BASELINE config 5 (stress) on one GPU"
    elif c1 and not args.residual:
        name = "BASELINE config 1 sizes"
    else:
        name = "custom sizes (not a BASELINE config)"
    metric = BASELINE_METRIC if c2 else f"rendered rays/sec (fwd+bwd) per GPU, {H}×{W}, {N} instances"
    field = "box + per-instance residual-MLP field, eikonal loss" if args.residual else "box-only field"
    if args.residual and args.mlp_split_bf16:
        field += " (MLP products on split-bf16 MFMA: VSRD_FLAG_MLP_SPLIT_BF16)"
    text = (f"{name}: dense frame, {V} views x {H}x{W} = {V * H * W} rays/step/GPU, {N} instances, {S} samples/ray "
            f"(pass 1: {S - 1}, pass 2: {2 * S - 1} points), {field}")
    return metric, text


# ---------------------------------------------------------------------------------------------------------------------
# launch: spawn the ranks ourselves when nobody else did
# ---------------------------------------------------------------------------------------------------------------------

def spawn_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start N copies of this script, one per GPU, with the torchrun environment.
    The parent initialises no GPU (counting devices does not, on this image) and returns the worst child exit code."""
    import socket
    import torch
    if not args.launcher_selftest and not args.ranks_share_gpu:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node has {have} visible GPU(s); one rank per GPU, no oversubscription")
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    children = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    codes = [child.wait() for child in children]
    return max(abs(code) for code in codes)


# ---------------------------------------------------------------------------------------------------------------------
# synthetic frame
# ---------------------------------------------------------------------------------------------------------------------

def kitti_intrinsics(height, width):
    from vsrd_amd import synthetic
    return synthetic.kitti_intrinsics(height, width)


def synthetic_frame(seed, num_views, height, width, num_instances):
    """vsrd_amd.synthetic.synthetic_frame (the frame launcher of the package optimises the same frames): KITTI-360 intrinsics, target
    E = I, sources shifted along z with a small yaw; raw box parameters ~ N(0, 0.5^2) with depth forced into 8-60 m."""
    from vsrd_amd import synthetic
    return synthetic.synthetic_frame(seed, num_views, height, width, num_instances)


def build_union(detector, temperature):
    """The soft-min union of the current boxes as the flat parameter block (what fields.flatten() produces from the
    sdfs.translation(sdfs.rotation(instance_field(sdfs.box(...)))) tree, built here in one cat)."""
    from vsrd_amd import fields
    out = detector()
    # (the orientations are rotation_matrix_y(cos, sin) of the detector's parameters: yaw_gradients -- VSRD_FLAG_YAW_GRADIENTS -- lets
    #  the kernels skip the adjoints of the five matrix entries that function keeps constant; every parameter gradient is unchanged)
    return fields.FieldBlock(fields.pack_instances(out["locations"][0], out["orientations"][0], out["dimensions"][0]),
                             float(temperature), None, None, yaw_gradients=True)


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (SURVEY.md §8d / BASELINE.md §2): the oracle on this host's cores, bounded sample, fwd and bwd separately
# ---------------------------------------------------------------------------------------------------------------------

def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(args, sched, frame, targets_of, hyper_state, cores):
    """``kind: "port"``: oracle/ (the CPU restatement pinned by the goldens) on a fixed subset of the SAME frame: rows of W rays
    through the objects of the target view, issued row by row (the reference's dense idiom, main.py:1011-1023), against the
    frame's own targets.  1 warm-up + 3 timed repeats, the fastest repeat is reported; forward (two passes + loss) and backward
    timed separately with time.perf_counter."""
    import torch
    from oracle import fields as ofields, rendering as orendering, geometry as ogeometry, losses as olosses
    torch.set_num_threads(cores)
    K, E, raw_loc, raw_dim, raw_ori = frame
    H, W, N, S = args.height, args.width, args.instances, args.samples
    num_rays = args.cpu_rays or (1408 if args.residual else 16384)
    cam, dirs = ogeometry.ray_casting((H, W), K[:1], E[:1])
    first = int(H * 0.55) * W                                        # rows through the objects
    first = max(0, min(first, H * W - num_rays))
    # residual fields: a quarter row per call -- the oracle carries forward-mode tangents through the MLP, and at a full row its
    # autograd state (tens of GB) is paged in afresh on every call (measured: 17 rays/s at 1408 rays per call, 75 at 352)
    chunk = args.cpu_chunk or (352 if args.residual else W)
    chunks = [(start, min(start + chunk, first + num_rays)) for start in range(first, first + num_rays, chunk)]
    flat_dirs = dirs[0].reshape(-1, 3)
    targets = targets_of(first, first + num_rays)                   # [num_rays, N] of view 0, from the device
    g = torch.Generator().manual_seed(1)
    uniforms = [(torch.rand(b - a, S, generator=g), torch.rand(b - a, S, generator=g)) for a, b in chunks]
    raws = [t[0].clone().requires_grad_(True) for t in (raw_loc, raw_dim, raw_ori)]
    mlp = None
    if args.residual:
        from vsrd_amd import models
        hyper = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256])
        hyper.load_state_dict(hyper_state["hyper"])
        embeddings = hyper_state["embeddings"].clone().requires_grad_(True)

    def one_pass():
        forward = backward = 0.0
        for (a, b), (uc, uf) in zip(chunks, uniforms):
            t0 = time.perf_counter()
            loc, dim, rot, _ = ogeometry.decode_box_parameters(*raws)
            weights = hyper(embeddings)[0] if args.residual else None
            union = ofields.InstanceUnion(loc, rot, dim, sched["temperature"], weights)
            out = orendering.hierarchical_render(union, cam[0], flat_dirs[a:b], (0.0, 100.0), S, sched["std"], sched["cosine_ratio"], uc, uf)
            loss = olosses.silhouette_loss(out.labels, targets[a - first:b - first])
            if args.residual:
                loss = loss + 0.01 * olosses.eikonal_loss(out.gradients)
            t1 = time.perf_counter()
            loss.backward()
            t2 = time.perf_counter()
            forward += t1 - t0
            backward += t2 - t1
        return forward, backward

    one_pass()                                                       # warm-up
    repeats = [one_pass() for _ in range(3)]
    forward, backward = min(repeats, key=lambda fb: fb[0] + fb[1])
    return dict(value=num_rays / (forward + backward), unit="rays/s", cores=cores, kind="port", cpu=cpu_model_name(),
                forward_s=forward, backward_s=backward, repeats=3,
                sample=f"{num_rays} rays of the target view in {len(chunks)} calls of <= {chunk} rays (fwd+bwd, N={N}, S={S}"
                       f"{', residual MLP + eikonal' if args.residual else ''}), the frame's own targets, oracle/ on {cores} threads, "
                       "fastest of 3 repeats after 1 warm-up")


# ---------------------------------------------------------------------------------------------------------------------
# committed rocprofv3 summaries (profiles/rNN*/counters.json, written by tools/summarize_profile.py)
# ---------------------------------------------------------------------------------------------------------------------

def committed_counters(kernel_symbols, key):
    """Per-STEP PMC counters summed over the kernels of one step (`kernel_symbols`: substrings of their names) from the newest committed
    profile of exactly this workload, or (None, None).  A step of the fused box kernel is one launch; a residual step is a front kernel
    and an MLP-adjoint kernel per chunk of rays plus the row reductions (tools/summarize_profile.py stores, per kernel, the sum over
    its dispatches divided by the steps of the profiled command as `<counter>_per_step`)."""
    import glob
    if isinstance(kernel_symbols, str):
        kernel_symbols = [kernel_symbols]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "counters.json")), reverse=True):
        try:
            data = json.load(open(path))
        except (OSError, ValueError):
            continue
        if data.get("workload_key") != key:
            continue
        total, used = {}, []
        for name, entry in data.get("kernels", {}).items():
            if not any(symbol in name for symbol in kernel_symbols):
                continue
            used.append(name)
            for counter, value in entry.items():
                if counter.endswith("_per_step"):
                    total[counter[:-len("_per_step")]] = total.get(counter[:-len("_per_step")], 0.0) + value
            if not any(c.endswith("_per_step") for c in entry):          # profiles written before round 3: one launch per step
                for counter, value in entry.items():
                    if isinstance(value, float) and counter != "rocprof_avg_ms":
                        total[counter] = total.get(counter, 0.0) + value
        if used:
            total["kernels"] = used
            # the matrix instructions of a step by kind: the kernels of split_front.hip (namespace vsrd_split) execute v_mfma_f32_16x16x32_bf16
            # only when launched with VSRD_FLAG_MLP_SPLIT_BF16 -- the only way they are launched (their residual_forward keeps the fp32 form
            # behind a wave-uniform branch on kMlpSplitBit that is never taken; `hipcc -S --cuda-device-only split_front.hip | grep v_mfma`:
            # residual_mlp_adjoint_split_kernel 72 bf16 / 0 fp32) -- every other kernel of the library v_mfma_f32_16x16x4_f32 only
            per_kernel = {name: data["kernels"][name].get("SQ_INSTS_MFMA_per_step", data["kernels"][name].get("SQ_INSTS_MFMA", 0.0)) for name in used}
            total["mfma_bf16_16x16x32"] = sum(v for name, v in per_kernel.items() if "vsrd_split::" in name)
            total["mfma_f32_16x16x4"] = sum(v for name, v in per_kernel.items() if "vsrd_split::" not in name)
            return total, os.path.relpath(path, ROOT)
    return None, None


def executed_view(counters, launch_ms):
    """Hardware view of one step from its PMC counters (wave-instruction counts summed over the step's kernels) and the live step duration.
    Two pipes, two peaks: fp32 vector arithmetic against the 157.3 TFLOP/s vector peak, matrix instructions priced BY INSTRUCTION --
    v_mfma_f32_16x16x4_f32 = 2 048 flop against the same 157.3 (it runs on the vector datapath), v_mfma_f32_16x16x32_bf16 = 16 384 flop
    against the dense bf16 peak of 2.5 PFLOP/s."""
    if not counters or "SQ_INSTS_VALU" not in counters:
        return None
    seconds = launch_ms * 1e-3
    fma, mul, add = (counters.get(k, 0.0) for k in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32"))
    trans = counters.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    mfma = counters.get("SQ_INSTS_MFMA", counters.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0))
    mfma_bf16 = counters.get("mfma_bf16_16x16x32", 0.0)
    mfma_f32 = counters.get("mfma_f32_16x16x4", mfma - mfma_bf16)
    valu_flop = (2.0 * fma + mul + add + trans) * 64.0
    f32_flop, bf16_flop = mfma_f32 * MFMA_16X16X4_FLOP, mfma_bf16 * MFMA_16X16X32_BF16_FLOP
    simd_cycles = NUM_SIMDS * PEAK_CLOCK_HZ * seconds
    # a wave64 fp32 VALU instruction occupies its SIMD's issue port for 2 cycles at peak (32 lanes/clk); MFMA 16x16x4 f32: 8 passes x 4 clk;
    # MFMA 16x16x32 bf16: 17-19 cycles measured (profiles/r05/mfma_bf16_overlap.txt)
    valu_slots = counters["SQ_INSTS_VALU"] - mfma
    matrix_cycles = 32.0 * mfma_f32 + 18.0 * mfma_bf16
    view = {"valu_tflops": valu_flop / seconds / 1e12, "valu_frac_of_fp32_vector_peak": valu_flop / seconds / 1e12 / FP32_PEAK_TF,
            "mfma_f32_tflops": f32_flop / seconds / 1e12, "mfma_bf16_tflops": bf16_flop / seconds / 1e12,
            "mfma_bf16_frac_of_bf16_peak": bf16_flop / seconds / 1e12 / BF16_MFMA_PEAK_TF, "bf16_peak_tflops": BF16_MFMA_PEAK_TF,
            "mfma_tflops": (f32_flop + bf16_flop) / seconds / 1e12,
            # the fp32 datapath (vector arithmetic + the fp32 matrix instruction, which shares it) against its 157.3 TFLOP/s
            "tflops": (valu_flop + f32_flop) / seconds / 1e12, "frac": (valu_flop + f32_flop) / seconds / 1e12 / FP32_PEAK_TF,
            "valu_issue_utilisation": 2.0 * valu_slots / simd_cycles,
            # what the SIMDs spent per vector instruction (matrix time taken off), next to what a stream of plain
            # multiply-adds costs at four waves per SIMD on this part (profiles/r01/op_rates.txt: 2.7-3.0, not the nominal 2)
            "cycles_per_valu_instruction": max(simd_cycles - matrix_cycles, 0.0) / max(valu_slots, 1.0),
            "measured_cycles_per_plain_fma": 2.9,
            "mfma_utilisation": f32_flop / seconds / 1e12 / FP32_PEAK_TF,
            "fma_share_of_fp32_ops": fma / max(fma + mul + add, 1.0),
            "valu_wave_instructions": counters["SQ_INSTS_VALU"], "mfma_wave_instructions": mfma,
            "mfma_f32_16x16x4_wave_instructions": mfma_f32, "mfma_bf16_16x16x32_wave_instructions": mfma_bf16,
            "lds_bank_conflict_share": (counters["SQ_LDS_BANK_CONFLICT"] / counters["SQ_LDS_IDX_ACTIVE"]
                                        if counters.get("SQ_LDS_IDX_ACTIVE") else None)}
    if counters.get("SQ_ACTIVE_INST_VALU"):
        # SQ_ACTIVE_INST_VALU counts, per SIMD, the quad-cycles in which a vector instruction (matrix ones included) is being issued:
        # x 4 over the step's SIMD-cycles = the share of time the vector issue port is busy
        view["valu_active_fraction"] = 4.0 * counters["SQ_ACTIVE_INST_VALU"] / simd_cycles
    return view


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------

def run_rank(args):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    selftest = args.launcher_selftest
    use_gpu = torch.cuda.is_available() and not selftest
    if not use_gpu and not selftest:
        raise SystemExit("bench.py measures the HIP path and needs a HIP device: there is no CPU fallback (the CPU oracle is only the "
                         "`cpu_baseline` leg).  `--launcher-selftest` exercises the multi-rank launch path without rendering.")
    share = args.ranks_share_gpu and use_gpu
    dev = torch.device("cuda", 0 if share else local_rank) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(dev)
    dist = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_gpu and not share:
            dist.init_process_group(backend="nccl", device_id=dev)     # RCCL; barrier / max-reduce / gather of the timings only
        else:
            dist.init_process_group(backend="gloo")                    # (RCCL refuses two ranks on one device)

    def fence():
        if distributed:
            dist.barrier()
        if use_gpu:
            torch.cuda.synchronize()

    sched = schedule_values(SCHEDULES[args.schedule])
    V, H, W, N, S = args.views, args.height, args.width, args.instances, args.samples
    R = V * H * W
    loss = None
    kernels = {}
    if selftest:
        def step(index):
            time.sleep(0.002 * (1 + rank))          # ranks differ: the reported time must be the slowest rank's
    else:
        import __graft_entry__
        if rank == 0:
            __graft_entry__.build()
        if distributed:
            dist.barrier()
        from vsrd_amd import models, rendering, profiling
        from vsrd_amd.rendering import renderers
        renderers.CULLING = not args.no_culling
        renderers.MLP_SPLIT_BF16 = renderers.MLP_SPLIT_BF16 or (args.residual and args.mlp_split_bf16)
        renderers.STEP_WAVE_PER_RAY = renderers.STEP_WAVE_PER_RAY or args.wave_per_ray
        # one frame per rank, and every rank a replica of the same synthetic frame (the scaling curve then isolates the
        # launcher; frames of a real shard differ in cost, which is load imbalance, not scaling); the Philox streams differ by rank
        frame = synthetic_frame(seed=0, num_views=V, height=H, width=W, num_instances=N)
        K, E, raw_loc, raw_dim, raw_ori = frame
        # ---- resident inputs (untimed) -------------------------------------------------------------------------------
        cam, dirs = rendering.ray_casting((H, W), K.to(dev), E.to(dev))                 # [V,3], [V,H,W,3]
        directions = dirs.reshape(-1, 3).contiguous()
        origins = cam[:, None, None, :].expand(V, H, W, 3).reshape(-1, 3).contiguous()   # per-pixel, as main.py:289-296
        detector = models.BoxParameters3D(1, N).to(dev)
        with torch.no_grad():
            detector.locations.copy_(raw_loc); detector.dimensions.copy_(raw_dim); detector.orientations.copy_(raw_ori)
            # targets: soft silhouettes of a perturbed copy of the boxes at the final (sharp) schedule
            perturbed = models.BoxParameters3D(1, N).to(dev)
            g = torch.Generator().manual_seed(1000 + rank)
            perturbed.locations.copy_(raw_loc + torch.randn(raw_loc.shape, generator=g) * 0.05)
            perturbed.dimensions.copy_(raw_dim + torch.randn(raw_dim.shape, generator=g) * 0.2)
            perturbed.orientations.copy_(raw_ori + torch.randn(raw_ori.shape, generator=g) * 0.1)
            targets = rendering.render_hierarchical(build_union(perturbed, 0.1), origins, directions, (0.0, 100.0), S, 0.1, 1.0,
                                                    seed=99, skip_exact_misses=True)["labels"].clamp(0.0, 1.0).contiguous()
        params = [detector.locations, detector.dimensions, detector.orientations]
        hyper = None
        hyper_state = None
        if args.residual:
            torch.manual_seed(rank)
            hyper = models.HyperDistanceField(48, [16, 16, 16, 16], 256, [256, 256, 256, 256]).to(dev)
            params += [detector.embeddings, *hyper.parameters()]
            hyper_state = dict(hyper={k: v.detach().cpu().clone() for k, v in hyper.state_dict().items()},
                               embeddings=detector.embeddings.detach().cpu().clone())
        optimizer = torch.optim.Adam(params, lr=1e-2)
        skip = not args.no_skip_misses and not args.residual     # eikonal needs every ray's gradients
        fused = not args.two_launch                               # one launch: render + silhouette BCE (+ eikonal) + adjoint

        def step(index):
            optimizer.zero_grad(set_to_none=True)
            union = build_union(detector, sched["temperature"])
            if hyper is not None:
                union.mlp_weights = hyper(detector.embeddings)[0].contiguous()
            if fused:
                value = rendering.silhouette_step(union, origins, directions, targets, (0.0, 100.0), S, sched["std"], sched["cosine_ratio"],
                                                  seed=rank, stream_offset=index, skip_exact_misses=skip, eikonal_ratio=0.01 if hyper is not None else 0.0)
            else:
                out = rendering.render_hierarchical(union, origins, directions, (0.0, 100.0), S, sched["std"], sched["cosine_ratio"],
                                                    seed=rank, stream_offset=index, skip_exact_misses=skip, return_gradients=hyper is not None)
                value = torch.nn.functional.binary_cross_entropy(out["labels"].clamp(1.0e-6, 1.0 - 1.0e-6), targets, reduction="none").mean()
                if hyper is not None:
                    value = value + 0.01 * ((out["gradients"].norm(dim=-1) - 1.0) ** 2).mean()
            value.backward()
            optimizer.step()
            return value

    # ---- W warm-up steps, then exactly K timed steps between barrier + synchronize on both sides -------------------------
    for i in range(args.warmup):
        step(i)
    fence()
    if selftest:
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        own = time.perf_counter() - t0                # this rank's own work, before it waits for the others
        fence()
        elapsed = time.perf_counter() - t0
    else:
        with profiling.kernel_timer() as timer:
            t0 = time.perf_counter()
            for i in range(args.steps):
                loss = step(args.warmup + i)
            torch.cuda.synchronize()
            own = time.perf_counter() - t0            # this rank's own work, before it waits for the others
            fence()
            elapsed = time.perf_counter() - t0
        kernels = timer.summary()
    per_rank_ms = [own / args.steps * 1e3]
    ranks_through = 1
    if distributed:
        wire = torch.device("cpu") if share else dev
        mine = torch.tensor([elapsed, own], device=wire, dtype=torch.float64)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank_ms = [float(t[1].item()) / args.steps * 1e3 for t in gathered]
        elapsed = max(float(t[0].item()) for t in gathered)           # MAX over ranks of the barrier-to-barrier time
        count = torch.ones(1, device=wire, dtype=torch.float64)
        dist.all_reduce(count)                                        # ranks that got here = ranks that passed every barrier
        ranks_through = int(round(float(count.item())))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        metric, workload = describe_workload(args)
        result = {
            "metric": metric, "value": ranks_through * R * args.steps / elapsed, "unit": "rays/s", "n_gpus": ranks_through,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "per_rank_ms_per_step": per_rank_ms,
        }
        if selftest:
            result.update(metric="launcher selftest: no rendering, NOT a measurement", value=0.0, unit="none",
                          config={"workload": "sleep(2 ms x (1 + rank)) per step on gloo/CPU: exercises spawn, rendezvous, barriers, "
                                              "max over ranks and the report only", "requested_gpus": args.gpus})
        else:
            result["config"] = {
                "workload": workload, "workload_key": workload_key(args),
                "schedule": f"{args.schedule}: T=std={sched['std']:.3f}, cosine_ratio={sched['cosine_ratio']:.2f}",
                "culling": not args.no_culling, "skip_exact_misses": skip, "rng": "in-kernel Philox4x32-10", "rays_per_gpu": R,
                "requested_gpus": args.gpus, "parallelism": f"frames sharded over {ranks_through} rank(s) (replicas of one synthetic frame), no data-path collective",
                "loss": ("silhouette BCE" + (" + 0.01 eikonal" if args.residual else "") + (" fused into the render kernel" if fused else " (torch elementwise)")) + " + Adam",
                "launches_per_step": "1 fused (render + loss + adjoint) + partial reductions" if fused else "forward, torch loss, backward",
                "final_loss": float(loss.detach()), "target_empty_fraction": float((targets.sum(-1) == 0).float().mean()),
            }
            if share:
                result["metric"] = "ranks share ONE GPU (--ranks-share-gpu): a test of the multi-rank path, NOT a scaling measurement; " + result["metric"]
            result.update(rooflines(args, kernels, R, fused))
            if world == 1 and not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(args, sched, frame, lambda a, b: targets[a:b].cpu(), hyper_state,
                                                      min(args.cpu_threads, os.cpu_count() or 1))
            default_workload = (V, H, W, N, S) == (9, 376, 1408, 16, 64) and not args.residual and fused and args.schedule == "mid" \
                and not args.no_culling and not args.no_skip_misses and not args.wave_per_ray
            if world == 1 and default_workload and not args.no_extra_regimes:
                del targets, directions, origins
                torch.cuda.empty_cache()
                result["extra_regimes"] = extra_regimes()
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def step_kernel_symbols(args, fused):
    """(C-ABI entry that dominates a step, names of the kernels it launches) -- what `roofline` is about."""
    N, S = args.instances, args.samples
    if not fused:
        return None, None
    if args.residual:
        return "vsrd_render_residual_step", ["residual_step_front_kernel", "residual_step_pair_kernel", "residual_mlp_adjoint",      # (..._kernel and vsrd_split::..._split_kernel)
                                             "render_residual_step_kernel", "reduce_item_rows_kernel", "reduce_item_segments_kernel", "pack_mlp_images_kernel"]
    from vsrd_amd.rendering import renderers
    dense = not renderers.STEP_WAVE_PER_RAY                              # api.hip: vsrd_render_silhouette_step
    if dense and S <= 64 and N <= 16:
        return "vsrd_render_silhouette_step", ["render_silhouette_quad_kernel"]          # four rays per wave
    if dense and S <= 128 and N <= 64:
        return "vsrd_render_silhouette_step", ["render_silhouette_pair_kernel"]          # two rays per wave
    return "vsrd_render_silhouette_step", ["render_silhouette_kernel<"]


def rooflines(args, kernels, R, fused):
    """`roofline` (the mandated object, for the dominant kernel(s) of a step) and `roofline_valu`, from launch durations measured live
    with HIP events on the launch stream (vsrd_amd/profiling.py) and the committed PMC counters of exactly this workload.

    Algorithmic bytes per ray: fused step = direction 12 + targets 4N (labels, label adjoints and the
    sorted distances never leave the chip); two-launch path = 12 + 4N per launch.
    `frac` is only ever a measured quantity over a hardware peak: bytes / s over the HBM peak, or EXECUTED flops (PMC wave-instruction
    counts x 64 lanes; MFMA 16x16x4 = 2048 flop) / s over the 157.3 TFLOP/s fp32 peak.  The SURVEY §8d flop model --
    3.5 (3S-2)(63N+45) per ray, + (3S-2) N 2*1617*9 for residual fields -- counts every instance at every sample, i.e. work the kernels
    cull: it is reported as `work_equivalent_tflops` (it can exceed the peak; it is a speed-up over a kernel that culls nothing, not a
    utilisation)."""
    N, S = args.instances, args.samples
    if fused:
        dominant, symbols = step_kernel_symbols(args, fused)
        fwd_n, fwd_ms = kernels[dominant]
        bwd_n, bwd_ms = 0, 0.0
        dom_ms = fwd_ms
    else:
        fwd_n, fwd_ms = kernels["vsrd_render_hierarchical_forward"]
        bwd_n, bwd_ms = kernels["vsrd_render_backward"]
        dominant, dom_ms, symbols = ("vsrd_render_backward", bwd_ms, ["render_backward_kernel"]) if bwd_ms >= fwd_ms else \
                                    ("vsrd_render_hierarchical_forward", fwd_ms, ["render_hierarchical_kernel"])
    dom_bytes = 12 + 4 * N
    counters, source = committed_counters(symbols, workload_key(args))
    traffic = None
    if counters and "FETCH_SIZE_bytes" in counters and "WRITE_SIZE_bytes" in counters:
        traffic = counters["FETCH_SIZE_bytes"] + counters["WRITE_SIZE_bytes"]
    executed = executed_view(counters, dom_ms)
    box_flop = 3.5 * (3 * S - 2) * (63 * N + 45)
    mlp_flop = (3 * S - 2) * N * 2 * 1617 * 9 if args.residual else 0.0
    total_ms = fwd_ms + bwd_ms
    achieved_gbs = R * dom_bytes / (dom_ms * 1e-3) / 1e9
    # SURVEY.md section 8d's API-faithful figure: B_api = 24 + 12 N for the two-launch path, + 24 (2S - 1) when the eikonal gradients cross the
    # boundary (config 3: 3 264 B per ray at N = 16, S = 64); `traffic_over_api_bytes` is the PMC traffic of a step against it
    api_bytes = 24 + 12 * N + (24 * (2 * S - 1) if args.residual else 0)
    hbm = {"bound": "hbm", "kernel": dominant, "kernels": counters.get("kernels") if counters else symbols, "achieved": achieved_gbs, "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source, "algorithmic_bytes_per_ray": dom_bytes,
           "launch_ms": dom_ms, "traffic_bytes_per_ray": traffic / R if traffic else None,
           "api_faithful_bytes_per_ray": api_bytes, "traffic_over_api_bytes": traffic / (R * api_bytes) if traffic else None,
           "traffic_over_algorithmic_bytes": traffic / (R * dom_bytes) if traffic else None,
           "note": "the fused path is compute bound (arithmetic intensity ~1e4 flop/B), not HBM bound (SURVEY.md §8d); see roofline_valu"}
    model_tf = R * (box_flop + mlp_flop) / (total_ms * 1e-3) / 1e12
    valu = {"bound": ("fp32 VALU (157.3 TFLOP/s); the matrix products on bf16 MFMA are priced against the 2.5 PFLOP/s bf16 peak in `executed`"
                      if args.residual and args.mlp_split_bf16 else
                      "fp32 VALU + fp32 MFMA (one datapath, 157.3 TFLOP/s)" if args.residual else "fp32-valu"),
            "achieved": executed["tflops"] if executed else None, "peak": FP32_PEAK_TF, "unit": "TFLOP/s", "frac": executed["frac"] if executed else None,
            "basis": ("executed flops: committed PMC counters of this workload (per step, all kernels of the step) / live step duration" if executed
                      else "no committed counters for this workload: no utilisation figure (see work_equivalent_tflops)"),
            "executed": executed, "executed_source": source,
            "work_equivalent_tflops": model_tf, "work_equivalent_flop_per_ray": box_flop + mlp_flop,
            "work_equivalent_note": "SURVEY §8d flop model: every instance at every sample, no culling -- a work rate, NOT a fraction of the peak",
            "forward_ms": fwd_ms, "backward_ms": bwd_ms, "launches": [fwd_n, bwd_n]}
    return {"roofline": hbm, "roofline_valu": valu}


# ---------------------------------------------------------------------------------------------------------------------
# extra regimes of the default run: measured AFTER the headline, each in a child process
# ---------------------------------------------------------------------------------------------------------------------

def extra_regimes():
    """Config 3 and config 5 at full size (bench.py itself with other sizes) and the reference's own regime -- 1000 importance-sampled
    rays x 100 samples per step, 3000 steps per frame (scripts/main.py:525-578, 620-651; README.md:128: "about 15 minutes" per frame on
    a V100) -- through FrameOptimizer as a replayed hipGraph (tools/native_mode_bench.py).  Children start after the parent's timed
    region and its report are complete; the parent only waits."""
    def child(cmd, keep, timeout=900):
        t0 = time.time()
        try:
            out = subprocess.run([sys.executable, *cmd], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
            lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not lines:
                return {"error": (out.stderr or out.stdout)[-300:], "command": "python " + " ".join(cmd)}
            record = json.loads(lines[-1])
        except (subprocess.TimeoutExpired, ValueError) as exc:
            return {"error": repr(exc)[:300], "command": "python " + " ".join(cmd)}
        small = {k: record[k] for k in keep if k in record}
        if "config" in record:
            small["workload"] = record["config"].get("workload")
            small["schedule"] = record["config"].get("schedule")
        for key in ("roofline", "roofline_valu"):
            if key in record:
                small[key] = {k: record[key].get(k) for k in ("bound", "kernel", "kernels", "achieved", "peak", "unit", "frac", "traffic", "work_equivalent_tflops",
                                                              "launch_ms", "traffic_over_api_bytes", "api_faithful_bytes_per_ray", "executed_source")
                              if k in record[key]}
                executed = record[key].get("executed")
                if executed:       # the utilisation figures a reader recomputes from profiles/: both pipes against their own peaks
                    small[key]["executed"] = {k: executed.get(k) for k in ("valu_tflops", "valu_frac_of_fp32_vector_peak", "mfma_f32_tflops", "mfma_bf16_tflops",
                                                                           "mfma_bf16_frac_of_bf16_peak", "valu_active_fraction", "valu_issue_utilisation",
                                                                           "mfma_f32_16x16x4_wave_instructions", "mfma_bf16_16x16x32_wave_instructions")}
        small["command"] = "python " + " ".join(cmd)
        small["wall_s"] = round(time.time() - t0, 1)
        return small
    dense = ["value", "unit", "ms_per_step", "steps", "warmup", "n_gpus"]
    native = ["phase", "graph", "steps_per_s", "ms_per_step", "seconds_per_3000_step_frame", "seconds_per_frame", "warmup_phase_seconds",
              "residual_phase_seconds", "steps", "rays_per_step", "samples_per_ray", "views", "instances", "final_loss", "mlp_products"]
    frames = ["value", "unit", "n_gpus", "frames", "seconds", "frames_per_s_per_gpu", "per_rank_seconds", "seconds_per_frame_per_gpu", "capture_seconds_per_frame",
              "restarts", "max_restarts", "mean_final_loss", "ranks", "procs_per_gpu", "frames_in_flight_per_process", "control_plane", "frame_batch", "queue",
              "mlp_products", "slot_setup_seconds", "graphs_per_slot", "frames_outside_slots", "frames_with_unhealthy_draws"]
    base = ["bench.py", "--no-cpu-baseline", "--no-extra-regimes"]
    tool = os.path.join("tools", "native_mode_bench.py")
    regimes = {
        "note": "measured after the headline's timed region, one child process each; config 2 above stays the metric's workload",
        "config3_full_size": child(base + ["--residual", "--steps", "3", "--warmup", "1"], dense),
        "config3_full_size_split_bf16": child(base + ["--residual", "--mlp-split-bf16", "--steps", "3", "--warmup", "1"], dense),
        "config5_one_gpu": child(base + ["--views", "17", "--height", "752", "--width", "2816", "--instances", "64", "--samples", "128", "--steps", "2", "--warmup", "1"], dense),
        "native_graph_box_only": child([tool, "--graph", "--steps", "300", "--json"], native),
        "native_graph_residual": child([tool, "--graph", "--residual", "--steps", "300", "--json"], native),
        "native_graph_whole_frame": child([tool, "--graph", "--whole-frame", "--json"], native),
        "native_graph_whole_frame_fp32_mlp": child([tool, "--graph", "--whole-frame", "--fp32-mlp", "--json"], native),
        "native_graph_whole_frame_batch16": child([tool, "--graph", "--whole-frame", "--batch", "16", "--json"], native + ["frame_batch", "seconds_per_batch", "setup_seconds"]),
        # frames/s, the unit the reference shards (README.md:128): whole frames through the frame launcher with its default layout,
        # checkpoints included (python bench.py --native = python -m vsrd_amd.launcher; on a node: --gpus 8).  Round 6: ONE rank process per GPU (RCCL
        # control plane) that steps a batch of frames together (optimization.FrameBatch; --frame-batch, default 16); the slots' set-up before the clock is
        # `slot_setup_seconds`
        "native_frames_per_s_one_process": child(["bench.py", "--native", "--gpus", "1", "--procs-per-gpu", "1", "--frames", "32"], frames, timeout=1200),
        "native_frames_per_s_one_process_fp32_mlp": child(["bench.py", "--native", "--gpus", "1", "--procs-per-gpu", "1", "--frames", "32", "--fp32-mlp"], frames, timeout=1200),
        # round 5's default layout, for comparison: two rank processes (gloo) with one frame each
        "native_frames_per_s_two_processes_r05": child(["bench.py", "--native", "--gpus", "1", "--procs-per-gpu", "2", "--frame-batch", "1", "--queue", "static", "--frames", "12"],
                                                       frames, timeout=1200),
    }
    # (`python bench.py --native --gpus 1` IS the one-process layout since round 6: the key of rounds 4-5 stays, it is the same record)
    regimes["native_frames_per_s"] = dict(regimes["native_frames_per_s_one_process"], note="the launcher's default layout: the record of native_frames_per_s_one_process")
    return regimes


def main():
    if "--native" in sys.argv[1:]:      # the frames/s entry point: same launch contract (spawns its ranks itself, or is a rank under torchrun)
        from vsrd_amd import launcher
        return launcher.main([a for a in sys.argv[1:] if a != "--native"])
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    run_rank(args)


if __name__ == "__main__":
    sys.exit(main() or 0)
