"""ctypes binding of libvsrd_hip.so (include/vsrd_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no CPU
fallback: if the shared object is missing, or a tensor is not on a HIP device, every entry
point raises.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VSRD_HIP_LIBRARY: an experiment build of the same ABI (tools/phase_timers.py, A/B macros); the product path is the in-tree default
LIBRARY_PATH = os.environ.get("VSRD_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libvsrd_hip.so")

ABI_VERSION = 8
MAX_INSTANCES = 64
MAX_SAMPLES = 256
INSTANCE_STRIDE = 16
MLP_WEIGHTS = 1617
FLAG_FINE_UNIFORMS_SORTED = 1
FLAG_SKIP_EXACT_MISSES = 2
FLAG_NO_CULLING = 4
FLAG_MLP_WEIGHTS_CENTRED = 8
FLAG_RUNNING_MINIMUM = 16
FLAG_GENERAL_ROTATIONS = 32
FLAG_RESIDUAL_SINGLE_KERNEL = 64
FLAG_RESIDUAL_WAVE_PER_RAY = 128
FLAG_STEP_WAVE_PER_RAY = 256
FLAG_STEP_SPLIT_RAY = 512
FLAG_YAW_GRADIENTS = 1024
FLAG_MLP_SPLIT_BF16 = 2048

c_float_p = ctypes.c_void_p  # device pointers travel as integers


class Field(ctypes.Structure):
    _fields_ = [
        ("num_instances", ctypes.c_int32),
        ("temperature", ctypes.c_float),
        ("instances", ctypes.c_void_p),
        ("mlp_weights", ctypes.c_void_p),
    ]


class RenderConfig(ctypes.Structure):
    _fields_ = [
        ("num_rays", ctypes.c_int32),
        ("num_samples", ctypes.c_int32),
        ("distance_near", ctypes.c_float),
        ("distance_far", ctypes.c_float),
        ("sdf_std_deviation", ctypes.c_float),
        ("cosine_ratio", ctypes.c_float),
        ("epsilon", ctypes.c_float),
        ("origin_stride", ctypes.c_int32),
        ("seed", ctypes.c_uint64),
        ("stream_offset", ctypes.c_uint64),
        ("flags", ctypes.c_uint32),
        ("device_schedule", ctypes.c_void_p),
        ("device_stream_offset", ctypes.c_void_p),
        ("ray_indices", ctypes.c_void_p),
        ("rays_per_origin", ctypes.c_int32),
        ("target_columns", ctypes.c_void_p),
        ("target_stride", ctypes.c_int32),
        ("out_distances", ctypes.c_void_p),
        ("out_coarse_weights", ctypes.c_void_p),
        ("out_u_coarse", ctypes.c_void_p),
        ("out_u_fine", ctypes.c_void_p),
        ("num_frames", ctypes.c_int32),          # frame batch (ABI 8): 0 or 1 = one frame
        ("frame_stride", ctypes.c_int64),        # bytes between the frames' copies of every buffer of the call
        ("adjoint_slots_per_item", ctypes.c_int32),   # vsrd_render_residual_step: slots per work item of the MLP adjoint (0: planned)
    ]


class FrameConfig(ctypes.Structure):
    _fields_ = [
        ("num_boxes", ctypes.c_int32),
        ("num_views", ctypes.c_int32),
        ("height", ctypes.c_float),
        ("width", ctypes.c_float),
        ("epsilon", ctypes.c_float),
        ("location_lo", ctypes.c_float * 3),
        ("location_hi", ctypes.c_float * 3),
        ("dimension_lo", ctypes.c_float * 3),
        ("dimension_hi", ctypes.c_float * 3),
        ("num_steps", ctypes.c_int32),
        ("max_temperature", ctypes.c_float),
        ("min_temperature", ctypes.c_float),
        ("max_std", ctypes.c_float),
        ("min_std", ctypes.c_float),
        ("weight_iou", ctypes.c_float),
        ("weight_l1", ctypes.c_float),
        ("weight_silhouette", ctypes.c_float),
        ("beta1", ctypes.c_float),
        ("beta2", ctypes.c_float),
        ("adam_epsilon", ctypes.c_float),
        ("lr_gamma", ctypes.c_float),
        ("num_frames", ctypes.c_int32),
        ("frame_stride", ctypes.c_int64),
    ]


class AdamTensors(ctypes.Structure):
    _fields_ = [
        ("parameter", ctypes.c_void_p),
        ("exp_avg", ctypes.c_void_p),
        ("exp_avg_sq", ctypes.c_void_p),
        ("step", ctypes.c_void_p),
        ("learning_rate", ctypes.c_void_p),
    ]


HYPER_LAYERS = 5


class Hypernetwork(ctypes.Structure):
    _fields_ = [
        ("num_instances", ctypes.c_int32),
        ("num_outputs", ctypes.c_int32),
        ("beta1", ctypes.c_float),
        ("beta2", ctypes.c_float),
        ("adam_epsilon", ctypes.c_float),
        ("lr_gamma", ctypes.c_float),
        ("embeddings", AdamTensors),
        ("weight_v", AdamTensors * HYPER_LAYERS),
        ("weight_g", AdamTensors * HYPER_LAYERS),
        ("bias", AdamTensors * HYPER_LAYERS),
        ("norm_weight", AdamTensors * (HYPER_LAYERS - 1)),
        ("norm_bias", AdamTensors * (HYPER_LAYERS - 1)),
        ("num_frames", ctypes.c_int32),
        ("frame_stride", ctypes.c_int64),
    ]


# symbol -> (restype, argtypes); mirrors include/vsrd_hip.h one to one
SIGNATURES = {
    "vsrd_abi_version": (ctypes.c_int32, []),
    "vsrd_error_string": (ctypes.c_char_p, [ctypes.c_int32]),
    "vsrd_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    "vsrd_render_backward_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "vsrd_ray_directions": (ctypes.c_int32, [c_float_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_float_p, ctypes.c_void_p]),
    "vsrd_field_eval": (ctypes.c_int32, [ctypes.POINTER(Field), c_float_p, ctypes.c_int64, c_float_p, c_float_p, c_float_p,
                                         ctypes.c_int32, ctypes.c_void_p]),
    "vsrd_sphere_trace": (ctypes.c_int32, [ctypes.POINTER(Field), c_float_p, ctypes.c_int32, c_float_p, ctypes.c_void_p, ctypes.c_int64,
                                           ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_int32, ctypes.c_int32,
                                           c_float_p, ctypes.c_void_p, ctypes.c_void_p]),
    "vsrd_polygon_soft_masks": (ctypes.c_int32, [c_float_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                 ctypes.c_void_p, ctypes.c_float, c_float_p, c_float_p, ctypes.c_void_p]),
    "vsrd_sample_stratified": (ctypes.c_int32, [ctypes.POINTER(RenderConfig), c_float_p, c_float_p, ctypes.c_void_p]),
    "vsrd_sample_importance": (ctypes.c_int32, [ctypes.POINTER(RenderConfig), c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "vsrd_render_forward": (ctypes.c_int32, [ctypes.POINTER(Field), ctypes.POINTER(RenderConfig), c_float_p, c_float_p, c_float_p,
                                             ctypes.c_int32, c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "vsrd_render_backward": (ctypes.c_int32, [ctypes.POINTER(Field), ctypes.POINTER(RenderConfig), c_float_p, c_float_p, c_float_p,
                                              ctypes.c_int32, c_float_p, c_float_p, c_float_p, ctypes.c_void_p, ctypes.c_size_t,
                                              c_float_p, c_float_p, ctypes.c_void_p]),
    "vsrd_render_hierarchical_forward": (ctypes.c_int32, [ctypes.POINTER(Field), ctypes.POINTER(RenderConfig), c_float_p, c_float_p,
                                                          c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                                          c_float_p, c_float_p, ctypes.c_void_p]),
    "vsrd_render_silhouette_step": (ctypes.c_int32, [ctypes.POINTER(Field), ctypes.POINTER(RenderConfig), c_float_p, c_float_p, c_float_p,
                                                     c_float_p, c_float_p, c_float_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t,
                                                     c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "vsrd_residual_step_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "vsrd_render_residual_step": (ctypes.c_int32, [ctypes.POINTER(Field), ctypes.POINTER(RenderConfig), c_float_p, c_float_p, c_float_p,
                                                   c_float_p, c_float_p, c_float_p, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                                   ctypes.c_size_t, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "vsrd_project_boxes_forward": (ctypes.c_int32, [c_float_p, c_float_p, c_float_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                                    ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                                    c_float_p, c_float_p, ctypes.c_void_p, ctypes.c_void_p]),
    "vsrd_field_eval_backward": (ctypes.c_int32, [ctypes.POINTER(Field), c_float_p, ctypes.c_int64, c_float_p, c_float_p, ctypes.c_int32,
                                                  ctypes.c_void_p, ctypes.c_size_t, c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "vsrd_sample_rays_workspace_bytes": (ctypes.c_size_t, []),
    "vsrd_sample_rays": (ctypes.c_int32, [c_float_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "vsrd_centre_mlp_weights": (ctypes.c_int32, [c_float_p, ctypes.c_int32, c_float_p, ctypes.c_void_p]),
    "vsrd_ray_table_bytes": (ctypes.c_size_t, [ctypes.c_int64]),
    "vsrd_ray_table_build": (ctypes.c_int32, [c_float_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "vsrd_sample_rays_table": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "vsrd_match_boxes": (ctypes.c_int32, [c_float_p, c_float_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "vsrd_linear_sum_assignment": (ctypes.c_int32, [c_float_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "vsrd_frame_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    "vsrd_frame_prologue": (ctypes.c_int32, [ctypes.POINTER(FrameConfig), c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, c_float_p, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "vsrd_frame_prologue_sample": (ctypes.c_int32, [ctypes.POINTER(FrameConfig), c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, c_float_p, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                                    ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "vsrd_frame_epilogue": (ctypes.c_int32, [ctypes.POINTER(FrameConfig), c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_float,
                                             ctypes.POINTER(AdamTensors), ctypes.POINTER(AdamTensors), ctypes.POINTER(AdamTensors),
                                             c_float_p, c_float_p, ctypes.c_void_p, c_float_p, c_float_p, ctypes.c_void_p]),
    "vsrd_hypernetwork_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int32]),
    "vsrd_hypernetwork_forward": (ctypes.c_int32, [ctypes.POINTER(Hypernetwork), ctypes.c_void_p, ctypes.c_size_t, c_float_p, c_float_p, ctypes.c_void_p]),
    "vsrd_hypernetwork_backward_step": (ctypes.c_int32, [ctypes.POINTER(Hypernetwork), ctypes.c_void_p, ctypes.c_size_t, c_float_p, ctypes.c_float,
                                                         ctypes.c_void_p]),
    "vsrd_project_boxes_backward": (ctypes.c_int32, [c_float_p, c_float_p, c_float_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                                     ctypes.c_int32, ctypes.c_float, c_float_p, ctypes.c_void_p, c_float_p,
                                                     ctypes.c_void_p]),
    "vsrd_selftest_wave": (ctypes.c_int32, [c_float_p, c_float_p, ctypes.c_void_p]),
    "vsrd_selftest_gelu": (ctypes.c_int32, [c_float_p, ctypes.c_int32, c_float_p, ctypes.c_void_p]),
}

E_INVALID_ARGUMENT, E_UNSUPPORTED, E_LAUNCH, E_WORKSPACE = -1, -2, -3, -4      # include/vsrd_hip.h: VSRD_E_*

_lock = threading.Lock()
_lib = None


class VsrdHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIBRARY_PATH):
                raise VsrdHipError(
                    f"{LIBRARY_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950).  vsrd_amd has no CPU fallback.")
            lib = ctypes.CDLL(LIBRARY_PATH)
            for name, (restype, argtypes) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = restype
                fn.argtypes = argtypes
            if lib.vsrd_abi_version() != ABI_VERSION:
                raise VsrdHipError(f"ABI mismatch: library {lib.vsrd_abi_version()} vs binding {ABI_VERSION}")
            _lib = lib
    return _lib


def check(code):
    if code != 0:
        raise VsrdHipError(f"libvsrd_hip: {load().vsrd_error_string(code).decode()} (code {code})")


def ptr(tensor):
    """Device pointer of a contiguous fp32 HIP tensor (None -> NULL)."""
    if tensor is None:
        return None
    if not tensor.is_cuda:
        raise VsrdHipError("vsrd_amd operates on HIP device tensors only (got a CPU tensor); there is no CPU fallback")
    if tensor.dtype != torch.float32:
        raise VsrdHipError(f"expected float32, got {tensor.dtype}")
    if not tensor.is_contiguous():
        raise VsrdHipError("internal error: non-contiguous tensor handed to the C ABI")
    return ctypes.c_void_p(tensor.data_ptr())


def iptr(tensor):
    """Device pointer of a contiguous int32 HIP tensor."""
    if not tensor.is_cuda or tensor.dtype != torch.int32 or not tensor.is_contiguous():
        raise VsrdHipError("expected a contiguous int32 HIP tensor")
    return ctypes.c_void_p(tensor.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_field(instances, temperature, mlp_weights=None):
    return Field(int(instances.shape[0]), float(temperature), ptr(instances).value,
                 None if mlp_weights is None else ptr(mlp_weights).value)


def make_config(num_rays, num_samples, distance_range, sdf_std_deviation, cosine_ratio, epsilon, origin_stride,
                seed=0, stream_offset=0, flags=0, schedule=None, gather=None, samples=None, frames=None, adjoint_slots_per_item=0):
    """`schedule`: optional device tensor float32 [3] = (temperature, sdf_std_deviation, cosine_ratio) read by the kernels at
    start instead of the by-value scalars; `stream_offset` may likewise be a device int64 tensor (hipGraph replay).
    `samples`: optional (distances [R,2S], coarse_weights [R,S-1], u_coarse [R,S], u_fine [R,S]) float32 device tensors (any may be None)
    that vsrd_render_silhouette_step fills with its own state between the passes (vsrd_render_config::out_*).
    `frames`: optional (num_frames, frame_stride in bytes) -- a batch of independent frames in one launch (include/vsrd_hip.h, ABI 8)."""
    schedule_ptr = offset_ptr = None
    if schedule is not None:
        if schedule.dtype != torch.float32 or schedule.numel() != 3 or not schedule.is_cuda or not schedule.is_contiguous():
            raise ValueError("schedule must be a contiguous float32 device tensor of 3 elements")
        schedule_ptr = schedule.data_ptr()
    if isinstance(stream_offset, torch.Tensor):
        if stream_offset.dtype != torch.int64 or stream_offset.numel() != 1 or not stream_offset.is_cuda:
            raise ValueError("a tensor stream_offset must be a device int64 scalar")
        offset_ptr, stream_offset = stream_offset.data_ptr(), 0
    ray_indices = target_columns = None
    rays_per_origin = target_stride = 0
    if gather is not None:        # (ray_indices int64 [R] or None, rays_per_origin, target_columns int32 [N] or None, target_stride): vsrd_render_config gather
        indices, rays_per_origin, columns, target_stride = gather
        if indices is not None:   # (None: a dense launch that only maps the target columns)
            if indices.dtype != torch.int64 or not indices.is_cuda or not indices.is_contiguous() or indices.numel() != int(num_rays):
                raise ValueError("ray_indices must be a contiguous int64 device tensor with one entry per ray")
            ray_indices = indices.data_ptr()
        if columns is not None:
            target_columns = iptr(columns).value
    outs = [None, None, None, None]
    if samples is not None:
        outs = [None if t is None else ptr(t).value for t in samples]
    return RenderConfig(int(num_rays), int(num_samples), float(distance_range[0]), float(distance_range[1]),
                        float(sdf_std_deviation), float(cosine_ratio), float(epsilon), int(origin_stride),
                        int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_offset) & 0xFFFFFFFFFFFFFFFF, int(flags), schedule_ptr, offset_ptr,
                        ray_indices, int(rays_per_origin), target_columns, int(target_stride), *outs,
                        *((1, 0) if frames is None else (int(frames[0]), int(frames[1]))), int(adjoint_slots_per_item))
