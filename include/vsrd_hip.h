/*
 * libvsrd_hip -- C ABI of the MI355X (gfx950) implementation of VSRD's instance-aware
 * volumetric silhouette renderer and the geometry around it.
 *
 * The reference (skmhrk1209/VSRD) is pure Python on PyTorch and defines no FFI; the
 * boundary it offers is the Python call surface of vsrd.rendering / vsrd.operations as
 * used by scripts/main.py.  Every entry point below names the
 * reference function it replaces (file:line under the reference checkout).  The Python
 * package vsrd_amd/ binds these symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - all pointers are DEVICE pointers to fp32 (or the stated type), caller-owned;
 *   - per-sample tensors are ray-major: [R, D] (the reference's sample-major [D, R, 1]
 *     is a permuted view of the same buffer);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), touches no
 *     global state and is re-entrant;
 *   - return value: 0 on success, a negative VSRD_E_* code otherwise (never throws);
 *     vsrd_error_string() describes a code.
 */
#ifndef VSRD_HIP_H
#define VSRD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSRD_ABI_VERSION 8

#define VSRD_OK 0
#define VSRD_E_INVALID_ARGUMENT (-1) /* null pointer, non-positive size, unsupported N / S */
#define VSRD_E_UNSUPPORTED (-2)      /* valid request this build does not implement        */
#define VSRD_E_LAUNCH (-3)           /* hipGetLastError() != hipSuccess after the launch   */
#define VSRD_E_WORKSPACE (-4)        /* workspace too small (see vsrd_workspace_bytes)     */

#define VSRD_MAX_INSTANCES 64  /* N, instances per soft-min union                         */
#define VSRD_MAX_SAMPLES 256   /* S, num_samples of hierarchical_volumetric_rendering      */
#define VSRD_INSTANCE_STRIDE 16 /* floats per packed instance: t[3] R[9] (row-major) dim[3] pad */
#define VSRD_MLP_WEIGHTS 1617  /* per-instance residual MLP, hyper_distance_field.py:18-25 */

/* ---- frame batches (ABI 8) --------------------------------------------------------------------------------------------------
 * The reference's unit of work is one target frame optimised for 3000 steps of 1000 rays (scripts/main.py:323-865), and its loop carries
 * a batch dimension: BoxParameters3D(batch_size, num_instances) (vsrd/models/detectors/box_parameters.py:34-49), lists over the batch
 * of distance fields, camera positions, ray directions and soft masks (scripts/main.py:525-651).  One frame's launches leave most of the
 * GPU idle (about one wave per SIMD), so the entry points of the per-frame step take a batch of INDEPENDENT frames in one launch each:
 *
 *     num_frames   B >= 2 (0 or 1: one frame, the meaning of every earlier ABI);
 *     frame_stride bytes, a positive multiple of 256: EVERY device pointer of the call -- arguments, struct members, workspaces --
 *                  addresses frame 0's buffer, and frame f's buffer lies f * frame_stride bytes behind it.
 *
 * The caller keeps each frame's buffers at the same offsets of one arena row per frame.  By-value arguments (sizes, seeds, flags, Adam's
 * constants) are shared; per-frame state (parameters, Adam's moments and counters, learning rates, the step counter that keys the Philox
 * streams, schedules, ray tables, scratch) lies behind the pointers and is therefore the frame's own.  There is no arithmetic across
 * frames: frame f's results are BIT-IDENTICAL to those of the same call on frame f alone (the launch is B copies of the one-frame grid,
 * tests/test_hip_step.py::test_frame_batch_walks_each_frames_own_trajectory).
 * The fields sit at the end of vsrd_render_config, vsrd_frame_config and vsrd_hypernetwork.  Entry points that accept B >= 2:
 * vsrd_frame_prologue, vsrd_frame_prologue_sample, vsrd_frame_epilogue, vsrd_hypernetwork_forward, vsrd_hypernetwork_backward_step,
 * vsrd_render_silhouette_step (gathered launches in the split-ray form: at most 2048 rays per frame) and vsrd_render_residual_step
 * (two-kernel form, one chunk of rays per frame); every other entry point returns VSRD_E_UNSUPPORTED for B >= 2. */

/* Field = temperature soft-min union of N oriented boxes (+ optional residual MLP).
 * Replaces the closure tree scripts/main.py:525-618 builds from
 *   vsrd/rendering/sdfs.py:9-37 (box, rotation, translation),
 *   scripts/main.py:433-492     (residual field/composition, instance_field, soft_union). */
typedef struct vsrd_field {
    int32_t num_instances;      /* N, 1..VSRD_MAX_INSTANCES                                */
    float temperature;          /* sdf_union_temperature, main.py:423-426                  */
    const float* instances;     /* [N, 16] packed: location, rotation matrix, half extents */
    const float* mlp_weights;   /* [N, 1617] or NULL (box-only)                            */
} vsrd_field;

/* Scalars of vsrd.rendering.hierarchical_volumetric_rendering (renderers.py:177-188). */
typedef struct vsrd_render_config {
    int32_t num_rays;           /* R                                                        */
    int32_t num_samples;        /* S ("num_samples"); pass 1 renders S-1, pass 2 2S-1 points */
    float distance_near;        /* distance_range[0]                                        */
    float distance_far;         /* distance_range[1]                                        */
    float sdf_std_deviation;    /* NeuS logistic std                                        */
    float cosine_ratio;         /* cos anneal ratio, renderers.py:234-239                   */
    float epsilon;              /* opacity denominator guard, default 1e-6                  */
    int32_t origin_stride;      /* floats between ray origins: 3, or 0 for one shared origin */
    uint64_t seed;              /* Philox key when uniforms are generated in-kernel         */
    uint64_t stream_offset;     /* Philox counter offset (step index)                       */
    uint32_t flags;             /* VSRD_FLAG_*                                              */
    /* Optional, DEVICE memory: per-step values that a captured hipGraph must not freeze in its kernel arguments.  When set, the
     * render kernels read them at start and use them INSTEAD of the by-value fields above / field->temperature
     * (the annealing schedules of scripts/main.py:420-431 and the Philox counter of the step). */
    const float* device_schedule;          /* [3] = soft-min temperature, sdf_std_deviation, cosine_ratio; or NULL */
    const uint64_t* device_stream_offset;  /* [1] replaces stream_offset; or NULL                                   */
    /* Optional gather (the two fused step entry points only; the others reject it): the step's rays are rows of FRAME-RESIDENT
     * tensors picked by index, so that no per-step copies of directions / origins / targets are made (scripts/main.py:629-671 indexes
     * multi_ray_directions, multi_camera_positions and multi_soft_masks with multi_ray_indices).  With ray_indices set, ray r reads
     * directions[ray_indices[r]], origins[ray_indices[r] / rays_per_origin] (one camera position per view, origin_stride 3) and
     * the targets row ray_indices[r]; labels (if requested) stay [R,N] in step order. */
    const int64_t* ray_indices;            /* [R] or NULL                                                           */
    int32_t rays_per_origin;               /* pixels per view (H * W); 0: origins are per ray, gathered like the directions */
    /* Optional column map of the targets: a targets row has target_stride columns (ground-truth instance order) and predicted
     * instance n is compared with column target_columns[n] (the Hungarian assignment, main.py:653-671); NULL: targets are [.,N]
     * in prediction order. */
    const int32_t* target_columns;         /* [N] or NULL                                                           */
    int32_t target_stride;                 /* columns of a targets row when target_columns is set                   */
    /* Optional outputs of vsrd_render_silhouette_step (ABI 7) and vsrd_render_residual_step (ABI 8: two-kernel form, num_samples in
     * (32, 64], one ray per wave; VSRD_E_UNSUPPORTED otherwise); every other entry point rejects them: the per-ray state between the
     * step's passes, written by the step launch itself -- what pass 1 of scripts/main.py:511-523 returns (`sampled_weights`), the
     * uniforms it used, and the sorted pass-2 distances its labels, loss and gradients were computed at.  Rows are in STEP order (ray r
     * of the launch).  A step that writes them takes the multi-ray or the one-ray kernels (never the split-ray form). */
    float* out_distances;                  /* [R,2S] or NULL; row of an exact miss (VSRD_FLAG_SKIP_EXACT_MISSES): [r,0] = NaN, rest untouched */
    float* out_coarse_weights;             /* [R,S-1] or NULL: pass 1's compositing weights                           */
    float* out_u_coarse;                   /* [R,S] or NULL: the stratified uniforms used                            */
    float* out_u_fine;                     /* [R,S] or NULL: the fine uniforms used (SORTED when drawn in the kernel) */
    /* Frame batch (ABI 8, see "frame batches" above): num_rays etc. are PER FRAME. */
    int32_t num_frames;                    /* 0 or 1: one frame                                                       */
    int64_t frame_stride;                  /* bytes between the frames' copies of every buffer of the call            */
    /* vsrd_render_residual_step, two-kernel form (ABI 8): slots (= 64-sample rounds of a ray) per work item of the MLP-adjoint kernel,
     * 1..64; 0: planned from the launch's own size (about 16384 items per frame: 4 slots at 1000 rays, at most 32 -- and 64 for launches that
     * leave 32 k items even then: config 3's chunks of 110 k rays).  An item keeps its instance's
     * weight adjoints in registers and leaves ONE partial row of 6.6 KB, so larger items mean fewer rows to write and to sum -- what a
     * batch of frames wants, whose items are B times as many (8 frames x 1000 rays: 16 slots, -14 % per frame-step) -- and coarser load
     * balance.  The order of summation, and with it the last bits of grad_mlp_weights, depends on this number and on nothing else of the
     * launch geometry: equal numbers give bit-identical results for a frame alone and in a batch.  (A number below the planned one
     * needs more rows than vsrd_residual_step_workspace_bytes provides for: VSRD_E_WORKSPACE.) */
    int32_t adjoint_slots_per_item;
} vsrd_render_config;

#define VSRD_FLAG_FINE_UNIFORMS_SORTED 1u /* u_fine is already sorted ascending per ray     */
#define VSRD_FLAG_SKIP_EXACT_MISSES 2u    /* rays whose coarse weights are all exactly 0 skip pass 2:
                                             labels = 0 (exact); distances/gradients/weights of such
                                             rays are NOT produced (fused-loss mode only).  In the kernels
                                             that put several rays in a wave (dense box-only launches) the
                                             rays of a wave share the soft-min mode, so the OTHER rays of a
                                             wave agree with the un-skipped run to rounding, not bit for bit */

#define VSRD_FLAG_MLP_WEIGHTS_CENTRED 8u   /* residual fields: the caller guarantees that, in each of the four linears that feed a
                                             LayerNorm (rows [out][in+1] of mlp_weights), every column -- bias column included -- has
                                             zero mean over the 16 output channels.  LayerNorm makes the field invariant to that
                                             centring, and gradients w.r.t. centred weights equal those w.r.t. the originals; the
                                             kernels then skip centring the weight operands on every evaluation.                     */

#define VSRD_FLAG_NO_CULLING 4u           /* evaluate every instance at every sample (A/B switch for the
                                             conservative soft-min culling described in DESIGN.md)        */

#define VSRD_FLAG_RUNNING_MINIMUM 16u      /* shift the soft-min by the running minimum of the instance distances instead of the
                                             lower bound known before the instance loop (A/B switch; the kernels fall back to it by
                                             themselves wherever the bound is unavailable or too loose)                              */

#define VSRD_FLAG_RESIDUAL_SINGLE_KERNEL 64u /* vsrd_render_residual_step: run the whole step in ONE kernel (render + loss + adjoint of a
                                             batch of rays per wave, 427 registers, one wave per SIMD) instead of the default two
                                             kernels per chunk of rays (front part + MLP adjoint distributed by instance, two waves per
                                             SIMD each); the results agree to rounding (A/B switch)                         */

#define VSRD_FLAG_RESIDUAL_WAVE_PER_RAY 128u /* vsrd_render_residual_step: keep one wave per ray in the front kernel.  By default every
                                             ray is split over the two waves of a workgroup when num_samples > 64 (four wave rounds: the
                                             split kernel fits two waves per SIMD, the unsplit one does not) and, for num_samples in
                                             (32, 64], when the launch has <= 2048 rays (the reference's own 1000 rays per step then put
                                             two waves on every SIMD).  A/B switch.                                                   */

#define VSRD_FLAG_GENERAL_ROTATIONS 32u     /* do not use the shortened rotation products the kernels select by themselves when every
                                             instance's rotation is exactly one about the y axis (r01 = r10 = r12 = r21 = 0, r11 = 1,
                                             what rotation_matrix_y produces); the results agree to rounding (A/B switch)              */

#define VSRD_FLAG_STEP_WAVE_PER_RAY 256u    /* vsrd_render_silhouette_step: keep one wave per ray.  By default dense launches
                                             (ray_indices == NULL) with num_samples <= 64 and <= 16 instances put FOUR consecutive rays
                                             in a wave (16 sample points of each per round): the wave-uniform instance culling then
                                             works on a quarter of the depth range, and one parameter-adjoint reduction serves four
                                             rays; the results agree to rounding (A/B switch)                              */

#define VSRD_FLAG_STEP_SPLIT_RAY 512u       /* vsrd_render_silhouette_step: split every ray over the two waves of a workgroup, half of its
                                             rounds each (num_samples in (32, 128]: 2 or 4 rounds of 64 pass-2 points; for fewer samples the flag has no
                                             effect and the launch keeps one wave per ray).  By default launches of at most 2048 rays that are
                                             not dense launches of the shapes above -- the reference's 1000 importance-sampled rays
                                             per step -- do this by themselves (two waves on every SIMD instead of one: a small launch
                                             is latency); the flag forces it for any launch, VSRD_FLAG_STEP_WAVE_PER_RAY forbids it; the
                                             results agree to rounding (A/B switch)                                                     */

#define VSRD_FLAG_YAW_GRADIENTS 1024u       /* the caller differentiates the rotations only through rotation_matrix_y(cos, sin)
                                             (box_parameters.py:5-13: what BoxParameters3D decodes to): grad_instances then carries the
                                             adjoints of r00, r02, r20, r22 only -- the five entries that rotation_matrix_y keeps constant
                                             (r01, r10, r11, r12, r21) get ZERO instead of their adjoint, which nobody would read.  Only
                                             honoured when every rotation of the field IS of that form (checked on the device), by the
                                             multi-ray kernels of vsrd_render_silhouette_step / vsrd_render_backward; never set it for
                                             rotation matrices that are parameters themselves                                          */

#define VSRD_FLAG_MLP_SPLIT_BF16 2048u      /* vsrd_render_residual_step (two-kernel form): the per-instance MLP's matrix products of the FRONT
                                             kernel (pass 1, pass 2: value and gradient of every residual) run on v_mfma_f32_16x16x32_bf16 with
                                             both operands split into two bfloat16 parts (x = hi + lo to 2^-18; fp32 accumulation) instead of the
                                             exact-fp32 matrix instruction.  The C ABI's default is the exact-fp32 form; optimization.FrameOptimizer / FrameBatch
                                             (the native loop) set the flag by default (OptimizationConfig.mlp_split_bf16).  Measured on the GPU under the
                                             residual goldens' tests with both forms, against the reference's goldens: labels within 1.1e-6 (tolerance
                                             1e-4), gradients within 4.2e-4 of the largest entry (tolerance 5e-3; exact fp32: 1.5e-4).  Other entry
                                             points ignore it.                                                                          */

int32_t vsrd_abi_version(void);
const char* vsrd_error_string(int32_t code);

/* Bytes of scratch vsrd_render_backward needs for a field of N instances (residual != 0: with per-instance MLP). */
size_t vsrd_workspace_bytes(int32_t num_instances, int32_t residual);

/* Bytes of scratch with which vsrd_render_backward runs its fastest form for a launch of num_rays rays with num_distances sorted
 * distances each (>= vsrd_workspace_bytes).  Residual fields: the adjoint of renderers.py:177-270 through the per-instance MLPs
 * (scripts/main.py:433-458) runs as two kernels per chunk of rays -- the renderer's part, then the MLP adjoints by instance over the
 * whole chunk, as in vsrd_render_residual_step -- when the workspace holds the seeds of at least min(num_rays, 64) rays (2560 bytes per
 * (ray, instance, 64-sample round)); with less it keeps the one-kernel form (one wave per SIMD). */
size_t vsrd_render_backward_workspace_bytes(int32_t num_instances, int32_t residual, int32_t num_distances, int32_t num_rays);

/* The same fusion for RESIDUAL fields (BASELINE config 3; the reference's steps after warm-up): two-pass render + silhouette BCE
 * (scripts/main.py:653-671) + eikonal term (main.py:679-687: mean over all R (2S-1) samples of (|grad sdf| - 1)^2) + adjoint, one
 * launch.  The differentiated quantity is  losses[0] + eikonal_ratio * losses[1]  with
 *   losses[0] = sum_{r,n} w_n BCE(...) * loss_scale      losses[1] = mean (|grad sdf| - 1)^2 ;
 * grad_instances [N,16] and grad_mlp_weights [N,1617] are its gradients.  Needs vsrd_residual_step_workspace_bytes(N, S, R) of
 * scratch (seeds of the MLP adjoint for a chunk of rays: the launch is cut into chunks of at most 3 GiB of them).
 * VSRD_FLAG_SKIP_EXACT_MISSES is ignored (the eikonal term needs every ray). */
size_t vsrd_residual_step_workspace_bytes(int32_t num_instances, int32_t num_samples, int32_t num_rays);
int32_t vsrd_render_residual_step(const vsrd_field* field, const vsrd_render_config* config,
                                  const float* origins, const float* directions, const float* u_coarse, const float* u_fine,
                                  const float* targets /* [R,N] */, const float* instance_weights /* [N] or NULL */, float loss_scale,
                                  float eikonal_ratio, void* workspace, size_t workspace_bytes,
                                  float* losses /* [2] */, float* grad_instances, float* grad_mlp_weights, float* labels /* [R,N] or NULL */,
                                  void* stream);

/* Hungarian matching of predicted to ground-truth 2-D boxes of the target view (scripts/main.py:374-386:
 * scipy.optimize.linear_sum_assignment(-torchvision.ops.distance_box_iou(pd, gt).cpu())), on the device -- no host
 * synchronisation, capturable in a hipGraph.  pd_boxes [P,4], gt_boxes [G,4] (x1,y1,x2,y2), 1 <= P,G <= 64.
 * pd_indices / gt_indices [min(P,G)] (int64): the matched pairs ordered by pd index, exactly what scipy returns
 * (same shortest-augmenting-path algorithm, float64 duals, same tie rule). */
int32_t vsrd_match_boxes(const float* pd_boxes, const float* gt_boxes, int32_t num_pd, int32_t num_gt,
                         int64_t* pd_indices, int64_t* gt_indices, void* stream);

/* The assignment alone on a given cost matrix [P,G] (float32, row-major): scipy.optimize.linear_sum_assignment(cost). */
int32_t vsrd_linear_sum_assignment(const float* cost, int32_t num_rows, int32_t num_cols,
                                   int64_t* row_indices, int64_t* col_indices, void* stream);

/* Importance sampling of rays (scripts/main.py:620-627: torch.multinomial(weights, num_rays, replacement=False)): num_samples
 * (<= 2048) distinct indices with probability proportional to weights [count] (>= 0), by ATen's own algorithm (exponential race,
 * keep the largest keys) with Philox4x32-10 keyed by (seed, stream_offset; index); deterministic in its arguments.
 * device_stream_offset (device memory, may be NULL) replaces stream_offset (hipGraph replay).  Needs vsrd_sample_rays_workspace_bytes()
 * of scratch.  indices [num_samples] int64, best key first; -1 fills the tail when fewer than num_samples weights are positive. */
size_t vsrd_sample_rays_workspace_bytes(void);
int32_t vsrd_sample_rays(const float* weights, int64_t count, int32_t num_samples, uint64_t seed, uint64_t stream_offset,
                         const uint64_t* device_stream_offset, void* workspace, size_t workspace_bytes, int64_t* indices, void* stream);

/* The same draw for weights that stay fixed over many calls (a frame's soft masks do not change over its 3000 steps,
 * scripts/main.py:204-265, 620-627): vsrd_ray_table_build turns weights [count] into a table once (fixed-point weights and their
 * 64-bit integer prefix sums: exact, so the table does not depend on the order of summation), and vsrd_sample_rays_table draws from it
 * in ONE single-workgroup launch: i.i.d. picks from the table in a fixed order, Philox4x32-10 keyed by (seed, stream_offset; pick
 * number), repeats skipped, the first num_samples (<= 2048) distinct picks kept -- successive sampling without replacement, the
 * distribution of torch.multinomial(replacement=False) and of vsrd_sample_rays (the sequences themselves differ from theirs).
 * remap (may be NULL): indices[i] = remap[pick] (the caller's compaction of the positive weights).  Skipping repeats needs about
 * num_samples / (weight mass outside the num_samples - 1 heaviest entries) picks; a call gives up after 32768, fills the rest with
 * repeats of its first picks and sets a sticky flag in the table (the uint32 at byte 28; reading it back is a host
 * synchronisation), so callers check that mass once per table and keep vsrd_sample_rays for weights that fail it
 * (vsrd_amd/rendering/samplers.py::RayTable.suits).
 * count < 2^32 - 1.  Fewer than num_samples positive weights: -1 fills the tail, as above. */
size_t vsrd_ray_table_bytes(int64_t count);
int32_t vsrd_ray_table_build(const float* weights, int64_t count, void* table, size_t table_bytes, void* stream);
int32_t vsrd_sample_rays_table(void* table, int64_t count, int32_t num_samples, uint64_t seed, uint64_t stream_offset,
                               const uint64_t* device_stream_offset, const int64_t* remap, int64_t* indices, void* stream);

/* vsrd.rendering.ray_casting (vsrd/rendering/utils.py:5-18), the per-pixel part:
 * directions[v,y,x,:] = normalize(inverse_projection[v] @ (x, y, 1)), integer pixel centres.
 * inverse_projection [V,9] = inv(E)[:3,:3] @ inv(K) (row-major), computed by the caller. */
int32_t vsrd_ray_directions(const float* inverse_projection, int32_t num_views, int32_t height, int32_t width,
                            float* directions /* [V,H,W,3] */, void* stream);

/* Evaluate the field at arbitrary points: what calling the distance_field closure does
 * (scripts/main.py:477-492) plus the autograd normal of renderers.py:76-113 / :218-228.
 * Any of distances [P], gradients [P,3], labels [P,N] may be NULL.
 * hard_union != 0 selects the arg-min union of scripts/main.py:494-509 (distances only). */
int32_t vsrd_field_eval(const vsrd_field* field, const float* positions /* [P,3] */, int64_t num_points,
                        float* distances, float* gradients, float* labels, int32_t hard_union, void* stream);

/* Adjoint of vsrd_field_eval w.r.t. the field parameters and the positions: the reference's closure call is differentiable
 * (autograd), e.g. sphere_tracing(differentiable=True) (renderers.py:59-72).  grad_distances [P] and/or grad_labels [P,N] (soft
 * union only) may be NULL; the analytic normal output of vsrd_field_eval is NOT differentiated (second order).
 * Needs vsrd_workspace_bytes(N, residual) of scratch.  grad_positions [P,3] or NULL; grad_instances [N,16];
 * grad_mlp_weights [N,1617] (required for residual fields). */
int32_t vsrd_field_eval_backward(const vsrd_field* field, const float* positions, int64_t num_points,
                                 const float* grad_distances, const float* grad_labels, int32_t hard_union,
                                 void* workspace, size_t workspace_bytes, float* grad_positions,
                                 float* grad_instances, float* grad_mlp_weights, void* stream);

/* vsrd.rendering.sphere_tracing, non-differentiable part (renderers.py:21-59): march each ray by the union distance until
 * |sdf| < convergence_criteria, it leaves the bounding sphere, or num_iterations is reached.
 * origins [R,3] (origin_stride 3) or [3] (stride 0); foreground [R] uint8 or NULL (= all finite origins);
 * bounding_radius <= 0: none; initialise != 0: start from the bounding-sphere entry point (renderers.py:36-43).
 * Outputs: positions [R,3], converged [R] uint8 (the reference's convergence_masks). */
int32_t vsrd_sphere_trace(const vsrd_field* field, const float* origins, int32_t origin_stride, const float* directions,
                          const uint8_t* foreground, int64_t num_rays, int32_t num_iterations, float convergence_criteria,
                          float bounding_radius, int32_t initialise, int32_t hard_union,
                          float* positions, uint8_t* converged, void* stream);

/* SoftRasterizer.make_distance_map + soft masks (vsrd/transforms/geometric_transforms.py:265-317), the step that produces the
 * path's targets: per instance, distance of every pixel to the closed polygon [count,2] (x,y), then
 * soft = sigmoid((inside ? d : -d) / temperature).  polygons [B,Pmax,2] (rows beyond counts[b] ignored), counts [B] int32,
 * inside [B,H,W] uint8 (required for soft_masks).  distance_maps / soft_masks [B,H,W], either may be NULL. */
int32_t vsrd_polygon_soft_masks(const float* polygons, const int32_t* counts, int32_t num_polygons, int32_t max_vertices,
                                int32_t height, int32_t width, const uint8_t* inside, float temperature,
                                float* distance_maps, float* soft_masks, void* stream);

/* samplers.quadrature_sampler over linspace bins (renderers.py:191-194, samplers.py:5-8).
 * u_coarse [R,S] in [0,1) -> distances [R,S]. */
int32_t vsrd_sample_stratified(const vsrd_render_config* config, const float* u_coarse, float* distances, void* stream);

/* Pass-2 distances: cat(coarse, inverse_transform_sampler(coarse, weights)) sorted
 * (renderers.py:198-210, samplers.py:11-36).  coarse_distances [R,S], coarse_weights [R,S-1],
 * u_fine [R,S] (raw draws, or sorted with VSRD_FLAG_FINE_UNIFORMS_SORTED) -> merged [R,2S] and/or the sampler's own
 * output fine [R,S] (either may be NULL). */
int32_t vsrd_sample_importance(const vsrd_render_config* config, const float* coarse_distances,
                               const float* coarse_weights, const float* u_fine, float* merged, float* fine, void* stream);

/* renderers.py:212-270 for given sorted distances [R,D]: evaluates the field and its normal at
 * the D-1 interval mid-points, converts to opacities, composites front to back.
 * Outputs: labels [R,N]; gradients [R,D-1,3] (may be NULL); weights [R,D-1] (may be NULL). */
int32_t vsrd_render_forward(const vsrd_field* field, const vsrd_render_config* config,
                            const float* origins, const float* directions,
                            const float* distances, int32_t num_distances,
                            float* labels, float* gradients, float* weights, void* stream);

/* Adjoint of vsrd_render_forward w.r.t. the packed instances (what autograd, including the
 * double-backward through the SDF normal, produces in the reference -- renderers.py:218-228).
 * grad_labels [R,N]; grad_gradients [R,D-1,3] or NULL; grad_weights [R,D-1] or NULL.
 * grad_instances [N,16] is OVERWRITTEN (pad column = 0); grad_mlp_weights [N,1617] likewise (required iff the field
 * carries mlp_weights, else NULL).  workspace: vsrd_workspace_bytes(N, residual). */
int32_t vsrd_render_backward(const vsrd_field* field, const vsrd_render_config* config,
                             const float* origins, const float* directions,
                             const float* distances, int32_t num_distances,
                             const float* grad_labels, const float* grad_gradients, const float* grad_weights,
                             void* workspace, size_t workspace_bytes,
                             float* grad_instances, float* grad_mlp_weights, void* stream);

/* The two-pass wrapper scripts/main.py:511-523 around renderers.py:177-270 in ONE launch:
 * pass 1 (stratified, no grad) -> importance sampling -> merge -> pass 2.
 * u_coarse / u_fine [R,S]: recorded uniforms, or NULL -> Philox4x32-10 keyed by (seed, stream_offset, ray).
 * Outputs: labels [R,N]; distances [R,2S] (needed by vsrd_render_backward; may be NULL for
 * inference); gradients [R,2S-1,3] / weights [R,2S-1] may be NULL;
 * coarse_weights [R,S-1] (may be NULL; ABI 7): pass 1's compositing weights, i.e. the `sampled_weights` pass 1 of
 * main.py:511-523 returns and samplers.py:11-36 turns into the importance samples (its `sampled_distances` are
 * torch.lerp(linspace bins, u_coarse), vsrd_sample_stratified);
 * u_coarse_out / u_fine_out [R,S] (may be NULL) export the uniforms actually used (u_fine_out: SORTED when they
 * were generated in the kernel).  Box-only launches that ask for neither `gradients` nor `weights` run in the
 * fused step's own mappings (several rays per wave), so these outputs are the step kernels' own pass-1 state. */
int32_t vsrd_render_hierarchical_forward(const vsrd_field* field, const vsrd_render_config* config,
                                         const float* origins, const float* directions,
                                         const float* u_coarse, const float* u_fine,
                                         float* labels, float* distances, float* gradients, float* weights, float* coarse_weights,
                                         float* u_coarse_out, float* u_fine_out, void* stream);

/* One fused optimisation-step launch for box-only fields: the two-pass render of vsrd_render_hierarchical_forward, the
 * silhouette loss of scripts/main.py:653-671  --  loss = loss_scale * sum_{r,n} w_n BCE(clamp(labels[r,n], 1e-6, 1-1e-6), targets[r,n])
 * --  and its gradient w.r.t. the packed instances, without writing any per-sample data to HBM.
 * targets [R,N] in prediction order; instance_weights [N] or NULL (all 1; 0 drops an unmatched instance);
 * loss_scale = 1 / (R * number of kept instances) gives the reference's mean.  Outputs: loss [1], grad_instances [N,16]
 * (d loss / d instances), labels [R,N] or NULL.  workspace: vsrd_workspace_bytes(N, 0). */
int32_t vsrd_render_silhouette_step(const vsrd_field* field, const vsrd_render_config* config,
                                    const float* origins, const float* directions, const float* u_coarse, const float* u_fine,
                                    const float* targets, const float* instance_weights, float loss_scale,
                                    void* workspace, size_t workspace_bytes,
                                    float* loss, float* grad_instances, float* labels, void* stream);

/* Multi-view projection of N boxes (8 corners each, world frame) into V cameras and reduction to clipped 2-D boxes:
 * scripts/main.py:339-362 = einsum with E + divide by w, vsrd.operations.project_box_3d / clip_lines_to_front
 * (geometric_operations.py:343-389) per box, torchvision.ops.clip_boxes_to_image.  One launch for all V*N boxes.
 * world_corners [N,8,3]; extrinsics [V,16]; intrinsics [V,9] (row-major); edges [E,2] corner indices (E <= 32).
 * height/width <= 0 skips the image clamp (plain project_box_3d).
 * Outputs: boxes_2d [V,N,4] = (x1,y1,x2,y2); camera_corners [V,N,8,3] or NULL; selection [V,N,4] (int32: which edge end
 * point attained each extreme, -1 = no gradient) -- consumed by the backward. */
int32_t vsrd_project_boxes_forward(const float* world_corners, const float* extrinsics, const float* intrinsics,
                                   const int32_t* edges, int32_t num_edges, int32_t num_views, int32_t num_boxes,
                                   int32_t height, int32_t width, float epsilon,
                                   float* boxes_2d, float* camera_corners, int32_t* selection, void* stream);

/* Adjoint of the above (torch min/max/clamp/where backward semantics): grad_boxes_2d [V,N,4] ->
 * grad_world_per_view [V,N,8,3]; the caller sums over V (fixed order: deterministic). */
int32_t vsrd_project_boxes_backward(const float* world_corners, const float* extrinsics, const float* intrinsics,
                                    const int32_t* edges, int32_t num_edges, int32_t num_views, int32_t num_boxes, float epsilon,
                                    const float* grad_boxes_2d, const int32_t* selection, float* grad_world_per_view, void* stream);

/* ---- the rest of one optimisation step of scripts/main.py around the render launch, for <= 64 boxes and <= 32 views ------------
 * In the reference these are a few hundred small ATen launches per step (box decode, V x N project_box_3d calls, DIoU matching
 * through scipy on the host, projection losses, schedules, autograd through all of it, Adam, ExponentialLR); here two
 * single-workgroup launches, so that the reference's native mode (1000 rays per step) is bound by its render launch. */
typedef struct vsrd_frame_config {
    int32_t num_boxes;              /* N (predictions = ground-truth slots, main.py:204-265 pads the sources to the target's N) */
    int32_t num_views;              /* V, view 0 = target view                                                               */
    float height, width;            /* image size for clip_boxes_to_image                                                     */
    float epsilon;                  /* project_box_3d / clip_lines_to_front guard, 1e-6                                       */
    float location_lo[3], location_hi[3];     /* BoxParameters3D.location_range (box_parameters.py:23-26)                     */
    float dimension_lo[3], dimension_hi[3];   /* BoxParameters3D.dimension_range (box_parameters.py:27-30)                    */
    int32_t num_steps;              /* optimisation steps of the schedules (config.json:167)                                  */
    float max_temperature, min_temperature, max_std, min_std;   /* config.json:230-233                                        */
    float weight_iou, weight_l1, weight_silhouette;             /* config.json:120-127                                        */
    float beta1, beta2, adam_epsilon; /* torch.optim.Adam defaults 0.9, 0.999, 1e-8                                           */
    float lr_gamma;                 /* ExponentialLR gamma (config.json:209-215)                                              */
    int32_t num_frames;             /* frame batch (ABI 8, "frame batches" above): 0 or 1 = one frame                         */
    int64_t frame_stride;           /* bytes between the frames' copies of every buffer of the call                           */
} vsrd_frame_config;

size_t vsrd_frame_scratch_bytes(int32_t num_views, int32_t num_boxes);

/* Before the render launch.  scripts/main.py:332 (BoxParameters3D.forward, box_parameters.py:124-146), :339-367 (projection),
 * :374-386 (matching: -DIoU + linear_sum_assignment, as vsrd_match_boxes), :391-415 (distance_box_iou_loss + smooth_l1_loss over
 * matched, visible instances), :420-431 (schedules from the device step counter).
 * raw_locations / raw_dimensions [N,3], raw_orientations [N,2]; extrinsics [V,16], intrinsics [V,9]; gt_boxes [V,N,4];
 * visible [V,N] uint8 (by ground-truth instance); step [1] int64 (device).
 * Outputs: instances [N,16] (the renderer's field block); pd_indices / gt_indices [N] int64; target_columns [N] int32 and
 * instance_weights [N] (for vsrd_render_config.target_columns / the step entry points); schedule [3] (for
 * vsrd_render_config.device_schedule); projection_losses [2] = (iou, l1); grad_raw [N,8] = d (weight_iou * iou + weight_l1 * l1) /
 * d (raw location 3 | raw dimension 3 | raw orientation 2). */
int32_t vsrd_frame_prologue(const vsrd_frame_config* config, const float* raw_locations, const float* raw_dimensions,
                            const float* raw_orientations, const float* extrinsics, const float* intrinsics, const float* gt_boxes,
                            const uint8_t* visible, const int64_t* step, void* scratch, size_t scratch_bytes,
                            float* instances, int64_t* pd_indices, int64_t* gt_indices, int32_t* target_columns, float* instance_weights,
                            float* schedule, float* projection_losses, float* grad_raw, void* stream);

/* vsrd_frame_prologue and the draw of the step's rays (vsrd_sample_rays_table keyed by the same device step counter `step`) in ONE launch
 * of two workgroups: neither needs the other, and as two launches on two streams they meet again through a cross-queue dependency that
 * costs as much as the draw itself.  ray_table / count / remap as for vsrd_sample_rays_table; ray_indices [num_rays] int64 out. */
int32_t vsrd_frame_prologue_sample(const vsrd_frame_config* config, const float* raw_locations, const float* raw_dimensions,
                                   const float* raw_orientations, const float* extrinsics, const float* intrinsics, const float* gt_boxes,
                                   const uint8_t* visible, const int64_t* step, void* scratch, size_t scratch_bytes,
                                   float* instances, int64_t* pd_indices, int64_t* gt_indices, int32_t* target_columns, float* instance_weights,
                                   float* schedule, float* projection_losses, float* grad_raw,
                                   void* ray_table, int64_t count, int32_t num_rays, uint64_t seed, const int64_t* remap, int64_t* ray_indices, void* stream);

/* torch.optim.Adam(capturable=True) state of one parameter tensor: all device memory, updated in place. */
typedef struct vsrd_adam_tensors {
    float* parameter;
    float* exp_avg;
    float* exp_avg_sq;
    float* step;                    /* [1] float32                                                                            */
    float* learning_rate;           /* [1] float32, multiplied by lr_gamma after the update (ExponentialLR)                   */
} vsrd_adam_tensors;

/* After the render launch.  scripts/main.py:855-865: total loss, backward through the decode (box_parameters.py:60-90), Adam step
 * on locations / dimensions / orientations, scheduler step, step counter.  grad_instances [N,16] = d (silhouette + eikonal_ratio *
 * eikonal) / d instances from the render step; grad_raw_projection [N,8] and projection_losses [2] from the prologue;
 * render_losses [2] = (silhouette, eikonal).  other_learning_rates: two more [1] float32 rates decayed by lr_gamma (embeddings,
 * hypernetwork groups), either may be NULL.  Outputs: record [5] = (iou, l1, silhouette, eikonal, total); raw_gradients [N,8] or NULL. */
int32_t vsrd_frame_epilogue(const vsrd_frame_config* config, const float* grad_instances, const float* grad_raw_projection,
                            const float* projection_losses, const float* render_losses, float eikonal_ratio,
                            const vsrd_adam_tensors* locations, const vsrd_adam_tensors* dimensions, const vsrd_adam_tensors* orientations,
                            float* other_learning_rate_0, float* other_learning_rate_1, int64_t* step,
                            float* record, float* raw_gradients, void* stream);

/* ---- the hypernetwork that generates the residual MLPs' weights (vsrd/models/fields/hyper_distance_field.py:27-55, 75-77) --------
 * embeddings [N,256] -> (VSRD_HYPER_LAYERS - 1) x [weight_norm(Linear 256 -> 256) -> LayerNorm(256, affine) -> exact GELU]
 * -> weight_norm(Linear 256 -> 1617) = mlp_weights [N,1617].  weight_norm over dim 0: W[o,:] = g[o] v[o,:] / |v[o,:]|.
 * Every tensor is given with its torch.optim.Adam(capturable=True) state: the backward entry point also takes the Adam step
 * (main.py:860-865; embeddings and hypernetwork are parameter groups 3 and 4 with their own rates) and decays the two rates.
 * With ATen this is ~130 launches of a few microseconds per step; here 6 forwards and 12 backwards. */
#define VSRD_HYPER_LAYERS 5
#define VSRD_HYPER_WIDTH 256
typedef struct vsrd_hypernetwork {
    int32_t num_instances;                                   /* N <= 64                                                       */
    int32_t num_outputs;                                     /* 1617                                                          */
    float beta1, beta2, adam_epsilon, lr_gamma;
    vsrd_adam_tensors embeddings;                            /* [N,256]                                                       */
    vsrd_adam_tensors weight_v[VSRD_HYPER_LAYERS];           /* [256,256] x 4, [1617,256]                                     */
    vsrd_adam_tensors weight_g[VSRD_HYPER_LAYERS];           /* [256] x 4, [1617]                                             */
    vsrd_adam_tensors bias[VSRD_HYPER_LAYERS];
    vsrd_adam_tensors norm_weight[VSRD_HYPER_LAYERS - 1];    /* LayerNorm gamma [256]                                         */
    vsrd_adam_tensors norm_bias[VSRD_HYPER_LAYERS - 1];
    int32_t num_frames;                                      /* frame batch (ABI 8, "frame batches" above): every frame its OWN hypernetwork, */
    int64_t frame_stride;                                    /* embeddings and Adam state, frame_stride bytes apart                           */
} vsrd_hypernetwork;

size_t vsrd_hypernetwork_workspace_bytes(int32_t num_instances);

/* mlp_weights [N,1617] (what HyperDistanceField.hypernetwork(embeddings) returns); centred [N,1617] or NULL: the same with the
 * column means of the four LayerNorm-feeding linears removed (VSRD_FLAG_MLP_WEIGHTS_CENTRED).  The workspace keeps the
 * activations for the backward call. */
int32_t vsrd_hypernetwork_forward(const vsrd_hypernetwork* net, void* workspace, size_t workspace_bytes,
                                  float* mlp_weights, float* centred, void* stream);

/* grad_mlp_weights [N,1617], multiplied by grad_scale (the loss weight) on the way in.  Backward through the hypernetwork, Adam on
 * every tensor of `net` (in place, torch's update), step counters + 1, both learning rates * lr_gamma.  Deterministic. */
int32_t vsrd_hypernetwork_backward_step(const vsrd_hypernetwork* net, void* workspace, size_t workspace_bytes,
                                        const float* grad_mlp_weights, float grad_scale, void* stream);

/* The centred copy of per-instance MLP weights [N,1617] that VSRD_FLAG_MLP_WEIGHTS_CENTRED announces: in the four linears that feed a
 * LayerNorm (hyper_distance_field.py:57-73) every column, bias column included, has its mean over the 16 output channels removed.
 * LayerNorm makes the field invariant to it, and the adjoints w.r.t. the centred weights are the adjoints w.r.t. the originals. */
int32_t vsrd_centre_mlp_weights(const float* mlp_weights, int32_t num_instances, float* centred, void* stream);

/* ---- self-tests of device building blocks (used by tests/, not by the product path) --------------------------------------------
 * vsrd_selftest_wave: the wave64 primitives of csrc/wave.h (sums, scans, the reduce-scatter butterflies) on one wave: in64 [64] ->
 * out [576].  vsrd_selftest_gelu: erf_gelu / erf_gelu_derivative of csrc/hypernetwork.h at n points: out[0..n) values, out[n..2n)
 * derivatives (the hypernetwork's exact GELU, hyper_distance_field.py:38-41). */
int32_t vsrd_selftest_wave(const float* in64, float* out576, void* stream);
int32_t vsrd_selftest_gelu(const float* x, int32_t n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VSRD_HIP_H */
