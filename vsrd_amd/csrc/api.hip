// C ABI of libvsrd_hip (include/vsrd_hip.h): argument validation, launch geometry, dispatch on the
// number of 64-sample rounds.  No torch types, no allocation, and no state beyond the experiment switches below, which are read from
// the environment ONCE (first call) and frozen: no entry point calls getenv per launch.
#include <cstdio>
#include <cstdlib>
#include "../../include/vsrd_hip.h"
#include "aux_kernels.h"
#include "quad_step.h"
#include "matching.h"
#include "ray_sampling.h"
#include "projection.h"
#include "frame_step.h"
#include "hypernetwork.h"
#include "split_front.h"

namespace {

using namespace vsrd;

constexpr int kMaxBlocks = 16384;             // 256 CUs x 8 workgroups: many more workgroups than are resident: rays differ a lot in cost and the dispatcher is the load balancer (r01: 2048 -> 16384 workgroups = -12 % time); also sizes the partial buffer

// Experiment / A-B switches of the library, all of them: read once, then constants of the process.
struct Switches {
    int grid_blocks;          // VSRD_GRID_BLOCKS=<n> (<= kMaxBlocks): cap of the render grids
    bool cap_to_residency;    // VSRD_CAP_TO_RESIDENCY=1: fit_to_residency really caps
    bool debug;               // VSRD_DEBUG: failed launches / LDS opt-ins are reported on stderr
    int split_max_rays;       // VSRD_SPLIT_MAX_RAYS: box-only step, launches of at most this many rays split every ray over two waves (0: never)
    bool no_full_shape;       // VSRD_NO_FULL_SHAPE: the step keeps the run-time-shape kernels (quad_step.h: kFull)
    int pair_max_rays;        // VSRD_PAIR_MAX_RAYS: residual step, 0 turns the pair kernel off for two-round launches
    int slots_per_item;       // VSRD_SLOTS_PER_ITEM: forced item size of the MLP adjoint (0: planned)
};
const Switches& switches() {
    static const Switches frozen = [] {
        auto number = [](const char* name, int fallback) { const char* e = getenv(name); return e ? atoi(e) : fallback; };
        Switches w;
        const int blocks = number("VSRD_GRID_BLOCKS", 0);
        w.grid_blocks = (blocks > 0 && blocks <= kMaxBlocks) ? blocks : kMaxBlocks;
        const char* cap = getenv("VSRD_CAP_TO_RESIDENCY");
        w.cap_to_residency = cap != nullptr && cap[0] == '1';
        w.debug = getenv("VSRD_DEBUG") != nullptr;
        w.split_max_rays = number("VSRD_SPLIT_MAX_RAYS", 2048);
        w.no_full_shape = getenv("VSRD_NO_FULL_SHAPE") != nullptr;
        w.pair_max_rays = number("VSRD_PAIR_MAX_RAYS", 2048);
        w.slots_per_item = number("VSRD_SLOTS_PER_ITEM", 0);
        return w;
    }();
    return frozen;
}
int grid_cap() { return switches().grid_blocks; }
constexpr size_t kLdsLimit = 160 * 1024;     // gfx950 LDS per CU
constexpr size_t kLdsDefault = 64 * 1024;    // dynamic LDS without opting in

struct Geometry {
    int blocks;
    int threads;
    size_t lds_bytes;
};

// Pick the number of waves per workgroup so the wave-private LDS partitions fit.
bool plan(int num_rays, size_t floats_per_wave, Geometry* g) {
    const size_t per_wave = floats_per_wave * sizeof(float);
    if (per_wave > kLdsLimit) return false;
    int waves = kMaxWavesPerBlock;
    while (waves > 1 && per_wave * waves > kLdsDefault) waves >>= 1;
    g->threads = waves * kWave;
    g->lds_bytes = per_wave * waves;
    const long long want = (static_cast<long long>(num_rays) + waves - 1) / waves;
    const int cap = grid_cap();
    g->blocks = static_cast<int>(want < 1 ? 1 : (want > cap ? cap : want));
    return true;
}

// Size a persistent launch to what is actually co-resident: with a static ray -> wave assignment a grid larger than the
// residency runs in "phases", and a partially filled last phase idles most of the chip (2048 workgroups at 6 resident per CU
// = 1.33 phases cost 2).  Performance only -- correctness never depends on residency (no inter-workgroup communication).
template <typename Kernel>
void fit_to_residency(Kernel kernel, Geometry* g) {
    static int num_cus = 0;
    if (num_cus == 0) {
        int device = 0;
        hipDeviceProp_t props;
        if (hipGetDevice(&device) == hipSuccess && hipGetDeviceProperties(&props, device) == hipSuccess) num_cus = props.multiProcessorCount;
        if (num_cus <= 0) num_cus = 256;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kernel), g->threads, g->lds_bytes) != hipSuccess || per_cu < 1)
        return;
    const int resident = per_cu * num_cus;
    // Measured on C2 (r01): capping the grid at `resident` (one phase) is 5 % SLOWER than 2048 workgroups -- rays differ a lot in
    // cost (culling, early outs) and the extra workgroups act as dynamic load balancing -- so the grid is left alone unless the
    // experiment switch asks for it.
    if (switches().cap_to_residency && g->blocks > resident) g->blocks = resident;
}

template <typename Kernel>
int opt_in_lds(Kernel kernel, size_t bytes) {
    if (bytes <= kLdsDefault) return VSRD_OK;
    const hipError_t error = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
    if (error != hipSuccess && switches().debug) fprintf(stderr, "libvsrd_hip: LDS opt-in of %zu bytes failed: %s\n", bytes, hipGetErrorString(error));
    return error == hipSuccess ? VSRD_OK : VSRD_E_LAUNCH;
}

int launch_status() {
    const hipError_t error = hipGetLastError();
    if (error == hipSuccess) return VSRD_OK;
    if (switches().debug) fprintf(stderr, "libvsrd_hip: launch failed: %s\n", hipGetErrorString(error));
    return VSRD_E_LAUNCH;
}

bool valid_field(const vsrd_field* f) {
    return f != nullptr && f->instances != nullptr && f->num_instances >= 1 && f->num_instances <= VSRD_MAX_INSTANCES &&
           f->temperature > 0.0f;
}

// Frame batches (include/vsrd_hip.h, ABI 8): B copies of the one-frame grid along blockIdx.y, every pointer moved by frame_stride bytes per frame.
constexpr int kMaxFrames = 1024;
struct Frames {
    int count;              // B (1: one frame)
    long long stride;       // bytes; 0 for one frame (the kernels test this, not the count)
};
bool frames_of(int32_t num_frames, int64_t frame_stride, Frames* f) {
    f->count = num_frames <= 1 ? 1 : num_frames;
    f->stride = 0;
    if (f->count == 1) return true;
    if (f->count > kMaxFrames || frame_stride <= 0 || (frame_stride & 255) != 0) return false;
    f->stride = frame_stride;
    return true;
}
// one fill for every frame's copy of a scratch region (masks, counters)
bool clear_frames(void* first, size_t bytes, const Frames& frames, hipStream_t s) {
    if (frames.count == 1) return hipMemsetAsync(first, 0, bytes, s) == hipSuccess;
    return hipMemset2DAsync(first, static_cast<size_t>(frames.stride), 0, bytes, static_cast<size_t>(frames.count), s) == hipSuccess;
}

bool wants_samples(const vsrd_render_config* c) { return c->out_distances || c->out_coarse_weights || c->out_u_coarse || c->out_u_fine; }

bool valid_config(const vsrd_render_config* c, bool gather_allowed = false, bool samples_allowed = false) {
    if (c == nullptr || c->num_rays < 0 || c->num_samples < 2 || c->num_samples > VSRD_MAX_SAMPLES || !(c->sdf_std_deviation > 0.0f) ||
        (c->origin_stride != 0 && c->origin_stride != 3))
        return false;
    Frames frames;
    if (!frames_of(c->num_frames, c->frame_stride, &frames)) return false;
    if (wants_samples(c) && !samples_allowed) return false;      // only vsrd_render_silhouette_step writes its samples out
    const bool gather = c->ray_indices != nullptr || c->target_columns != nullptr;
    if (gather && !gather_allowed) return false;                 // only the fused step kernels read through an index
    if (c->ray_indices != nullptr && c->rays_per_origin < 0) return false;
    if (c->target_columns != nullptr && c->target_stride < 1) return false;
    return true;
}

FieldArgs field_args(const vsrd_field* f) {
    FieldArgs a;
    a.num_instances = f->num_instances;
    a.inv_t = 1.0f / f->temperature;
    return a;
}

RenderArgs render_args(const vsrd_render_config* c) {
    RenderArgs a;
    a.num_rays = c->num_rays;
    a.num_samples = c->num_samples;
    a.near = c->distance_near;
    a.far = c->distance_far;
    a.sh.inv_t = 0.0f;  // filled from the field
    a.sh.cull = 0.0f;   // computed in-kernel (field.h: field_bounds)
    a.sh.reach = -1.0f; // likewise
    a.sh.yaw = false;   // likewise
    a.sh.yaw_gradients = false;
    a.sh.std = c->sdf_std_deviation;
    a.sh.inv_std = 1.0f / c->sdf_std_deviation;
    a.sh.ratio = c->cosine_ratio;
    a.sh.eps = c->epsilon;
    a.sh.mlp_bits = 0u;  // set in-kernel from the flags
    a.sh.mlp_stride = kMlpWeights;   // (the two front kernels of the residual step switch to the image table under VSRD_FLAG_MLP_SPLIT_BF16)
    a.origin_stride = c->origin_stride;
    a.seed = c->seed;
    a.stream_offset = c->stream_offset;
    a.flags = c->flags;
    a.dynamic = c->device_schedule;
    a.dynamic_offset = reinterpret_cast<const unsigned long long*>(c->device_stream_offset);
    a.ray_indices = reinterpret_cast<const long long*>(c->ray_indices);
    a.rays_per_origin = c->rays_per_origin;
    a.target_columns = c->target_columns;
    a.target_stride = c->target_stride;
    a.out_distances = c->out_distances; a.out_coarse_weights = c->out_coarse_weights;
    a.out_u_coarse = c->out_u_coarse; a.out_u_fine = c->out_u_fine;
    a.frame_stride = (c->num_frames > 1) ? c->frame_stride : 0;      // (only the two fused step entry points launch batches; the others reject them)
    return a;
}

// rounds of 64 lanes needed for `points` sample points, rounded up to the instantiated {1,2,4,8}
int rounds_for(int points) {
    const int need = (points + kWave - 1) / kWave;
    if (need <= 1) return 1;
    if (need <= 2) return 2;
    if (need <= 4) return 4;
    if (need <= 8) return 8;
    return 0;
}

// The residual adjoint needs >= 17 KB of LDS per wave (residual.h: kMlpLdsFloats): at most two waves fit the 64 KB of a workgroup.
constexpr int kResidualWaves = 2;
constexpr int kMaxBlocksResidual = 512;     // residual adjoint: one wave per SIMD resident; more workgroups than CUs = load balancing (each wave owns N x 1617 partials)

// Residual adjoint, per wave: the residual jets (value + local gradient) the forward sweep leaves for the per-instance phase
// [round <= 4][N][64] float4 + a word per (round, instance) (render_kernels.h: jet_wave_float4s), and the seeds that phase leaves for the MLP adjoint [ray of batch][round <= 4][N][7][64].
size_t residual_jet_floats(int num_instances, bool residual) {
    return residual ? static_cast<size_t>(kMaxBlocksResidual) * kResidualWaves * jet_wave_float4s(4, num_instances) * 4 : 0;
}
size_t residual_cache_floats(int num_instances, bool residual) {
    return residual_jet_floats(num_instances, residual) +
           (residual ? static_cast<size_t>(kMaxBlocksResidual) * kResidualWaves * kMlpBatch * 4 * num_instances * kSeedFloats * kWave : 0);
}

size_t partial_floats(int num_instances, bool residual) {
    const size_t box = static_cast<size_t>(kMaxBlocks) * kMaxWavesPerBlock * num_instances * kGradStride;
    const size_t mlp = residual ? static_cast<size_t>(kMaxBlocksResidual) * kResidualWaves * num_instances * kMlpWeights : 0;
    return box + mlp + residual_cache_floats(num_instances, residual);
}

}  // namespace

extern "C" {

int32_t vsrd_abi_version(void) { return VSRD_ABI_VERSION; }

const char* vsrd_error_string(int32_t code) {
    switch (code) {
        case VSRD_OK: return "ok";
        case VSRD_E_INVALID_ARGUMENT: return "invalid argument";
        case VSRD_E_UNSUPPORTED: return "unsupported configuration";
        case VSRD_E_LAUNCH: return "HIP launch failed";
        case VSRD_E_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}

size_t vsrd_workspace_bytes(int32_t num_instances, int32_t residual) {
    if (num_instances < 1 || num_instances > VSRD_MAX_INSTANCES) return 0;
    return partial_floats(num_instances, residual != 0) * sizeof(float);
}

int32_t vsrd_ray_directions(const float* inverse_projection, int32_t num_views, int32_t height, int32_t width,
                            float* directions, void* stream) {
    if (!inverse_projection || !directions || num_views < 1 || height < 1 || width < 1) return VSRD_E_INVALID_ARGUMENT;
    const size_t total = static_cast<size_t>(num_views) * height * width;
    const int blocks = static_cast<int>(std::min<size_t>((total + 255) / 256, 8192));
    hipLaunchKernelGGL(ray_directions_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       inverse_projection, num_views, height, width, directions);
    return launch_status();
}

int32_t vsrd_field_eval(const vsrd_field* field, const float* positions, int64_t num_points,
                        float* distances, float* gradients, float* labels, int32_t hard_union, void* stream) {
    if (!valid_field(field) || (!positions && num_points > 0) || num_points < 0) return VSRD_E_INVALID_ARGUMENT;
    if (hard_union && labels) return VSRD_E_INVALID_ARGUMENT;
    if (num_points == 0) return VSRD_OK;
    const int blocks = static_cast<int>(std::min<int64_t>((num_points + 255) / 256, 8192));
    if (field->mlp_weights != nullptr)
        hipLaunchKernelGGL(field_eval_kernel<true>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), field_args(field),
                           field->instances, field->mlp_weights, positions, static_cast<long long>(num_points), distances, gradients, labels, hard_union);
    else
        hipLaunchKernelGGL(field_eval_kernel<false>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), field_args(field),
                           field->instances, field->mlp_weights, positions, static_cast<long long>(num_points), distances, gradients, labels, hard_union);
    return launch_status();
}

int32_t vsrd_field_eval_backward(const vsrd_field* field, const float* positions, int64_t num_points,
                                 const float* grad_distances, const float* grad_labels, int32_t hard_union,
                                 void* workspace, size_t workspace_bytes, float* grad_positions,
                                 float* grad_instances, float* grad_mlp_weights, void* stream) {
    if (!valid_field(field) || !grad_instances || num_points < 0 || (!positions && num_points > 0)) return VSRD_E_INVALID_ARGUMENT;
    if (hard_union && grad_labels) return VSRD_E_INVALID_ARGUMENT;
    const bool residual = field->mlp_weights != nullptr;
    if (residual && !grad_mlp_weights) return VSRD_E_INVALID_ARGUMENT;
    const int N = field->num_instances;
    if (!workspace || workspace_bytes < vsrd_workspace_bytes(N, residual)) return VSRD_E_WORKSPACE;
    const hipStream_t s = static_cast<hipStream_t>(stream);
    const int row = N * kGradStride, mlp_row = N * kMlpWeights;
    if (num_points == 0) {
        if (residual && hipMemsetAsync(grad_mlp_weights, 0, mlp_row * sizeof(float), s) != hipSuccess) return VSRD_E_LAUNCH;
        return hipMemsetAsync(grad_instances, 0, row * sizeof(float), s) == hipSuccess ? VSRD_OK : VSRD_E_LAUNCH;
    }
    Geometry g;
    const size_t per_wave = (static_cast<size_t>(residual ? kMlpLdsFloats : 0) + row + 3) & ~static_cast<size_t>(3);
    if (!plan(static_cast<int>(std::min<int64_t>((num_points + kWave - 1) / kWave, 1 << 30)), per_wave, &g)) return VSRD_E_UNSUPPORTED;
    if (residual && g.threads > kResidualWaves * kWave) return VSRD_E_UNSUPPORTED;
    g.blocks = std::min(g.blocks, residual ? kMaxBlocksResidual : 2048);
    float* partials = static_cast<float*>(workspace);
    float* mlp_partials = partials + static_cast<size_t>(kMaxBlocks) * kMaxWavesPerBlock * row;
    const FieldArgs f = field_args(field);
    if (residual)
        hipLaunchKernelGGL(field_eval_backward_kernel<true>, dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, f, field->instances, field->mlp_weights,
                           positions, static_cast<long long>(num_points), grad_distances, grad_labels, hard_union, grad_positions, partials, mlp_partials);
    else
        hipLaunchKernelGGL(field_eval_backward_kernel<false>, dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, f, field->instances, field->mlp_weights,
                           positions, static_cast<long long>(num_points), grad_distances, grad_labels, hard_union, grad_positions, partials, mlp_partials);
    if (launch_status() != VSRD_OK) return VSRD_E_LAUNCH;
    const int num_waves = g.blocks * (g.threads / kWave);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(row), dim3(256), 0, s, partials, num_waves, row, grad_instances);
    if (residual)
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(mlp_row), dim3(256), 0, s, mlp_partials, num_waves, mlp_row, grad_mlp_weights);
    return launch_status();
}

int32_t vsrd_sphere_trace(const vsrd_field* field, const float* origins, int32_t origin_stride, const float* directions,
                          const uint8_t* foreground, int64_t num_rays, int32_t num_iterations, float convergence_criteria,
                          float bounding_radius, int32_t initialise, int32_t hard_union,
                          float* positions, uint8_t* converged, void* stream) {
    if (!valid_field(field) || num_rays < 0 || num_iterations < 0 || (origin_stride != 0 && origin_stride != 3)) return VSRD_E_INVALID_ARGUMENT;
    if (num_rays == 0) return VSRD_OK;
    if (!origins || !directions || !positions || !converged) return VSRD_E_INVALID_ARGUMENT;
    const int blocks = static_cast<int>(std::min<int64_t>((num_rays + 255) / 256, 16384));
    const hipStream_t s = static_cast<hipStream_t>(stream);
    if (field->mlp_weights != nullptr)
        hipLaunchKernelGGL(sphere_trace_kernel<true>, dim3(blocks), dim3(256), 0, s, field_args(field), field->instances, field->mlp_weights,
                           origins, origin_stride, directions, foreground, static_cast<long long>(num_rays), num_iterations,
                           convergence_criteria, bounding_radius, initialise, hard_union, positions, converged);
    else
        hipLaunchKernelGGL(sphere_trace_kernel<false>, dim3(blocks), dim3(256), 0, s, field_args(field), field->instances, field->mlp_weights,
                           origins, origin_stride, directions, foreground, static_cast<long long>(num_rays), num_iterations,
                           convergence_criteria, bounding_radius, initialise, hard_union, positions, converged);
    return launch_status();
}

int32_t vsrd_polygon_soft_masks(const float* polygons, const int32_t* counts, int32_t num_polygons, int32_t max_vertices,
                                int32_t height, int32_t width, const uint8_t* inside, float temperature,
                                float* distance_maps, float* soft_masks, void* stream) {
    if (num_polygons < 0 || max_vertices < 1 || height < 1 || width < 1 || temperature <= 0.0f) return VSRD_E_INVALID_ARGUMENT;
    if (num_polygons == 0) return VSRD_OK;
    if (!polygons || !counts || (!distance_maps && !soft_masks) || (soft_masks && !inside)) return VSRD_E_INVALID_ARGUMENT;
    const int pixels = height * width;
    const dim3 grid(static_cast<unsigned>(std::min((pixels + 255) / 256, 4096)), static_cast<unsigned>(num_polygons));
    hipLaunchKernelGGL(polygon_soft_mask_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), polygons, counts, num_polygons,
                       max_vertices, height, width, inside, temperature, distance_maps, soft_masks);
    return launch_status();
}

int32_t vsrd_sample_stratified(const vsrd_render_config* config, const float* u_coarse, float* distances, void* stream) {
    if (config != nullptr && config->num_frames > 1) return VSRD_E_UNSUPPORTED;      // (frame batches: the per-frame step entry points only)
    if (!valid_config(config)) return VSRD_E_INVALID_ARGUMENT;
    if (config->num_rays == 0) return VSRD_OK;
    if (!u_coarse || !distances) return VSRD_E_INVALID_ARGUMENT;
    const size_t total = static_cast<size_t>(config->num_rays) * config->num_samples;
    const int blocks = static_cast<int>(std::min<size_t>((total + 255) / 256, 8192));
    hipLaunchKernelGGL(sample_stratified_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       render_args(config), u_coarse, distances);
    return launch_status();
}

int32_t vsrd_sample_importance(const vsrd_render_config* config, const float* coarse_distances,
                               const float* coarse_weights, const float* u_fine, float* merged, float* fine, void* stream) {
    if (config != nullptr && config->num_frames > 1) return VSRD_E_UNSUPPORTED;      // (frame batches: the per-frame step entry points only)
    if (!valid_config(config)) return VSRD_E_INVALID_ARGUMENT;
    if (config->num_rays == 0) return VSRD_OK;
    if (!coarse_distances || !coarse_weights || !u_fine || (!merged && !fine)) return VSRD_E_INVALID_ARGUMENT;
    Geometry g;
    if (!plan(config->num_rays, wave_lds_floats(config->num_samples, 0), &g)) return VSRD_E_UNSUPPORTED;
    const RenderArgs c = render_args(config);
    const hipStream_t s = static_cast<hipStream_t>(stream);
    const int rounds = rounds_for(config->num_samples);
#define VSRD_LAUNCH(K)                                                                                                  \
    hipLaunchKernelGGL(sample_importance_kernel<K>, dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, c, coarse_distances, \
                       coarse_weights, u_fine, merged, fine)
    switch (rounds) {
        case 1: VSRD_LAUNCH(1); break;
        case 2: VSRD_LAUNCH(2); break;
        case 4: VSRD_LAUNCH(4); break;
        default: return VSRD_E_UNSUPPORTED;
    }
#undef VSRD_LAUNCH
    return launch_status();
}

int32_t vsrd_render_forward(const vsrd_field* field, const vsrd_render_config* config,
                            const float* origins, const float* directions,
                            const float* distances, int32_t num_distances,
                            float* labels, float* gradients, float* weights, void* stream) {
    if (config != nullptr && config->num_frames > 1) return VSRD_E_UNSUPPORTED;      // (frame batches: the per-frame step entry points only)
    if (!valid_field(field) || !valid_config(config) || num_distances < 2 || num_distances > 2 * VSRD_MAX_SAMPLES)
        return VSRD_E_INVALID_ARGUMENT;
    if (config->num_rays == 0) return VSRD_OK;   // empty batch: buffers may be null
    if (!origins || !directions || !distances || !labels) return VSRD_E_INVALID_ARGUMENT;
    const bool residual = field->mlp_weights != nullptr;
    Geometry g;
    if (!plan(config->num_rays, static_cast<size_t>(forward_lds_floats(num_distances, field->num_instances, residual)), &g)) return VSRD_E_UNSUPPORTED;
    const FieldArgs f = field_args(field);
    RenderArgs c = render_args(config);
    c.sh.inv_t = f.inv_t;
    const hipStream_t s = static_cast<hipStream_t>(stream);
#define VSRD_LAUNCH(K, RES)                                                                                                  \
    do {                                                                                                                       \
        if (opt_in_lds(render_forward_kernel<K, RES>, g.lds_bytes) != VSRD_OK) return VSRD_E_LAUNCH;                         \
        fit_to_residency(render_forward_kernel<K, RES>, &g);                                                                  \
        hipLaunchKernelGGL((render_forward_kernel<K, RES>), dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, f, field->instances, \
                           field->mlp_weights, c, origins, directions, distances, num_distances, labels, gradients, weights);   \
    } while (0)
    const int rounds = rounds_for(num_distances - 1);
    if (residual) {
        switch (rounds) {
            case 1: VSRD_LAUNCH(1, true); break;
            case 2: VSRD_LAUNCH(2, true); break;
            case 4: VSRD_LAUNCH(4, true); break;
            default: return VSRD_E_UNSUPPORTED;
        }
    } else {
        switch (rounds) {
            case 1: VSRD_LAUNCH(1, false); break;
            case 2: VSRD_LAUNCH(2, false); break;
            case 4: VSRD_LAUNCH(4, false); break;
            case 8: VSRD_LAUNCH(8, false); break;
            default: return VSRD_E_UNSUPPORTED;
        }
    }
#undef VSRD_LAUNCH
    return launch_status();
}

namespace {
// (defined with the residual step's launch plan, below)
int32_t render_backward_split(const vsrd_field* field, const vsrd_render_config* config, const float* origins, const float* directions,
                              const float* distances, int32_t num_distances, const float* grad_labels, const float* grad_gradients,
                              const float* grad_weights, void* workspace, size_t workspace_bytes, float* grad_instances, float* grad_mlp_weights,
                              void* stream, bool* taken);
}

int32_t vsrd_render_backward(const vsrd_field* field, const vsrd_render_config* config,
                             const float* origins, const float* directions,
                             const float* distances, int32_t num_distances,
                             const float* grad_labels, const float* grad_gradients, const float* grad_weights,
                             void* workspace, size_t workspace_bytes,
                             float* grad_instances, float* grad_mlp_weights, void* stream) {
    if (config != nullptr && config->num_frames > 1) return VSRD_E_UNSUPPORTED;      // (frame batches: the per-frame step entry points only)
    if (!valid_field(field) || !valid_config(config) || !grad_instances || !workspace || num_distances < 2 ||
        num_distances > 2 * VSRD_MAX_SAMPLES)
        return VSRD_E_INVALID_ARGUMENT;
    if (config->num_rays > 0 && (!origins || !directions || !distances || !grad_labels)) return VSRD_E_INVALID_ARGUMENT;
    const bool residual = field->mlp_weights != nullptr;
    if (residual && !grad_mlp_weights) return VSRD_E_INVALID_ARGUMENT;
    const int N = field->num_instances;
    if (workspace_bytes < vsrd_workspace_bytes(N, residual)) return VSRD_E_WORKSPACE;
    if (residual && config->num_rays > 0 && !(config->flags & VSRD_FLAG_RESIDUAL_SINGLE_KERNEL)) {
        // two kernels per chunk of rays (front part + MLP adjoint by instance) when the workspace holds a chunk's seeds
        bool taken = false;
        const int32_t status = render_backward_split(field, config, origins, directions, distances, num_distances, grad_labels, grad_gradients,
                                                     grad_weights, workspace, workspace_bytes, grad_instances, grad_mlp_weights, stream, &taken);
        if (taken) return status;
    }
    const hipStream_t s = static_cast<hipStream_t>(stream);
    const int row = N * kGradStride;
    const int mlp_row = N * kMlpWeights;
    if (config->num_rays == 0) {
        if (residual && hipMemsetAsync(grad_mlp_weights, 0, mlp_row * sizeof(float), s) != hipSuccess) return VSRD_E_LAUNCH;
        return hipMemsetAsync(grad_instances, 0, row * sizeof(float), s) == hipSuccess ? VSRD_OK : VSRD_E_LAUNCH;
    }
    Geometry g;
    // box-only fields, label adjoints only: the mappings of the fused step (quad_step.h) -- four rays per wave for N <= 16 and <= 128
    // distances, two for N <= 64 and <= 256
    if (!residual && !grad_gradients && !grad_weights && !(config->flags & VSRD_FLAG_STEP_WAVE_PER_RAY) && N <= kPairMaxInstances &&
        num_distances <= 2 * kPairMaxSamples) {
        const bool quad = N <= kQuadMaxInstances && num_distances <= 2 * kQuadMaxSamples;
        const int lanes = quad ? kRowLanes : 32, rays_per_wave = kWave / lanes, half = (num_distances + 1) / 2;
        const int shape_instances = quad ? kQuadMaxInstances : kPairMaxInstances;
        const bool full = num_distances == 8 * lanes && 2 * N > shape_instances && !switches().no_full_shape;      // (2 S = 8 x lanes: BASELINE configs 2 and 5)
        if (!plan((config->num_rays + rays_per_wave - 1) / rays_per_wave, static_cast<size_t>(quad_lds_floats(half, full ? shape_instances : N, lanes)), &g))
            return VSRD_E_UNSUPPORTED;
        const FieldArgs f = field_args(field);
        RenderArgs c = render_args(config);
        c.sh.inv_t = f.inv_t;
        float* partials = static_cast<float*>(workspace);
        // two kernels on one grid (quad_step.h: backward_rows_kernel_body): behind the waves' partial rows, two summary words and one byte per group of rays
        const size_t num_groups = (static_cast<size_t>(config->num_rays) + rays_per_wave - 1) / rays_per_wave;
        const size_t flag_bytes = 16 + ((num_groups + 15) & ~static_cast<size_t>(15));
        const int waves_per_block = g.threads / kWave;
        if (flag_bytes + static_cast<size_t>(waves_per_block) * row * sizeof(float) > workspace_bytes) return VSRD_E_WORKSPACE;
        const int max_waves = static_cast<int>((workspace_bytes - flag_bytes) / sizeof(float) / row);
        if (g.blocks * waves_per_block > max_waves) g.blocks = max_waves / waves_per_block;
        unsigned char* redo_flags = reinterpret_cast<unsigned char*>(partials + static_cast<size_t>(g.blocks) * waves_per_block * row);
        if (hipMemsetAsync(redo_flags, 0, 16, s) != hipSuccess) return VSRD_E_LAUNCH;
#define VSRD_LAUNCH_BACKWARD_ROWS(KERNEL, FULL)                                                                                           \
        hipLaunchKernelGGL((KERNEL<true, FULL>), dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, f, field->instances, c, origins, directions,   \
                           distances, num_distances, grad_labels, partials, redo_flags);                                                \
        hipLaunchKernelGGL((KERNEL<false, false>), dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, f, field->instances, c, origins, directions, \
                           distances, num_distances, grad_labels, partials, redo_flags)
        if (quad) { if (full) { VSRD_LAUNCH_BACKWARD_ROWS(render_backward_quad_kernel, true); } else { VSRD_LAUNCH_BACKWARD_ROWS(render_backward_quad_kernel, false); } }
        else { if (full) { VSRD_LAUNCH_BACKWARD_ROWS(render_backward_pair_kernel, true); } else { VSRD_LAUNCH_BACKWARD_ROWS(render_backward_pair_kernel, false); } }
#undef VSRD_LAUNCH_BACKWARD_ROWS
        if (launch_status() != VSRD_OK) return VSRD_E_LAUNCH;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(row), dim3(256), 0, s, partials, g.blocks * (g.threads / kWave), row, grad_instances);
        return launch_status();
    }
    if (!plan(config->num_rays, static_cast<size_t>(backward_lds_floats(num_distances, N, residual)), &g)) return VSRD_E_UNSUPPORTED;
    if (residual && g.threads > kResidualWaves * kWave) return VSRD_E_UNSUPPORTED;      // cannot happen (LDS per wave), but the scratch is sized for it
    if (residual && g.blocks > kMaxBlocksResidual) g.blocks = kMaxBlocksResidual;
    const FieldArgs f = field_args(field);
    RenderArgs c = render_args(config);
    c.sh.inv_t = f.inv_t;
    float* partials = static_cast<float*>(workspace);
    float* mlp_partials = partials + static_cast<size_t>(kMaxBlocks) * kMaxWavesPerBlock * row;
    float* jets = mlp_partials + (residual ? static_cast<size_t>(kMaxBlocksResidual) * kResidualWaves * mlp_row : 0);
    float4* residual_cache = reinterpret_cast<float4*>(jets);
    float* seed_cache = jets + residual_jet_floats(N, residual);
#define VSRD_LAUNCH(K, RES)                                                                                                    \
    fit_to_residency(render_backward_kernel<K, RES>, &g);                                                                        \
    if (residual && g.blocks > kMaxBlocksResidual) g.blocks = kMaxBlocksResidual;                                                \
    hipLaunchKernelGGL((render_backward_kernel<K, RES>), dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, f, field->instances,    \
                       field->mlp_weights, c, origins, directions, distances, num_distances, grad_labels, grad_gradients,         \
                       grad_weights, partials, mlp_partials, residual_cache, seed_cache)
    const int rounds = rounds_for(num_distances - 1);
    if (residual) {
        switch (rounds) {
            case 1: VSRD_LAUNCH(1, true); break;
            case 2: VSRD_LAUNCH(2, true); break;
            case 4: VSRD_LAUNCH(4, true); break;
            default: return VSRD_E_UNSUPPORTED;
        }
    } else {
        switch (rounds) {
            case 1: VSRD_LAUNCH(1, false); break;
            case 2: VSRD_LAUNCH(2, false); break;
            case 4: VSRD_LAUNCH(4, false); break;
            default: return VSRD_E_UNSUPPORTED;
        }
    }
#undef VSRD_LAUNCH
    if (launch_status() != VSRD_OK) return VSRD_E_LAUNCH;
    const int num_waves = g.blocks * (g.threads / kWave);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(row), dim3(256), 0, s, partials, num_waves, row, grad_instances);
    if (residual)
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(mlp_row), dim3(256), 0, s, mlp_partials, num_waves, mlp_row, grad_mlp_weights);
    return launch_status();
}

int32_t vsrd_render_hierarchical_forward(const vsrd_field* field, const vsrd_render_config* config,
                                         const float* origins, const float* directions,
                                         const float* u_coarse, const float* u_fine,
                                         float* labels, float* distances, float* gradients, float* weights, float* coarse_weights,
                                         float* u_coarse_out, float* u_fine_out, void* stream) {
    if (config != nullptr && config->num_frames > 1) return VSRD_E_UNSUPPORTED;      // (frame batches: the per-frame step entry points only)
    if (!valid_field(field) || !valid_config(config)) return VSRD_E_INVALID_ARGUMENT;
    if (config->num_rays == 0) return VSRD_OK;   // empty batch: buffers may be null
    if (!origins || !directions || !labels) return VSRD_E_INVALID_ARGUMENT;
    const bool residual = field->mlp_weights != nullptr;
    if ((config->flags & VSRD_FLAG_SKIP_EXACT_MISSES) && (gradients || weights)) return VSRD_E_INVALID_ARGUMENT;
    const int S = config->num_samples;
    Geometry g;
    const FieldArgs f = field_args(field);
    RenderArgs c = render_args(config);
    c.sh.inv_t = f.inv_t;
    const hipStream_t s = static_cast<hipStream_t>(stream);
    // labels, distances, pass 1's weights and the uniforms only (no per-sample gradients / weights of pass 2), box-only field: the forward
    // in the mappings of the fused step (quad_step.h) -- four rays per wave for N <= 16 and S <= 64, two for N <= 64 and S <= 128
    if (!residual && !gradients && !weights && !(config->flags & VSRD_FLAG_STEP_WAVE_PER_RAY) &&
        S <= kPairMaxSamples && field->num_instances <= kPairMaxInstances) {
        const bool quad = S <= kQuadMaxSamples && field->num_instances <= kQuadMaxInstances;
        const int lanes = quad ? kRowLanes : 32, rays_per_wave = kWave / lanes;
        // two kernels on one grid, the hot one instantiated for a full shape (quad_step.h: hierarchical_rows_kernel_body), as in vsrd_render_silhouette_step
        const int shape_instances = quad ? kQuadMaxInstances : kPairMaxInstances;
        const bool full = S == 4 * lanes && 2 * field->num_instances > shape_instances && !switches().no_full_shape;
        if (!plan((config->num_rays + rays_per_wave - 1) / rays_per_wave,
                  static_cast<size_t>(quad_lds_floats(S, full ? shape_instances : field->num_instances, lanes)), &g))
            return VSRD_E_UNSUPPORTED;
#define VSRD_LAUNCH_FORWARD_ROWS(KERNEL, FULL)                                                                                            \
        hipLaunchKernelGGL((KERNEL<true, FULL>), dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, f, field->instances, c, origins,         \
                           directions, u_coarse, u_fine, labels, distances, coarse_weights, u_coarse_out, u_fine_out);                   \
        hipLaunchKernelGGL((KERNEL<false, false>), dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, f, field->instances, c, origins,       \
                           directions, u_coarse, u_fine, labels, distances, coarse_weights, u_coarse_out, u_fine_out)
        if (quad) { if (full) { VSRD_LAUNCH_FORWARD_ROWS(render_hierarchical_quad_kernel, true); } else { VSRD_LAUNCH_FORWARD_ROWS(render_hierarchical_quad_kernel, false); } }
        else { if (full) { VSRD_LAUNCH_FORWARD_ROWS(render_hierarchical_pair_kernel, true); } else { VSRD_LAUNCH_FORWARD_ROWS(render_hierarchical_pair_kernel, false); } }
#undef VSRD_LAUNCH_FORWARD_ROWS
        return launch_status();
    }
    if (!plan(config->num_rays, static_cast<size_t>(hierarchical_lds_floats(S, field->num_instances, residual)), &g)) return VSRD_E_UNSUPPORTED;
#define VSRD_LAUNCH(K, RES)                                                                                                  \
    do {                                                                                                                       \
        if (opt_in_lds(render_hierarchical_kernel<K, RES>, g.lds_bytes) != VSRD_OK) return VSRD_E_LAUNCH;                    \
        fit_to_residency(render_hierarchical_kernel<K, RES>, &g);                                                             \
        hipLaunchKernelGGL((render_hierarchical_kernel<K, RES>), dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, f,            \
                           field->instances, field->mlp_weights, c, origins, directions, u_coarse, u_fine, labels, distances,   \
                           gradients, weights, coarse_weights, u_coarse_out, u_fine_out);                                                     \
    } while (0)
    const int rounds = rounds_for(2 * S - 1);
    if (residual) {
        switch (rounds) {
            case 1: VSRD_LAUNCH(1, true); break;
            case 2: VSRD_LAUNCH(2, true); break;
            case 4: VSRD_LAUNCH(4, true); break;
            default: return VSRD_E_UNSUPPORTED;
        }
    } else {
        switch (rounds) {
            case 1: VSRD_LAUNCH(1, false); break;
            case 2: VSRD_LAUNCH(2, false); break;
            case 4: VSRD_LAUNCH(4, false); break;
            case 8: VSRD_LAUNCH(8, false); break;
            default: return VSRD_E_UNSUPPORTED;
        }
    }
#undef VSRD_LAUNCH
    return launch_status();
}

int32_t vsrd_project_boxes_forward(const float* world_corners, const float* extrinsics, const float* intrinsics,
                                   const int32_t* edges, int32_t num_edges, int32_t num_views, int32_t num_boxes,
                                   int32_t height, int32_t width, float epsilon,
                                   float* boxes_2d, float* camera_corners, int32_t* selection, void* stream) {
    if (num_views < 0 || num_boxes < 0 || num_edges < 1 || num_edges > kMaxEdges) return VSRD_E_INVALID_ARGUMENT;
    if (num_views == 0 || num_boxes == 0) return VSRD_OK;
    if (!world_corners || !extrinsics || !intrinsics || !edges || !boxes_2d || !selection) return VSRD_E_INVALID_ARGUMENT;
    const int total = num_views * num_boxes;
    hipLaunchKernelGGL(project_boxes_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       world_corners, extrinsics, intrinsics, edges, num_edges, num_views, num_boxes,
                       static_cast<float>(height), static_cast<float>(width), epsilon, boxes_2d, camera_corners, selection);
    return launch_status();
}

int32_t vsrd_project_boxes_backward(const float* world_corners, const float* extrinsics, const float* intrinsics,
                                    const int32_t* edges, int32_t num_edges, int32_t num_views, int32_t num_boxes, float epsilon,
                                    const float* grad_boxes_2d, const int32_t* selection, float* grad_world_per_view, void* stream) {
    if (num_views < 0 || num_boxes < 0 || num_edges < 1 || num_edges > kMaxEdges) return VSRD_E_INVALID_ARGUMENT;
    if (num_views == 0 || num_boxes == 0) return VSRD_OK;
    if (!world_corners || !extrinsics || !intrinsics || !edges || !grad_boxes_2d || !selection || !grad_world_per_view)
        return VSRD_E_INVALID_ARGUMENT;
    const int total = num_views * num_boxes;
    hipLaunchKernelGGL(project_boxes_backward_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       world_corners, extrinsics, intrinsics, edges, num_views, num_boxes, epsilon, grad_boxes_2d, selection,
                       grad_world_per_view);
    return launch_status();
}

namespace {
constexpr int kSplitBlocks = 1024;                       // render_silhouette_split_kernel: x 2 waves = two waves on each of the 1024 SIMDs
int split_max_rays() { return switches().split_max_rays; }   // box-only step: launches of at most this many rays split every ray over two waves (0: never)
}  // namespace

int32_t vsrd_render_silhouette_step(const vsrd_field* field, const vsrd_render_config* config,
                                    const float* origins, const float* directions, const float* u_coarse, const float* u_fine,
                                    const float* targets, const float* instance_weights, float loss_scale,
                                    void* workspace, size_t workspace_bytes,
                                    float* loss, float* grad_instances, float* labels, void* stream) {
    if (!valid_field(field) || !valid_config(config, true, true) || !workspace || !loss || !grad_instances) return VSRD_E_INVALID_ARGUMENT;
    if (field->mlp_weights != nullptr) return VSRD_E_UNSUPPORTED;            // box-only fast path
    const int N = field->num_instances;
    if (workspace_bytes < vsrd_workspace_bytes(N, 0)) return VSRD_E_WORKSPACE;
    const hipStream_t s = static_cast<hipStream_t>(stream);
    const int row = N * kGradStride;
    Frames frames;
    frames_of(config->num_frames, config->frame_stride, &frames);           // (validated by valid_config)
    if (frames.count > 1 && (config->num_rays == 0 || wants_samples(config))) return VSRD_E_UNSUPPORTED;
    if (config->num_rays == 0) {
        if (hipMemsetAsync(loss, 0, sizeof(float), s) != hipSuccess) return VSRD_E_LAUNCH;
        return hipMemsetAsync(grad_instances, 0, row * sizeof(float), s) == hipSuccess ? VSRD_OK : VSRD_E_LAUNCH;
    }
    if (!origins || !directions || !targets) return VSRD_E_INVALID_ARGUMENT;
    const int S = config->num_samples;
    const int rounds = rounds_for(2 * S - 1);
    if (rounds < 1 || rounds > 4) return VSRD_E_UNSUPPORTED;
    // dense launches of the benchmark shapes: four neighbouring rays per wave (quad_step.h).  Gathered rays (the reference's 1000
    // importance-sampled rays per step) are neither neighbours nor enough to fill the chip four to a wave.
    const bool force_split = (config->flags & VSRD_FLAG_STEP_SPLIT_RAY) && !(config->flags & VSRD_FLAG_STEP_WAVE_PER_RAY);
    // (dense = rays AND targets read row by row: the multi-ray kernels know neither ray_indices nor the target column map)
    const bool dense = config->ray_indices == nullptr && config->target_columns == nullptr && !(config->flags & VSRD_FLAG_STEP_WAVE_PER_RAY) && !force_split;
    const bool quad = dense && S <= kQuadMaxSamples && N <= kQuadMaxInstances;
    // ... and for more instances or samples than that (BASELINE config 5: N = 64, S = 128) two rays per wave, 32 lanes each
    const bool pair = dense && !quad && S <= kPairMaxSamples && N <= kPairMaxInstances;
    // ... and for launches too small to put two waves on every SIMD one ray per wave (the reference's 1000 sampled rays per step: 1000
    // waves on 1024 SIMDs, pure latency), a ray split over the two waves of a workgroup (render_silhouette_split_kernel)
    // (a step that writes its samples out -- vsrd_render_config::out_* -- keeps one ray per wave: the split form has no such outputs)
    const bool split = !quad && !pair && !(config->flags & VSRD_FLAG_STEP_WAVE_PER_RAY) && !wants_samples(config) && (rounds == 2 || rounds == 4) &&
                       (force_split || config->num_rays <= split_max_rays()) &&
                       static_cast<size_t>(split_lds_floats(S, N)) * sizeof(float) <= kLdsLimit / 2;
    if (split) {
        const int blocks = config->num_rays < kSplitBlocks ? config->num_rays : kSplitBlocks;
        const int num_waves = blocks * kPairWaves;
        if (static_cast<size_t>(num_waves) * (row + 1) * sizeof(float) > workspace_bytes) return VSRD_E_WORKSPACE;
        if (frames.count > 1 && static_cast<long long>(workspace_bytes) > frames.stride) return VSRD_E_INVALID_ARGUMENT;      // (the frames' workspaces would overlap)
        const size_t lds_bytes = static_cast<size_t>(split_lds_floats(S, N)) * sizeof(float);
        const FieldArgs f = field_args(field);
        RenderArgs c = render_args(config);
        c.sh.inv_t = f.inv_t;
        float* partials = static_cast<float*>(workspace);
        float* loss_partials = partials + static_cast<size_t>(num_waves) * row;
#define VSRD_LAUNCH_SPLIT(K, FRAMES)                                                                                            \
        do {                                                                                                                      \
            if (opt_in_lds(render_silhouette_split_kernel<K, FRAMES>, lds_bytes) != VSRD_OK) return VSRD_E_LAUNCH;               \
            hipLaunchKernelGGL((render_silhouette_split_kernel<K, FRAMES>), dim3(blocks, frames.count), dim3(kPairWaves * kWave), lds_bytes, s, f, field->instances, c, \
                               origins, directions, u_coarse, u_fine, targets, instance_weights, loss_scale, labels, partials, loss_partials); \
        } while (0)
        if (frames.count > 1) { if (rounds == 2) VSRD_LAUNCH_SPLIT(2, true); else VSRD_LAUNCH_SPLIT(4, true); }
        else if (rounds == 2) VSRD_LAUNCH_SPLIT(2, false); else VSRD_LAUNCH_SPLIT(4, false);
#undef VSRD_LAUNCH_SPLIT
        if (launch_status() != VSRD_OK) return VSRD_E_LAUNCH;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(row + 1, frames.count), dim3(256), 0, s, partials, num_waves, row, grad_instances, nullptr, loss_partials, 1, loss,
                           frames.stride);
        return launch_status();
    }
    if (frames.count > 1) return VSRD_E_UNSUPPORTED;     // a batch of frames runs in the split-ray form only (gathered launches of at most 2048 rays per frame)
    const int lanes = quad ? kRowLanes : 32, rays_per_wave = kWave / lanes;
    // a launch that fills its shape (S = lanes x rounds of the shape, more than half of its instance slots: BASELINE configs 2 and 5)
    // runs the hot kernel instantiated for that S with the instance tables padded to the shape's count (quad_step.h: kFull)
    const int shape_instances = quad ? kQuadMaxInstances : kPairMaxInstances;
    // (exactly the launches the dispatch below gives a <4, true, true> kernel: S = 4 x lanes)
    const bool full = (quad || pair) && S == 4 * lanes && 2 * N > shape_instances && !switches().no_full_shape;
    Geometry g;
    const size_t per_wave = (quad || pair) ? static_cast<size_t>(quad_lds_floats(S, full ? shape_instances : N, lanes)) : static_cast<size_t>(wave_lds_floats(S, N)) + N + row;
    if (!plan((quad || pair) ? (config->num_rays + rays_per_wave - 1) / rays_per_wave : config->num_rays, per_wave, &g)) return VSRD_E_UNSUPPORTED;
#ifdef VSRD_INSTANCE_LDS
    if (quad && full) g.lds_bytes += static_cast<size_t>(N) * kInstanceStride * sizeof(float);      // experiment: the instance block next to the waves' partitions
#endif
    // the partial buffer holds one gradient row per wave; the loss partials live in the tail of the same row budget, and behind them
    // (multi-ray mappings) one byte per group of rays: "the hot kernel could not serve this group" (quad_step.h)
    const size_t num_groups = (quad || pair) ? (static_cast<size_t>(config->num_rays) + rays_per_wave - 1) / rays_per_wave : 0;
    const size_t flag_bytes = (quad || pair) ? 16 + ((num_groups + 15) & ~static_cast<size_t>(15)) : 0;      // summary words + one byte per group
    if (flag_bytes + static_cast<size_t>(g.threads / kWave) * (row + 1) * sizeof(float) > vsrd_workspace_bytes(N, 0)) return VSRD_E_WORKSPACE;
    const int max_waves_for_loss = static_cast<int>((vsrd_workspace_bytes(N, 0) - flag_bytes) / sizeof(float) / (row + 1));
    if (g.blocks * (g.threads / kWave) > max_waves_for_loss) g.blocks = max_waves_for_loss / (g.threads / kWave);
    const FieldArgs f = field_args(field);
    RenderArgs c = render_args(config);
    c.sh.inv_t = f.inv_t;
    const int num_waves = g.blocks * (g.threads / kWave);
    float* partials = static_cast<float*>(workspace);
    float* loss_partials = partials + static_cast<size_t>(num_waves) * row;
    unsigned char* redo_flags = reinterpret_cast<unsigned char*>(loss_partials + num_waves);
#define VSRD_LAUNCH(K)                                                                                                          \
    do {                                                                                                                          \
        if (opt_in_lds(render_silhouette_kernel<K>, g.lds_bytes) != VSRD_OK) return VSRD_E_LAUNCH;                               \
        hipLaunchKernelGGL(render_silhouette_kernel<K>, dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, f, field->instances, c,    \
                           origins, directions, u_coarse, u_fine, targets, instance_weights, loss_scale, labels, partials,        \
                           loss_partials);                                                                                       \
    } while (0)
// two kernels on one grid: the hot body (rotations about y, fixed soft-min shift), then everything it left over (quad_step.h)
#define VSRD_LAUNCH_ROWS(KERNEL, K, FULL)                                                                                         \
    do {                                                                                                                          \
        if (opt_in_lds(KERNEL<K, true, FULL>, g.lds_bytes) != VSRD_OK || opt_in_lds(KERNEL<K, false, false>, g.lds_bytes) != VSRD_OK) return VSRD_E_LAUNCH; \
        if (hipMemsetAsync(redo_flags, 0, 16, s) != hipSuccess) return VSRD_E_LAUNCH;                                               \
        hipLaunchKernelGGL((KERNEL<K, true, FULL>), dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, f, field->instances, c,        \
                           origins, directions, u_coarse, u_fine, targets, instance_weights, loss_scale, labels, partials,        \
                           loss_partials, redo_flags);                                                                           \
        hipLaunchKernelGGL((KERNEL<K, false, false>), dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, f, field->instances, c,      \
                           origins, directions, u_coarse, u_fine, targets, instance_weights, loss_scale, labels, partials,        \
                           loss_partials, redo_flags);                                                                           \
    } while (0)
#define VSRD_LAUNCH_QUAD(K) VSRD_LAUNCH_ROWS(render_silhouette_quad_kernel, K, false)
#define VSRD_LAUNCH_PAIR(K) VSRD_LAUNCH_ROWS(render_silhouette_pair_kernel, K, false)
    if (quad) {
        if (S <= 16) VSRD_LAUNCH_QUAD(1);
        else if (S <= 32) VSRD_LAUNCH_QUAD(2);
        else if (full) VSRD_LAUNCH_ROWS(render_silhouette_quad_kernel, 4, true);
        else VSRD_LAUNCH_QUAD(4);
    } else if (pair) {
        if (S <= 64) VSRD_LAUNCH_PAIR(2);
        else if (full) VSRD_LAUNCH_ROWS(render_silhouette_pair_kernel, 4, true);
        else VSRD_LAUNCH_PAIR(4);
    } else {
        switch (rounds) {
            case 1: VSRD_LAUNCH(1); break;
            case 2: VSRD_LAUNCH(2); break;
            case 4: VSRD_LAUNCH(4); break;
            default: return VSRD_E_UNSUPPORTED;
        }
    }
#undef VSRD_LAUNCH
#undef VSRD_LAUNCH_QUAD
#undef VSRD_LAUNCH_PAIR
#undef VSRD_LAUNCH_ROWS
    if (launch_status() != VSRD_OK) return VSRD_E_LAUNCH;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(row + 1), dim3(256), 0, s, partials, num_waves, row, grad_instances, nullptr, loss_partials, 1, loss);
    return launch_status();
}

static int32_t residual_step_single_kernel(const vsrd_field* field, const vsrd_render_config* config,
                                  const float* origins, const float* directions, const float* u_coarse, const float* u_fine,
                                  const float* targets, const float* instance_weights, float loss_scale, float eikonal_ratio,
                                  void* workspace, size_t workspace_bytes,
                                  float* losses, float* grad_instances, float* grad_mlp_weights, float* labels, void* stream) {
    if (!valid_field(field) || !valid_config(config, true) || !workspace || !losses || !grad_instances || !grad_mlp_weights) return VSRD_E_INVALID_ARGUMENT;
    if (field->mlp_weights == nullptr) return VSRD_E_INVALID_ARGUMENT;       // box-only fields: vsrd_render_silhouette_step
    const int N = field->num_instances;
    if (workspace_bytes < vsrd_workspace_bytes(N, 1)) return VSRD_E_WORKSPACE;
    const hipStream_t s = static_cast<hipStream_t>(stream);
    const int row = N * kGradStride, mlp_row = N * kMlpWeights;
    if (config->num_rays == 0) {
        if (hipMemsetAsync(losses, 0, 2 * sizeof(float), s) != hipSuccess) return VSRD_E_LAUNCH;
        if (hipMemsetAsync(grad_mlp_weights, 0, mlp_row * sizeof(float), s) != hipSuccess) return VSRD_E_LAUNCH;
        return hipMemsetAsync(grad_instances, 0, row * sizeof(float), s) == hipSuccess ? VSRD_OK : VSRD_E_LAUNCH;
    }
    if (!origins || !directions || !targets) return VSRD_E_INVALID_ARGUMENT;
    const int S = config->num_samples;
    const int num_points = 2 * S - 1;
    const int rounds = rounds_for(num_points);
    Geometry g;
    if (!plan(config->num_rays, static_cast<size_t>(residual_step_lds_floats(S, N)), &g)) return VSRD_E_UNSUPPORTED;
    if (g.threads > kResidualWaves * kWave) return VSRD_E_UNSUPPORTED;
    if (g.blocks > kMaxBlocksResidual) g.blocks = kMaxBlocksResidual;
    const FieldArgs f = field_args(field);
    RenderArgs c = render_args(config);
    c.sh.inv_t = f.inv_t;
    const int num_waves = g.blocks * (g.threads / kWave);
    float* partials = static_cast<float*>(workspace);
    float* loss_partials = partials + static_cast<size_t>(num_waves) * row;             // inside the (much larger) box-partial region
    float* mlp_partials = partials + static_cast<size_t>(kMaxBlocks) * kMaxWavesPerBlock * row;
    float* jets = mlp_partials + static_cast<size_t>(kMaxBlocksResidual) * kResidualWaves * mlp_row;
    float4* residual_cache = reinterpret_cast<float4*>(jets);
    float* seed_cache = jets + residual_jet_floats(N, true);
    // d/dg of  eikonal_ratio * mean over [R, 2S-1] of (|g| - 1)^2 ;  the reported eikonal term is the plain mean
    const float eikonal_norm = 1.0f / (static_cast<float>(config->num_rays) * static_cast<float>(num_points));
    const float eikonal_scale = eikonal_ratio * eikonal_norm;
#define VSRD_LAUNCH(K)                                                                                                             \
    do {                                                                                                                             \
        if (opt_in_lds(render_residual_step_kernel<K>, g.lds_bytes) != VSRD_OK) return VSRD_E_LAUNCH;                                \
        hipLaunchKernelGGL(render_residual_step_kernel<K>, dim3(g.blocks), dim3(g.threads), g.lds_bytes, s, f, field->instances,       \
                           field->mlp_weights, c, origins, directions, u_coarse, u_fine, targets, instance_weights, loss_scale,       \
                           eikonal_scale, eikonal_norm, labels, partials, mlp_partials, residual_cache, seed_cache, loss_partials);  \
    } while (0)
    switch (rounds) {
        case 1: VSRD_LAUNCH(1); break;
        case 2: VSRD_LAUNCH(2); break;
        case 4: VSRD_LAUNCH(4); break;
        default: return VSRD_E_UNSUPPORTED;
    }
#undef VSRD_LAUNCH
    if (launch_status() != VSRD_OK) return VSRD_E_LAUNCH;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(row), dim3(256), 0, s, partials, num_waves, row, grad_instances);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(mlp_row), dim3(256), 0, s, mlp_partials, num_waves, mlp_row, grad_mlp_weights);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(2), dim3(256), 0, s, loss_partials, num_waves, 2, losses);
    return launch_status();
}


// ---- the split form (render_kernels.h: residual_step_front_kernel + residual_mlp_adjoint_kernel), the default ---------------------
namespace {

constexpr size_t kSeedBudgetBytes = size_t(6) << 30;     // seeds of one chunk of rays: N x chunk x rounds x 2560 B (a launch is cut into chunks that fit)
constexpr int kFrontBlocks = 512;                        // x 4 waves = two waves on each of the 1024 SIMDs
constexpr int kMlpAdjointBlocks = 4096;                  // single-wave workgroups, dynamic item fetch (two per SIMD are resident)
constexpr int kPairBlocks = 1024;                        // residual_step_pair_kernel: x 2 waves = two waves on each of the 1024 SIMDs
constexpr int kPairMaxRays = 2048;                       // two-round launches of at most this many rays split each ray over two waves

int pair_max_rays() { return switches().pair_max_rays; }    // experiment switch: 0 turns the pair kernel off

struct ResidualStepPlan {
    int rounds, chunk, front_blocks, front_waves, slots_per_item, items_per_instance;
    bool pair;                     // few rays: residual_step_pair_kernel (a ray split over the two waves of a workgroup)
    long long slots_per_instance;
    size_t front_lds;
    // workspace layout, in floats from the base
    size_t box_partials, loss_partials, jets, box_extra, counter, segment_sums, seeds, item_rows, masks, item_flags, images, total_bytes;
};

static bool plan_residual_step(int N, int S, int num_rays, bool allow_pair, ResidualStepPlan* p, int slots_per_item = 0) {
    p->rounds = rounds_for(2 * S - 1);
    if (p->rounds < 1 || p->rounds > 4) return false;
    // four rounds (S in (64, 128]): the one-wave-per-ray kernel needs 308 registers (one wave per SIMD), the pair kernel 256 -- the pair
    // kernel wins at every launch size (132 k rays, S = 100: 90.3 against 97.9 ms); two rounds: only while a launch cannot fill the SIMDs
    p->pair = allow_pair && num_rays >= 1 && (p->rounds == 4 || (p->rounds == 2 && num_rays <= pair_max_rays()));
    p->front_lds = p->pair ? static_cast<size_t>(residual_pair_lds_floats(S, N)) * sizeof(float)
                           : static_cast<size_t>(residual_front_lds_floats(S, N)) * sizeof(float) * kMaxWavesPerBlock;
    if (p->front_lds > kLdsLimit / 2) return false;     // two workgroups per CU = two waves per SIMD (one wave per ray at N = 64, S = 128: 93 KB -> the single-kernel form)
    const size_t per_ray = static_cast<size_t>(N) * p->rounds * kSeedFloats * kWave * sizeof(float);
    long long chunk = static_cast<long long>(kSeedBudgetBytes / per_ray);
    if (chunk < 256) chunk = 256;
    if (chunk > num_rays) chunk = num_rays < 1 ? 1 : num_rays;
    p->chunk = static_cast<int>(chunk);
    const long long want = (chunk + kMaxWavesPerBlock - 1) / kMaxWavesPerBlock;
    p->front_blocks = static_cast<int>(want > kFrontBlocks ? kFrontBlocks : want);
    p->front_waves = p->front_blocks * kMaxWavesPerBlock;
    if (p->pair) {
        p->front_blocks = static_cast<int>(chunk > kPairBlocks ? kPairBlocks : chunk);
        p->front_waves = p->front_blocks * kPairWaves;
    }
    p->slots_per_instance = chunk * p->rounds;
    // items: enough of them to balance ~2000 waves, few enough that an item amortises its partial row (6.6 KB written and read once)
    long long per_item = (p->slots_per_instance * N) / 16384;
    // (round 6: launches that leave >= 32 k items even at the largest item -- config 3's chunks of 110 k rays -- take 64 slots per item: same
    //  speed as 32 or 48 (3.66-3.70 Mrays/s, tools/gpu_r06q.sh), half the partial rows written and reduced: 35 -> 17 GB per config-3 step)
    per_item = per_item < 4 ? 4 : (per_item >= 128 ? 64 : (per_item > 32 ? 32 : per_item));
    const int forced = switches().slots_per_item;                        // experiment switch
    if (forced >= 1 && forced <= 64) per_item = forced;
    if (slots_per_item >= 1 && slots_per_item <= 64) per_item = slots_per_item;      // vsrd_render_config::adjoint_slots_per_item
    p->slots_per_item = static_cast<int>(per_item);
    p->items_per_instance = static_cast<int>((p->slots_per_instance + per_item - 1) / per_item);
    size_t at = 0;
    auto take = [&](size_t floats) { const size_t here = at; at += (floats + 3) & ~size_t(3); return here; };
    p->box_partials = take(static_cast<size_t>(p->front_waves) * N * kGradStride);
    p->loss_partials = take(static_cast<size_t>(p->front_waves) * 2);
    p->jets = take(static_cast<size_t>(p->front_waves) * jet_wave_float4s(p->rounds, N) * 4);
    p->box_extra = take(static_cast<size_t>(N) * kGradStride);
    p->segment_sums = take(static_cast<size_t>(N) * kItemSegments * kItemRowFloats);
    p->seeds = take(static_cast<size_t>(N) * p->slots_per_instance * kSeedFloats * kWave);
    p->item_rows = take(static_cast<size_t>(N) * p->items_per_instance * kItemRowFloats);
    p->masks = take((static_cast<size_t>(N) * p->slots_per_instance + 3) / 4);
    p->counter = take(4);                                // right behind the masks: one memset clears both
    p->item_flags = take((static_cast<size_t>(N) * p->items_per_instance + 3) / 4);
    p->images = take(static_cast<size_t>(N) * kMlpImageWords);       // VSRD_FLAG_MLP_SPLIT_BF16: the instances' operand images (residual.h)
    p->total_bytes = at * sizeof(float);
    return true;
}

// vsrd_render_backward on residual fields in the split form.  The plan is the residual step's, with the chunk of rays sized to what the
// caller's workspace holds (at least min(num_rays, 64) rays, else the caller falls back to render_backward_kernel<K, true>).
bool plan_backward_split(int N, int num_distances, int num_rays, size_t budget_bytes, ResidualStepPlan* p) {
    p->rounds = rounds_for(num_distances - 1);
    if (p->rounds < 1 || p->rounds > 4) return false;
    p->pair = false;
    p->front_lds = static_cast<size_t>(backward_front_lds_floats(num_distances, N)) * sizeof(float) * kMaxWavesPerBlock;
    if (p->front_lds > kLdsDefault) return false;
    const size_t per_ray = static_cast<size_t>(N) * p->rounds * kSeedFloats * kWave * sizeof(float);
    long long chunk = static_cast<long long>(kSeedBudgetBytes / per_ray);
    if (chunk > num_rays) chunk = num_rays < 1 ? 1 : num_rays;
    for (;; chunk = chunk * 3 / 4) {                                       // the largest chunk whose whole layout fits the budget
        if (chunk < (num_rays < 64 ? num_rays : 64)) return false;
        p->chunk = static_cast<int>(chunk);
        const long long want = (chunk + kMaxWavesPerBlock - 1) / kMaxWavesPerBlock;
        p->front_blocks = static_cast<int>(want > kFrontBlocks ? kFrontBlocks : want);
        p->front_waves = p->front_blocks * kMaxWavesPerBlock;
        p->slots_per_instance = chunk * p->rounds;
        long long per_item = (p->slots_per_instance * N) / 16384;
        per_item = per_item < 4 ? 4 : (per_item > 32 ? 32 : per_item);
        p->slots_per_item = static_cast<int>(per_item);
        p->items_per_instance = static_cast<int>((p->slots_per_instance + per_item - 1) / per_item);
        size_t at = 0;
        auto take = [&](size_t floats) { const size_t here = at; at += (floats + 3) & ~size_t(3); return here; };
        p->box_partials = take(static_cast<size_t>(p->front_waves) * N * kGradStride);
        p->loss_partials = take(4);
        p->jets = take(static_cast<size_t>(p->front_waves) * jet_wave_float4s(p->rounds, N) * 4);
        p->box_extra = take(static_cast<size_t>(N) * kGradStride);
        p->segment_sums = take(static_cast<size_t>(N) * kItemSegments * kItemRowFloats);
        p->seeds = take(static_cast<size_t>(N) * p->slots_per_instance * kSeedFloats * kWave);
        p->item_rows = take(static_cast<size_t>(N) * p->items_per_instance * kItemRowFloats);
        p->masks = take((static_cast<size_t>(N) * p->slots_per_instance + 3) / 4);
        p->counter = take(4);
        p->item_flags = take((static_cast<size_t>(N) * p->items_per_instance + 3) / 4);
        p->total_bytes = at * sizeof(float);
        if (p->total_bytes <= budget_bytes) return true;
        if (chunk <= 1) return false;
    }
}

int32_t render_backward_split(const vsrd_field* field, const vsrd_render_config* config, const float* origins, const float* directions,
                              const float* distances, int32_t num_distances, const float* grad_labels, const float* grad_gradients,
                              const float* grad_weights, void* workspace, size_t workspace_bytes, float* grad_instances, float* grad_mlp_weights,
                              void* stream, bool* taken) {
    const int N = field->num_instances;
    ResidualStepPlan p;
    *taken = plan_backward_split(N, num_distances, config->num_rays, workspace_bytes, &p);
    if (!*taken) return VSRD_OK;
    const hipStream_t s = static_cast<hipStream_t>(stream);
    const int row = N * kGradStride;
    const FieldArgs f = field_args(field);
    RenderArgs c = render_args(config);
    c.sh.inv_t = f.inv_t;
    float* base = static_cast<float*>(workspace);
    float* box_partials = base + p.box_partials;
    float4* jets = reinterpret_cast<float4*>(base + p.jets);
    float* box_extra = base + p.box_extra;
    unsigned* counter = reinterpret_cast<unsigned*>(base + p.counter);
    float* segment_sums = base + p.segment_sums;
    float* seeds = base + p.seeds;
    float* item_rows = base + p.item_rows;
    unsigned char* masks = reinterpret_cast<unsigned char*>(base + p.masks);
    unsigned char* item_flags = reinterpret_cast<unsigned char*>(base + p.item_flags);
    const unsigned mlp_bits = (config->flags & VSRD_FLAG_MLP_WEIGHTS_CENTRED) ? kMlpCentredBit : 0u;
    const size_t adjoint_lds = (static_cast<size_t>(kMlpWbarFloats) + static_cast<size_t>(kMlpStashTiles) * kTileFloats) * sizeof(float);
    int chunk_index = 0;
    for (int first = 0; first < config->num_rays; first += p.chunk, ++chunk_index) {
        const int rays = std::min(p.chunk, config->num_rays - first);
        const long long used_slots = static_cast<long long>(rays) * p.rounds;
        if (hipMemsetAsync(masks, 0, (p.counter - p.masks + 4) * sizeof(float), s) != hipSuccess) return VSRD_E_LAUNCH;   // tile masks + the item counter's 16-byte slot (a multiple of 16 bytes: one fill kernel, not two)
#define VSRD_LAUNCH(K)                                                                                                                   \
        hipLaunchKernelGGL(render_backward_front_kernel<K>, dim3(p.front_blocks), dim3(kBlockThreads), p.front_lds, s, f, field->instances, \
                           field->mlp_weights, c, origins, directions, distances, num_distances, grad_labels, grad_gradients, grad_weights,  \
                           box_partials, jets, seeds, masks, p.slots_per_instance, first, rays, chunk_index > 0 ? 1 : 0)
        switch (p.rounds) {
            case 1: VSRD_LAUNCH(1); break;
            case 2: VSRD_LAUNCH(2); break;
            case 4: VSRD_LAUNCH(4); break;
            default: return VSRD_E_UNSUPPORTED;
        }
#undef VSRD_LAUNCH
        hipLaunchKernelGGL(residual_mlp_adjoint_kernel, dim3(kMlpAdjointBlocks), dim3(kWave), adjoint_lds, s, field->instances,
                           field->mlp_weights, N, mlp_bits, seeds, masks, p.slots_per_instance, used_slots, p.items_per_instance, p.slots_per_item,
                           counter, item_rows, item_flags, 0ll);
        hipLaunchKernelGGL(reduce_item_rows_kernel, dim3(N, (kItemRowFloats + 255) / 256, kItemSegments), dim3(256), 0, s, item_rows, item_flags,
                           p.items_per_instance, segment_sums, 0ll);
        hipLaunchKernelGGL(reduce_item_segments_kernel, dim3(N, (kItemRowFloats + 255) / 256), dim3(256), 0, s, segment_sums, grad_mlp_weights, box_extra,
                           chunk_index > 0 ? 1 : 0, 0ll);
        if (launch_status() != VSRD_OK) return VSRD_E_LAUNCH;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(row), dim3(256), 0, s, box_partials, p.front_waves, row, grad_instances, box_extra);
    return launch_status();
}

}  // namespace

size_t vsrd_render_backward_workspace_bytes(int32_t num_instances, int32_t residual, int32_t num_distances, int32_t num_rays) {
    if (num_instances < 1 || num_instances > VSRD_MAX_INSTANCES || num_distances < 2 || num_distances > 2 * VSRD_MAX_SAMPLES || num_rays < 0) return 0;
    size_t need = vsrd_workspace_bytes(num_instances, residual);
    ResidualStepPlan p;
    if (residual && num_rays > 0 && plan_backward_split(num_instances, num_distances, num_rays, ~size_t(0), &p) && p.total_bytes > need) need = p.total_bytes;
    return need;
}

size_t vsrd_residual_step_workspace_bytes(int32_t num_instances, int32_t num_samples, int32_t num_rays) {
    if (num_instances < 1 || num_instances > VSRD_MAX_INSTANCES || num_samples < 2 || num_samples > VSRD_MAX_SAMPLES || num_rays < 0) return 0;
    ResidualStepPlan p;
    size_t need = vsrd_workspace_bytes(num_instances, 1);                 // (VSRD_FLAG_RESIDUAL_SINGLE_KERNEL and shapes the split form does not take)
    for (int allow_pair = 0; allow_pair < 2; ++allow_pair)                // (either front kernel: VSRD_FLAG_RESIDUAL_WAVE_PER_RAY)
        if (plan_residual_step(num_instances, num_samples, num_rays, allow_pair != 0, &p) && p.total_bytes > need) need = p.total_bytes;
    return need;
}

int32_t vsrd_render_residual_step(const vsrd_field* field, const vsrd_render_config* config,
                                  const float* origins, const float* directions, const float* u_coarse, const float* u_fine,
                                  const float* targets, const float* instance_weights, float loss_scale, float eikonal_ratio,
                                  void* workspace, size_t workspace_bytes,
                                  float* losses, float* grad_instances, float* grad_mlp_weights, float* labels, void* stream) {
    if (!valid_field(field) || !valid_config(config, true, true) || !workspace || !losses || !grad_instances || !grad_mlp_weights) return VSRD_E_INVALID_ARGUMENT;
    if (field->mlp_weights == nullptr) return VSRD_E_INVALID_ARGUMENT;       // box-only fields: vsrd_render_silhouette_step
    const int N = field->num_instances, S = config->num_samples;
    ResidualStepPlan p;
    // (a step that writes its samples out -- vsrd_render_config::out_* -- keeps one ray per wave: the split-ray and the one-kernel forms have no such outputs)
    const bool allow_pair = !(config->flags & VSRD_FLAG_RESIDUAL_WAVE_PER_RAY) && !wants_samples(config);
    if (wants_samples(config) && (config->flags & VSRD_FLAG_RESIDUAL_SINGLE_KERNEL)) return VSRD_E_UNSUPPORTED;
    Frames frames;
    frames_of(config->num_frames, config->frame_stride, &frames);           // (validated by valid_config)
    if (config->adjoint_slots_per_item < 0 || config->adjoint_slots_per_item > 64) return VSRD_E_INVALID_ARGUMENT;
    const bool one_kernel = (config->flags & VSRD_FLAG_RESIDUAL_SINGLE_KERNEL) || config->num_rays == 0 ||
                            !plan_residual_step(N, S, config->num_rays, allow_pair, &p, config->adjoint_slots_per_item);
    // a batch of frames: the two-kernel form with every frame's rays in ONE chunk (each frame its own seeds, item counter and rows, frame_stride apart)
    if (frames.count > 1 && (one_kernel || p.chunk < config->num_rays || static_cast<long long>(p.total_bytes) > frames.stride)) return VSRD_E_UNSUPPORTED;
    if (one_kernel && wants_samples(config)) return VSRD_E_UNSUPPORTED;
    if (one_kernel)
        return residual_step_single_kernel(field, config, origins, directions, u_coarse, u_fine, targets, instance_weights, loss_scale, eikonal_ratio,
                                           workspace, workspace_bytes, losses, grad_instances, grad_mlp_weights, labels, stream);
    if (workspace_bytes < p.total_bytes) return VSRD_E_WORKSPACE;
    if (!origins || !directions || !targets) return VSRD_E_INVALID_ARGUMENT;
    const hipStream_t s = static_cast<hipStream_t>(stream);
    const int row = N * kGradStride, mlp_row = N * kMlpWeights;
    const int num_points = 2 * S - 1;
    const FieldArgs f = field_args(field);
    RenderArgs c = render_args(config);
    c.sh.inv_t = f.inv_t;
    float* base = static_cast<float*>(workspace);
    float* box_partials = base + p.box_partials;
    float* loss_partials = base + p.loss_partials;
    float4* jets = reinterpret_cast<float4*>(base + p.jets);
    float* box_extra = base + p.box_extra;
    unsigned* counter = reinterpret_cast<unsigned*>(base + p.counter);
    float* segment_sums = base + p.segment_sums;
    float* seeds = base + p.seeds;
    float* item_rows = base + p.item_rows;
    unsigned char* masks = reinterpret_cast<unsigned char*>(base + p.masks);
    unsigned char* item_flags = reinterpret_cast<unsigned char*>(base + p.item_flags);
    const float eikonal_norm = 1.0f / (static_cast<float>(config->num_rays) * static_cast<float>(num_points));
    const float eikonal_scale = eikonal_ratio * eikonal_norm;
    const unsigned mlp_bits = (config->flags & VSRD_FLAG_MLP_WEIGHTS_CENTRED) ? kMlpCentredBit : 0u;
    const size_t adjoint_lds = (static_cast<size_t>(kMlpWbarFloats) + static_cast<size_t>(kMlpStashTiles) * kTileFloats) * sizeof(float);
    // VSRD_FLAG_MLP_SPLIT_BF16: the front kernels of the OTHER translation unit (split_front.hip), which read the instances' split-bf16
    // operand images instead of their weights
    const bool split = (config->flags & VSRD_FLAG_MLP_SPLIT_BF16) != 0u;
    const float* front_weights = field->mlp_weights;
    if (split) {
        unsigned* images = reinterpret_cast<unsigned*>(base + p.images);
        if (vsrd_split_front::pack_images(field->mlp_weights, N, (config->flags & VSRD_FLAG_MLP_WEIGHTS_CENTRED) ? 1 : 0, images, frames.count, frames.stride, s) != vsrd_split_front::kOk)
            return VSRD_E_LAUNCH;
        front_weights = reinterpret_cast<const float*>(images);
    }
    int chunk_index = 0;
    for (int first = 0; first < config->num_rays; first += p.chunk, ++chunk_index) {
        const int rays = std::min(p.chunk, config->num_rays - first);
        const long long used_slots = static_cast<long long>(rays) * p.rounds;
        if (!clear_frames(masks, (p.counter - p.masks + 4) * sizeof(float), frames, s)) return VSRD_E_LAUNCH;   // tile masks + the item counter's 16-byte slot (a multiple of 16 bytes: one fill kernel, not two)
#define VSRD_FRONT_ARGS                                                                                                                  \
        f, field->instances, front_weights, c, origins, directions, u_coarse, u_fine, targets, instance_weights, loss_scale, eikonal_scale, \
        eikonal_norm, labels, box_partials, jets, loss_partials, seeds, masks, p.slots_per_instance, first, rays, chunk_index > 0 ? 1 : 0
#define VSRD_LAUNCH(K)                                                                                                                   \
        if (opt_in_lds(residual_step_front_kernel<K>, p.front_lds) != VSRD_OK) return VSRD_E_LAUNCH;                                     \
        if (frames.count > 1) return VSRD_E_UNSUPPORTED;      /* (a batch's frames are launches of <= 2048 rays: the pair kernel) */                    \
        if (wants_samples(config)) return VSRD_E_UNSUPPORTED; /* (the samples are written by the two-round instantiation below) */                      \
        hipLaunchKernelGGL(residual_step_front_kernel<K>, dim3(p.front_blocks), dim3(kBlockThreads), p.front_lds, s, VSRD_FRONT_ARGS)
#define VSRD_LAUNCH_PAIR(K)                                                                                                              \
        if (frames.count > 1) {                                                                                                          \
            if (opt_in_lds(residual_step_pair_kernel<K, true>, p.front_lds) != VSRD_OK) return VSRD_E_LAUNCH;                            \
            hipLaunchKernelGGL((residual_step_pair_kernel<K, true>), dim3(p.front_blocks, frames.count), dim3(kPairWaves * kWave), p.front_lds, s, VSRD_FRONT_ARGS); \
        } else {                                                                                                                         \
            if (opt_in_lds(residual_step_pair_kernel<K>, p.front_lds) != VSRD_OK) return VSRD_E_LAUNCH;                                  \
            hipLaunchKernelGGL(residual_step_pair_kernel<K>, dim3(p.front_blocks), dim3(kPairWaves * kWave), p.front_lds, s, VSRD_FRONT_ARGS); \
        }
        if (split) {
            vsrd_split_front::FrontLaunch launch = {&f, sizeof f, &c, sizeof c, field->instances, front_weights, origins, directions, u_coarse, u_fine, targets, instance_weights,
                                                    loss_scale, eikonal_scale, eikonal_norm, labels, box_partials, jets, loss_partials, seeds, masks, p.slots_per_instance,
                                                    first, rays, chunk_index > 0 ? 1 : 0, p.pair, p.rounds, p.front_blocks, p.front_lds, wants_samples(config), frames.count};
            const int code = vsrd_split_front::launch_front(launch, s);
            if (code != vsrd_split_front::kOk) return code == vsrd_split_front::kUnsupported ? VSRD_E_UNSUPPORTED : VSRD_E_LAUNCH;
        } else if (p.pair) {
            switch (p.rounds) {
                case 2: VSRD_LAUNCH_PAIR(2); break;
                case 4: VSRD_LAUNCH_PAIR(4); break;
                default: return VSRD_E_UNSUPPORTED;
            }
        } else {
            if (p.rounds == 2 && wants_samples(config) && frames.count == 1) {         // vsrd_render_config::out_*: num_samples in (32, 64]
                if (opt_in_lds(residual_step_front_kernel<2, true>, p.front_lds) != VSRD_OK) return VSRD_E_LAUNCH;
                hipLaunchKernelGGL((residual_step_front_kernel<2, true>), dim3(p.front_blocks), dim3(kBlockThreads), p.front_lds, s, VSRD_FRONT_ARGS);
            } else switch (p.rounds) {
                case 1: VSRD_LAUNCH(1); break;
                case 2: VSRD_LAUNCH(2); break;
                case 4: VSRD_LAUNCH(4); break;
                default: return VSRD_E_UNSUPPORTED;
            }
        }
#undef VSRD_LAUNCH
#undef VSRD_LAUNCH_PAIR
#undef VSRD_FRONT_ARGS
        if (split) {
            if (vsrd_split_front::launch_adjoint(kMlpAdjointBlocks, field->instances, front_weights, N, seeds, masks, p.slots_per_instance, used_slots,
                                                 p.items_per_instance, p.slots_per_item, counter, item_rows, item_flags, frames.count, frames.stride, s) != vsrd_split_front::kOk)
                return VSRD_E_LAUNCH;
        } else {
            hipLaunchKernelGGL(residual_mlp_adjoint_kernel, dim3(kMlpAdjointBlocks, frames.count), dim3(kWave), adjoint_lds, s, field->instances,
                               field->mlp_weights, N, mlp_bits, seeds, masks, p.slots_per_instance, used_slots, p.items_per_instance, p.slots_per_item,
                               counter, item_rows, item_flags, frames.stride);
        }
        hipLaunchKernelGGL(reduce_item_rows_kernel, dim3(N, (kItemRowFloats + 255) / 256, kItemSegments * frames.count), dim3(256), 0, s, item_rows, item_flags,
                           p.items_per_instance, segment_sums, frames.stride);
        hipLaunchKernelGGL(reduce_item_segments_kernel, dim3(N, (kItemRowFloats + 255) / 256, frames.count), dim3(256), 0, s, segment_sums, grad_mlp_weights, box_extra,
                           chunk_index > 0 ? 1 : 0, frames.stride);
        if (launch_status() != VSRD_OK) return VSRD_E_LAUNCH;
    }
    (void)mlp_row;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(row + 2, frames.count), dim3(256), 0, s, box_partials, p.front_waves, row, grad_instances, box_extra, loss_partials, 2, losses,
                       frames.stride);
    return launch_status();
}

size_t vsrd_frame_scratch_bytes(int32_t num_views, int32_t num_boxes) {
    if (num_views < 1 || num_views > kFrameMaxViews || num_boxes < 1 || num_boxes > kFrameMaxBoxes) return 0;
    return frame_scratch_floats(num_views, num_boxes) * sizeof(float);
}

namespace {
static bool frame_args(const vsrd_frame_config* c, FrameStepArgs* a) {
    if (!c || c->num_boxes < 1 || c->num_boxes > kFrameMaxBoxes || c->num_views < 1 || c->num_views > kFrameMaxViews || c->num_steps < 1) return false;
    a->num_boxes = c->num_boxes; a->num_views = c->num_views;
    a->height = c->height; a->width = c->width; a->epsilon = c->epsilon;
    for (int j = 0; j < 3; ++j) {
        a->location_lo[j] = c->location_lo[j]; a->location_hi[j] = c->location_hi[j];
        a->dimension_lo[j] = c->dimension_lo[j]; a->dimension_hi[j] = c->dimension_hi[j];
    }
    a->num_steps = c->num_steps;
    a->max_temperature = c->max_temperature; a->min_temperature = c->min_temperature; a->max_std = c->max_std; a->min_std = c->min_std;
    a->weight_iou = c->weight_iou; a->weight_l1 = c->weight_l1; a->weight_silhouette = c->weight_silhouette;
    a->beta1 = c->beta1; a->beta2 = c->beta2; a->adam_epsilon = c->adam_epsilon; a->lr_gamma = c->lr_gamma;
    Frames frames;
    if (!frames_of(c->num_frames, c->frame_stride, &frames)) return false;
    a->frame_stride = frames.stride;
    return true;
}
}  // namespace

int32_t vsrd_frame_prologue(const vsrd_frame_config* config, const float* raw_locations, const float* raw_dimensions,
                            const float* raw_orientations, const float* extrinsics, const float* intrinsics, const float* gt_boxes,
                            const uint8_t* visible, const int64_t* step, void* scratch, size_t scratch_bytes,
                            float* instances, int64_t* pd_indices, int64_t* gt_indices, int32_t* target_columns, float* instance_weights,
                            float* schedule, float* projection_losses, float* grad_raw, void* stream) {
    FrameStepArgs a;
    if (!frame_args(config, &a)) return VSRD_E_INVALID_ARGUMENT;
    if (!raw_locations || !raw_dimensions || !raw_orientations || !extrinsics || !intrinsics || !gt_boxes || !visible || !step || !instances ||
        !pd_indices || !gt_indices || !target_columns || !instance_weights || !schedule || !projection_losses || !grad_raw)
        return VSRD_E_INVALID_ARGUMENT;
    if (!scratch || scratch_bytes < vsrd_frame_scratch_bytes(a.num_views, a.num_boxes)) return VSRD_E_WORKSPACE;
    FrameBuffers b;
    b.raw_locations = raw_locations; b.raw_dimensions = raw_dimensions; b.raw_orientations = raw_orientations;
    b.extrinsics = extrinsics; b.intrinsics = intrinsics; b.gt_boxes = gt_boxes; b.visible = visible;
    b.step = reinterpret_cast<const long long*>(step);
    b.scratch = static_cast<float*>(scratch);
    b.instances = instances;
    b.pd_indices = reinterpret_cast<long long*>(pd_indices); b.gt_indices = reinterpret_cast<long long*>(gt_indices);
    b.target_map = target_columns; b.instance_weights = instance_weights; b.schedule = schedule; b.losses = projection_losses; b.grad_raw = grad_raw;
    hipLaunchKernelGGL(frame_prologue_kernel, dim3(1, config->num_frames > 1 ? config->num_frames : 1), dim3(kFrameThreads), 0, static_cast<hipStream_t>(stream), a, b);
    return launch_status();
}

int32_t vsrd_frame_prologue_sample(const vsrd_frame_config* config, const float* raw_locations, const float* raw_dimensions,
                                   const float* raw_orientations, const float* extrinsics, const float* intrinsics, const float* gt_boxes,
                                   const uint8_t* visible, const int64_t* step, void* scratch, size_t scratch_bytes,
                                   float* instances, int64_t* pd_indices, int64_t* gt_indices, int32_t* target_columns, float* instance_weights,
                                   float* schedule, float* projection_losses, float* grad_raw,
                                   void* ray_table, int64_t count, int32_t num_rays, uint64_t seed, const int64_t* remap, int64_t* ray_indices, void* stream) {
    FrameStepArgs a;
    if (!frame_args(config, &a)) return VSRD_E_INVALID_ARGUMENT;
    if (!raw_locations || !raw_dimensions || !raw_orientations || !extrinsics || !intrinsics || !gt_boxes || !visible || !step || !instances ||
        !pd_indices || !gt_indices || !target_columns || !instance_weights || !schedule || !projection_losses || !grad_raw)
        return VSRD_E_INVALID_ARGUMENT;
    if (!ray_table || !ray_indices || count < 1 || count >= 0xffffffffll || num_rays < 1 || num_rays > kSampleMax) return VSRD_E_INVALID_ARGUMENT;
    if (!scratch || scratch_bytes < vsrd_frame_scratch_bytes(a.num_views, a.num_boxes)) return VSRD_E_WORKSPACE;
    FrameBuffers b;
    b.raw_locations = raw_locations; b.raw_dimensions = raw_dimensions; b.raw_orientations = raw_orientations;
    b.extrinsics = extrinsics; b.intrinsics = intrinsics; b.gt_boxes = gt_boxes; b.visible = visible;
    b.step = reinterpret_cast<const long long*>(step);
    b.scratch = static_cast<float*>(scratch);
    b.instances = instances;
    b.pd_indices = reinterpret_cast<long long*>(pd_indices); b.gt_indices = reinterpret_cast<long long*>(gt_indices);
    b.target_map = target_columns; b.instance_weights = instance_weights; b.schedule = schedule; b.losses = projection_losses; b.grad_raw = grad_raw;
    // (a refused LDS opt-in is "this device / build cannot run the combined launch" -- VSRD_E_UNSUPPORTED, on which the caller falls back to
    //  vsrd_frame_prologue + vsrd_sample_rays_table -- and not a failed launch, which must not be mistaken for it)
    if (opt_in_lds(frame_prologue_sample_kernel, kTableLdsBytes) != VSRD_OK) return VSRD_E_UNSUPPORTED;
    hipLaunchKernelGGL(frame_prologue_sample_kernel, dim3(2, config->num_frames > 1 ? config->num_frames : 1), dim3(kTableThreads), kTableLdsBytes, static_cast<hipStream_t>(stream), a, b,
                       static_cast<RayTableHeader*>(ray_table), static_cast<long long>(count), num_rays, seed, reinterpret_cast<const unsigned long long*>(step),
                       reinterpret_cast<const long long*>(remap), reinterpret_cast<long long*>(ray_indices));
    return launch_status();
}

int32_t vsrd_frame_epilogue(const vsrd_frame_config* config, const float* grad_instances, const float* grad_raw_projection,
                            const float* projection_losses, const float* render_losses, float eikonal_ratio,
                            const vsrd_adam_tensors* locations, const vsrd_adam_tensors* dimensions, const vsrd_adam_tensors* orientations,
                            float* other_learning_rate_0, float* other_learning_rate_1, int64_t* step,
                            float* record, float* raw_gradients, void* stream) {
    FrameStepArgs a;
    if (!frame_args(config, &a)) return VSRD_E_INVALID_ARGUMENT;
    if (!grad_instances || !grad_raw_projection || !projection_losses || !render_losses || !locations || !dimensions || !orientations || !step || !record)
        return VSRD_E_INVALID_ARGUMENT;
    const vsrd_adam_tensors* groups[3] = {locations, dimensions, orientations};
    AdamTensors tensors[3];
    for (int k = 0; k < 3; ++k) {
        if (!groups[k]->parameter || !groups[k]->exp_avg || !groups[k]->exp_avg_sq || !groups[k]->step || !groups[k]->learning_rate) return VSRD_E_INVALID_ARGUMENT;
        tensors[k] = AdamTensors{groups[k]->parameter, groups[k]->exp_avg, groups[k]->exp_avg_sq, groups[k]->step, groups[k]->learning_rate};
    }
    EpilogueBuffers e;
    e.grad_instances = grad_instances; e.grad_raw_projection = grad_raw_projection; e.projection_losses = projection_losses;
    e.render_losses = render_losses; e.eikonal_ratio = eikonal_ratio;
    e.locations = tensors[0]; e.dimensions = tensors[1]; e.orientations = tensors[2];
    e.other_learning_rates[0] = other_learning_rate_0; e.other_learning_rates[1] = other_learning_rate_1;
    e.step = reinterpret_cast<long long*>(step); e.record = record; e.raw_gradients = raw_gradients;
    hipLaunchKernelGGL(frame_epilogue_kernel, dim3(1, config->num_frames > 1 ? config->num_frames : 1), dim3(kFrameMaxBoxes), 0, static_cast<hipStream_t>(stream), a, e);
    return launch_status();
}

namespace {
struct HyperPlan {
    size_t activations;        // float offset of z[l] = activations + l * N * 256, l < 4
    size_t inv_norm;           // 4 x 256 + 1632
    size_t gz;                 // 5 x N x 256: the output adjoints of the four hidden linears, then the embeddings' gradient
    size_t partials;           // final_blocks x N x 256
    size_t norm_partials;      // 4 x N x 2 x 256
    size_t total;
    int final_blocks;
};
static HyperPlan plan_hypernetwork(int num_instances) {
    const size_t slab = static_cast<size_t>(num_instances) * kHyperWidth;
    HyperPlan p;
    p.final_blocks = (kMlpWeights + kHyperWaves - 1) / kHyperWaves;
    p.activations = 0;
    p.inv_norm = p.activations + (VSRD_HYPER_LAYERS - 1) * slab;
    p.gz = p.inv_norm + (VSRD_HYPER_LAYERS - 1) * kHyperWidth + kMlpWbarFloats;
    p.partials = p.gz + (VSRD_HYPER_LAYERS) * slab;
    p.norm_partials = p.partials + static_cast<size_t>(p.final_blocks) * slab;
    p.total = p.norm_partials + (VSRD_HYPER_LAYERS - 1) * 2 * slab;
    return p;
}
static bool valid_adam(const vsrd_adam_tensors& t) { return t.parameter && t.exp_avg && t.exp_avg_sq && t.step && t.learning_rate; }
static AdamTensors adam_tensors(const vsrd_adam_tensors& t) { return AdamTensors{t.parameter, t.exp_avg, t.exp_avg_sq, t.step, t.learning_rate}; }
static bool valid_hypernetwork(const vsrd_hypernetwork* net) {
    if (!net || net->num_instances < 1 || net->num_instances > VSRD_MAX_INSTANCES || net->num_outputs != kMlpWeights) return false;
    if (!valid_adam(net->embeddings)) return false;
    for (int l = 0; l < VSRD_HYPER_LAYERS; ++l) {
        if (!valid_adam(net->weight_v[l]) || !valid_adam(net->weight_g[l]) || !valid_adam(net->bias[l])) return false;
        if (l + 1 < VSRD_HYPER_LAYERS && (!valid_adam(net->norm_weight[l]) || !valid_adam(net->norm_bias[l]))) return false;
    }
    return true;
}
}  // namespace

size_t vsrd_hypernetwork_workspace_bytes(int32_t num_instances) {
    if (num_instances < 1 || num_instances > VSRD_MAX_INSTANCES) return 0;
    return plan_hypernetwork(num_instances).total * sizeof(float);
}

int32_t vsrd_hypernetwork_forward(const vsrd_hypernetwork* net, void* workspace, size_t workspace_bytes,
                                  float* mlp_weights, float* centred, void* stream) {
    if (!valid_hypernetwork(net) || !mlp_weights) return VSRD_E_INVALID_ARGUMENT;
    Frames frames;
    if (!frames_of(net->num_frames, net->frame_stride, &frames)) return VSRD_E_INVALID_ARGUMENT;
    const int N = net->num_instances;
    const HyperPlan p = plan_hypernetwork(N);
    if (!workspace || workspace_bytes < p.total * sizeof(float)) return VSRD_E_WORKSPACE;
    float* ws = static_cast<float*>(workspace);
    const hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t slab = static_cast<size_t>(N) * kHyperWidth, lds = slab * sizeof(float);
    if (opt_in_lds(hyper_linear_forward_kernel, lds) != VSRD_OK) return VSRD_E_LAUNCH;
    constexpr int kLast = VSRD_HYPER_LAYERS - 1;
    static_assert(kLast == kHyperHidden, "csrc/hypernetwork.h is written for four hidden blocks");
    HyperHiddenForward hidden;
    for (int l = 0; l < kLast; ++l) {
        hidden.v[l] = net->weight_v[l].parameter; hidden.g[l] = net->weight_g[l].parameter; hidden.b[l] = net->bias[l].parameter;
        hidden.z[l] = ws + p.activations + l * slab;
        hidden.inv_norm[l] = ws + p.inv_norm + l * kHyperWidth;
        if (l + 1 < kLast) { hidden.gamma[l] = net->norm_weight[l].parameter; hidden.beta[l] = net->norm_bias[l].parameter; }
    }
    hipLaunchKernelGGL(hyper_hidden_forward_kernel, dim3(N, frames.count), dim3(kHyperChainThreads), 0, s, net->embeddings.parameter, hidden, frames.stride);
    hipLaunchKernelGGL(hyper_linear_forward_kernel, dim3((kMlpWeights + kHyperWaves - 1) / kHyperWaves, frames.count), dim3(kHyperThreads), lds, s,
                       ws + p.activations + (kLast - 1) * slab, net->norm_weight[kLast - 1].parameter, net->norm_bias[kLast - 1].parameter,
                       net->weight_v[kLast].parameter, net->weight_g[kLast].parameter, net->bias[kLast].parameter, kMlpWeights, N, mlp_weights,
                       ws + p.inv_norm + kLast * kHyperWidth, frames.stride);
    if (centred) hipLaunchKernelGGL(hyper_centre_kernel, dim3(N, frames.count), dim3(128), 0, s, mlp_weights, N, centred, frames.stride);
    return launch_status();
}

int32_t vsrd_centre_mlp_weights(const float* mlp_weights, int32_t num_instances, float* centred, void* stream) {
    if (!mlp_weights || !centred || num_instances < 1 || num_instances > VSRD_MAX_INSTANCES) return VSRD_E_INVALID_ARGUMENT;
    hipLaunchKernelGGL(hyper_centre_kernel, dim3(num_instances), dim3(128), 0, static_cast<hipStream_t>(stream), mlp_weights, num_instances, centred);
    return launch_status();
}

int32_t vsrd_hypernetwork_backward_step(const vsrd_hypernetwork* net, void* workspace, size_t workspace_bytes,
                                        const float* grad_mlp_weights, float grad_scale, void* stream) {
    if (!valid_hypernetwork(net) || !grad_mlp_weights) return VSRD_E_INVALID_ARGUMENT;
    Frames frames;
    if (!frames_of(net->num_frames, net->frame_stride, &frames)) return VSRD_E_INVALID_ARGUMENT;
    const int N = net->num_instances;
    const HyperPlan p = plan_hypernetwork(N);
    if (!workspace || workspace_bytes < p.total * sizeof(float)) return VSRD_E_WORKSPACE;
    float* ws = static_cast<float*>(workspace);
    const hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t slab = static_cast<size_t>(N) * kHyperWidth, lds = (slab + kHyperWaves * kHyperWidth + static_cast<size_t>(N) * kHyperWaves) * sizeof(float);
    if (opt_in_lds(hyper_linear_backward_kernel, lds) != VSRD_OK || opt_in_lds(hyper_hidden_update_kernel, lds) != VSRD_OK) return VSRD_E_LAUNCH;
    constexpr int kLast = VSRD_HYPER_LAYERS - 1;
    const HyperAdam adam{net->beta1, net->beta2, net->adam_epsilon};
    auto z = [&](int l) { return ws + p.activations + l * slab; };              // output of linear l (l < 4)
    auto gz = [&](int l) { return ws + p.gz + l * slab; };                       // its adjoint; gz(4): the embeddings' gradient
    auto inv_norm = [&](int l) { return ws + p.inv_norm + l * kHyperWidth; };
    auto norm_shares = [&](int k) { return ws + p.norm_partials + k * 2 * slab; };
    // the final linear: its own update and its workgroups' shares of the input adjoint; their sum through the norm behind linear 3
    const int final_blocks = (kMlpWeights + kHyperWaves - 1) / kHyperWaves;
    hipLaunchKernelGGL(hyper_linear_backward_kernel, dim3(final_blocks, frames.count), dim3(kHyperThreads), lds, s, z(kLast - 1), net->norm_weight[kLast - 1].parameter,
                       net->norm_bias[kLast - 1].parameter, grad_mlp_weights, grad_scale, inv_norm(kLast), kMlpWeights, N, adam_tensors(net->weight_v[kLast]),
                       adam_tensors(net->weight_g[kLast]), adam_tensors(net->bias[kLast]), adam, ws + p.partials, frames.stride);
    hipLaunchKernelGGL(hyper_norm_backward_kernel, dim3(N, frames.count), dim3(kHyperNormThreads * kHyperNormSplit), 0, s, ws + p.partials, final_blocks, z(kLast - 1), N,
                       net->norm_weight[kLast - 1].parameter, net->norm_bias[kLast - 1].parameter, gz(kLast - 1), norm_shares(kLast - 1), frames.stride);
    // the chain through the hidden blocks (one workgroup per instance), then every hidden linear's own update
    HyperHiddenBackward chain;
    HyperHiddenUpdate update;
    HyperNorms norms;
    HyperStepCounters counters;
    counters.count = 0;
    counters.step[counters.count++] = net->embeddings.step;
    for (int l = 0; l < VSRD_HYPER_LAYERS; ++l) {
        counters.step[counters.count++] = net->weight_v[l].step;
        counters.step[counters.count++] = net->weight_g[l].step;
        counters.step[counters.count++] = net->bias[l].step;
        if (l == kLast) break;
        counters.step[counters.count++] = net->norm_weight[l].step;
        counters.step[counters.count++] = net->norm_bias[l].step;
        norms.gamma[l] = adam_tensors(net->norm_weight[l]);
        norms.beta[l] = adam_tensors(net->norm_bias[l]);
        chain.v[l] = net->weight_v[l].parameter; chain.g[l] = net->weight_g[l].parameter; chain.inv_norm[l] = inv_norm(l);
        chain.gz[l] = gz(l);
        if (l + 1 < kLast) { chain.gamma[l] = net->norm_weight[l].parameter; chain.beta[l] = net->norm_bias[l].parameter; chain.z[l] = z(l); chain.norm_partials[l] = norm_shares(l); }
        update.x[l] = l == 0 ? net->embeddings.parameter : z(l - 1);
        update.gamma[l] = l == 0 ? nullptr : net->norm_weight[l - 1].parameter;
        update.beta[l] = l == 0 ? nullptr : net->norm_bias[l - 1].parameter;
        update.gz[l] = gz(l); update.inv_norm[l] = inv_norm(l);
        update.v[l] = adam_tensors(net->weight_v[l]); update.g[l] = adam_tensors(net->weight_g[l]); update.b[l] = adam_tensors(net->bias[l]);
    }
    chain.embedding_bar = gz(kLast);
    hipLaunchKernelGGL(hyper_hidden_backward_kernel, dim3(N, frames.count), dim3(kHyperChainThreads), 0, s, chain, frames.stride);
    hipLaunchKernelGGL(hyper_hidden_update_kernel, dim3(kHyperHidden * (kHyperWidth / kHyperWaves), frames.count), dim3(kHyperThreads), lds, s, update, N, adam, frames.stride);
    hipLaunchKernelGGL(hyper_finish_kernel, dim3(1, frames.count), dim3(kHyperChainThreads), 0, s, norms, ws + p.norm_partials, adam_tensors(net->embeddings), gz(kLast), N, adam,
                       counters, net->embeddings.learning_rate, net->weight_v[0].learning_rate, net->lr_gamma, frames.stride);
    return launch_status();
}

size_t vsrd_sample_rays_workspace_bytes(void) { return sizeof(SampleScratch); }

int32_t vsrd_sample_rays(const float* weights, int64_t count, int32_t num_samples, uint64_t seed, uint64_t stream_offset,
                         const uint64_t* device_stream_offset, void* workspace, size_t workspace_bytes, int64_t* indices, void* stream) {
    if (!weights || !workspace || !indices || count < 1 || num_samples < 1 || num_samples > kSampleMax) return VSRD_E_INVALID_ARGUMENT;
    if (workspace_bytes < sizeof(SampleScratch)) return VSRD_E_WORKSPACE;
    const hipStream_t s = static_cast<hipStream_t>(stream);
    SampleScratch* scratch = static_cast<SampleScratch*>(workspace);
    const unsigned long long* device_step = reinterpret_cast<const unsigned long long*>(device_stream_offset);
    const long long want = (count + 255) / 256;
    const int blocks = static_cast<int>(want > 4096 ? 4096 : want);
    hipLaunchKernelGGL(sample_clear_kernel, dim3(1), dim3(256), 0, s, scratch);
    // every workgroup ends with one global atomic per occupied bin (~100): few, fat workgroups keep that tail short
    hipLaunchKernelGGL(keys_histogram_kernel, dim3(blocks > 512 ? 512 : blocks), dim3(256), 0, s, weights, static_cast<long long>(count), seed, stream_offset, device_step, scratch);
    hipLaunchKernelGGL(threshold_kernel, dim3(1), dim3(kWave), 0, s, scratch, num_samples);
    hipLaunchKernelGGL(collect_kernel, dim3(blocks), dim3(256), 0, s, weights, static_cast<long long>(count), seed, stream_offset, device_step, scratch);
    hipLaunchKernelGGL(select_kernel, dim3(1), dim3(1024), 0, s, scratch, num_samples, reinterpret_cast<long long*>(indices));
    return launch_status();
}

size_t vsrd_ray_table_bytes(int64_t count) { return count < 1 ? 0 : ray_table_bytes(count); }

int32_t vsrd_ray_table_build(const float* weights, int64_t count, void* table, size_t table_bytes, void* stream) {
    if (!weights || !table || count < 1 || count >= 0xffffffffll) return VSRD_E_INVALID_ARGUMENT;
    if (table_bytes < ray_table_bytes(count)) return VSRD_E_WORKSPACE;
    const hipStream_t s = static_cast<hipStream_t>(stream);
    RayTableHeader* header = static_cast<RayTableHeader*>(table);
    if (hipMemsetAsync(header, 0, sizeof(RayTableHeader), s) != hipSuccess) return VSRD_E_LAUNCH;
    const long long blocks = ray_table_blocks(count);
    hipLaunchKernelGGL(ray_table_max_kernel, dim3(static_cast<unsigned>(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, s, weights, static_cast<long long>(count), header);
    hipLaunchKernelGGL(ray_table_sums_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, weights, static_cast<long long>(count), header);
    hipLaunchKernelGGL(ray_table_offsets_kernel, dim3(1), dim3(1024), 0, s, static_cast<long long>(count), header);
    hipLaunchKernelGGL(ray_table_fill_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, weights, static_cast<long long>(count), header);
    const long long boundaries = (1ll << ray_table_guide_bits(count)) + 1;
    hipLaunchKernelGGL(ray_table_guide_kernel, dim3(static_cast<unsigned>((boundaries + 255) / 256)), dim3(256), 0, s, static_cast<long long>(count), header);
    return launch_status();
}

int32_t vsrd_sample_rays_table(void* table, int64_t count, int32_t num_samples, uint64_t seed, uint64_t stream_offset,
                               const uint64_t* device_stream_offset, const int64_t* remap, int64_t* indices, void* stream) {
    if (!table || !indices || count < 1 || count >= 0xffffffffll || num_samples < 1 || num_samples > kSampleMax) return VSRD_E_INVALID_ARGUMENT;
    if (opt_in_lds(sample_table_kernel, kTableLdsBytes) != VSRD_OK) return VSRD_E_LAUNCH;
    hipLaunchKernelGGL(sample_table_kernel, dim3(1), dim3(kTableThreads), kTableLdsBytes, static_cast<hipStream_t>(stream), static_cast<RayTableHeader*>(table),
                       static_cast<long long>(count), num_samples, seed, stream_offset, reinterpret_cast<const unsigned long long*>(device_stream_offset),
                       reinterpret_cast<const long long*>(remap), reinterpret_cast<long long*>(indices));
    return launch_status();
}

int32_t vsrd_match_boxes(const float* pd_boxes, const float* gt_boxes, int32_t num_pd, int32_t num_gt,
                         int64_t* pd_indices, int64_t* gt_indices, void* stream) {
    if (!pd_boxes || !gt_boxes || !pd_indices || !gt_indices || num_pd < 1 || num_gt < 1 || num_pd > 64 || num_gt > 64) return VSRD_E_INVALID_ARGUMENT;
    hipLaunchKernelGGL(match_kernel, dim3(1), dim3(kWave), 0, static_cast<hipStream_t>(stream), nullptr, pd_boxes, gt_boxes, num_pd, num_gt,
                       reinterpret_cast<long long*>(pd_indices), reinterpret_cast<long long*>(gt_indices));
    return launch_status();
}

int32_t vsrd_linear_sum_assignment(const float* cost, int32_t num_rows, int32_t num_cols, int64_t* row_indices, int64_t* col_indices, void* stream) {
    if (!cost || !row_indices || !col_indices || num_rows < 1 || num_cols < 1 || num_rows > 64 || num_cols > 64) return VSRD_E_INVALID_ARGUMENT;
    hipLaunchKernelGGL(match_kernel, dim3(1), dim3(kWave), 0, static_cast<hipStream_t>(stream), cost, nullptr, nullptr, num_rows, num_cols,
                       reinterpret_cast<long long*>(row_indices), reinterpret_cast<long long*>(col_indices));
    return launch_status();
}

#ifdef VSRD_PHASE_TIMERS
// Experiments only (tools/phase_timers.py): read (and optionally clear) the per-phase tick totals of the fused step kernels.
int32_t vsrd_debug_phase_cycles(unsigned long long* out16, int32_t reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return VSRD_E_LAUNCH;
    if (reset) {
        unsigned long long zeros[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), zeros, sizeof(zeros)) != hipSuccess) return VSRD_E_LAUNCH;
    }
    return VSRD_OK;
}
#endif

// Not part of the public header: exercised by tests/test_hip_wave.py.
int32_t vsrd_selftest_wave(const float* in64, float* out512, void* stream) {
    if (!in64 || !out512) return VSRD_E_INVALID_ARGUMENT;
    hipLaunchKernelGGL(wave_selftest_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), in64, out512);  // writes 576 floats
    return launch_status();
}

// erf_gelu / erf_gelu_derivative of csrc/hypernetwork.h at n points: out[0..n) values, out[n..2n) derivatives
// (tests/test_hip_step.py::test_hypernetwork_gelu_error_bound).
int32_t vsrd_selftest_gelu(const float* x, int32_t n, float* out, void* stream) {
    if (!x || !out || n < 1) return VSRD_E_INVALID_ARGUMENT;
    hipLaunchKernelGGL(gelu_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, out);
    return launch_status();
}

}  // extern "C"
