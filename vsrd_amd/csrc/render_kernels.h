// __global__ kernels of the render path (forward, fused hierarchical forward, backward).
// Grid: persistent-ish; 256-thread workgroups = 4 independent waves, each looping over rays.
#pragma once
#include "render.h"

namespace vsrd {

constexpr int kMaxWavesPerBlock = 4;
constexpr int kBlockThreads = kMaxWavesPerBlock * kWave;   // launch bound; the host may launch fewer waves

__device__ __forceinline__ int waves_per_block() { return static_cast<int>(blockDim.x) >> 6; }
constexpr int kGradStride = kInstanceStride;   // grad_instances rows are [t(3) R(9) dim(3) pad]
constexpr int kSeedFloats = 7;                 // per sample: local position p (3), d_bar, gl_bar (3); x - t = R p is rebuilt by the reader (round 6; 10 before)
constexpr int kMlpBatch = 8;                   // rays whose MLP adjoints are run together, instance-major (adjoint_phase_mlp)

// residual fields: every wave also owns kMlpWbarFloats floats for residual_forward's staged weight operands (Shading::mlp_lds)
__host__ __device__ constexpr int forward_weights_floats(bool residual) { return residual ? kMlpWbarFloats : 0; }

// Floats of LDS one wave of render_backward_kernel owns (a multiple of 4: the partitions stay 16-byte aligned).
__host__ __device__ constexpr int backward_lds_floats(int num_distances, int num_instances, bool residual) {
    return (forward_weights_floats(residual) + (residual ? kMlpLdsFloats + kMlpBatch * 4 * num_instances : 0) + num_distances + num_instances +
            num_instances * kGradStride + cull_coef_floats(num_instances) + 3) & ~3;
}

// Floats of LDS one wave of render_forward_kernel owns: sorted distances, [N][64] distance cache, culling coefficients.
__host__ __device__ constexpr int forward_lds_floats(int num_distances, int num_instances, bool residual = false) {
    return (forward_weights_floats(residual) + num_distances + num_instances * kWave + cull_coef_floats(num_instances) + 3) & ~3;
}

// ... and of render_hierarchical_kernel.
__host__ __device__ constexpr int hierarchical_lds_floats(int num_samples, int num_instances, bool residual) {
    return (forward_weights_floats(residual) + wave_lds_floats(num_samples, num_instances) + 3) & ~3;
}

// The instance block [N,16] travels as its own `const float* __restrict__` kernel argument (not inside this
// struct): only then does the compiler prove the uniform loads in the instance loops are not clobbered by the
// kernel's stores and select SMEM (s_load_dwordx8/x4) instead of VMEM for them.
struct FieldArgs {
    int num_instances;
    float inv_t;
};

struct RenderArgs {
    int num_rays;
    int num_samples;
    float near, far;
    Shading sh;
    int origin_stride;
    unsigned long long seed, stream_offset;
    unsigned flags;
    const float* dynamic;                         // vsrd_render_config::device_schedule
    const unsigned long long* dynamic_offset;     // vsrd_render_config::device_stream_offset
    const long long* ray_indices;                 // vsrd_render_config::ray_indices (fused step kernels only)
    int rays_per_origin;
    const int* target_columns;
    int target_stride;
    // vsrd_render_config::out_* (the fused box-only step only): the step's own pass-1 weights, uniforms and sorted pass-2 distances
    float* out_distances;
    float* out_coarse_weights;
    float* out_u_coarse;
    float* out_u_fine;
    long long frame_stride;                       // vsrd_render_config::frame_stride (bytes; wave.h: frame batches), 0 for one frame
};

// Frame f of a batch (wave.h): the pointers inside the argument block move with every other pointer of the launch.
__device__ __forceinline__ void shift_frame(RenderArgs& c, long long shift) {
    c.dynamic = of_frame(c.dynamic, shift); c.dynamic_offset = of_frame(c.dynamic_offset, shift);
    c.ray_indices = of_frame(c.ray_indices, shift); c.target_columns = of_frame(c.target_columns, shift);
    c.out_distances = of_frame(c.out_distances, shift); c.out_coarse_weights = of_frame(c.out_coarse_weights, shift);
    c.out_u_coarse = of_frame(c.out_u_coarse, shift); c.out_u_fine = of_frame(c.out_u_fine, shift);
}

// Row of the frame-resident tensors that step ray `ray` reads (vsrd_render_config::ray_indices), as a wave-uniform value.
__device__ __forceinline__ long long source_row(const RenderArgs& c, int ray) {
    return c.ray_indices ? c.ray_indices[ray] : static_cast<long long>(ray);
}

__device__ __forceinline__ Ray load_ray_gathered(const RenderArgs& c, const float* __restrict__ origins, const float* __restrict__ directions, long long row) {
    const long long origin_row = (c.ray_indices && c.rays_per_origin > 0) ? row / c.rays_per_origin : row;
    const float* o = origins + origin_row * c.origin_stride;
    const float* d = directions + row * 3;
    Ray r;
    r.ox = uniform(o[0]); r.oy = uniform(o[1]); r.oz = uniform(o[2]);
    r.rx = uniform(d[0]); r.ry = uniform(d[1]); r.rz = uniform(d[2]);
    return r;
}

// Target of predicted instance `lane` for the ray read from row `row` (vsrd_render_config::target_columns).
__device__ __forceinline__ float load_target(const RenderArgs& c, const float* __restrict__ targets, long long row, int lane, int N) {
    if (lane >= N) return 0.0f;
    if (c.target_columns == nullptr) return targets[row * N + lane];
    const int column = c.target_columns[lane];
    return (column >= 0) ? targets[row * c.target_stride + column] : 0.0f;
}

// Phase clocks (experiments only: -DVSRD_PHASE_TIMERS, tools/phase_timers.py): every wave adds the s_memtime ticks it spends in each
// phase of the fused step kernels to a global table that vsrd_debug_phase_cycles reads back.  Compiled out by default.
#ifdef VSRD_PHASE_TIMERS
__device__ unsigned long long g_phase_cycles[16];
struct PhaseClock {
    unsigned long long last, acc[8];
    __device__ __forceinline__ PhaseClock() { for (int i = 0; i < 8; ++i) acc[i] = 0ull; last = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void mark(int phase) { const unsigned long long now = __builtin_readcyclecounter(); acc[phase] += now - last; last = now; }
    __device__ __forceinline__ void flush(int lane) { if (lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_phase_cycles[i], acc[i]); }
};
#define VSRD_PHASE_CLOCK() PhaseClock phase_clock
#define VSRD_PHASE(i) phase_clock.mark(i)
#define VSRD_PHASE_FLUSH(lane) phase_clock.flush(lane)
#else
#define VSRD_PHASE_CLOCK()
#define VSRD_PHASE(i)
#define VSRD_PHASE_FLUSH(lane)
#endif

// Per-step scalars that live on the device (hipGraph replay): applied to the kernel's private copies of its arguments.
__device__ __forceinline__ void apply_device_schedule(FieldArgs& f, RenderArgs& c) {
    if (c.dynamic != nullptr) {
        const float temperature = c.dynamic[0], std = c.dynamic[1];
        f.inv_t = 1.0f / temperature;
        c.sh.inv_t = f.inv_t;
        c.sh.std = std;
        c.sh.inv_std = 1.0f / std;
        c.sh.ratio = c.dynamic[2];
    }
    if (c.dynamic_offset != nullptr) c.stream_offset = *c.dynamic_offset;
}

__device__ __forceinline__ Ray load_ray(const float* __restrict__ origins, const float* __restrict__ directions,
                                        int origin_stride, int ray) {
    const float* o = origins + static_cast<size_t>(ray) * origin_stride;
    const float* d = directions + static_cast<size_t>(ray) * 3;
    Ray r;
    r.ox = uniform(o[0]); r.oy = uniform(o[1]); r.oz = uniform(o[2]);
    r.rx = uniform(d[0]); r.ry = uniform(d[1]); r.rz = uniform(d[2]);
    return r;
}

// ---------------------------------------------------------------------------------------------------
// renderers.py:212-270 at given sorted distances [R,D]
// ---------------------------------------------------------------------------------------------------
template <int kRounds, bool kResidual>
__global__ __launch_bounds__(kBlockThreads) void render_forward_kernel(
    FieldArgs f, const float* __restrict__ instances, const float* __restrict__ mlp, RenderArgs c, const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ distances, int num_distances,
    float* __restrict__ labels, float* __restrict__ gradients, float* __restrict__ weights) {
    apply_device_schedule(f, c);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = wave_in_block();
    const int lane = lane_id();
    const int per_wave = forward_lds_floats(num_distances, f.num_instances, kResidual);
    float* dist = lds + wave * per_wave + forward_weights_floats(kResidual);
    float* dcache = dist + num_distances;
    float* coef = dcache + f.num_instances * kWave;
    Shading sh = c.sh;
    sh.mlp_lds = lds + wave * per_wave;
    const FieldBounds bounds = field_bounds(instances, f.num_instances, f.inv_t, kResidual, c.flags);
    sh.cull = bounds.margin; sh.reach = bounds.reach; sh.yaw = bounds.yaw;
    sh.mlp_bits = (c.flags & 8u) ? kMlpCentredBit : 0u;
    const int stride = static_cast<int>(gridDim.x) * waves_per_block();
    for (int ray = static_cast<int>(blockIdx.x) * waves_per_block() + wave; ray < c.num_rays; ray += stride) {
        const Ray r = load_ray(origins, directions, c.origin_stride, ray);
        const float* src = distances + static_cast<size_t>(ray) * num_distances;
        for (int idx = lane; idx < num_distances; idx += kWave) dist[idx] = src[idx];
        const RayCull rc = cull_ray_setup(instances, f.num_instances, r.ox, r.oy, r.oz, r.rx, r.ry, r.rz, coef, lane);   // (syncs the wave's LDS)
        float w[kRounds];
        float* g_out = gradients ? gradients + static_cast<size_t>(ray) * (num_distances - 1) * 3 : nullptr;
        float* w_out = weights ? weights + static_cast<size_t>(ray) * (num_distances - 1) : nullptr;
        const float label = render_pass<kRounds, true, kResidual>(instances, mlp, f.num_instances, sh, r, rc, dist, num_distances, dcache, w, g_out, w_out);
        if (lane < f.num_instances) labels[static_cast<size_t>(ray) * f.num_instances + lane] = label;
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------------------------
// scripts/main.py:511-523 (two-pass wrapper) in one launch
// ---------------------------------------------------------------------------------------------------
// Stratified distances (samplers.py:5-8) and the sorted fine uniforms of one ray, into the wave's LDS (l.coarse, l.usorted).
// In-kernel randomness: Philox4x32-10 per (ray, sample).  The S fine uniforms are needed SORTED (samplers.py:22); instead of drawing
// and sorting, they are generated in order as normalised partial sums of S+1 exponential spacings (the order statistics of S iid
// uniforms have exactly this distribution): one log + one wave scan instead of a sort.  Supplied uniforms (parity tests, the
// API-faithful path) are rank-sorted in LDS unless the caller says they are sorted.
template <int kRoundsS>
__device__ __forceinline__ void stage_ray_samples(const WaveLds& l, const RenderArgs& c, int S, int ray, const float* __restrict__ u_coarse,
                                                  const float* __restrict__ u_fine, float* __restrict__ u_coarse_out,
                                                  float* __restrict__ u_fine_out, bool sorted_input, int lane) {
    const size_t row = static_cast<size_t>(ray) * S;
    const bool philox = (u_coarse == nullptr || u_fine == nullptr);
    float spacing[kRoundsS];
    float running = 0.0f, extra_spacing = 0.0f;
#pragma unroll
    for (int k = 0; k < kRoundsS; ++k) {
        const int idx = k * kWave + lane;
        spacing[k] = 0.0f;
        if (k * kWave >= S) continue;
        float uc = 0.0f, uf = 0.0f;
        if (philox) {
            const Philox4 rnd = philox4x32_10(static_cast<uint32_t>(ray), static_cast<uint32_t>(idx),
                                              static_cast<uint32_t>(c.stream_offset), static_cast<uint32_t>(c.stream_offset >> 32),
                                              static_cast<uint32_t>(c.seed), static_cast<uint32_t>(c.seed >> 32));
            uc = uniform_from_bits(rnd.x);
            uf = uniform_from_bits(rnd.y);
            if (k == 0) extra_spacing = -fast_log(1.0f - read_lane(uniform_from_bits(rnd.z), 0));   // the (S+1)-th spacing
        }
        const bool valid = idx < S;
        if (u_coarse != nullptr && valid) uc = u_coarse[row + idx];
        if (u_fine != nullptr && valid) uf = u_fine[row + idx];
        if (valid) {
            const float lo = torch_linspace(c.near, c.far, S + 1, idx);
            const float hi = torch_linspace(c.near, c.far, S + 1, idx + 1);
            l.coarse[idx] = torch_lerp(lo, hi, uc);
            if (u_coarse_out != nullptr) u_coarse_out[row + idx] = uc;
        }
        if (u_fine == nullptr) {
            const float e = valid ? -fast_log(1.0f - uf) : 0.0f;          // exponential spacing, > 0
            const float inclusive = wave_inclusive_sum(e) + running;
            spacing[k] = inclusive;
            running = read_lane(inclusive, kWave - 1);
        } else if (valid) {
            (sorted_input ? l.usorted : l.uraw)[idx] = uf;
            if (u_fine_out != nullptr) u_fine_out[row + idx] = uf;
        }
    }
    if (u_fine == nullptr) {
        const float inv_total = fast_rcp(running + extra_spacing);
#pragma unroll
        for (int k = 0; k < kRoundsS; ++k) {
            const int idx = k * kWave + lane;
            if (idx < S) {
                const float sorted_u = fminf(spacing[k] * inv_total, 0.99999994f);
                l.usorted[idx] = sorted_u;
                if (u_fine_out != nullptr) u_fine_out[row + idx] = sorted_u;
            }
        }
    }
    wave_lds_sync();
    if (u_fine != nullptr && !sorted_input) {
        rank_sort<kRoundsS>(l.uraw, l.usorted, S);
        wave_lds_sync();
    }
}

template <int kRounds, bool kResidual>
__global__ __launch_bounds__(kBlockThreads) void render_hierarchical_kernel(
    FieldArgs f, const float* __restrict__ instances, const float* __restrict__ mlp, RenderArgs c, const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ u_coarse, const float* __restrict__ u_fine,
    float* __restrict__ labels, float* __restrict__ distances, float* __restrict__ gradients, float* __restrict__ weights,
    float* __restrict__ coarse_weights, float* __restrict__ u_coarse_out, float* __restrict__ u_fine_out) {
    apply_device_schedule(f, c);
    constexpr int kRoundsS = (kRounds + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = wave_in_block();
    const int lane = lane_id();
    const int S = c.num_samples;
    float* mine = lds + wave * hierarchical_lds_floats(S, f.num_instances, kResidual);
    const WaveLds l = carve_lds(mine + forward_weights_floats(kResidual), S, f.num_instances);
    const bool sorted_input = (u_fine != nullptr) && (c.flags & 1u);
    Shading sh = c.sh;
    sh.mlp_lds = mine;
    const FieldBounds bounds = field_bounds(instances, f.num_instances, f.inv_t, kResidual, c.flags);
    sh.cull = bounds.margin; sh.reach = bounds.reach; sh.yaw = bounds.yaw;
    sh.mlp_bits = (c.flags & 8u) ? kMlpCentredBit : 0u;
    const int stride = static_cast<int>(gridDim.x) * waves_per_block();
    for (int ray = static_cast<int>(blockIdx.x) * waves_per_block() + wave; ray < c.num_rays; ray += stride) {
        const Ray r = load_ray(origins, directions, c.origin_stride, ray);
        const RayCull rc = cull_ray_setup(instances, f.num_instances, r.ox, r.oy, r.oz, r.rx, r.ry, r.rz, l.cull, lane);
        stage_ray_samples<kRoundsS>(l, c, S, ray, u_coarse, u_fine, u_coarse_out, u_fine_out, sorted_input, lane);
        // ---- pass 1: coarse weights only (no labels, no outputs) ----------------------------------
        float w1[kRoundsS];
        render_pass<kRoundsS, false, kResidual>(instances, mlp, f.num_instances, sh, r, rc, l.coarse, S, l.dcache, w1, nullptr, nullptr);
        if (coarse_weights != nullptr) {                             // pass 1's compositing weights (main.py:511-523: what pass 1 hands to pass 2)
#pragma unroll
            for (int k = 0; k < kRoundsS; ++k)
                if (k * kWave + lane < S - 1) coarse_weights[static_cast<size_t>(ray) * (S - 1) + k * kWave + lane] = w1[k];
        }
        if (c.flags & 2u) {
            float total = 0.0f;
#pragma unroll
            for (int k = 0; k < kRoundsS; ++k) total += wave_sum(w1[k]);
            if (total == 0.0f) {                                    // wave-uniform: exact miss
                if (lane < f.num_instances) labels[static_cast<size_t>(ray) * f.num_instances + lane] = 0.0f;
                // sentinel row: the backward skips it (the exact adjoint of an exact miss is zero)
                if (distances != nullptr && lane == 0) distances[static_cast<size_t>(ray) * 2 * S] = __builtin_nanf("");
                wave_lds_sync();
                continue;
            }
        }
        // ---- importance sampling + merge (samplers.py:11-36, renderers.py:198-210) ---------------
        importance_merge<kRoundsS>(l, S, w1);
        // ---- pass 2 ----------------------------------------------------------------------------------
        float w2[kRounds];
        const int D = 2 * S;
        float* g_out = gradients ? gradients + static_cast<size_t>(ray) * (D - 1) * 3 : nullptr;
        float* w_out = weights ? weights + static_cast<size_t>(ray) * (D - 1) : nullptr;
        const float label = render_pass<kRounds, true, kResidual>(instances, mlp, f.num_instances, sh, r, rc, l.merged, D, l.dcache, w2, g_out, w_out);
        if (lane < f.num_instances) labels[static_cast<size_t>(ray) * f.num_instances + lane] = label;
        if (distances != nullptr) {
            float* dst = distances + static_cast<size_t>(ray) * D;
            for (int idx = lane; idx < D; idx += kWave) dst[idx] = l.merged[idx];
        }
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------------------------
// Adjoint of render_forward w.r.t. the packed instances.  Derivation: DESIGN.md "Backward";
// float64 blueprint: oracle/analytic.py (phase A / phase B).
// ---------------------------------------------------------------------------------------------------
struct SampleAdjoint {
    float x, y, z;        // sample position
    float m, inv_z, us;   // soft-min shift, 1/Z, u - m
    float wgt;            // compositing weight
    float lam_z;          // sum_n lambda_n w_n
    float u_bar;          // dL/du
    float gbx, gby, gbz;  // dL/dg
    float A, B;           // g_bar . g,  g_bar . sum_i w_i grad d_i
};

// Per-ray state of the adjoint, one or a few samples per lane.
template <int kRounds>
struct RayAdjoint {
    SampleAdjoint sa[kRounds];
    Opacity op[kRounds];
    float gx[kRounds], gy[kRounds], gz[kRounds], trans[kRounds], delta[kRounds];
    // culling decisions of the forward sweep (wave-uniform, bit i = instance i): evaluated in the round at all.  The later phases read
    // these instead of repeating the bound test per instance and round.
    unsigned long long near_any[kRounds];
    bool last_running;       // the last round's soft-min used the running minimum: its cache rows hold distances, not soft-min terms
};

// The residual jets of one wave: [round][instance][lane] float4 (value, local gradient) followed by one word per (round, instance) that
// says WHICH lanes hold one.  residual_forward evaluates 16-point tiles (residual.h: tile_plan); lanes of tiles it skipped have a zero jet
// and -- round 6 -- no store: on config 3 three quarters of the float4s were zeros, written through the L2s (the waves' jets are 64 MB,
// beyond them) and read back by the label mix and the per-instance phase: 84 of the step's 288 GB.  The word is the tile plan
// (bits 0..3: tiles evaluated, bits 4..9: the lane the tiles are counted from); readers take zero where no jet was stored.
__host__ __device__ constexpr size_t jet_wave_float4s(int rounds, int num_instances) {
    return static_cast<size_t>(rounds) * num_instances * kWave + (static_cast<size_t>(rounds) * num_instances + 3) / 4;
}
template <int kRounds>
__device__ __forceinline__ unsigned* jet_codes(float4* rcache, int N) { return reinterpret_cast<unsigned*>(rcache + static_cast<size_t>(kRounds) * N * kWave); }
template <int kRounds>
__device__ __forceinline__ const unsigned* jet_codes(const float4* rcache, int N) { return reinterpret_cast<const unsigned*>(rcache + static_cast<size_t>(kRounds) * N * kWave); }
__device__ __forceinline__ unsigned jet_code(unsigned long long need) {
    const TilePlan plan = tile_plan(need);
    return plan.tiles | (static_cast<unsigned>(plan.start) << 4);
}
__device__ __forceinline__ bool jet_stored(unsigned code, int lane) { return (((code & 15u) >> (((lane - static_cast<int>(code >> 4)) & 63) >> 4)) & 1u) != 0u; }
// The jet of (round k, instance i) at this lane; `code` is wave-uniform (one load of one address, made scalar).
template <int kRounds>
__device__ __forceinline__ float4 load_jet(const float4* rcache, int N, int k, int i, int lane) {
    const unsigned code = __builtin_amdgcn_readfirstlane(jet_codes<kRounds>(rcache, N)[k * N + i]);
    return jet_stored(code, lane) ? rcache[(k * N + i) * kWave + lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// The instance loop of the forward sweep for round k (render.h: union_loop, plus what the adjoint keeps: the 16-lane row masks and
// the residual jets).  Instances that fail the exact test are cleared from st.near_any[k]: the later phases never see them.
template <int kRounds, bool kResidual, bool kCacheD, bool kRunning, bool kYaw>
__device__ __forceinline__ UnionSums sweep_union_loop(RayAdjoint<kRounds>& st, int k, const float* __restrict__ instances, const float* __restrict__ mlp, int N,
                                                      const Shading& sh, const RoundCull& cull, float floor, const float* lam,
                                                      float* dcache, int lane, float4* rcache, unsigned long long live) {
    UnionSums sums = union_init(kRunning, floor);
    float best = cull.nearest_hi;
    for (unsigned long long todo = st.near_any[k]; todo != 0ull; todo &= todo - 1ull) {
        const int i = __builtin_ctzll(todo);
        const Instance in = load_instance(instances, i);
        BoxEval e = box_value<kYaw>(in, st.sa[k].x, st.sa[k].y, st.sa[k].z);
        const unsigned long long near = __ballot(!(e.d - best > sh.cull));
        if (near == 0ull) { st.near_any[k] &= ~(1ull << i); continue; }
        best = fminf(best, e.d);
        box_gradient<kYaw>(e, in);
        if (kResidual) {
            const Residual res = residual_forward_packed(mlp + i * sh.mlp_stride, e.px, e.py, e.pz, near & live, sh.mlp_bits, sh.mlp_lds);
            add_residual<kYaw>(e, in, res);
            if (rcache) {
                const unsigned code = jet_code(near & live);                      // (the plan residual_forward_packed just ran)
                if (jet_stored(code, lane)) rcache[(k * N + i) * kWave + lane] = make_float4(res.value, res.gx, res.gy, res.gz);
                if (lane == 0) jet_codes<kRounds>(rcache, N)[k * N + i] = code;
            }
        }
        // the distance cache feeds the label sums of the round: with a fixed shift it can hold the soft-min term itself
        const float term = union_accumulate<kRunning>(sums, e.d, e.gwx, e.gwy, e.gwz, lam ? lam[i] : 0.0f, sh.inv_t);
        if (kCacheD) dcache[i * kWave + lane] = term;
    }
    return sums;
}

// Phase A, forward sweep: union sums (with Lambda = sum_n lambda_n e_n when `lam` is given), opacity, transmittance.
// kCacheD (fused loss kernel): the instance distances of the current round are cached in `dcache` [N][64] and turned into the
// ray's labels right away; the return value then holds label n in lane n.
template <int kRounds, bool kResidual, bool kCacheD>
__device__ __forceinline__ float adjoint_forward_sweep(RayAdjoint<kRounds>& st, const float* __restrict__ instances, const float* __restrict__ mlp,
                                                       int N, const Shading& sh, const Ray& r, const RayCull& rc, const float* dist, int num_points,
                                                       const float* lam, float* dcache, int lane, float4* rcache = nullptr,
                                                       int first_point = 0, float* carry_out = nullptr) {
    const float inv_t = sh.inv_t;
    float label = 0.0f;
    float carry = 1.0f;
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
        const int s = first_point + k * kWave + lane;
        const bool valid = s < num_points;
        const int s0 = valid ? s : (num_points - 1);
        const float d0 = dist[s0], d1 = dist[s0 + 1];
        st.delta[k] = d1 - d0;
        const float mid = (d0 + d1) / 2.0f;
        st.sa[k].x = r.ox + r.rx * mid; st.sa[k].y = r.oy + r.ry * mid; st.sa[k].z = r.oz + r.rz * mid;
        // culling (field.h): which instances this round has to evaluate
        RoundCull cull;
        st.near_any[k] = cull_round_mask<kCacheD>(rc, N, mid, sh.cull, dcache, lane, &cull);
        const unsigned long long live = __ballot(valid);                    // padding lanes repeat the last point: their tiles skip the MLP
        UnionSums sums;
        const float floor = cull.nearest_lo - sh.reach;
        bool running = sh.reach < 0.0f || wave_any(!((cull.nearest_hi + 1.0f - floor) * sh.inv_t <= kUnionFloorSpan));   // wave-uniform (render.h: eval_union)
        if (!running) {
            sums = sh.yaw ? sweep_union_loop<kRounds, kResidual, kCacheD, false, true>(st, k, instances, mlp, N, sh, cull, floor, lam, dcache, lane, rcache, live)
                          : sweep_union_loop<kRounds, kResidual, kCacheD, false, false>(st, k, instances, mlp, N, sh, cull, floor, lam, dcache, lane, rcache, live);
            running = wave_any(!(sums.Z >= kUnionTinyZ));                    // the fixed shift underflowed somewhere: repeat the round
        }
        if (running) sums = sweep_union_loop<kRounds, kResidual, kCacheD, true, false>(st, k, instances, mlp, N, sh, cull, 0.0f, lam, dcache, lane, rcache, live);
        const UnionValue v = union_finish(sums, inv_t);
        st.op[k] = opacity_of(v, r, st.delta[k], sh);
        const float alpha = valid ? st.op[k].alpha : 0.0f;
        st.op[k].alpha = alpha;
        const float inclusive = wave_inclusive_product(1.0f - alpha);
        st.trans[k] = carry * wave_shift_up(inclusive, 1.0f, lane);
        carry *= read_lane(inclusive, kWave - 1);
        st.sa[k].m = v.m; st.sa[k].inv_z = v.inv_z; st.sa[k].us = v.us;
        st.sa[k].wgt = st.trans[k] * alpha;
        st.sa[k].lam_z = sums.L * v.inv_z;
        st.gx[k] = v.gx; st.gy[k] = v.gy; st.gz[k] = v.gz;
        // B needs sum_i w_i grad d_i: stash it in (gbx,gby,gbz) until the reverse sweep
        st.sa[k].gbx = v.b0x; st.sa[k].gby = v.b0y; st.sa[k].gbz = v.b0z;
        if (kCacheD && first_point + k * kWave < num_points) {
            const float scale = st.sa[k].wgt * v.inv_z;
            if (k == kRounds - 1) st.last_running = running;
            if (running) {                                                   // the cache holds distances
                for (unsigned long long todo = st.near_any[k]; todo != 0ull; todo &= todo - 1ull) {
                    const int i = __builtin_ctzll(todo);
                    const float total = wave_sum(fast_exp(-(dcache[i * kWave + lane] - v.m) * inv_t) * scale);
                    label = (lane == i) ? (label + total) : label;
                }
            } else {                                                         // the cache holds exp(-(d_i - m)/T) (sweep_union_loop)
                for (unsigned long long todo = st.near_any[k]; todo != 0ull; todo &= todo - 1ull) {
                    const int i = __builtin_ctzll(todo);
                    const float total = wave_sum(dcache[i * kWave + lane] * scale);
                    label = (lane == i) ? (label + total) : label;
                }
            }
        }
    }
    if (carry_out != nullptr) *carry_out = carry;
    return label;
}

// Phase A, reverse sweep: labels -> weights -> opacity -> (u_bar, g_bar).  Returns (wave-uniformly) whether any adjoint is non-zero.
template <int kRounds>
__device__ __forceinline__ bool adjoint_reverse_sweep(RayAdjoint<kRounds>& st, const Shading& sh, const Ray& r, int num_points,
                                                      const float* grad_weights_row, const float* grad_gradients_row, int lane,
                                                      float eikonal_scale = 0.0f, int first_point = 0, float suffix_carry = 0.0f) {
    bool any_flow = false;
#pragma unroll
    for (int k = kRounds - 1; k >= 0; --k) {
        const int s = first_point + k * kWave + lane;
        const bool valid = s < num_points;
        float w_bar = st.sa[k].lam_z;
        if (grad_weights_row != nullptr && valid) w_bar += grad_weights_row[s];
        const float contrib = valid ? w_bar * st.sa[k].wgt : 0.0f;
        const float rev_inclusive = wave_inclusive_sum(wave_reverse(contrib, lane));
        const float suffix_inclusive = wave_reverse(rev_inclusive, lane);
        const float Q = suffix_inclusive - contrib + suffix_carry;          // sum over later samples
        suffix_carry += read_lane(rev_inclusive, kWave - 1);
        const float alpha = st.op[k].alpha;
        const float alpha_bar = w_bar * st.trans[k] - Q * fast_rcp(1.0f - alpha);
        const float x_bar = (valid && st.op[k].xx > 0.0f) ? alpha_bar : 0.0f;
        const float inv_pe = fast_rcp(st.op[k].phi_p + sh.eps);
        const float phi_p_bar = x_bar * (st.op[k].phi_n + sh.eps) * inv_pe * inv_pe;
        const float phi_n_bar = -x_bar * inv_pe;
        const float sp_bar = phi_p_bar * st.op[k].phi_p * (1.0f - st.op[k].phi_p) * sh.inv_std;
        const float sn_bar = phi_n_bar * st.op[k].phi_n * (1.0f - st.op[k].phi_n) * sh.inv_std;
        const float u_bar = sp_bar + sn_bar;
        const float cprime_bar = (sn_bar - sp_bar) * st.delta[k] / 2.0f;
        const float slope = (1.0f - sh.ratio) * ((0.5f - 0.5f * st.op[k].cosine > 0.0f) ? 0.5f : 0.0f)
                          + sh.ratio * ((-st.op[k].cosine > 0.0f) ? 1.0f : 0.0f);
        const float cos_bar = cprime_bar * slope;
        const float nbx = cos_bar * r.rx, nby = cos_bar * r.ry, nbz = cos_bar * r.rz;
        const float n_dot = st.op[k].nx * nbx + st.op[k].ny * nby + st.op[k].nz * nbz;
        float gbx = (nbx - st.op[k].nx * n_dot) * st.op[k].inv_gn;
        float gby = (nby - st.op[k].ny * n_dot) * st.op[k].inv_gn;
        float gbz = (nbz - st.op[k].nz * n_dot) * st.op[k].inv_gn;
        if (grad_gradients_row != nullptr && valid) {
            const float* gg = grad_gradients_row + s * 3;
            gbx += gg[0]; gby += gg[1]; gbz += gg[2];
        }
        if (eikonal_scale != 0.0f) {        // fused eikonal term (main.py:679-687): d/dg of eikonal_scale * (|g| - 1)^2, torch's norm' (0) = 0
            const float norm = fast_sqrt(st.gx[k] * st.gx[k] + st.gy[k] * st.gy[k] + st.gz[k] * st.gz[k]);
            const float pull = (norm > 0.0f) ? eikonal_scale * 2.0f * (norm - 1.0f) * fast_rcp(norm) : 0.0f;
            gbx += pull * st.gx[k]; gby += pull * st.gy[k]; gbz += pull * st.gz[k];
        }
        if (!valid) { gbx = 0.0f; gby = 0.0f; gbz = 0.0f; }
        st.sa[k].B = gbx * st.sa[k].gbx + gby * st.sa[k].gby + gbz * st.sa[k].gbz;
        st.sa[k].A = gbx * st.gx[k] + gby * st.gy[k] + gbz * st.gz[k];
        st.sa[k].gbx = gbx; st.sa[k].gby = gby; st.sa[k].gbz = gbz;
        st.sa[k].u_bar = valid ? u_bar : 0.0f;
        if (!valid) { st.sa[k].wgt = 0.0f; st.sa[k].lam_z = 0.0f; }
        any_flow = any_flow || (st.sa[k].u_bar != 0.0f) || (gbx != 0.0f) || (gby != 0.0f) || (gbz != 0.0f) || (st.sa[k].wgt != 0.0f);
    }
    return __ballot(any_flow) != 0ull;
}

// Label-adjoint mix of the fused step kernels: Lambda_s = sum_n lambda_n w_{s,n} for the samples of every round.  The last round's
// soft-min terms are still in the wave's distance cache; for the earlier rounds the box distances are re-evaluated (value only,
// ~30 instructions) instead of keeping every round's [N][64] cache in LDS, which would cost render_silhouette_kernel a workgroup
// per CU; residual fields add the residual value the sweep left in its jet cache.
template <int kRounds, bool kResidual, bool kYaw>
__device__ __forceinline__ void adjoint_label_mix(RayAdjoint<kRounds>& st, const float* __restrict__ instances, int N, float inv_t, int num_points,
                                                  const float* lam, int lane, const float4* rcache, const float* dcache, int first_point = 0) {
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
        if (first_point + k * kWave >= num_points) continue;
        float acc = 0.0f;
        if (k == kRounds - 1) {
            if (st.last_running) {
                for (unsigned long long todo = st.near_any[k]; todo != 0ull; todo &= todo - 1ull) {
                    const int i = __builtin_ctzll(todo);
                    if (lam[i] != 0.0f) acc += lam[i] * fast_exp(-(dcache[i * kWave + lane] - st.sa[k].m) * inv_t);
                }
            } else {
                for (unsigned long long todo = st.near_any[k]; todo != 0ull; todo &= todo - 1ull) {
                    const int i = __builtin_ctzll(todo);
                    acc += lam[i] * dcache[i * kWave + lane];
                }
            }
        } else {
            for (unsigned long long todo = st.near_any[k]; todo != 0ull; todo &= todo - 1ull) {   // the instances the forward sweep evaluated
                const int i = __builtin_ctzll(todo);
                if (lam[i] == 0.0f) continue;                               // wave-uniform
                const Instance in = load_instance_as<!kResidual>(instances, i);
                float d = box_value<kYaw>(in, st.sa[k].x, st.sa[k].y, st.sa[k].z).d;
                if (kResidual) d += load_jet<kRounds>(rcache, N, k, i, lane).x;
                acc += lam[i] * fast_exp(-(d - st.sa[k].m) * inv_t);
            }
        }
        st.sa[k].lam_z = acc * st.sa[k].inv_z;
    }
}

// x - t of a sample from its position p in the instance's frame: p = R^T (x - t), so x - t = R p.  The seeds carried it as three more
// floats until round 6 (write only what the adjoint cannot rebuild); nine multiply-adds per point where the MLP
// adjoint spends thousands, and three tenths of the seed traffic.
struct SeedOffset { float x, y, z; };
__device__ __forceinline__ SeedOffset seed_offset(const Instance& in, float px, float py, float pz) {
    return {fmaf(in.r02, pz, fmaf(in.r01, py, in.r00 * px)), fmaf(in.r12, pz, fmaf(in.r11, py, in.r10 * px)), fmaf(in.r22, pz, fmaf(in.r21, py, in.r20 * px))};
}

// Where the per-instance phase leaves the seeds (kSeedFloats x 64 floats per (round, instance), the points that matter first) and
// their COUNTS (0 .. 64 leading columns; "masks" for historical reasons: rounds 2 and 3 stored 4-bit tile masks) for the MLP adjoint.  Two layouts: wave-private batches (render_backward_kernel / render_residual_step_kernel: [round][instance], 32-bit masks
// in LDS) and the launch-wide dense table of the split residual step ([instance][ray of chunk][round], byte masks in global memory).
struct SeedSink {
    float* seeds;                    // base of this ray's seeds
    long long instance_stride;       // floats between instances
    long long round_stride;          // floats between rounds
    unsigned* masks32;               // [round * mask_round_stride + instance * mask_instance_stride] (LDS), or nullptr
    unsigned char* masks8;           // the same in bytes (global), or nullptr
    long long mask_instance_stride, mask_round_stride;
};

__device__ __forceinline__ SeedSink batch_seed_sink(float* ray_seeds, unsigned* ray_masks, int N) {
    return SeedSink{ray_seeds, static_cast<long long>(kSeedFloats) * kWave, static_cast<long long>(N) * kSeedFloats * kWave, ray_masks, nullptr, 1, N};
}

// Phase B: per instance, adjoint of (d_i, grad d_i) w.r.t. (t, R, dim) (and the residual MLP weights), accumulated into the wave's
// LDS rows G [N,16]; for residual fields it leaves the seeds and tile masks of the MLP adjoint (adjoint_phase_mlp) instead of running it.
template <int kRounds, bool kResidual, bool kYaw>
__device__ __forceinline__ void adjoint_phase_b(const RayAdjoint<kRounds>& st, const float* __restrict__ instances, const float* __restrict__ mlp,
                                                int N, float inv_t, int num_points, const float* lam, float* G, int lane,
                                                const float4* rcache, const SeedSink& sink, float cull_margin = 3.0e38f, int first_point = 0) {
    unsigned long long todo = 0ull;                                           // instances evaluated in some round of this ray
#pragma unroll
    for (int k = 0; k < kRounds; ++k) todo |= (first_point + k * kWave < num_points) ? st.near_any[k] : 0ull;
    for (; todo != 0ull; todo &= todo - 1ull) {
        const int i = __builtin_ctzll(todo);
        bool active[kRounds];
        int counts[kRounds];                                                  // residual fields: leading seed columns that matter, per round
        bool any_active = false;
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {                                   // the forward sweep's culling decisions (scalar bit tests)
            active[k] = (first_point + k * kWave < num_points) && ((st.near_any[k] >> i) & 1ull) != 0ull;
            any_active = any_active || active[k];
            counts[k] = 0;
        }
        if (kResidual && !any_active && lane < kRounds) {                     // (an instance no round of this ray evaluated: nothing for the MLP adjoint)
            const long long at = lane * sink.mask_round_stride + i * sink.mask_instance_stride;
            if (sink.masks32) sink.masks32[at] = 0u;
            else sink.masks8[at] = 0;
        }
        if (!any_active) continue;                                            // negligible for this ray (field.h culling)
        const Instance in = load_instance_as<!kResidual>(instances, i);
        const float lam_i = lam[i];
        float at0 = 0, at1 = 0, at2 = 0, ad0 = 0, ad1 = 0, ad2 = 0;
        float r00 = 0, r01 = 0, r02 = 0, r10 = 0, r11 = 0, r12 = 0, r20 = 0, r21 = 0, r22 = 0;
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            if (!active[k]) continue;
            BoxEval e = eval_box<kYaw>(in, st.sa[k].x, st.sa[k].y, st.sa[k].z);
            unsigned long long need = ~0ull;
            if (kResidual) {
                // The lanes whose points the MLP adjoint must see: those on which the instance is not negligible -- box distance within the
                // culling margin of the union value u >= min_j d_j, a superset of the exact criterion of the forward sweep (field.h) that
                // needs no state of it; lanes without a point of their own (padding) never.  The rest (soft-min weight < exp(-18)) is
                // what the forward sweep's tile masks dropped at tile granularity in rounds 2 and 3.
                const bool own = first_point + k * kWave + lane < num_points;
                need = __ballot(own && !(e.d - (st.sa[k].m + st.sa[k].us) > cull_margin));
                // the residual jet of this (round, instance) was left by the forward sweep
                const float4 res = load_jet<kRounds>(rcache, N, k, i, lane);
                add_residual<kYaw>(e, in, Residual{res.x, res.y, res.z, res.w});
            }
            const float ds = e.d - st.sa[k].m;
            const float w = fast_exp(-ds * inv_t) * st.sa[k].inv_z;
            const float cc = w * (1.0f - (ds - st.sa[k].us) * inv_t);
            // g_bar in the instance's frame, R^T g_bar: both beta = g_bar . grad d_i = (R^T g_bar) . gl  and  gl_bar = R^T (cc g_bar) need it,
            // and the world-frame gradient R gl of this instance is then not needed at all (one rotation per pair instead of two)
            const float gx_ = st.sa[k].gbx, gy_ = st.sa[k].gby, gz_ = st.sa[k].gbz;
            const float rgx = kYaw ? fmaf(in.r20, gz_, in.r00 * gx_) : fmaf(in.r20, gz_, fmaf(in.r10, gy_, in.r00 * gx_));
            const float rgy = kYaw ? gy_ : fmaf(in.r21, gz_, fmaf(in.r11, gy_, in.r01 * gx_));
            const float rgz = kYaw ? fmaf(in.r22, gz_, in.r02 * gx_) : fmaf(in.r22, gz_, fmaf(in.r12, gy_, in.r02 * gx_));
            const float beta = rgx * e.glx + rgy * e.gly + rgz * e.glz;
            const float d_bar = st.sa[k].u_bar * cc
                              + inv_t * (-beta * cc + w * st.sa[k].A - beta * w + cc * st.sa[k].B)
                              - inv_t * w * st.sa[k].wgt * (lam_i - st.sa[k].lam_z);
            const float gwbx = cc * gx_, gwby = cc * gy_, gwbz = cc * gz_;
            const float glbx = cc * rgx, glby = cc * rgy, glbz = cc * rgz;       // gl_bar_j = sum_k R_kj gw_bar_k
            const float sx = sign_of(e.px), sy = sign_of(e.py), sz = sign_of(e.pz);
            const float vx = sx * glbx, vy = sy * glby, vz = sz * glbz;
            const float inv_n = box_inverse_norm(e);
            const float hx = fmaxf(e.qx, 0.0f) * inv_n, hy = fmaxf(e.qy, 0.0f) * inv_n, hz = fmaxf(e.qz, 0.0f) * inv_n;
            const float hv = hx * vx + hy * vy + hz * vz;
            const float qbx = d_bar * e.hx + ((e.qx > 0.0f) ? (vx - hx * hv) * inv_n : 0.0f);
            const float qby = d_bar * e.hy + ((e.qy > 0.0f) ? (vy - hy * hv) * inv_n : 0.0f);
            const float qbz = d_bar * e.hz + ((e.qz > 0.0f) ? (vz - hz * hv) * inv_n : 0.0f);
            const float pbx = sx * qbx, pby = sy * qby, pbz = sz * qbz;
            ad0 -= qbx; ad1 -= qby; ad2 -= qbz;
            if (kResidual) {        // seeds of the residual adjoint (main.py:451-458): value adjoint d_bar, local-gradient adjoint gl_bar;
                // left for the MLP adjoint with the points that matter gathered into the leading columns (residual.h: packed_column)
                const int column = packed_column(need, lane, counts[k]);
                float* dst = sink.seeds + k * sink.round_stride + i * sink.instance_stride + column;
                // Round 6: the launch-wide table's readers (residual_mlp_adjoint_kernel: the points of an item's slots as ONE stream) take the
                // leading counts[k] columns of a slot and nothing else, so the other lanes' seeds -- three quarters of the 285 GB per config-3
                // step that made the step's HBM traffic 28x its API-faithful bytes -- are not written.  (The wave-private batches of the
                // one-kernel forms are read tile by tile: every lane writes there.)
                if (sink.masks8 == nullptr || column < counts[k]) {
                    dst[0 * kWave] = e.px; dst[1 * kWave] = e.py; dst[2 * kWave] = e.pz; dst[3 * kWave] = d_bar;
                    dst[4 * kWave] = glbx; dst[5 * kWave] = glby; dst[6 * kWave] = glbz;
                }
            }
            r00 += e.relx * pbx + gwbx * e.glx; r01 += e.relx * pby + gwbx * e.gly; r02 += e.relx * pbz + gwbx * e.glz;
            r10 += e.rely * pbx + gwby * e.glx; r11 += e.rely * pby + gwby * e.gly; r12 += e.rely * pbz + gwby * e.glz;
            r20 += e.relz * pbx + gwbz * e.glx; r21 += e.relz * pby + gwbz * e.gly; r22 += e.relz * pbz + gwbz * e.glz;
            if (kYaw) {
                at0 -= in.r00 * pbx + in.r02 * pbz; at1 -= pby; at2 -= in.r20 * pbx + in.r22 * pbz;
            } else {
                at0 -= in.r00 * pbx + in.r01 * pby + in.r02 * pbz;
                at1 -= in.r10 * pbx + in.r11 * pby + in.r12 * pbz;
                at2 -= in.r20 * pbx + in.r21 * pby + in.r22 * pbz;
            }
        }
        if (kResidual && lane < kRounds) {                                    // how many leading seed columns of which rounds the MLP adjoint has to visit
            int mine_count = 0;
#pragma unroll
            for (int k = 0; k < kRounds; ++k) mine_count = (lane == k) ? counts[k] : mine_count;
            const long long at = lane * sink.mask_round_stride + i * sink.mask_instance_stride;
            if (sink.masks32) sink.masks32[at] = static_cast<unsigned>(mine_count);
            else sink.masks8[at] = static_cast<unsigned char>(mine_count);
        }
        // one reduce-scatter butterfly: lane j (< 16) receives parameter j of instance i and keeps it in its LDS slot
        const float packed[16] = {at0, at1, at2, r00, r01, r02, r10, r11, r12, r20, r21, r22, ad0, ad1, ad2, 0.0f};
        const float mine = wave_reduce16_scatter(packed, lane);
        if (lane < kGradStride) G[i * kGradStride + lane] += mine;
    }
}

// Phase B, residual part: the MLP adjoint of every (ray of the batch, instance, round) the box phase left seeds for.
// * It runs AFTER the box phases, when no per-ray adjoint state is live: residual_backward (inlined here, its only call site) needs
//   ~400 registers, and as a function called from inside the box phase it saved and restored 217 of them per call -- 200 GB of
//   scratch traffic per launch on the C3-shaped bench.  The seeds (7 floats per sample) go through the workspace instead.
// * It is instance-major over a batch of kBatch rays: the 1617 weight adjoints of an instance are accumulated in LDS over the whole
//   batch and flushed into the wave's global partial row once per (batch, instance) instead of once per (ray, instance) (that
//   read-modify-write was 27 of the remaining 30 GB), the weight operands are loaded once, and one butterfly serves the batch.
// Its result p_bar = dL/d(local position) adds  rel (x) p_bar  to the rotation adjoint and  -R p_bar  to the translation adjoint.
template <int kRounds, int kBatch>
__device__ __forceinline__ void adjoint_phase_mlp(const float* __restrict__ instances, const float* __restrict__ mlp, int N, float* G, float* wbar,
                                                  float* my_mlp, int lane, const float* seeds, const unsigned* masks, unsigned mlp_bits) {
    for (int i = 0; i < N; ++i) {
        unsigned any = 0u;
        for (int slot = 0; slot < kBatch * kRounds; ++slot) any |= masks[slot * N + i];
        if (__builtin_amdgcn_readfirstlane(any) == 0u) continue;
        const Instance in = load_instance(instances, i);
        float at0 = 0, at1 = 0, at2 = 0;
        float r00 = 0, r01 = 0, r02 = 0, r10 = 0, r11 = 0, r12 = 0, r20 = 0, r21 = 0, r22 = 0;
#pragma unroll 1
        for (int slot = 0; slot < kBatch * kRounds; ++slot) {           // slot = ray-of-batch * kRounds + round
            const unsigned count = __builtin_amdgcn_readfirstlane(masks[slot * N + i]);
            if (count == 0u) continue;
            const unsigned rows = tiles_of_count(static_cast<int>(count));
            const float* src = seeds + static_cast<size_t>(slot * N + i) * (kSeedFloats * kWave) + lane;
            const float px = src[0 * kWave], py = src[1 * kWave], pz = src[2 * kWave];
            const SeedOffset rel = seed_offset(in, px, py, pz);
            const float relx = rel.x, rely = rel.y, relz = rel.z;
            const ResidualAdjoint ra = residual_backward(mlp + i * kMlpWeights, px, py, pz, src[3 * kWave],
                                                         src[4 * kWave], src[5 * kWave], src[6 * kWave], wbar, lane, rows | mlp_bits);
            r00 += relx * ra.px; r01 += relx * ra.py; r02 += relx * ra.pz;
            r10 += rely * ra.px; r11 += rely * ra.py; r12 += rely * ra.pz;
            r20 += relz * ra.px; r21 += relz * ra.py; r22 += relz * ra.pz;
            at0 -= in.r00 * ra.px + in.r01 * ra.py + in.r02 * ra.pz;
            at1 -= in.r10 * ra.px + in.r11 * ra.py + in.r12 * ra.pz;
            at2 -= in.r20 * ra.px + in.r21 * ra.py + in.r22 * ra.pz;
        }
        const float packed[16] = {at0, at1, at2, r00, r01, r02, r10, r11, r12, r20, r21, r22, 0.0f, 0.0f, 0.0f, 0.0f};
        const float mine = wave_reduce16_scatter(packed, lane);
        if (lane < kGradStride) G[i * kGradStride + lane] += mine;
        // flush this instance's MLP weight adjoints into the wave's global row (wave-private RMW)
        float* dst = my_mlp + static_cast<size_t>(i) * kMlpWeights;
        wave_lds_sync();
        for (int idx = lane; idx < kMlpWeights; idx += kWave) { dst[idx] += wbar[idx]; wbar[idx] = 0.0f; }
        wave_lds_sync();
    }
}

template <int kRounds, bool kResidual>
__global__ __launch_bounds__(kBlockThreads) void render_backward_kernel(
    FieldArgs f, const float* __restrict__ instances, const float* __restrict__ mlp, RenderArgs c,
    const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ distances, int num_distances,
    const float* __restrict__ grad_labels, const float* __restrict__ grad_gradients, const float* __restrict__ grad_weights,
    float* __restrict__ partials, float* __restrict__ mlp_partials, float4* __restrict__ residual_cache, float* __restrict__ seed_cache) {
    apply_device_schedule(f, c);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = wave_in_block();
    const int lane = lane_id();
    const int N = f.num_instances;
    const int num_points = num_distances - 1;
    const int per_wave = backward_lds_floats(num_distances, N, kResidual);
    float* wbar = lds + wave * per_wave + forward_weights_floats(kResidual);   // residual only: [1617] MLP weight adjoints of the current
    float* dist = wbar + (kResidual ? kMlpLdsFloats : 0);                // instance + the transposition scratch (residual.h), 16 B aligned
    float* lam = dist + num_distances;
    float* G = lam + N;
    float* coef = G + N * kGradStride;                                   // per-ray culling coefficients (field.h: RayCull)
    unsigned* masks = reinterpret_cast<unsigned*>(coef + cull_coef_floats(N));   // residual only: [kMlpBatch][kRounds][N] tile masks of the MLP adjoint
    for (int idx = lane; idx < N * kGradStride; idx += kWave) G[idx] = 0.0f;
    const size_t wave_global0 = static_cast<size_t>(blockIdx.x) * waves_per_block() + wave;
    float* my_mlp = kResidual ? mlp_partials + wave_global0 * (static_cast<size_t>(N) * kMlpWeights) : nullptr;
    // per-wave workspace of the residual adjoint: [round][instance][lane] residual jets (float4); [ray of batch][round][instance][7][lane] seeds
    float4* rcache = kResidual ? residual_cache + wave_global0 * jet_wave_float4s(kRounds, N) : nullptr;
    float* seeds = kResidual ? seed_cache + wave_global0 * (static_cast<size_t>(kMlpBatch) * kRounds * N * kSeedFloats * kWave) : nullptr;
    if (kResidual) {
        for (int idx = lane; idx < N * kMlpWeights; idx += kWave) my_mlp[idx] = 0.0f;
        for (int idx = lane; idx < kMlpWeights; idx += kWave) wbar[idx] = 0.0f;
    }
    Shading sh = c.sh;
    sh.mlp_lds = lds + wave * per_wave;
    const FieldBounds bounds = field_bounds(instances, N, f.inv_t, kResidual, c.flags);
    sh.cull = bounds.margin; sh.reach = bounds.reach; sh.yaw = bounds.yaw;
    sh.mlp_bits = (c.flags & 8u) ? kMlpCentredBit : 0u;
    const int stride = static_cast<int>(gridDim.x) * waves_per_block();
    constexpr int kBatch = kResidual ? kMlpBatch : 1;
    for (int first = static_cast<int>(blockIdx.x) * waves_per_block() + wave; first < c.num_rays; first += stride * kBatch) {
#pragma unroll 1
        for (int b = 0; b < kBatch; ++b) {
            const int ray = first + b * stride;
            unsigned* ray_masks = masks + b * kRounds * N;
            float* ray_seeds = kResidual ? seeds + static_cast<size_t>(b) * (kRounds * N * kSeedFloats * kWave) : nullptr;
            wave_lds_sync();
            if (kResidual) {
                for (int idx = lane; idx < kRounds * N; idx += kWave) ray_masks[idx] = 0u;      // a ray that exits early leaves no work
            }
            if (ray >= c.num_rays) continue;
            const float lam_lane = (lane < N) ? grad_labels[static_cast<size_t>(ray) * N + lane] : 0.0f;
            if (grad_gradients == nullptr && grad_weights == nullptr && wave_max(fabsf(lam_lane)) == 0.0f) continue;  // nothing flows back
            const Ray r = load_ray(origins, directions, c.origin_stride, ray);
            const float* src = distances + static_cast<size_t>(ray) * num_distances;
            for (int idx = lane; idx < num_distances; idx += kWave) dist[idx] = src[idx];
            if (lane < N) lam[lane] = lam_lane;
            const RayCull rc = cull_ray_setup(instances, N, r.ox, r.oy, r.oz, r.rx, r.ry, r.rz, coef, lane);       // (syncs the wave's LDS)
            if (dist[0] != dist[0]) continue;                            // NaN sentinel: ray skipped by the forward
            RayAdjoint<kRounds> st;
            adjoint_forward_sweep<kRounds, kResidual, false>(st, instances, mlp, N, sh, r, rc, dist, num_points, lam, nullptr, lane, rcache);
            const float* gw_row = grad_weights ? grad_weights + static_cast<size_t>(ray) * num_points : nullptr;
            const float* gg_row = grad_gradients ? grad_gradients + static_cast<size_t>(ray) * num_points * 3 : nullptr;
            if (!adjoint_reverse_sweep<kRounds>(st, sh, r, num_points, gw_row, gg_row, lane)) continue;      // exact zero adjoint
            const SeedSink sink = batch_seed_sink(ray_seeds, ray_masks, N);
            if (sh.yaw) adjoint_phase_b<kRounds, kResidual, true>(st, instances, mlp, N, f.inv_t, num_points, lam, G, lane, rcache, sink, sh.cull);
            else adjoint_phase_b<kRounds, kResidual, false>(st, instances, mlp, N, f.inv_t, num_points, lam, G, lane, rcache, sink, sh.cull);
        }
        if (kResidual) {
            wave_lds_sync();                                                 // masks: written by lanes < kRounds, read by all
            adjoint_phase_mlp<kRounds, kBatch>(instances, mlp, N, G, wbar, my_mlp, lane, seeds, masks, sh.mlp_bits);
        }
    }
    wave_lds_sync();
    float* out = partials + wave_global0 * (N * kGradStride);
    for (int idx = lane; idx < N * kGradStride; idx += kWave) out[idx] = G[idx];
}

// ---------------------------------------------------------------------------------------------------
// Fused optimisation-step kernel: two-pass render + silhouette BCE + adjoint, one launch, nothing per-sample in HBM.
//   scripts/main.py:511-523 (two-pass wrapper), :653-671 (mean BCE of clamp(labels, 1e-6, 1-1e-6) against the soft masks),
//   and the backward of both.  The BCE is element-wise in (ray, instance), so each wave can turn its ray's labels into
//   label adjoints on the spot and run the adjoint sweep on the pass-2 state it still holds in registers: no saved
//   distances, no recomputation of pass 2, no second launch.
// targets [R,N] are already in prediction order (the host applies the Hungarian permutation); instance_weights [N]
// (0 = unmatched instance) and loss_scale = 1 / (R * matched) reproduce the mean over kept elements.
// ---------------------------------------------------------------------------------------------------
template <int kRounds>
__global__ __launch_bounds__(kBlockThreads, (kRounds <= 2) ? 4 : 2) void render_silhouette_kernel(
    FieldArgs f, const float* __restrict__ instances, RenderArgs c, const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ u_coarse, const float* __restrict__ u_fine,
    const float* __restrict__ targets, const float* __restrict__ instance_weights, float loss_scale,
    float* __restrict__ labels_out, float* __restrict__ partials, float* __restrict__ loss_partials) {
    apply_device_schedule(f, c);
    constexpr int kRoundsS = (kRounds + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = wave_in_block();
    const int lane = lane_id();
    const int S = c.num_samples;
    const int N = f.num_instances;
    const int per_wave = wave_lds_floats(S, N) + N + N * kGradStride;
    float* base = lds + wave * per_wave;
    WaveLds l = carve_lds(base, S, N);                                     // l.dcache: [N][64], one round at a time
    float* lam = base + wave_lds_floats(S, N);
    float* G = lam + N;
    for (int idx = lane; idx < N * kGradStride; idx += kWave) G[idx] = 0.0f;
    const bool sorted_input = (u_fine != nullptr) && (c.flags & 1u);
    Shading sh = c.sh;
    const FieldBounds bounds = field_bounds(instances, N, f.inv_t, false, c.flags);
    sh.cull = bounds.margin; sh.reach = bounds.reach; sh.yaw = bounds.yaw;
    sh.mlp_bits = 0u;
    sh.mlp_lds = nullptr;
    const float weight_lane = (lane < N) ? (instance_weights ? instance_weights[lane] : 1.0f) : 0.0f;
    float loss_acc = 0.0f;                                                  // lane n: this wave's BCE sum of instance n
    const int D = 2 * S, num_points = D - 1;
    const int stride = static_cast<int>(gridDim.x) * waves_per_block();
    VSRD_PHASE_CLOCK();
    for (int ray = static_cast<int>(blockIdx.x) * waves_per_block() + wave; ray < c.num_rays; ray += stride) {
        wave_lds_sync();
        VSRD_PHASE(7);
        const long long row = source_row(c, ray);
        const Ray r = load_ray_gathered(c, origins, directions, row);
        const float target = load_target(c, targets, row, lane, N);
        const RayCull rc = cull_ray_setup(instances, N, r.ox, r.oy, r.oz, r.rx, r.ry, r.rz, l.cull, lane);
        stage_ray_samples<kRoundsS>(l, c, S, ray, u_coarse, u_fine, c.out_u_coarse, c.out_u_fine, sorted_input, lane);
        VSRD_PHASE(0);
        // ---- pass 1 ------------------------------------------------------------------------------------
        float w1[kRoundsS];
        render_pass<kRoundsS, false, false>(instances, nullptr, N, sh, r, rc, l.coarse, S, l.dcache, w1, nullptr, nullptr);
        VSRD_PHASE(1);
        if (c.out_coarse_weights != nullptr) {
#pragma unroll
            for (int k = 0; k < kRoundsS; ++k)
                if (k * kWave + lane < S - 1) c.out_coarse_weights[static_cast<size_t>(ray) * (S - 1) + k * kWave + lane] = w1[k];
        }
        float coarse_total = 0.0f;
#pragma unroll
        for (int k = 0; k < kRoundsS; ++k) coarse_total += wave_sum(w1[k]);
        float label = 0.0f;
        bool rendered = false;
        RayAdjoint<kRounds> st;
        if ((c.flags & 2u) && coarse_total == 0.0f) {
            if (c.out_distances != nullptr && lane == 0) c.out_distances[static_cast<size_t>(ray) * D] = __builtin_nanf("");      // sentinel row
        } else {                                                            // (exact miss: labels are exactly 0, adjoint exactly 0)
            rendered = true;
            importance_merge<kRoundsS>(l, S, w1);
            if (c.out_distances != nullptr) {
                float* dst = c.out_distances + static_cast<size_t>(ray) * D;
                for (int idx = lane; idx < D; idx += kWave) dst[idx] = l.merged[idx];
            }
            VSRD_PHASE(2);
            // ---- pass 2 with the adjoint's state kept in registers ------------------------------------
            label = adjoint_forward_sweep<kRounds, false, true>(st, instances, nullptr, N, sh, r, rc, l.merged, num_points, nullptr, l.dcache, lane);
            VSRD_PHASE(3);
        }
        if (labels_out != nullptr && lane < N) labels_out[static_cast<size_t>(ray) * N + lane] = label;
        // ---- silhouette BCE and its gradient (main.py:653-671; torch clamp / binary_cross_entropy backward) -----------------
        const float p = fminf(fmaxf(label, 1.0e-6f), 1.0f - 1.0e-6f);
        const float bce = -(target * logf(p) + (1.0f - target) * logf(1.0f - p));
        loss_acc += (lane < N) ? weight_lane * bce : 0.0f;
        const bool inside_clamp = (label >= 1.0e-6f) && (label <= 1.0f - 1.0e-6f);
        const float lam_lane = (lane < N && inside_clamp) ? weight_lane * loss_scale * (p - target) / fmaxf(p * (1.0f - p), 1.0e-12f) : 0.0f;
        if (!rendered || wave_max(fabsf(lam_lane)) == 0.0f) continue;
        if (lane < N) lam[lane] = lam_lane;
        wave_lds_sync();
        if (sh.yaw) adjoint_label_mix<kRounds, false, true>(st, instances, N, sh.inv_t, num_points, lam, lane, nullptr, l.dcache);
        else adjoint_label_mix<kRounds, false, false>(st, instances, N, sh.inv_t, num_points, lam, lane, nullptr, l.dcache);
        if (!adjoint_reverse_sweep<kRounds>(st, sh, r, num_points, nullptr, nullptr, lane)) continue;
        VSRD_PHASE(4);
        const SeedSink no_sink = {};
        if (sh.yaw) adjoint_phase_b<kRounds, false, true>(st, instances, nullptr, N, f.inv_t, num_points, lam, G, lane, nullptr, no_sink);
        else adjoint_phase_b<kRounds, false, false>(st, instances, nullptr, N, f.inv_t, num_points, lam, G, lane, nullptr, no_sink);
        VSRD_PHASE(5);
    }
    wave_lds_sync();
    VSRD_PHASE_FLUSH(lane);
    const size_t wave_global = static_cast<size_t>(blockIdx.x) * waves_per_block() + wave;
    float* out = partials + wave_global * (N * kGradStride);
    for (int idx = lane; idx < N * kGradStride; idx += kWave) out[idx] = G[idx];
    const float loss_total = wave_sum(loss_acc);
    if (lane == 0) loss_partials[wave_global] = loss_total * loss_scale;
}

// ---------------------------------------------------------------------------------------------------
// Fused optimisation-step kernel for RESIDUAL fields (BASELINE config 3; scripts/main.py steps >= warm-up): two-pass render +
// silhouette BCE + eikonal term (main.py:679-687, mean over every sample of every ray of (|grad sdf| - 1)^2) + the whole adjoint in
// one launch.  As in render_silhouette_kernel the adjoint runs on the pass-2 state the wave still holds (no second evaluation of
// pass 2, no saved distances / gradients in HBM); as in render_backward_kernel<K, true> the MLP adjoints of a batch of rays run
// afterwards, instance-major (adjoint_phase_mlp).  Loss = loss_partials[.][0] + eikonal_ratio * loss_partials[.][1] summed over waves.
// ---------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int residual_step_lds_floats(int num_samples, int num_instances) {
    return (kMlpWbarFloats + kMlpLdsFloats + kMlpBatch * 4 * num_instances + wave_lds_floats(num_samples, num_instances) + num_instances + num_instances * kGradStride + 3) & ~3;
}

#ifndef VSRD_RESIDUAL_WAVES_PER_EU
#define VSRD_RESIDUAL_WAVES_PER_EU 1
#endif
template <int kRounds>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(VSRD_RESIDUAL_WAVES_PER_EU, VSRD_RESIDUAL_WAVES_PER_EU))) void render_residual_step_kernel(
    FieldArgs f, const float* __restrict__ instances, const float* __restrict__ mlp, RenderArgs c,
    const float* __restrict__ origins, const float* __restrict__ directions, const float* __restrict__ u_coarse, const float* __restrict__ u_fine,
    const float* __restrict__ targets, const float* __restrict__ instance_weights, float loss_scale, float eikonal_scale, float eikonal_norm,
    float* __restrict__ labels_out, float* __restrict__ partials, float* __restrict__ mlp_partials,
    float4* __restrict__ residual_cache, float* __restrict__ seed_cache, float* __restrict__ loss_partials) {
    apply_device_schedule(f, c);
    constexpr int kRoundsS = (kRounds + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = wave_in_block();
    const int lane = lane_id();
    const int S = c.num_samples;
    const int N = f.num_instances;
    float* wbar = lds + wave * residual_step_lds_floats(S, N) + kMlpWbarFloats;   // [1617] + transposition scratch (residual.h), 16 B aligned
    unsigned* masks = reinterpret_cast<unsigned*>(wbar + kMlpLdsFloats);  // [kMlpBatch][kRounds][N]
    float* base = wbar + kMlpLdsFloats + kMlpBatch * 4 * N;
    WaveLds l = carve_lds(base, S, N);                                     // l.dcache: [N][64], one round at a time
    float* lam = base + wave_lds_floats(S, N);
    float* G = lam + N;
    for (int idx = lane; idx < N * kGradStride; idx += kWave) G[idx] = 0.0f;
    const size_t wave_global = static_cast<size_t>(blockIdx.x) * waves_per_block() + wave;
    float* my_mlp = mlp_partials + wave_global * (static_cast<size_t>(N) * kMlpWeights);
    float4* rcache = residual_cache + wave_global * jet_wave_float4s(kRounds, N);
    float* seeds = seed_cache + wave_global * (static_cast<size_t>(kMlpBatch) * kRounds * N * kSeedFloats * kWave);
    for (int idx = lane; idx < N * kMlpWeights; idx += kWave) my_mlp[idx] = 0.0f;
    for (int idx = lane; idx < kMlpWeights; idx += kWave) wbar[idx] = 0.0f;
    const bool sorted_input = (u_fine != nullptr) && (c.flags & 1u);
    Shading sh = c.sh;
    sh.mlp_lds = wbar - kMlpWbarFloats;
    const FieldBounds bounds = field_bounds(instances, N, f.inv_t, true, c.flags);
    sh.cull = bounds.margin; sh.reach = bounds.reach; sh.yaw = bounds.yaw;
    sh.mlp_bits = (c.flags & 8u) ? kMlpCentredBit : 0u;
    const float weight_lane = (lane < N) ? (instance_weights ? instance_weights[lane] : 1.0f) : 0.0f;
    float loss_acc = 0.0f, eikonal_acc = 0.0f;
    const int D = 2 * S, num_points = D - 1;
    const int stride = static_cast<int>(gridDim.x) * waves_per_block();
    VSRD_PHASE_CLOCK();
    for (int first = static_cast<int>(blockIdx.x) * waves_per_block() + wave; first < c.num_rays; first += stride * kMlpBatch) {
#pragma unroll 1
        for (int b = 0; b < kMlpBatch; ++b) {
            const int ray = first + b * stride;
            unsigned* ray_masks = masks + b * kRounds * N;
            float* ray_seeds = seeds + static_cast<size_t>(b) * (kRounds * N * kSeedFloats * kWave);
            wave_lds_sync();
            for (int idx = lane; idx < kRounds * N; idx += kWave) ray_masks[idx] = 0u;
            if (ray >= c.num_rays) continue;
            const long long row = source_row(c, ray);
            const Ray r = load_ray_gathered(c, origins, directions, row);
            const float target = load_target(c, targets, row, lane, N);
            const RayCull rc = cull_ray_setup(instances, N, r.ox, r.oy, r.oz, r.rx, r.ry, r.rz, l.cull, lane);
            stage_ray_samples<kRoundsS>(l, c, S, ray, u_coarse, u_fine, nullptr, nullptr, sorted_input, lane);
            VSRD_PHASE(0);
            float w1[kRoundsS];
            render_pass<kRoundsS, false, true>(instances, mlp, N, sh, r, rc, l.coarse, S, l.dcache, w1, nullptr, nullptr);
            VSRD_PHASE(1);
            importance_merge<kRoundsS>(l, S, w1);
            VSRD_PHASE(2);
            RayAdjoint<kRounds> st;
            const float label = adjoint_forward_sweep<kRounds, true, true>(st, instances, mlp, N, sh, r, rc, l.merged, num_points, nullptr, l.dcache, lane, rcache);
            VSRD_PHASE(3);
            if (labels_out != nullptr && lane < N) labels_out[static_cast<size_t>(ray) * N + lane] = label;
            // silhouette BCE and its gradient (as render_silhouette_kernel)
            const float p = fminf(fmaxf(label, 1.0e-6f), 1.0f - 1.0e-6f);
            const float bce = -(target * logf(p) + (1.0f - target) * logf(1.0f - p));
            loss_acc += (lane < N) ? weight_lane * bce : 0.0f;
            const bool inside_clamp = (label >= 1.0e-6f) && (label <= 1.0f - 1.0e-6f);
            const float lam_lane = (lane < N && inside_clamp) ? weight_lane * loss_scale * (p - target) / fmaxf(p * (1.0f - p), 1.0e-12f) : 0.0f;
            if (lane < N) lam[lane] = lam_lane;
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < kRounds; ++k) {
                if (k * kWave >= num_points) continue;
                // eikonal value of this round's samples
                const float norm = fast_sqrt(st.gx[k] * st.gx[k] + st.gy[k] * st.gy[k] + st.gz[k] * st.gz[k]);
                eikonal_acc += (k * kWave + lane < num_points) ? (norm - 1.0f) * (norm - 1.0f) : 0.0f;
            }
            if (sh.yaw) adjoint_label_mix<kRounds, true, true>(st, instances, N, sh.inv_t, num_points, lam, lane, rcache, l.dcache);
            else adjoint_label_mix<kRounds, true, false>(st, instances, N, sh.inv_t, num_points, lam, lane, rcache, l.dcache);
            if (!adjoint_reverse_sweep<kRounds>(st, sh, r, num_points, nullptr, nullptr, lane, eikonal_scale)) continue;
            VSRD_PHASE(4);
            const SeedSink sink = batch_seed_sink(ray_seeds, ray_masks, N);
            if (sh.yaw) adjoint_phase_b<kRounds, true, true>(st, instances, mlp, N, f.inv_t, num_points, lam, G, lane, rcache, sink, sh.cull);
            else adjoint_phase_b<kRounds, true, false>(st, instances, mlp, N, f.inv_t, num_points, lam, G, lane, rcache, sink, sh.cull);
            VSRD_PHASE(5);
        }
        wave_lds_sync();                                                     // masks: written by lanes < kRounds, read by all
        VSRD_PHASE(7);
        adjoint_phase_mlp<kRounds, kMlpBatch>(instances, mlp, N, G, wbar, my_mlp, lane, seeds, masks, sh.mlp_bits);
        VSRD_PHASE(6);
    }
    wave_lds_sync();
    VSRD_PHASE_FLUSH(lane);
    float* out = partials + wave_global * (N * kGradStride);
    for (int idx = lane; idx < N * kGradStride; idx += kWave) out[idx] = G[idx];
    const float loss_total = wave_sum(loss_acc), eikonal_total = wave_sum(eikonal_acc);
    if (lane == 0) { loss_partials[2 * wave_global] = loss_total * loss_scale; loss_partials[2 * wave_global + 1] = eikonal_total * eikonal_norm; }
}

// ---------------------------------------------------------------------------------------------------
// The same step as TWO kernels per chunk of rays (the default for residual fields; VSRD_FLAG_RESIDUAL_SINGLE_KERNEL selects the
// kernel above).  Why: on gfx950 one wave per SIMD issues a VALU instruction only every ~5 cycles, two waves every ~2.7 (tools/micro/
// mfma_overlap.hip; fp32 MFMAs share the datapath and overlap with nothing), and the fused kernel needs 427 registers -- one wave per
// SIMD -- because the MLP adjoint's state (four layers of [LayerNorm -> GELU] state, the weight operands in both orientations, 45
// weight-adjoint accumulators) and the renderer's per-ray state must fit one allocation.  Split at the point where only the seeds
// are live, each half fits 256 registers (two waves per SIMD), and the MLP adjoint can be distributed by INSTANCE over the whole
// launch: weight operands and weight-adjoint accumulators stay in registers over ~32 point sets instead of one batch of 8 rays.
//   residual_step_front_kernel     pass 1, sampling, pass 2, silhouette BCE, eikonal term, reverse sweep, per-instance box adjoint;
//                                  leaves seeds [N][slot][7][64] and 4-bit tile masks [N][slot] (slot = ray of chunk * rounds + round)
//   residual_mlp_adjoint_kernel    work items (instance, <= 64 consecutive slots), fetched dynamically by single-wave workgroups;
//                                  ONE partial row [1617 weight adjoints | 16 box adjoints] per item, so the result does not depend
//                                  on which wave ran which item (reduce_item_rows_kernel sums the rows in a fixed order)
// ---------------------------------------------------------------------------------------------------
constexpr int kItemRowFloats = kMlpWbarFloats + kGradStride;         // 1632 + 16

// (the staging area in front of a wave's partition holds the instance's fp32 weights or its split-bf16 operand image: kMlpStageFloats)
__host__ __device__ constexpr int residual_front_lds_floats(int num_samples, int num_instances) {
    return (kMlpStageFloats + wave_lds_floats(num_samples, num_instances) + num_instances + num_instances * kGradStride + 3) & ~3;
}
// residual_forward's mode bits of a launch (VSRD_FLAG_MLP_WEIGHTS_CENTRED, VSRD_FLAG_MLP_SPLIT_BF16: then `mlp` IS the image table)
__device__ __forceinline__ unsigned residual_mode_bits(unsigned flags) { return ((flags & 8u) ? kMlpCentredBit : 0u) | ((flags & 2048u) ? kMlpSplitBit : 0u); }
__device__ __forceinline__ int residual_weight_stride(unsigned flags) { return (flags & 2048u) ? kMlpImageWords : kMlpWeights; }

// (four rounds -- S in (64, 128], the reference's own S = 100 -- hold twice the per-ray adjoint state and do not fit 256 registers)
// kExport (round 6): the instantiation that writes the step's own samples out (vsrd_render_config::out_*; launched only when a caller asks for them,
// two-round launches only: in the kernel every launch runs, the few extra live values cost 6-12 registers and, in the split-bf16 unit, spills)
template <int kRounds, bool kExport = false>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(kRounds <= 2 ? 2 : 1, 2))) void residual_step_front_kernel(
    FieldArgs f, const float* __restrict__ instances, const float* __restrict__ mlp, RenderArgs c,
    const float* __restrict__ origins, const float* __restrict__ directions, const float* __restrict__ u_coarse, const float* __restrict__ u_fine,
    const float* __restrict__ targets, const float* __restrict__ instance_weights, float loss_scale, float eikonal_scale, float eikonal_norm,
    float* __restrict__ labels_out, float* __restrict__ partials, float4* __restrict__ residual_cache, float* __restrict__ loss_partials,
    float* __restrict__ seed_table, unsigned char* __restrict__ mask_table, long long slots_per_instance, int chunk_base, int chunk_rays, int accumulate) {
    apply_device_schedule(f, c);
    constexpr int kRoundsS = (kRounds + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = wave_in_block();
    const int lane = lane_id();
    const int S = c.num_samples;
    const int N = f.num_instances;
    float* base = lds + wave * residual_front_lds_floats(S, N) + kMlpStageFloats;
    WaveLds l = carve_lds(base, S, N);                                     // l.dcache: [N][64], one round at a time
    float* lam = base + wave_lds_floats(S, N);
    float* G = lam + N;
    const size_t wave_global = static_cast<size_t>(blockIdx.x) * waves_per_block() + wave;
    float* out = partials + wave_global * (N * kGradStride);
    for (int idx = lane; idx < N * kGradStride; idx += kWave) G[idx] = accumulate ? out[idx] : 0.0f;      // chunks of one call add up
    float4* rcache = residual_cache + wave_global * jet_wave_float4s(kRounds, N);
    const bool sorted_input = (u_fine != nullptr) && (c.flags & 1u);
    Shading sh = c.sh;
    sh.mlp_lds = base - kMlpStageFloats;
    const FieldBounds bounds = field_bounds(instances, N, f.inv_t, true, c.flags);
    sh.cull = bounds.margin; sh.reach = bounds.reach; sh.yaw = bounds.yaw;
    sh.mlp_bits = residual_mode_bits(c.flags);
    sh.mlp_stride = residual_weight_stride(c.flags);
    const float weight_lane = (lane < N) ? (instance_weights ? instance_weights[lane] : 1.0f) : 0.0f;
    float loss_acc = 0.0f, eikonal_acc = 0.0f;
    const int D = 2 * S, num_points = D - 1;
    const int stride = static_cast<int>(gridDim.x) * waves_per_block();
    for (int local = static_cast<int>(wave_global); local < chunk_rays; local += stride) {
        const int ray = chunk_base + local;
        wave_lds_sync();
        const long long row = source_row(c, ray);
        const Ray r = load_ray_gathered(c, origins, directions, row);
        const float target = load_target(c, targets, row, lane, N);
        const RayCull rc = cull_ray_setup(instances, N, r.ox, r.oy, r.oz, r.rx, r.ry, r.rz, l.cull, lane);
        stage_ray_samples<kRoundsS>(l, c, S, ray, u_coarse, u_fine, kExport ? c.out_u_coarse : nullptr, kExport ? c.out_u_fine : nullptr, sorted_input, lane);
        float w1[kRoundsS];
        render_pass<kRoundsS, false, true>(instances, mlp, N, sh, r, rc, l.coarse, S, l.dcache, w1, nullptr, nullptr);
        if (kExport && c.out_coarse_weights != nullptr) {                  // vsrd_render_config::out_* (ABI 8: the residual step too): the step's own samples
#pragma unroll
            for (int k = 0; k < kRoundsS; ++k)
                if (k * kWave + lane < S - 1) c.out_coarse_weights[static_cast<size_t>(ray) * (S - 1) + k * kWave + lane] = w1[k];
        }
        importance_merge<kRoundsS>(l, S, w1);
        if (kExport && c.out_distances != nullptr) {
            float* dst = c.out_distances + static_cast<size_t>(ray) * D;
            for (int idx = lane; idx < D; idx += kWave) dst[idx] = l.merged[idx];
        }
        RayAdjoint<kRounds> st;
        const float label = adjoint_forward_sweep<kRounds, true, true>(st, instances, mlp, N, sh, r, rc, l.merged, num_points, nullptr, l.dcache, lane, rcache);
        if (labels_out != nullptr && lane < N) labels_out[static_cast<size_t>(ray) * N + lane] = label;
        // silhouette BCE and its gradient (as render_silhouette_kernel)
        const float p = fminf(fmaxf(label, 1.0e-6f), 1.0f - 1.0e-6f);
        const float bce = -(target * logf(p) + (1.0f - target) * logf(1.0f - p));
        loss_acc += (lane < N) ? weight_lane * bce : 0.0f;
        const bool inside_clamp = (label >= 1.0e-6f) && (label <= 1.0f - 1.0e-6f);
        const float lam_lane = (lane < N && inside_clamp) ? weight_lane * loss_scale * (p - target) / fmaxf(p * (1.0f - p), 1.0e-12f) : 0.0f;
        if (lane < N) lam[lane] = lam_lane;
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            if (k * kWave >= num_points) continue;
            const float norm = fast_sqrt(st.gx[k] * st.gx[k] + st.gy[k] * st.gy[k] + st.gz[k] * st.gz[k]);
            eikonal_acc += (k * kWave + lane < num_points) ? (norm - 1.0f) * (norm - 1.0f) : 0.0f;
        }
        if (sh.yaw) adjoint_label_mix<kRounds, true, true>(st, instances, N, sh.inv_t, num_points, lam, lane, rcache, l.dcache);
        else adjoint_label_mix<kRounds, true, false>(st, instances, N, sh.inv_t, num_points, lam, lane, rcache, l.dcache);
        if (!adjoint_reverse_sweep<kRounds>(st, sh, r, num_points, nullptr, nullptr, lane, eikonal_scale)) continue;      // (masks stay 0)
        const long long slot0 = static_cast<long long>(local) * kRounds;
        const SeedSink sink = {seed_table + slot0 * (kSeedFloats * kWave), slots_per_instance * (kSeedFloats * kWave), static_cast<long long>(kSeedFloats) * kWave,
                               nullptr, mask_table + slot0, slots_per_instance, 1};
        if (sh.yaw) adjoint_phase_b<kRounds, true, true>(st, instances, mlp, N, f.inv_t, num_points, lam, G, lane, rcache, sink, sh.cull);
        else adjoint_phase_b<kRounds, true, false>(st, instances, mlp, N, f.inv_t, num_points, lam, G, lane, rcache, sink, sh.cull);
    }
    wave_lds_sync();
    for (int idx = lane; idx < N * kGradStride; idx += kWave) out[idx] = G[idx];
    const float loss_total = wave_sum(loss_acc), eikonal_total = wave_sum(eikonal_acc);
    if (lane == 0) {
        float* mine = loss_partials + 2 * wave_global;
        mine[0] = (accumulate ? mine[0] : 0.0f) + loss_total * loss_scale;
        mine[1] = (accumulate ? mine[1] : 0.0f) + eikonal_total * eikonal_norm;
    }
}

// vsrd_render_backward for RESIDUAL fields as two kernels per chunk of rays (round 3; what an unchanged main.py reaches through
// hierarchical_volumetric_rendering(...).backward(), scripts/main.py:511-523, 629-687): this front part -- forward sweep at the saved
// distances, reverse sweep on the incoming adjoints (labels, SDF gradients for the eikonal term, weights), per-instance box adjoint --
// leaves the MLP adjoint's seeds in the dense table, and residual_mlp_adjoint_kernel runs them by instance over the whole chunk, exactly
// as in vsrd_render_residual_step.  render_backward_kernel<K, true> kept the r01 structure (seeds in wave-private batches, the MLP
// adjoint in the same kernel: 393 registers, one wave per SIMD); it remains the fallback when the workspace cannot hold a chunk.
__host__ __device__ constexpr int backward_front_lds_floats(int num_distances, int num_instances) {
    return (kMlpWbarFloats + num_distances + num_instances + num_instances * kGradStride + cull_coef_floats(num_instances) + 3) & ~3;
}

template <int kRounds>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_waves_per_eu(kRounds <= 2 ? 2 : 1, 2))) void render_backward_front_kernel(
    FieldArgs f, const float* __restrict__ instances, const float* __restrict__ mlp, RenderArgs c,
    const float* __restrict__ origins, const float* __restrict__ directions, const float* __restrict__ distances, int num_distances,
    const float* __restrict__ grad_labels, const float* __restrict__ grad_gradients, const float* __restrict__ grad_weights,
    float* __restrict__ partials, float4* __restrict__ residual_cache, float* __restrict__ seed_table, unsigned char* __restrict__ mask_table,
    long long slots_per_instance, int chunk_base, int chunk_rays, int accumulate) {
    apply_device_schedule(f, c);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = wave_in_block();
    const int lane = lane_id();
    const int N = f.num_instances;
    const int num_points = num_distances - 1;
    float* mine = lds + wave * backward_front_lds_floats(num_distances, N);
    float* dist = mine + kMlpWbarFloats;
    float* lam = dist + num_distances;
    float* G = lam + N;
    float* coef = G + N * kGradStride;
    const size_t wave_global = static_cast<size_t>(blockIdx.x) * waves_per_block() + wave;
    float* out = partials + wave_global * (N * kGradStride);
    for (int idx = lane; idx < N * kGradStride; idx += kWave) G[idx] = accumulate ? out[idx] : 0.0f;      // chunks of one call add up
    float4* rcache = residual_cache + wave_global * jet_wave_float4s(kRounds, N);
    Shading sh = c.sh;
    sh.mlp_lds = mine;
    const FieldBounds bounds = field_bounds(instances, N, f.inv_t, true, c.flags);
    sh.cull = bounds.margin; sh.reach = bounds.reach; sh.yaw = bounds.yaw;
    sh.mlp_bits = (c.flags & 8u) ? kMlpCentredBit : 0u;
    const int stride = static_cast<int>(gridDim.x) * waves_per_block();
    for (int local = static_cast<int>(wave_global); local < chunk_rays; local += stride) {
        const int ray = chunk_base + local;
        wave_lds_sync();
        const float lam_lane = (lane < N) ? grad_labels[static_cast<size_t>(ray) * N + lane] : 0.0f;
        if (grad_gradients == nullptr && grad_weights == nullptr && wave_max(fabsf(lam_lane)) == 0.0f) continue;   // nothing flows back (masks stay 0)
        const Ray r = load_ray(origins, directions, c.origin_stride, ray);
        const float* src = distances + static_cast<size_t>(ray) * num_distances;
        for (int idx = lane; idx < num_distances; idx += kWave) dist[idx] = src[idx];
        if (lane < N) lam[lane] = lam_lane;
        const RayCull rc = cull_ray_setup(instances, N, r.ox, r.oy, r.oz, r.rx, r.ry, r.rz, coef, lane);           // (syncs the wave's LDS)
        if (dist[0] != dist[0]) continue;                                    // NaN sentinel: ray skipped by the forward
        RayAdjoint<kRounds> st;
        adjoint_forward_sweep<kRounds, true, false>(st, instances, mlp, N, sh, r, rc, dist, num_points, lam, nullptr, lane, rcache);
        const float* gw_row = grad_weights ? grad_weights + static_cast<size_t>(ray) * num_points : nullptr;
        const float* gg_row = grad_gradients ? grad_gradients + static_cast<size_t>(ray) * num_points * 3 : nullptr;
        if (!adjoint_reverse_sweep<kRounds>(st, sh, r, num_points, gw_row, gg_row, lane)) continue;
        const long long slot0 = static_cast<long long>(local) * kRounds;
        const SeedSink sink = {seed_table + slot0 * (kSeedFloats * kWave), slots_per_instance * (kSeedFloats * kWave), static_cast<long long>(kSeedFloats) * kWave,
                               nullptr, mask_table + slot0, slots_per_instance, 1};
        if (sh.yaw) adjoint_phase_b<kRounds, true, true>(st, instances, mlp, N, f.inv_t, num_points, lam, G, lane, rcache, sink, sh.cull);
        else adjoint_phase_b<kRounds, true, false>(st, instances, mlp, N, f.inv_t, num_points, lam, G, lane, rcache, sink, sh.cull);
    }
    wave_lds_sync();
    for (int idx = lane; idx < N * kGradStride; idx += kWave) out[idx] = G[idx];
}

// The front kernel with a ray split over the two waves of a workgroup, each wave taking half of its rounds.  Used (api.hip:
// plan_residual_step) for four-round launches (S in (64, 128], the reference's own S = 100: half the per-ray adjoint state per wave
// fits 256 registers -- two waves per SIMD -- where residual_step_front_kernel<4> needs 308) and for small two-round launches (1000 rays
// then put two waves on every SIMD instead of one; at one wave per SIMD a VALU instruction issues every ~5 cycles, at two every ~2.7).
// The rounds of a ray are coupled only through three scalars, exchanged in LDS: the transmittance entering a round (a product over
// the rounds before it), the labels (a sum over rounds) and the reverse sweep's suffix sum (over the rounds after it); importance
// sampling stays with wave 0.  Outputs as residual_step_front_kernel (seed slot = ray * kRounds + round; one partial row per wave).
constexpr int kPairWaves = 2;

__host__ __device__ constexpr int residual_pair_wave_floats(int num_instances) {
    return (kMlpStageFloats + num_instances * kWave + cull_coef_floats(num_instances) + num_instances + num_instances * kGradStride + 3) & ~3;
}
__host__ __device__ constexpr int residual_pair_lds_floats(int num_samples, int num_instances) {
    return ((7 * num_samples + 3) & ~3) + 16 + kPairWaves * kWave + kPairWaves * residual_pair_wave_floats(num_instances);
}

// kFrames: the instantiation for a batch of frames (wave.h): the arguments are moved to the workgroup's frame.  The one-frame instantiation
// is the code it always was -- both kernels sit at their register limits, and the pointer arithmetic costs the batched form spills.
template <int kRounds, bool kFrames = false>
__global__ __launch_bounds__(kPairWaves * kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void residual_step_pair_kernel(
    FieldArgs f, const float* __restrict__ instances, const float* __restrict__ mlp, RenderArgs c,
    const float* __restrict__ origins, const float* __restrict__ directions, const float* __restrict__ u_coarse, const float* __restrict__ u_fine,
    const float* __restrict__ targets, const float* __restrict__ instance_weights, float loss_scale, float eikonal_scale, float eikonal_norm,
    float* __restrict__ labels_out, float* __restrict__ partials, float4* __restrict__ residual_cache, float* __restrict__ loss_partials,
    float* __restrict__ seed_table, unsigned char* __restrict__ mask_table, long long slots_per_instance, int chunk_base, int chunk_rays, int accumulate) {
    static_assert(kRounds % kPairWaves == 0, "a wave takes kRounds / 2 rounds of pass 2");
    if constexpr (kFrames) {               // a batch of frames (wave.h): this workgroup's frame is blockIdx.y
        const long long shift = frame_shift(c.frame_stride, blockIdx.y);
        shift_frame(c, shift);
        VSRD_OF_FRAME_NONNULL(instances, shift); VSRD_OF_FRAME_NONNULL(mlp, shift); VSRD_OF_FRAME_NONNULL(origins, shift); VSRD_OF_FRAME_NONNULL(directions, shift);
        VSRD_OF_FRAME(u_coarse, shift); VSRD_OF_FRAME(u_fine, shift); VSRD_OF_FRAME_NONNULL(targets, shift); VSRD_OF_FRAME(instance_weights, shift);
        VSRD_OF_FRAME(labels_out, shift); VSRD_OF_FRAME_NONNULL(partials, shift); VSRD_OF_FRAME_NONNULL(residual_cache, shift); VSRD_OF_FRAME_NONNULL(loss_partials, shift);
        VSRD_OF_FRAME_NONNULL(seed_table, shift); VSRD_OF_FRAME_NONNULL(mask_table, shift);
    }
    apply_device_schedule(f, c);
    constexpr int kRoundsS = (kRounds + 1) / 2;                               // rounds of pass 1 (all of them staged and merged by wave 0)
    constexpr int kMine = kRounds / kPairWaves;                               // rounds of pass 2 per wave
    constexpr int kMineS = (kRoundsS + kPairWaves - 1) / kPairWaves;          // rounds of pass 1 per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = wave_in_block();
    const int lane = lane_id();
    const int S = c.num_samples;
    const int N = f.num_instances;
    // shared by the two waves: the ray's sample arrays, the exchange scalars, the label halves
    WaveLds l = carve_lds(lds, S, 0);
    float* xchg = lds + ((7 * S + 3) & ~3);                                   // [0..1] pass-1 products, [2..3] pass-2 products, [4..5] suffix sums
    float* label_half = xchg + 16;                                            // [2][64]
    float* mine = label_half + kPairWaves * kWave + wave * residual_pair_wave_floats(N);
    l.dcache = mine + kMlpStageFloats;
    l.cull = l.dcache + N * kWave;
    float* lam = l.cull + cull_coef_floats(N);
    float* G = lam + N;
    const size_t wave_global = static_cast<size_t>(blockIdx.x) * kPairWaves + wave;
    float* out = partials + wave_global * (N * kGradStride);
    for (int idx = lane; idx < N * kGradStride; idx += kWave) G[idx] = accumulate ? out[idx] : 0.0f;
    float4* rcache = residual_cache + wave_global * jet_wave_float4s(kRounds, N);
    const bool sorted_input = (u_fine != nullptr) && (c.flags & 1u);
    Shading sh = c.sh;
    sh.mlp_lds = mine;
    const FieldBounds bounds = field_bounds(instances, N, f.inv_t, true, c.flags);
    sh.cull = bounds.margin; sh.reach = bounds.reach; sh.yaw = bounds.yaw;
    sh.mlp_bits = residual_mode_bits(c.flags);
    sh.mlp_stride = residual_weight_stride(c.flags);
    const float weight_lane = (lane < N) ? (instance_weights ? instance_weights[lane] : 1.0f) : 0.0f;
    float loss_acc = 0.0f, eikonal_acc = 0.0f;
    const int D = 2 * S, num_points = D - 1;
    // pass 2: the ray's 16-point tiles are divided evenly (S = 100: 199 points = 13 tiles -> 7 + 6, not the 8 + 5 of a split at point
    // 128); a wave renders [first_point, my_points) and passes my_points where the helpers expect the ray's number of points
    const int split = ((num_points + 31) / 32) * 16;
    const int first_coarse = wave * kMineS * kWave, first_point = wave == 0 ? 0 : split, my_points = wave == 0 ? split : num_points;
    // (the two waves of a ray only meet in LDS: block_lds_barrier() does not wait for a wave's global stores -- the cached jets, the seeds
    // of the ray before -- as __syncthreads() would)
    for (int local = static_cast<int>(blockIdx.x); local < chunk_rays; local += static_cast<int>(gridDim.x)) {
        const int ray = chunk_base + local;
        block_lds_barrier();                                                      // the previous ray's arrays are no longer read
        const long long row = source_row(c, ray);
        const Ray r = load_ray_gathered(c, origins, directions, row);
        const float target = load_target(c, targets, row, lane, N);
        const RayCull rc = cull_ray_setup(instances, N, r.ox, r.oy, r.oz, r.rx, r.ry, r.rz, l.cull, lane);
        if (wave == 0) stage_ray_samples<kRoundsS>(l, c, S, ray, u_coarse, u_fine, nullptr, nullptr, sorted_input, lane);
        block_lds_barrier();
        // ---- pass 1: each wave its coarse rounds; the compositing weights meet in l.fine (free until the merge writes it) ----------
        float w1[kMineS];
        float through = 1.0f;
        render_pass<kMineS, false, true>(instances, mlp, N, sh, r, rc, l.coarse, S, l.dcache, w1, nullptr, nullptr, first_coarse, &through);
        if (lane == 0) xchg[wave] = through;
        block_lds_barrier();
        {
            const float entering = (wave == 0) ? 1.0f : xchg[0];
#pragma unroll
            for (int k = 0; k < kMineS; ++k) {
                const int idx = first_coarse + k * kWave + lane;
                if (idx < S) l.fine[idx] = w1[k] * entering;
            }
        }
        block_lds_barrier();
        if (wave == 0) {
            float weights[kRoundsS];
#pragma unroll
            for (int k = 0; k < kRoundsS; ++k) weights[k] = (k * kWave + lane < S) ? l.fine[k * kWave + lane] : 0.0f;
            wave_lds_sync();
            importance_merge<kRoundsS>(l, S, weights);
        }
        block_lds_barrier();
        // ---- pass 2: each wave its rounds of the merged samples, with the adjoint's state kept in registers ------------------------
        RayAdjoint<kMine> st;
        float label = adjoint_forward_sweep<kMine, true, true>(st, instances, mlp, N, sh, r, rc, l.merged, my_points, nullptr, l.dcache, lane, rcache,
                                                                 first_point, &through);
        if (lane == 0) xchg[2 + wave] = through;
        block_lds_barrier();
        if (wave != 0) {
            const float entering = xchg[2];
#pragma unroll
            for (int k = 0; k < kMine; ++k) { st.trans[k] *= entering; st.sa[k].wgt *= entering; }
            label *= entering;
        }
        label_half[wave * kWave + lane] = label;
        block_lds_barrier();
        label = label_half[lane] + label_half[kWave + lane];
        if (wave == 0 && labels_out != nullptr && lane < N) labels_out[static_cast<size_t>(ray) * N + lane] = label;
        // silhouette BCE and its gradient (as render_silhouette_kernel); the loss is counted by wave 0
        const float p = fminf(fmaxf(label, 1.0e-6f), 1.0f - 1.0e-6f);
        const float bce = -(target * logf(p) + (1.0f - target) * logf(1.0f - p));
        loss_acc += (wave == 0 && lane < N) ? weight_lane * bce : 0.0f;
        const bool inside_clamp = (label >= 1.0e-6f) && (label <= 1.0f - 1.0e-6f);
        const float lam_lane = (lane < N && inside_clamp) ? weight_lane * loss_scale * (p - target) / fmaxf(p * (1.0f - p), 1.0e-12f) : 0.0f;
        if (lane < N) lam[lane] = lam_lane;
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < kMine; ++k) {
            if (first_point + k * kWave >= my_points) continue;
            const float norm = fast_sqrt(st.gx[k] * st.gx[k] + st.gy[k] * st.gy[k] + st.gz[k] * st.gz[k]);
            eikonal_acc += (first_point + k * kWave + lane < my_points) ? (norm - 1.0f) * (norm - 1.0f) : 0.0f;
        }
        if (sh.yaw) adjoint_label_mix<kMine, true, true>(st, instances, N, sh.inv_t, my_points, lam, lane, rcache, l.dcache, first_point);
        else adjoint_label_mix<kMine, true, false>(st, instances, N, sh.inv_t, my_points, lam, lane, rcache, l.dcache, first_point);
        // the reverse sweep's sum over LATER samples: wave 0 needs the total of wave 1's rounds
        float later = 0.0f;
#pragma unroll
        for (int k = kMine - 1; k >= 0; --k) {
            const bool valid = first_point + k * kWave + lane < my_points;
            later += wave_sum(valid ? st.sa[k].lam_z * st.sa[k].wgt : 0.0f);
        }
        if (lane == 0) xchg[4 + wave] = later;
        block_lds_barrier();
        const float suffix = (wave == 0) ? xchg[5] : 0.0f;
        if (adjoint_reverse_sweep<kMine>(st, sh, r, my_points, nullptr, nullptr, lane, eikonal_scale, first_point, suffix)) {      // (else: masks stay 0)
            const long long slot0 = static_cast<long long>(local) * kRounds + wave * kMine;
            const SeedSink sink = {seed_table + slot0 * (kSeedFloats * kWave), slots_per_instance * (kSeedFloats * kWave), static_cast<long long>(kSeedFloats) * kWave,
                                   nullptr, mask_table + slot0, slots_per_instance, 1};
            if (sh.yaw) adjoint_phase_b<kMine, true, true>(st, instances, mlp, N, f.inv_t, my_points, lam, G, lane, rcache, sink, sh.cull, first_point);
            else adjoint_phase_b<kMine, true, false>(st, instances, mlp, N, f.inv_t, my_points, lam, G, lane, rcache, sink, sh.cull, first_point);
        }
    }
    wave_lds_sync();
    for (int idx = lane; idx < N * kGradStride; idx += kWave) out[idx] = G[idx];
    const float loss_total = wave_sum(loss_acc), eikonal_total = wave_sum(eikonal_acc);
    if (lane == 0) {
        float* row = loss_partials + 2 * wave_global;
        row[0] = (accumulate ? row[0] : 0.0f) + loss_total * loss_scale;
        row[1] = (accumulate ? row[1] : 0.0f) + eikonal_total * eikonal_norm;
    }
}

// The same split for BOX-ONLY fields: render_silhouette_kernel's step with a ray's rounds divided over the two waves of a workgroup.
// For the reference's own launches -- 1000 importance-sampled rays per step (main.py:620-651) -- one wave per ray leaves one wave on
// each SIMD, where a VALU instruction issues every ~5 cycles instead of every ~2.7 and the launch is pure latency (53 us); two waves
// per ray put two on every SIMD with half the rounds each.  Same helpers, same exchange through LDS as above; one partial row and
// one loss partial per wave, as render_silhouette_kernel writes them.
__host__ __device__ constexpr int split_wave_floats(int num_instances) {
    return (num_instances * kWave + cull_coef_floats(num_instances) + num_instances + num_instances * kGradStride + 3) & ~3;
}
__host__ __device__ constexpr int split_lds_floats(int num_samples, int num_instances) {
    return ((7 * num_samples + 3) & ~3) + 16 + kPairWaves * kWave + kPairWaves * split_wave_floats(num_instances);
}

template <int kRounds, bool kFrames = false>      // (kFrames: as residual_step_pair_kernel)
__global__ __launch_bounds__(kPairWaves * kWave) void render_silhouette_split_kernel(
    FieldArgs f, const float* __restrict__ instances, RenderArgs c, const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ u_coarse, const float* __restrict__ u_fine,
    const float* __restrict__ targets, const float* __restrict__ instance_weights, float loss_scale,
    float* __restrict__ labels_out, float* __restrict__ partials, float* __restrict__ loss_partials) {
    static_assert(kRounds % kPairWaves == 0, "a wave takes kRounds / 2 rounds of pass 2");
    if constexpr (kFrames) {               // a batch of frames (wave.h): this workgroup's frame is blockIdx.y
        const long long shift = frame_shift(c.frame_stride, blockIdx.y);
        shift_frame(c, shift);
        VSRD_OF_FRAME_NONNULL(instances, shift); VSRD_OF_FRAME_NONNULL(origins, shift); VSRD_OF_FRAME_NONNULL(directions, shift);
        VSRD_OF_FRAME(u_coarse, shift); VSRD_OF_FRAME(u_fine, shift); VSRD_OF_FRAME_NONNULL(targets, shift); VSRD_OF_FRAME(instance_weights, shift);
        VSRD_OF_FRAME(labels_out, shift); VSRD_OF_FRAME_NONNULL(partials, shift); VSRD_OF_FRAME_NONNULL(loss_partials, shift);
    }
    apply_device_schedule(f, c);
    constexpr int kRoundsS = (kRounds + 1) / 2;                               // rounds of pass 1 (all of them staged and merged by wave 0)
    constexpr int kMine = kRounds / kPairWaves;                               // rounds of pass 2 per wave
    constexpr int kMineS = (kRoundsS + kPairWaves - 1) / kPairWaves;          // rounds of pass 1 per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = wave_in_block();
    const int lane = lane_id();
    const int S = c.num_samples;
    const int N = f.num_instances;
    WaveLds l = carve_lds(lds, S, 0);                                         // shared by the two waves: the ray's sample arrays
    float* xchg = lds + ((7 * S + 3) & ~3);                                   // [0..1] pass-1 products, [2..3] pass-2 products, [4..5] suffix sums, [6..7] pass-1 weight sums
    float* label_half = xchg + 16;                                            // [2][64]
    float* mine = label_half + kPairWaves * kWave + wave * split_wave_floats(N);
    l.dcache = mine;
    l.cull = l.dcache + N * kWave;
    float* lam = l.cull + cull_coef_floats(N);
    float* G = lam + N;
    for (int idx = lane; idx < N * kGradStride; idx += kWave) G[idx] = 0.0f;
    const bool sorted_input = (u_fine != nullptr) && (c.flags & 1u);
    Shading sh = c.sh;
    const FieldBounds bounds = field_bounds(instances, N, f.inv_t, false, c.flags);
    sh.cull = bounds.margin; sh.reach = bounds.reach; sh.yaw = bounds.yaw;
    sh.mlp_bits = 0u;
    sh.mlp_lds = nullptr;
    const float weight_lane = (lane < N) ? (instance_weights ? instance_weights[lane] : 1.0f) : 0.0f;
    float loss_acc = 0.0f;
    const int D = 2 * S, num_points = D - 1;
    const int split = ((num_points + 31) / 32) * 16;                          // the ray's 16-point tiles divided evenly, as above
    const int first_coarse = wave * kMineS * kWave, first_point = wave == 0 ? 0 : split, my_points = wave == 0 ? split : num_points;
    for (int ray = static_cast<int>(blockIdx.x); ray < c.num_rays; ray += static_cast<int>(gridDim.x)) {
        block_lds_barrier();                                                  // the previous ray's arrays are no longer read
        const long long row = source_row(c, ray);
        const Ray r = load_ray_gathered(c, origins, directions, row);
        const float target = load_target(c, targets, row, lane, N);
        const RayCull rc = cull_ray_setup(instances, N, r.ox, r.oy, r.oz, r.rx, r.ry, r.rz, l.cull, lane);
        if (wave == 0) stage_ray_samples<kRoundsS>(l, c, S, ray, u_coarse, u_fine, nullptr, nullptr, sorted_input, lane);
        block_lds_barrier();
        // ---- pass 1: each wave its coarse rounds; the compositing weights meet in l.fine (free until the merge writes it) ----------
        float w1[kMineS];
        float through = 1.0f;
        render_pass<kMineS, false, false>(instances, nullptr, N, sh, r, rc, l.coarse, S, l.dcache, w1, nullptr, nullptr, first_coarse, &through);
        float my_total = 0.0f;
#pragma unroll
        for (int k = 0; k < kMineS; ++k) my_total += wave_sum(w1[k]);
        if (lane == 0) { xchg[wave] = through; xchg[6 + wave] = my_total; }
        block_lds_barrier();
        // exact miss (VSRD_FLAG_SKIP_EXACT_MISSES): every coarse weight of the ray is exactly 0 -- labels exactly 0, adjoint exactly 0
        const bool missed = (c.flags & 2u) && (xchg[6] == 0.0f) && (xchg[0] * xchg[7] == 0.0f);
        float label = 0.0f;
        RayAdjoint<kMine> st;
        if (!missed) {                                                        // (uniform over the workgroup: both waves read the same sums)
            const float entering = (wave == 0) ? 1.0f : xchg[0];
#pragma unroll
            for (int k = 0; k < kMineS; ++k) {
                const int idx = first_coarse + k * kWave + lane;
                if (idx < S) l.fine[idx] = w1[k] * entering;
            }
            block_lds_barrier();
            if (wave == 0) {
                float weights[kRoundsS];
#pragma unroll
                for (int k = 0; k < kRoundsS; ++k) weights[k] = (k * kWave + lane < S) ? l.fine[k * kWave + lane] : 0.0f;
                wave_lds_sync();
                importance_merge<kRoundsS>(l, S, weights);
            }
            block_lds_barrier();
            // ---- pass 2: each wave its rounds of the merged samples, with the adjoint's state kept in registers --------------------
            label = adjoint_forward_sweep<kMine, false, true>(st, instances, nullptr, N, sh, r, rc, l.merged, my_points, nullptr, l.dcache, lane, nullptr,
                                                                first_point, &through);
            if (lane == 0) xchg[2 + wave] = through;
            block_lds_barrier();
            if (wave != 0) {
                const float entered = xchg[2];
#pragma unroll
                for (int k = 0; k < kMine; ++k) { st.trans[k] *= entered; st.sa[k].wgt *= entered; }
                label *= entered;
            }
            label_half[wave * kWave + lane] = label;
            block_lds_barrier();
            label = label_half[lane] + label_half[kWave + lane];
        }
        if (wave == 0 && labels_out != nullptr && lane < N) labels_out[static_cast<size_t>(ray) * N + lane] = label;
        // silhouette BCE and its gradient (as render_silhouette_kernel); the loss is counted by wave 0
        const float p = fminf(fmaxf(label, 1.0e-6f), 1.0f - 1.0e-6f);
        const float bce = -(target * logf(p) + (1.0f - target) * logf(1.0f - p));
        loss_acc += (wave == 0 && lane < N) ? weight_lane * bce : 0.0f;
        const bool inside_clamp = (label >= 1.0e-6f) && (label <= 1.0f - 1.0e-6f);
        const float lam_lane = (lane < N && inside_clamp) ? weight_lane * loss_scale * (p - target) / fmaxf(p * (1.0f - p), 1.0e-12f) : 0.0f;
        if (missed || wave_max(fabsf(lam_lane)) == 0.0f) continue;           // (uniform over the workgroup: label and target are)
        if (lane < N) lam[lane] = lam_lane;
        wave_lds_sync();
        if (sh.yaw) adjoint_label_mix<kMine, false, true>(st, instances, N, sh.inv_t, my_points, lam, lane, nullptr, l.dcache, first_point);
        else adjoint_label_mix<kMine, false, false>(st, instances, N, sh.inv_t, my_points, lam, lane, nullptr, l.dcache, first_point);
        // the reverse sweep's sum over LATER samples: wave 0 needs the total of wave 1's rounds
        float later = 0.0f;
#pragma unroll
        for (int k = kMine - 1; k >= 0; --k) {
            const bool valid = first_point + k * kWave + lane < my_points;
            later += wave_sum(valid ? st.sa[k].lam_z * st.sa[k].wgt : 0.0f);
        }
        if (lane == 0) xchg[4 + wave] = later;
        block_lds_barrier();
        const float suffix = (wave == 0) ? xchg[5] : 0.0f;
        if (adjoint_reverse_sweep<kMine>(st, sh, r, my_points, nullptr, nullptr, lane, 0.0f, first_point, suffix)) {
            const SeedSink no_sink = {};
            if (sh.yaw) adjoint_phase_b<kMine, false, true>(st, instances, nullptr, N, f.inv_t, my_points, lam, G, lane, nullptr, no_sink, sh.cull, first_point);
            else adjoint_phase_b<kMine, false, false>(st, instances, nullptr, N, f.inv_t, my_points, lam, G, lane, nullptr, no_sink, sh.cull, first_point);
        }
    }
    wave_lds_sync();
    const size_t wave_global = static_cast<size_t>(blockIdx.x) * kPairWaves + wave;
    float* out = partials + wave_global * (N * kGradStride);
    for (int idx = lane; idx < N * kGradStride; idx += kWave) out[idx] = G[idx];
    const float loss_total = wave_sum(loss_acc);
    if (lane == 0) loss_partials[wave_global] = loss_total * loss_scale;
}

// One wave = one workgroup (its own staged weights and transposition scratch: 19.3 KB of LDS, eight workgroups per CU), so waves
// never wait for each other: the active fraction of an item varies from 0 to 1.  Work item = (instance, kSlotsPerItem consecutive
// slots), fetched with one global atomic; ONE partial row per item and a flag whether it holds anything.
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void residual_mlp_adjoint_kernel(
    const float* __restrict__ instances, const float* __restrict__ mlp, int N, unsigned mlp_bits, const float* __restrict__ seed_table,
    const unsigned char* __restrict__ mask_table, long long slots_per_instance, long long used_slots, int items_per_instance, int slots_per_item,
    unsigned* __restrict__ next_item, float* __restrict__ item_rows, unsigned char* __restrict__ item_flags, long long frame_stride) {
    {                                      // a batch of frames (wave.h): blockIdx.y's workgroups fetch the items of frame blockIdx.y, from its own counter
        const long long shift = frame_shift(frame_stride, blockIdx.y);          // (0 for one frame; straight-line code: hipcc 7.2's register allocator
        VSRD_OF_FRAME_NONNULL(instances, shift); VSRD_OF_FRAME_NONNULL(mlp, shift); VSRD_OF_FRAME_NONNULL(seed_table, shift);      //  crashed on the branchy form)
        VSRD_OF_FRAME_NONNULL(mask_table, shift); VSRD_OF_FRAME_NONNULL(next_item, shift); VSRD_OF_FRAME_NONNULL(item_rows, shift);
        VSRD_OF_FRAME_NONNULL(item_flags, shift);
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = lane_id();
    const LdsFloats staged = (LdsFloats)lds;                                   // [1632] this item's instance weights, centred
    const LdsFloats scratch = staged + kMlpWbarFloats;                         // kMlpStashTiles tiles
    const LdsWeights wt = {staged, lane >> 4, lane & 15};
    const int num_items = N * items_per_instance;
    int staged_instance = -1;
    while (true) {
        int item = 0;
        if (lane == 0) item = static_cast<int>(atomicAdd(next_item, 1u));
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= num_items) break;
        const int i = item / items_per_instance;
        const long long first = static_cast<long long>(item - i * items_per_instance) * slots_per_item;
        const unsigned char* item_counts = mask_table + static_cast<long long>(i) * slots_per_instance + first;
        // Lane k looks at slot k of the item (slots_per_item <= 64): the number of leading seed columns that matter (adjoint_phase_b).
        // Round 4: the points of ALL the item's slots are taken as one stream and cut into 16-point tiles -- a slot of its own ends in a
        // partly filled tile (half a tile per (ray, round, instance) on average: a fifth of the tiles of pass 2, tests/tile_statistics.py),
        // the stream only once.  Element e of the stream = column e - begin_k of slot k, k = the first slot whose inclusive count exceeds e.
        const int my_count = (lane < slots_per_item && first + lane < used_slots) ? static_cast<int>(item_counts[lane]) : 0;
        const int inclusive = static_cast<int>(wave_inclusive_sum(static_cast<float>(my_count)));        // (exact: at most 64 x 64)
        const int total = __builtin_amdgcn_readlane(inclusive, kWave - 1);
        if (total == 0) {
            if (lane == 0) item_flags[item] = 0;
            continue;
        }
        if (i != staged_instance) {
            wave_lds_sync();
            stage_centred_weights_wave(staged, mlp + static_cast<size_t>(i) * kMlpWeights, mlp_bits != 0u, lane);
            staged_instance = i;
            wave_lds_sync();
        }
        const Instance in = load_instance(instances, i);
        MlpAdjoint s;
        s.clear();
        float at0 = 0, at1 = 0, at2 = 0;
        float r00 = 0, r01 = 0, r02 = 0, r10 = 0, r11 = 0, r12 = 0, r20 = 0, r21 = 0, r22 = 0;
        const int begin = inclusive - my_count;
        const float* item_seeds = seed_table + (static_cast<long long>(i) * slots_per_instance + first) * (kSeedFloats * kWave);
#pragma unroll 1
        for (int base = 0; base < total; base += kWave) {
            const int e = base + lane;
            const bool valid = e < total;
            int k = 0;                                                         // number of slots whose inclusive count is <= e (binary lifting over the lanes)
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const int probe = __builtin_amdgcn_ds_bpermute((k + step - 1) << 2, inclusive);
                k += (probe <= e) ? step : 0;
            }
            const int column = e - __builtin_amdgcn_ds_bpermute(k << 2, begin);
            const float* src = item_seeds + (valid ? static_cast<long long>(k) * (kSeedFloats * kWave) + column : 0);
            // lanes beyond the stream's end: zero seeds (the adjoint is linear in them) at the origin
            const float px = valid ? src[0 * kWave] : 0.0f, py = valid ? src[1 * kWave] : 0.0f, pz = valid ? src[2 * kWave] : 0.0f;
            const float res_bar = valid ? src[3 * kWave] : 0.0f;
            const float glx = valid ? src[4 * kWave] : 0.0f, gly = valid ? src[5 * kWave] : 0.0f, glz = valid ? src[6 * kWave] : 0.0f;
            const SeedOffset rel = seed_offset(in, px, py, pz);               // (0 at the origin: lanes beyond the stream's end)
            const float relx = rel.x, rely = rel.y, relz = rel.z;
            const unsigned rows = tiles_of_count(min(total - base, kWave));
            const ResidualAdjoint ra = mlp_adjoint_points<true>(s, wt, px, py, pz, res_bar, glx, gly, glz, scratch, lane, rows);
            r00 += relx * ra.px; r01 += relx * ra.py; r02 += relx * ra.pz;
            r10 += rely * ra.px; r11 += rely * ra.py; r12 += rely * ra.pz;
            r20 += relz * ra.px; r21 += relz * ra.py; r22 += relz * ra.pz;
            at0 -= in.r00 * ra.px + in.r01 * ra.py + in.r02 * ra.pz;
            at1 -= in.r10 * ra.px + in.r11 * ra.py + in.r12 * ra.pz;
            at2 -= in.r20 * ra.px + in.r21 * ra.py + in.r22 * ra.pz;
        }
        float* row = item_rows + static_cast<size_t>(item) * kItemRowFloats;
        mlp_adjoint_flush<true>(s, row, lane);
        const float packed[16] = {at0, at1, at2, r00, r01, r02, r10, r11, r12, r20, r21, r22, 0.0f, 0.0f, 0.0f, 0.0f};
        const float mine = wave_reduce16_scatter(packed, lane);
        if (lane < kGradStride) row[kMlpWbarFloats + lane] = mine;
        if (lane == 0) item_flags[item] = 1;
    }
}

#ifdef VSRD_SPLIT_BF16      // (csrc/split_front.hip only)
constexpr int kMlpSplitScratchTiles = 4;
// residual_mlp_adjoint_kernel on the split-bf16 products (VSRD_FLAG_MLP_SPLIT_BF16; residual.h: mlp_adjoint_points_split): the instance's
// operand IMAGE is staged instead of its weights (9.4 KB) and the scratch is four tiles (the encoder features are recomputed, not
// stashed): 14.5 KB of LDS per single-wave workgroup, eight per CU as before.  Same work items, same rows, same reductions.
// One wave = one workgroup (its own staged image and transposition scratch), so waves
// never wait for each other: the active fraction of an item varies from 0 to 1.  Work item = (instance, kSlotsPerItem consecutive
// slots), fetched with one global atomic; ONE partial row per item and a flag whether it holds anything.
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void residual_mlp_adjoint_split_kernel(
    const float* __restrict__ instances, const float* __restrict__ mlp, int N, unsigned mlp_bits, const float* __restrict__ seed_table,
    const unsigned char* __restrict__ mask_table, long long slots_per_instance, long long used_slots, int items_per_instance, int slots_per_item,
    unsigned* __restrict__ next_item, float* __restrict__ item_rows, unsigned char* __restrict__ item_flags, long long frame_stride) {
    {                                      // a batch of frames (wave.h): blockIdx.y's workgroups fetch the items of frame blockIdx.y, from its own counter
        const long long shift = frame_shift(frame_stride, blockIdx.y);          // (0 for one frame; straight-line code: hipcc 7.2's register allocator
        VSRD_OF_FRAME_NONNULL(instances, shift); VSRD_OF_FRAME_NONNULL(mlp, shift); VSRD_OF_FRAME_NONNULL(seed_table, shift);      //  crashed on the branchy form)
        VSRD_OF_FRAME_NONNULL(mask_table, shift); VSRD_OF_FRAME_NONNULL(next_item, shift); VSRD_OF_FRAME_NONNULL(item_rows, shift);
        VSRD_OF_FRAME_NONNULL(item_flags, shift);
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = lane_id();
    const LdsWords staged = (LdsWords)lds;                                     // [kMlpImageWords] this item's instance: the operand image
    const LdsFloats scratch = (LdsFloats)lds + kMlpImageWords;                 // kMlpSplitScratchTiles tiles
    const SplitWeights wt = {staged, lane, lane >> 4};
    const int num_items = N * items_per_instance;
    int staged_instance = -1;
    while (true) {
        int item = 0;
        if (lane == 0) item = static_cast<int>(atomicAdd(next_item, 1u));
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= num_items) break;
        const int i = item / items_per_instance;
        const long long first = static_cast<long long>(item - i * items_per_instance) * slots_per_item;
        const unsigned char* item_counts = mask_table + static_cast<long long>(i) * slots_per_instance + first;
        // Lane k looks at slot k of the item (slots_per_item <= 64): the number of leading seed columns that matter (adjoint_phase_b).
        // Round 4: the points of ALL the item's slots are taken as one stream and cut into 16-point tiles -- a slot of its own ends in a
        // partly filled tile (half a tile per (ray, round, instance) on average: a fifth of the tiles of pass 2, tests/tile_statistics.py),
        // the stream only once.  Element e of the stream = column e - begin_k of slot k, k = the first slot whose inclusive count exceeds e.
        const int my_count = (lane < slots_per_item && first + lane < used_slots) ? static_cast<int>(item_counts[lane]) : 0;
        const int inclusive = static_cast<int>(wave_inclusive_sum(static_cast<float>(my_count)));        // (exact: at most 64 x 64)
        const int total = __builtin_amdgcn_readlane(inclusive, kWave - 1);
        if (total == 0) {
            if (lane == 0) item_flags[item] = 0;
            continue;
        }
        if (i != staged_instance) {
            wave_lds_sync();
            stage_image_wave(staged, mlp + static_cast<size_t>(i) * kMlpImageWords, lane);      // (`mlp`: the image table of pack_mlp_images_kernel)
            staged_instance = i;
            wave_lds_sync();
        }
        const Instance in = load_instance(instances, i);
        MlpAdjoint s;
        s.clear();
        float at0 = 0, at1 = 0, at2 = 0;
        float r00 = 0, r01 = 0, r02 = 0, r10 = 0, r11 = 0, r12 = 0, r20 = 0, r21 = 0, r22 = 0;
        const int begin = inclusive - my_count;
        const float* item_seeds = seed_table + (static_cast<long long>(i) * slots_per_instance + first) * (kSeedFloats * kWave);
#pragma unroll 1
        for (int base = 0; base < total; base += kWave) {
            const int e = base + lane;
            const bool valid = e < total;
            int k = 0;                                                         // number of slots whose inclusive count is <= e (binary lifting over the lanes)
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const int probe = __builtin_amdgcn_ds_bpermute((k + step - 1) << 2, inclusive);
                k += (probe <= e) ? step : 0;
            }
            const int column = e - __builtin_amdgcn_ds_bpermute(k << 2, begin);
            const float* src = item_seeds + (valid ? static_cast<long long>(k) * (kSeedFloats * kWave) + column : 0);
            // lanes beyond the stream's end: zero seeds (the adjoint is linear in them) at the origin
            const float px = valid ? src[0 * kWave] : 0.0f, py = valid ? src[1 * kWave] : 0.0f, pz = valid ? src[2 * kWave] : 0.0f;
            const float res_bar = valid ? src[3 * kWave] : 0.0f;
            const float glx = valid ? src[4 * kWave] : 0.0f, gly = valid ? src[5 * kWave] : 0.0f, glz = valid ? src[6 * kWave] : 0.0f;
            const SeedOffset rel = seed_offset(in, px, py, pz);               // (0 at the origin: lanes beyond the stream's end)
            const float relx = rel.x, rely = rel.y, relz = rel.z;
            const unsigned rows = tiles_of_count(min(total - base, kWave));
            const ResidualAdjoint ra = mlp_adjoint_points_split(s, wt, px, py, pz, res_bar, glx, gly, glz, scratch, lane, rows);
            r00 += relx * ra.px; r01 += relx * ra.py; r02 += relx * ra.pz;
            r10 += rely * ra.px; r11 += rely * ra.py; r12 += rely * ra.pz;
            r20 += relz * ra.px; r21 += relz * ra.py; r22 += relz * ra.pz;
            at0 -= in.r00 * ra.px + in.r01 * ra.py + in.r02 * ra.pz;
            at1 -= in.r10 * ra.px + in.r11 * ra.py + in.r12 * ra.pz;
            at2 -= in.r20 * ra.px + in.r21 * ra.py + in.r22 * ra.pz;
        }
        float* row = item_rows + static_cast<size_t>(item) * kItemRowFloats;
        mlp_adjoint_flush<true>(s, row, lane);
        const float packed[16] = {at0, at1, at2, r00, r01, r02, r10, r11, r12, r20, r21, r22, 0.0f, 0.0f, 0.0f, 0.0f};
        const float mine = wave_reduce16_scatter(packed, lane);
        if (lane < kGradStride) row[kMlpWbarFloats + lane] = mine;
        if (lane == 0) item_flags[item] = 1;
    }
}

#endif

// VSRD_FLAG_MLP_SPLIT_BF16: the operand images of a launch's instances (residual.h: pack_mlp_image), one workgroup per instance, once per
// vsrd_render_residual_step call (N x 9.4 KB: the front kernels stage an instance's image where they staged its 6.5 KB of weights).
#ifdef VSRD_SPLIT_BF16      // (csrc/split_front.hip only)
__global__ __launch_bounds__(256) void pack_mlp_images_kernel(const float* __restrict__ weights, int centred, unsigned* __restrict__ images, long long frame_stride) {
    if (frame_stride != 0) { const long long shift = frame_shift(frame_stride, blockIdx.y); VSRD_OF_FRAME(weights, shift); VSRD_OF_FRAME(images, shift); }
    __shared__ __attribute__((aligned(16))) float staged[kMlpWbarFloats];
    const int n = blockIdx.x;
    stage_centred_weights((LdsFloats)staged, weights + static_cast<size_t>(n) * kMlpWeights, centred != 0, static_cast<int>(threadIdx.x), static_cast<int>(blockDim.x));
    __syncthreads();
    pack_mlp_image((LdsFloats)staged, images + static_cast<size_t>(n) * kMlpImageWords, static_cast<int>(threadIdx.x), static_cast<int>(blockDim.x));
}
#endif

// grad_mlp [N,1617] (+)= sum over the item rows of instance i (fixed order); box_extra [N,16] likewise (the MLP's dL/dp chained into t, R).
// Two deterministic stages: kItemSegments contiguous row ranges per instance are summed in parallel (grid N x 7 x segments), then the
// segment sums in a fixed order.
constexpr int kItemSegments = 32;

__global__ __launch_bounds__(256) void reduce_item_rows_kernel(const float* __restrict__ item_rows, const unsigned char* __restrict__ item_flags,
                                                               int rows_per_instance, float* __restrict__ segment_sums, long long frame_stride) {
    const int i = blockIdx.x, segment = blockIdx.z % kItemSegments;
    if (frame_stride != 0) {               // a batch of frames (wave.h): the grid's z extent is frames x segments
        const long long shift = frame_shift(frame_stride, blockIdx.z / kItemSegments);
        VSRD_OF_FRAME(item_rows, shift); VSRD_OF_FRAME(item_flags, shift); VSRD_OF_FRAME(segment_sums, shift);
    }
    const int idx = blockIdx.y * blockDim.x + threadIdx.x;
    if (idx >= kItemRowFloats) return;
    const int per_segment = (rows_per_instance + kItemSegments - 1) / kItemSegments;
    const int begin = segment * per_segment, end = min(begin + per_segment, rows_per_instance);
    const float* rows = item_rows + static_cast<size_t>(i) * rows_per_instance * kItemRowFloats + idx;
    const unsigned char* flags = item_flags + static_cast<size_t>(i) * rows_per_instance;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int r = begin;
    for (; r + 4 <= end; r += 4) {                                      // (four fixed partial sums: the order never depends on the flags' pattern)
        if (flags[r]) a0 += rows[static_cast<size_t>(r) * kItemRowFloats];
        if (flags[r + 1]) a1 += rows[static_cast<size_t>(r + 1) * kItemRowFloats];
        if (flags[r + 2]) a2 += rows[static_cast<size_t>(r + 2) * kItemRowFloats];
        if (flags[r + 3]) a3 += rows[static_cast<size_t>(r + 3) * kItemRowFloats];
    }
    for (; r < end; ++r)
        if (flags[r]) a0 += rows[static_cast<size_t>(r) * kItemRowFloats];
    segment_sums[(static_cast<size_t>(i) * kItemSegments + segment) * kItemRowFloats + idx] = (a0 + a1) + (a2 + a3);
}

// grad_mlp [N,1617] (+)= the segment sums of instance i; box_extra [N,16] likewise (the MLP's dL/dp chained into t and R).
__global__ __launch_bounds__(256) void reduce_item_segments_kernel(const float* __restrict__ segment_sums, float* __restrict__ grad_mlp,
                                                                   float* __restrict__ box_extra, int accumulate, long long frame_stride) {
    if (frame_stride != 0) { const long long shift = frame_shift(frame_stride, blockIdx.z); VSRD_OF_FRAME(segment_sums, shift); VSRD_OF_FRAME(grad_mlp, shift); VSRD_OF_FRAME(box_extra, shift); }
    const int i = blockIdx.x;
    const int idx = blockIdx.y * blockDim.x + threadIdx.x;
    if (idx >= kItemRowFloats) return;
    float total = 0.0f;
    for (int segment = 0; segment < kItemSegments; ++segment) total += segment_sums[(static_cast<size_t>(i) * kItemSegments + segment) * kItemRowFloats + idx];
    if (idx < kMlpWeights) {
        float* dst = grad_mlp + static_cast<size_t>(i) * kMlpWeights + idx;
        *dst = (accumulate ? *dst : 0.0f) + total;
    } else if (idx >= kMlpWbarFloats) {
        float* dst = box_extra + i * kGradStride + (idx - kMlpWbarFloats);
        *dst = (accumulate ? *dst : 0.0f) + total;
    }
}

// Deterministic second stage: grad[idx] = sum over waves of partials[wave][idx].  A second table (the loss partials of the fused steps:
// rows of `row2` numbers, summed into out2) rides in the same launch: workgroups row .. row + row2 - 1.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partials, int num_waves, int row, float* __restrict__ out,
                                                              const float* __restrict__ extra = nullptr,
                                                              const float* __restrict__ partials2 = nullptr, int row2 = 0, float* __restrict__ out2 = nullptr,
                                                              long long frame_stride = 0) {
    if (frame_stride != 0) {               // a batch of frames (wave.h): frame blockIdx.y
        const long long shift = frame_shift(frame_stride, blockIdx.y);
        VSRD_OF_FRAME(partials, shift); VSRD_OF_FRAME(out, shift); VSRD_OF_FRAME(extra, shift); VSRD_OF_FRAME(partials2, shift); VSRD_OF_FRAME(out2, shift);
    }
    __shared__ float scratch[256 / kWave];
    const bool second = static_cast<int>(blockIdx.x) >= row;
    const int idx = second ? static_cast<int>(blockIdx.x) - row : static_cast<int>(blockIdx.x);
    const float* table = second ? partials2 : partials;
    const int pitch = second ? row2 : row;
    float acc = 0.0f;
    for (int w = threadIdx.x; w < num_waves; w += blockDim.x) acc += table[static_cast<size_t>(w) * pitch + idx];
    acc = wave_sum(acc);
    if (lane_id() == 0) scratch[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.0f;
        for (int k = 0; k < 256 / kWave; ++k) total += scratch[k];
        if (second) out2[idx] = total;
        else out[idx] = total + (extra ? extra[idx] : 0.0f);
    }
}

}  // namespace vsrd
