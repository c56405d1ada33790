// Kernels around the render path: per-pixel ray directions, field evaluation at arbitrary points,
// and the two samplers as stand-alone launches (the API-faithful two-call path).
#pragma once
#include "render_kernels.h"

namespace vsrd {

// vsrd/rendering/utils.py:5-18: dir = normalize(M @ (x, y, 1)), M = inv(E)[:3,:3] @ inv(K).
// One thread per pixel; a wave writes 768 contiguous bytes.
__global__ __launch_bounds__(256) void ray_directions_kernel(const float* __restrict__ inverse_projection, int num_views,
                                                             int height, int width, float* __restrict__ directions) {
    const size_t pixels = static_cast<size_t>(height) * width;
    const size_t total = pixels * num_views;
    for (size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const int view = static_cast<int>(idx / pixels);
        const size_t pixel = idx - static_cast<size_t>(view) * pixels;
        const float py = static_cast<float>(pixel / width);
        const float px = static_cast<float>(pixel % width);
        const float* m = inverse_projection + view * 9;
        const float dx = m[0] * px + m[1] * py + m[2];
        const float dy = m[3] * px + m[4] * py + m[5];
        const float dz = m[6] * px + m[7] * py + m[8];
        const float inv = 1.0f / fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1.0e-12f);
        directions[idx * 3 + 0] = dx * inv;
        directions[idx * 3 + 1] = dy * inv;
        directions[idx * 3 + 2] = dz * inv;
    }
}

// The distance_field closure evaluated at arbitrary points (main.py:477-509) plus its analytic normal.
// One thread per point; the instance loop is still wave-uniform (scalar parameter loads).
template <bool kResidual>
__global__ __launch_bounds__(256) void field_eval_kernel(FieldArgs f, const float* __restrict__ instances, const float* __restrict__ mlp,
                                                         const float* __restrict__ positions, long long num_points,
                                                         float* __restrict__ distances, float* __restrict__ gradients,
                                                         float* __restrict__ labels, int hard_union) {
    // residual fields: this wave's staging area for residual_forward's weight operands (static LDS: 4 waves x 6.5 KB)
    __shared__ float forward_weights_all[kResidual ? (256 / kWave) * kMlpWbarFloats : 1];
    float* forward_weights = forward_weights_all + (kResidual ? wave_in_block() * kMlpWbarFloats : 0);
    // whole waves stay in the loop (the residual MLP is wave-cooperative): tail lanes evaluate the last point and store nothing
    for (long long base = static_cast<long long>(blockIdx.x) * blockDim.x + (threadIdx.x & ~(kWave - 1)); base < num_points;
         base += static_cast<long long>(gridDim.x) * blockDim.x) {
        const bool valid = base + lane_id() < num_points;
        const long long idx = valid ? base + lane_id() : num_points - 1;
        const float x = positions[idx * 3 + 0], y = positions[idx * 3 + 1], z = positions[idx * 3 + 2];
        if (hard_union) {
            float best = 3.0e38f, bx = 0.0f, by = 0.0f, bz = 0.0f;
            for (int i = 0; i < f.num_instances; ++i) {
                const BoxEval e = eval_instance<kResidual>(load_instance(instances, i), kResidual ? mlp + i * kMlpWeights : nullptr, x, y, z, forward_weights);
                if (e.d < best) { best = e.d; bx = e.gwx; by = e.gwy; bz = e.gwz; }   // argmin: first minimum
            }
            if (distances && valid) distances[idx] = best;
            if (gradients && valid) { gradients[idx * 3 + 0] = bx; gradients[idx * 3 + 1] = by; gradients[idx * 3 + 2] = bz; }
            continue;
        }
        UnionSums sums = union_init();
        for (int i = 0; i < f.num_instances; ++i) {
            const BoxEval e = eval_instance<kResidual>(load_instance(instances, i), kResidual ? mlp + i * kMlpWeights : nullptr, x, y, z, forward_weights);
            union_accumulate(sums, e.d, e.gwx, e.gwy, e.gwz, 0.0f, f.inv_t);
        }
        const UnionValue v = union_finish(sums, f.inv_t);
        if (distances && valid) distances[idx] = v.u;
        if (gradients && valid) { gradients[idx * 3 + 0] = v.gx; gradients[idx * 3 + 1] = v.gy; gradients[idx * 3 + 2] = v.gz; }
        if (labels) {
            for (int i = 0; i < f.num_instances; ++i) {
                const BoxEval e = eval_instance<kResidual>(load_instance(instances, i), kResidual ? mlp + i * kMlpWeights : nullptr, x, y, z, forward_weights);
                if (valid) labels[idx * f.num_instances + i] = fast_exp(-(e.d - v.m) * f.inv_t) * v.inv_z;
            }
        }
    }
}

// Adjoint of field_eval_kernel: calling the distance_field closure is differentiable in the reference (autograd through
// scripts/main.py:433-509), e.g. for sphere_tracing(differentiable=True) (renderers.py:59-72) whose Newton step depends on the field
// parameters through sdf(x).  Given dL/d(distances) [P] and (soft union) dL/d(labels) [P,N]:
//   soft:  d_bar_i = u_bar c_i - w_i (lambda_i - sum_j lambda_j w_j) / T,   c_i = w_i (1 - (d_i - u) / T)      (phase B of the renderer's
//   hard:  d_bar_i = u_bar [i = argmin]                                                                          adjoint with g_bar = 0)
// then the box adjoint of d_i (and the residual MLP's) and  x_bar = sum_i d_bar_i grad d_i.  One wave = 64 points, the instance
// loop is wave-uniform; per-wave partial rows + reduce_partials_kernel as in the render adjoint (deterministic).
template <bool kResidual>
__global__ __launch_bounds__(kBlockThreads) void field_eval_backward_kernel(
    FieldArgs f, const float* __restrict__ instances, const float* __restrict__ mlp, const float* __restrict__ positions, long long num_points,
    const float* __restrict__ grad_distances, const float* __restrict__ grad_labels, int hard_union,
    float* __restrict__ grad_positions, float* __restrict__ partials, float* __restrict__ mlp_partials) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float forward_weights_all[kResidual ? kMaxWavesPerBlock * kMlpWbarFloats : 1];      // residual_forward's staged weights
    float* forward_weights = forward_weights_all + (kResidual ? wave_in_block() * kMlpWbarFloats : 0);
    const int wave = wave_in_block();
    const int lane = lane_id();
    const int N = f.num_instances;
    const int per_wave = ((kResidual ? kMlpLdsFloats : 0) + N * kGradStride + 3) & ~3;
    float* wbar = lds + wave * per_wave;                                  // residual only (residual.h)
    float* G = wbar + (kResidual ? kMlpLdsFloats : 0);
    for (int idx = lane; idx < N * kGradStride; idx += kWave) G[idx] = 0.0f;
    const size_t wave_global = static_cast<size_t>(blockIdx.x) * waves_per_block() + wave;
    float* my_mlp = kResidual ? mlp_partials + wave_global * (static_cast<size_t>(N) * kMlpWeights) : nullptr;
    if (kResidual) {
        for (int idx = lane; idx < N * kMlpWeights; idx += kWave) my_mlp[idx] = 0.0f;
        for (int idx = lane; idx < kMlpWeights; idx += kWave) wbar[idx] = 0.0f;
    }
    wave_lds_sync();
    const long long stride = static_cast<long long>(gridDim.x) * waves_per_block() * kWave;
    for (long long base = static_cast<long long>(wave_global) * kWave; base < num_points; base += stride) {
        const bool valid = base + lane < num_points;
        const long long idx = valid ? base + lane : num_points - 1;
        const float x = positions[idx * 3 + 0], y = positions[idx * 3 + 1], z = positions[idx * 3 + 2];
        const float u_bar = (valid && grad_distances) ? grad_distances[idx] : 0.0f;
        // ---- forward quantities of the union ------------------------------------------------------------------------------
        UnionSums sums = union_init();
        float best = 3.0e38f;
        int best_index = 0;
        for (int i = 0; i < N; ++i) {
            const BoxEval e = eval_instance<kResidual>(load_instance(instances, i), kResidual ? mlp + i * kMlpWeights : nullptr, x, y, z, forward_weights);
            const float lambda = (valid && grad_labels && !hard_union) ? grad_labels[idx * N + i] : 0.0f;
            union_accumulate(sums, e.d, e.gwx, e.gwy, e.gwz, lambda, f.inv_t);
            if (e.d < best) { best = e.d; best_index = i; }                   // argmin: first minimum
        }
        const UnionValue v = union_finish(sums, f.inv_t);
        const float lam_z = sums.L * v.inv_z;
        // ---- per instance ------------------------------------------------------------------------------------------------------
        float xb = 0.0f, yb = 0.0f, zb = 0.0f;
        for (int i = 0; i < N; ++i) {
            const Instance in = load_instance(instances, i);
            const BoxEval e = eval_instance<kResidual>(in, kResidual ? mlp + i * kMlpWeights : nullptr, x, y, z, forward_weights);
            float d_bar;
            if (hard_union) {
                d_bar = (i == best_index) ? u_bar : 0.0f;
            } else {
                const float ds = e.d - v.m;
                const float w = fast_exp(-ds * f.inv_t) * v.inv_z;
                const float cc = w * (1.0f - (ds - v.us) * f.inv_t);
                const float lambda = (valid && grad_labels) ? grad_labels[idx * N + i] : 0.0f;
                d_bar = u_bar * cc - f.inv_t * w * (lambda - lam_z);
            }
            xb += d_bar * e.gwx; yb += d_bar * e.gwy; zb += d_bar * e.gwz;
            // box adjoint of d_i with no gradient adjoint: q_bar = d_bar h, p_bar = sign(p) q_bar (+ the residual's)
            const float qbx = d_bar * e.hx, qby = d_bar * e.hy, qbz = d_bar * e.hz;
            float pbx = sign_of(e.px) * qbx, pby = sign_of(e.py) * qby, pbz = sign_of(e.pz) * qbz;
            if (kResidual) {
                const ResidualAdjoint ra = residual_backward(mlp + i * kMlpWeights, e.px, e.py, e.pz, d_bar, 0.0f, 0.0f, 0.0f, wbar, lane, 0xFu);
                pbx += ra.px; pby += ra.py; pbz += ra.pz;
            }
            const float packed[16] = {-(in.r00 * pbx + in.r01 * pby + in.r02 * pbz), -(in.r10 * pbx + in.r11 * pby + in.r12 * pbz),
                                      -(in.r20 * pbx + in.r21 * pby + in.r22 * pbz),
                                      e.relx * pbx, e.relx * pby, e.relx * pbz, e.rely * pbx, e.rely * pby, e.rely * pbz,
                                      e.relz * pbx, e.relz * pby, e.relz * pbz, -qbx, -qby, -qbz, 0.0f};
            const float mine = wave_reduce16_scatter(packed, lane);
            if (lane < kGradStride) G[i * kGradStride + lane] += mine;
            if (kResidual) {
                float* dst = my_mlp + static_cast<size_t>(i) * kMlpWeights;
                wave_lds_sync();
                for (int k = lane; k < kMlpWeights; k += kWave) { dst[k] += wbar[k]; wbar[k] = 0.0f; }
                wave_lds_sync();
            }
        }
        if (grad_positions && valid) { grad_positions[idx * 3 + 0] = xb; grad_positions[idx * 3 + 1] = yb; grad_positions[idx * 3 + 2] = zb; }
    }
    wave_lds_sync();
    float* out = partials + wave_global * (N * kGradStride);
    for (int idx = lane; idx < N * kGradStride; idx += kWave) out[idx] = G[idx];
}

// Union distance only (no normal): what sphere tracing evaluates per step.
template <bool kResidual>
__device__ __forceinline__ float union_distance(const FieldArgs& f, const float* __restrict__ instances, const float* __restrict__ mlp,
                                                float x, float y, float z, int hard_union, float* forward_weights) {
    float m = 3.0e38f, Z = 0.0f, S1 = 0.0f;           // online soft-min (field.h), value part only
    for (int i = 0; i < f.num_instances; ++i) {
        const BoxEval e = eval_instance<kResidual>(load_instance(instances, i), kResidual ? mlp + i * kMlpWeights : nullptr, x, y, z, forward_weights);
        const float d = e.d;
        if (hard_union) { m = fminf(m, d); continue; }
        const bool lower = d < m;
        const float gap = lower ? (m - d) : (d - m);
        const float ex = fast_exp(-gap * f.inv_t);
        const float scale = lower ? ex : 1.0f, w = lower ? 1.0f : ex;
        S1 = scale * (S1 + (lower ? gap : 0.0f) * Z) + w * (lower ? 0.0f : gap);
        Z = scale * Z + w;
        m = lower ? d : m;
    }
    return hard_union ? m : (m + S1 * fast_rcp(Z));
}

// vsrd.rendering.sphere_tracing (renderers.py:21-59, the non-differentiable part), one thread per ray.  The reference
// evaluates every ray until ALL have stopped (a host sync per iteration, :57); a stopped ray no longer moves, so stopping
// each thread on its own condition gives identical positions and masks.
//   origins: origin_stride 3 (per ray) or 0 (shared); foreground [R] (uint8, in/out semantics of `foreground_masks`)
//   bounding_radius <= 0: no bounding sphere.  initialise != 0: start at the sphere entry point (renderers.py:36-43).
template <bool kResidual>
__global__ __launch_bounds__(256) void sphere_trace_kernel(FieldArgs f, const float* __restrict__ instances, const float* __restrict__ mlp,
                                                           const float* __restrict__ origins, int origin_stride,
                                                           const float* __restrict__ directions, const unsigned char* __restrict__ foreground,
                                                           long long num_rays, int num_iterations, float criteria, float bounding_radius,
                                                           int initialise, int hard_union,
                                                           float* __restrict__ positions, unsigned char* __restrict__ converged) {
    // residual fields: this wave's staging area for residual_forward's weight operands (static LDS: 4 waves x 6.5 KB)
    __shared__ float forward_weights_all[kResidual ? (256 / kWave) * kMlpWbarFloats : 1];
    float* forward_weights = forward_weights_all + (kResidual ? wave_in_block() * kMlpWbarFloats : 0);
    for (long long base = static_cast<long long>(blockIdx.x) * blockDim.x + (threadIdx.x & ~(kWave - 1)); base < num_rays;
         base += static_cast<long long>(gridDim.x) * blockDim.x) {                 // whole waves (see field_eval_kernel)
        const bool valid = base + lane_id() < num_rays;
        const long long idx = valid ? base + lane_id() : num_rays - 1;
        const float* o = origins + idx * origin_stride;
        float px = o[0], py = o[1], pz = o[2];
        const float dx = directions[idx * 3 + 0], dy = directions[idx * 3 + 1], dz = directions[idx * 3 + 2];
        bool fg = foreground ? (foreground[idx] != 0) : (isfinite(px) && isfinite(py) && isfinite(pz));
        if (bounding_radius > 0.0f && initialise) {                    // sphere_intersection, renderers.py:9-18
            const float a = dx * dx + dy * dy + dz * dz;
            const float b = dx * px + dy * py + dz * pz;
            const float cc = px * px + py * py + pz * pz - bounding_radius * bounding_radius;
            const float disc = b * b - a * cc;
            const bool hit = disc >= 0.0f;
            const float t = (-b - sqrtf(disc)) / a;
            if (hit) { px += dx * t; py += dy * t; pz += dz * t; }
            fg = fg && hit;
        }
        bool conv = false, done = !valid;
        for (int it = 0; it < num_iterations; ++it) {
            const float sd = union_distance<kResidual>(f, instances, mlp, px, py, pz, hard_union, forward_weights);     // all lanes, stopped or not
            if (!done) {
                if (fg && !conv) { px += dx * sd; py += dy * sd; pz += dz * sd; }
                if (bounding_radius > 0.0f) fg = fg && (sqrtf(px * px + py * py + pz * pz) < bounding_radius);
                conv = fabsf(sd) < criteria;
                done = !fg || conv;
            }
            if (!wave_any(!done)) break;
        }
        if (valid) {
            positions[idx * 3 + 0] = px; positions[idx * 3 + 1] = py; positions[idx * 3 + 2] = pz;
            converged[idx] = conv ? 1 : 0;
        }
    }
}

// SoftRasterizer.make_distance_map + the soft-mask formula (vsrd/transforms/geometric_transforms.py:265-317): distance of every
// pixel (integer centres) to the closed polygon of an instance mask, then sigmoid(+-distance / temperature).  One thread per
// pixel; the loop over polygon sides is wave-uniform (vertices via scalar loads).  The reference broadcasts a [HW, P, 2] tensor
// per instance on the CPU inside the dataset; here it is one launch for all instances of a frame.
__global__ __launch_bounds__(256) void polygon_soft_mask_kernel(const float* __restrict__ polygons, const int* __restrict__ counts,
                                                                int num_polygons, int max_vertices, int height, int width,
                                                                const unsigned char* __restrict__ inside, float temperature,
                                                                float* __restrict__ distance_maps, float* __restrict__ soft_masks) {
    const int b = blockIdx.y;
    const int count = counts[b];
    const float* poly = polygons + static_cast<size_t>(b) * max_vertices * 2;
    const int pixels = height * width;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < pixels; idx += gridDim.x * blockDim.x) {
        const float px = static_cast<float>(idx % width), py = static_cast<float>(idx / width);
        float best = 3.0e38f;
        for (int k = 0; k < count; ++k) {
            const int kn = (k + 1 == count) ? 0 : k + 1;                       // torch.roll(polygons, -1)
            const float ax = poly[2 * k], ay = poly[2 * k + 1];
            const float sx = poly[2 * kn] - ax, sy = poly[2 * kn + 1] - ay;
            const float rx = px - ax, ry = py - ay;
            float ratio = (sx * rx + sy * ry) / (sx * sx + sy * sy + 1.0e-6f);
            ratio = fminf(fmaxf(ratio, 0.0f), 1.0f);
            const float nx = rx - sx * ratio, ny = ry - sy * ratio;
            best = fminf(best, sqrtf(nx * nx + ny * ny));
        }
        const size_t out = static_cast<size_t>(b) * pixels + idx;
        if (distance_maps) distance_maps[out] = best;
        if (soft_masks) {
            const float sdf = inside[out] ? best : -best;
            soft_masks[out] = 1.0f / (1.0f + expf(-sdf / temperature));
        }
    }
}

// renderers.py:191-194 + samplers.py:5-8, one thread per (ray, bin).
__global__ __launch_bounds__(256) void sample_stratified_kernel(RenderArgs c, const float* __restrict__ u_coarse, float* __restrict__ distances) {
    const size_t total = static_cast<size_t>(c.num_rays) * c.num_samples;
    for (size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const int k = static_cast<int>(idx % c.num_samples);
        const float lo = torch_linspace(c.near, c.far, c.num_samples + 1, k);
        const float hi = torch_linspace(c.near, c.far, c.num_samples + 1, k + 1);
        distances[idx] = torch_lerp(lo, hi, u_coarse[idx]);
    }
}

// samplers.py:11-36 + renderers.py:198-210 as its own launch (wave per ray).
template <int kRoundsS>
__global__ __launch_bounds__(kBlockThreads) void sample_importance_kernel(
    RenderArgs c, const float* __restrict__ coarse_distances, const float* __restrict__ coarse_weights,
    const float* __restrict__ u_fine, float* __restrict__ merged, float* __restrict__ fine) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = wave_in_block();
    const int lane = lane_id();
    const int S = c.num_samples;
    const WaveLds l = carve_lds(lds + wave * wave_lds_floats(S, 0), S);
    const bool sorted_input = (c.flags & 1u) != 0;
    const int stride = static_cast<int>(gridDim.x) * waves_per_block();
    for (int ray = static_cast<int>(blockIdx.x) * waves_per_block() + wave; ray < c.num_rays; ray += stride) {
        const size_t row = static_cast<size_t>(ray) * S;
        float w1[kRoundsS];
#pragma unroll
        for (int k = 0; k < kRoundsS; ++k) {
            const int idx = k * kWave + lane;
            w1[k] = 0.0f;
            if (idx < S) {
                l.coarse[idx] = coarse_distances[row + idx];
                (sorted_input ? l.usorted : l.uraw)[idx] = u_fine[row + idx];
                if (idx < S - 1) w1[k] = coarse_weights[static_cast<size_t>(ray) * (S - 1) + idx];
            }
        }
        wave_lds_sync();
        if (!sorted_input) {
            rank_sort<kRoundsS>(l.uraw, l.usorted, S);
            wave_lds_sync();
        }
        importance_merge<kRoundsS>(l, S, w1);
        if (merged != nullptr) {
            float* dst = merged + static_cast<size_t>(ray) * 2 * S;
            for (int idx = lane; idx < 2 * S; idx += kWave) dst[idx] = l.merged[idx];
        }
        if (fine != nullptr) {                                                   // inverse_transform_sampler's own return value
            for (int idx = lane; idx < S; idx += kWave) fine[row + idx] = l.fine[idx];
        }
        wave_lds_sync();
    }
}

// Self-test of the wave primitives (tests/test_hip_wave.py): out[0..63] sum, [64..127] inclusive sum,
// [128..191] inclusive product, [192..255] max, [256..319] reverse, [320..383] shift-up.
__global__ void wave_selftest_kernel(const float* __restrict__ in, float* __restrict__ out) {
    const int lane = lane_id();
    const float v = in[lane];
    out[lane] = wave_sum(v);
    out[64 + lane] = wave_inclusive_sum(v);
    out[128 + lane] = wave_inclusive_product(1.0f + 0.01f * v);
    out[192 + lane] = wave_max(v);
    out[256 + lane] = wave_reverse(v, lane);
    out[320 + lane] = wave_shift_up(v, -7.0f, lane);
    const Philox4 r = philox4x32_10(static_cast<uint32_t>(lane), 1u, 2u, 3u, 0xdeadbeefu, 0x12345678u);
    out[384 + lane] = uniform_from_bits(r.x);
    out[448 + lane] = torch_linspace(0.0f, 100.0f, 65, lane);
    float many[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) many[j] = v * static_cast<float>(j + 1) + static_cast<float>(j);
    out[512 + lane] = wave_reduce16_scatter(many, lane);   // lane l: sum over lanes of many[l & 15]
}

}  // namespace vsrd
