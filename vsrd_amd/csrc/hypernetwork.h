// The hypernetwork that turns the per-instance embeddings into the weights of the per-instance residual MLPs, forward, backward and
// Adam, as a handful of small launches (native mode, hipGraph replay).
//
// Reference semantics (vsrd/models/fields/hyper_distance_field.py:27-55, 75-77; configs/.../config.json:143-156, 166-215):
//   4 x [weight_norm(Linear 256 -> 256) -> LayerNorm(256, affine) -> exact GELU] -> weight_norm(Linear 256 -> 1617)
//   weight_norm (dim = 0):  W[o, :] = g[o] v[o, :] / |v[o, :]|
//   torch.optim.Adam on every tensor (embeddings 1e-3, hypernetwork 1e-4), ExponentialLR
// With torch these are ~130 element-wise / GEMM / reduction launches of 3-5 us per step for <= 64 rows of 256 numbers (0.4 of the
// 1.56 ms of a residual step at 1000 rays); here 3 launches forwards and 5 backwards (round 2: 6 and 12, one per layer -- but the
// instances only meet in the parameter gradients, so the four hidden blocks run as ONE workgroup per instance in either direction).  The parameters, Adam's moments and step
// counters are the torch module's and torch.optim.Adam's own device tensors, updated in place (state dicts stay what they are).
//
// Mapping: one wave = one output row of a linear at a time, the lane holds 4 of its 256 inputs; the activations of all N instances
// sit in LDS; sums over lanes are DPP wave reductions; everything that crosses workgroups is summed in a fixed order (deterministic).
#pragma once
#include "frame_step.h"

namespace vsrd {

constexpr int kHyperWidth = 256;
constexpr int kHyperThreads = 512;           // the linears: 8 waves = 8 output rows per workgroup
constexpr int kHyperWaves = kHyperThreads / kWave;
constexpr int kHyperNormThreads = 256;       // the LayerNorm backward: one instance per workgroup, one channel per thread
constexpr int kHyperNormSplit = 4;           // ... and four threads per channel for the sum over the final linear's workgroups
constexpr float kHyperNormEps = 1.0e-5f;

struct HyperAdam { float beta1, beta2, epsilon; };

// The erf-form GELU (torch's default, hyper_distance_field.py:30-55) and its derivative through residual.h's normal cdf / pdf pair: erf
// by Abramowitz & Stegun 7.1.26, an APPROXIMATION with |error| <= 1.5e-7 on erf -- one f32 ulp of a cdf near 1, ~17 instructions, but a
// large RELATIVE error of the cdf in the far negative tail (a < -5, where a * cdf(a) < 2e-6 anyway).  Absolute error of the GELU and of
// its derivative over [-8, 8]: < 4e-7 (tests/test_hip_step.py::test_hypernetwork_gelu_error_bound); -DVSRD_LIBM_ERF builds libm's erff
// for parity runs.  libm's erff is ~300 instructions with divergent branches, and the part between two
// linears is ONE wave's work while fifteen wait: 4 erff per lane were 4.5 of the 8 us per layer (tools/hyper_timers.py).
#ifdef VSRD_LIBM_ERF
__device__ __forceinline__ float erf_gelu(float a) { return 0.5f * a * (1.0f + erff(a * 0.70710678118654752f)); }
__device__ __forceinline__ float erf_gelu_derivative(float a) {
    return 0.5f * (1.0f + erff(a * 0.70710678118654752f)) + a * 0.39894228040143268f * expf(-0.5f * a * a);
}
#else
__device__ __forceinline__ float erf_gelu(float a) { return a * gauss(a).cdf; }
__device__ __forceinline__ float erf_gelu_derivative(float a) { const Gauss n = gauss(a); return n.cdf + a * n.pdf; }
#endif

__global__ void gelu_selftest_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    const int i = static_cast<int>(blockIdx.x) * 256 + static_cast<int>(threadIdx.x);
    if (i < n) { out[i] = erf_gelu(x[i]); out[n + i] = erf_gelu_derivative(x[i]); }
}

// torch.optim.Adam's update of one tensor, the per-step scalars computed once per thread.  (The step counters are advanced by
// hyper_finish_kernel, after every reader of the step.)
struct AdamStep {
    float step_size, inv_sqrt_bc2, one_minus_beta1, beta2, epsilon;
    __device__ __forceinline__ AdamStep(const AdamTensors& t, const HyperAdam& a) {
        const float step = *t.step + 1.0f;
        step_size = *t.learning_rate / (1.0f - powf(a.beta1, step));
        inv_sqrt_bc2 = 1.0f / sqrtf(1.0f - powf(a.beta2, step));
        one_minus_beta1 = 1.0f - a.beta1;
        beta2 = a.beta2;
        epsilon = a.epsilon;
    }
    __device__ __forceinline__ void apply(const AdamTensors& t, size_t index, float grad) const {
        apply(t, index, grad, t.exp_avg[index], t.exp_avg_sq[index], t.parameter[index]);
    }
    // (the old moments and the old parameter requested by the caller long before the gradient is known: a load is ~2 us away)
    __device__ __forceinline__ void apply(const AdamTensors& t, size_t index, float grad, float old_m, float old_v, float old_p) const {
        const float m = old_m + (grad - old_m) * one_minus_beta1;
        const float v = old_v * beta2 + (1.0f - beta2) * grad * grad;
        t.exp_avg[index] = m;
        t.exp_avg_sq[index] = v;
        t.parameter[index] = old_p - step_size * m / (sqrtf(v) * inv_sqrt_bc2 + epsilon);
    }
};

// The input activations of a linear for all instances into LDS: x itself (first layer: the embeddings), or GELU(LayerNorm(x) gamma + beta).
// One wave per instance at a time, 4 channels per lane.  `h` [N][256].
__device__ __forceinline__ void stage_hyper_input(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  int num_instances, float* h) {
    const int wave = static_cast<int>(threadIdx.x) >> 6, lane = lane_id();
    for (int n = wave; n < num_instances; n += kHyperWaves) {
        const float4 v = *reinterpret_cast<const float4*>(x + static_cast<size_t>(n) * kHyperWidth + 4 * lane);
        float out[4] = {v.x, v.y, v.z, v.w};
        if (gamma != nullptr) {
            const float mean = wave_sum(out[0] + out[1] + out[2] + out[3]) * (1.0f / kHyperWidth);
            float var = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { out[j] -= mean; var += out[j] * out[j]; }
            const float inv_std = rsqrtf(wave_sum(var) * (1.0f / kHyperWidth) + kHyperNormEps);
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = erf_gelu(out[j] * inv_std * gamma[4 * lane + j] + beta[4 * lane + j]);
        }
        *reinterpret_cast<float4*>(h + n * kHyperWidth + 4 * lane) = make_float4(out[0], out[1], out[2], out[3]);
    }
    __syncthreads();
}

// z[n, o] = g[o] (v[o, :] . h[n, :]) / |v[o, :]| + b[o];  inv_norm[o] = 1 / |v[o, :]| kept for the backward.
__global__ __launch_bounds__(kHyperThreads) void hyper_linear_forward_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ v,
    const float* __restrict__ g, const float* __restrict__ b, int num_rows, int num_instances, float* __restrict__ z, float* __restrict__ inv_norm,
    long long frame_stride) {
    if (frame_stride != 0) {               // a batch of frames (wave.h): frame blockIdx.y, every frame its own hypernetwork
        const long long shift = frame_shift(frame_stride, blockIdx.y);
        VSRD_OF_FRAME(x, shift); VSRD_OF_FRAME(gamma, shift); VSRD_OF_FRAME(beta, shift); VSRD_OF_FRAME(v, shift); VSRD_OF_FRAME(g, shift); VSRD_OF_FRAME(b, shift);
        VSRD_OF_FRAME(z, shift); VSRD_OF_FRAME(inv_norm, shift);
    }
    extern __shared__ __attribute__((aligned(16))) float h[];
    const int wave = static_cast<int>(threadIdx.x) >> 6, lane = lane_id();
    const int o = static_cast<int>(blockIdx.x) * kHyperWaves + wave;
    float4 row = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (o < num_rows) row = *reinterpret_cast<const float4*>(v + static_cast<size_t>(o) * kHyperWidth + 4 * lane);   // in flight during the staging
    stage_hyper_input(x, gamma, beta, num_instances, h);
    if (o >= num_rows) return;
    const float inv = rsqrtf(wave_sum(row.x * row.x + row.y * row.y + row.z * row.z + row.w * row.w));
    const float scale = g[o] * inv, bias = b[o];
    for (int n = 0; n < num_instances; ++n) {
        const float4 a = *reinterpret_cast<const float4*>(h + n * kHyperWidth + 4 * lane);
        const float dot = wave_sum(row.x * a.x + row.y * a.y + row.z * a.z + row.w * a.w);
        if (lane == 0) z[static_cast<size_t>(n) * num_rows + o] = dot * scale + bias;
    }
    if (lane == 0) inv_norm[o] = inv;
}

// The four hidden blocks of ONE instance per workgroup (the instances do not meet before the losses): four dependent launches of the
// kernel above become one.  16 waves hold 16 output rows each (all 16 row loads in flight at once); the LayerNorm + GELU between two
// linears is one wave's work (4 channels per lane).  z[l] [N][256]: the linears' outputs, kept for the
// backward; inv_norm[l] [256] likewise (every workgroup computes the same values; the first one stores them).
constexpr int kHyperChainThreads = 1024;
constexpr int kHyperChainWaves = kHyperChainThreads / kWave;
constexpr int kHyperHidden = 4;

struct HyperHiddenForward {
    const float* v[kHyperHidden]; const float* g[kHyperHidden]; const float* b[kHyperHidden];
    const float* gamma[kHyperHidden - 1]; const float* beta[kHyperHidden - 1];        // the norm BEHIND linear l feeds linear l + 1
    float* z[kHyperHidden]; float* inv_norm[kHyperHidden];
};

// (Frame batches, wave.h: the pointer tables of these kernels are indexed by run-time layer numbers and therefore stay where they are, in
//  the kernel arguments -- a modified copy would live in scratch memory; every pointer is moved to the workgroup's frame where it is read.)
__global__ __launch_bounds__(kHyperChainThreads) void hyper_hidden_forward_kernel(const float* __restrict__ embeddings, HyperHiddenForward net, long long frame_stride) {
    const long long shift = frame_shift(frame_stride, blockIdx.y);      // this workgroup's frame (0 for one frame)
    VSRD_OF_FRAME(embeddings, shift);
    __shared__ __attribute__((aligned(16))) float h[kHyperWidth];
    __shared__ __attribute__((aligned(16))) float out[kHyperWidth];
    const int n = blockIdx.x, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6), lane = lane_id();    // (wave: scalar row addresses)
#ifdef VSRD_PHASE_TIMERS
    const unsigned long long t_start = wall_clock64();
#define VSRD_HYPER_AT(slot, thread) do { if (threadIdx.x == (thread) && n == 0 && l == 0) g_phase_cycles[slot] = wall_clock64() - t_start; } while (0)   /* tools/hyper_timers.py */
#else
#define VSRD_HYPER_AT(slot, thread) do { } while (0)
#endif
    constexpr int kRows = kHyperWidth / kHyperChainWaves;         // 16 rows per wave: o = wave + 16 r
    // A global load is ~2 us away from a lone workgroup, so nothing is requested where it is needed: the rows (and their g, b: lane r
    // holds row r's) of a layer are requested while the layer before finishes, the norms' affines at the start.  The 32 sums over
    // the lanes that a wave owes per layer (16 rows x {v . h, v . v}) are two reduce-scatters (wave.h: ~50 instructions for 16 sums,
    // lane r ends up with row r's) instead of 32 DPP chains of 7: with the chains a layer took 7 us of which 4.5 were the slowest
    // wave's chains (DPP and v_readlane issue every 4.4 cycles, tools/micro/op_rates.hip).
    float4 row[kRows];
    float g_row = 0.0f, b_row = 0.0f;
    auto request = [&](int l) {
        const float* v_l = of_frame(net.v[l], shift);
#pragma unroll
        for (int r = 0; r < kRows; ++r) row[r] = *reinterpret_cast<const float4*>(v_l + static_cast<size_t>(wave + kHyperChainWaves * r) * kHyperWidth + 4 * lane);
        if (lane < kRows) { g_row = of_frame(net.g[l], shift)[wave + kHyperChainWaves * lane]; b_row = of_frame(net.b[l], shift)[wave + kHyperChainWaves * lane]; }
    };
    request(0);
    __shared__ __attribute__((aligned(16))) float affine[kHyperHidden - 1][2][kHyperWidth];      // gamma, beta of the three norms in between
    if (wave < kHyperHidden - 1) {
        *reinterpret_cast<float4*>(&affine[wave][0][4 * lane]) = *reinterpret_cast<const float4*>(of_frame(net.gamma[wave], shift) + 4 * lane);
        *reinterpret_cast<float4*>(&affine[wave][1][4 * lane]) = *reinterpret_cast<const float4*>(of_frame(net.beta[wave], shift) + 4 * lane);
    }
    if (wave == kHyperHidden - 1)
        *reinterpret_cast<float4*>(h + 4 * lane) = *reinterpret_cast<const float4*>(embeddings + static_cast<size_t>(n) * kHyperWidth + 4 * lane);
#pragma unroll 1                                                   // (unrolled, the scheduler overlaps the layers and spills the rows)
    for (int l = 0; l < kHyperHidden; ++l) {
        float inv;                                                 // 1 / |v[o]| of row o = wave + 16 lane: does not need h, so it is computed while wave 0 prepares h
        {
            float squares[kRows];
#pragma unroll
            for (int r = 0; r < kRows; ++r) squares[r] = (row[r].x * row[r].x + row[r].y * row[r].y) + (row[r].z * row[r].z + row[r].w * row[r].w);
            inv = rsqrtf(wave_reduce16_scatter(squares, lane));    // lane r (of every row of 16 lanes): row r's sum
        }
        __builtin_amdgcn_sched_barrier(0);                         // (both reductions in flight at once spill)
        block_lds_barrier();                                       // h of this layer is complete
        VSRD_HYPER_AT(8, 64);
        {
            const float4 a = *reinterpret_cast<const float4*>(h + 4 * lane);
            float dots[kRows];
#pragma unroll
            for (int r = 0; r < kRows; ++r) dots[r] = (row[r].x * a.x + row[r].y * a.y) + (row[r].z * a.z + row[r].w * a.w);
            const float dot = wave_reduce16_scatter(dots, lane);
            VSRD_HYPER_AT(9, 64);
            if (lane < kRows) {
                const int o = wave + kHyperChainWaves * lane;
                out[o] = dot * g_row * inv + b_row;
                if (n == 0) of_frame(net.inv_norm[l], shift)[o] = inv;
            }
            if (l + 1 < kHyperHidden) request(l + 1);
            VSRD_HYPER_AT(10, 64);
        }
        block_lds_barrier();                                       // out is complete; nobody reads h any more
        VSRD_HYPER_AT(11, 0);
        if (wave == 0) {
            const float4 zv = *reinterpret_cast<const float4*>(out + 4 * lane);
            *reinterpret_cast<float4*>(of_frame(net.z[l], shift) + static_cast<size_t>(n) * kHyperWidth + 4 * lane) = zv;
            if (l + 1 < kHyperHidden) {                            // (the norm behind the last hidden linear belongs to the final linear's staging)
                const float4 gv = *reinterpret_cast<const float4*>(&affine[l][0][4 * lane]), bv = *reinterpret_cast<const float4*>(&affine[l][1][4 * lane]);
                float y[4] = {zv.x, zv.y, zv.z, zv.w};
                const float gam[4] = {gv.x, gv.y, gv.z, gv.w}, bet[4] = {bv.x, bv.y, bv.z, bv.w};
                const float mean = wave_sum(y[0] + y[1] + y[2] + y[3]) * (1.0f / kHyperWidth);
                float var = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) { y[j] -= mean; var += y[j] * y[j]; }
                const float inv_std = rsqrtf(wave_sum(var) * (1.0f / kHyperWidth) + kHyperNormEps);
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = erf_gelu(y[j] * inv_std * gam[j] + bet[j]);
                *reinterpret_cast<float4*>(h + 4 * lane) = make_float4(y[0], y[1], y[2], y[3]);
                VSRD_HYPER_AT(12, 0);
            }
        }
    }
}
#undef VSRD_HYPER_AT

// Centre the generated weights for the render kernels (VSRD_FLAG_MLP_WEIGHTS_CENTRED; rendering/renderers.py::_centre_mlp): in the
// four linears of the per-instance MLP that feed a LayerNorm, remove each column's mean over the 16 output channels.
__global__ __launch_bounds__(128) void hyper_centre_kernel(const float* __restrict__ weights, int num_instances, float* __restrict__ centred, long long frame_stride = 0) {
    if (frame_stride != 0) { const long long shift = frame_shift(frame_stride, blockIdx.y); VSRD_OF_FRAME(weights, shift); VSRD_OF_FRAME(centred, shift); }
    const int n = blockIdx.x;
    const float* src = weights + static_cast<size_t>(n) * kMlpWeights;
    float* dst = centred + static_cast<size_t>(n) * kMlpWeights;
    for (int idx = kMlpHead + static_cast<int>(threadIdx.x); idx < kMlpWeights; idx += blockDim.x) dst[idx] = src[idx];      // the head is not centred
    for (int column = threadIdx.x; column < kMlpRow0 + 3 * kMlpRow; column += blockDim.x) {
        const bool first = column < kMlpRow0;
        const int l = first ? 0 : (column - kMlpRow0) / kMlpRow, col = first ? column : (column - kMlpRow0) % kMlpRow;
        const int base = first ? col : kMlpLayer1 + l * kMlpBlock + col, pitch = first ? kMlpRow0 : kMlpRow;
        float mean = 0.0f;
        for (int r = 0; r < kMlpHidden; ++r) mean += src[base + r * pitch];
        mean *= 1.0f / kMlpHidden;
        for (int r = 0; r < kMlpHidden; ++r) dst[base + r * pitch] = src[base + r * pitch] - mean;
    }
}

// Backward of one weight-normed linear for the 8 rows of this workgroup (one per wave), then Adam on those rows:
//   grad_W[o, :] = sum_n gz[n, o] h[n, :],  grad_b[o] = sum_n gz[n, o],
//   grad_g[o] = (grad_W[o, :] . v[o, :]) / |v|,   grad_v[o, :] = g / |v| (grad_W[o, :] - v[o, :] (grad_W . v) / |v|^2)
// and this workgroup's share of the input adjoint  partial_gh[block][n, :] = sum_{o in block} gz[n, o] W[o, :]  (W before the update):
// the waves leave their W rows and gz columns in LDS and all threads sum the 8 rows in a fixed order.  gz is scaled by
// `grad_scale` on the way in.  LDS: h [N][256] | rows [8][256] | zbar [N][8].
// (partial_gh == nullptr: the rows' own update only -- hyper_hidden_update_kernel, whose input adjoints come from hyper_hidden_backward_kernel.)
__device__ __forceinline__ void hyper_rows_backward(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ gz, float grad_scale,
    const float* __restrict__ inv_norm, int num_rows, int num_instances, const AdamTensors& v, const AdamTensors& g, const AdamTensors& b,
    const HyperAdam& adam, float* __restrict__ partial_gh, int block, float* lds) {
    float* h = lds;
    float* rows = lds + num_instances * kHyperWidth;
    float* zbar = rows + kHyperWaves * kHyperWidth;
    const int wave = static_cast<int>(threadIdx.x) >> 6, lane = lane_id();
    const int o = block * kHyperWaves + wave;
    const bool live = o < num_rows;
    // Everything this row will need is requested up front, in one round trip (a global load is ~2 us away and the kernel is a chain of
    // them otherwise): the row, Adam's moments of the row, the N output adjoints of the row (lane n holds instance n's), g, 1 / |v|.
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 row = zero4, old_m = zero4, old_v = zero4;
    float z_bar_lane = 0.0f, inv = 0.0f, g_o = 0.0f, g_m = 0.0f, g_v = 0.0f, b_m = 0.0f, b_v = 0.0f, b_p = 0.0f;
    if (live) {
        const size_t at = static_cast<size_t>(o) * kHyperWidth + 4 * lane;
        row = *reinterpret_cast<const float4*>(v.parameter + at);
        old_m = *reinterpret_cast<const float4*>(v.exp_avg + at);
        old_v = *reinterpret_cast<const float4*>(v.exp_avg_sq + at);
        if (lane < num_instances) z_bar_lane = gz[static_cast<size_t>(lane) * num_rows + o] * grad_scale;
        inv = inv_norm[o];
        g_o = g.parameter[o];
        if (lane == 0) {                                           // (g's and the bias's Adam state: stepped by lane 0 at the end)
            g_m = g.exp_avg[o]; g_v = g.exp_avg_sq[o];
            b_m = b.exp_avg[o]; b_v = b.exp_avg_sq[o]; b_p = b.parameter[o];
        }
    }
    const AdamStep step_v(v, adam), step_g(g, adam), step_b(b, adam);
    stage_hyper_input(x, gamma, beta, num_instances, h);
    float scale = 0.0f;
    if (live) {
        scale = g_o * inv;
        float gw[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gb = 0.0f;
        for (int n = 0; n < num_instances; ++n) {
            const float z_bar = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z_bar_lane), __builtin_amdgcn_readfirstlane(n)));
            const float4 a = *reinterpret_cast<const float4*>(h + n * kHyperWidth + 4 * lane);
            gw[0] += z_bar * a.x; gw[1] += z_bar * a.y; gw[2] += z_bar * a.z; gw[3] += z_bar * a.w;
            gb += z_bar;
            if (lane == 0) zbar[n * kHyperWaves + wave] = z_bar;
        }
        const float dot = wave_sum(gw[0] * row.x + gw[1] * row.y + gw[2] * row.z + gw[3] * row.w);
        const float pull = dot * inv * inv;
        const float rv[4] = {row.x, row.y, row.z, row.w}, mv[4] = {old_m.x, old_m.y, old_m.z, old_m.w}, vv[4] = {old_v.x, old_v.y, old_v.z, old_v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) step_v.apply(v, static_cast<size_t>(o) * kHyperWidth + 4 * lane + j, scale * (gw[j] - rv[j] * pull), mv[j], vv[j], rv[j]);
        if (lane == 0) {
            step_g.apply(g, o, dot * inv, g_m, g_v, g_o);
            step_b.apply(b, o, gb, b_m, b_v, b_p);
        }
    } else if (lane == 0) {
        for (int n = 0; n < num_instances; ++n) zbar[n * kHyperWaves + wave] = 0.0f;
    }
    if (partial_gh == nullptr) return;                             // (uniform over the workgroup)
    *reinterpret_cast<float4*>(rows + wave * kHyperWidth + 4 * lane) = make_float4(scale * row.x, scale * row.y, scale * row.z, scale * row.w);
    __syncthreads();
    const int channel = static_cast<int>(threadIdx.x) & (kHyperWidth - 1);
    float column[kHyperWaves];
#pragma unroll
    for (int w = 0; w < kHyperWaves; ++w) column[w] = rows[w * kHyperWidth + channel];
    float* out = partial_gh + static_cast<size_t>(block) * num_instances * kHyperWidth;
    for (int n = static_cast<int>(threadIdx.x) >> 8; n < num_instances; n += kHyperThreads / kHyperWidth) {
        float acc = 0.0f;
#pragma unroll
        for (int w = 0; w < kHyperWaves; ++w) acc += zbar[n * kHyperWaves + w] * column[w];
        out[n * kHyperWidth + channel] = acc;
    }
}

__global__ __launch_bounds__(kHyperThreads) void hyper_linear_backward_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ gz, float grad_scale,
    const float* __restrict__ inv_norm, int num_rows, int num_instances, AdamTensors v, AdamTensors g, AdamTensors b, HyperAdam adam,
    float* __restrict__ partial_gh, long long frame_stride) {
    if (frame_stride != 0) {
        const long long shift = frame_shift(frame_stride, blockIdx.y);
        VSRD_OF_FRAME(x, shift); VSRD_OF_FRAME(gamma, shift); VSRD_OF_FRAME(beta, shift); VSRD_OF_FRAME(gz, shift); VSRD_OF_FRAME(inv_norm, shift); VSRD_OF_FRAME(partial_gh, shift);
        shift_frame(v, shift); shift_frame(g, shift); shift_frame(b, shift);
    }
    extern __shared__ __attribute__((aligned(16))) float lds[];
    hyper_rows_backward(x, gamma, beta, gz, grad_scale, inv_norm, num_rows, num_instances, v, g, b, adam, partial_gh, static_cast<int>(blockIdx.x), lds);
}

// The four hidden linears' own updates in one launch (32 workgroups of 8 rows each): their output adjoints gz[l] are all known once
// hyper_hidden_backward_kernel has walked the chain.  LDS: h [N][256] | rows [8][256] | zbar [N][8] (as above).
struct HyperHiddenUpdate {
    const float* x[4]; const float* gamma[4]; const float* beta[4]; const float* gz[4]; const float* inv_norm[4];
    AdamTensors v[4], g[4], b[4];
};

__global__ __launch_bounds__(kHyperThreads) void hyper_hidden_update_kernel(HyperHiddenUpdate u, int num_instances, HyperAdam adam, long long frame_stride) {
    const long long shift = frame_shift(frame_stride, blockIdx.y);      // this workgroup's frame (0 for one frame)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kBlocksPerLayer = kHyperWidth / kHyperWaves;
    const int l = static_cast<int>(blockIdx.x) / kBlocksPerLayer, block = static_cast<int>(blockIdx.x) % kBlocksPerLayer;
    AdamTensors v = u.v[l], g = u.g[l], b = u.b[l];
    shift_frame(v, shift); shift_frame(g, shift); shift_frame(b, shift);
    hyper_rows_backward(of_frame(u.x[l], shift), of_frame(u.gamma[l], shift), of_frame(u.beta[l], shift), of_frame(u.gz[l], shift), 1.0f, of_frame(u.inv_norm[l], shift),
                        kHyperWidth, num_instances, v, g, b, adam, nullptr, block, lds);
}

__device__ __forceinline__ float block_sum_256(float value, float* scratch) {       // 4 waves; scratch [4]; fixed order
    const float partial = wave_sum(value);
    __syncthreads();
    if (lane_id() == 0) scratch[threadIdx.x >> 6] = partial;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// Backward of [LayerNorm(affine) -> GELU] that produced the input of the linear above, for one instance per workgroup, one channel
// per thread: sums the workgroups' partial input adjoints, chains through GELU and LayerNorm.
//   a = gamma y + beta, h = gelu(a):  a_bar = h_bar gelu'(a),  y_bar = a_bar gamma,  z_bar = (y_bar - mean(y_bar) - y mean(y_bar y)) / std
// and leaves this instance's share of gamma_bar = a_bar y, beta_bar = a_bar in norm_partials [N][2][256] (hyper_finish_kernel sums
// them at the end of the step).  Used behind the final linear, whose 203 workgroups' shares have to be summed across workgroups.
__global__ __launch_bounds__(kHyperNormThreads * kHyperNormSplit) void hyper_norm_backward_kernel(
    const float* __restrict__ partial_gh, int num_partials, const float* __restrict__ z_prev, int num_instances,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ gz_out, float* __restrict__ norm_partials, long long frame_stride) {
    if (frame_stride != 0) {
        const long long shift = frame_shift(frame_stride, blockIdx.y);
        VSRD_OF_FRAME(partial_gh, shift); VSRD_OF_FRAME(z_prev, shift); VSRD_OF_FRAME(gamma, shift); VSRD_OF_FRAME(beta, shift); VSRD_OF_FRAME(gz_out, shift);
        VSRD_OF_FRAME(norm_partials, shift);
    }
    __shared__ float scratch[4];
    __shared__ float quarter[kHyperNormSplit][kHyperWidth];
    const int n = blockIdx.x, c = static_cast<int>(threadIdx.x) & (kHyperWidth - 1), q = static_cast<int>(threadIdx.x) >> 8;
    const size_t stride = static_cast<size_t>(num_instances) * kHyperWidth;
    const float* src = partial_gh + static_cast<size_t>(n) * kHyperWidth + c;
    const float y_in = z_prev[static_cast<size_t>(n) * kHyperWidth + c], gam = gamma[c], bet = beta[c];      // (requested with the shares: one round trip)
    // 203 shares per channel behind the final linear: four threads per channel sum every fourth one (16 loads in flight each), then
    // thread c adds the four in a fixed order and the other twelve waves leave
    float part = 0.0f;
#pragma unroll 16
    for (int p = q; p < num_partials; p += kHyperNormSplit) part += src[p * stride];
    quarter[q][c] = part;
    __syncthreads();
    if (q != 0) return;
    const float hb = (quarter[0][c] + quarter[1][c]) + (quarter[2][c] + quarter[3][c]);
    float y = y_in;
    const float mean = block_sum_256(y, scratch) * (1.0f / kHyperWidth);
    y -= mean;
    const float inv_std = rsqrtf(block_sum_256(y * y, scratch) * (1.0f / kHyperWidth) + kHyperNormEps);
    y *= inv_std;
    const float a_bar = hb * erf_gelu_derivative(y * gam + bet);
    const float y_bar = a_bar * gam;
    const float m1 = block_sum_256(y_bar, scratch) * (1.0f / kHyperWidth);
    const float m2 = block_sum_256(y_bar * y, scratch) * (1.0f / kHyperWidth);
    gz_out[static_cast<size_t>(n) * kHyperWidth + c] = (y_bar - m1 - y * m2) * inv_std;
    norm_partials[(static_cast<size_t>(n) * 2 + 0) * kHyperWidth + c] = a_bar * y;
    norm_partials[(static_cast<size_t>(n) * 2 + 1) * kHyperWidth + c] = a_bar;
}

// The adjoint chain through the four hidden blocks of ONE instance per workgroup: eight dependent launches (linear backward, norm
// backward, four times) become this one plus hyper_hidden_update_kernel.  Given gz[3] = the adjoint of the last hidden linear's output
// (hyper_norm_backward_kernel behind the final linear), for l = 3 .. 0:
//   a_bar[i] = sum_o gz[l][o] g_l[o] / |v_l[o]| v_l[o, i]      twelve waves sum 22 rows each (4 channels per lane), twelve partial sums
//   l > 0:  through [LayerNorm -> GELU] behind linear l - 1 (one wave, 4 channels per lane; the arithmetic of hyper_norm_backward_kernel)
//           -> gz[l - 1], and this instance's share of that norm's gamma_bar / beta_bar
//   l = 0:  a_bar is the gradient of the embedding row -> embedding_bar [N][256] (stepped by hyper_finish_kernel, after every reader)
struct HyperHiddenBackward {
    const float* v[kHyperHidden]; const float* g[kHyperHidden]; const float* inv_norm[kHyperHidden];
    const float* gamma[kHyperHidden - 1]; const float* beta[kHyperHidden - 1]; const float* z[kHyperHidden - 1];
    float* gz[kHyperHidden];                  // [N][256] each; gz[3] is the input
    float* norm_partials[kHyperHidden - 1];   // [N][2][256] each
    float* embedding_bar;                     // [N][256]
};

__global__ __launch_bounds__(kHyperChainThreads) void hyper_hidden_backward_kernel(HyperHiddenBackward net, long long frame_stride) {
    const long long shift = frame_shift(frame_stride, blockIdx.y);      // this workgroup's frame (0 for one frame)
    constexpr int kWorkers = kHyperChainWaves - kHyperHidden;      // waves 4 .. 15 hold the weights: rows o = (wave - 4) + 12 r, 4 input channels per lane
    constexpr int kRows = (kHyperWidth + kWorkers - 1) / kWorkers; // 22 (the last ones of some waves fall off the end)
    __shared__ __attribute__((aligned(16))) float scaled[kWorkers * kRows];         // gz[l][o] g[o] / |v[o]|, zero behind row 255
    __shared__ __attribute__((aligned(16))) float partial[kWorkers][kHyperWidth];
    const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = lane_id();
    // A global load is ~2 us away from a lone workgroup, so nothing is requested where it is needed: a layer's weights are requested
    // while the layer before finishes, and the one-wave part of step l is the work of wave l (l = 0 .. 3: waves that hold no weights),
    // which has held everything it needs (z, gamma, beta of the norm behind linear l - 1 and that linear's g / |v|) since the start.
    // Two code paths with the same two barriers per layer: in one path the register allocator keeps the weights alive across the erf code.
    if (wave >= kHyperHidden) {                                    // ---- the workers
        const int first = __builtin_amdgcn_readfirstlane(wave) - kHyperHidden;      // (wave-uniform row addresses: scalar base, one lane offset)
        float4 row[kRows];
        auto request = [&](int l) {
            const float* v_l = of_frame(net.v[l], shift);
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const int o = first + kWorkers * r;
                row[r] = o < kHyperWidth ? *reinterpret_cast<const float4*>(v_l + static_cast<size_t>(o) * kHyperWidth + 4 * lane) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        };
        request(kHyperHidden - 1);
#pragma unroll 1
        for (int l = kHyperHidden - 1; l >= 0; --l) {
            block_lds_barrier();                                       // scaled is complete
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const float weight = scaled[first + kWorkers * r];
                acc.x += weight * row[r].x; acc.y += weight * row[r].y; acc.z += weight * row[r].z; acc.w += weight * row[r].w;
            }
            *reinterpret_cast<float4*>(&partial[first][4 * lane]) = acc;
            __builtin_amdgcn_sched_barrier(0);                     // (the next layer's loads must not be hoisted above this layer's sums)
            if (l > 0) request(l - 1);
            block_lds_barrier();                                       // partial is complete; nobody reads scaled any more
        }
        return;
    }
    // ---- waves 0 .. 3
    float4 zv = make_float4(0.0f, 0.0f, 0.0f, 0.0f), gam = zv, bet = zv, weight_scale = zv;
    if (wave >= 1) {                                               // wave l: the norm behind linear k = l - 1
        const int k = wave - 1;
        zv = *reinterpret_cast<const float4*>(of_frame(net.z[k], shift) + static_cast<size_t>(n) * kHyperWidth + 4 * lane);
        gam = *reinterpret_cast<const float4*>(of_frame(net.gamma[k], shift) + 4 * lane);
        bet = *reinterpret_cast<const float4*>(of_frame(net.beta[k], shift) + 4 * lane);
        const float4 g = *reinterpret_cast<const float4*>(of_frame(net.g[k], shift) + 4 * lane), inv = *reinterpret_cast<const float4*>(of_frame(net.inv_norm[k], shift) + 4 * lane);
        weight_scale = make_float4(g.x * inv.x, g.y * inv.y, g.z * inv.z, g.w * inv.w);
    }
    scaled[tid] = of_frame(net.gz[kHyperHidden - 1], shift)[static_cast<size_t>(n) * kHyperWidth + tid] * of_frame(net.g[kHyperHidden - 1], shift)[tid] *
                  of_frame(net.inv_norm[kHyperHidden - 1], shift)[tid];
    if (tid < kWorkers * kRows - kHyperWidth) scaled[kHyperWidth + tid] = 0.0f;
#pragma unroll
    for (int l = kHyperHidden - 1; l >= 0; --l) {
        block_lds_barrier();
        block_lds_barrier();
        if (wave != l) continue;
        float hb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int p = 0; p < kWorkers; ++p) {                       // fixed order
            const float4 share = *reinterpret_cast<const float4*>(&partial[p][4 * lane]);
            hb[0] += share.x; hb[1] += share.y; hb[2] += share.z; hb[3] += share.w;
        }
        if (l == 0) {
            *reinterpret_cast<float4*>(of_frame(net.embedding_bar, shift) + static_cast<size_t>(n) * kHyperWidth + 4 * lane) = make_float4(hb[0], hb[1], hb[2], hb[3]);
            continue;
        }
        const int k = l - 1;                                       // the norm behind linear k
        float y[4] = {zv.x, zv.y, zv.z, zv.w};
        const float gamma[4] = {gam.x, gam.y, gam.z, gam.w}, beta[4] = {bet.x, bet.y, bet.z, bet.w};
        const float next_scale[4] = {weight_scale.x, weight_scale.y, weight_scale.z, weight_scale.w};
        const float mean = wave_sum(y[0] + y[1] + y[2] + y[3]) * (1.0f / kHyperWidth);
        float var = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { y[j] -= mean; var += y[j] * y[j]; }
        const float inv_std = rsqrtf(wave_sum(var) * (1.0f / kHyperWidth) + kHyperNormEps);
        float a_bar[4], y_bar[4], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] *= inv_std;
            a_bar[j] = hb[j] * erf_gelu_derivative(y[j] * gamma[j] + beta[j]);
            y_bar[j] = a_bar[j] * gamma[j];
            s1 += y_bar[j]; s2 += y_bar[j] * y[j];
        }
        const float m1 = wave_sum(s1) * (1.0f / kHyperWidth), m2 = wave_sum(s2) * (1.0f / kHyperWidth);
        float out[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { out[j] = (y_bar[j] - m1 - y[j] * m2) * inv_std; scaled[4 * lane + j] = out[j] * next_scale[j]; }
        *reinterpret_cast<float4*>(of_frame(net.gz[k], shift) + static_cast<size_t>(n) * kHyperWidth + 4 * lane) = make_float4(out[0], out[1], out[2], out[3]);
        float* shares = of_frame(net.norm_partials[k], shift) + static_cast<size_t>(n) * 2 * kHyperWidth;
        *reinterpret_cast<float4*>(shares + 4 * lane) = make_float4(a_bar[0] * y[0], a_bar[1] * y[1], a_bar[2] * y[2], a_bar[3] * y[3]);
        *reinterpret_cast<float4*>(shares + kHyperWidth + 4 * lane) = make_float4(a_bar[0], a_bar[1], a_bar[2], a_bar[3]);
    }
}

// The end of the step, one workgroup, after every other kernel has read what it changes: Adam on the LayerNorm affines (gamma_bar /
// beta_bar = the instances' shares summed in instance order) and on the embeddings, then -- behind a barrier, so that every thread has
// read the counters and the rates -- the step counters of all the tensors stepped in this backward advance and the two learning
// rates decay (ExponentialLR steps after the optimiser).
struct HyperNorms { AdamTensors gamma[4], beta[4]; };
struct HyperStepCounters { float* step[32]; int count; };

__global__ __launch_bounds__(kHyperChainThreads) void hyper_finish_kernel(HyperNorms norms, const float* __restrict__ norm_partials, AdamTensors embeddings,
                                                                          const float* __restrict__ embedding_bar, int num_instances, HyperAdam adam,
                                                                          HyperStepCounters counters, float* embedding_lr, float* hyper_lr, float gamma,
                                                                          long long frame_stride) {
    const long long shift = frame_shift(frame_stride, blockIdx.y);      // a batch of frames (wave.h): one workgroup per frame (0 for one frame)
    VSRD_OF_FRAME(norm_partials, shift); shift_frame(embeddings, shift); VSRD_OF_FRAME(embedding_bar, shift);
    VSRD_OF_FRAME(embedding_lr, shift); VSRD_OF_FRAME(hyper_lr, shift);
    const int tid = threadIdx.x;
    const int k = tid >> 8, c = tid & (kHyperWidth - 1);          // 4 norms x 256 channels
    AdamTensors norm_gamma = norms.gamma[k], norm_beta = norms.beta[k];      // (the table stays in the kernel arguments: k is a run-time index)
    shift_frame(norm_gamma, shift); shift_frame(norm_beta, shift);
    // every operand is requested before anything is computed (one round trip of ~2 us instead of five)
    constexpr int kMaxPerThread = VSRD_MAX_INSTANCES * kHyperWidth / kHyperChainThreads;          // 16 embedding entries per thread at N = 64
    const float counter = (tid < counters.count) ? *of_frame(counters.step[tid], shift) : 0.0f;      // (the table itself stays in the kernel arguments)
    const float old_gamma[3] = {norm_gamma.exp_avg[c], norm_gamma.exp_avg_sq[c], norm_gamma.parameter[c]};
    const float old_beta[3] = {norm_beta.exp_avg[c], norm_beta.exp_avg_sq[c], norm_beta.parameter[c]};
    float e_grad[kMaxPerThread], e_m[kMaxPerThread], e_v[kMaxPerThread], e_p[kMaxPerThread];
#pragma unroll
    for (int it = 0; it < kMaxPerThread; ++it) {
        const int idx = tid + it * kHyperChainThreads;
        const bool in = idx < num_instances * kHyperWidth;
        e_grad[it] = in ? embedding_bar[idx] : 0.0f;
        e_m[it] = in ? embeddings.exp_avg[idx] : 0.0f;
        e_v[it] = in ? embeddings.exp_avg_sq[idx] : 0.0f;
        e_p[it] = in ? embeddings.parameter[idx] : 0.0f;
    }
    const float* src = norm_partials + static_cast<size_t>(k) * num_instances * 2 * kHyperWidth;
    float dgamma = 0.0f, dbeta = 0.0f;
    for (int n = 0; n < num_instances; ++n) {
        dgamma += src[(static_cast<size_t>(n) * 2 + 0) * kHyperWidth + c];
        dbeta += src[(static_cast<size_t>(n) * 2 + 1) * kHyperWidth + c];
    }
    const AdamStep step_gamma(norm_gamma, adam), step_beta(norm_beta, adam), step_embeddings(embeddings, adam);
    step_gamma.apply(norm_gamma, c, dgamma, old_gamma[0], old_gamma[1], old_gamma[2]);
    step_beta.apply(norm_beta, c, dbeta, old_beta[0], old_beta[1], old_beta[2]);
#pragma unroll
    for (int it = 0; it < kMaxPerThread; ++it) {
        const int idx = tid + it * kHyperChainThreads;
        if (idx < num_instances * kHyperWidth) step_embeddings.apply(embeddings, idx, e_grad[it], e_m[it], e_v[it], e_p[it]);
    }
    block_lds_barrier();          // every thread has used the counters and the rates it read (an execution barrier: no need to drain the stores above)
    if (tid < counters.count) *of_frame(counters.step[tid], shift) = counter + 1.0f;
    if (tid == 0) { *embedding_lr *= gamma; *hyper_lr *= gamma; }
}

}  // namespace vsrd
