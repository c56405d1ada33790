// The hypernetwork that turns the per-instance embeddings into the weights of the per-instance residual MLPs, forward, backward and
// Adam, as a handful of small launches (native mode, hipGraph replay).
//
// Reference semantics (vsrd/models/fields/hyper_distance_field.py:27-55, 75-77; configs/.../config.json:143-156, 166-215):
//   4 x [weight_norm(Linear 256 -> 256) -> LayerNorm(256, affine) -> exact GELU] -> weight_norm(Linear 256 -> 1617)
//   weight_norm (dim = 0):  W[o, :] = g[o] v[o, :] / |v[o, :]|
//   torch.optim.Adam on every tensor (embeddings 1e-3, hypernetwork 1e-4), ExponentialLR
// With torch these are ~130 element-wise / GEMM / reduction launches of 3-5 us per step for <= 64 rows of 256 numbers (0.4 of the
// 1.56 ms of a residual step at 1000 rays); here 6 launches forwards and 12 backwards.  The parameters, Adam's moments and step
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
constexpr float kHyperNormEps = 1.0e-5f;

struct HyperAdam { float beta1, beta2, epsilon; };

__device__ __forceinline__ float gelu_exact(float a) { return 0.5f * a * (1.0f + erff(a * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_exact_derivative(float a) {
    return 0.5f * (1.0f + erff(a * 0.70710678118654752f)) + a * 0.39894228040143268f * expf(-0.5f * a * a);
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
        const float m = t.exp_avg[index] + (grad - t.exp_avg[index]) * one_minus_beta1;
        const float v = t.exp_avg_sq[index] * beta2 + (1.0f - beta2) * grad * grad;
        t.exp_avg[index] = m;
        t.exp_avg_sq[index] = v;
        t.parameter[index] -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + epsilon);
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
            for (int j = 0; j < 4; ++j) out[j] = gelu_exact(out[j] * inv_std * gamma[4 * lane + j] + beta[4 * lane + j]);
        }
        *reinterpret_cast<float4*>(h + n * kHyperWidth + 4 * lane) = make_float4(out[0], out[1], out[2], out[3]);
    }
    __syncthreads();
}

// z[n, o] = g[o] (v[o, :] . h[n, :]) / |v[o, :]| + b[o];  inv_norm[o] = 1 / |v[o, :]| kept for the backward.
__global__ __launch_bounds__(kHyperThreads) void hyper_linear_forward_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ v,
    const float* __restrict__ g, const float* __restrict__ b, int num_rows, int num_instances, float* __restrict__ z, float* __restrict__ inv_norm) {
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

// Centre the generated weights for the render kernels (VSRD_FLAG_MLP_WEIGHTS_CENTRED; rendering/renderers.py::_centre_mlp): in the
// four linears of the per-instance MLP that feed a LayerNorm, remove each column's mean over the 16 output channels.
__global__ __launch_bounds__(128) void hyper_centre_kernel(const float* __restrict__ weights, int num_instances, float* __restrict__ centred) {
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
__global__ __launch_bounds__(kHyperThreads) void hyper_linear_backward_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ gz, float grad_scale,
    const float* __restrict__ inv_norm, int num_rows, int num_instances, AdamTensors v, AdamTensors g, AdamTensors b, HyperAdam adam,
    float* __restrict__ partial_gh) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* h = lds;
    float* rows = lds + num_instances * kHyperWidth;
    float* zbar = rows + kHyperWaves * kHyperWidth;
    const int wave = static_cast<int>(threadIdx.x) >> 6, lane = lane_id();
    const int o = static_cast<int>(blockIdx.x) * kHyperWaves + wave;
    const bool live = o < num_rows;
    float4 row = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (live) row = *reinterpret_cast<const float4*>(v.parameter + static_cast<size_t>(o) * kHyperWidth + 4 * lane);
    const AdamStep step_v(v, adam), step_g(g, adam), step_b(b, adam);
    stage_hyper_input(x, gamma, beta, num_instances, h);
    float scale = 0.0f;
    if (live) {
        const float inv = inv_norm[o];
        scale = g.parameter[o] * inv;
        float gw[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gb = 0.0f;
        for (int n = 0; n < num_instances; ++n) {
            const float z_bar = gz[static_cast<size_t>(n) * num_rows + o] * grad_scale;
            const float4 a = *reinterpret_cast<const float4*>(h + n * kHyperWidth + 4 * lane);
            gw[0] += z_bar * a.x; gw[1] += z_bar * a.y; gw[2] += z_bar * a.z; gw[3] += z_bar * a.w;
            gb += z_bar;
            if (lane == 0) zbar[n * kHyperWaves + wave] = z_bar;
        }
        const float dot = wave_sum(gw[0] * row.x + gw[1] * row.y + gw[2] * row.z + gw[3] * row.w);
        const float pull = dot * inv * inv;
        const float rv[4] = {row.x, row.y, row.z, row.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) step_v.apply(v, static_cast<size_t>(o) * kHyperWidth + 4 * lane + j, scale * (gw[j] - rv[j] * pull));
        if (lane == 0) {
            step_g.apply(g, o, dot * inv);
            step_b.apply(b, o, gb);
        }
    } else if (lane == 0) {
        for (int n = 0; n < num_instances; ++n) zbar[n * kHyperWaves + wave] = 0.0f;
    }
    *reinterpret_cast<float4*>(rows + wave * kHyperWidth + 4 * lane) = make_float4(scale * row.x, scale * row.y, scale * row.z, scale * row.w);
    __syncthreads();
    const int channel = static_cast<int>(threadIdx.x) & (kHyperWidth - 1);
    float column[kHyperWaves];
#pragma unroll
    for (int w = 0; w < kHyperWaves; ++w) column[w] = rows[w * kHyperWidth + channel];
    float* out = partial_gh + static_cast<size_t>(blockIdx.x) * num_instances * kHyperWidth;
    for (int n = static_cast<int>(threadIdx.x) >> 8; n < num_instances; n += kHyperThreads / kHyperWidth) {
        float acc = 0.0f;
#pragma unroll
        for (int w = 0; w < kHyperWaves; ++w) acc += zbar[n * kHyperWaves + w] * column[w];
        out[n * kHyperWidth + channel] = acc;
    }
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
// and leaves this instance's share of gamma_bar = a_bar y, beta_bar = a_bar in norm_partials [N][2][256] (hyper_norm_adam_kernel sums
// them at the end of the step).  is_embedding: the input was the embeddings themselves; h_bar is their gradient and they are stepped.
__global__ __launch_bounds__(kHyperNormThreads) void hyper_norm_backward_kernel(
    const float* __restrict__ partial_gh, int num_partials, const float* __restrict__ z_prev, int num_instances,
    const float* __restrict__ gamma, const float* __restrict__ beta, AdamTensors embeddings, int is_embedding, HyperAdam adam,
    float* __restrict__ gz_out, float* __restrict__ norm_partials) {
    __shared__ float scratch[4];
    const int n = blockIdx.x, c = threadIdx.x;
    const size_t stride = static_cast<size_t>(num_instances) * kHyperWidth;
    const float* src = partial_gh + static_cast<size_t>(n) * kHyperWidth + c;
    float hb = 0.0f;
#pragma unroll 8
    for (int p = 0; p < num_partials; ++p) hb += src[p * stride];
    if (is_embedding) {
        const AdamStep step(embeddings, adam);
        step.apply(embeddings, static_cast<size_t>(n) * kHyperWidth + c, hb);
        return;
    }
    float y = z_prev[static_cast<size_t>(n) * kHyperWidth + c];
    const float mean = block_sum_256(y, scratch) * (1.0f / kHyperWidth);
    y -= mean;
    const float inv_std = rsqrtf(block_sum_256(y * y, scratch) * (1.0f / kHyperWidth) + kHyperNormEps);
    y *= inv_std;
    const float gam = gamma[c];
    const float a_bar = hb * gelu_exact_derivative(y * gam + beta[c]);
    const float y_bar = a_bar * gam;
    const float m1 = block_sum_256(y_bar, scratch) * (1.0f / kHyperWidth);
    const float m2 = block_sum_256(y_bar * y, scratch) * (1.0f / kHyperWidth);
    gz_out[static_cast<size_t>(n) * kHyperWidth + c] = (y_bar - m1 - y * m2) * inv_std;
    norm_partials[(static_cast<size_t>(n) * 2 + 0) * kHyperWidth + c] = a_bar * y;
    norm_partials[(static_cast<size_t>(n) * 2 + 1) * kHyperWidth + c] = a_bar;
}

// Adam on the LayerNorm affines, all four norms in one launch (workgroup = norm, thread = channel) after the whole backward chain has
// read them: gamma_bar / beta_bar = the instances' shares summed in instance order.
struct HyperNorms { AdamTensors gamma[4], beta[4]; };

__global__ __launch_bounds__(kHyperNormThreads) void hyper_norm_adam_kernel(HyperNorms norms, const float* __restrict__ norm_partials, int num_instances, HyperAdam adam) {
    const int k = blockIdx.x, c = threadIdx.x;
    const float* src = norm_partials + static_cast<size_t>(k) * num_instances * 2 * kHyperWidth;
    float dgamma = 0.0f, dbeta = 0.0f;
    for (int n = 0; n < num_instances; ++n) {
        dgamma += src[(static_cast<size_t>(n) * 2 + 0) * kHyperWidth + c];
        dbeta += src[(static_cast<size_t>(n) * 2 + 1) * kHyperWidth + c];
    }
    const AdamStep step_gamma(norms.gamma[k], adam), step_beta(norms.beta[k], adam);
    step_gamma.apply(norms.gamma[k], c, dgamma);
    step_beta.apply(norms.beta[k], c, dbeta);
}

// After every kernel of the step has read them: advance the step counters of all the tensors stepped above and decay the two
// learning rates (ExponentialLR steps after the optimiser).
struct HyperStepCounters { float* step[32]; int count; };

__global__ void hyper_finish_kernel(HyperStepCounters counters, float* embedding_lr, float* hyper_lr, float gamma) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < counters.count) *counters.step[idx] += 1.0f;
    if (idx == 0) { *embedding_lr *= gamma; *hyper_lr *= gamma; }
}

}  // namespace vsrd
