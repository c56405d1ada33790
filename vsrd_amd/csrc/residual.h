// Per-instance residual SDF: sigmoid(MLP_w(encode((|x|, y, z) / 100)) - 1) as a first-order jet (value + gradient
// w.r.t. the local position), evaluated per lane with wave-uniform weights.
//
// Reference semantics (CPU restatement: oracle/fields.py, float64 blueprint incl. the adjoint: oracle/analytic_mlp.py):
//   residual_distance_field / residual_composition   scripts/main.py:433-458
//   SinusoidalEncoder (8 octaves, [coord][freq][cos,sin])   vsrd/models/encoders/sinusoidal_encoder.py:12-18
//   HyperDistanceField.distance_field (49->16, 3 x 17->16, 17->1; LayerNorm without affine + exact GELU
//   between layers; weights [out][in+1], bias last)          vsrd/models/fields/hyper_distance_field.py:57-73
//   gradient = autograd.grad(sdf, positions)                 vsrd/rendering/renderers.py:218-228 (forward-mode here)
//
// All 1617 weights of instance i are wave-uniform: they are read with scalar loads and used as SGPR operands;
// the activations (16 channels x (value + 3 tangents)) live in registers with compile-time indices.
#pragma once
#include "wave.h"

namespace vsrd {

constexpr int kMlpWeights = 1617;
constexpr int kMlpHidden = 16;
constexpr int kMlpFeatures = 48;
constexpr float kPositionScale = 100.0f;      // max(distance_range), main.py:441
constexpr float kLayerNormEps = 1.0e-5f;

struct Jet16 {
    float v[kMlpHidden];        // value
    float t[3][kMlpHidden];     // tangents d/d(folded, scaled position)
};

__device__ __forceinline__ float gauss_pdf(float y) { return fast_exp(-0.5f * y * y) * 0.3989422804014327f; }
__device__ __forceinline__ float gauss_cdf(float y) { return 0.5f * (1.0f + erff(y * 0.7071067811865476f)); }

// First layer: encoder fused with Linear(48 -> 16).  Row o of the weight block is [W[o][0..47], bias].
__device__ __forceinline__ void mlp_first_layer(const float* __restrict__ w, float f0, float f1, float f2, Jet16& z) {
    float feat[3][16], dfeat[3][16];
    const float f[3] = {f0, f1, f2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float octave = static_cast<float>(1 << k);
            const float x = f[c] * octave;                       // exact scaling; sin(pi x), cos(pi x)
            const float s = sinpif(x), co = cospif(x);
            const float omega = octave * 3.14159265358979323846f;
            feat[c][2 * k] = co; feat[c][2 * k + 1] = s;
            dfeat[c][2 * k] = -omega * s; dfeat[c][2 * k + 1] = omega * co;
        }
    }
#pragma unroll
    for (int o = 0; o < kMlpHidden; ++o) {
        const float* row = w + o * (kMlpFeatures + 1);
        float acc = row[kMlpFeatures];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float tan = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float wj = row[c * 16 + j];
                acc += wj * feat[c][j];
                tan += wj * dfeat[c][j];
            }
            z.t[c][o] = tan;
        }
        z.v[o] = acc;
    }
}

// LayerNorm (no affine) followed by exact GELU on a jet, in place.
__device__ __forceinline__ void mlp_norm_gelu(Jet16& z) {
    float mean = 0.0f;
#pragma unroll
    for (int o = 0; o < kMlpHidden; ++o) mean += z.v[o];
    mean *= (1.0f / kMlpHidden);
    float var = 0.0f;
#pragma unroll
    for (int o = 0; o < kMlpHidden; ++o) { z.v[o] -= mean; var += z.v[o] * z.v[o]; }
    const float inv_s = __builtin_amdgcn_rsqf(var * (1.0f / kMlpHidden) + kLayerNormEps);
    float tmean[3] = {0.0f, 0.0f, 0.0f}, q[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int o = 0; o < kMlpHidden; ++o) {
        z.v[o] *= inv_s;                                                     // y
#pragma unroll
        for (int c = 0; c < 3; ++c) { tmean[c] += z.t[c][o]; q[c] += z.v[o] * z.t[c][o]; }
    }
#pragma unroll
    for (int o = 0; o < kMlpHidden; ++o) {
        const float y = z.v[o];
        const float cdf = gauss_cdf(y);
        const float g1 = cdf + y * gauss_pdf(y);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float dy = (z.t[c][o] - tmean[c] * (1.0f / kMlpHidden) - y * q[c] * (1.0f / kMlpHidden)) * inv_s;
            z.t[c][o] = dy * g1;
        }
        z.v[o] = y * cdf;
    }
}

// Linear(16 -> kOut) on a jet; weight rows are [W[o][0..15], bias].
template <int kOut>
__device__ __forceinline__ void mlp_linear(const float* __restrict__ w, const Jet16& a, Jet16& z) {
#pragma unroll
    for (int o = 0; o < kOut; ++o) {
        const float* row = w + o * (kMlpHidden + 1);
        float acc = row[kMlpHidden], t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
#pragma unroll
        for (int j = 0; j < kMlpHidden; ++j) {
            const float wj = row[j];
            acc += wj * a.v[j];
            t0 += wj * a.t[0][j]; t1 += wj * a.t[1][j]; t2 += wj * a.t[2][j];
        }
        z.v[o] = acc; z.t[0][o] = t0; z.t[1][o] = t1; z.t[2][o] = t2;
    }
}

struct Residual { float value; float gx, gy, gz; };

// residual(p) and d residual / d p for local position p, instance weights w (wave-uniform pointer).
__device__ __forceinline__ Residual residual_forward(const float* __restrict__ w, float px, float py, float pz) {
    const float fold = (px > 0.0f) ? 1.0f : ((px < 0.0f) ? -1.0f : 0.0f);
    const float inv = 1.0f / kPositionScale;
    Jet16 a, z;
    mlp_first_layer(w, fabsf(px) * inv, py * inv, pz * inv, z);
    const float* wl = w + (kMlpFeatures + 1) * kMlpHidden;
#pragma unroll
    for (int layer = 0; layer < 3; ++layer) {
        mlp_norm_gelu(z);
        a = z;
        mlp_linear<kMlpHidden>(wl, a, z);
        wl += (kMlpHidden + 1) * kMlpHidden;
    }
    mlp_norm_gelu(z);
    a = z;
    mlp_linear<1>(wl, a, z);
    Residual r;
    r.value = fast_rcp(1.0f + fast_exp(-(z.v[0] - 1.0f)));
    const float kappa = r.value * (1.0f - r.value) * inv;
    r.gx = kappa * z.t[0][0] * fold; r.gy = kappa * z.t[1][0]; r.gz = kappa * z.t[2][0];
    return r;
}

}  // namespace vsrd
