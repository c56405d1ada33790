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

// Start of a weight row: makes the row pointer opaque to the optimiser at this point, so the scalar loads of the row are
// issued here and not hoisted to the top of the function -- with ~100 rows of 16 SGPRs each in flight the compiler otherwise
// spills SGPRs into VGPR lanes (59k v_readlane in the adjoint before this).
// `after` is any value produced by the previous row: the data dependence is what keeps the rows (and their loads) in order.
template <typename W>
__device__ __forceinline__ W row_begin(W row, float after) {
    asm volatile("" : "+s"(row) : "v"(after));
    return row;
}

// First layer: encoder fused with Linear(48 -> 16).  Row o of the weight block is [W[o][0..47], bias].
template <typename W>
__device__ __forceinline__ void mlp_first_layer(W w, float f0, float f1, float f2, Jet16& z) {
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
    float chain = f0;
#pragma unroll
    for (int o = 0; o < kMlpHidden; ++o) {
        const W row = row_begin(w + o * (kMlpFeatures + 1), chain);
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
        chain = acc;
        __builtin_amdgcn_sched_barrier(0);
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
template <int kOut, typename W>
__device__ __forceinline__ void mlp_linear(W w, const Jet16& a, Jet16& z) {
    float chain = a.v[0];
#pragma unroll
    for (int o = 0; o < kOut; ++o) {
        const W row = row_begin(w + o * (kMlpHidden + 1), chain);
        float acc = row[kMlpHidden], t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
#pragma unroll
        for (int j = 0; j < kMlpHidden; ++j) {
            const float wj = row[j];
            acc += wj * a.v[j];
            t0 += wj * a.t[0][j]; t1 += wj * a.t[1][j]; t2 += wj * a.t[2][j];
        }
        z.v[o] = acc; z.t[0][o] = t0; z.t[1][o] = t1; z.t[2][o] = t2;
        chain = t2;
        __builtin_amdgcn_sched_barrier(0);
    }
}

struct Residual { float value; float gx, gy, gz; };

// A pointer that arrives in VGPRs (function argument of a non-inlined call) made provably wave-uniform again, so that the
// weight loads behind it are selected as scalar loads.
// constant address space (4): read-only for the whole launch (the MLP weights are never written by these kernels), which is what
// lets uniform loads through it be selected as SMEM inside a non-inlined function.
using GlobalFloats = const __attribute__((address_space(4))) float*;

__device__ __forceinline__ GlobalFloats uniform_pointer(const float* p) {
    const unsigned long long bits = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits >> 32));
    return reinterpret_cast<GlobalFloats>((static_cast<unsigned long long>(hi) << 32) | lo);   // SGPR pair
}

// residual(p) and d residual / d p for local position p, instance weights w (wave-uniform pointer).
// NOT inlined: the fully unrolled MLP is ~9k instructions; one copy per kernel instead of one per call site keeps the
// residual kernels from being megabytes of straight-line code (the 64 KB instruction cache is the scarce resource here).
__device__ __attribute__((noinline)) Residual residual_forward(const float* w_in, float px, float py, float pz) {
    const GlobalFloats w = uniform_pointer(w_in);
    const float fold = (px > 0.0f) ? 1.0f : ((px < 0.0f) ? -1.0f : 0.0f);
    const float inv = 1.0f / kPositionScale;
    Jet16 a, z;
    mlp_first_layer(w, fabsf(px) * inv, py * inv, pz * inv, z);
    GlobalFloats wl = w + (kMlpFeatures + 1) * kMlpHidden;
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

// ---- adjoint ------------------------------------------------------------------------------------------------
// Blueprint: oracle/analytic_mlp.py::backward (float64, checked against autograd).  One lane = one sample.
// Weight adjoints are reduced over the 64 lanes 16 values at a time with the reduce-scatter butterfly and added
// into the wave's LDS row `wbar` [1617] (lane j < 16 owns column j of the current weight row: thread-private
// addresses, no hazards).

// What the adjoint of one [LayerNorm -> GELU -> Linear] block needs, recomputed from the block's input jet.
struct BlockState {
    float y[kMlpHidden], g1[kMlpHidden], g2[kMlpHidden];   // normalised value, GELU', pdf(y)(2 - y^2)
    float dy[3][kMlpHidden];
    float a[kMlpHidden], da[3][kMlpHidden];                // activations fed to the linear
    float q[3];                                            // mean(y * dz_c)
    float inv_s;
};

__device__ __forceinline__ void block_state(const Jet16& z, BlockState& b) {
    float mean = 0.0f;
#pragma unroll
    for (int o = 0; o < kMlpHidden; ++o) mean += z.v[o];
    mean *= (1.0f / kMlpHidden);
    float var = 0.0f;
#pragma unroll
    for (int o = 0; o < kMlpHidden; ++o) { b.y[o] = z.v[o] - mean; var += b.y[o] * b.y[o]; }
    b.inv_s = __builtin_amdgcn_rsqf(var * (1.0f / kMlpHidden) + kLayerNormEps);
    float tmean[3] = {0.0f, 0.0f, 0.0f};
    b.q[0] = b.q[1] = b.q[2] = 0.0f;
#pragma unroll
    for (int o = 0; o < kMlpHidden; ++o) {
        b.y[o] *= b.inv_s;
#pragma unroll
        for (int c = 0; c < 3; ++c) { tmean[c] += z.t[c][o]; b.q[c] += b.y[o] * z.t[c][o]; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { tmean[c] *= (1.0f / kMlpHidden); b.q[c] *= (1.0f / kMlpHidden); }
#pragma unroll
    for (int o = 0; o < kMlpHidden; ++o) {
        const float y = b.y[o];
        const float cdf = gauss_cdf(y), pdf = gauss_pdf(y);
        b.g1[o] = cdf + y * pdf;
        b.g2[o] = pdf * (2.0f - y * y);
        b.a[o] = y * cdf;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            b.dy[c][o] = (z.t[c][o] - tmean[c] - y * b.q[c]) * b.inv_s;
            b.da[c][o] = b.dy[c][o] * b.g1[o];
        }
    }
}

// P(v) = (v - mean(v) - y mean(v y)) / s  (LayerNorm's symmetric Jacobian), in place on 16 values.
__device__ __forceinline__ void layer_norm_adjoint(float (&v)[kMlpHidden], const BlockState& b) {
    float m = 0.0f, my = 0.0f;
#pragma unroll
    for (int o = 0; o < kMlpHidden; ++o) { m += v[o]; my += v[o] * b.y[o]; }
    m *= (1.0f / kMlpHidden); my *= (1.0f / kMlpHidden);
#pragma unroll
    for (int o = 0; o < kMlpHidden; ++o) v[o] = (v[o] - m - b.y[o] * my) * b.inv_s;
}

// Adjoint of one block: (z_bar, dz_bar) of the linear's kOut outputs -> adjoint of the block's input jet (returned in zb),
// weight adjoints of the linear accumulated into wbar (row-major [kOut][17]).
template <int kOut, typename W>
__device__ __forceinline__ void block_adjoint(W w, const Jet16& z_in, Jet16& zb, float* wbar, int lane) {
    BlockState b;
    block_state(z_in, b);
    float a_bar[kMlpHidden], da_bar[3][kMlpHidden];
#pragma unroll
    for (int j = 0; j < kMlpHidden; ++j) { a_bar[j] = 0.0f; da_bar[0][j] = da_bar[1][j] = da_bar[2][j] = 0.0f; }
    float bias[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) bias[o] = 0.0f;
    // (1) weight adjoints: outer products reduced over the wave -- no weights needed
#pragma unroll
    for (int o = 0; o < kOut; ++o) {
        float prod[16];
#pragma unroll
        for (int j = 0; j < kMlpHidden; ++j)
            prod[j] = zb.v[o] * b.a[j] + zb.t[0][o] * b.da[0][j] + zb.t[1][o] * b.da[1][j] + zb.t[2][o] * b.da[2][j];
        const float r = wave_reduce16_scatter(prod, lane);
        if (lane < kMlpHidden) wbar[o * (kMlpHidden + 1) + lane] += r;
        bias[o] = zb.v[o];
    }
    {
        const float r = wave_reduce16_scatter(bias, lane);
        if (lane < kOut) wbar[lane * (kMlpHidden + 1) + kMlpHidden] += r;
    }
    // (2) activation adjoints: transposed mat-vec, one weight row (16 SGPRs) live at a time
    float chain = b.inv_s;
#pragma unroll
    for (int o = 0; o < kOut; ++o) {
        const W row = row_begin(w + o * (kMlpHidden + 1), chain);
#pragma unroll
        for (int j = 0; j < kMlpHidden; ++j) {
            const float wj = row[j];
            a_bar[j] += wj * zb.v[o];
            da_bar[0][j] += wj * zb.t[0][o]; da_bar[1][j] += wj * zb.t[1][o]; da_bar[2][j] += wj * zb.t[2][o];
        }
        chain = da_bar[2][kMlpHidden - 1];
        __builtin_amdgcn_sched_barrier(0);
    }
    // GELU jet adjoint
    float y_bar[kMlpHidden];
    float dyb[3][kMlpHidden];
    float s_bar = 0.0f;
    float dot[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < kMlpHidden; ++j) {
        float cross = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            cross += da_bar[c][j] * b.dy[c][j];
            dyb[c][j] = da_bar[c][j] * b.g1[j];
            s_bar -= dyb[c][j] * b.dy[c][j];
            dot[c] += dyb[c][j] * b.y[j];
        }
        y_bar[j] = a_bar[j] * b.g1[j] + cross * b.g2[j];
    }
    s_bar *= b.inv_s;
    // LayerNorm jet adjoint: explicit dependence of dy on (y, s), then through y = (z - mean)/s
#pragma unroll
    for (int j = 0; j < kMlpHidden; ++j) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            y_bar[j] -= (dyb[c][j] * b.q[c] + z_in.t[c][j] * dot[c] * (1.0f / kMlpHidden)) * b.inv_s;
    }
    layer_norm_adjoint(y_bar, b);
#pragma unroll
    for (int c = 0; c < 3; ++c) layer_norm_adjoint(dyb[c], b);
#pragma unroll
    for (int j = 0; j < kMlpHidden; ++j) {
        zb.v[j] = y_bar[j] + s_bar * b.y[j] * (1.0f / kMlpHidden);
        zb.t[0][j] = dyb[0][j]; zb.t[1][j] = dyb[1][j]; zb.t[2][j] = dyb[2][j];
    }
}

struct ResidualAdjoint { float px, py, pz; };

// Adjoint of residual_forward at local position p: res_bar = dL/d residual, (gbx,gby,gbz) = dL/d(grad_p residual).
// Accumulates dL/dw into wbar (LDS, [1617]); returns dL/dp.
__device__ __attribute__((noinline)) ResidualAdjoint residual_backward(const float* w_in, float px, float py, float pz,
                                                                       float res_bar, float gbx, float gby, float gbz, float* wbar, int lane) {
    const GlobalFloats w = uniform_pointer(w_in);
    const float fold = (px > 0.0f) ? 1.0f : ((px < 0.0f) ? -1.0f : 0.0f);
    const float inv = 1.0f / kPositionScale;
    const float f[3] = {fabsf(px) * inv, py * inv, pz * inv};
    const float folds[3] = {fold, 1.0f, 1.0f};
    // ---- forward; only z0 (first-layer output) and z2 (second hidden linear's output) are kept: z1 and z3 are recomputed
    //      from them when their block's adjoint runs (two extra hidden layers instead of 128 more live registers) ---------
    Jet16 z0, z2, a, out;
    mlp_first_layer(w, f[0], f[1], f[2], z0);
    const GlobalFloats w1 = w + (kMlpFeatures + 1) * kMlpHidden;
    const GlobalFloats w2 = w1 + (kMlpHidden + 1) * kMlpHidden;
    const GlobalFloats w3 = w2 + (kMlpHidden + 1) * kMlpHidden;
    const GlobalFloats w4 = w3 + (kMlpHidden + 1) * kMlpHidden;
    {
        Jet16 z1, z3;
        a = z0; mlp_norm_gelu(a); mlp_linear<kMlpHidden>(w1, a, z1);
        mlp_norm_gelu(z1); mlp_linear<kMlpHidden>(w2, z1, z2);
        a = z2; mlp_norm_gelu(a); mlp_linear<kMlpHidden>(w3, a, z3);
        mlp_norm_gelu(z3); mlp_linear<1>(w4, z3, out);
    }
    const float res = fast_rcp(1.0f + fast_exp(-(out.v[0] - 1.0f)));
    const float kappa = res * (1.0f - res);
    const float gb[3] = {gbx, gby, gbz};
    // ---- adjoint of the sigmoid head -----------------------------------------------------------------------------------
    Jet16 zb;
    float kappa_bar = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        zb.t[c][0] = gb[c] * kappa * folds[c] * inv;
        kappa_bar += gb[c] * out.t[c][0] * folds[c] * inv;
    }
    zb.v[0] = (res_bar + kappa_bar * (1.0f - 2.0f * res)) * kappa;
    // ---- blocks 4..1 ---------------------------------------------------------------------------------------------------
    const int off1 = (kMlpFeatures + 1) * kMlpHidden, blk = (kMlpHidden + 1) * kMlpHidden;
    {
        Jet16 z3;
        a = z2; mlp_norm_gelu(a); mlp_linear<kMlpHidden>(w3, a, z3);
        block_adjoint<1>(w4, z3, zb, wbar + off1 + 3 * blk, lane);
    }
    block_adjoint<kMlpHidden>(w3, z2, zb, wbar + off1 + 2 * blk, lane);
    {
        Jet16 z1;
        a = z0; mlp_norm_gelu(a); mlp_linear<kMlpHidden>(w1, a, z1);
        block_adjoint<kMlpHidden>(w2, z1, zb, wbar + off1 + blk, lane);
    }
    block_adjoint<kMlpHidden>(w1, z0, zb, wbar + off1, lane);
    // ---- first layer + encoder -----------------------------------------------------------------------------------------
    float f_bar[3] = {0.0f, 0.0f, 0.0f};
    float bias[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) bias[o] = zb.v[o];
    {
        const float r = wave_reduce16_scatter(bias, lane);
        if (lane < kMlpHidden) wbar[lane * (kMlpFeatures + 1) + kMlpFeatures] += r;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float feat[16], dfeat[16], feat_bar[16], dfeat_bar[16];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float octave = static_cast<float>(1 << k);
            const float x = f[c] * octave;
            const float s = sinpif(x), co = cospif(x);
            const float omega = octave * 3.14159265358979323846f;
            feat[2 * k] = co; feat[2 * k + 1] = s;
            dfeat[2 * k] = -omega * s; dfeat[2 * k + 1] = omega * co;
            feat_bar[2 * k] = feat_bar[2 * k + 1] = 0.0f;
            dfeat_bar[2 * k] = dfeat_bar[2 * k + 1] = 0.0f;
        }
#pragma unroll
        for (int o = 0; o < kMlpHidden; ++o) {
            float prod[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) prod[j] = zb.v[o] * feat[j] + zb.t[c][o] * dfeat[j];
            const float r = wave_reduce16_scatter(prod, lane);
            if (lane < 16) wbar[o * (kMlpFeatures + 1) + c * 16 + lane] += r;
        }
        float chain = feat[0];
#pragma unroll
        for (int o = 0; o < kMlpHidden; ++o) {
            const GlobalFloats row = row_begin(w + o * (kMlpFeatures + 1) + c * 16, chain);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float wj = row[j];
                feat_bar[j] += wj * zb.v[o];
                dfeat_bar[j] += wj * zb.t[c][o];
            }
            chain = dfeat_bar[15];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float omega = static_cast<float>(1 << k) * 3.14159265358979323846f;
            const float co = feat[2 * k], s = feat[2 * k + 1];
            f_bar[c] += feat_bar[2 * k] * (-omega * s) + feat_bar[2 * k + 1] * (omega * co)
                      + dfeat_bar[2 * k] * (-omega * omega * co) + dfeat_bar[2 * k + 1] * (-omega * omega * s);
        }
    }
    ResidualAdjoint r;
    r.px = f_bar[0] * fold * inv; r.py = f_bar[1] * inv; r.pz = f_bar[2] * inv;
    return r;
}

}  // namespace vsrd
