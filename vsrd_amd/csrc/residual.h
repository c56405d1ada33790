// Per-instance residual SDF on the matrix cores: sigmoid(MLP_w(encode((|x|, y, z) / 100)) - 1) as a first-order jet
// (value + gradient w.r.t. the local position) for the 64 samples of a wave.
//
// Reference semantics (CPU restatement: oracle/fields.py; float64 blueprint of the adjoint: oracle/analytic_mlp.py):
//   residual_distance_field / residual_composition          scripts/main.py:433-458
//   SinusoidalEncoder (8 octaves, [coord][freq][cos,sin])   vsrd/models/encoders/sinusoidal_encoder.py:12-18
//   HyperDistanceField.distance_field (49->16, 3 x 17->16, 17->1; LayerNorm without affine + exact GELU between the
//   layers; weights [out][in+1], bias last)                 vsrd/models/fields/hyper_distance_field.py:57-73
//   gradient = autograd.grad(sdf, positions)                vsrd/rendering/renderers.py:218-228 (forward mode here)
//
// Mapping.  The wave holds 64 sample points (lane = point).  The MLP runs on TILES of 16 points: for tile q the lane
// (g = lane >> 4, m = lane & 15) works on point 16 q + m, and a 16-channel activation of that point is spread over the four
// lanes {m, m+16, m+32, m+48} x 4 registers: register j of row g is channel 4 g + j.  That is exactly the C/D layout of
// v_mfma_f32_16x16x4_f32 (col = lane & 15 = point, row = 4 (lane >> 4) + reg = channel), and -- because the k index of a B
// operand is also lane >> 4 -- register s of a layer's output IS the B operand of k-step s of the next layer when the weight
// (A) operand of that step is W[o = lane & 15][4 g + s].  Layers therefore chain with no data movement; the per-point
// reductions of LayerNorm run over 4 registers and 2 row swaps (v_permlane16/32_swap).  The exact-f32 MFMA has the same
// peak as the f32 VALU (MI355X_MICROARCH.md: 64 FLOP/clk/SIMD) but frees the VALU for the encoder / LayerNorm / GELU work,
// needs one VGPR per weight operand instead of one SGPR per weight (the scalar version spilled SGPRs and waited on a scalar
// load per weight row), and is bitwise a k-ordered fmaf chain.
//
// All functions here are WAVE-COOPERATIVE: they must be called with all 64 lanes active.
#pragma once
#include "wave.h"

namespace vsrd {

constexpr int kMlpWeights = 1617;
constexpr int kMlpHidden = 16;
constexpr int kMlpFeatures = 48;
constexpr int kMlpRow0 = kMlpFeatures + 1;                   // first-layer row: 48 weights + bias
constexpr int kMlpRow = kMlpHidden + 1;                      // hidden row: 16 weights + bias
constexpr int kMlpLayer1 = kMlpRow0 * kMlpHidden;            // 784: offset of the first hidden linear
constexpr int kMlpBlock = kMlpRow * kMlpHidden;              // 272
constexpr int kMlpHead = kMlpLayer1 + 3 * kMlpBlock;         // 1600: [w4[0..15], b4]
constexpr float kPositionScale = 100.0f;                     // max(distance_range), main.py:441
constexpr float kLayerNormEps = 1.0e-5f;
constexpr float kPi = 3.14159265358979323846f;
constexpr unsigned kMlpCentredBit = 0x10u;                   // in the tile-mask argument: the weights arrive centred (VSRD_FLAG_MLP_WEIGHTS_CENTRED)
constexpr int kMlpStartShift = 8;                            // ... and bits 8..13: the lane the tiles are counted from (residual_forward)
constexpr unsigned kMlpSplitBit = 0x20u;                     // ... and: the weight pointer is the instance's split-bf16 operand IMAGE (VSRD_FLAG_MLP_SPLIT_BF16, below)

using f32x4 = __attribute__((ext_vector_type(4))) float;

// The two entry points are NOT inlined: one copy per kernel keeps the residual kernels inside the instruction cache.
#define VSRD_RESIDUAL_FN __device__ __attribute__((noinline))

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }
__device__ __forceinline__ float hsum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]); }

// Sum over the four rows of 16 lanes: every lane (g, m) receives sum_g' v(g', m): two row swaps (v_permlane16_swap / v_permlane32_swap).
// Rounds 2-4 used ONE fp32 MFMA with an all-ones A operand instead (D[i][m] = sum_k 1 * B[k][m], k = row) on the assumption that the
// matrix instruction runs beside the vector ones.  It does not (tools/micro/mfma_valu_interleave.hip: 32.5 cycles alone, ~42 in a stream
// of vector instructions, nothing hidden), and the swaps cost ~18: round 5, same box, same build otherwise: config 3 3.05 -> 3.13 Mrays/s,
// the native residual step 0.838 -> 0.820 ms.  -DVSRD_ROWS_SUM_MFMA: the matrix form.
#ifndef VSRD_ROWS_SUM_MFMA
__device__ __forceinline__ float rows_sum(float v) { return add_xor32(add_xor16(v)); }
#else
__device__ __forceinline__ float rows_sum(float v) { return mfma4(1.0f, v, f32x4{0.0f, 0.0f, 0.0f, 0.0f})[0]; }
#endif

// Value of lane 16 q + (lane & 15): row q broadcast to all four rows.  `start` (wave-uniform): the rows are counted from lane `start`
// on, cyclically (residual_forward with rotated tiles; ds_bpermute takes the lane index modulo 64 by itself).
__device__ __forceinline__ float from_row(float v, int q, int lane, int start = 0) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((((q << 4) | (lane & 15)) + start) << 2, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float rotate_lanes(float v, int byte_address) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_address, __builtin_bit_cast(int, v)));
}

// Standard normal pdf and cdf (exact GELU = y * cdf(y)).  erf through Abramowitz & Stegun 7.1.26,
//   erf(x) = 1 - (a1 t + ... + a5 t^5) exp(-x^2), t = 1 / (1 + p x), x >= 0, |error| <= 1.5e-7 (one f32 ulp of a cdf near 1),
// which shares its exponential exp(-y^2 / 2) with the pdf: ~17 instructions for (cdf, pdf) instead of ~60 through erff.
struct Gauss { float cdf, pdf; };
__device__ __forceinline__ Gauss gauss(float y) {
#ifdef VSRD_LIBM_ERF
    return {0.5f * (1.0f + erff(y * 0.7071067811865476f)), fast_exp(-0.5f * y * y) * 0.3989422804014327f};
#else
    const float e = fast_exp(-0.5f * y * y);
    const float t = fast_rcp(1.0f + 0.3275911f * 0.7071067811865476f * fabsf(y));
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    const float half_erfc = 0.5f * poly * e;                      // 0.5 erfc(|y| / sqrt 2)
    return {(y < 0.0f) ? half_erfc : 1.0f - half_erfc, e * 0.3989422804014327f};
#endif
}

// One 16-point tile of a 16-channel jet: value and the three tangents d/d(folded, scaled local position).
struct TileJet { f32x4 v, t[3]; };

// A operands and C initialisers of one instance for the forward pass, per lane (g, o = lane & 15).
struct ForwardWeights {
    float a0[12];     // W0[o][16 c + 4 g + s']  at index 4 c + s'   (k-step of the first layer: coordinate c, feature 4 g + s')
    float a[3][4];    // W_l[o][4 g + s]
    f32x4 b0, b[3];   // biases as accumulator initialisers: channel 4 g + j
    f32x4 w4;         // head weights of channel 4 g + j
    float b4;
};

// The weight pointer reaches the non-inlined functions in VGPRs: made wave-uniform again (SGPR base + per-lane offset
// addressing) and typed as global memory so the loads are global_load, not flat_load.
using GlobalWeights = const __attribute__((address_space(1))) float*;

__device__ __forceinline__ GlobalWeights uniform_weights(const float* p) {
    const unsigned long long bits = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits >> 32));
    return reinterpret_cast<GlobalWeights>((static_cast<unsigned long long>(hi) << 32) | lo);
}

__device__ __forceinline__ const float* uniform_weights_generic(const float* p) {
    const unsigned long long bits = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits >> 32));
    return reinterpret_cast<const float*>((static_cast<unsigned long long>(hi) << 32) | lo);
}

// Sum over the 16 lanes of each row (every lane of the row receives it).
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_move<kDppQuadXor1>(0.0f, v);
    v += dpp_move<kDppQuadXor2>(0.0f, v);
    v += dpp_move<kDppRowHalfMirror>(0.0f, v);
    v += dpp_move<kDppRowMirror>(0.0f, v);
    return v;
}

// LayerNorm's mean subtraction is folded into the weights: with C = I - 11^T/16, LN(W a + b) = LN(C W a + C b) and C W a + C b has zero
// channel mean by construction -- so every linear that feeds a LayerNorm (all four) is loaded with its column means (over the output
// channel = over the 16 lanes of a row) and its bias mean removed, and the norm needs no mean / tangent-mean reductions (4 of its 8).
// The adjoint is unchanged: the LayerNorm Jacobian P already produces zero-mean adjoints, so W_bar = z_bar a^T and W^T z_bar hold for
// the original W.
__device__ __forceinline__ void load_forward_weights(GlobalWeights w, int lane, ForwardWeights& fw, bool centred) {
    const int g = lane >> 4, o = lane & 15;
    const GlobalWeights row0 = w + o * kMlpRow0 + 4 * g;
#pragma unroll
    for (int s = 0; s < 12; ++s) fw.a0[s] = row0[(s >> 2) * 16 + (s & 3)];
#pragma unroll
    for (int j = 0; j < 4; ++j) fw.b0[j] = w[(4 * g + j) * kMlpRow0 + kMlpFeatures];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const GlobalWeights wl = w + kMlpLayer1 + l * kMlpBlock;
#pragma unroll
        for (int s = 0; s < 4; ++s) fw.a[l][s] = wl[o * kMlpRow + 4 * g + s];
#pragma unroll
        for (int j = 0; j < 4; ++j) fw.b[l][j] = wl[(4 * g + j) * kMlpRow + kMlpHidden];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) fw.w4[j] = w[kMlpHead + 4 * g + j];
    fw.b4 = w[kMlpHead + kMlpHidden];
    if (centred) return;                                           // wave-uniform: the caller did it once for the whole launch
#pragma unroll
    for (int s = 0; s < 12; ++s) fw.a0[s] -= row_sum16(fw.a0[s]) * (1.0f / kMlpHidden);
    fw.b0 -= splat4(rows_sum(hsum4(fw.b0)) * (1.0f / kMlpHidden));
#pragma unroll
    for (int l = 0; l < 3; ++l) {
#pragma unroll
        for (int s = 0; s < 4; ++s) fw.a[l][s] -= row_sum16(fw.a[l][s]) * (1.0f / kMlpHidden);
        fw.b[l] -= splat4(rows_sum(hsum4(fw.b[l])) * (1.0f / kMlpHidden));
    }
}

// sin(pi x) and cos(pi x) for |x| < 2^22: x = n / 2 + r with n = rint(2 x) and |r| <= 1/4 exactly (x, n / 2 and their difference
// are fp32-exact), Taylor polynomials on |pi r| <= pi / 4 (truncation 1.8e-9 / 2.4e-8), quadrant fix-up by n mod 4.
// ~24 instructions for the pair instead of ~47 through sinpif + cospif.
__device__ __forceinline__ void sincospi(float x, float& sine, float& cosine) {
#ifdef VSRD_LIBM_SINCOS
    sine = sinpif(x); cosine = cospif(x);
#else
    const float n = rintf(2.0f * x);
    const float r = fmaf(n, -0.5f, x);
    const int quadrant = static_cast<int>(n);
    const float r2 = r * r;
    const float sr = r * (3.14159265358979324f + r2 * (-5.16771278004997003f + r2 * (2.55016403987734548f + r2 * (-0.599264529320792077f + r2 * 0.0821458866111282288f))));
    const float cr = 1.0f + r2 * (-4.93480220054467931f + r2 * (4.05871212641676822f + r2 * (-1.33526276885458950f + r2 * 0.235330630358893205f)));
    const bool odd = (quadrant & 1) != 0;
    const unsigned s_bits = __float_as_uint(odd ? cr : sr) ^ ((static_cast<unsigned>(quadrant) & 2u) << 30);          // n mod 4 in {2, 3}: -
    const unsigned c_bits = __float_as_uint(odd ? sr : cr) ^ ((static_cast<unsigned>(quadrant + 1) & 2u) << 30);      // n mod 4 in {1, 2}: -
    sine = __uint_as_float(s_bits); cosine = __uint_as_float(c_bits);
#endif
}

// Encoder features of one tile: row g evaluates octaves 2 g and 2 g + 1 of every coordinate, i.e. features 4 g .. 4 g + 3
// ([cos, sin] per octave) of each 16-feature coordinate block, and their derivatives w.r.t. the scaled coordinate.
struct TileFeatures { f32x4 f[3], d[3]; };

__device__ __forceinline__ void encode_tile(float f0, float f1, float f2, int g, TileFeatures& e) {
    const float base = (g == 0) ? 1.0f : ((g == 1) ? 4.0f : ((g == 2) ? 16.0f : 64.0f));
    const float f[3] = {f0, f1, f2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // octave 2 g through the exact range reduction of sincospi, octave 2 g + 1 = twice the angle by the double-angle formulas
        // (3 instead of ~24 instructions; one doubling adds ~1e-7 to the 1e-7 of the pair it starts from)
        float s0, c0;
        sincospi(f[c] * base, s0, c0);                           // exact scaling; sin(pi x), cos(pi x)
        const float s1 = 2.0f * s0 * c0, c1 = fmaf(-2.0f * s0, s0, 1.0f);
        const float omega0 = base * kPi, omega1 = 2.0f * base * kPi;
        e.f[c][0] = c0; e.f[c][1] = s0; e.f[c][2] = c1; e.f[c][3] = s1;
        e.d[c][0] = -omega0 * s0; e.d[c][1] = omega0 * c0; e.d[c][2] = -omega1 * s1; e.d[c][3] = omega1 * c1;
    }
}

// Encoder fused with Linear(48 -> 16).  The tangent w.r.t. coordinate c only sees the 16 features of that coordinate.
__device__ __forceinline__ void first_layer_tile(const ForwardWeights& fw, const TileFeatures& e, TileJet& z) {
    z.v = fw.b0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        z.t[c] = splat4(0.0f);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            z.v = mfma4(fw.a0[4 * c + s], e.f[c][s], z.v);
            z.t[c] = mfma4(fw.a0[4 * c + s], e.d[c][s], z.t[c]);
        }
    }
}

// Linear(16 -> 16) on a tile jet.
__device__ __forceinline__ void linear_tile(const float (&a)[4], f32x4 bias, f32x4 in_v, f32x4 in_t0, f32x4 in_t1, f32x4 in_t2, TileJet& out) {
    out.v = bias;
    out.t[0] = out.t[1] = out.t[2] = splat4(0.0f);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        out.v = mfma4(a[s], in_v[s], out.v);
        out.t[0] = mfma4(a[s], in_t0[s], out.t[0]);
        out.t[1] = mfma4(a[s], in_t1[s], out.t[1]);
        out.t[2] = mfma4(a[s], in_t2[s], out.t[2]);
    }
}
__device__ __forceinline__ void linear_tile(const float (&a)[4], f32x4 bias, const TileJet& in, TileJet& out) {
    linear_tile(a, bias, in.v, in.t[0], in.t[1], in.t[2], out);
}

// LayerNorm (no affine) followed by exact GELU on a tile jet, in place.
// The jet arrives with zero channel mean (value and tangents): see load_forward_weights.
__device__ __forceinline__ void norm_gelu_tile(TileJet& z) {
    const float var = rows_sum(dot4(z.v, z.v)) * (1.0f / kMlpHidden);
    const float inv_s = __builtin_amdgcn_rsqf(var + kLayerNormEps);
    const f32x4 y = z.v * splat4(inv_s);
    float q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = rows_sum(dot4(y, z.t[c])) * (1.0f / kMlpHidden);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const Gauss n = gauss(y[j]);
        const float g1 = n.cdf + y[j] * n.pdf;
#pragma unroll
        for (int c = 0; c < 3; ++c) z.t[c][j] = (z.t[c][j] - y[j] * q[c]) * inv_s * g1;
        z.v[j] = y[j] * n.cdf;
    }
}

struct Residual { float value; float gx, gy, gz; };

// ---- adjoint ------------------------------------------------------------------------------------------------------
// Blueprint: oracle/analytic_mlp.py::backward (float64, checked against autograd).  Same tile layout as the forward.
//   activation adjoints  a_bar = W^T z_bar        : MFMA with the transposed weight operands  W[4 g + s][i = lane & 15]
//   weight adjoints      W_bar += z_bar a^T        : MFMA whose k index is the POINT: both operands are needed with the
//                                                    channel along lane & 15 and the point along (lane >> 4, k-step), i.e.
//                                                    transposed; the tiles go through a wave-private LDS scratch
//                                                    (4 ds_write_b32 + 1 ds_read_b128 per tile, conflict-free with a row
//                                                    pitch of 20 floats).  The accumulators come out as W_bar[4 g + j][lane & 15]
//                                                    -- the row-major layout of the weight block itself.
// Per call (64 points x 1 instance) the accumulators live in registers; they are added into the wave's LDS row
// `wbar` [1617] at the end (each lane owns its addresses).

constexpr int kTilePitch = 20;                               // floats per channel row of a staged tile (16 points + 4 pad)
constexpr int kTileFloats = 16 * kTilePitch;
constexpr int kMlpWbarFloats = 1632;                         // wbar [1617] padded to a multiple of 4 floats
constexpr int kMlpScratchTiles = 8;
constexpr int kMlpStashTiles = 10;                           // mlp_adjoint_points<true>: + six tiles of encoder features and tangents
constexpr int kMlpLdsFloats = kMlpWbarFloats + kMlpScratchTiles * kTileFloats;   // per-wave LDS of the residual adjoint

// Compiler-only ordering between LDS accesses of different lanes of this wave (DS operations of a wave execute in order).
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Stage a tile (register j of lane (g, m) = X[4 g + j][m]) / fetch it transposed: lane (g, r) <- X[r][4 g + s], s = 0..3.
// LDS reached through a non-inlined call arrives as a generic pointer; typed back to the local address space so the accesses
// are ds_read / ds_write, not flat_*.
using LdsFloats = __attribute__((address_space(3))) float*;

__device__ __forceinline__ void stage_tile(LdsFloats tile, f32x4 x, int lane) {
    LdsFloats dst = tile + (lane >> 4) * 4 * kTilePitch + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j * kTilePitch] = x[j];
}
__device__ __forceinline__ f32x4 fetch_tile(LdsFloats tile, int lane) {
    return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(tile + (lane & 15) * kTilePitch + (lane >> 4) * 4);
}

struct BackwardWeights {    // transposed A operands, lane (g, i = lane & 15)
    float at[3][4];         // W_l[4 g + s][i]
    float at0[3][4];        // W0[4 g + s][16 c + i]
};

__device__ __forceinline__ void load_backward_weights(GlobalWeights w, int lane, BackwardWeights& bw) {
    const int g = lane >> 4, i = lane & 15;
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int s = 0; s < 4; ++s) bw.at[l][s] = w[kMlpLayer1 + l * kMlpBlock + (4 * g + s) * kMlpRow + i];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s) bw.at0[c][s] = w[(4 * g + s) * kMlpRow0 + 16 * c + i];
}

// ---- single-tangent jets (the adjoint) -------------------------------------------------------------------------------------------
// The loss sees the three tangents of the MLP output only through the linear form  sum_c gl_bar_c * kappa * fold_c / 100 * dout_c, and a
// Jacobian-vector product is linear in its direction: with  delta = gl_bar * fold / 100  (per point) that form is kappa times the
// tangent of `out` ALONG delta.  So the adjoint pushes the ONE tangent  sum_c delta_c dfeat_c  through the network (seed kappa)
// instead of the three unit tangents (seeds kappa delta_c): the hidden layers carry 2 columns instead of 4 -- half their MFMAs, half
// the LayerNorm / GELU jet algebra, half the staged tiles -- for the same derivatives (oracle/analytic_mlp.py: backward_directional,
// checked against autograd).  delta is a constant of the adjoint (gl_bar is a seed), so nothing flows back into it.
struct TileJet1 { f32x4 v, t; };

// What the adjoint of one [LayerNorm -> GELU] needs, recomputed from the block's input jet.
struct TileState {
    f32x4 y, g1, g2, a;         // normalised value, GELU', pdf(y)(2 - y^2), activation
    f32x4 dy;                   // tangent of y; the activation tangent is dy * g1
    float q;                    // mean(y * dz)
    float inv_s;
    __device__ __forceinline__ f32x4 da() const { return dy * g1; }
};

__device__ __forceinline__ void tile_state(const TileJet1& z, TileState& b) {      // z has zero channel mean (load_forward_weights)
    const float var = rows_sum(dot4(z.v, z.v)) * (1.0f / kMlpHidden);
    b.inv_s = __builtin_amdgcn_rsqf(var + kLayerNormEps);
    b.y = z.v * splat4(b.inv_s);
    b.q = rows_sum(dot4(b.y, z.t)) * (1.0f / kMlpHidden);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float y = b.y[j];
        const Gauss n = gauss(y);
        b.g1[j] = n.cdf + y * n.pdf;
        b.g2[j] = n.pdf * (2.0f - y * y);
        b.a[j] = y * n.cdf;
        b.dy[j] = (z.t[j] - y * b.q) * b.inv_s;
    }
}

// Linear(16 -> 16) on a single-tangent jet.
__device__ __forceinline__ void linear_tile(const float (&a)[4], f32x4 bias, f32x4 in_v, f32x4 in_t, TileJet1& out) {
    out.v = bias;
    out.t = splat4(0.0f);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        out.v = mfma4(a[s], in_v[s], out.v);
        out.t = mfma4(a[s], in_t[s], out.t);
    }
}

// P(v) = (v - mean(v) - y mean(v y)) / s  (LayerNorm's symmetric Jacobian), in place on a tile.
__device__ __forceinline__ void layer_norm_adjoint_tile(f32x4& v, const TileState& b) {
    const float m = rows_sum(hsum4(v)) * (1.0f / kMlpHidden);
    const float my = rows_sum(dot4(v, b.y)) * (1.0f / kMlpHidden);
    v = (v - splat4(m) - b.y * splat4(my)) * splat4(b.inv_s);
}

// Adjoint of [LayerNorm -> GELU] on a single-tangent jet: (a_bar, da_bar) of the activations -> adjoint zb of the block's input jet.
// With P the LayerNorm Jacobian above, P(dz) = dy, so the input tangent itself is not needed:
//   z_bar = P(y_bar - dyb q / s) - dy mean(dyb y) / s + y s_bar / 16,   dz_bar = P(dyb).
__device__ __forceinline__ void gelu_norm_adjoint_tile(const TileState& b, f32x4 a_bar, f32x4 da_bar, TileJet1& zb) {
    f32x4 y_bar, dyb;
    float s_part = 0.0f, dot_part = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        dyb[j] = da_bar[j] * b.g1[j];
        s_part -= dyb[j] * b.dy[j];
        dot_part += dyb[j] * b.y[j];
        y_bar[j] = a_bar[j] * b.g1[j] + da_bar[j] * b.dy[j] * b.g2[j] - dyb[j] * b.q * b.inv_s;
    }
    const float s_bar = rows_sum(s_part) * b.inv_s;
    const float dot = rows_sum(dot_part) * (b.inv_s * (1.0f / kMlpHidden));
    layer_norm_adjoint_tile(y_bar, b);
    layer_norm_adjoint_tile(dyb, b);
    zb.v = y_bar + b.y * splat4(s_bar * (1.0f / kMlpHidden)) - b.dy * splat4(dot);
    zb.t = dyb;
}

__device__ __forceinline__ float pick4(f32x4 v, int m) { return (m == 0) ? v[0] : ((m == 1) ? v[1] : ((m == 2) ? v[2] : v[3])); }
__device__ __forceinline__ f32x4 row_sum16(f32x4 v) { return f32x4{row_sum16(v[0]), row_sum16(v[1]), row_sum16(v[2]), row_sum16(v[3])}; }

struct ResidualAdjoint { float px, py, pz; };

// ---- the adjoint proper, in pieces, so that a caller can keep the weight-adjoint accumulators in registers over MANY point sets of
// one instance (residual_mlp_adjoint_kernel, render_kernels.h) or over one (residual_backward) -----------------------------------------
struct MlpAdjoint {
    f32x4 acc_w[3], acc_w0[3], acc_b[3], acc_b0, acc_w4;    // W_bar[4 g + j][lane & 15] of the three hidden blocks / the three coordinate blocks
    float acc_b4;                                            // of the first layer; bias / head adjoints per point (summed over points at the end)
    __device__ __forceinline__ void clear() {
        acc_b0 = splat4(0.0f); acc_w4 = splat4(0.0f); acc_b4 = 0.0f;
#pragma unroll
        for (int l = 0; l < 3; ++l) { acc_w[l] = splat4(0.0f); acc_w0[l] = splat4(0.0f); acc_b[l] = splat4(0.0f); }
    }
};

// Where the MFMA weight operands come from.  RegisterWeights: 69 registers per lane, loaded once per point set (the wave-private
// callers).  LdsWeights: the instance's 1617 weights staged once per work item in the workgroup's LDS (centred: stage_centred_weights)
// and read operand by operand where they are used -- what lets residual_mlp_adjoint_kernel fit 256 registers (two waves per SIMD).
struct RegisterWeights {
    ForwardWeights fw;
    BackwardWeights bw;
    __device__ __forceinline__ float a0(int c, int k) const { return fw.a0[4 * c + k]; }      // W0[o][16 c + 4 g + k]
    __device__ __forceinline__ f32x4 b0() const { return fw.b0; }
    __device__ __forceinline__ float a(int l, int k) const { return fw.a[l][k]; }             // W_l[o][4 g + k]
    __device__ __forceinline__ f32x4 b(int l) const { return fw.b[l]; }
    __device__ __forceinline__ f32x4 w4() const { return fw.w4; }
    __device__ __forceinline__ float b4() const { return fw.b4; }
    __device__ __forceinline__ float at(int l, int k) const { return bw.at[l][k]; }           // W_l[4 g + k][i]
    __device__ __forceinline__ float at0(int c, int k) const { return bw.at0[c][k]; }         // W0[4 g + k][16 c + i]
};

struct LdsWeights {
    LdsFloats w;      // [1617] row-major, the four LayerNorm-fed linears centred
    int g, o;         // lane >> 4, lane & 15
    __device__ __forceinline__ float a0(int c, int k) const { return w[o * kMlpRow0 + 16 * c + 4 * g + k]; }
    __device__ __forceinline__ f32x4 b0() const { return f32x4{w[(4 * g + 0) * kMlpRow0 + kMlpFeatures], w[(4 * g + 1) * kMlpRow0 + kMlpFeatures],
                                                              w[(4 * g + 2) * kMlpRow0 + kMlpFeatures], w[(4 * g + 3) * kMlpRow0 + kMlpFeatures]}; }
    __device__ __forceinline__ float a(int l, int k) const { return w[kMlpLayer1 + l * kMlpBlock + o * kMlpRow + 4 * g + k]; }
    __device__ __forceinline__ f32x4 b(int l) const {
        const LdsFloats wl = w + kMlpLayer1 + l * kMlpBlock + kMlpHidden;
        return f32x4{wl[(4 * g + 0) * kMlpRow], wl[(4 * g + 1) * kMlpRow], wl[(4 * g + 2) * kMlpRow], wl[(4 * g + 3) * kMlpRow]};
    }
    __device__ __forceinline__ f32x4 w4() const { return f32x4{w[kMlpHead + 4 * g], w[kMlpHead + 4 * g + 1], w[kMlpHead + 4 * g + 2], w[kMlpHead + 4 * g + 3]}; }
    __device__ __forceinline__ float b4() const { return w[kMlpHead + kMlpHidden]; }
    __device__ __forceinline__ float at(int l, int k) const { return w[kMlpLayer1 + l * kMlpBlock + (4 * g + k) * kMlpRow + o]; }
    __device__ __forceinline__ float at0(int c, int k) const { return w[(4 * g + k) * kMlpRow0 + 16 * c + o]; }
};

// Stage one instance's weights into LDS with the centring of load_forward_weights applied (column means over the 16 output channels
// of the first linear and the three hidden ones removed, bias column included; skipped when the caller centred them for the launch).
// The transposed operands read the same copy: the adjoints they multiply have zero channel mean, so W^T z_bar = (C W)^T z_bar.
// Cooperative over `threads` threads of a workgroup; the caller synchronises before and after.
__device__ __forceinline__ void stage_centred_weights(LdsFloats dst, const float* __restrict__ w, bool centred, int thread, int threads) {
    for (int idx = thread; idx < kMlpWeights; idx += threads) dst[idx] = w[idx];
    if (centred) return;
    __syncthreads();
    for (int column = thread; column < kMlpRow0 + 3 * kMlpRow; column += threads) {       // 49 + 3 x 17 columns, 16 rows each
        const bool first = column < kMlpRow0;
        const int l = first ? 0 : (column - kMlpRow0) / kMlpRow, col = first ? column : (column - kMlpRow0) % kMlpRow;
        const int base = first ? col : kMlpLayer1 + l * kMlpBlock + col, pitch = first ? kMlpRow0 : kMlpRow;
        float mean = 0.0f;
        for (int r = 0; r < kMlpHidden; ++r) mean += dst[base + r * pitch];
        mean *= 1.0f / kMlpHidden;
        for (int r = 0; r < kMlpHidden; ++r) dst[base + r * pitch] -= mean;
    }
}

// The same by the 64 lanes of one wave (single-wave workgroups).
__device__ __forceinline__ void stage_centred_weights_wave(LdsFloats dst, const float* __restrict__ w, bool centred, int lane) {
    for (int idx = lane; idx < kMlpWeights; idx += kWave) dst[idx] = w[idx];
    if (centred) return;
    wave_lds_order();
    for (int column = lane; column < kMlpRow0 + 3 * kMlpRow; column += kWave) {
        const bool first = column < kMlpRow0;
        const int l = first ? 0 : (column - kMlpRow0) / kMlpRow, col = first ? column : (column - kMlpRow0) % kMlpRow;
        const int base = first ? col : kMlpLayer1 + l * kMlpBlock + col, pitch = first ? kMlpRow0 : kMlpRow;
        float mean = 0.0f;
        for (int r = 0; r < kMlpHidden; ++r) mean += dst[base + r * pitch];
        mean *= 1.0f / kMlpHidden;
        for (int r = 0; r < kMlpHidden; ++r) dst[base + r * pitch] -= mean;
    }
}

__device__ __forceinline__ void load_register_weights(RegisterWeights& w, const float* w_in, int lane, bool centred) {
    const GlobalWeights global = uniform_weights(w_in);
    load_forward_weights(global, lane, w.fw, centred);
    load_backward_weights(global, lane, w.bw);
}

// residual(p) and d residual / d p for the local positions p of the wave's 64 points, instance weights w (wave-uniform).
// Value by one forward column, gradient by one REVERSE column (oracle/analytic_mlp.py: forward_reverse) instead of three forward
// tangents:  a_bar_3 = w4,  z_bar_l = P_l(a_bar_l * gelu'(y_l)),  a_bar_{l-1} = W_l^T z_bar_l,  feat_bar = W_0^T z_bar_0,
// d out / d f_c = sum_j feat_bar_{c,j} dfeat_{c,j}: 64 instead of 92 MFMAs per 16-point tile, and the LayerNorm / GELU algebra on two
// columns instead of four.
// NOT inlined: one copy per kernel keeps the residual kernels' code inside the instruction cache.
// `tiles_in`: wave-uniform mask of the 16-point tiles to evaluate; lanes of the other tiles return 0.
// `weights_lds`: kMlpWbarFloats floats of this wave's LDS.  The instance's weights are staged there (26 coalesced loads per lane
// instead of 69 gathers of 64 scattered addresses each) and read operand by operand: held in registers they would cost the CALLERS
// their second wave per SIMD (a callee's registers, and the AGPRs it parks callee-saved ones in, count towards its callers).
// One 16-point tile of residual_forward on the exact-fp32 matrix instruction: value `v` and the three tangents `t` of the MLP output
// w.r.t. the folded, scaled coordinates (every lane of the tile's column receives them).
__device__ __forceinline__ void forward_tile_f32(const LdsWeights& wt, const TileFeatures& e, float& v, float (&t)[3]) {
    // ---- forward column ---------------------------------------------------------------------------------------------------
    f32x4 z = wt.b0();
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) z = mfma4(wt.a0(c, k), e.f[c][k], z);
    f32x4 y[4], g1[4];
    float inv_s[4];
    v = 0.0f;
#pragma unroll
    for (int l = 0; l < 4; ++l) {                            // z has zero channel mean (load_forward_weights)
        const float var = rows_sum(dot4(z, z)) * (1.0f / kMlpHidden);
        inv_s[l] = __builtin_amdgcn_rsqf(var + kLayerNormEps);
        y[l] = z * splat4(inv_s[l]);
        f32x4 a;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const Gauss n = gauss(y[l][j]);
            a[j] = y[l][j] * n.cdf;
            g1[l][j] = n.cdf + y[l][j] * n.pdf;
        }
        if (l < 3) {
            z = wt.b(l);
#pragma unroll
            for (int k = 0; k < 4; ++k) z = mfma4(wt.a(l, k), a[k], z);
        } else {
            v = rows_sum(dot4(wt.w4(), a)) + wt.b4();
        }
    }
    // ---- reverse column ---------------------------------------------------------------------------------------------------
    f32x4 a_bar = wt.w4(), z_bar;
#pragma unroll
    for (int l = 3; l >= 0; --l) {
        const f32x4 u = a_bar * g1[l];
        const float m = rows_sum(hsum4(u)) * (1.0f / kMlpHidden);
        const float my = rows_sum(dot4(u, y[l])) * (1.0f / kMlpHidden);
        z_bar = (u - splat4(m) - y[l] * splat4(my)) * splat4(inv_s[l]);
        if (l > 0) {
            a_bar = splat4(0.0f);
#pragma unroll
            for (int k = 0; k < 4; ++k) a_bar = mfma4(wt.at(l - 1, k), z_bar[k], a_bar);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f32x4 feat_bar = splat4(0.0f);
#pragma unroll
        for (int k = 0; k < 4; ++k) feat_bar = mfma4(wt.at0(c, k), z_bar[k], feat_bar);
        t[c] = rows_sum(dot4(feat_bar, e.d[c]));
    }
}

// ---- the same MLP on v_mfma_f32_16x16x32_bf16 with BOTH operands split into two bfloat16 parts (round 5; VSRD_FLAG_MLP_SPLIT_BF16) --------
// x = hi + lo, hi = bf16(x), lo = bf16(x - hi), both rounded to nearest even: |x - hi - lo| <= 2^-18 |x|.  The 16 x 16 layers have K = 16
// channels and the instruction contracts K = 32 per lane group of four channels: a lane's eight slots hold [part A of its four channels |
// part B of its four channels].  With the weight (A) operand [w_hi(4) | w_lo(4)] -- four words straight from the instance's operand
// image -- and the activation (B) operand [a_hi(4) | a_hi(4)], then [a_lo(4) | a_lo(4)], two instructions give all four products
// (w_hi + w_lo)(a_hi + a_lo) in fp32 accumulation: 2 x 16.6 cycles where the exact-fp32 form takes 4 x 32.5 (tools/micro/
// mfma_valu_interleave.hip; in a stream of vector instructions the fp32 form costs another ~10 cycles per switch), plus 16 vector
// instructions for the split of a four-channel operand.  The C / D layout is that of the fp32 instruction, so layers still chain with no
// data movement, and everything between the products (LayerNorm, GELU, the row sums) is the fp32 code above.
// What it costs in accuracy, measured on the GPU under the residual goldens' tests with both product forms (tests/test_hip_render.py,
// test_hip_step.py: the `mlp_products` fixture), against the reference's goldens: labels within 1.1e-6 (tolerance 1e-4), gradients within 4.2e-4 of the largest entry (tolerance 5e-3; exact fp32: 1.5e-4).
// (tests/split_bf16_emulation.py, the CPU emulation that came first -- the oracle with both operands of every linear replaced by their
// two-part sums -- had bounded the move at 2.1e-6 / 7.8e-4.)
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using LdsWords = __attribute__((address_space(3))) unsigned*;

constexpr int kImgOperandWords = 4 * kWave;                  // one A operand: lane l reads words [4 l, 4 l + 4) = {hi(k0, k1), hi(k2, k3), lo(k0, k1), lo(k2, k3)}
constexpr int kImgFirst = 0;                                 // operands 0..2: W0[o][16 c + 4 g + s]   (first layer, coordinate block c)
constexpr int kImgHidden = 3;                                // operands 3..5: W_l[o][4 g + s]
constexpr int kImgHiddenT = 6;                               // operands 6..8: W_l[4 g + s][i]        (transposed: the reverse column)
constexpr int kImgOperands = 9;
constexpr int kImgTail = kImgOperands * kImgOperandWords;    // then fp32: b0[16] | b_l[16] x 3 | w4[16] | b4 | pad
constexpr int kImgTailFloats = 96;
constexpr int kMlpImageWords = kImgTail + kImgTailFloats;    // 2400 words (9.4 KB) per instance; lane-linear, so every operand read is conflict-free
constexpr int kMlpStageFloats = kMlpImageWords > kMlpWbarFloats ? kMlpImageWords : kMlpWbarFloats;   // a wave's staging area in kernels that take either form

// The code below is compiled only in the translation unit that defines VSRD_SPLIT_BF16 (csrc/split_front.hip): under the library's
// scheduler strategy (-amdgpu-sched-strategy=iterative-ilp) hipcc 7.2 crashes on it -- the iterative scheduler leaves the live intervals
// of this function inconsistent and the register allocator's spill-weight pass (VirtRegAuxInfo::isRematerializable) follows a copy to a
// value that is not there; which source shapes trigger it changes with every edit (the reverse column as a loop, the row sums as lane
// swaps ...).  split_front.hip holds the residual step's front kernels with these products and is compiled with the default strategy;
// api.hip (everything else, iterative-ilp) never sees this code.
#ifdef VSRD_SPLIT_BF16
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// {hi(x0, x1), hi(x2, x3), lo(x0, x1), lo(x2, x3)}
#ifdef VSRD_SPLIT_TRUNCATE
// Experiment (round 6, VERDICT r05 item 3 "cut the VALU count"; profiles/r06_c3_bf16/variants.txt): both parts TRUNCATED to bfloat16 -- the high
// half-words picked by v_perm_b32, no v_cvt_pk_bf16_f32 (a double-rate instruction: 5.6 cycles against 2.9, profiles/r05/split_rates.txt): six
// full-rate instructions per pair of values instead of two conversions + four; |x - hi - lo| <= 2^-16 |x| instead of 2^-18.  Measured: +2.9 % on
// config 3, +3.0 % on the batched native residual step, goldens inside their tolerances -- NOT the default: four times the product error for 3 %.
__device__ __forceinline__ unsigned pack_high_halves(float lo, float hi) {        // {bits 31..16 of lo, bits 31..16 of hi} as one word, lo in the low half
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ u32x4 split4(float x0, float x1, float x2, float x3) {
    const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u), r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
    const float r2 = x2 - __uint_as_float(__float_as_uint(x2) & 0xffff0000u), r3 = x3 - __uint_as_float(__float_as_uint(x3) & 0xffff0000u);
    return u32x4{pack_high_halves(x0, x1), pack_high_halves(x2, x3), pack_high_halves(r0, r1), pack_high_halves(r2, r3)};
}
#else
__device__ __forceinline__ u32x4 split4(float x0, float x1, float x2, float x3) {
    const unsigned h01 = cvt_pk_bf16(x0, x1), h23 = cvt_pk_bf16(x2, x3);
    const float r0 = x0 - __uint_as_float(h01 << 16), r1 = x1 - __uint_as_float(h01 & 0xffff0000u);
    const float r2 = x2 - __uint_as_float(h23 << 16), r3 = x3 - __uint_as_float(h23 & 0xffff0000u);
    return u32x4{h01, h23, cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3)};
}
#endif
__device__ __forceinline__ u32x4 split4(f32x4 x) { return split4(x[0], x[1], x[2], x[3]); }
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// c += W x, W = an image operand, x = the split `s` = [x_hi | x_lo] of the lane's four channels.  The operand comes in both orders of
// its halves, a = [w_hi | w_lo] and swapped = [w_lo | w_hi] (the same 16 bytes of the image read a second time, half by half: no second
// image): a . s = w_hi x_hi + w_lo x_lo, swapped . s = w_lo x_hi + w_hi x_lo -- all four partial products with the split used AS IT IS.
// kDup: the first form, a . [x_hi | x_hi] + a . [x_lo | x_lo], whose two activation operands cost four register moves per product (144 of
// the 1 930 vector instructions of an adjoint tile) but which needs four registers less per live operand.
// -DVSRD_SPLIT_DUP: that form everywhere; -DVSRD_SPLIT_DUP_FORWARD: in the forward tiles of the front kernels only.
struct SplitOperand { u32x4 a, swapped; };
#ifdef VSRD_SPLIT_DUP
constexpr bool kSplitDupForward = true, kSplitDupAdjoint = true;
#elif defined(VSRD_SPLIT_DUP_FORWARD)
constexpr bool kSplitDupForward = true, kSplitDupAdjoint = false;
#else
constexpr bool kSplitDupForward = false, kSplitDupAdjoint = false;
#endif
template <bool kDup>
__device__ __forceinline__ f32x4 mfma_split(const SplitOperand& w, u32x4 s, f32x4 c) {
    if (kDup) {
        c = mfma_bf16(w.a, u32x4{s[0], s[1], s[0], s[1]}, c);
        return mfma_bf16(w.a, u32x4{s[2], s[3], s[2], s[3]}, c);
    }
    c = mfma_bf16(w.a, s, c);
    return mfma_bf16(w.swapped, s, c);
}

// One instance's operand image from its (centred) weights: what pack_mlp_images_kernel (render_kernels.h) runs per instance.
// `w`: the 1617 weights in LDS, centred (stage_centred_weights); thread `thread` of `threads`.
__device__ __forceinline__ void pack_mlp_image(const LdsFloats w, unsigned* __restrict__ image, int thread, int threads) {
    for (int idx = thread; idx < kImgOperands * kWave; idx += threads) {
        const int p = idx / kWave, lane = idx % kWave, g = lane >> 4, o = lane & 15;
        float x[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (p < kImgHidden) x[s] = w[o * kMlpRow0 + 16 * (p - kImgFirst) + 4 * g + s];
            else if (p < kImgHiddenT) x[s] = w[kMlpLayer1 + (p - kImgHidden) * kMlpBlock + o * kMlpRow + 4 * g + s];
            else x[s] = w[kMlpLayer1 + (p - kImgHiddenT) * kMlpBlock + (4 * g + s) * kMlpRow + o];
        }
        const u32x4 words = split4(x[0], x[1], x[2], x[3]);
        *reinterpret_cast<u32x4*>(image + p * kImgOperandWords + 4 * lane) = words;
    }
    float* tail = reinterpret_cast<float*>(image + kImgTail);
    for (int idx = thread; idx < kImgTailFloats; idx += threads) {
        float value = 0.0f;
        if (idx < 16) value = w[idx * kMlpRow0 + kMlpFeatures];
        else if (idx < 64) value = w[kMlpLayer1 + ((idx - 16) >> 4) * kMlpBlock + ((idx - 16) & 15) * kMlpRow + kMlpHidden];
        else if (idx < 80) value = w[kMlpHead + (idx - 64)];
        else if (idx == 80) value = w[kMlpHead + kMlpHidden];
        tail[idx] = value;
    }
}

struct SplitWeights {
    LdsWords image;    // the staged operand image of the instance
    int lane, g;
    template <bool kDup>
    __device__ __forceinline__ SplitOperand operand(int p) const {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const __attribute__((address_space(3))) u32x2* at = reinterpret_cast<const __attribute__((address_space(3))) u32x2*>(image + p * kImgOperandWords + 4 * lane);
        const u32x2 hi = at[0], lo = at[1];
        if (kDup) return {u32x4{hi[0], hi[1], lo[0], lo[1]}, u32x4{0u, 0u, 0u, 0u}};
        // The swapped operand: a second pair of reads through addresses the optimiser cannot see through -- otherwise it keeps the first
        // pair and builds the swapped operand with four register moves, the moves this form is there to avoid; each half through its
        // own address, because a merged ds_read2_b64 returns the halves in address order and the moves are back.
        unsigned again_lo = static_cast<unsigned>(reinterpret_cast<unsigned long>(at + 1)), again_hi = static_cast<unsigned>(reinterpret_cast<unsigned long>(at));
        asm volatile("" : "+v"(again_lo));
        asm volatile("" : "+v"(again_hi));
        const u32x2 lo2 = *reinterpret_cast<const __attribute__((address_space(3))) u32x2*>(static_cast<unsigned long>(again_lo));
        const u32x2 hi2 = *reinterpret_cast<const __attribute__((address_space(3))) u32x2*>(static_cast<unsigned long>(again_hi));
        return {u32x4{hi[0], hi[1], lo[0], lo[1]}, u32x4{lo2[0], lo2[1], hi2[0], hi2[1]}};
    }
    __device__ __forceinline__ f32x4 tail4(int at) const {
        return *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(image + kImgTail + at + 4 * g);      // channels 4 g .. 4 g + 3
    }
    __device__ __forceinline__ f32x4 b0() const { return tail4(0); }
    __device__ __forceinline__ f32x4 b(int l) const { return tail4(16 + 16 * l); }
    __device__ __forceinline__ f32x4 w4() const { return tail4(64); }
    __device__ __forceinline__ float b4() const { return __uint_as_float(image[kImgTail + 80]); }
};

__device__ __forceinline__ void stage_image_wave(LdsWords dst, const float* image, int lane) {
    const unsigned long long bits = reinterpret_cast<unsigned long long>(image);
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(bits >> 32));
    const __attribute__((address_space(1))) u32x4* src = reinterpret_cast<const __attribute__((address_space(1))) u32x4*>((static_cast<unsigned long long>(hi) << 32) | lo);
    for (int idx = lane; idx < kMlpImageWords / 4; idx += kWave)
        *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(dst + 4 * idx) = src[idx];
}

// forward_tile_f32 on the split-bf16 products.  The first layer's tangent goes FORWARDS here (X_c = W0_c dfeat_c next to z = W0 feat:
// the same weight operand, read once), so t_c = sum over channels of z_bar0 . X_c needs no transposed first-layer operand -- 3 KB of LDS
// the MLP-adjoint kernel does not have -- for the same number of matrix instructions.
__device__ __forceinline__ void forward_tile_split(const SplitWeights& wt, const TileFeatures& e, float& v, float (&t)[3]) {
    f32x4 z = wt.b0();
    f32x4 x[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const SplitOperand a = wt.operand<kSplitDupForward>(kImgFirst + c);
        z = mfma_split<kSplitDupForward>(a, split4(e.f[c]), z);
        x[c] = mfma_split<kSplitDupForward>(a, split4(e.d[c]), splat4(0.0f));
    }
    f32x4 y[4], g1[4];
    float inv_s[4];
    v = 0.0f;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const float var = rows_sum(dot4(z, z)) * (1.0f / kMlpHidden);
        inv_s[l] = __builtin_amdgcn_rsqf(var + kLayerNormEps);
        y[l] = z * splat4(inv_s[l]);
        f32x4 a;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const Gauss n = gauss(y[l][j]);
            a[j] = y[l][j] * n.cdf;
            g1[l][j] = n.cdf + y[l][j] * n.pdf;
        }
        if (l < 3) z = mfma_split<kSplitDupForward>(wt.operand<kSplitDupForward>(kImgHidden + l), split4(a), wt.b(l));
        else v = rows_sum(dot4(wt.w4(), a)) + wt.b4();
    }
    f32x4 a_bar = wt.w4(), z_bar;
#pragma unroll
    for (int l = 3; l >= 0; --l) {
        const f32x4 u = a_bar * g1[l];
        const float m = rows_sum(hsum4(u)) * (1.0f / kMlpHidden);
        const float my = rows_sum(dot4(u, y[l])) * (1.0f / kMlpHidden);
        z_bar = (u - splat4(m) - y[l] * splat4(my)) * splat4(inv_s[l]);
        if (l > 0) a_bar = mfma_split<kSplitDupForward>(wt.operand<kSplitDupForward>(kImgHiddenT + l - 1), split4(z_bar), splat4(0.0f));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c] = rows_sum(dot4(z_bar, x[c]));
}

#endif  // VSRD_SPLIT_BF16

VSRD_RESIDUAL_FN Residual residual_forward(const float* w_in, float px, float py, float pz, unsigned tiles_in, float* weights_lds) {
    const int lane = lane_id();
    const int g = lane >> 4;
    const unsigned tiles = __builtin_amdgcn_readfirstlane(tiles_in);
    const int start = static_cast<int>((tiles >> kMlpStartShift) & 63u);      // tile q = lanes start + 16 q ... start + 16 q + 15 (mod 64): tile_plan, below
    const bool split = (tiles & kMlpSplitBit) != 0u;                           // (wave-uniform) w_in is the instance's operand image
    const LdsFloats staged = (LdsFloats)weights_lds;
    wave_lds_order();                                            // (the previous call's operand reads are done)
#ifdef VSRD_SPLIT_BF16
    if (split) stage_image_wave((LdsWords)weights_lds, w_in, lane);
    else
#endif
    stage_centred_weights_wave(staged, uniform_weights_generic(w_in), (tiles & kMlpCentredBit) != 0u, lane);
    wave_lds_order();
#ifdef VSRD_PROBE_STAGE_TWICE          // probe: what one staging costs (the difference to the normal build)
    asm volatile("" ::: "memory");
    stage_centred_weights_wave(staged, uniform_weights_generic(w_in) + (tiles >> 30), (tiles & kMlpCentredBit) != 0u, lane);
    wave_lds_order();
#endif
    const LdsWeights wt = {staged, g, lane & 15};
#ifdef VSRD_SPLIT_BF16
    const SplitWeights swt = {(LdsWords)weights_lds, lane, g};
#else
    (void)split;
#endif
    const float fold = (px > 0.0f) ? 1.0f : ((px < 0.0f) ? -1.0f : 0.0f);
    const float inv = 1.0f / kPositionScale;
    const float f0 = fabsf(px) * inv, f1 = py * inv, f2 = pz * inv;
    float out_v = 0.0f, out_t0 = 0.0f, out_t1 = 0.0f, out_t2 = 0.0f;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        if (!((tiles >> q) & 1u)) continue;
        TileFeatures e;
        encode_tile(from_row(f0, q, lane, start), from_row(f1, q, lane, start), from_row(f2, q, lane, start), g, e);
        float v, t[3];
#ifdef VSRD_SPLIT_BF16
        if (split) forward_tile_split(swt, e, v, t);
        else
#endif
        forward_tile_f32(wt, e, v, t);
        const bool mine = (g == q);                              // row q of tile q holds point 16 q + m = this lane
        out_v = mine ? v : out_v; out_t0 = mine ? t[0] : out_t0; out_t1 = mine ? t[1] : out_t1; out_t2 = mine ? t[2] : out_t2;
    }
    Residual r;
    r.value = ((tiles >> g) & 1u) ? fast_rcp(1.0f + fast_exp(-(out_v - 1.0f))) : 0.0f;
    float kappa = r.value * (1.0f - r.value) * inv;
    if (start != 0) {                                            // (wave-uniform) lane (g, m) holds the result of lane start + 16 g + m: send it home
        const int back = (lane - start) << 2;
        r.value = rotate_lanes(r.value, back); kappa = rotate_lanes(kappa, back);
        out_t0 = rotate_lanes(out_t0, back); out_t1 = rotate_lanes(out_t1, back); out_t2 = rotate_lanes(out_t2, back);
    }
    r.gx = kappa * out_t0 * fold; r.gy = kappa * out_t1; r.gz = kappa * out_t2;
    return r;
}

// ---- packing the points that need the MLP into as few 16-point tiles as possible (round 4) ------------------------------------------
// The MLP runs tile by tile (16 consecutive lanes); which lanes NEED an instance's residual is a ballot (`need`: the lanes on which the
// instance is not negligible, field.h culling).  Along a ray the needed samples of an instance are (nearly always) ONE run of
// consecutive samples, which straddles tile boundaries wherever it happens to start.  Counting the tiles from the run's first lane
// (residual_forward: `start`) packs it into ceil(run / 16) tiles: 9 % fewer tiles in pass 1 and 4 % in pass 2 at the mid schedule
// (tests/tile_statistics.py; gathering the needed lanes one by one would save 0.5 % more).  A need-set with holes that the plain row
// mask serves with fewer tiles keeps the plain mask (start = 0).
struct TilePlan { int start; unsigned tiles; };      // wave-uniform: count the tiles from lane `start`, evaluate those of the 4-bit mask
__device__ __forceinline__ TilePlan tile_plan(unsigned long long need) {
    TilePlan p;
    p.start = 0;
    p.tiles = ((need & 0xFFFFull) ? 1u : 0u) | ((need & 0xFFFF0000ull) ? 2u : 0u) | ((need & 0xFFFF00000000ull) ? 4u : 0u) |
              ((need & 0xFFFF000000000000ull) ? 8u : 0u);
#ifndef VSRD_NO_TILE_PACKING
    if (need != 0ull) {
        const int first = __builtin_ctzll(need);
        const int run = 64 - __builtin_clzll(need) - first;
        const int packed = (run + 15) >> 4;
        if (packed < __builtin_popcount(p.tiles)) { p.start = first; p.tiles = (1u << packed) - 1u; }
    }
#endif
    return p;
}

// Where lane `lane`'s point goes when the points of `need` are gathered into the leading columns of a 64-column row (the MLP adjoint's
// seeds): the lanes that need the instance keep their order in columns 0 .. count - 1, the others follow (a permutation of the 64
// columns, so every column is written).  `count` = the number of leading columns that matter.
__device__ __forceinline__ int packed_column(unsigned long long need, int lane, int& count) {
    const int rank = static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(need >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(need), 0u)));
    count = __builtin_popcountll(need);
    return ((need >> lane) & 1ull) ? rank : (count + lane - rank);
}
__device__ __forceinline__ unsigned tiles_of_count(int count) { return (1u << ((count + 15) >> 4)) - 1u; }      // 0 .. 64 points -> 4-bit tile mask

// residual_forward for the lanes of `need` (other lanes of an evaluated tile get their residual as well, lanes of skipped tiles 0).
// `bits`: kMlpCentredBit or 0.
__device__ __forceinline__ Residual residual_forward_packed(const float* w_in, float px, float py, float pz, unsigned long long need, unsigned bits,
                                                            float* weights_lds) {
    const TilePlan plan = tile_plan(need);
    return residual_forward(w_in, px, py, pz, plan.tiles | bits | (static_cast<unsigned>(plan.start) << kMlpStartShift), weights_lds);
}

// Register j of lane (g, m) of a staged tile, back in the layout stage_tile took it from.
__device__ __forceinline__ f32x4 unstage_tile(LdsFloats tile, int lane) {
    const LdsFloats src = tile + (lane >> 4) * 4 * kTilePitch + (lane & 15);
    return f32x4{src[0], src[kTilePitch], src[2 * kTilePitch], src[3 * kTilePitch]};
}

// Adjoint of residual_forward at the wave's local positions p: res_bar = dL/d residual, (gbx, gby, gbz) = dL/d(grad_p residual).
// Accumulates dL/dw into `s`; `scratch`: the wave's transposition scratch; returns dL/dp per lane.
// kStash: the encoder features and their tangent wait in LDS (tiles 4..9 of a 10-tile scratch) from the first layer to the first
// layer's adjoint instead of in 36 registers (they have to be staged there for the weight adjoint anyway).
template <bool kStash, typename Weights>
__device__ __forceinline__ ResidualAdjoint mlp_adjoint_points(MlpAdjoint& s, const Weights& wt, float px, float py, float pz, float res_bar,
                                                              float gbx, float gby, float gbz, LdsFloats scratch, int lane, unsigned tiles_in) {
    const int g = lane >> 4;
    const unsigned tiles = __builtin_amdgcn_readfirstlane(tiles_in);
    const float fold = (px > 0.0f) ? 1.0f : ((px < 0.0f) ? -1.0f : 0.0f);
    const float inv = 1.0f / kPositionScale;
    const float f0 = fabsf(px) * inv, f1 = py * inv, f2 = pz * inv;
    // the direction of the single tangent, per point: delta = gl_bar * fold / 100 (header comment of this section)
    const float d0 = gbx * fold * inv, d1 = gby * inv, d2 = gbz * inv;
    const float base = (g == 0) ? 1.0f : ((g == 1) ? 4.0f : ((g == 2) ? 16.0f : 64.0f));
    const float omega[2] = {base * kPi, 2.0f * base * kPi};
    constexpr int kFeatureTile = kStash ? 4 : 2;             // first of the six tiles (f_0, tangent_0, f_1, tangent_1, f_2, tangent_2)
    ResidualAdjoint mine = {0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        if (!((tiles >> q) & 1u)) continue;
        const float tfold = from_row(fold, q, lane);
        const float delta[3] = {from_row(d0, q, lane), from_row(d1, q, lane), from_row(d2, q, lane)};
        const float t_res_bar = from_row(res_bar, q, lane);
        // ---- forward, keeping the [LayerNorm -> GELU] state of every layer ----------------------------------------------------
        TileState st[4];
        TileFeatures e;                                          // (kStash: dead after the first layer)
        f32x4 tangent[3];                                        // d features / d epsilon along delta, per coordinate block
        {
            encode_tile(from_row(f0, q, lane), from_row(f1, q, lane), from_row(f2, q, lane), g, e);
#pragma unroll
            for (int c = 0; c < 3; ++c) tangent[c] = e.d[c] * splat4(delta[c]);
            TileJet1 z;
            z.v = wt.b0();
            z.t = splat4(0.0f);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = wt.a0(c, k);
                    z.v = mfma4(a, e.f[c][k], z.v);
                    z.t = mfma4(a, tangent[c][k], z.t);
                }
            }
            if (kStash) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    stage_tile(scratch + (kFeatureTile + 2 * c) * kTileFloats, e.f[c], lane);
                    stage_tile(scratch + (kFeatureTile + 2 * c + 1) * kTileFloats, tangent[c], lane);
                }
            }
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                tile_state(z, st[l]);
                const f32x4 in_t = st[l].da();
                z.v = wt.b(l);
                z.t = splat4(0.0f);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = wt.a(l, k);
                    z.v = mfma4(a, st[l].a[k], z.v);
                    z.t = mfma4(a, in_t[k], z.t);
                }
            }
            tile_state(z, st[3]);
        }
        const f32x4 w4 = wt.w4();
        const float out_v = rows_sum(dot4(w4, st[3].a)) + wt.b4();
        const float out_t = rows_sum(dot4(w4, st[3].da()));      // tangent of the MLP output along delta  (= kappa_bar of the 3-tangent form)
        // ---- sigmoid head:  L = res_bar * res + kappa * out_t,  res = sigmoid(out - 1),  kappa = res (1 - res) ------------------------------
        const float res = fast_rcp(1.0f + fast_exp(-(out_v - 1.0f)));
        const float kappa = res * (1.0f - res);
        const float zb_t = kappa;
        const float zb_v = (t_res_bar + out_t * (1.0f - 2.0f * res)) * kappa;
        // ---- block 4: LayerNorm -> GELU -> Linear(16 -> 1) -----------------------------------------------------------------
        TileJet1 zb;
        s.acc_w4 += splat4(zb_v) * st[3].a + splat4(zb_t) * st[3].da();
        s.acc_b4 += zb_v;
        gelu_norm_adjoint_tile(st[3], w4 * splat4(zb_v), w4 * splat4(zb_t), zb);
        // ---- blocks 3..1: LayerNorm -> GELU -> Linear(16 -> 16); zb is the adjoint of the linear's output ------------------
#pragma unroll
        for (int l = 2; l >= 0; --l) {
            s.acc_b[l] += zb.v;
            stage_tile(scratch + 0 * kTileFloats, zb.v, lane);
            stage_tile(scratch + 1 * kTileFloats, st[l].a, lane);
            stage_tile(scratch + 2 * kTileFloats, zb.t, lane);
            stage_tile(scratch + 3 * kTileFloats, st[l].da(), lane);
            wave_lds_order();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const f32x4 xt = fetch_tile(scratch + (2 * k) * kTileFloats, lane), yt = fetch_tile(scratch + (2 * k + 1) * kTileFloats, lane);
#pragma unroll
                for (int j = 0; j < 4; ++j) s.acc_w[l] = mfma4(xt[j], yt[j], s.acc_w[l]);
            }
            wave_lds_order();
            f32x4 a_bar = splat4(0.0f), da_bar = splat4(0.0f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = wt.at(l, k);
                a_bar = mfma4(a, zb.v[k], a_bar);
                da_bar = mfma4(a, zb.t[k], da_bar);
            }
            TileJet1 zin_bar;
            gelu_norm_adjoint_tile(st[l], a_bar, da_bar, zin_bar);
            zb = zin_bar;
        }
        // ---- first layer + encoder; zb is the adjoint of z[0] -----------------------------------------------------------------
        s.acc_b0 += zb.v;
        stage_tile(scratch + 0 * kTileFloats, zb.v, lane);
        stage_tile(scratch + 1 * kTileFloats, zb.t, lane);
        if (!kStash) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                stage_tile(scratch + (kFeatureTile + 2 * c) * kTileFloats, e.f[c], lane);
                stage_tile(scratch + (kFeatureTile + 2 * c + 1) * kTileFloats, tangent[c], lane);
            }
        }
        wave_lds_order();
        const f32x4 xv = fetch_tile(scratch + 0 * kTileFloats, lane), xt = fetch_tile(scratch + 1 * kTileFloats, lane);
        float f_bar[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const f32x4 ft = fetch_tile(scratch + (kFeatureTile + 2 * c) * kTileFloats, lane);
            const f32x4 dt = fetch_tile(scratch + (kFeatureTile + 2 * c + 1) * kTileFloats, lane);
            const f32x4 feat = kStash ? unstage_tile(scratch + (kFeatureTile + 2 * c) * kTileFloats, lane) : e.f[c];
            f32x4 feat_bar = splat4(0.0f), tangent_bar = splat4(0.0f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = wt.at0(c, k);
                s.acc_w0[c] = mfma4(xv[k], ft[k], s.acc_w0[c]);
                s.acc_w0[c] = mfma4(xt[k], dt[k], s.acc_w0[c]);
                feat_bar = mfma4(a, zb.v[k], feat_bar);
                tangent_bar = mfma4(a, zb.t[k], tangent_bar);
            }
            // features of this row: (cos, sin) of octaves 2 g and 2 g + 1;  d cos = -omega sin,  d sin = omega cos;
            // d (delta dfeat) / d f = -delta omega^2 feat
            float part = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const float co = feat[2 * kk], si = feat[2 * kk + 1], om = omega[kk];
                part += om * (feat_bar[2 * kk + 1] * co - feat_bar[2 * kk] * si) - (delta[c] * om * om) * (tangent_bar[2 * kk] * co + tangent_bar[2 * kk + 1] * si);
            }
            f_bar[c] = rows_sum(part);
        }
        wave_lds_order();
        if (g == q) { mine.px = f_bar[0] * tfold * inv; mine.py = f_bar[1] * inv; mine.pz = f_bar[2] * inv; }
    }
    return mine;
}

// Add the accumulators into a row-major weight row `dst` [1617] (LDS or global; every lane owns its addresses, so plain read-add-write)
// and clear them.
// kAssign: store instead of add (every one of the 1617 entries is owned by exactly one lane, so a row is fully written).
template <bool kAssign = false, typename Row>
__device__ __forceinline__ void mlp_adjoint_flush(MlpAdjoint& s, Row dst, int lane) {
    const int g = lane >> 4, m = lane & 15;
    auto put = [&](int index, float value) { dst[index] = kAssign ? value : dst[index] + value; };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int l = 0; l < 3; ++l) put(kMlpLayer1 + l * kMlpBlock + (4 * g + j) * kMlpRow + m, s.acc_w[l][j]);
#pragma unroll
        for (int c = 0; c < 3; ++c) put((4 * g + j) * kMlpRow0 + 16 * c + m, s.acc_w0[c][j]);
    }
    // biases / head weights: sum over the 16 points of the row, lane m < 4 of row g stores channel 4 g + m
    const float b0 = pick4(row_sum16(s.acc_b0), m), w4 = pick4(row_sum16(s.acc_w4), m);
    float bl[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) bl[l] = pick4(row_sum16(s.acc_b[l]), m);
    const float b4 = row_sum16(s.acc_b4);
    if (m < 4) {
        put((4 * g + m) * kMlpRow0 + kMlpFeatures, b0);
        put(kMlpHead + 4 * g + m, w4);
#pragma unroll
        for (int l = 0; l < 3; ++l) put(kMlpLayer1 + l * kMlpBlock + (4 * g + m) * kMlpRow + kMlpHidden, bl[l]);
    }
    if (lane == 0) put(kMlpHead + kMlpHidden, b4);
    s.clear();
}

// One point set of one instance, start to finish: adds dL/dw into `mlp_lds` (the wave's LDS: wbar [1617], then the transposition
// scratch); returns dL/dp per lane.  (residual_mlp_adjoint_kernel, with many point sets per instance, uses the pieces directly.)
__device__ __forceinline__ ResidualAdjoint residual_backward(const float* w_in, float px, float py, float pz,
                                                             float res_bar, float gbx, float gby, float gbz, float* mlp_lds, int lane, unsigned tiles_in) {
    const LdsFloats wbar = (LdsFloats)mlp_lds;
    MlpAdjoint s;
    s.clear();
    RegisterWeights wt;
    load_register_weights(wt, w_in, lane, (__builtin_amdgcn_readfirstlane(tiles_in) & kMlpCentredBit) != 0u);
    const ResidualAdjoint mine = mlp_adjoint_points<false>(s, wt, px, py, pz, res_bar, gbx, gby, gbz, wbar + kMlpWbarFloats, lane, tiles_in);
    mlp_adjoint_flush(s, wbar, lane);
    return mine;
}

#ifdef VSRD_SPLIT_BF16
// ---- mlp_adjoint_points on the split-bf16 products (csrc/split_front.hip only) ---------------------------------------------------------------
// Same algebra and tile layout as mlp_adjoint_points; what changes:
//   * every product W x / W^T z_bar is mfma_split on an operand of the instance's image (2 bf16 instructions for 4 fp32 ones);
//   * the weight adjoints  W_bar += z_bar a^T + dz_bar da^T  contract over the tile's 16 points AND the two columns: K = 32 is ONE bf16
//     instruction per pair of parts -- hi.hi, hi.lo, lo.hi, lo.lo: four instructions where the fp32 form takes eight; the operands are the
//     transposed tiles (fetch_tile, as before) split in registers;
//   * the first layer's input adjoint goes FORWARDS:  f_bar_c = z_bar . (W0_c dfeat_c) + dz_bar . (W0_c d(tangent_c)/df)  needs the forward
//     operand of the image only -- no transposed first-layer image, 3 KB of LDS the kernel does not have;
//   * the encoder features are recomputed where the first layer's adjoint needs them instead of waiting in six LDS tiles (or 36
//     registers): with the 9.4 KB image the kernel's LDS is image + four scratch tiles = 14.5 KB, eight workgroups per CU as before.
struct SplitPair { u32x4 hi, lo; };            // two columns of one transposed tile: {v_hi(4 points), t_hi(4 points)}, {v_lo, t_lo}
__device__ __forceinline__ SplitPair split_pair(f32x4 v, f32x4 t) {
    const u32x4 sv = split4(v), st = split4(t);
    return {u32x4{sv[0], sv[1], st[0], st[1]}, u32x4{sv[2], sv[3], st[2], st[3]}};
}
// acc[o][i] += sum over the tile's points p of  xv[o][p] yv[i][p] + xt[o][p] yt[i][p]
__device__ __forceinline__ f32x4 mfma_outer(const SplitPair& x, const SplitPair& y, f32x4 acc) {
    acc = mfma_bf16(x.hi, y.hi, acc);
    acc = mfma_bf16(x.hi, y.lo, acc);
    acc = mfma_bf16(x.lo, y.hi, acc);
#ifdef VSRD_SPLIT_OUTER_3        // experiment (round 6, profiles/r06_c3_bf16/variants.txt): without the lo.lo partial product (2^-16 of the term)
    return acc;
#else
    return mfma_bf16(x.lo, y.lo, acc);
#endif
}

__device__ __forceinline__ ResidualAdjoint mlp_adjoint_points_split(MlpAdjoint& s, const SplitWeights& wt, float px, float py, float pz, float res_bar,
                                                                    float gbx, float gby, float gbz, LdsFloats scratch, int lane, unsigned tiles_in) {
    const int g = lane >> 4;
    const unsigned tiles = __builtin_amdgcn_readfirstlane(tiles_in);
    const float fold = (px > 0.0f) ? 1.0f : ((px < 0.0f) ? -1.0f : 0.0f);
    const float inv = 1.0f / kPositionScale;
    const float f0 = fabsf(px) * inv, f1 = py * inv, f2 = pz * inv;
    const float d0 = gbx * fold * inv, d1 = gby * inv, d2 = gbz * inv;      // the direction of the single tangent (mlp_adjoint_points)
    const float base = (g == 0) ? 1.0f : ((g == 1) ? 4.0f : ((g == 2) ? 16.0f : 64.0f));
    const float omega[2] = {base * kPi, 2.0f * base * kPi};
    ResidualAdjoint mine = {0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        if (!((tiles >> q) & 1u)) continue;
        const float tfold = from_row(fold, q, lane);
        const float delta[3] = {from_row(d0, q, lane), from_row(d1, q, lane), from_row(d2, q, lane)};
        const float t_res_bar = from_row(res_bar, q, lane);
        const float tf[3] = {from_row(f0, q, lane), from_row(f1, q, lane), from_row(f2, q, lane)};
        // ---- forward, keeping the [LayerNorm -> GELU] state of every layer ----------------------------------------------------
        TileState st[4];
        {
            TileFeatures e;
            encode_tile(tf[0], tf[1], tf[2], g, e);
            TileJet1 z;
            z.v = wt.b0();
            z.t = splat4(0.0f);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const SplitOperand a = wt.operand<kSplitDupAdjoint>(kImgFirst + c);
                z.v = mfma_split<kSplitDupAdjoint>(a, split4(e.f[c]), z.v);
                z.t = mfma_split<kSplitDupAdjoint>(a, split4(e.d[c] * splat4(delta[c])), z.t);
            }
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                tile_state(z, st[l]);
                const SplitOperand a = wt.operand<kSplitDupAdjoint>(kImgHidden + l);
                z.v = mfma_split<kSplitDupAdjoint>(a, split4(st[l].a), wt.b(l));
                z.t = mfma_split<kSplitDupAdjoint>(a, split4(st[l].da()), splat4(0.0f));
            }
            tile_state(z, st[3]);
        }
        const f32x4 w4 = wt.w4();
        const float out_v = rows_sum(dot4(w4, st[3].a)) + wt.b4();
        const float out_t = rows_sum(dot4(w4, st[3].da()));
        const float res = fast_rcp(1.0f + fast_exp(-(out_v - 1.0f)));
        const float kappa = res * (1.0f - res);
        const float zb_t = kappa;
        const float zb_v = (t_res_bar + out_t * (1.0f - 2.0f * res)) * kappa;
        TileJet1 zb;
        s.acc_w4 += splat4(zb_v) * st[3].a + splat4(zb_t) * st[3].da();
        s.acc_b4 += zb_v;
        gelu_norm_adjoint_tile(st[3], w4 * splat4(zb_v), w4 * splat4(zb_t), zb);
        // ---- blocks 3..1 ----------------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int l = 2; l >= 0; --l) {
            s.acc_b[l] += zb.v;
            stage_tile(scratch + 0 * kTileFloats, zb.v, lane);
            stage_tile(scratch + 1 * kTileFloats, st[l].a, lane);
            stage_tile(scratch + 2 * kTileFloats, zb.t, lane);
            stage_tile(scratch + 3 * kTileFloats, st[l].da(), lane);
            wave_lds_order();
            {
                const SplitPair x = split_pair(fetch_tile(scratch + 0 * kTileFloats, lane), fetch_tile(scratch + 2 * kTileFloats, lane));
                const SplitPair y = split_pair(fetch_tile(scratch + 1 * kTileFloats, lane), fetch_tile(scratch + 3 * kTileFloats, lane));
                s.acc_w[l] = mfma_outer(x, y, s.acc_w[l]);
            }
            wave_lds_order();
            const SplitOperand at = wt.operand<kSplitDupAdjoint>(kImgHiddenT + l);
            const f32x4 a_bar = mfma_split<kSplitDupAdjoint>(at, split4(zb.v), splat4(0.0f));
            const f32x4 da_bar = mfma_split<kSplitDupAdjoint>(at, split4(zb.t), splat4(0.0f));
            TileJet1 zin_bar;
            gelu_norm_adjoint_tile(st[l], a_bar, da_bar, zin_bar);
            zb = zin_bar;
        }
        // ---- first layer + encoder; zb is the adjoint of z[0] -------------------------------------------------------------------------------
        s.acc_b0 += zb.v;
        stage_tile(scratch + 0 * kTileFloats, zb.v, lane);
        stage_tile(scratch + 1 * kTileFloats, zb.t, lane);
        TileFeatures e;
        encode_tile(tf[0], tf[1], tf[2], g, e);                      // (recomputed: see the header comment)
        wave_lds_order();
        const SplitPair x = split_pair(fetch_tile(scratch + 0 * kTileFloats, lane), fetch_tile(scratch + 1 * kTileFloats, lane));
        float f_bar[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            stage_tile(scratch + 2 * kTileFloats, e.f[c], lane);
            stage_tile(scratch + 3 * kTileFloats, e.d[c] * splat4(delta[c]), lane);
            wave_lds_order();
            {
                const SplitPair y = split_pair(fetch_tile(scratch + 2 * kTileFloats, lane), fetch_tile(scratch + 3 * kTileFloats, lane));
                s.acc_w0[c] = mfma_outer(x, y, s.acc_w0[c]);
            }
            wave_lds_order();
            // d feat / d f = e.d;  d (delta dfeat) / d f = -delta omega^2 feat
            const f32x4 second = {-delta[c] * omega[0] * omega[0] * e.f[c][0], -delta[c] * omega[0] * omega[0] * e.f[c][1],
                                  -delta[c] * omega[1] * omega[1] * e.f[c][2], -delta[c] * omega[1] * omega[1] * e.f[c][3]};
            const SplitOperand a = wt.operand<kSplitDupAdjoint>(kImgFirst + c);
            const f32x4 x1 = mfma_split<kSplitDupAdjoint>(a, split4(e.d[c]), splat4(0.0f));
            const f32x4 x2 = mfma_split<kSplitDupAdjoint>(a, split4(second), splat4(0.0f));
            f_bar[c] = rows_sum(dot4(zb.v, x1) + dot4(zb.t, x2));
        }
        if (g == q) { mine.px = f_bar[0] * tfold * inv; mine.py = f_bar[1] * inv; mine.pz = f_bar[2] * inv; }
    }
    return mine;
}
#endif  // VSRD_SPLIT_BF16

}  // namespace vsrd
