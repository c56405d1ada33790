// Per-instance oriented-box SDF, its analytic gradient, and the online temperature soft-min union.
//
// Reference semantics (see oracle/fields.py for the CPU restatement):
//   box / rotation / translation           vsrd/rendering/sdfs.py:5-37
//   instance_field + soft_union            scripts/main.py:460-492
//   grad = autograd.grad(sdf, positions)   vsrd/rendering/renderers.py:218-228  (analytic here)
//
// Mapping: the instance loop is WAVE-UNIFORM (every lane evaluates instance i at its own sample),
// so the 16 packed floats of instance i are fetched with scalar loads and used as SGPR operands.
#pragma once
#include "wave.h"
#include "residual.h"

namespace vsrd {

constexpr int kInstanceStride = 16;  // VSRD_INSTANCE_STRIDE
constexpr float kNormEpsilon = 1.0e-6f;  // sdfs.py:5

struct Instance {
    float tx, ty, tz;
    float r00, r01, r02, r10, r11, r12, r20, r21, r22;
    float dx, dy, dz;
};

__device__ __forceinline__ Instance load_instance(const float* __restrict__ instances, int i) {
    const float* p = instances + i * kInstanceStride;  // uniform address -> s_load_dwordx16
    Instance v;
    v.tx = p[0]; v.ty = p[1]; v.tz = p[2];
    v.r00 = p[3]; v.r01 = p[4]; v.r02 = p[5];
    v.r10 = p[6]; v.r11 = p[7]; v.r12 = p[8];
    v.r20 = p[9]; v.r21 = p[10]; v.r22 = p[11];
    v.dx = p[12]; v.dy = p[13]; v.dz = p[14];
    return v;
}

// A/B experiment (-DVSRD_INSTANCE_VGPR, box-only loops): the parameters as VGPRs (vector loads of one address) instead of SGPR
// operands -- a VALU instruction with an SGPR operand issues in 4.3 cycles, the same one on VGPRs in 2.5 (tools/micro/pk_rate.hip).
template <bool kVector>
__device__ __forceinline__ Instance load_instance_as(const float* __restrict__ instances, int i) {
#ifdef VSRD_INSTANCE_VGPR
    if (kVector) {
        int zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
        const float4* p = reinterpret_cast<const float4*>(instances + i * kInstanceStride + zero);
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        Instance v;
        v.tx = a.x; v.ty = a.y; v.tz = a.z;
        v.r00 = a.w; v.r01 = b.x; v.r02 = b.y;
        v.r10 = b.z; v.r11 = b.w; v.r12 = c.x;
        v.r20 = c.y; v.r21 = c.z; v.r22 = c.w;
        v.dx = d.x; v.dy = d.y; v.dz = d.z;
        return v;
    }
#endif
    return load_instance(instances, i);
}

// Everything phase B of the backward needs about one (sample, instance) pair.
struct BoxEval {
    float relx, rely, relz;   // x - t
    float px, py, pz;         // local position  p = (x - t) @ R
    float qx, qy, qz;         // |p| - half extents
    float hx, hy, hz;         // d d / d q
    float nrm;                // sqrt(sum relu(q)^2 + 1e-6)
    float inv;                // 1 / nrm
    float qmax;               // max_j q_j
    float d;                  // signed distance
    float glx, gly, glz;      // local gradient  sign(p) * h
    float gwx, gwy, gwz;      // world gradient  R @ gl
};

__device__ __forceinline__ float sign_of(float v) { return (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f); }


// First half of the box evaluation: local position and signed distance (what the exact culling test of the render loops needs).
// kYaw: the rotation is one about the y axis with r01 = r10 = r12 = r21 = 0 and r11 = 1 EXACTLY (what rotation_matrix_y,
// box_parameters.py:5-13, produces; field_bounds checks it for the whole field): the products with those entries are left out,
// which changes the results by rounding at most (x * 0 = 0, x * 1 = x) and saves 5 of the 9 multiply-adds of each rotation.
template <bool kYaw = false>
__device__ __forceinline__ BoxEval box_value(const Instance& in, float x, float y, float z) {
    BoxEval e;
    e.relx = x - in.tx; e.rely = y - in.ty; e.relz = z - in.tz;
    // row vector times R (sdfs.py:34): p_j = sum_k rel_k R_kj
    // (explicit fma chains: every instantiation rounds the local position the same way -- the box SDF has kinks (arg max, relu),
    //  and a last-bit difference in p moves samples across them, which shows up as 1e-4 differences in summed gradients)
    if (kYaw) {
        e.px = fmaf(e.relz, in.r20, e.relx * in.r00);
        e.py = e.rely;
        e.pz = fmaf(e.relz, in.r22, e.relx * in.r02);
    } else {
        e.px = fmaf(e.relz, in.r20, fmaf(e.rely, in.r10, e.relx * in.r00));
        e.py = fmaf(e.relz, in.r21, fmaf(e.rely, in.r11, e.relx * in.r01));
        e.pz = fmaf(e.relz, in.r22, fmaf(e.rely, in.r12, e.relx * in.r02));
    }
    e.qx = fabsf(e.px) - in.dx; e.qy = fabsf(e.py) - in.dy; e.qz = fabsf(e.pz) - in.dz;
    const float ax = fmaxf(e.qx, 0.0f), ay = fmaxf(e.qy, 0.0f), az = fmaxf(e.qz, 0.0f);
    // |v| = s rsq(s) and 1 / |v| = rsq(s): ONE transcendental per evaluation instead of v_sqrt + v_rcp (a quarter-rate instruction costs
    // three multiply-adds).  Both forms are within 1.5 ulp of the exact norm; round 4: config 2 229 -> 234 Mrays/s, config 5 30.1 -> 30.7.
    // -DVSRD_BOX_SQRT_RCP: the form of rounds 1-3.
#ifndef VSRD_BOX_SQRT_RCP
    const float squares = ax * ax + ay * ay + az * az + kNormEpsilon;
    e.inv = fast_rsq(squares);
    e.nrm = squares * e.inv;
#ifndef VSRD_NO_NORM_PIN
    asm volatile("" : "+v"(e.nrm));      // (a rounded product in EVERY instantiation: without this some of them fuse it into the subtraction below,
                                         //  and kernels that differ in the last bit of a distance differ by 1e-4 in summed gradients -- see above)
#endif
#else
    e.nrm = fast_sqrt(ax * ax + ay * ay + az * az + kNormEpsilon);
    e.inv = fast_rcp(e.nrm);
#endif
    e.qmax = fmaxf(fmaxf(e.qx, e.qy), e.qz);                            // one v_max3_f32
    e.d = e.nrm - fmaxf(-e.qmax, 0.0f);
    return e;
}

__device__ __forceinline__ float box_inverse_norm(const BoxEval& e) {
    return e.inv;
}

// gw_k = sum_j R_kj gl_j
template <bool kYaw>
__device__ __forceinline__ void rotate_to_world(BoxEval& e, const Instance& in) {
    if (kYaw) {
        e.gwx = fmaf(in.r02, e.glz, in.r00 * e.glx);
        e.gwy = e.gly;
        e.gwz = fmaf(in.r22, e.glz, in.r20 * e.glx);
    } else {
        e.gwx = fmaf(in.r02, e.glz, fmaf(in.r01, e.gly, in.r00 * e.glx));
        e.gwy = fmaf(in.r12, e.glz, fmaf(in.r11, e.gly, in.r10 * e.glx));
        e.gwz = fmaf(in.r22, e.glz, fmaf(in.r21, e.gly, in.r20 * e.glx));
    }
}

// Second half: analytic gradient in the local and the world frame.
//   d d / d q_j = relu(q_j) / nrm + [inside] [j = arg max q]      (torch.max: the FIRST maximal index; its backward routes there)
//   gl_j        = sign(p_j) * that                                (torch.sign(0) = 0)
// Inside the box (q_max < 0) every relu(q_j) is 0, and p_j = 0 makes relu(q_j) = 0 (half extents are positive): per axis the
// magnitude is one select between relu(q_j)/nrm and 1, the select mask carries "p_j != 0" (lane-mask logic on the scalar unit), and
// the sign is a bit-field insert; the arg max is a v_max3 and two compares.  (Six vector instructions fewer than the obvious
// form, and no faster: the fused step kernel is bound by its multiply-add stream, tools/micro/op_rates.hip.)
template <bool kYaw = false>
__device__ __forceinline__ void box_gradient(BoxEval& e, const Instance& in) {
    const float inv = box_inverse_norm(e);
    const bool inside = e.qmax < 0.0f;
    const bool first_x = e.qx == e.qmax;
    const bool first_y = !first_x && (e.qy == e.qmax);
    const bool one_x = inside && first_x, one_y = inside && first_y, one_z = inside && !first_x && !first_y;
    const float tx = fmaxf(e.qx, 0.0f) * inv, ty = fmaxf(e.qy, 0.0f) * inv, tz = fmaxf(e.qz, 0.0f) * inv;
    e.hx = one_x ? 1.0f : tx; e.hy = one_y ? 1.0f : ty; e.hz = one_z ? 1.0f : tz;      // (only the adjoint uses h itself)
    e.glx = __builtin_copysignf((one_x && e.px != 0.0f) ? 1.0f : tx, e.px);
    e.gly = __builtin_copysignf((one_y && e.py != 0.0f) ? 1.0f : ty, e.py);
    e.glz = __builtin_copysignf((one_z && e.pz != 0.0f) ? 1.0f : tz, e.pz);
    rotate_to_world<kYaw>(e, in);
}

template <bool kYaw = false>
__device__ __forceinline__ BoxEval eval_box(const Instance& in, float x, float y, float z) {
    BoxEval e = box_value<kYaw>(in, x, y, z);
    box_gradient<kYaw>(e, in);
    return e;
}

// d_i = box(p) + residual(p), local gradient likewise (scripts/main.py:451-458).
template <bool kYaw = false>
__device__ __forceinline__ void add_residual(BoxEval& e, const Instance& in, const Residual& r) {
    e.d += r.value;
    e.glx += r.gx; e.gly += r.gy; e.glz += r.gz;
    rotate_to_world<kYaw>(e, in);
}

// Box (+ optional residual MLP) evaluation of instance i.  `mlp` is the wave-uniform weight row of the instance.
// `tiles`: wave-uniform 4-bit mask of the 16-lane rows that need the residual (rows_with(ballot of un-culled lanes)); the lanes
// of the other rows get residual 0 -- their soft-min weight is below exp(-tau) on the box distance alone (culling, below).
// `keep`: where to leave the residual jet for the adjoint (render_backward_kernel caches it instead of re-evaluating the MLP).
template <bool kResidual>
__device__ __forceinline__ BoxEval eval_instance(const Instance& in, const float* __restrict__ mlp, float x, float y, float z, float* mlp_lds,
                                                 unsigned tiles = 0xFu, Residual* keep = nullptr) {
    BoxEval e = eval_box(in, x, y, z);
    if (kResidual) {
        const Residual r = residual_forward(mlp, e.px, e.py, e.pz, tiles, mlp_lds);      // wave-cooperative: all 64 lanes active
        add_residual(e, in, r);
        if (keep) *keep = r;
    }
    return e;
}

// ---- conservative instance culling -------------------------------------------------------------------
// For a sample x, instance i's soft-min weight relative to the best instance is at most
//   exp(-(LB_i - UB) / T),  LB_i = |x - t_i| (1 - k) - |dim_i|  <=  d_i(x),   UB = min_j |x - t_j| (1 + k) + 2e-3  >=  min_j d_j(x)
// (a box lies inside its circumscribed sphere and contains its centre; k = 2e-4 absorbs rotation matrices that are
// orthonormal only to ~1e-4).  When LB_i - UB > tau T on EVERY lane of the wave the instance is skipped for that round:
// with tau = 18 its weight is below exp(-18) = 1.5e-8 < 2^-24, i.e. below half an ulp of the soft-min normaliser.
// The decision is wave-uniform (one ballot), so a skipped instance costs six instructions instead of ~75.
// Second, exact stage (render loops): an instance that passes the bound test gets its box distance evaluated (value only, ~30
// instructions); if  d_i - min(UB, smallest box distance among the instances evaluated before it) > tau T  on every lane -- the same
// criterion with an exact left-hand side -- the instance is dropped for the round before its gradient, its residual MLP, its
// soft-min terms and, later, its label sum and both adjoint phases are computed.  (At T = 0.1 the sphere bounds keep 9 of 16
// instances per round of the benchmark scene, the exact test 3.5.)
constexpr float kCullTau = 18.0f;
constexpr float kCullSlack = 2.0e-4f;

// 4-bit mask of the 16-lane rows in which `ballot` has a lane set (scalar arithmetic on the ballot).
__device__ __forceinline__ unsigned rows_with(unsigned long long ballot) {
    return ((ballot & 0xFFFFull) ? 1u : 0u) | ((ballot & 0xFFFF0000ull) ? 2u : 0u) | ((ballot & 0xFFFF00000000ull) ? 4u : 0u) |
           ((ballot & 0xFFFF000000000000ull) ? 8u : 0u);
}

// The bound test itself works on SQUARED centre distances along the ray: x(t) = o + r t gives
//   |x(t) - t_i|^2 = a_i + b_i t + (r.r) t^2,   a_i = |o - t_i|^2,  b_i = 2 (o - t_i).r
// so a sample costs two FMAs per instance (a_i, b_i: per-ray values, wave-uniform LDS reads) instead of three subtractions,
// three multiply-adds and a square root, and the test  LB_i - UB > tau T  is compared in squares:
//   cull  <=>  d2_i - E > ((UB + tau T + |dim_i|) / (1 - k))^2,   UB = sqrt(min_j d2_j + E) (1 + k) + 2e-3
// E bounds the rounding error of the quadratic form AND of the sample position itself (x is computed as o + r t in fp32):
// both are below 6 * 2^-24 * (max(|o - t_i|, |o|) + |r| |t|)^2 = 3.6e-7 (...)^2; E = 2e-6 (...)^2.
constexpr float kCullQuadSlack = 2.0e-6f;
constexpr int kCullCoefs = 4;                      // a_i, b_i, |dim_i| / (1 - k), pad  (per instance, in the wave's LDS)

__host__ __device__ constexpr int cull_coef_floats(int num_instances) { return kCullCoefs * num_instances; }

struct RayCull {
    const float* coef;   // LDS [N][kCullCoefs]
    float c2;            // r.r
    float rnorm;         // |r|
    float reach;         // max(max_i |o - t_i|, |o|): scale of the error bound E
};

struct RoundCull {       // one round (64 samples) of one ray
    float err;           // E of this lane's sample
    float limit;         // (UB + margin) / (1 - k) of this lane's sample
    float nearest_hi;    // upper bound of the nearest centre distance (hence of the smallest box distance): sqrt(min_j d2_j + E) (1 + k)
    float nearest_lo;    // lower bound of the nearest centre distance of this lane's sample: sqrt(max(min_j d2_j - E, 0)) (1 - k)
};

// Per ray: lane i prepares instance i's coefficients (N <= 64 = VSRD_MAX_INSTANCES).
__device__ __forceinline__ RayCull cull_ray_setup(const float* __restrict__ instances, int num_instances, float ox, float oy, float oz,
                                                  float rx, float ry, float rz, float* coef, int lane) {
    float a = 0.0f;
    if (lane < num_instances) {
        const float* p = instances + lane * kInstanceStride;
        const float ex = ox - p[0], ey = oy - p[1], ez = oz - p[2];
        a = ex * ex + ey * ey + ez * ez;
        coef[kCullCoefs * lane + 0] = a;
        coef[kCullCoefs * lane + 1] = 2.0f * (ex * rx + ey * ry + ez * rz);
        coef[kCullCoefs * lane + 2] = fast_sqrt(p[12] * p[12] + p[13] * p[13] + p[14] * p[14]) * (1.0f / (1.0f - kCullSlack));
    }
    RayCull rc;
    rc.coef = coef;
    rc.c2 = rx * rx + ry * ry + rz * rz;
    rc.rnorm = fast_sqrt(rc.c2);
    rc.reach = fast_sqrt(fmaxf(wave_max(a), ox * ox + oy * oy + oz * oz));
    wave_lds_sync();
    return rc;
}

// Squared centre distance of instance i at ray parameter t (ct = c2 * t).
__device__ __forceinline__ float centre_distance2(const RayCull& rc, int i, float t, float ct) {
    return fmaf(t, ct + rc.coef[kCullCoefs * i + 1], rc.coef[kCullCoefs * i + 0]);
}

// After the minimum of the squared centre distances over the instances is known.
// `inner` (field_bounds; 0: unknown): a radius every box contains around its centre.  A point at distance c from the centre of box j has
// d_j <= c - inner (outside: the inscribed ball is no farther than that; inside: every face is at least inner - c away), so the smallest box
// distance is at most (nearest centre distance) - inner: a tighter start for the exact test's running minimum and a tighter bound test.
// Round 6, opt-in (-DVSRD_CULL_INNER; docs/OPTLOG.md round 6 item 7b) for the kernels of quad_step.h: a round of the benchmark scene carries 0.7 candidates that fail the exact test and 0.3-0.4 survivors no
// point of it needs (tests/survivor_statistics.py), because the minimum starts from the nearest CENTRE; 0.8 m less of a 1.2-1.6 m slack:
// config 2 232.4 -> 239.2 Mrays/s, config 5 31.4 -> 32.5, two-launch config 2 169.6 -> 171.7 (same box, tools/gpu_r06t.sh).
__device__ __forceinline__ RoundCull cull_round(const RayCull& rc, float t, float nearest2, float margin, float inner = 0.0f) {
    RoundCull c;
    const float s = rc.reach + rc.rnorm * fabsf(t);
    c.err = kCullQuadSlack * s * s;
    const float nearest = fast_sqrt(fmaxf(nearest2, 0.0f) + c.err);
    c.nearest_hi = nearest * (1.0f + kCullSlack) - inner;
    c.limit = (c.nearest_hi + margin) * (1.0f / (1.0f - kCullSlack));
    c.nearest_lo = fast_sqrt(fmaxf(nearest2 - c.err, 0.0f)) * (1.0f - kCullSlack);
    return c;
}

// Lanes on which instance i may matter (NaN-safe: an undecidable comparison keeps the instance).
__device__ __forceinline__ unsigned long long cull_near(const RayCull& rc, const RoundCull& c, int i, float d2) {
    const float reach = c.limit + rc.coef[kCullCoefs * i + 2];
    return __ballot(!(d2 > fmaf(reach, reach, c.err)));
}

// The culling pre-pass of one round: bit i of the result = instance i has to be evaluated (wave-uniform).
// Two branch-free loops, so that the LDS reads of consecutive instances are in flight together: squared centre distances (kept in
// d2cache[i][lane] when kCache, else recomputed) and their minimum; then one ballot per instance.
template <bool kCache>
__device__ __forceinline__ unsigned long long cull_round_mask(const RayCull& rc, int num_instances, float t, float margin, float* d2cache, int lane,
                                                              RoundCull* round_out) {
    const float ct = rc.c2 * t;
    float nearest2 = 3.0e38f;
    int i = 0;
    for (; i + 4 <= num_instances; i += 4) {          // coefficient reads first, then the stores (they may alias for the compiler)
        float d2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) d2[j] = centre_distance2(rc, i + j, t, ct);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (kCache) d2cache[(i + j) * kWave + lane] = d2[j];
            nearest2 = fminf(nearest2, d2[j]);
        }
    }
    for (; i < num_instances; ++i) {
        const float d2 = centre_distance2(rc, i, t, ct);
        if (kCache) d2cache[i * kWave + lane] = d2;
        nearest2 = fminf(nearest2, d2);
    }
    const RoundCull cull = cull_round(rc, t, nearest2, margin);
    unsigned long long mask = 0ull;
    for (i = 0; i + 4 <= num_instances; i += 4) {
        float d2[4], radius[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d2[j] = kCache ? d2cache[(i + j) * kWave + lane] : centre_distance2(rc, i + j, t, ct);
            radius[j] = rc.coef[kCullCoefs * (i + j) + 2];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float reach = cull.limit + radius[j];
            mask |= (__ballot(!(d2[j] > fmaf(reach, reach, cull.err))) != 0ull) ? (1ull << (i + j)) : 0ull;
        }
    }
    for (; i < num_instances; ++i) {
        const float d2 = kCache ? d2cache[i * kWave + lane] : centre_distance2(rc, i, t, ct);
        mask |= (cull_near(rc, cull, i, d2) != 0ull) ? (1ull << i) : 0ull;
    }
    if (round_out) *round_out = cull;
    return mask;
}

// Per-launch bounds of the field (wave-uniform; every wave computes them once):
//   margin  tau * T + slack of the culling test, or +huge (culling off) when some rotation matrix is not orthonormal to 1e-4;
//   reach   max_i |dim_i| + slack: every d_i(x) >= (nearest centre distance) (1 - k) - reach, which gives the soft-min a shift it
//           knows BEFORE the instance loop (union_accumulate); < 0 when that bound is unavailable (non-orthonormal rotations) or
//           reach / T is so large that exp(-(d - floor)/T) could underflow for the best instance.
struct FieldBounds { float margin, reach; bool yaw; float inner; };   // yaw: every rotation is exactly of the form box_value<true> assumes

__device__ __forceinline__ FieldBounds field_bounds(const float* __restrict__ instances, int num_instances, float inv_t, bool residual, unsigned flags) {
    float worst = 0.0f, rmax = 0.0f, rmin = 3.0e38f;
    bool yaw = true;
    for (int i = 0; i < num_instances; ++i) {
        const Instance in = load_instance(instances, i);
        yaw = yaw && in.r01 == 0.0f && in.r10 == 0.0f && in.r12 == 0.0f && in.r21 == 0.0f && in.r11 == 1.0f;
        const float g00 = in.r00 * in.r00 + in.r10 * in.r10 + in.r20 * in.r20, g11 = in.r01 * in.r01 + in.r11 * in.r11 + in.r21 * in.r21;
        const float g22 = in.r02 * in.r02 + in.r12 * in.r12 + in.r22 * in.r22, g01 = in.r00 * in.r01 + in.r10 * in.r11 + in.r20 * in.r21;
        const float g02 = in.r00 * in.r02 + in.r10 * in.r12 + in.r20 * in.r22, g12 = in.r01 * in.r02 + in.r11 * in.r12 + in.r21 * in.r22;
        worst = fmaxf(worst, fmaxf(fmaxf(fabsf(g00 - 1.0f), fabsf(g11 - 1.0f)), fmaxf(fabsf(g22 - 1.0f), fmaxf(fabsf(g01), fmaxf(fabsf(g02), fabsf(g12))))));
        rmax = fmaxf(rmax, fast_sqrt(in.dx * in.dx + in.dy * in.dy + in.dz * in.dz));
        rmin = fminf(rmin, fminf(in.dx, fminf(in.dy, in.dz)));
    }
    const bool orthonormal = worst < 1.0e-4f;                      // (a NaN parameter fails the test: no culling, running minimum)
    FieldBounds b;
    // residual fields: d_i = box + residual, residual in (0, 1): the upper bound of the best distance grows by 1
    b.margin = (orthonormal && !(flags & 4u)) ? (kCullTau / inv_t + 2.0e-3f + (residual ? 1.0f : 0.0f)) : 3.0e38f;
    const float reach = rmax + 2.0e-3f;
    b.reach = (orthonormal && !(flags & 16u) && (reach + 1.0f) * inv_t <= 50.0f) ? reach : -1.0f;
    b.yaw = yaw && !(flags & 32u);
    // the ball every box contains around its centre (cull_round); box-only fields with orthonormal rotations and positive extents only
    b.inner = (orthonormal && !residual && num_instances > 0 && rmin > 4.0e-3f) ? rmin * (1.0f - 1.0e-4f) - 2.0e-3f : 0.0f;
    return b;
}

// Online soft-min over the instances with a shift m (a lower bound of the d_i close to the smallest, or the running minimum) so
// that neither the exponentials nor the (d_i - u)/T factor of the union gradient lose digits at small temperature.
//   e_i = exp(-(d_i - m)/T), Z = sum e_i, S1 = sum e_i (d_i - m),
//   G0 = sum e_i gw_i, G1 = sum e_i (d_i - m) gw_i, L = sum e_i lambda_i (backward only)
struct UnionSums {
    float m, Z, S1;
    float g0x, g0y, g0z, g1x, g1y, g1z;
    float L;
};

// `floor` >= 0 ... any finite value: a lower bound of every d_i known before the loop (fixed shift, see union_accumulate);
// running = true: the shift is the running minimum.
__device__ __forceinline__ UnionSums union_init(bool running = true, float floor = 0.0f) {
    UnionSums s;
    s.m = running ? 3.0e38f : floor;  // 3e38 is finite: the first instance rescales the (all-zero) sums by exp(-3e38/T) = 0, no inf*0
    s.Z = 0.0f; s.S1 = 0.0f;
    s.g0x = s.g0y = s.g0z = 0.0f; s.g1x = s.g1y = s.g1z = 0.0f; s.L = 0.0f;
    return s;
}

// running (wave-uniform) = false: s.m is a lower bound of all d_i that lies within ~50 T of the smallest (field_bounds / RoundCull):
// no exponential overflows, the best one is >= e^-50, and every sum is a plain multiply-add: 12 instructions.  The caller checks Z
// afterwards and repeats the round with running = true if it underflowed (never seen; far-extrapolated samples are the candidates).
// running = true: one exponential per instance as well, but either the old sums are rescaled (new minimum) or the new term is:
// ~36 instructions, valid for any inputs.
// (kRunning is a template parameter, and the callers instantiate their whole instance loop once per value: a run-time branch
//  inside the loop costs eight register copies per instance where the two paths meet.)
// Returns the instance's soft-min term exp(-(d - s.m)/T) when the shift is fixed (kRunning = false; with the running minimum the
// term is relative to a shift that may still move, and the return value is d itself).
// kLambda = false: the caller has no label adjoints to mix (s.L stays untouched: one multiply-add fewer per instance).
template <bool kRunning = true, bool kLambda = true>
__device__ __forceinline__ float union_accumulate(UnionSums& s, float d, float gwx, float gwy, float gwz, float lambda, float inv_t) {
    if (!kRunning) {
        const float dd = d - s.m;
        const float e = fast_exp(-dd * inv_t);
        const float edd = e * dd;
        s.Z += e; s.S1 += edd;
        s.g0x += e * gwx; s.g0y += e * gwy; s.g0z += e * gwz;
        s.g1x += edd * gwx; s.g1y += edd * gwy; s.g1z += edd * gwz;
        if (kLambda) s.L += e * lambda;
        return e;
    }
    const bool lower = d < s.m;
    const float gap = lower ? (s.m - d) : (d - s.m);           // >= 0
    const float ex = fast_exp(-gap * inv_t);
    const float scale = lower ? ex : 1.0f;                      // multiplies the old sums
    const float e = lower ? 1.0f : ex;                          // weight of the new term
    const float shift = lower ? gap : 0.0f;                     // old (d_j - m) grow by the drop of m
    const float dd = lower ? 0.0f : gap;                        // new term's (d_i - m)
    s.g1x = scale * (s.g1x + shift * s.g0x) + e * dd * gwx;
    s.g1y = scale * (s.g1y + shift * s.g0y) + e * dd * gwy;
    s.g1z = scale * (s.g1z + shift * s.g0z) + e * dd * gwz;
    s.S1 = scale * (s.S1 + shift * s.Z) + e * dd;
    s.g0x = scale * s.g0x + e * gwx;
    s.g0y = scale * s.g0y + e * gwy;
    s.g0z = scale * s.g0z + e * gwz;
    if (kLambda) s.L = scale * s.L + e * lambda;
    s.Z = scale * s.Z + e;
    s.m = lower ? d : s.m;
    return d;
}

// Smallest normaliser the fixed-shift sums are trusted with: terms down to e^-18 of it are still normal numbers.
constexpr float kUnionTinyZ = 1.0e-28f;
// Largest (upper bound of the smallest distance - floor) / T a round may have to use the floor: the best term is >= e^-60 then.
constexpr float kUnionFloorSpan = 60.0f;

struct UnionValue {
    float u;            // union distance  sum_i w_i d_i
    float gx, gy, gz;   // its gradient    sum_i w_i (1 - (d_i - u)/T) grad d_i
    float m, inv_z;     // soft-min shift and 1/Z:  w_i = exp(-(d_i - m)/T) * inv_z
    float us;           // u - m  (>= 0)
    float b0x, b0y, b0z;  // sum_i w_i grad d_i
};

__device__ __forceinline__ UnionValue union_finish(const UnionSums& s, float inv_t) {
    UnionValue v;
    v.m = s.m;
    v.inv_z = fast_rcp(s.Z);
    v.us = s.S1 * v.inv_z;
    v.u = s.m + v.us;
    v.b0x = s.g0x * v.inv_z; v.b0y = s.g0y * v.inv_z; v.b0z = s.g0z * v.inv_z;
    const float k = 1.0f + v.us * inv_t;
    v.gx = k * v.b0x - inv_t * (s.g1x * v.inv_z);
    v.gy = k * v.b0y - inv_t * (s.g1y * v.inv_z);
    v.gz = k * v.b0z - inv_t * (s.g1z * v.inv_z);
    return v;
}

}  // namespace vsrd
