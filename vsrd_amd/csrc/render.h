// Wave-per-ray rendering core: one wavefront = one ray, one lane = one sample point.
//
// Reference semantics (CPU restatement: oracle/rendering.py):
//   stratified + inverse-transform sampling   vsrd/rendering/samplers.py:5-36
//   hierarchical_volumetric_rendering         vsrd/rendering/renderers.py:177-270
//   two-pass wrapper (pass 1 without grad)    scripts/main.py:511-523
//
// Per-ray state lives in registers (one or a few samples per lane) and in a small wave-private
// LDS partition (sorted distances, cdf, per-instance distance cache); nothing per-sample ever
// goes to HBM unless the caller asks for the API-faithful outputs.
#pragma once
#include "field.h"

namespace vsrd {

struct Ray { float ox, oy, oz, rx, ry, rz; };

struct Shading {
    float inv_t;    // 1 / soft-union temperature
    float cull;     // culling margin (field.h), wave-uniform; +huge disables culling
    float reach;    // soft-min floor: every d_i >= nearest centre distance - reach (field.h: field_bounds); < 0: running minimum
    float inner;    // radius of the ball every box contains around its centre (field.h: cull_round), 0: not used (only the kernels of quad_step.h set it)
    bool yaw;       // every rotation is exactly a rotation about y (field.h: box_value<true>)
    bool yaw_gradients;  // ... and the caller differentiates only through rotation_matrix_y (VSRD_FLAG_YAW_GRADIENTS): the adjoints of the
                         //     five constant entries of the rotation are not accumulated
    float std;      // sdf_std_deviation
    float inv_std;  // 1 / sdf_std_deviation
    float ratio;    // cosine_ratio
    float eps;      // epsilon (opacity denominator)
    unsigned mlp_bits;   // kMlpCentredBit when VSRD_FLAG_MLP_WEIGHTS_CENTRED is set (OR-ed into the residual's tile mask), kMlpSplitBit when the
                         // launch's `mlp` pointer is the table of split-bf16 operand images (VSRD_FLAG_MLP_SPLIT_BF16), else 0
    int mlp_stride;      // floats between two instances' rows of `mlp`: kMlpWeights, or kMlpImageWords for the image table
    float* mlp_lds;      // residual fields: kMlpWbarFloats (kMlpStageFloats in the kernels that take images) floats of the wave's LDS for residual_forward's staged weights
};

// torch.lerp(a, b, w): two-sided formula of ATen's lerp kernel.
__device__ __forceinline__ float torch_lerp(float a, float b, float w) {
    const float diff = b - a;
    return (fabsf(w) < 0.5f) ? (a + w * diff) : (b - diff * (1.0f - w));
}

// torch.linspace(start, end, steps)[i] (ATen: symmetric evaluation from both ends).
__device__ __forceinline__ float torch_linspace(float start, float end, int steps, int i) {
    const float step = (end - start) / static_cast<float>(steps - 1);
    return (i < steps / 2) ? (start + step * static_cast<float>(i)) : (end - step * static_cast<float>(steps - i - 1));
}

__device__ __forceinline__ float sigmoidf(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

// NeuS section-point opacity, renderers.py:228-248, with every intermediate the adjoint needs.
struct Opacity {
    float inv_gn;           // 1 / max(|g|, 1e-12)
    float nx, ny, nz;       // normal
    float cosine;           // r . n
    float cprime;           // annealed, negated cosine
    float phi_p, phi_n;     // logistic cdf at the previous / next section point
    float xx;               // (phi_p - phi_n) / (phi_p + eps)
    float alpha;            // relu(xx)
};

__device__ __forceinline__ Opacity opacity_of(const UnionValue& v, const Ray& ray, float delta, const Shading& sh) {
    Opacity o;
    const float gn = fast_sqrt(v.gx * v.gx + v.gy * v.gy + v.gz * v.gz);
    o.inv_gn = fast_rcp(fmaxf(gn, 1.0e-12f));                       // F.normalize eps
    o.nx = v.gx * o.inv_gn; o.ny = v.gy * o.inv_gn; o.nz = v.gz * o.inv_gn;
    o.cosine = ray.rx * o.nx + ray.ry * o.ny + ray.rz * o.nz;
    const float a = fmaxf(-o.cosine * 0.5f + 0.5f, 0.0f);
    const float b = fmaxf(-o.cosine, 0.0f);
    o.cprime = -torch_lerp(a, b, sh.ratio);
    const float half = o.cprime * delta / 2.0f;
    o.phi_p = sigmoidf((v.u - half) * sh.inv_std);
    o.phi_n = sigmoidf((v.u + half) * sh.inv_std);
    o.xx = (o.phi_p - o.phi_n) * fast_rcp(o.phi_p + sh.eps);
    o.alpha = fmaxf(o.xx, 0.0f);
    return o;
}

__device__ __forceinline__ bool wave_any(bool pred) { return __ballot(pred) != 0ull; }

// Evaluate the union at the sample of ray parameter t (position x, y, z): culling pre-pass (field.h) over the squared centre
// distances, then a uniform loop over the instances that survive it, with scalar parameter loads.
// `lam` (LDS, or nullptr) are the per-instance label adjoints accumulated into sums.L by the backward.
// On return bit i of *near_out says whether instance i was evaluated this round (wave-uniform) and, with kCacheDistances,
// dcache[i][lane] holds d_i for those instances.
// The instance loop of eval_union: the instances of `evaluated` (bound test, field.h) that also pass the exact test -- the others
// are cleared from the mask --, accumulated with a fixed soft-min shift `floor` or (kRunning) the running minimum.
// `live`: the lanes that hold a point of their own (the others repeat the last one and carry no weight): only their tiles run the MLP.
template <bool kCacheDistances, bool kResidual, bool kRunning, bool kYaw>
__device__ __forceinline__ UnionSums union_loop(const float* __restrict__ instances, const float* __restrict__ mlp, unsigned long long& evaluated,
                                                const Shading& sh, const RoundCull& cull, float floor, float x, float y, float z,
                                                float* dcache, int lane, const float* lam, unsigned long long live = ~0ull) {
    UnionSums sums = union_init(kRunning, floor);
    float best = cull.nearest_hi;                                               // upper bound of the smallest box distance, per lane
    for (unsigned long long todo = evaluated; todo != 0ull; todo &= todo - 1ull) {    // wave-uniform: the instances that survive
        const int i = __builtin_ctzll(todo);
        const Instance in = load_instance_as<!kResidual>(instances, i);
        BoxEval e = box_value<kYaw>(in, x, y, z);
        const unsigned long long near = __ballot(!(e.d - best > sh.cull));      // (NaN-safe: an undecidable comparison keeps the instance)
        if (near == 0ull) { evaluated &= ~(1ull << i); continue; }
        best = fminf(best, e.d);
        box_gradient<kYaw>(e, in);
        if (kResidual) add_residual<kYaw>(e, in, residual_forward_packed(mlp + i * sh.mlp_stride, e.px, e.py, e.pz, near & live, sh.mlp_bits, sh.mlp_lds));
        if (kCacheDistances) dcache[i * kWave + lane] = e.d;
        union_accumulate<kRunning>(sums, e.d, e.gwx, e.gwy, e.gwz, lam ? lam[i] : 0.0f, sh.inv_t);
    }
    return sums;
}

template <bool kCacheDistances, bool kResidual>
__device__ __forceinline__ UnionValue eval_union(const float* __restrict__ instances, const float* __restrict__ mlp, int num_instances,
                                                 const Shading& sh, const RayCull& rc, float t, float x, float y, float z, float* dcache, int lane,
                                                 const float* lam, unsigned long long* near_out, float* lam_z_out, unsigned long long live = ~0ull) {
    RoundCull cull;
    unsigned long long evaluated = cull_round_mask<true>(rc, num_instances, t, sh.cull, dcache, lane, &cull);
    UnionSums sums;
    // wave-uniform: no floor for this field, or a floor too far below the minimum on some lane (samples extrapolated to 1e6 m,
    // where the bounds are hundreds of metres wide): such rounds go straight to the running minimum
    const float floor = cull.nearest_lo - sh.reach;
    bool running = sh.reach < 0.0f || wave_any(!((cull.nearest_hi + 1.0f - floor) * sh.inv_t <= kUnionFloorSpan));
    if (!running) {
        sums = sh.yaw ? union_loop<kCacheDistances, kResidual, false, true>(instances, mlp, evaluated, sh, cull, floor, x, y, z, dcache, lane, lam, live)
                      : union_loop<kCacheDistances, kResidual, false, false>(instances, mlp, evaluated, sh, cull, floor, x, y, z, dcache, lane, lam, live);
        running = wave_any(!(sums.Z >= kUnionTinyZ));                           // the fixed shift underflowed somewhere: repeat the round
    }
    if (running) sums = union_loop<kCacheDistances, kResidual, true, false>(instances, mlp, evaluated, sh, cull, 0.0f, x, y, z, dcache, lane, lam, live);
    const UnionValue v = union_finish(sums, sh.inv_t);
    if (near_out) *near_out = evaluated;
    if (lam_z_out) *lam_z_out = sums.L * v.inv_z;
    return v;
}

// Render the D-1 interval mid-points of the sorted distances `dist` (LDS, wave-private).
//   weights[k]  <- compositing weight of sample (k*64 + lane)         (0 for padding lanes)
//   return      <- lane i holds label i (sum_s w_s * softmin weight)  when kLabels
//   grad_out / weight_out: this ray's [D-1,3] / [D-1] rows in HBM, or nullptr.
// first_point / carry_out: a ray split over the waves of a workgroup (render_kernels.h: residual_step_pair_kernel) -- this wave renders
// the kRounds rounds from point `first_point` on with an entering transmittance of 1 and reports what its rounds let through.
template <int kRounds, bool kLabels, bool kResidual>
__device__ __forceinline__ float render_pass(const float* __restrict__ instances, const float* __restrict__ mlp, int num_instances, const Shading& sh,
                                             const Ray& ray, const RayCull& rc, const float* dist, int num_distances, float* dcache,
                                             float (&weights)[kRounds], float* grad_out, float* weight_out,
                                             int first_point = 0, float* carry_out = nullptr) {
    const int lane = lane_id();
    const int num_points = num_distances - 1;
    float carry = 1.0f;       // transmittance entering this round (wave-uniform)
    float label_acc = 0.0f;
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
        weights[k] = 0.0f;
        if (first_point + k * kWave >= num_points) continue;         // wave-uniform
        const int s = first_point + k * kWave + lane;
        const bool valid = s < num_points;
        const int s0 = valid ? s : (num_points - 1);
        const float d0 = dist[s0], d1 = dist[s0 + 1];
        const float delta = d1 - d0;
        const float mid = (d0 + d1) / 2.0f;
        const float x = ray.ox + ray.rx * mid, y = ray.oy + ray.ry * mid, z = ray.oz + ray.rz * mid;
        unsigned long long evaluated;
        // (padding lanes repeat the last point: the residual MLP skips 16-lane tiles that hold nothing else -- at the reference's S = 100
        //  the last round of either pass is mostly padding)
        const UnionValue v = eval_union<kLabels, kResidual>(instances, mlp, num_instances, sh, rc, mid, x, y, z, dcache, lane, nullptr, &evaluated, nullptr,
                                                            __ballot(valid));
        const Opacity op = opacity_of(v, ray, delta, sh);
        const float alpha = valid ? op.alpha : 0.0f;
        const float inclusive = wave_inclusive_product(1.0f - alpha);
        const float trans = carry * wave_shift_up(inclusive, 1.0f, lane);
        const float w = trans * alpha;
        carry *= read_lane(inclusive, kWave - 1);
        weights[k] = w;
        if (grad_out != nullptr && valid) {
            grad_out[s * 3 + 0] = v.gx; grad_out[s * 3 + 1] = v.gy; grad_out[s * 3 + 2] = v.gz;
        }
        if (weight_out != nullptr && valid) weight_out[s] = w;
        if (kLabels) {
            const float scale = w * v.inv_z;
            for (unsigned long long todo = evaluated; todo != 0ull; todo &= todo - 1ull) {     // (culled instances: weight < exp(-18))
                const int i = __builtin_ctzll(todo);
                const float e = fast_exp(-(dcache[i * kWave + lane] - v.m) * sh.inv_t) * scale;
                const float total = wave_sum(e);
                label_acc = (lane == i) ? (label_acc + total) : label_acc;
            }
        }
    }
    if (carry_out != nullptr) *carry_out = carry;      // (rays split over waves: the transmittance this wave's rounds let through)
    return label_acc;
}

// Number of elements of the sorted LDS array a[0..n), n >= 1, that are  < v  (kStrict) or <= v, by binary lifting: the count is
// built from its highest bit down (`top` = the largest power of two <= n, wave-uniform), four vector instructions and one LDS read per
// step (the lo / hi interval form took ten).  Probes beyond n read the largest element a[n - 1]: if that one is below, all n are, and
// the final clamp returns n; otherwise they answer "not below", as elements beyond the end would.
template <bool kStrict>
__device__ __forceinline__ int count_below(const float* a, int n, float v, int top) {
    int pos = 0;
    for (int stride = top; stride >= 1; stride >>= 1) {
        const float probe = a[min(pos + stride - 1, n - 1)];
        pos += (kStrict ? (probe < v) : (probe <= v)) ? stride : 0;
    }
    return min(pos, n);
}

// `top` of count_below for arrays of n elements.
__device__ __forceinline__ int search_iterations(int n) { return 1 << (31 - __builtin_clz(static_cast<unsigned>(n))); }

// Per-wave LDS partition, in floats.
struct WaveLds {
    float* coarse;    // [S]   pass-1 distances
    float* cdf;       // [S]   cdf[0] = 0, cdf[k+1] = cumsum pdf
    float* uraw;      // [S]   raw fine uniforms
    float* usorted;   // [S]   sorted fine uniforms
    float* fine;      // [S]   importance samples
    float* merged;    // [2S]  sorted union
    float* dcache;    // [N,64] per-instance distances of the current round
    float* cull;      // [N,4]  per-ray culling coefficients (field.h: RayCull)
};

__host__ __device__ constexpr int wave_lds_floats(int num_samples, int num_instances) {
    return 7 * num_samples + num_instances * kWave + cull_coef_floats(num_instances);
}

__device__ __forceinline__ WaveLds carve_lds(float* base, int num_samples, int num_instances = 0) {
    WaveLds l;
    l.coarse = base; l.cdf = base + num_samples; l.uraw = base + 2 * num_samples; l.usorted = base + 3 * num_samples;
    l.fine = base + 4 * num_samples; l.merged = base + 5 * num_samples; l.dcache = base + 7 * num_samples;
    l.cull = l.dcache + num_instances * kWave;
    return l;
}

// Sort S uniforms (one per lane and round) by ranking: rank = #{smaller} + #{equal with lower index}.
template <int kRounds>
__device__ __forceinline__ void rank_sort(const float* raw, float* sorted, int n) {
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
        if (k * kWave >= n) continue;
        const int idx = k * kWave + lane;
        const bool valid = idx < n;
        const float v = raw[valid ? idx : 0];
        int rank = 0;
        for (int j = 0; j < n; ++j) {                                // LDS broadcast reads
            const float o = raw[j];
            rank += ((o < v) || (o == v && j < idx)) ? 1 : 0;
        }
        if (valid) sorted[rank] = v;
    }
}

// samplers.py:11-36 + renderers.py:198-210: importance samples from (coarse, weights), merged with
// the coarse distances into l.merged[0..2S).  weights[k] holds w of sample k*64+lane (0 when >= S-1).
// l.usorted must hold the sorted uniforms.
template <int kRoundsS>
__device__ __forceinline__ void importance_merge(const WaveLds& l, int S, const float (&weights)[kRoundsS]) {
    const int lane = lane_id();
    // pdf = w / max(sum |w|, 1e-12); cdf = cumsum
    float total = 0.0f;
#pragma unroll
    for (int k = 0; k < kRoundsS; ++k) total += wave_sum(fabsf(weights[k]));
    const float denom = fmaxf(total, 1.0e-12f);
    float running = 0.0f;
#pragma unroll
    for (int k = 0; k < kRoundsS; ++k) {
        if (k * kWave >= S - 1) continue;
        const int idx = k * kWave + lane;
        const float inclusive = wave_inclusive_sum(weights[k] / denom) + running;
        if (idx < S - 1) l.cdf[idx + 1] = inclusive;
        running = read_lane(inclusive, kWave - 1);
    }
    if (lane == 0) l.cdf[0] = 0.0f;
    wave_lds_sync();
    // inverse transform
    const int iters = search_iterations(S);
    float fine_max = -3.0e38f;
#pragma unroll
    for (int k = 0; k < kRoundsS; ++k) {
        if (k * kWave >= S) continue;
        const int j = k * kWave + lane;
        const bool valid = j < S;
        const float u = l.usorted[valid ? j : (S - 1)];
        int upper = count_below<true>(l.cdf, S, u, iters);          // searchsorted(right=False)
        upper = min(max(upper, 1), S - 1);
        const float c_lo = l.cdf[upper - 1], c_hi = l.cdf[upper];
        const float b_lo = l.coarse[upper - 1], b_hi = l.coarse[upper];
        const float t = (u - c_lo) / (c_hi - c_lo + 1.0e-6f);
        float sample = torch_lerp(b_lo, b_hi, t);
        // The samples are non-decreasing in exact arithmetic; a running maximum removes one-ulp
        // inversions of the two-sided lerp so that the rank merge below is a valid permutation.
        float scan = valid ? sample : -3.0e38f;
        scan = fmaxf(scan, dpp_move<kDppRowShr1>(scan, scan));
        scan = fmaxf(scan, dpp_move<kDppRowShr2>(scan, scan));
        scan = fmaxf(scan, dpp_move<kDppRowShr4>(scan, scan));
        scan = fmaxf(scan, dpp_move<kDppRowShr8>(scan, scan));
        scan = fmaxf(scan, dpp_move<kDppRowBcast15, 0xa>(scan, scan));
        scan = fmaxf(scan, dpp_move<kDppRowBcast31, 0xc>(scan, scan));
        sample = fmaxf(scan, fine_max);
        fine_max = fmaxf(fine_max, read_lane(scan, kWave - 1));
        if (valid) l.fine[j] = sample;
    }
    wave_lds_sync();
    // merge by rank (ties: coarse first)
#pragma unroll
    for (int k = 0; k < kRoundsS; ++k) {
        if (k * kWave >= S) continue;
        const int j = k * kWave + lane;
        const bool valid = j < S;
        const int jj = valid ? j : (S - 1);
        const float a = l.coarse[jj], b = l.fine[jj];
        const int rank_a = jj + count_below<true>(l.fine, S, a, iters);
        const int rank_b = jj + count_below<false>(l.coarse, S, b, iters);
        if (valid) { l.merged[rank_a] = a; l.merged[rank_b] = b; }
    }
    wave_lds_sync();
}

}  // namespace vsrd
