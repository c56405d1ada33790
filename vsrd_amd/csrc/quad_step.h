// Box-only fields with SEVERAL NEIGHBOURING RAYS PER WAVE: the fused optimisation step (vsrd_render_silhouette_step), the two-pass
// forward (vsrd_render_hierarchical_forward, labels / distances only) and the adjoint at saved distances (vsrd_render_backward, label
// adjoints only) for dense launches.
//
// Same arithmetic and semantics as the one-ray kernels of render_kernels.h (scripts/main.py:511-523, 653-671,
// vsrd/rendering/renderers.py:177-270, samplers.py:5-36), different mapping:
//   render_silhouette_kernel        one wave = one ray,   lane = sample,                              rounds of 64 consecutive samples
//   render_silhouette_quad_kernel   one wave = four rays, lane = (ray = lane / 16, sample = lane % 16),  rounds of 4 x 16 samples   (N <= 16, S <= 64)
//   render_silhouette_pair_kernel   one wave = two rays,  lane = (ray = lane / 32, sample = lane % 32),  rounds of 2 x 32 samples   (N <= 64, S <= 128)
// Why: the instance culling (field.h) is wave-uniform, so its granularity is what one round covers.  64 consecutive samples span
// half a ray and keep 11.1 (pass 1) / 7.4 (pass 2) of 16 instances at the mid schedule of BASELINE config 2; 16 consecutive samples of
// four neighbouring pixels cover a quarter of that depth range at nearly the same place and keep 5.8 / 4.8 (tests/cull_statistics.py):
// the instance loops -- 78 % of the step -- shrink by 35-48 %, and one parameter-adjoint butterfly serves four rays instead of one.
// What it costs: the per-ray state of the adjoint lives in registers for all rounds of pass 2 (8 rounds x 6-7 floats), so everything
// else about a sample is re-derived where it is needed (opacity in the reverse sweep, the sample position from the sorted distances)
// or waits in LDS (transmittance, C1..C3); compositing scans are scans over a ray's lanes (DPP) with per-ray carries; the per-ray
// constants live in LDS.  DESIGN.md section 2b.
#pragma once
#include "render_kernels.h"

namespace vsrd {

constexpr int kQuadRays = 4;             // rays per wave of the 16-lane shape (BASELINE config 2)
constexpr int kRowLanes = 16;
constexpr int kQuadMaxInstances = 16;    // 16-lane shape: lane (ray, n) owns label n of its ray
constexpr int kQuadMaxSamples = 64;
constexpr int kPairMaxInstances = 64;    // 32-lane shape (two rays per wave): lane (ray, c) owns labels c and 32 + c
constexpr int kPairMaxSamples = 128;

// ---- primitives over the kL lanes of one ray (kL = 16: a DPP row; kL = 32: two rows); every lane receives its ray's result -------------
constexpr int kDppRowShl1 = 0x101, kDppRowShl2 = 0x102, kDppRowShl4 = 0x104, kDppRowShl8 = 0x108;

__device__ __forceinline__ float max_xor16(float v) {          // max with the lane 16 away (wave.h: add_xor16)
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
// value of a fixed lane of the wave (byte address of that lane for ds_bpermute: RowLanes::first / ::last)
__device__ __forceinline__ float lane_gather(float v, int byte_address) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_address, __builtin_bit_cast(int, v)));
}

struct RowLanes {
    int lane, row, col;       // row = ray of the wave, col = lane within the ray
    int first, last;          // ds_bpermute addresses of the ray's first / last lane
};
template <int kL>
__device__ __forceinline__ RowLanes row_lanes(int lane) {
    RowLanes r;
    r.lane = lane; r.row = r.lane / kL; r.col = r.lane % kL;
    r.first = (r.lane & ~(kL - 1)) << 2; r.last = (r.lane | (kL - 1)) << 2;
    return r;
}

template <int kL>
__device__ __forceinline__ float seg_sum(float v) {
    v += dpp_move<kDppQuadXor1>(0.0f, v);
    v += dpp_move<kDppQuadXor2>(0.0f, v);
    v += dpp_move<kDppRowHalfMirror>(0.0f, v);
    v += dpp_move<kDppRowMirror>(0.0f, v);
    return kL == 32 ? add_xor16(v) : v;
}
template <int kL>
__device__ __forceinline__ float seg_max(float v) {
    v = fmaxf(v, dpp_move<kDppQuadXor1>(v, v));
    v = fmaxf(v, dpp_move<kDppQuadXor2>(v, v));
    v = fmaxf(v, dpp_move<kDppRowHalfMirror>(v, v));
    v = fmaxf(v, dpp_move<kDppRowMirror>(v, v));
    return kL == 32 ? max_xor16(v) : v;
}
template <int kL>
__device__ __forceinline__ float seg_inclusive_sum(float v) {          // prefix over the ray's lanes, its first lane first
    v += dpp_move<kDppRowShr1>(0.0f, v);
    v += dpp_move<kDppRowShr2>(0.0f, v);
    v += dpp_move<kDppRowShr4>(0.0f, v);
    v += dpp_move<kDppRowShr8>(0.0f, v);
    if (kL == 32) v += dpp_move<kDppRowBcast15, 0xa>(0.0f, v);         // rows 1, 3 += the last lane of rows 0, 2
    return v;
}
template <int kL>
__device__ __forceinline__ float seg_inclusive_product(float v) {
#ifndef VSRD_NO_DPP_INPLACE
    v = mul_shr8(mul_shr4(mul_shr2(mul_shr1(v))));
#else
    v *= dpp_move<kDppRowShr1>(1.0f, v);
    v *= dpp_move<kDppRowShr2>(1.0f, v);
    v *= dpp_move<kDppRowShr4>(1.0f, v);
    v *= dpp_move<kDppRowShr8>(1.0f, v);
#endif
    if (kL == 32) v *= dpp_move<kDppRowBcast15, 0xa>(1.0f, v);
    return v;
}
template <int kL>
__device__ __forceinline__ float seg_inclusive_max(float v) {
#ifndef VSRD_NO_DPP_INPLACE
    v = max_shr8(max_shr4(max_shr2(max_shr1(v))));
#else
    v = fmaxf(v, dpp_move<kDppRowShr1>(v, v));
    v = fmaxf(v, dpp_move<kDppRowShr2>(v, v));
    v = fmaxf(v, dpp_move<kDppRowShr4>(v, v));
    v = fmaxf(v, dpp_move<kDppRowShr8>(v, v));
#endif
    if (kL == 32) v = fmaxf(v, dpp_move<kDppRowBcast15, 0xa>(v, v));
    return v;
}
template <int kL>
__device__ __forceinline__ float seg_suffix_sum(float v, const RowLanes& rl) {     // inclusive sum over this and the LATER lanes of the ray
    v += dpp_move<kDppRowShl1>(0.0f, v);
    v += dpp_move<kDppRowShl2>(0.0f, v);
    v += dpp_move<kDppRowShl4>(0.0f, v);
    v += dpp_move<kDppRowShl8>(0.0f, v);
    if (kL == 32) {                                                                // the first row of the ray += the second row's total
        const float upper = lane_gather(v, ((rl.lane | 31) - 15) << 2);           // lane 16 of the ray holds its second row's total
        v += (rl.col < 16) ? upper : 0.0f;
    }
    return v;
}
// value of the previous lane of the ray, `first` for the ray's first lane
template <int kL>
__device__ __forceinline__ float seg_shift_up(float v, float first, const RowLanes& rl) {
    if (kL == 16) return dpp_move<kDppRowShr1>(first, v);
    const float previous = lane_gather(v, (rl.lane - 1) << 2);
    return rl.col == 0 ? first : previous;
}

// The lane id as a value the optimiser cannot see through: what is derived from it inside a loop iteration (LDS addresses, sample
// indices, stratification bounds ...) is re-derived there in a few integer instructions instead of being hoisted out of the loop
// and held -- or spilled -- over all of it (render_silhouette_quad_kernel: ~50 such registers).
__device__ __forceinline__ int opaque_lane_id() {
    int lane = lane_id();
    asm volatile("" : "+v"(lane));
    return lane;
}

// Experiment (-DVSRD_INSTANCE_LDS): the hot full-shape kernel stages the instance block in the workgroup's LDS and the instance loops read
// it with four ds_read_b128 of one address (a broadcast): the parameters are VGPR operands then -- an FMA with an SGPR operand issues in
// 4.4 cycles, on VGPRs in 2.8 (profiles/r01/op_rates.txt), and the instance loops have ~14 of them per (sample, instance).
__device__ __forceinline__ Instance load_instance_block(const float* __restrict__ instances, int i) {
#ifdef VSRD_INSTANCE_LDS
    const float4* p = reinterpret_cast<const float4*>(instances + i * kInstanceStride);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    Instance v;
    v.tx = a.x; v.ty = a.y; v.tz = a.z;
    v.r00 = a.w; v.r01 = b.x; v.r02 = b.y;
    v.r10 = b.z; v.r11 = b.w; v.r12 = c.x;
    v.r20 = c.y; v.r21 = c.z; v.r22 = c.w;
    v.dx = d.x; v.dy = d.y; v.dz = d.z;
    return v;
#else
    return load_instance(instances, i);
#endif
}

// ---- per-wave LDS ------------------------------------------------------------------------------------------------------------------
//   4 x [ coarse S | fine S (first the sorted uniforms) | 16 pad | merged 2S (first: raw uniforms | cdf) | 16 + 4 pad ]   the first pad takes
//       the tail of the per-point array below when 2S - 1 points are padded to whole rounds of 16, the second the tail of C2
//   dcache [N][64]     soft-min terms of the current round (label sums; the last round's serve the label mix)
//   4 x [ N x (a, b, radius, lambda) | 4 pad ]  culling coefficients of (ray, instance) (field.h: RayCull) + the ray's label adjoints
//   4 x [ ox oy oz rx | ry rz reach pad ]       the rays themselves: re-read where they are needed instead of living in 9 registers
// After importance_merge the coarse | fine part of a row holds, per pass-2 point, first the transmittance (forward sweep -> reverse
// sweep) and then the interval mid-point (reverse sweep -> per-instance phase); the reverse sweep also turns merged[s + 1] into C2 of
// point s once the distances of its round have been read (the rounds run backwards, so merged[16 q] stays for round q - 1).
// (the sorted distances are followed by `lanes` + 4 floats: the reverse sweep writes C2 of point s at merged[s + 1] for EVERY lane of a
//  round, padding lanes included, i.e. up to index roundup(2S - 1, lanes) -- without the room a row's padding lanes overwrote the next
//  row's transmittances of round 0 whenever 2S - 1 is not just below a multiple of `lanes`: found by the shape sweep of round 3)
constexpr int kRowRayFloats = 8;           // a ray in LDS: ox oy oz rx | ry rz reach pad
__host__ __device__ constexpr int quad_row_floats(int num_samples, int lanes = kRowLanes) { return 4 * num_samples + 2 * lanes + 4; }
__host__ __device__ constexpr int quad_merged_offset(int num_samples, int lanes = kRowLanes) { return 2 * num_samples + lanes; }
__host__ __device__ constexpr int quad_coef_floats(int num_instances) { return kCullCoefs * num_instances + 4; }
__host__ __device__ constexpr int quad_rounds_s(int num_samples, int lanes = kRowLanes) {          // rounds of `lanes` coarse points: 1, 2 or 4
    return num_samples <= lanes ? 1 : (num_samples <= 2 * lanes ? 2 : 4);
}
#ifndef VSRD_CACHE_SLOTS
#define VSRD_CACHE_SLOTS 22       // (what three workgroups per CU leave room for at S = 128, N = 64: 52 160 of 54 613 bytes)
#endif
static_assert(VSRD_CACHE_SLOTS >= 16 && VSRD_CACHE_SLOTS <= 32, "quad_label_sums gathers a survivor's rank within the 32 lanes of a ray, and the rows are reused as the [16][64] C1 / C3 area");
constexpr int kCacheSlots = VSRD_CACHE_SLOTS;   // (>= 16: the rows also hold C1 / C3 of the eight pass-2 rounds) shapes with more instances than lanes per ray: soft-min terms of the first 16 survivors of a round
// Two rays per wave: the terms of survivor k of a round start at float k * kSlotStride (64 lanes + 4: rows stay 16-byte aligned and
// sixteen lanes reading sixteen DIFFERENT rows with ds_read_b128 touch every bank once -- quad_label_sums), and one more row holds the
// lanes' label scales.  The same floats hold C1 / C3 of the eight pass-2 rounds afterwards ([16][64]).
#ifndef VSRD_LABEL_LOOP
constexpr int kSlotStride = kWave + 4;
#else
constexpr int kSlotStride = kWave;
#endif
__host__ __device__ constexpr int quad_cache_floats(int num_samples, int num_instances, int lanes = kRowLanes) {   // soft-min terms of a round, later C1 / C3 of every pass-2 round ([rows][64])
    return lanes == kRowLanes ? (num_instances > 4 * quad_rounds_s(num_samples, lanes) ? num_instances : 4 * quad_rounds_s(num_samples, lanes)) * kWave
                              : (kCacheSlots + 1) * kSlotStride;
}
__host__ __device__ constexpr int quad_lds_floats(int num_samples, int num_instances, int lanes = kRowLanes) {
    return (kWave / lanes) * (quad_row_floats(num_samples, lanes) + quad_coef_floats(num_instances) + kRowRayFloats) + quad_cache_floats(num_samples, num_instances, lanes);
}

// Stratified distances and sorted fine uniforms of the lane's ray (render_kernels.h: stage_ray_samples, per 16-lane row).
template <int kL, int kRoundsS>
__device__ __forceinline__ void quad_stage_samples(float* rowbase, const RenderArgs& c, int S, int ray, const float* __restrict__ u_coarse,
                                                   const float* __restrict__ u_fine, bool sorted_input, const RowLanes& rl,
                                                   float* __restrict__ u_coarse_out = nullptr, float* __restrict__ u_fine_out = nullptr, bool write = false) {
    float* coarse = rowbase;
    float* usorted = rowbase + S;
    float* uraw = rowbase + quad_merged_offset(S, kL);
    const size_t urow = static_cast<size_t>(ray) * S;
    const bool philox = (u_coarse == nullptr || u_fine == nullptr);
    float spacing[kRoundsS];
    float running = 0.0f, extra_spacing = 0.0f;
#pragma unroll
    for (int k = 0; k < kRoundsS; ++k) {
        const int idx = k * kL + rl.col;
        spacing[k] = 0.0f;
        if (k * kL >= S) continue;
        float uc = 0.0f, uf = 0.0f;
        if (philox) {
            const Philox4 rnd = philox4x32_10(static_cast<uint32_t>(ray), static_cast<uint32_t>(idx),
                                              static_cast<uint32_t>(c.stream_offset), static_cast<uint32_t>(c.stream_offset >> 32),
                                              static_cast<uint32_t>(c.seed), static_cast<uint32_t>(c.seed >> 32));
            uc = uniform_from_bits(rnd.x);
            uf = uniform_from_bits(rnd.y);
            if (k == 0) extra_spacing = -fast_log(1.0f - lane_gather(uniform_from_bits(rnd.z), rl.first));   // the (S+1)-th spacing
        }
        const bool valid = idx < S;
        if (u_coarse != nullptr && valid) uc = u_coarse[urow + idx];
        if (u_fine != nullptr && valid) uf = u_fine[urow + idx];
        if (valid) {
            const float lo = torch_linspace(c.near, c.far, S + 1, idx);
            const float hi = torch_linspace(c.near, c.far, S + 1, idx + 1);
            coarse[idx] = torch_lerp(lo, hi, uc);
            if (u_coarse_out != nullptr && write) u_coarse_out[urow + idx] = uc;
        }
        if (u_fine == nullptr) {
            const float e = valid ? -fast_log(1.0f - uf) : 0.0f;
            const float inclusive = seg_inclusive_sum<kL>(e) + running;
            spacing[k] = inclusive;
            running = lane_gather(inclusive, rl.last);
        } else if (valid) {
            (sorted_input ? usorted : uraw)[idx] = uf;
            if (u_fine_out != nullptr && write) u_fine_out[urow + idx] = uf;
        }
    }
    if (u_fine == nullptr) {
        const float inv_total = fast_rcp(running + extra_spacing);
#pragma unroll
        for (int k = 0; k < kRoundsS; ++k) {
            const int idx = k * kL + rl.col;
            if (idx < S) {
                const float sorted_u = fminf(spacing[k] * inv_total, 0.99999994f);
                usorted[idx] = sorted_u;
                if (u_fine_out != nullptr && write) u_fine_out[urow + idx] = sorted_u;      // (the uniforms actually used: already sorted)
            }
        }
    }
    wave_lds_sync();
    if (u_fine != nullptr && !sorted_input) {             // rank sort of the row's raw draws (parity tests, the API-faithful path)
#pragma unroll
        for (int k = 0; k < kRoundsS; ++k) {
            if (k * kL >= S) continue;
            const int idx = k * kL + rl.col;
            const bool valid = idx < S;
            const float v = uraw[valid ? idx : 0];
            int rank = 0;
            for (int j = 0; j < S; ++j) {
                const float o = uraw[j];
                rank += ((o < v) || (o == v && j < idx)) ? 1 : 0;
            }
            if (valid) usorted[rank] = v;
        }
        wave_lds_sync();
    }
}

// Number of elements of the sorted LDS array a[0..n), n <= 64, that are < v (kStrict) or <= v: the same count as render.h's
// count_below, by binary lifting (strides 32 ... 1, then the one element a stride-1 step can leave open): four vector instructions
// and one LDS read per step where the lo / hi interval form takes ten.  Probes beyond n read the largest element a[n - 1]: if that
// one is below, all n are, and the final clamp returns n.
template <bool kStrict>
__device__ __forceinline__ int count_below_64(const float* a, int n, float v) {
    int pos = 0;
#pragma unroll
    for (int stride = 32; stride >= 1; stride >>= 1) {
        const float probe = a[min(pos + stride - 1, n - 1)];
        pos += (kStrict ? (probe < v) : (probe <= v)) ? stride : 0;
    }
    const float probe = a[min(pos, n - 1)];
    pos += (kStrict ? (probe < v) : (probe <= v)) ? 1 : 0;
    return min(pos, n);
}

template <int kL, bool kStrict>
__device__ __forceinline__ int seg_count_below(const float* a, int n, float v) {
    if (kL == kRowLanes) return count_below_64<kStrict>(a, n, v);                 // S <= 64
    return count_below<kStrict>(a, n, v, search_iterations(n));                    // render.h: any n
}

// samplers.py:11-36 + renderers.py:198-210 for the lane's ray (render.h: importance_merge, per row): w[k] = coarse weight of point
// k * 16 + col (0 beyond S - 2); on return merged[0..2S) is the sorted union of the coarse and the fine distances.
template <int kL, int kRoundsS>
__device__ __forceinline__ void quad_importance_merge(float* rowbase, int S, const float (&w)[kRoundsS], const RowLanes& rl) {
    float* coarse = rowbase;
    float* fine = rowbase + S;                    // holds the sorted uniforms on entry: lane j turns u[j] into fine[j] in place
    float* merged = rowbase + quad_merged_offset(S, kL);
    float* cdf = merged + S;                      // dead before the merge writes there
    float total = 0.0f;
#pragma unroll
    for (int k = 0; k < kRoundsS; ++k) total += seg_sum<kL>(fabsf(w[k]));
    const float denom = fmaxf(total, 1.0e-12f);
    float running = 0.0f;
#pragma unroll
    for (int k = 0; k < kRoundsS; ++k) {
        if (k * kL >= S - 1) continue;
        const int idx = k * kL + rl.col;
        const float inclusive = seg_inclusive_sum<kL>(w[k] / denom) + running;
        if (idx < S - 1) cdf[idx + 1] = inclusive;
        running = lane_gather(inclusive, rl.last);
    }
    if (rl.col == 0) cdf[0] = 0.0f;
    wave_lds_sync();
    float fine_max = -3.0e38f;
#pragma unroll
    for (int k = 0; k < kRoundsS; ++k) {
        if (k * kL >= S) continue;
        const int j = k * kL + rl.col;
        const bool valid = j < S;
        const float u = fine[valid ? j : (S - 1)];
        int upper = seg_count_below<kL, true>(cdf, S, u);
        upper = min(max(upper, 1), S - 1);
        const float c_lo = cdf[upper - 1], c_hi = cdf[upper];
        const float b_lo = coarse[upper - 1], b_hi = coarse[upper];
        const float t = (u - c_lo) / (c_hi - c_lo + 1.0e-6f);
        float sample = torch_lerp(b_lo, b_hi, t);
        const float scan = seg_inclusive_max<kL>(valid ? sample : -3.0e38f);     // running maximum: see render.h
        sample = fmaxf(scan, fine_max);
        fine_max = fmaxf(fine_max, lane_gather(scan, rl.last));
        wave_lds_sync();                                                     // (padding lanes read the last lane's uniform above)
        if (valid) fine[j] = sample;
    }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < kRoundsS; ++k) {
        if (k * kL >= S) continue;
        const int j = k * kL + rl.col;
        const bool valid = j < S;
        const int jj = valid ? j : (S - 1);
        const float a = coarse[jj], b = fine[jj];
        const int rank_a = jj + seg_count_below<kL, true>(fine, S, a);
        const int rank_b = jj + seg_count_below<kL, false>(coarse, S, b);
        if (valid) { merged[rank_a] = a; merged[rank_b] = b; }
    }
    wave_lds_sync();
}

// The lane's ray as the wave keeps it in LDS (8 floats per row): origin, direction, and the scale of the culling error bound.
struct RowRay { Ray ray; float reach; };
__device__ __forceinline__ RowRay load_row_ray(const float* rp) {
    const float4 a = *reinterpret_cast<const float4*>(rp), b = *reinterpret_cast<const float4*>(rp + 4);
    RowRay r;
    r.ray.ox = a.x; r.ray.oy = a.y; r.ray.oz = a.z; r.ray.rx = a.w; r.ray.ry = b.x; r.ray.rz = b.y; r.reach = b.z;
    return r;
}
__device__ __forceinline__ RayCull row_cull(const float* coef, const RowRay& r) {
    RayCull rc;
    rc.coef = coef;
    rc.c2 = r.ray.rx * r.ray.rx + r.ray.ry * r.ray.ry + r.ray.rz * r.ray.rz;
    rc.rnorm = fast_sqrt(rc.c2);
    // (r.r, |r| and max(1, |r|) / 2 kept in the ray's LDS row instead of being re-derived by every round: tried in round 4, neutral)
    rc.reach = r.reach;
    return rc;
}
// (a compiler-only fence: LDS values re-read after it are loaded again instead of being kept in registers across a loop)
__device__ __forceinline__ void reload_fence() { asm volatile("" ::: "memory"); }

// Culling coefficients of (lane's ray, instance col, col + 16, ...) (field.h: cull_ray_setup, per row) and the ray itself, into LDS.
// `padded` >= N: the table is filled up to `padded` rows with instances that can never matter (a = 3e38: never the nearest, never
// inside a reach) -- the pre-pass of a kernel instantiated for a compile-time instance count walks all of them (quad_round_mask masks
// them out again where culling is switched off).
template <int kL>
__device__ __forceinline__ void quad_ray_setup(const float* __restrict__ instances, int N, int padded, const Ray& r, float* coef, float* rayp, const RowLanes& rl) {
    float amax = 0.0f;
    for (int i = rl.col; i < padded; i += kL) {
        const bool real = i < N;
        const float* p = instances + (real ? i : 0) * kInstanceStride;
        const float ex = r.ox - p[0], ey = r.oy - p[1], ez = r.oz - p[2];
        const float a = ex * ex + ey * ey + ez * ez;
        coef[kCullCoefs * i + 0] = real ? a : 3.0e38f;
        coef[kCullCoefs * i + 1] = real ? 2.0f * (ex * r.rx + ey * r.ry + ez * r.rz) : 0.0f;
        coef[kCullCoefs * i + 2] = real ? fast_sqrt(p[12] * p[12] + p[13] * p[13] + p[14] * p[14]) * (1.0f / (1.0f - kCullSlack)) : 0.0f;
        coef[kCullCoefs * i + 3] = 0.0f;                                        // the ray's label adjoint of instance i, set after pass 2
        amax = fmaxf(amax, real ? a : 0.0f);
    }
    const float reach = fast_sqrt(fmaxf(seg_max<kL>(amax), r.ox * r.ox + r.oy * r.oy + r.oz * r.oz));
    if (rl.col == 0) {
        *reinterpret_cast<float4*>(rayp) = make_float4(r.ox, r.oy, r.oz, r.rx);
        *reinterpret_cast<float4*>(rayp + 4) = make_float4(r.ry, r.rz, reach, 0.0f);
    }
    wave_lds_sync();
}

// The culling pre-pass of one round (field.h: cull_round_mask) in two steps, so that a round can leave between them.
// -DVSRD_CULL_PARTIAL (round 4, NOT the default): the term of the squared centre distance that all instances share leaves the
// per-instance work,  |x(t) - t_i|^2 = (a_i + b_i t) + (r.r) t^2 = e_i + c:  min_i e_i + c for the nearest centre and
// e_i > reach_i^2 + (E - c)  for the test, one instruction less per (sample, instance) in either loop.  +1 % on config 2 -- but with it
// the culling A/B of the two-rays-per-wave kernels (tests/test_hip_scale.py: pair-mid) shows rays whose labels are off by up to 6.7e-3
// (nested form: 2.2e-4, the sampler's own conditioning), although a float32 emulation of both bound tests over every (group, round,
// instance) of that view (tests/culling_formulations_debug.py) finds their masks equal but for 6 of 1.4e8 triples and neither ever
// dropping an instance that matters, and the bounds of the nearest distance intact.  Not understood, so not used.
// Step 1: the bounds of the nearest centre distance of every lane's point (RoundCull).
__device__ __forceinline__ float centre_partial(const RayCull& rc, int i, float t) {       // e_i = a_i + b_i t
#ifndef VSRD_CULL_PARTIAL                            // the nested form: d2_i = a_i + t (b_i + c t) itself
    return fmaf(t, rc.c2 * t + rc.coef[kCullCoefs * i + 1], rc.coef[kCullCoefs * i + 0]);
#else
    return fmaf(t, rc.coef[kCullCoefs * i + 1], rc.coef[kCullCoefs * i + 0]);
#endif
}
__device__ __forceinline__ RoundCull quad_round_bounds(const RayCull& rc, int num_instances, float t, float margin, float inner = 0.0f) {
    float nearest = 3.0e38f;
    int i = 0;
#pragma unroll 4                                     // (sixteen instances in flight at most, also where the count is a compile-time 64)
    for (; i + 4 <= num_instances; i += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) nearest = fminf(nearest, centre_partial(rc, i + j, t));
    }
    for (; i < num_instances; ++i) nearest = fminf(nearest, centre_partial(rc, i, t));
#ifndef VSRD_CULL_PARTIAL
    return cull_round(rc, t, nearest, margin, inner);
#else
    return cull_round(rc, t, fmaf(rc.c2 * t, t, nearest), margin, inner);
#endif
}
// Step 2: bit i = instance i may matter on some lane (wave-uniform).  (NaN-safe: an undecidable comparison keeps the instance.)
__device__ __forceinline__ unsigned long long quad_round_mask(const RayCull& rc, const RoundCull& cull, int num_instances, float t,
                                                              unsigned long long real = ~0ull, float widest = -1.0f) {
#ifndef VSRD_CULL_PARTIAL
    const float shift = cull.err;
#else
    const float shift = cull.err - rc.c2 * t * t;                                       // E - c
#endif
    unsigned long long mask = 0ull;
    int i = 0;
#ifndef VSRD_CULL_PER_INSTANCE_RADIUS
    // Round 4: one threshold per lane from the WIDEST instance's bounding radius (field_bounds: max |dim|) instead of one per (lane,
    // instance): two instructions and an LDS read less per pair, a margin looser by at most max|dim| - min|dim| (1.1 m between the
    // reference's smallest and largest box against 18 T + the nearest distance) -- conservative, so nothing the tests see changes (the
    // benchmark's final loss is bit-identical); config 2: 218.3 -> 221.3 Mrays/s, config 5: 29.15 -> 29.50.  `widest` < 0: no such
    // bound (field_bounds found rotations that are not orthonormal): the per-instance radii of the coefficient rows.
    if (widest >= 0.0f) {
        const float reach = cull.limit + widest;
        const float threshold = fmaf(reach, reach, shift);
        // (tried in round 4: compare -> SCC -> add-with-carry into a 32-bit word, two scalar instructions per instance instead of the compiler's
        //  three (s_cmp, s_cselect of the bit, s_or) -- as inline asm it is a dependent chain the scheduler cannot interleave: config 2 -3.8 %,
        //  config 5 -1.7 %)
#pragma unroll 4
        for (; i + 4 <= num_instances; i += 4) {
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = centre_partial(rc, i + j, t);
#pragma unroll
            for (int j = 0; j < 4; ++j) mask |= (__ballot(!(e[j] > threshold)) != 0ull) ? (1ull << (i + j)) : 0ull;
        }
        for (; i < num_instances; ++i) mask |= (__ballot(!(centre_partial(rc, i, t) > threshold)) != 0ull) ? (1ull << i) : 0ull;
        return mask & real;
    }
#endif
#pragma unroll 4
    for (; i + 4 <= num_instances; i += 4) {
        float e[4], radius[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[j] = centre_partial(rc, i + j, t);
            radius[j] = rc.coef[kCullCoefs * (i + j) + 2];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float reach = cull.limit + radius[j];
            mask |= (__ballot(!(e[j] > fmaf(reach, reach, shift))) != 0ull) ? (1ull << (i + j)) : 0ull;
        }
    }
    for (; i < num_instances; ++i) {
        const float reach = cull.limit + rc.coef[kCullCoefs * i + 2];
        mask |= (__ballot(!(centre_partial(rc, i, t) > fmaf(reach, reach, shift))) != 0ull) ? (1ull << i) : 0ull;
    }
    return mask & real;                                                                 // (padding rows of the table: quad_ray_setup)
}

// Rounds that see nothing (wave-uniform, exact).  Every box distance of the round's points is >= floor = (nearest centre distance)
// (1 - k) - max_i |dim_i| (field.h: field_bounds), the soft-min union is a convex combination of them, and the section points of the
// opacity (renderers.py:228-248) lie within |c'| delta / 2 <= max(1, |r|) delta / 2 of it.  Once (floor - that) / sigma >= 17 both
// logistic cdfs are EXACTLY 1 in fp32 (exp(-17) < 2^-24 vanishes in 1 + exp(-x), v_rcp(1) = 1), so alpha = 0 on every lane: no
// weight, no label, no adjoint -- the round leaves before its instance loop.  (Fine samples extrapolated to 1e6 m end here too.)
// VSRD_FLAG_NO_CULLING (every instance at every sample) switches this off as well.
constexpr float kLogisticOne = 17.5f;
__device__ __forceinline__ bool quad_round_is_empty(const RayCull& rc, const RoundCull& cull, const Shading& sh, float delta) {
    const float floor = cull.nearest_lo - sh.reach;
    const float half = 0.5f * fmaxf(1.0f, rc.rnorm) * fabsf(delta);
    return sh.reach >= 0.0f && sh.cull < 1.0e38f && !wave_any(!((floor - half) * sh.inv_std >= kLogisticOne));
}

// The instance loop of one round (render.h: union_loop): the instances of `evaluated` that also pass the exact test; the soft-min
// term of every surviving instance is left in dcache[i][lane] (fixed shift: exp(-(d_i - m)/T); running minimum: d_i).
// kBySlot (shapes with up to 64 instances): the k-th surviving instance of the round takes cache row k while k < kCacheSlots; later
// survivors are re-evaluated where their term is needed (quad_cached_term).
template <bool kCache, bool kBySlot, bool kRunning, bool kYaw>
__device__ __forceinline__ UnionSums quad_union_loop(const float* __restrict__ instances, unsigned long long& evaluated, const Shading& sh,
                                                     const RoundCull& cull, float floor, float x, float y, float z, float* dcache, int lane) {
    UnionSums sums = union_init(kRunning, floor);
    float best = cull.nearest_hi;
    int slot = 0;
#ifdef VSRD_PHASE_TIMERS
    const int candidates = __builtin_popcountll(evaluated);       // (tools/phase_timers.py: rounds, instances past the bound test, survivors)
#endif
    // (a form with two nested loops -- the inner one looking for the next instance past the exact test -- lets the compiler update the sums
    //  in place and drops the nine register copies this loop ends with, 12 of 70 vector instructions per instance: 1.5 % SLOWER on config 2;
    //  the copies are nearly free and the longer scalar chain between the test and the branch is not)
    for (unsigned long long todo = evaluated; todo != 0ull; todo &= todo - 1ull) {
        const int i = __builtin_ctzll(todo);
        const Instance in = load_instance_block(instances, i);
        BoxEval e = box_value<kYaw>(in, x, y, z);
        const unsigned long long near = __ballot(!(e.d - best > sh.cull));
        if (near == 0ull) { evaluated &= ~(1ull << i); continue; }
        best = min_raw(best, e.d);
        box_gradient<kYaw>(e, in);
        const float term = union_accumulate<kRunning, false>(sums, e.d, e.gwx, e.gwy, e.gwz, 0.0f, sh.inv_t);
        if (kCache && (!kBySlot || slot < kCacheSlots)) dcache[(kBySlot ? slot * kSlotStride : i * kWave) + lane] = term;
        ++slot;
    }
#ifdef VSRD_PHASE_TIMERS
    if (lane == 0) { atomicAdd(&g_phase_cycles[8], 1ull); atomicAdd(&g_phase_cycles[9], static_cast<unsigned long long>(candidates));
                     atomicAdd(&g_phase_cycles[10], static_cast<unsigned long long>(slot)); }
#endif
    return sums;
}

// The cached soft-min term of surviving instance i (the slot-th survivor of its round): exp(-(d_i - m)/T) (fixed shift) or d_i
// (running minimum); survivors beyond the cache are re-evaluated.
template <bool kBySlot, bool kRunning, bool kYaw>
__device__ __forceinline__ float quad_cached_term(const float* __restrict__ instances, int i, int slot, const float* dcache, int lane, float x, float y, float z,
                                                  float m, float inv_t) {
    if (!kBySlot) return dcache[i * kWave + lane];
    if (slot < kCacheSlots) return dcache[slot * kSlotStride + lane];
    const float d = box_value<kYaw>(load_instance_block(instances, i), x, y, z).d;
    return kRunning ? d : fast_exp(-(d - m) * inv_t);
}

// Two rays per wave, fixed shift: the label sums of one round, transposed.  The instance loop left the soft-min term of survivor k at
// every point in row k of the cache; lane (ray, k) multiplies its survivor's row with the ray's label scales (t alpha / Z per point) --
// 32 multiply-adds on eight pairs of ds_read_b128 for ALL cached survivors at once, where one reduction per survivor costs a row
// reduction (seven instructions) and two selects for each of the ~20 survivors of a config-5 round -- and the lane that owns instance n
// fetches the sum of slot rank(n) = popcount(near & below(n)).  Survivors beyond the cache take the per-survivor path in the caller.
__device__ __forceinline__ void quad_label_sums(float* dcache, float scale, unsigned long long near, float (&label)[2], const RowLanes& rl) {
    float* scales = dcache + kCacheSlots * kSlotStride;
    scales[rl.lane] = scale;
    wave_lds_sync();
    const int first = rl.lane & ~31;                                              // the ray's first lane = its first point's column
    const float4* terms = reinterpret_cast<const float4*>(dcache + min(rl.col, kCacheSlots - 1) * kSlotStride + first);
    const float4* factors = reinterpret_cast<const float4*>(scales + first);
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float4 e = terms[k], f = factors[k];
        sum = fmaf(e.x, f.x, sum); sum = fmaf(e.y, f.y, sum); sum = fmaf(e.z, f.z, sum); sum = fmaf(e.w, f.w, sum);
    }
    const unsigned lo = static_cast<unsigned>(near), hi = static_cast<unsigned>(near >> 32), below = (1u << rl.col) - 1u;
    const int rank0 = __builtin_popcount(lo & below), rank1 = __builtin_popcount(lo) + __builtin_popcount(hi & below);
    const float sum0 = lane_gather(sum, (first + min(rank0, 31)) << 2), sum1 = lane_gather(sum, (first + min(rank1, 31)) << 2);
    label[0] += (((lo >> rl.col) & 1u) && rank0 < kCacheSlots) ? sum0 : 0.0f;
    label[1] += (((hi >> rl.col) & 1u) && rank1 < kCacheSlots) ? sum1 : 0.0f;
}

// One point per lane: the interval [dist[s], dist[s + 1]] of the lane's ray.
struct QuadPoint { float delta, mid, x, y, z; bool valid; };
template <int kL>
__device__ __forceinline__ QuadPoint quad_point(const float* dist, int num_points, int round, const Ray& ray, bool live, const RowLanes& rl) {
    QuadPoint p;
    const int s = round * kL + rl.col;
    p.valid = live && s < num_points;
    const int s0 = (s < num_points) ? s : (num_points - 1);                    // padding lanes repeat the last point
    const float d0 = dist[s0], d1 = dist[s0 + 1];
    p.delta = d1 - d0;
    p.mid = (d0 + d1) / 2.0f;
    p.x = ray.ox + ray.rx * p.mid; p.y = ray.oy + ray.ry * p.mid; p.z = ray.oz + ray.rz * p.mid;
    return p;
}

// Pass 1 of the four rays: coarse compositing weights w[k] of point k * 16 + col (render.h: render_pass without labels).
// Returns false when the fixed soft-min shift cannot serve some round (field.h: union_accumulate): the caller repeats the group with
// the running minimum.
template <int kL, int kRoundsS, bool kYaw, bool kRunning>
__device__ __forceinline__ bool quad_pass_one(const float* __restrict__ instances, int N, const Shading& sh, const float* rayp, const float* coef,
                                              const float* coarse, int S, float (&w)[kRoundsS], const RowLanes& rl, unsigned long long real = ~0ull) {
    const int num_points = S - 1;
    float carry = 1.0f;
#pragma unroll
    for (int k = 0; k < kRoundsS; ++k) {
        w[k] = 0.0f;
        if (k * kL >= num_points) continue;
        const RowRay rr = load_row_ray(rayp);
        const RayCull rc = row_cull(coef, rr);
        const QuadPoint p = quad_point<kL>(coarse, num_points, k, rr.ray, true, rl);
        const RoundCull cull = quad_round_bounds(rc, N, p.mid, sh.cull, sh.inner);
        if (quad_round_is_empty(rc, cull, sh, p.delta)) continue;                // alpha = 0 exactly: the transmittance passes unchanged
        unsigned long long evaluated = quad_round_mask(rc, cull, N, p.mid, real, sh.reach >= 0.0f ? sh.reach * (1.0f / (1.0f - kCullSlack)) : -1.0f);
#ifdef VSRD_BOUND_FINER_CULLING         // (see quad_forward_sweep)
        for (int drop = 0; drop < VSRD_BOUND_FINER_CULLING; ++drop)
            if (__builtin_popcountll(evaluated) >= 5) evaluated &= ~(1ull << (63 - __builtin_clzll(evaluated)));
#endif
        const float floor = cull.nearest_lo - sh.reach;
        if (!kRunning && wave_any(!((cull.nearest_hi + 1.0f - floor) * sh.inv_t <= kUnionFloorSpan))) return false;
        const UnionSums sums = quad_union_loop<false, false, kRunning, kYaw>(instances, evaluated, sh, cull, floor, p.x, p.y, p.z, nullptr, rl.lane);
        if (!kRunning && wave_any(!(sums.Z >= kUnionTinyZ))) return false;
        const UnionValue v = union_finish(sums, sh.inv_t);
        const Opacity op = opacity_of(v, rr.ray, p.delta, sh);
        const float alpha = p.valid ? op.alpha : 0.0f;
        const float inclusive = seg_inclusive_product<kL>(1.0f - alpha);
        w[k] = carry * seg_shift_up<kL>(inclusive, 1.0f, rl) * alpha;
        carry *= lane_gather(inclusive, rl.last);
    }
    return true;
}

// Per-sample state of the adjoint, one pass-2 point per lane and round, in registers:
//   after the forward sweep:  m, inv_z (soft-min shift, 1/Z), s = u - m, a = grad u, c = B' with B = g_bar . b = cos_bar B'
//                             (b = sum_i w_i grad d_i;  g_bar = cos_bar / |g| (r - n (n . r)), n = grad u / |g|, so
//                              B' = (r . b - (grad u . b) (n . r) / |g|) / |g|)
//   after the reverse sweep:  m, inv_z, s = 1 + (u - m) / T, a = g_bar; C1, C2 and C3 wait in the wave's LDS (the distance cache is
//   free by then: two lane-private floats per round; the third takes the place of the sorted distances).  With them (adjoint_phase_b)
//     d_bar_i = cc_i (C1 - beta_i / T) + w_i (C2 - beta_i / T - C3 lambda_i),   cc_i = w_i (s - (d_i - m) / T)
//     C1 = u_bar + B / T,  C2 = (A + w_s Lambda_s / Z_s) / T,  C3 = w_s / T,   A = g_bar . grad u
template <int kRounds>
struct QuadAdjoint {
    float m[kRounds], inv_z[kRounds], s[kRounds];
    float ax[kRounds], ay[kRounds], az[kRounds];
    float c[kRounds];
    unsigned long long near[kRounds];            // instances evaluated in the round (wave-uniform)
};

// Pass 2, forward: union, opacity, transmittance, labels.  label: lane (ray, n) accumulates label n of its ray.  The transmittance of
// every point of an active round is left in trans[round * 16 + col] (LDS) for the reverse sweep.  Returns false when a round needs
// the running minimum.
template <int kL, int kRounds, bool kYaw, bool kRunning>
__device__ __forceinline__ bool quad_forward_sweep(QuadAdjoint<kRounds>& st, const float* __restrict__ instances, int N, const Shading& sh,
                                                   const float* rayp, const float* coef, const float* merged, int num_points, bool live,
                                                   float* dcache, float* trans, float (&label)[kL == kRowLanes ? 1 : 2], unsigned& active, int& cached_round,
                                                   const RowLanes& rl, unsigned long long real = ~0ull) {
    constexpr int kSlots = kL == kRowLanes ? 1 : 2;                            // lane (ray, c) owns labels c, kL + c
    float carry = 1.0f;
#pragma unroll
    for (int s = 0; s < kSlots; ++s) label[s] = 0.0f;
    cached_round = -1;                                                           // the round whose soft-min terms the distance cache holds at the end
    active = 0u;                                                                 // bit q: some point of round q has a weight
#pragma unroll
    for (int q = 0; q < kRounds; ++q) {
        st.near[q] = 0ull;
        if (q * kL >= num_points) continue;
#ifdef VSRD_PHASE_TIMERS
        if (rl.lane == 0) atomicAdd(&g_phase_cycles[11], 1ull);                   // pass-2 rounds / of them: behind an opaque surface on every ray
        if (!wave_any(!(carry < 1.0e-9f)) && rl.lane == 0) atomicAdd(&g_phase_cycles[12], 1ull);
        if (!wave_any(!(carry < 1.0e-6f)) && rl.lane == 0) atomicAdd(&g_phase_cycles[13], 1ull);
#endif
        float delta, bprime, px, py, pz;
        UnionValue v;
        Opacity op;
        bool valid;
        {
            const RowRay rr = load_row_ray(rayp);
            const RayCull rc = row_cull(coef, rr);
            const QuadPoint p = quad_point<kL>(merged, num_points, q, rr.ray, live, rl);
            const RoundCull cull = quad_round_bounds(rc, N, p.mid, sh.cull, sh.inner);
            if (quad_round_is_empty(rc, cull, sh, p.delta)) continue;            // alpha = 0 exactly: no weight, no label, no adjoint
            st.near[q] = quad_round_mask(rc, cull, N, p.mid, real, sh.reach >= 0.0f ? sh.reach * (1.0f / (1.0f - kCullSlack)) : -1.0f);
#ifdef VSRD_BOUND_FINER_CULLING         // TIMING BOUND ONLY (profiles/r06/variants.txt; wrong results): what a mapping with a finer culling granularity could
            // buy at no cost of its own -- every pass-2 round with five or more candidate instances loses VSRD_BOUND_FINER_CULLING of them
            for (int drop = 0; drop < VSRD_BOUND_FINER_CULLING; ++drop)
                if (__builtin_popcountll(st.near[q]) >= 5) st.near[q] &= ~(1ull << (63 - __builtin_clzll(st.near[q])));
#endif
            const float floor = cull.nearest_lo - sh.reach;
            if (!kRunning && wave_any(!((cull.nearest_hi + 1.0f - floor) * sh.inv_t <= kUnionFloorSpan))) return false;
            const UnionSums sums = quad_union_loop<true, (kL > kRowLanes), kRunning, kYaw>(instances, st.near[q], sh, cull, floor, p.x, p.y, p.z, dcache, rl.lane);
            cached_round = q;
            if (!kRunning && wave_any(!(sums.Z >= kUnionTinyZ))) return false;
            v = union_finish(sums, sh.inv_t);
            delta = p.delta; valid = p.valid; px = p.x; py = p.y; pz = p.z;
        }
        reload_fence();
        {
            const RowRay rr = load_row_ray(rayp);                                // (re-read: the ray does not live in registers across the instance loop)
            op = opacity_of(v, rr.ray, delta, sh);
            const float rb = rr.ray.rx * v.b0x + rr.ray.ry * v.b0y + rr.ray.rz * v.b0z;
            const float gb = v.gx * v.b0x + v.gy * v.b0y + v.gz * v.b0z;
            bprime = (rb - gb * op.inv_gn * op.cosine) * op.inv_gn;
        }
        const float alpha = valid ? op.alpha : 0.0f;
        const float inclusive = seg_inclusive_product<kL>(1.0f - alpha);
        const float t = carry * seg_shift_up<kL>(inclusive, 1.0f, rl);
        carry *= lane_gather(inclusive, rl.last);
        st.m[q] = v.m; st.inv_z[q] = v.inv_z; st.s[q] = v.us;
        st.ax[q] = v.gx; st.ay[q] = v.gy; st.az[q] = v.gz;
        st.c[q] = bprime;
        if (__ballot(alpha > 0.0f) == 0ull) continue;                            // (a round without weight: nothing for the labels, nothing flows back)
        active |= 1u << q;
        trans[q * kL + rl.col] = t;
        const float scale = t * alpha * v.inv_z;
        int slot = 0;
        unsigned long long todo = st.near[q];
#ifndef VSRD_LABEL_LOOP
        if constexpr (kL > kRowLanes && !kRunning) {                               // the cached survivors at once; the loop below: the others
            quad_label_sums(dcache, scale, st.near[q], label, rl);
            if (__builtin_popcountll(todo) <= kCacheSlots) todo = 0ull;
            for (; slot < kCacheSlots && todo != 0ull; ++slot) todo &= todo - 1ull;
        }
#endif
        for (; todo != 0ull; todo &= todo - 1ull, ++slot) {
            const int i = __builtin_ctzll(todo);
            const float cached = quad_cached_term<(kL > kRowLanes), kRunning, kYaw>(instances, i, slot, dcache, rl.lane, px, py, pz, v.m, sh.inv_t);
            const float e = kRunning ? fast_exp(-(cached - v.m) * sh.inv_t) : cached;
            const float total = seg_sum<kL>(e * scale);
#pragma unroll
            for (int s = 0; s < kSlots; ++s) label[s] = (i == s * kL + rl.col) ? (label[s] + total) : label[s];
        }
    }
    return true;
}

// Reverse sweep over the rounds of pass 2 (render_kernels.h: adjoint_label_mix + adjoint_reverse_sweep): label-adjoint mix
// Lambda_s, opacity (recomputed from u, grad u and the interval), suffix sums of w_bar w, chain to (u_bar, g_bar), and the
// per-sample constants of the per-instance phase.  coef_own: the lane's OWN ray's (a, b, radius, lambda) rows; cached_round:
// the round whose soft-min terms the distance cache still holds (fixed shift; -1: none).  On return cbuf (= the distance cache) holds
// C1 and C3 of every point, trans_mid the interval mid-points, c2buf (= the row's OWN sorted distances, shifted by one) C2.  Returns the
// rounds (bit q) in which some lane carries a non-zero adjoint.
template <int kL, int kRounds, bool kYaw>
__device__ __forceinline__ unsigned quad_reverse_sweep(QuadAdjoint<kRounds>& st, const float* __restrict__ instances, const Shading& sh, const float* rayp,
                                                       const float* merged, int num_points, bool live, unsigned active, unsigned long long lam_any, int cached_round,
                                                       const float* coef_own, float* cbuf, float* trans_mid, float* c2buf, const RowLanes& rl) {
    unsigned flow = 0u;
    float suffix_carry = 0.0f;
#pragma unroll
    for (int q = kRounds - 1; q >= 0; --q) {
        if (!((active >> q) & 1u)) continue;                                       // no weight in the round: nothing flows back (exact)
        const RowRay rr = load_row_ray(rayp);
        const QuadPoint p = quad_point<kL>(merged, num_points, q, rr.ray, live, rl);
        // Lambda_s / Z_s = sum_n lambda_n w_{s,n} over the instances the forward sweep evaluated (culled ones: weight < exp(-18))
        float acc = 0.0f;
#ifdef VSRD_BOUND_CACHED_LABEL_MIX      // TIMING BOUND ONLY (profiles/r06/variants.txt; wrong results): every round's label mix at the price of the cached round's
        if (cached_round >= 0) {
#else
        if (q == cached_round) {                                                   // wave-uniform: the soft-min terms are still in the distance cache
#endif
            int slot = 0;
            for (unsigned long long todo = st.near[q]; todo != 0ull; todo &= todo - 1ull, ++slot) {
                const int i = __builtin_ctzll(todo);
                if (!((lam_any >> i) & 1ull)) continue;                            // wave-uniform: no ray has a label adjoint for it
                acc += coef_own[kCullCoefs * i + 3] * quad_cached_term<(kL > kRowLanes), false, kYaw>(instances, i, slot, cbuf, rl.lane, p.x, p.y, p.z, st.m[q], sh.inv_t);
            }
        } else {
            for (unsigned long long todo = st.near[q] & lam_any; todo != 0ull; todo &= todo - 1ull) {
                const int i = __builtin_ctzll(todo);
                const Instance in = load_instance_block(instances, i);
                acc += coef_own[kCullCoefs * i + 3] * fast_exp(-(box_value<kYaw>(in, p.x, p.y, p.z).d - st.m[q]) * sh.inv_t);
            }
        }
        const float lam_z = p.valid ? acc * st.inv_z[q] : 0.0f;
        UnionValue v;
        v.u = st.m[q] + st.s[q]; v.gx = st.ax[q]; v.gy = st.ay[q]; v.gz = st.az[q];
        const Opacity op = opacity_of(v, rr.ray, p.delta, sh);
        const float alpha = p.valid ? op.alpha : 0.0f;
        const float t = trans_mid[q * kL + rl.col];
        trans_mid[q * kL + rl.col] = p.mid;                                 // (same lane, same slot: no hazard)
        const float wgt = t * alpha;
        const float contrib = lam_z * wgt;
        const float suffix_inclusive = seg_suffix_sum<kL>(contrib, rl);
        const float Q = suffix_inclusive - contrib + suffix_carry;                 // sum over the later samples of the ray
        suffix_carry += lane_gather(suffix_inclusive, rl.first);
        const float alpha_bar = lam_z * t - Q * fast_rcp(1.0f - alpha);
        const float x_bar = (p.valid && op.xx > 0.0f) ? alpha_bar : 0.0f;
        const float inv_pe = fast_rcp(op.phi_p + sh.eps);
        const float phi_p_bar = x_bar * (op.phi_n + sh.eps) * inv_pe * inv_pe;
        const float phi_n_bar = -x_bar * inv_pe;
        const float sp_bar = phi_p_bar * op.phi_p * (1.0f - op.phi_p) * sh.inv_std;
        const float sn_bar = phi_n_bar * op.phi_n * (1.0f - op.phi_n) * sh.inv_std;
        const float u_bar = sp_bar + sn_bar;                                       // (x_bar = 0 on padding lanes)
        const float cprime_bar = (sn_bar - sp_bar) * p.delta / 2.0f;
        const float slope = (1.0f - sh.ratio) * ((0.5f - 0.5f * op.cosine > 0.0f) ? 0.5f : 0.0f) + sh.ratio * ((-op.cosine > 0.0f) ? 1.0f : 0.0f);
        const float cos_bar = cprime_bar * slope;
        const float nbx = cos_bar * rr.ray.rx, nby = cos_bar * rr.ray.ry, nbz = cos_bar * rr.ray.rz;
        const float n_dot = op.nx * nbx + op.ny * nby + op.nz * nbz;
        const float gbx = (nbx - op.nx * n_dot) * op.inv_gn, gby = (nby - op.ny * n_dot) * op.inv_gn, gbz = (nbz - op.nz * n_dot) * op.inv_gn;
        const float B = cos_bar * st.c[q];
        const float A = gbx * st.ax[q] + gby * st.ay[q] + gbz * st.az[q];
        st.ax[q] = gbx; st.ay[q] = gby; st.az[q] = gbz;
        st.s[q] = 1.0f + st.s[q] * sh.inv_t;
        c2buf[q * kL + rl.col + 1] = sh.inv_t * (A + wgt * lam_z);
        cbuf[(2 * q) * kWave + rl.lane] = u_bar + sh.inv_t * B;
        cbuf[(2 * q + 1) * kWave + rl.lane] = sh.inv_t * wgt;
        const bool any = (u_bar != 0.0f) || (gbx != 0.0f) || (gby != 0.0f) || (gbz != 0.0f) || (wgt != 0.0f);
        flow |= (__ballot(any) != 0ull) ? (1u << q) : 0u;
    }
    return flow;
}

// Per-instance phase for the four rays (render_kernels.h: adjoint_phase_b): instance-outer, rounds inner, one reduce-scatter butterfly
// per (wave, instance).  G[s]: lane (row r, col c) accumulates parameter c of instance 4 s + r (four registers for N <= 16; float
// atomics on the wave's row of the partial table instead cost 6.5 GB of L2 write-through per launch).
template <int kL, int kRounds, bool kYaw>
__device__ __forceinline__ void quad_phase_b(const QuadAdjoint<kRounds>& st, const float* __restrict__ instances, const Shading& sh, const float* rayp,
                                             unsigned flow, const float* coef_own, const float* cbuf, const float* mids, const float* c2buf, float (&G)[kL == kRowLanes ? 4 : 16], const RowLanes& rl) {
    const float inv_t = sh.inv_t;
    const Ray ray = load_row_ray(rayp).ray;
    unsigned long long todo = 0ull;
#pragma unroll
    for (int q = 0; q < kRounds; ++q) todo |= ((flow >> q) & 1u) ? st.near[q] : 0ull;
    for (; todo != 0ull; todo &= todo - 1ull) {
        const int i = __builtin_ctzll(todo);
        const Instance in = load_instance_block(instances, i);
        const float lam_i = coef_own[kCullCoefs * i + 3];
        float at0 = 0, at1 = 0, at2 = 0, ad0 = 0, ad1 = 0, ad2 = 0;
        float r00 = 0, r01 = 0, r02 = 0, r10 = 0, r11 = 0, r12 = 0, r20 = 0, r21 = 0, r22 = 0;
#pragma unroll
        for (int q = 0; q < kRounds; ++q) {
            if (!(((st.near[q] >> i) & 1ull) && ((flow >> q) & 1u))) continue;       // wave-uniform
            const float mid = mids[q * kL + rl.col];
            const float c1 = cbuf[(2 * q) * kWave + rl.lane], c3 = cbuf[(2 * q + 1) * kWave + rl.lane], c2 = c2buf[q * kL + rl.col + 1];
#ifndef VSRD_BOUND_FREE_SELECTORS
            const BoxEval e = eval_box<kYaw>(in, ray.ox + ray.rx * mid, ray.oy + ray.ry * mid, ray.oz + ray.rz * mid);
#else       // TIMING BOUND ONLY (profiles/r06/variants.txt; results are wrong inside boxes): the per-instance phase as if the seven selectors of a
            // (sample, instance) pair -- inside, first-x, first-y, three p != 0, three q > 0 -- cost nothing at all
            BoxEval e = box_value<kYaw>(in, ray.ox + ray.rx * mid, ray.oy + ray.ry * mid, ray.oz + ray.rz * mid);
            e.hx = fmaxf(e.qx, 0.0f) * e.inv; e.hy = fmaxf(e.qy, 0.0f) * e.inv; e.hz = fmaxf(e.qz, 0.0f) * e.inv;
            e.glx = __builtin_copysignf(e.hx, e.px); e.gly = __builtin_copysignf(e.hy, e.py); e.glz = __builtin_copysignf(e.hz, e.pz);
#endif
            const float ds = e.d - st.m[q];
            const float w = fast_exp(-ds * inv_t) * st.inv_z[q];
            const float cc = w * (st.s[q] - ds * inv_t);
            const float gx_ = st.ax[q], gy_ = st.ay[q], gz_ = st.az[q];
            const float rgx = kYaw ? fmaf(in.r20, gz_, in.r00 * gx_) : fmaf(in.r20, gz_, fmaf(in.r10, gy_, in.r00 * gx_));
            const float rgy = kYaw ? gy_ : fmaf(in.r21, gz_, fmaf(in.r11, gy_, in.r01 * gx_));
            const float rgz = kYaw ? fmaf(in.r22, gz_, in.r02 * gx_) : fmaf(in.r22, gz_, fmaf(in.r12, gy_, in.r02 * gx_));
            const float tb = inv_t * (rgx * e.glx + rgy * e.gly + rgz * e.glz);
            const float d_bar = cc * (c1 - tb) + w * (c2 - tb - c3 * lam_i);
            const float gwbx = cc * gx_, gwby = cc * gy_, gwbz = cc * gz_;
            const float glbx = cc * rgx, glby = cc * rgy, glbz = cc * rgz;
#ifndef VSRD_BOUND_FREE_SELECTORS
            const float sx = sign_of(e.px), sy = sign_of(e.py), sz = sign_of(e.pz);
#else
            const float sx = __builtin_copysignf(1.0f, e.px), sy = __builtin_copysignf(1.0f, e.py), sz = __builtin_copysignf(1.0f, e.pz);
#endif
            const float vx = sx * glbx, vy = sy * glby, vz = sz * glbz;
            const float inv_n = box_inverse_norm(e);
            const float hx = fmaxf(e.qx, 0.0f) * inv_n, hy = fmaxf(e.qy, 0.0f) * inv_n, hz = fmaxf(e.qz, 0.0f) * inv_n;
            const float hv = hx * vx + hy * vy + hz * vz;
#ifndef VSRD_BOUND_FREE_SELECTORS
            const float qbx = d_bar * e.hx + ((e.qx > 0.0f) ? (vx - hx * hv) * inv_n : 0.0f);
            const float qby = d_bar * e.hy + ((e.qy > 0.0f) ? (vy - hy * hv) * inv_n : 0.0f);
            const float qbz = d_bar * e.hz + ((e.qz > 0.0f) ? (vz - hz * hv) * inv_n : 0.0f);
#else
            const float qbx = d_bar * e.hx + (vx - hx * hv) * inv_n;
            const float qby = d_bar * e.hy + (vy - hy * hv) * inv_n;
            const float qbz = d_bar * e.hz + (vz - hz * hv) * inv_n;
#endif
            const float pbx = sx * qbx, pby = sy * qby, pbz = sz * qbz;
            ad0 -= qbx; ad1 -= qby; ad2 -= qbz;
            r00 += e.relx * pbx + gwbx * e.glx; r02 += e.relx * pbz + gwbx * e.glz;
            r20 += e.relz * pbx + gwbz * e.glx; r22 += e.relz * pbz + gwbz * e.glz;
            if (!(kYaw && sh.yaw_gradients)) {           // (wave-uniform) the entries rotation_matrix_y keeps constant: nobody reads their adjoints
                r01 += e.relx * pby + gwbx * e.gly;
                r10 += e.rely * pbx + gwby * e.glx; r11 += e.rely * pby + gwby * e.gly; r12 += e.rely * pbz + gwby * e.glz;
                r21 += e.relz * pby + gwbz * e.gly;
            }
            if (kYaw) {
                at0 -= in.r00 * pbx + in.r02 * pbz; at1 -= pby; at2 -= in.r20 * pbx + in.r22 * pbz;
            } else {
                at0 -= in.r00 * pbx + in.r01 * pby + in.r02 * pbz;
                at1 -= in.r10 * pbx + in.r11 * pby + in.r12 * pbz;
                at2 -= in.r20 * pbx + in.r21 * pby + in.r22 * pbz;
            }
        }
        const float packed[16] = {at0, at1, at2, r00, r01, r02, r10, r11, r12, r20, r21, r22, ad0, ad1, ad2, 0.0f};
        const float mine = wave_reduce16_scatter(packed, rl.lane);            // every lane: the wave's sum of parameter (lane & 15)
        const float add = ((rl.lane >> 4) == (i & 3)) ? mine : 0.0f;             // lane (16-lane row r, c) owns parameter c of instances r, 4 + r, ...
        if (kL == kRowLanes) {
#pragma unroll
            for (int s = 0; s < 4; ++s) G[s] += ((i >> 2) == s) ? add : 0.0f;
        } else {                                                                  // 16 registers: branch on the (wave-uniform) register index
            switch (i >> 2) {
#define VSRD_G_CASE(s) case s: G[s] += add; break;
                VSRD_G_CASE(0) VSRD_G_CASE(1) VSRD_G_CASE(2) VSRD_G_CASE(3) VSRD_G_CASE(4) VSRD_G_CASE(5) VSRD_G_CASE(6) VSRD_G_CASE(7)
                VSRD_G_CASE(8) VSRD_G_CASE(9) VSRD_G_CASE(10) VSRD_G_CASE(11) VSRD_G_CASE(12) VSRD_G_CASE(13) VSRD_G_CASE(14) VSRD_G_CASE(15)
#undef VSRD_G_CASE
                default: break;
            }
        }
    }
}

#ifdef VSRD_PHASE_TIMERS
#define VSRD_QUAD_CLOCK_PARAM , PhaseClock& phase_clock
#define VSRD_QUAD_CLOCK_ARG , phase_clock
#else
#define VSRD_QUAD_CLOCK_PARAM
#define VSRD_QUAD_CLOCK_ARG
#endif

// Everything one wave does for one group of four rays: sampling, pass 1, pass 2, silhouette BCE (main.py:653-671; torch clamp /
// binary_cross_entropy backward), label adjoints, reverse sweep, per-instance phase.  One soft-min mode per instantiation (kRunning:
// the running minimum; else the shift known before the instance loop, field.h); returns false -- before any side effect other than
// LDS staging -- when the fixed shift cannot serve some round of the group: the caller then runs the group again with kRunning.
// The three instantiations share no state, so none of it crosses a control-flow merge.
template <int kL, int kRoundsS, bool kYaw, bool kRunning>
__device__ __forceinline__ bool quad_step_body(const float* __restrict__ instances, int N, int NP, const RenderArgs& c, const Shading& sh, int first_ray,
                                               const float* __restrict__ origins, const float* __restrict__ directions,
                                               const float* __restrict__ u_coarse, const float* __restrict__ u_fine, bool sorted_input,
                                               const float* __restrict__ targets, const float* __restrict__ instance_weights, float loss_scale,
                                               float* __restrict__ labels_out, float* stage, float* dcache, float* coefs, float* rays,
                                               float (&G)[kL == kRowLanes ? 4 : 16], float& loss_acc, const RowLanes& rl VSRD_QUAD_CLOCK_PARAM) {
    constexpr int kRounds = 2 * kRoundsS;
    constexpr int kSlots = kL == kRowLanes ? 1 : 2;                          // lane (ray, c) owns the labels of instances c, kL + c
    VSRD_PHASE(7);
    const int S = c.num_samples;
    const int num_points = 2 * S - 1;
    const int my_ray = first_ray + rl.row;
    const bool alive = my_ray < c.num_rays;
    const int ray = alive ? my_ray : (c.num_rays - 1);                    // rows beyond the launch repeat its last ray and contribute nothing
    const long long src = source_row(c, ray);
    float* rowbase = stage + rl.row * quad_row_floats(S, kL);
    float* coef_own = coefs + rl.row * quad_coef_floats(NP);
    const unsigned long long real = NP > N ? ((1ull << N) - 1ull) : ~0ull;            // (NP > N only in instantiations with NP <= 64 and N < NP)
    {
        const long long origin_row = (c.ray_indices && c.rays_per_origin > 0) ? src / c.rays_per_origin : src;
        const float* o = origins + origin_row * c.origin_stride;
        const float* d = directions + src * 3;
        Ray r;
        r.ox = o[0]; r.oy = o[1]; r.oz = o[2]; r.rx = d[0]; r.ry = d[1]; r.rz = d[2];
        quad_ray_setup<kL>(instances, N, NP, r, coef_own, rays + rl.row * kRowRayFloats, rl);
    }
    quad_stage_samples<kL, kRoundsS>(rowbase, c, S, ray, u_coarse, u_fine, sorted_input, rl, c.out_u_coarse, c.out_u_fine, alive);
    VSRD_PHASE(0);
    // ---- pass 1 ----------------------------------------------------------------------------------------------------------------
    float w1[kRoundsS];
    if (!quad_pass_one<kL, kRoundsS, kYaw, kRunning>(instances, NP, sh, rays + rl.row * kRowRayFloats, coef_own, rowbase, S, w1, rl, real)) return false;
    VSRD_PHASE(1);
    if (c.out_coarse_weights != nullptr && alive) {          // (vsrd_render_config::out_*: the step's own state between its passes; wave-uniform pointers)
#pragma unroll
        for (int k = 0; k < kRoundsS; ++k)
            if (k * kL + rl.col < S - 1) c.out_coarse_weights[static_cast<size_t>(my_ray) * (S - 1) + k * kL + rl.col] = w1[k];
    }
    float coarse_total = 0.0f;
#pragma unroll
    for (int k = 0; k < kRoundsS; ++k) coarse_total += seg_sum<kL>(w1[k]);
    // exact misses (VSRD_FLAG_SKIP_EXACT_MISSES): labels exactly 0, adjoint exactly 0
    const bool live = alive && !((c.flags & 2u) && coarse_total == 0.0f);
    const unsigned long long live_lanes = __ballot(live);
    float label[kSlots];
#pragma unroll
    for (int s = 0; s < kSlots; ++s) label[s] = 0.0f;
    unsigned active = 0u;
    int cached_round = -1;
    QuadAdjoint<kRounds> st;
    // rows that do not take part shadow the first live row (same points, same votes in the culling ballots, zero weight)
    const int data_row = live ? rl.row : (live_lanes != 0ull ? (__builtin_ctzll(live_lanes) / kL) : rl.row);
    const float* rayp = rays + data_row * kRowRayFloats;
    const float* merged = stage + data_row * quad_row_floats(S, kL) + quad_merged_offset(S, kL);
    float* trans_mid = rowbase;                                              // (the row's own: a shadow row's transmittances are all 1)
    if (live_lanes != 0ull) {
        quad_importance_merge<kL, kRoundsS>(rowbase, S, w1, rl);
        if (c.out_distances != nullptr && live) {            // the sorted pass-2 distances of the row's own ray (the reverse sweep overwrites them later)
            float* dst = c.out_distances + static_cast<size_t>(my_ray) * (2 * S);
            const float* own = rowbase + quad_merged_offset(S, kL);
            for (int idx = rl.col; idx < 2 * S; idx += kL) dst[idx] = own[idx];
        }
        VSRD_PHASE(2);
        if (!quad_forward_sweep<kL, kRounds, kYaw, kRunning>(st, instances, NP, sh, rayp, coefs + data_row * quad_coef_floats(NP), merged, num_points, live, dcache,
                                                         trans_mid, label, active, cached_round, rl, real)) return false;
        VSRD_PHASE(3);
    }
    if (c.out_distances != nullptr && alive && !live && rl.col == 0) c.out_distances[static_cast<size_t>(my_ray) * (2 * S)] = __builtin_nanf("");   // exact miss: sentinel row
    // ---- loss and label adjoints -------------------------------------------------------------------------------------------------
    unsigned long long lam_any = 0ull;                                       // bit n: some ray has a label adjoint for instance n
    float lam_lane[kSlots];
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        const int n = s * kL + rl.col;
        const bool mine = alive && n < N;
        const float lab = live ? label[s] : 0.0f;
        if (labels_out != nullptr && mine) labels_out[static_cast<size_t>(my_ray) * N + n] = lab;
        const float target = mine ? load_target(c, targets, src, n, N) : 0.0f;
        const float weight_lane = mine ? (instance_weights ? instance_weights[n] : 1.0f) : 0.0f;
        const float p = fminf(fmaxf(lab, 1.0e-6f), 1.0f - 1.0e-6f);
        const float bce = -(target * logf(p) + (1.0f - target) * logf(1.0f - p));
        loss_acc += weight_lane * bce;
        const bool inside_clamp = (lab >= 1.0e-6f) && (lab <= 1.0f - 1.0e-6f);
        lam_lane[s] = (live && inside_clamp) ? weight_lane * loss_scale * (p - target) / fmaxf(p * (1.0f - p), 1.0e-12f) : 0.0f;
        unsigned long long any = __ballot(lam_lane[s] != 0.0f);              // lane (ray r, c): bit kL r + c -> fold the rays
        any |= any >> 32;
        if (kL == kRowLanes) { any |= any >> 16; any &= 0xFFFFull; } else { any &= 0xFFFFFFFFull; }
        lam_any |= any << (s * kL);
    }
    if (active == 0u || lam_any == 0ull) return true;
#pragma unroll
    for (int s = 0; s < kSlots; ++s)
        if (s * kL + rl.col < N) coef_own[kCullCoefs * (s * kL + rl.col) + 3] = lam_lane[s];
    wave_lds_sync();
    // ---- adjoint -------------------------------------------------------------------------------------------------------------------
    const unsigned flow = quad_reverse_sweep<kL, kRounds, kYaw>(st, instances, sh, rayp, merged, num_points, live, active, lam_any, kRunning ? -1 : cached_round,
                                                            coef_own, dcache, trans_mid, rowbase + quad_merged_offset(S, kL), rl);
    VSRD_PHASE(4);
    if (flow != 0u) quad_phase_b<kL, kRounds, kYaw>(st, instances, sh, rayp, flow, coef_own, dcache, trans_mid, rowbase + quad_merged_offset(S, kL), G, rl);
    VSRD_PHASE(5);
    return true;
}

#ifndef VSRD_QUAD_WAVES_PER_EU
#define VSRD_QUAD_WAVES_PER_EU 4
#endif
// kL = lanes per ray: 16 (four rays per wave; N <= 16, S <= 64: BASELINE config 2) or 32 (two rays per wave; N <= 64, S <= 128: BASELINE
// config 5, and config-2-shaped frames with more than 16 instances).
//
// One launch of the step = TWO kernels on the same grid (round 4; one kernel with the three bodies inlined held the registers of all
// of them: 114 spilled VGPRs and 2.8 GB of scratch write-through per launch at config 2):
//   kHot   rotations about y + the soft-min shift known before the instance loop -- what BoxParameters3D produces at ordinary
//          temperatures.  Runs every group when the field allows it (field_bounds: orthonormal, all rotations about y, reach / T small)
//          and records per group whether the fixed shift served it (redo_flags[group]; 1 = some round needs the running minimum --
//          samples extrapolated to 1e6 m next to ordinary ones); does nothing otherwise.
//   !kHot  everything else: all groups when the hot kernel did not run (general rotations with the fixed shift, running minimum as
//          its own fallback; or the running minimum throughout), else only the groups the hot kernel flagged.  The same wave owns the
//          same groups in both kernels (same grid, same stride), so its partial row is written by the hot kernel and added to here:
//          one fixed order of summation, bit-identical repeats.
//
// kFull (hot kernel only): the launch fills its shape -- S = kL * kRoundsS samples per ray (64 / 128: BASELINE configs 2 and 5) and
// more than half of the kL-lane shape's instance slots (N in 9..16 / 33..64) -- so S is a compile-time constant and the instance
// tables are padded to the shape's full count: every LDS offset, every loop bound of the culling pre-pass and the searches, the
// stratification and the point counts fold into the code.  Measured on config 2: 190 -> 222 Mrays/s (S alone: 210) -- the kernel is
// bound by VALU issue (profiles/r04), and a fifth of what it issued was address and bound arithmetic on two launch constants.
template <int kL, int kRoundsS, bool kHot, bool kFull>
__device__ __forceinline__ void silhouette_rows_kernel_body(
    FieldArgs f, const float* __restrict__ instances, RenderArgs c, const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ u_coarse, const float* __restrict__ u_fine,
    const float* __restrict__ targets, const float* __restrict__ instance_weights, float loss_scale,
    float* __restrict__ labels_out, float* __restrict__ partials, float* __restrict__ loss_partials, unsigned char* __restrict__ redo_flags) {
    constexpr int kRays = kWave / kL;
    constexpr int kG = kL == kRowLanes ? 4 : 16;
    // redo_flags[0..16): [0] number of groups the hot kernel flagged, [1] 1 once the hot kernel has run (both zeroed by the host before the
    // hot launch); the per-group flags follow.  The usual case -- the hot kernel ran and flagged nothing -- costs the second kernel two
    // scalar loads per wave (it was 0.18 ms of a 21.7 ms step while every wave walked its groups' flags).
    unsigned* redo_summary = reinterpret_cast<unsigned*>(redo_flags);
    redo_flags += 16;
    if (!kHot && __builtin_amdgcn_readfirstlane(static_cast<int>(redo_summary[1])) != 0 &&
        __builtin_amdgcn_readfirstlane(static_cast<int>(redo_summary[0])) == 0) return;
    apply_device_schedule(f, c);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = wave_in_block();
    const int lane0 = lane_id();
    static_assert(kHot || !kFull, "only the hot kernel is instantiated for a full shape");
    if (kFull) c.num_samples = kL * kRoundsS;                                   // (the host launches kFull for exactly this S)
    // (these kernels take dense launches only -- no ray_indices, no target column map: vsrd_render_silhouette_step sends every other launch
    //  to the one-ray kernels -- so the three fields are compile-time nothing here)
    c.ray_indices = nullptr; c.target_columns = nullptr; c.rays_per_origin = 0;
    const int S = c.num_samples;
    const int N = f.num_instances;
    const int NP = kFull ? (kL == kRowLanes ? kQuadMaxInstances : kPairMaxInstances) : N;      // rows of the instance tables
    float* stage = lds + wave * quad_lds_floats(S, NP, kL);
    float* dcache = stage + kRays * quad_row_floats(S, kL);
    float* coefs = dcache + quad_cache_floats(S, NP, kL);
    float* rays = coefs + kRays * quad_coef_floats(NP);
    const bool sorted_input = (u_fine != nullptr) && (c.flags & 1u);
    Shading sh = c.sh;
    const FieldBounds bounds = field_bounds(instances, N, f.inv_t, false, c.flags);
    sh.cull = bounds.margin; sh.reach = bounds.reach; sh.yaw = bounds.yaw;
#ifdef VSRD_CULL_INNER           // (opt-in, round 6: the culling bounds from the boxes' inscribed ball, field.h: cull_round; +3 %, docs/OPTLOG.md round 6 item 7b)
    sh.inner = bounds.inner;
#else
    sh.inner = 0.0f;
#endif
    sh.yaw_gradients = sh.yaw && (c.flags & 1024u) != 0u;                       // VSRD_FLAG_YAW_GRADIENTS
    sh.mlp_bits = 0u;
    sh.mlp_lds = nullptr;
    const bool hot_runs = sh.reach >= 0.0f && sh.yaw;                          // (wave-uniform, the same in both kernels of the launch)
    if (kHot && !hot_runs) return;
    if (kHot && blockIdx.x == 0 && threadIdx.x == 0) redo_summary[1] = 1u;
#ifdef VSRD_INSTANCE_LDS
    if (kHot && kFull && kL == kRowLanes) {                                     // (the host adds N x 16 floats behind the waves' partitions)
        float* block = lds + waves_per_block() * quad_lds_floats(S, NP, kL);
        for (int idx = static_cast<int>(threadIdx.x); idx < N * kInstanceStride; idx += static_cast<int>(blockDim.x)) block[idx] = instances[idx];
        __syncthreads();
        instances = block;
    }
#endif
    float loss_acc = 0.0f;
    float G[kG];
#pragma unroll
    for (int s = 0; s < kG; ++s) G[s] = 0.0f;
    const int num_waves = static_cast<int>(gridDim.x) * waves_per_block();
    const int wave_global = static_cast<int>(blockIdx.x) * waves_per_block() + wave;
    const int num_groups = (c.num_rays + kRays - 1) / kRays;
    bool touched = false;                                                       // !kHot behind the hot kernel: did this wave redo a group?
    VSRD_PHASE_CLOCK();
    for (int group = wave_global; group < num_groups; group += num_waves) {
        const int first_ray = group * kRays;
        if (!kHot && hot_runs) {
            if (__builtin_amdgcn_readfirstlane(static_cast<int>(redo_flags[group])) == 0) continue;
            touched = true;
        }
        const RowLanes rl = row_lanes<kL>(opaque_lane_id());
        wave_lds_sync();
        if (kHot) {
            const bool done = quad_step_body<kL, kRoundsS, true, false>(instances, N, NP, c, sh, first_ray, origins, directions, u_coarse, u_fine, sorted_input, targets,
                                                                        instance_weights, loss_scale, labels_out, stage, dcache, coefs, rays, G, loss_acc, rl VSRD_QUAD_CLOCK_ARG);
            if (lane0 == 0) {
                redo_flags[group] = done ? 0 : 1;
                if (!done) atomicAdd(redo_summary, 1u);
            }
        } else {
            bool done = false;
            if (sh.reach >= 0.0f && !hot_runs) {
                done = quad_step_body<kL, kRoundsS, false, false>(instances, N, NP, c, sh, first_ray, origins, directions, u_coarse, u_fine, sorted_input, targets,
                                                                  instance_weights, loss_scale, labels_out, stage, dcache, coefs, rays, G, loss_acc, rl VSRD_QUAD_CLOCK_ARG);
                if (!done) wave_lds_sync();
            }
            if (!done) quad_step_body<kL, kRoundsS, false, true>(instances, N, NP, c, sh, first_ray, origins, directions, u_coarse, u_fine, sorted_input, targets,
                                                                 instance_weights, loss_scale, labels_out, stage, dcache, coefs, rays, G, loss_acc, rl VSRD_QUAD_CLOCK_ARG);
        }
    }
    VSRD_PHASE(7);
    VSRD_PHASE_FLUSH(lane0);
    if (!kHot && hot_runs && !touched) return;                                  // (the usual case: nothing was left over)
    // the wave's row of the partial-gradient table (summed over the waves by reduce_partials_kernel): lane (16-lane row r, c) of register s
    // holds parameter c of instance 4 s + r
    const bool add = !kHot && hot_runs;
    float* out = partials + static_cast<size_t>(wave_global) * (N * kGradStride);
#pragma unroll
    for (int s = 0; s < kG; ++s)
        if (s * kWave + lane0 < N * kGradStride) out[s * kWave + lane0] = add ? (out[s * kWave + lane0] + G[s]) : G[s];
    const float loss_total = wave_sum(loss_acc);
    if (lane0 == 0) loss_partials[wave_global] = add ? (loss_partials[wave_global] + loss_total * loss_scale) : (loss_total * loss_scale);
}

template <int kRoundsS, bool kHot, bool kFull>
__global__ __launch_bounds__(kBlockThreads, VSRD_QUAD_WAVES_PER_EU) void render_silhouette_quad_kernel(
    FieldArgs f, const float* __restrict__ instances, RenderArgs c, const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ u_coarse, const float* __restrict__ u_fine,
    const float* __restrict__ targets, const float* __restrict__ instance_weights, float loss_scale,
    float* __restrict__ labels_out, float* __restrict__ partials, float* __restrict__ loss_partials, unsigned char* __restrict__ redo_flags) {
    silhouette_rows_kernel_body<kRowLanes, kRoundsS, kHot, kFull>(f, instances, c, origins, directions, u_coarse, u_fine, targets, instance_weights, loss_scale, labels_out,
                                                           partials, loss_partials, redo_flags);
}

// ---- the two-pass forward alone (vsrd_render_hierarchical_forward: scripts/main.py:511-523 as one launch) in the same mappings -----------
// Labels [R,N] and, when asked for, the sorted pass-2 distances [R,2S] the backward works from (rays skipped as exact misses get the
// NaN sentinel render_backward_kernel looks for).  Launches that also want the per-sample gradients / weights or the uniforms back keep
// the one-ray kernel (render_hierarchical_kernel).
template <int kL, int kRoundsS, bool kYaw, bool kRunning>
__device__ __forceinline__ bool rows_forward_body(const float* __restrict__ instances, int N, int NP, const RenderArgs& c, const Shading& sh, int first_ray,
                                                  const float* __restrict__ origins, const float* __restrict__ directions,
                                                  const float* __restrict__ u_coarse, const float* __restrict__ u_fine, bool sorted_input,
                                                  float* __restrict__ labels_out, float* __restrict__ distances_out, float* __restrict__ coarse_weights_out,
                                                  float* __restrict__ u_coarse_out, float* __restrict__ u_fine_out,
                                                  float* stage, float* dcache, float* coefs, float* rays, const RowLanes& rl) {
    constexpr int kRounds = 2 * kRoundsS;
    constexpr int kSlots = kL == kRowLanes ? 1 : 2;
    const int S = c.num_samples;
    const int num_points = 2 * S - 1;
    const int my_ray = first_ray + rl.row;
    const bool alive = my_ray < c.num_rays;
    const int ray = alive ? my_ray : (c.num_rays - 1);
    float* rowbase = stage + rl.row * quad_row_floats(S, kL);
    float* coef_own = coefs + rl.row * quad_coef_floats(NP);
    const unsigned long long real = NP > N ? ((1ull << N) - 1ull) : ~0ull;            // (NP > N: the full-shape instantiation's padded instance tables, quad_step_body)
    {
        const float* o = origins + static_cast<size_t>(ray) * c.origin_stride;
        const float* d = directions + static_cast<size_t>(ray) * 3;
        Ray r;
        r.ox = o[0]; r.oy = o[1]; r.oz = o[2]; r.rx = d[0]; r.ry = d[1]; r.rz = d[2];
        quad_ray_setup<kL>(instances, N, NP, r, coef_own, rays + rl.row * kRowRayFloats, rl);
    }
    quad_stage_samples<kL, kRoundsS>(rowbase, c, S, ray, u_coarse, u_fine, sorted_input, rl, u_coarse_out, u_fine_out, alive);
    float w1[kRoundsS];
    if (!quad_pass_one<kL, kRoundsS, kYaw, kRunning>(instances, NP, sh, rays + rl.row * kRowRayFloats, coef_own, rowbase, S, w1, rl, real)) return false;
    if (coarse_weights_out != nullptr && alive) {            // pass 1's compositing weights: what pass 1 of main.py:511-523 hands to pass 2
#pragma unroll
        for (int k = 0; k < kRoundsS; ++k)
            if (k * kL + rl.col < S - 1) coarse_weights_out[static_cast<size_t>(my_ray) * (S - 1) + k * kL + rl.col] = w1[k];
    }
    float coarse_total = 0.0f;
#pragma unroll
    for (int k = 0; k < kRoundsS; ++k) coarse_total += seg_sum<kL>(w1[k]);
    const bool live = alive && !((c.flags & 2u) && coarse_total == 0.0f);
    const unsigned long long live_lanes = __ballot(live);
    float label[kSlots];
#pragma unroll
    for (int s = 0; s < kSlots; ++s) label[s] = 0.0f;
    if (live_lanes != 0ull) {
        unsigned active = 0u;
        int cached_round = -1;
        QuadAdjoint<kRounds> st;
        const int data_row = live ? rl.row : (__builtin_ctzll(live_lanes) / kL);
        quad_importance_merge<kL, kRoundsS>(rowbase, S, w1, rl);
        if (!quad_forward_sweep<kL, kRounds, kYaw, kRunning>(st, instances, NP, sh, rays + data_row * kRowRayFloats, coefs + data_row * quad_coef_floats(NP),
                                                              stage + data_row * quad_row_floats(S, kL) + quad_merged_offset(S, kL), num_points, live, dcache,
                                                              rowbase, label, active, cached_round, rl, real)) return false;
    }
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        const int n = s * kL + rl.col;
        if (alive && n < N) labels_out[static_cast<size_t>(my_ray) * N + n] = live ? label[s] : 0.0f;
    }
    if (distances_out != nullptr && alive) {
        float* dst = distances_out + static_cast<size_t>(my_ray) * (2 * S);
        if (live) {
            const float* merged = rowbase + quad_merged_offset(S, kL);
            for (int idx = rl.col; idx < 2 * S; idx += kL) dst[idx] = merged[idx];
        } else if (rl.col == 0) {
            dst[0] = __builtin_nanf("");                                     // sentinel row: the backward skips it (exact miss: exact zero adjoint)
        }
    }
    return true;
}

// Round 6: the two-launch path's forward is TWO kernels on one grid as well, and the hot one is instantiated for a full
// shape like the fused step's (kFull: S and the instance count compile-time constants).  The forward has no scratch to keep flags in (the C
// ABI gives vsrd_render_hierarchical_forward none), so a group the hot body cannot serve is marked IN ITS OUTPUT: NaN in the first label of
// its first ray, which the second kernel looks for (one scalar load per group) and overwrites with the group's real labels.
template <int kL, int kRoundsS, bool kHot, bool kFull>
__device__ __forceinline__ void hierarchical_rows_kernel_body(
    FieldArgs f, const float* __restrict__ instances, RenderArgs c, const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ u_coarse, const float* __restrict__ u_fine, float* __restrict__ labels_out, float* __restrict__ distances_out,
    float* __restrict__ coarse_weights_out, float* __restrict__ u_coarse_out, float* __restrict__ u_fine_out) {
    constexpr int kRays = kWave / kL;
    static_assert(kHot || !kFull, "only the hot kernel is instantiated for a full shape");
    apply_device_schedule(f, c);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = wave_in_block();
    if (kFull) c.num_samples = kL * kRoundsS;                                   // (the host launches kFull for exactly this S)
    const int S = c.num_samples;
    const int N = f.num_instances;
    const int NP = kFull ? (kL == kRowLanes ? kQuadMaxInstances : kPairMaxInstances) : N;      // rows of the instance tables
    float* stage = lds + wave * quad_lds_floats(S, NP, kL);
    float* dcache = stage + kRays * quad_row_floats(S, kL);
    float* coefs = dcache + quad_cache_floats(S, NP, kL);
    float* rays = coefs + kRays * quad_coef_floats(NP);
    const bool sorted_input = (u_fine != nullptr) && (c.flags & 1u);
    Shading sh = c.sh;
    const FieldBounds bounds = field_bounds(instances, N, f.inv_t, false, c.flags);
    sh.cull = bounds.margin; sh.reach = bounds.reach; sh.yaw = bounds.yaw;
#ifdef VSRD_CULL_INNER           // (opt-in, round 6: the culling bounds from the boxes' inscribed ball, field.h: cull_round; +3 %, docs/OPTLOG.md round 6 item 7b)
    sh.inner = bounds.inner;
#else
    sh.inner = 0.0f;
#endif
    sh.mlp_bits = 0u;
    sh.mlp_lds = nullptr;
    const bool hot_runs = sh.reach >= 0.0f && sh.yaw;                          // (wave-uniform, the same in both kernels of the launch)
    if (kHot && !hot_runs) return;
    const int num_waves = static_cast<int>(gridDim.x) * waves_per_block();
    const int wave_global = static_cast<int>(blockIdx.x) * waves_per_block() + wave;
    const int num_groups = (c.num_rays + kRays - 1) / kRays;
    if (!kHot && hot_runs) {            // the usual case -- nothing marked -- costs this wave ONE vector load: lane k looks at the mark of the wave's k-th group
        bool any = false;
        for (int base = wave_global; base < num_groups; base += kWave * num_waves) {
            const long long group = static_cast<long long>(base) + static_cast<long long>(lane_id()) * num_waves;
            const float mark = group < num_groups ? labels_out[static_cast<size_t>(group) * kRays * N] : 0.0f;
            any = any || __ballot(mark != mark) != 0ull;
        }
        if (!any) return;
    }
    for (int group = wave_global; group < num_groups; group += num_waves) {
        const int first_ray = group * kRays;
        if (!kHot && hot_runs) {                                                // behind the hot kernel: only the groups it marked
            const float mark = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, labels_out[static_cast<size_t>(first_ray) * N])));
            if (mark == mark) continue;
        }
        const RowLanes rl = row_lanes<kL>(opaque_lane_id());
        wave_lds_sync();
        if (kHot) {
            const bool done = rows_forward_body<kL, kRoundsS, true, false>(instances, N, NP, c, sh, first_ray, origins, directions, u_coarse, u_fine, sorted_input,
                                                                           labels_out, distances_out, coarse_weights_out, u_coarse_out, u_fine_out, stage, dcache, coefs, rays, rl);
            if (!done && lane_id() == 0) labels_out[static_cast<size_t>(first_ray) * N] = __builtin_nanf("");
        } else {
            bool done = false;
            if (sh.reach >= 0.0f && !hot_runs) {
                done = rows_forward_body<kL, kRoundsS, false, false>(instances, N, NP, c, sh, first_ray, origins, directions, u_coarse, u_fine, sorted_input,
                                                                     labels_out, distances_out, coarse_weights_out, u_coarse_out, u_fine_out, stage, dcache, coefs, rays, rl);
                if (!done) wave_lds_sync();
            }
            if (!done) rows_forward_body<kL, kRoundsS, false, true>(instances, N, NP, c, sh, first_ray, origins, directions, u_coarse, u_fine, sorted_input,
                                                                    labels_out, distances_out, coarse_weights_out, u_coarse_out, u_fine_out, stage, dcache, coefs, rays, rl);
        }
    }
}

template <bool kHot, bool kFull>
__global__ __launch_bounds__(kBlockThreads, 4) void render_hierarchical_quad_kernel(
    FieldArgs f, const float* __restrict__ instances, RenderArgs c, const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ u_coarse, const float* __restrict__ u_fine, float* __restrict__ labels_out, float* __restrict__ distances_out,
    float* __restrict__ coarse_weights_out, float* __restrict__ u_coarse_out, float* __restrict__ u_fine_out) {
    hierarchical_rows_kernel_body<kRowLanes, 4, kHot, kFull>(f, instances, c, origins, directions, u_coarse, u_fine, labels_out, distances_out, coarse_weights_out, u_coarse_out, u_fine_out);
}
template <bool kHot, bool kFull>
__global__ __launch_bounds__(kBlockThreads, 4) void render_hierarchical_pair_kernel(
    FieldArgs f, const float* __restrict__ instances, RenderArgs c, const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ u_coarse, const float* __restrict__ u_fine, float* __restrict__ labels_out, float* __restrict__ distances_out,
    float* __restrict__ coarse_weights_out, float* __restrict__ u_coarse_out, float* __restrict__ u_fine_out) {
    hierarchical_rows_kernel_body<32, 4, kHot, kFull>(f, instances, c, origins, directions, u_coarse, u_fine, labels_out, distances_out, coarse_weights_out, u_coarse_out, u_fine_out);
}

// ---- the adjoint at saved distances (vsrd_render_backward, box-only fields, label adjoints only) in the same mappings ------------------
// renderers.py:212-270 backwards for the rays of one wave: sorted distances [R,D] from the forward, grad_labels [R,N]; what
// hierarchical_volumetric_rendering(...).backward() of the reference reaches for a box-only field (scripts/main.py:511-523, 629-671).
// Launches that also carry adjoints of the per-sample gradients / weights keep the one-ray kernel (render_backward_kernel).
template <int kL, int kRounds, bool kYaw, bool kRunning>
__device__ __forceinline__ bool rows_backward_body(const float* __restrict__ instances, int N, int NP, const RenderArgs& c, const Shading& sh, int first_ray,
                                                   const float* __restrict__ origins, const float* __restrict__ directions,
                                                   const float* __restrict__ distances, int num_distances, const float* __restrict__ grad_labels,
                                                   float* stage, float* dcache, float* coefs, float* rays, float (&G)[kL == kRowLanes ? 4 : 16], const RowLanes& rl) {
    constexpr int kSlots = kL == kRowLanes ? 1 : 2;
    const int half = (num_distances + 1) / 2;                                // the row layout of the step kernels for S = ceil(D / 2)
    const int num_points = num_distances - 1;
    const int my_ray = first_ray + rl.row;
    const bool alive = my_ray < c.num_rays;
    const int ray = alive ? my_ray : (c.num_rays - 1);
    float* rowbase = stage + rl.row * quad_row_floats(half, kL);
    float* coef_own = coefs + rl.row * quad_coef_floats(NP);
    const unsigned long long real = NP > N ? ((1ull << N) - 1ull) : ~0ull;            // (NP > N: the full-shape instantiation's padded instance tables)
    {
        const float* o = origins + static_cast<size_t>(ray) * c.origin_stride;
        const float* d = directions + static_cast<size_t>(ray) * 3;
        Ray r;
        r.ox = o[0]; r.oy = o[1]; r.oz = o[2]; r.rx = d[0]; r.ry = d[1]; r.rz = d[2];
        quad_ray_setup<kL>(instances, N, NP, r, coef_own, rays + rl.row * kRowRayFloats, rl);      // (clears the label adjoints, syncs)
    }
    const float* src = distances + static_cast<size_t>(ray) * num_distances;
    float* own_merged = rowbase + quad_merged_offset(half, kL);
    for (int idx = rl.col; idx < num_distances; idx += kL) own_merged[idx] = src[idx];
    float lam_lane[kSlots];
    float biggest = 0.0f;
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        const int n = s * kL + rl.col;
        lam_lane[s] = (alive && n < N) ? grad_labels[static_cast<size_t>(my_ray) * N + n] : 0.0f;
        biggest = fmaxf(biggest, fabsf(lam_lane[s]));
    }
    const float first_distance = src[0];
    // rays the forward skipped (NaN sentinel) and rays without a label adjoint: nothing flows back
    const bool live = alive && first_distance == first_distance && seg_max<kL>(biggest) > 0.0f;
    const unsigned long long live_lanes = __ballot(live);
    if (live_lanes == 0ull) return true;
    unsigned long long lam_any = 0ull;
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        if (!live) lam_lane[s] = 0.0f;
        unsigned long long any = __ballot(lam_lane[s] != 0.0f);
        any |= any >> 32;
        if (kL == kRowLanes) { any |= any >> 16; any &= 0xFFFFull; } else { any &= 0xFFFFFFFFull; }
        lam_any |= any << (s * kL);
        if (s * kL + rl.col < N) coef_own[kCullCoefs * (s * kL + rl.col) + 3] = lam_lane[s];
    }
    wave_lds_sync();
    const int data_row = live ? rl.row : (__builtin_ctzll(live_lanes) / kL);
    const float* rayp = rays + data_row * kRowRayFloats;
    const float* merged = stage + data_row * quad_row_floats(half, kL) + quad_merged_offset(half, kL);
    float label[kSlots];
    unsigned active = 0u;
    int cached_round = -1;
    QuadAdjoint<kRounds> st;
    if (!quad_forward_sweep<kL, kRounds, kYaw, kRunning>(st, instances, NP, sh, rayp, coefs + data_row * quad_coef_floats(NP), merged, num_points, live, dcache,
                                                          rowbase, label, active, cached_round, rl, real)) return false;
    if (active == 0u) return true;
    const unsigned flow = quad_reverse_sweep<kL, kRounds, kYaw>(st, instances, sh, rayp, merged, num_points, live, active, lam_any, kRunning ? -1 : cached_round,
                                                                coef_own, dcache, rowbase, own_merged, rl);
    if (flow != 0u) quad_phase_b<kL, kRounds, kYaw>(st, instances, sh, rayp, flow, coef_own, dcache, rowbase, own_merged, G, rl);
    return true;
}

// Round 6: the adjoint launch is TWO kernels on one grid, like the fused step (silhouette_rows_kernel_body above).
// With its three bodies inlined into one loop render_backward_quad_kernel held 128 registers with 135 of them spilled (160 bytes of scratch per
// lane: the registers of the bodies it was not running) -- and this is the backward every `loss.backward()` of the recommended one-launch
// render_hierarchical path takes.
//   kHot   rotations about y + fixed soft-min shift (what BoxParameters3D decodes to); records per group whether that body served it
//   !kHot  everything else: all groups when the hot kernel did not run, else only the groups it flagged, adding to the same partial rows
template <int kL, bool kHot, bool kFull>
__device__ __forceinline__ void backward_rows_kernel_body(
    FieldArgs f, const float* __restrict__ instances, RenderArgs c, const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ distances, int num_distances, const float* __restrict__ grad_labels, float* __restrict__ partials,
    unsigned char* __restrict__ redo_flags) {
    constexpr int kRays = kWave / kL;
    constexpr int kG = kL == kRowLanes ? 4 : 16;
    constexpr int kRounds = 8;                                               // up to 8 kL points
    static_assert(kHot || !kFull, "only the hot kernel is instantiated for a full shape");
    if (kFull) num_distances = 8 * kL;                                       // (the host launches kFull for exactly 2 S = 8 kL distances per ray: configs 2 and 5)
    unsigned* redo_summary = reinterpret_cast<unsigned*>(redo_flags);        // [0] groups flagged, [1] the hot kernel ran (zeroed by the host)
    redo_flags += 16;
    if (!kHot && __builtin_amdgcn_readfirstlane(static_cast<int>(redo_summary[1])) != 0 &&
        __builtin_amdgcn_readfirstlane(static_cast<int>(redo_summary[0])) == 0) return;
    apply_device_schedule(f, c);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wave = wave_in_block();
    const int lane0 = lane_id();
    const int half = (num_distances + 1) / 2;
    const int N = f.num_instances;
    const int NP = kFull ? (kL == kRowLanes ? kQuadMaxInstances : kPairMaxInstances) : N;      // rows of the instance tables
    float* stage = lds + wave * quad_lds_floats(half, NP, kL);
    float* dcache = stage + kRays * quad_row_floats(half, kL);
    float* coefs = dcache + quad_cache_floats(half, NP, kL);
    float* rays = coefs + kRays * quad_coef_floats(NP);
    Shading sh = c.sh;
    const FieldBounds bounds = field_bounds(instances, N, f.inv_t, false, c.flags);
    sh.cull = bounds.margin; sh.reach = bounds.reach; sh.yaw = bounds.yaw;
#ifdef VSRD_CULL_INNER           // (opt-in, round 6: the culling bounds from the boxes' inscribed ball, field.h: cull_round; +3 %, docs/OPTLOG.md round 6 item 7b)
    sh.inner = bounds.inner;
#else
    sh.inner = 0.0f;
#endif
    sh.yaw_gradients = sh.yaw && (c.flags & 1024u) != 0u;                       // VSRD_FLAG_YAW_GRADIENTS
    sh.mlp_bits = 0u;
    sh.mlp_lds = nullptr;
    const bool hot_runs = sh.reach >= 0.0f && sh.yaw;                          // (wave-uniform, the same in both kernels of the launch)
    if (kHot && !hot_runs) return;
    if (kHot && blockIdx.x == 0 && threadIdx.x == 0) redo_summary[1] = 1u;
    float G[kG];
#pragma unroll
    for (int s = 0; s < kG; ++s) G[s] = 0.0f;
    const int num_waves = static_cast<int>(gridDim.x) * waves_per_block();
    const int wave_global = static_cast<int>(blockIdx.x) * waves_per_block() + wave;
    const int num_groups = (c.num_rays + kRays - 1) / kRays;
    bool touched = false;
    for (int group = wave_global; group < num_groups; group += num_waves) {
        const int first_ray = group * kRays;
        if (!kHot && hot_runs) {
            if (__builtin_amdgcn_readfirstlane(static_cast<int>(redo_flags[group])) == 0) continue;
            touched = true;
        }
        const RowLanes rl = row_lanes<kL>(opaque_lane_id());
        wave_lds_sync();
        if (kHot) {
            const bool done = rows_backward_body<kL, kRounds, true, false>(instances, N, NP, c, sh, first_ray, origins, directions, distances, num_distances, grad_labels,
                                                                           stage, dcache, coefs, rays, G, rl);
            if (lane0 == 0) {
                redo_flags[group] = done ? 0 : 1;
                if (!done) atomicAdd(redo_summary, 1u);
            }
        } else {
            bool done = false;
            if (sh.reach >= 0.0f && !hot_runs) {
                done = rows_backward_body<kL, kRounds, false, false>(instances, N, NP, c, sh, first_ray, origins, directions, distances, num_distances, grad_labels,
                                                                     stage, dcache, coefs, rays, G, rl);
                if (!done) wave_lds_sync();
            }
            if (!done) rows_backward_body<kL, kRounds, false, true>(instances, N, NP, c, sh, first_ray, origins, directions, distances, num_distances, grad_labels,
                                                                    stage, dcache, coefs, rays, G, rl);
        }
    }
    if (!kHot && hot_runs && !touched) return;                                  // (the usual case: nothing was left over)
    const bool add = !kHot && hot_runs;
    float* out = partials + static_cast<size_t>(wave_global) * (N * kGradStride);
#pragma unroll
    for (int s = 0; s < kG; ++s)
        if (s * kWave + lane0 < N * kGradStride) out[s * kWave + lane0] = add ? (out[s * kWave + lane0] + G[s]) : G[s];
}

template <bool kHot, bool kFull>
__global__ __launch_bounds__(kBlockThreads, 4) void render_backward_quad_kernel(
    FieldArgs f, const float* __restrict__ instances, RenderArgs c, const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ distances, int num_distances, const float* __restrict__ grad_labels, float* __restrict__ partials, unsigned char* __restrict__ redo_flags) {
    backward_rows_kernel_body<kRowLanes, kHot, kFull>(f, instances, c, origins, directions, distances, num_distances, grad_labels, partials, redo_flags);
}
template <bool kHot, bool kFull>
__global__ __launch_bounds__(kBlockThreads, 3) void render_backward_pair_kernel(
    FieldArgs f, const float* __restrict__ instances, RenderArgs c, const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ distances, int num_distances, const float* __restrict__ grad_labels, float* __restrict__ partials, unsigned char* __restrict__ redo_flags) {
    backward_rows_kernel_body<32, kHot, kFull>(f, instances, c, origins, directions, distances, num_distances, grad_labels, partials, redo_flags);
}

// Two rays per wave, 32 lanes each (kRoundsS = 2: S <= 64; 4: S <= 128).
#ifndef VSRD_PAIR_WAVES_PER_EU
#define VSRD_PAIR_WAVES_PER_EU 3
#endif
template <int kRoundsS, bool kHot, bool kFull>
__global__ __launch_bounds__(kBlockThreads, VSRD_PAIR_WAVES_PER_EU) void render_silhouette_pair_kernel(
    FieldArgs f, const float* __restrict__ instances, RenderArgs c, const float* __restrict__ origins, const float* __restrict__ directions,
    const float* __restrict__ u_coarse, const float* __restrict__ u_fine,
    const float* __restrict__ targets, const float* __restrict__ instance_weights, float loss_scale,
    float* __restrict__ labels_out, float* __restrict__ partials, float* __restrict__ loss_partials, unsigned char* __restrict__ redo_flags) {
    silhouette_rows_kernel_body<32, kRoundsS, kHot, kFull>(f, instances, c, origins, directions, u_coarse, u_fine, targets, instance_weights, loss_scale, labels_out,
                                                    partials, loss_partials, redo_flags);
}

}  // namespace vsrd
