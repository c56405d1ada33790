// Wave64 primitives for gfx950: one wavefront renders one ray, one lane owns one sample.
// Everything here is wave-local: no workgroup barrier is ever needed on the render path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace vsrd {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

// Wave index inside the workgroup as a *scalar* (SGPR) value so that everything indexed by it
// (LDS partitions, ray ids, per-instance parameter loads) stays on the scalar unit.
__device__ __forceinline__ int wave_in_block() {
    return __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
}

// ---- frame batches (include/vsrd_hip.h, ABI 8: vsrd_frame_batch) ---------------------------------------------------------------
// A launch over B frames is B copies of the one-frame grid stacked along blockIdx.y (kernels whose own grid is 2-D or 3-D: along
// blockIdx.z), and copy f works on frame f's buffers: EVERY device pointer of the launch addresses frame 0's buffer, and frame f's lies
// `frame_stride` bytes x f behind it (the caller lays the frames' buffers out in one arena with one stride).  Nothing else differs
// between the copies -- same blockIdx.x / gridDim.x, same by-value arguments -- so frame f's results are bit for bit those of a launch
// on frame f alone.  `of_frame` keeps the argument's provenance (a byte offset on the same object, nullptr stays nullptr): the
// compiler still selects scalar loads for the wave-uniform parameter reads.
__device__ __forceinline__ long long frame_shift(long long frame_stride, unsigned frame) { return frame_stride * static_cast<long long>(frame); }
template <typename T>
__device__ __forceinline__ T* of_frame(T* p, long long shift) {
    return p ? reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_cv<T>::type*>(p)) + shift) : p;
}
// for arguments declared `T* __restrict__ p`: VSRD_OF_FRAME(p, shift) re-points the argument itself
#define VSRD_OF_FRAME(p, shift) p = of_frame(p, shift)
// ... and for arguments that are never null (no select)
template <typename T>
__device__ __forceinline__ T* of_frame_nonnull(T* p, long long shift) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_cv<T>::type*>(p)) + shift);
}
#define VSRD_OF_FRAME_NONNULL(p, shift) p = of_frame_nonnull(p, shift)

__device__ __forceinline__ float uniform(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ float read_lane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// LDS written by some lanes of this wave and read by others: DS operations of one wave execute in
// order, so only the compiler has to be stopped from reordering; the workgroup-scope fence also
// drains lgkmcnt, which is cheap and keeps this robust against scheduling changes.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Workgroup barrier for data handed over through LDS only: waits for this wave's LDS operations, not for its global loads and
// stores (__syncthreads() drains vmcnt too -- a prefetch requested before the barrier would be waited for AT the barrier, and a
// global store by one wave would hold everybody for its ~2 us round trip).
__device__ __forceinline__ void block_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- DPP building blocks (gfx9 family: row_shr, row_bcast15/31 are available on gfx950) ----------
template <int kCtrl, int kRowMask = 0xf, int kBankMask = 0xf>
__device__ __forceinline__ float dpp_move(float identity, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v),
                                                                 kCtrl, kRowMask, kBankMask, false));
}
// In-place scan steps "v = v op v[lane - n]" inside a row of 16 as ONE DPP instruction: a lane whose source lies outside its row is
// disabled by the instruction itself (no bound_ctrl) and keeps its value, so no identity register is needed -- the compiler's form of
// `v *= dpp_move(1.0f, v)` is v_mov 1.0 + v_mov_dpp + v_mul (and for fmaxf two canonicalising v_max on top).  The wait states a VALU
// write needs before a DPP read are in the string (-DVSRD_NO_DPP_INPLACE: the compiler's form).
#ifndef VSRD_NO_DPP_INPLACE
#define VSRD_DPP_STEP(NAME, OP, CTRL)                                                                              \
    __device__ __forceinline__ float NAME(float v) {                                                               \
        asm volatile("s_nop 1\n\t" OP " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(v));                \
        return v;                                                                                                  \
    }
VSRD_DPP_STEP(mul_shr1, "v_mul_f32_dpp", "row_shr:1") VSRD_DPP_STEP(mul_shr2, "v_mul_f32_dpp", "row_shr:2")
VSRD_DPP_STEP(mul_shr4, "v_mul_f32_dpp", "row_shr:4") VSRD_DPP_STEP(mul_shr8, "v_mul_f32_dpp", "row_shr:8")
VSRD_DPP_STEP(max_shr1, "v_max_f32_dpp", "row_shr:1") VSRD_DPP_STEP(max_shr2, "v_max_f32_dpp", "row_shr:2")
VSRD_DPP_STEP(max_shr4, "v_max_f32_dpp", "row_shr:4") VSRD_DPP_STEP(max_shr8, "v_max_f32_dpp", "row_shr:8")
#undef VSRD_DPP_STEP
// min / max of two values that are known not to be signalling NaNs: one instruction (fminf / fmaxf canonicalise both operands first
// whenever the compiler cannot prove them canonical, e.g. a value carried around a loop)
__device__ __forceinline__ float min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#else
__device__ __forceinline__ float min_raw(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float max_raw(float a, float b) { return fmaxf(a, b); }
#endif

constexpr int kDppQuadXor1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141;
constexpr int kDppRowMirror = 0x140;
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

// Sum over the 64 lanes; the result is returned as a wave-uniform value.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<kDppQuadXor1>(0.0f, v);
    v += dpp_move<kDppQuadXor2>(0.0f, v);
    v += dpp_move<kDppRowHalfMirror>(0.0f, v);
    v += dpp_move<kDppRowMirror>(0.0f, v);                 // every lane: sum of its row of 16
    v += dpp_move<kDppRowBcast15, 0xa>(0.0f, v);           // rows 1,3 += row 0,2
    v += dpp_move<kDppRowBcast31, 0xc>(0.0f, v);           // rows 2,3 += rows 0+1
    return read_lane(v, 63);
}

__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<kDppQuadXor1>(v, v));
    v = fmaxf(v, dpp_move<kDppQuadXor2>(v, v));
    v = fmaxf(v, dpp_move<kDppRowHalfMirror>(v, v));
    v = fmaxf(v, dpp_move<kDppRowMirror>(v, v));
    v = fmaxf(v, dpp_move<kDppRowBcast15, 0xa>(v, v));
    v = fmaxf(v, dpp_move<kDppRowBcast31, 0xc>(v, v));
    return read_lane(v, 63);
}

// Inclusive prefix sum / product over the lanes (lane 0 first).
__device__ __forceinline__ float wave_inclusive_sum(float v) {
    v += dpp_move<kDppRowShr1>(0.0f, v);
    v += dpp_move<kDppRowShr2>(0.0f, v);
    v += dpp_move<kDppRowShr4>(0.0f, v);
    v += dpp_move<kDppRowShr8>(0.0f, v);                   // scan inside each row of 16
    v += dpp_move<kDppRowBcast15, 0xa>(0.0f, v);
    v += dpp_move<kDppRowBcast31, 0xc>(0.0f, v);
    return v;
}

__device__ __forceinline__ float wave_inclusive_product(float v) {
    v *= dpp_move<kDppRowShr1>(1.0f, v);
    v *= dpp_move<kDppRowShr2>(1.0f, v);
    v *= dpp_move<kDppRowShr4>(1.0f, v);
    v *= dpp_move<kDppRowShr8>(1.0f, v);
    v *= dpp_move<kDppRowBcast15, 0xa>(1.0f, v);
    v *= dpp_move<kDppRowBcast31, 0xc>(1.0f, v);
    return v;
}

// lane ^ 16 / lane ^ 32 exchange sums with the gfx950 row-swap instructions (v_permlane16_swap / v_permlane32_swap):
// swapping a register with itself leaves {even rows duplicated, odd rows duplicated}, whose sum is the xor butterfly.
// (inline asm: with hipcc 7.2 the second result of __builtin_amdgcn_permlane{16,32}_swap is assigned the first
//  result's register; the two wait states LLVM's hazard rule "VALU write -> v_permlane read" needs are in the string)
__device__ __forceinline__ float add_xor16(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float add_xor32(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// Reduce 16 values per lane over the whole wave with a reduce-scatter butterfly: after the call EVERY lane l holds
// the sum over the 64 lanes of v[l & 15].  ~50 VALU ops instead of 16 x (6 DPP + readlane): each stage pairs the
// lanes of a row through a DPP permutation (mirror, half-mirror, quad xor 2, quad xor 1) and halves the number of
// live values, so lane i of a row ends up owning value i; two row swaps then fold the four rows.
__device__ __forceinline__ float wave_reduce16_scatter(const float (&v)[16], int lane) {
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
    float a[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const float keep = b3 ? v[8 + m] : v[m];
        const float send = b3 ? v[m] : v[8 + m];
        a[m] = keep + dpp_move<kDppRowMirror>(0.0f, send);
    }
    float b[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float keep = b2 ? a[4 + m] : a[m];
        const float send = b2 ? a[m] : a[4 + m];
        b[m] = keep + dpp_move<kDppRowHalfMirror>(0.0f, send);
    }
    float c[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const float keep = b1 ? b[2 + m] : b[m];
        const float send = b1 ? b[m] : b[2 + m];
        c[m] = keep + dpp_move<kDppQuadXor2>(0.0f, send);
    }
    const float keep = b0 ? c[1] : c[0];
    const float send = b0 ? c[0] : c[1];
    const float r = keep + dpp_move<kDppQuadXor1>(0.0f, send);   // lane i of each row: row-sum of value i
    return add_xor32(add_xor16(r));
}

// Value of lane (l-1), `first` for lane 0 / value of lane (l+1), `last` for lane 63.
__device__ __forceinline__ float wave_shift_up(float v, float first, int lane) {
    const float t = __shfl_up(v, 1, kWave);
    return lane == 0 ? first : t;
}
__device__ __forceinline__ float wave_reverse(float v, int lane) { return __shfl(v, kWave - 1 - lane, kWave); }

// ---- fast fp32 math -----------------------------------------------------------------------------------
// One hardware transcendental each (v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 / v_exp_f32, <= 1 ulp) instead of the
// correctly-rounded expansions (~10 instructions for a division).  The reference tolerance is 1e-4 on the
// silhouettes (BASELINE.json), five orders of magnitude above what these change; VSRD_PRECISE_MATH=1
// restores the IEEE sequences for A/B runs.
#ifndef VSRD_PRECISE_MATH
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
#else
__device__ __forceinline__ float fast_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float fast_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return 1.0f / sqrtf(x); }
__device__ __forceinline__ float fast_exp(float x) { return expf(x); }
__device__ __forceinline__ float fast_log(float x) { return logf(x); }
#endif

// ---- Philox4x32-10 (counter-based; the same generator family torch uses on device) --------------
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    constexpr uint32_t kM0 = 0xD2511F53u, kM1 = 0xCD9E8D57u, kW0 = 0x9E3779B9u, kW1 = 0xBB67AE85u;
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint32_t hi0 = __umulhi(kM0, c0), lo0 = kM0 * c0;
        const uint32_t hi1 = __umulhi(kM1, c2), lo1 = kM1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += kW0; k1 += kW1;
    }
    return {c0, c1, c2, c3};
}

// 24 random bits -> [0, 1) exactly as torch's uniform transformation for float.
__device__ __forceinline__ float uniform_from_bits(uint32_t bits) { return static_cast<float>(bits >> 8) * 5.9604644775390625e-08f; }

}  // namespace vsrd
