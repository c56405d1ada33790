// Importance sampling of rays: `num_samples` distinct indices drawn without replacement with probability proportional to
// `weights` -- what scripts/main.py:620-627 does with torch.multinomial(weights, num_rays, replacement=False).
//
// Same algorithm as ATen's multinomial-without-replacement (an exponential race: key_i = w_i / E_i, E_i ~ Exp(1), keep the k
// largest keys), written so that it is a handful of streaming launches with no host involvement and no 9M-element sort:
//   1. keys_histogram_kernel   one pass over the weights: key from Philox4x32-10(seed, step; i), histogram of the key's top 12 bits
//   2. threshold_kernel        the bin in which the k-th largest key lies
//   3. collect_kernel          second pass (keys are re-generated, not stored): append every key in or above that bin
//                              (k + ~k/8 candidates: P(key > t) ~ sum(w) / t, and a bin is 12.5 % wide)
//   4. select_kernel           one workgroup sorts the candidates (bitonic, LDS) by (key desc, index asc) and emits the first k
// The output is a deterministic function of (weights, seed, step): the candidate SET does not depend on the append order and the
// final sort breaks ties by index.  ATen's version sorts all M keys with a device radix sort (0.3 ms at M = 9M on MI355X, and its
// captured form faults on replay in torch 2.10+rocm7.0); this one reads the weights twice (72 MB at M = 9M).
#pragma once
#include "wave.h"

namespace vsrd {

constexpr int kSampleBins = 4096;                 // top 12 bits of a non-negative float: 8 exponent bits + 3 mantissa bits
constexpr int kSampleCandidates = 4096;           // capacity of the candidate list and of the final sort (48 KB of LDS)
constexpr int kSampleMax = 2048;                  // largest k: k + k/8 + slack must fit the candidate list

struct SampleScratch {                            // device workspace of one sampling call
    unsigned histogram[kSampleBins];
    unsigned threshold_bin;
    unsigned num_candidates;
    unsigned overflow;                            // sticky: some call found more candidates than the list holds (the surplus was dropped)
    unsigned pad;
    float candidate_keys[kSampleCandidates];
    long long candidate_indices[kSampleCandidates];
};

__device__ __forceinline__ float race_key(float weight, long long index, unsigned long long seed, unsigned long long step) {
    const Philox4 rnd = philox4x32_10(static_cast<uint32_t>(index), static_cast<uint32_t>(index >> 32), static_cast<uint32_t>(step),
                                      static_cast<uint32_t>(step >> 32) ^ 0x52415953u /* "RAYS": a stream of its own */,
                                      static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
    const float u = uniform_from_bits(rnd.x);                     // [0, 1)
    const float e = -__logf(1.0f - u) + 1.0e-30f;                 // Exp(1), never 0
    return (weight > 0.0f) ? weight / e : 0.0f;
}

__global__ __launch_bounds__(256) void sample_clear_kernel(SampleScratch* scratch) {
    for (int i = threadIdx.x; i < kSampleBins; i += blockDim.x) scratch->histogram[i] = 0u;
    if (threadIdx.x == 0) { scratch->threshold_bin = 0u; scratch->num_candidates = 0u; }     // (overflow is sticky: the host clears it)
}

__global__ __launch_bounds__(256) void keys_histogram_kernel(const float* __restrict__ weights, long long count, unsigned long long seed,
                                                             unsigned long long step, const unsigned long long* __restrict__ device_step,
                                                             SampleScratch* scratch) {
    __shared__ unsigned local[kSampleBins];
    if (device_step != nullptr) step = *device_step;
    for (int i = threadIdx.x; i < kSampleBins; i += blockDim.x) local[i] = 0u;
    __syncthreads();
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += static_cast<long long>(gridDim.x) * blockDim.x) {
        const float key = race_key(weights[i], i, seed, step);
        if (key > 0.0f) atomicAdd(&local[__float_as_uint(key) >> 20], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSampleBins; i += blockDim.x)
        if (local[i] != 0u) atomicAdd(&scratch->histogram[i], local[i]);
}

// The bin b with  count(bins > b) < k <= count(bins >= b);  0 when fewer than k weights are positive (then everything positive is kept).
// One wave: lane l owns the 64 bins [4096 - 64 (l + 1), 4096 - 64 l), i.e. lane 0 the highest keys.
__global__ __launch_bounds__(kWave) void threshold_kernel(SampleScratch* scratch, int k) {
    const int lane = lane_id();
    const int top = kSampleBins - 64 * lane;                       // one past my highest bin
    unsigned mine = 0u;
    for (int b = 0; b < 64; ++b) mine += scratch->histogram[top - 1 - b];
    unsigned inclusive = mine;
#pragma unroll
    for (int offset = 1; offset < kWave; offset <<= 1) {
        const unsigned other = __shfl_up(inclusive, offset, kWave);
        if (lane >= offset) inclusive += other;
    }
    const unsigned above = inclusive - mine;                       // keys in the chunks above mine
    const bool holds = above < static_cast<unsigned>(k) && inclusive >= static_cast<unsigned>(k);
    if (holds) {
        unsigned running = above;
        int bin = top - 1;
        for (; bin > top - 64; --bin) {
            running += scratch->histogram[bin];
            if (running >= static_cast<unsigned>(k)) break;
        }
        scratch->threshold_bin = static_cast<unsigned>(bin);
    }
    // no lane holds it when fewer than k keys are positive: threshold_bin stays 0 (sample_clear_kernel)
}

__global__ __launch_bounds__(256) void collect_kernel(const float* __restrict__ weights, long long count, unsigned long long seed,
                                                      unsigned long long step, const unsigned long long* __restrict__ device_step,
                                                      SampleScratch* scratch) {
    if (device_step != nullptr) step = *device_step;
    const unsigned threshold = scratch->threshold_bin;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += static_cast<long long>(gridDim.x) * blockDim.x) {
        const float key = race_key(weights[i], i, seed, step);
        if (key > 0.0f && (__float_as_uint(key) >> 20) >= threshold) {
            const unsigned slot = atomicAdd(&scratch->num_candidates, 1u);
            if (slot < kSampleCandidates) { scratch->candidate_keys[slot] = key; scratch->candidate_indices[slot] = i; }
            else scratch->overflow = 1u;          // a crowded threshold bin: which candidates were kept depends on the atomic order
        }
    }
}

// (key desc, index asc) order on (key, index) pairs.
__device__ __forceinline__ bool race_before(float ka, long long ia, float kb, long long ib) { return (ka > kb) || (ka == kb && ia < ib); }

// One workgroup sorts the n (~1.1 k) candidates by (key desc, index asc) with a bitonic network over the next power of two of n
// (LDS) and emits the first k: the order is unique, so the output is deterministic.
__global__ __launch_bounds__(1024) void select_kernel(SampleScratch* scratch, int k, long long* __restrict__ indices) {
    __shared__ float keys[kSampleCandidates];
    __shared__ long long ids[kSampleCandidates];
    const int n = static_cast<int>(min(scratch->num_candidates, static_cast<unsigned>(kSampleCandidates)));
    int padded = 2;
    while (padded < n) padded <<= 1;
    for (int i = threadIdx.x; i < padded; i += blockDim.x) {
        keys[i] = (i < n) ? scratch->candidate_keys[i] : -1.0f;                      // empty slots sort last
        ids[i] = (i < n) ? scratch->candidate_indices[i] : 0x7fffffffffffffffll;
    }
    __syncthreads();
    for (int size = 2; size <= padded; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < padded / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool best_first = (lo & size) == 0;
                const bool in_order = race_before(keys[lo], ids[lo], keys[hi], ids[hi]);
                if (in_order != best_first) {
                    const float tk = keys[lo]; keys[lo] = keys[hi]; keys[hi] = tk;
                    const long long ti = ids[lo]; ids[lo] = ids[hi]; ids[hi] = ti;
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < k; i += blockDim.x) indices[i] = (i < padded && keys[i] > 0.0f) ? ids[i] : -1ll;   // -1: fewer than k positive weights
}

}  // namespace vsrd
