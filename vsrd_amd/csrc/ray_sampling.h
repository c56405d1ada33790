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

// ---- the same draw from a per-frame table: one launch per step ----------------------------------------------------------------------
// A frame's importance weights do not change over its 3000 steps (scripts/main.py:204-265 builds the soft masks once, :620-627 draws
// from them every step), so the two passes over all the weights above can be paid once per frame instead of once per step:
//   build (once per frame)   fixed-point weights  f_i = max(1, floor(w_i * 2^b / max w))  for w_i > 0 (b = 62 - ceil(log2 count), at most
//                            52), 0 otherwise, and their inclusive prefix sums as 64-bit integers.  Integer sums are exact in any order,
//                            so the table is monotone and deterministic whatever the scan tree (a floating-point scan is neither).
//   draw (once per step)     ONE workgroup: candidates j = 0, 1, 2, ... are i.i.d. picks from the table (Philox4x32-10 keyed by (seed,
//                            step; j) -> 64 random bits -> target in [0, total) -> first index whose prefix sum exceeds it: a guide
//                            table indexed by the top bits brackets it to ~1 entry, an 8-ary search finishes -- a search from scratch
//                            is 56 scattered loads per pick, and the 115 k of a draw through the L1 of the one CU took 60 us); a
//                            candidate is ACCEPTED unless an earlier candidate picked the
//                            same index (LDS hash table, atomicMin of the candidate number per index); the first k accepted candidates,
//                            in candidate order, are the sample.
// Drawing i.i.d. and skipping repeats IS successive sampling without replacement, which is also what the exponential race produces
// (the k largest keys w_i / E_i are the first k distinct arrivals of independent Poisson processes of rates w_i): the same distribution
// over ordered samples as vsrd_sample_rays and torch.multinomial(replacement=False).  What differs is the tail behaviour: the race
// always finishes in two passes; skipping repeats needs about k / (weight mass outside the k - 1 heaviest entries) candidates, so the
// caller checks that mass once per frame (rendering/samplers.py::RayTable.suits) and keeps the race for frames that fail it.  A draw
// that is still short after kTableRounds x 2048 candidates fills the rest with repeats of its first picks (valid indices, no longer
// distinct) and sets the table's sticky `incomplete` flag.
constexpr int kTableChunk = 4096;                 // weights per workgroup of the build kernels (256 threads x 16)
constexpr int kTableThreads = 1024;               // the draw: one workgroup, two candidates per thread and round
constexpr int kTableRounds = 16;
constexpr int kTableSlots = 8192;                 // hash table: <= 2048 accepted + 2048 new indices per round -> load <= 0.5
constexpr unsigned kTableEmpty = 0xffffffffu;

struct RayTableHeader {                           // 64 bytes in front of the prefix sums
    unsigned long long total;                     // sum of the fixed-point weights
    unsigned long long num_positive;
    unsigned long long last_positive;             // largest index with a positive weight
    unsigned int max_bits;                        // bit pattern of the largest weight (non-negative floats order like unsigned integers)
    unsigned int incomplete;                      // sticky: some draw ran out of candidates (its tail repeats its first picks)
    unsigned long long pad[4];
};
static_assert(sizeof(RayTableHeader) == 64, "RayTableHeader is the first 64 bytes of the table");

__host__ __device__ inline long long ray_table_blocks(long long count) { return (count + kTableChunk - 1) / kTableChunk; }
// The guide: 2^b >= count buckets over the 64 random bits of a pick (bucket = their top b bits); guide[g] = the pick of the first bit
// pattern of bucket g, guide[2^b] = the last positive index.  Picks are monotone in the bits, so a pick of bucket g lies in
// [guide[g], guide[g + 1]] -- on average less than one table entry wide, against the whole table for a search from scratch.
__host__ __device__ inline int ray_table_guide_bits(long long count) {
    int bits = 10;
    while (bits < 26 && (1ll << bits) < count) ++bits;
    return bits;
}
__host__ __device__ inline size_t ray_table_guide_offset(long long count) {       // bytes from the start of the table
    return sizeof(RayTableHeader) + sizeof(unsigned long long) * (static_cast<size_t>(count) + static_cast<size_t>(ray_table_blocks(count)));
}
__host__ __device__ inline size_t ray_table_bytes(long long count) {
    return ray_table_guide_offset(count) + sizeof(unsigned) * ((static_cast<size_t>(1) << ray_table_guide_bits(count)) + 2);
}
__device__ __forceinline__ unsigned long long* ray_table_sums(RayTableHeader* table) { return reinterpret_cast<unsigned long long*>(table + 1); }
__device__ __forceinline__ const unsigned long long* ray_table_sums(const RayTableHeader* table) { return reinterpret_cast<const unsigned long long*>(table + 1); }

__device__ __forceinline__ int ray_table_bits(long long count) {
    int log2_count = 0;
    while ((1ll << log2_count) < count) ++log2_count;
    const int bits = 62 - log2_count;
    return bits > 52 ? 52 : bits;
}

__device__ __forceinline__ unsigned long long ray_table_fixed(float weight, double scale) {
    if (!(weight > 0.0f)) return 0ull;
    const unsigned long long f = static_cast<unsigned long long>(static_cast<double>(weight) * scale);
    return f > 0ull ? f : 1ull;                                    // every positive weight stays drawable
}

// Exclusive prefix sum of one value per thread over the workgroup (Hillis-Steele in LDS; integers: exact in any order).  `lds` holds
// blockDim.x + 1 entries; the workgroup's total is left in lds[blockDim.x].
__device__ __forceinline__ unsigned long long block_exclusive_sum_u64(unsigned long long value, unsigned long long* lds) {
    const int tid = threadIdx.x, n = blockDim.x;
    __syncthreads();
    lds[tid] = value;
    __syncthreads();
    unsigned long long inclusive = value;
    for (int offset = 1; offset < n; offset <<= 1) {
        const unsigned long long other = (tid >= offset) ? lds[tid - offset] : 0ull;
        __syncthreads();
        inclusive += other;
        lds[tid] = inclusive;
        __syncthreads();
    }
    if (tid == n - 1) lds[n] = inclusive;
    __syncthreads();
    return inclusive - value;
}

__global__ __launch_bounds__(256) void ray_table_max_kernel(const float* __restrict__ weights, long long count, RayTableHeader* table) {
    unsigned best = 0u;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += static_cast<long long>(gridDim.x) * blockDim.x) {
        const float w = weights[i];
        if (w > 0.0f) best = max(best, __float_as_uint(w));
    }
#pragma unroll
    for (int offset = kWave / 2; offset > 0; offset >>= 1) best = max(best, static_cast<unsigned>(__shfl_xor(static_cast<int>(best), offset, kWave)));
    if (lane_id() == 0 && best != 0u) atomicMax(&table->max_bits, best);
}

__device__ __forceinline__ double ray_table_scale(const RayTableHeader* table, long long count) {
    const float largest = __uint_as_float(table->max_bits);
    return largest > 0.0f ? ldexp(1.0, ray_table_bits(count)) / static_cast<double>(largest) : 0.0;
}

// Pass 1: the sum of each chunk of 4096 fixed-point weights (behind the prefix sums), the number of positive weights, the last of them.
__global__ __launch_bounds__(256) void ray_table_sums_kernel(const float* __restrict__ weights, long long count, RayTableHeader* table) {
    __shared__ unsigned long long lds[256 + 1];
    const double scale = ray_table_scale(table, count);
    const long long first = static_cast<long long>(blockIdx.x) * kTableChunk + static_cast<long long>(threadIdx.x) * 16;
    unsigned long long sum = 0ull, positives = 0ull;
    long long last = -1;
    for (int e = 0; e < 16; ++e) {
        const long long i = first + e;
        const unsigned long long f = (i < count) ? ray_table_fixed(weights[i], scale) : 0ull;
        sum += f;
        if (f != 0ull) { ++positives; last = i; }
    }
    block_exclusive_sum_u64(sum, lds);
    if (threadIdx.x == 0) (ray_table_sums(table) + count)[blockIdx.x] = lds[256];
    block_exclusive_sum_u64(positives, lds);
    if (threadIdx.x == 0 && lds[256] != 0ull) atomicAdd(&table->num_positive, lds[256]);
    // the workgroup's last positive index: the threads' indices ascend with the thread number, so it is the last thread that has one
    __shared__ int last_thread;
    if (threadIdx.x == 0) last_thread = -1;
    __syncthreads();
    if (last >= 0) atomicMax(&last_thread, static_cast<int>(threadIdx.x));
    __syncthreads();
    if (last_thread == static_cast<int>(threadIdx.x)) atomicMax(&table->last_positive, static_cast<unsigned long long>(last));      // one global atomic per workgroup
}

// Pass 2 (one workgroup): chunk sums -> exclusive offsets, in place; the grand total.
__global__ __launch_bounds__(1024) void ray_table_offsets_kernel(long long count, RayTableHeader* table) {
    __shared__ unsigned long long lds[1024 + 1];
    unsigned long long* sums = ray_table_sums(table) + count;
    const long long blocks = ray_table_blocks(count);
    const long long per_thread = (blocks + 1023) / 1024;
    const long long begin = static_cast<long long>(threadIdx.x) * per_thread;
    unsigned long long mine = 0ull;
    for (long long b = begin; b < begin + per_thread && b < blocks; ++b) mine += sums[b];
    unsigned long long running = block_exclusive_sum_u64(mine, lds);
    for (long long b = begin; b < begin + per_thread && b < blocks; ++b) {
        const unsigned long long s = sums[b];
        sums[b] = running;
        running += s;
    }
    if (threadIdx.x == 0) table->total = lds[1024];
}

// Pass 3: the inclusive prefix sums themselves.
__global__ __launch_bounds__(256) void ray_table_fill_kernel(const float* __restrict__ weights, long long count, RayTableHeader* table) {
    __shared__ unsigned long long lds[256 + 1];
    const double scale = ray_table_scale(table, count);
    unsigned long long* cdf = ray_table_sums(table);
    const long long first = static_cast<long long>(blockIdx.x) * kTableChunk + static_cast<long long>(threadIdx.x) * 16;
    unsigned long long f[16], sum = 0ull;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const long long i = first + e;
        f[e] = (i < count) ? ray_table_fixed(weights[i], scale) : 0ull;
        sum += f[e];
    }
    unsigned long long running = cdf[count + blockIdx.x] + block_exclusive_sum_u64(sum, lds);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        running += f[e];
        if (first + e < count) cdf[first + e] = running;
    }
}

// First index in [lo, hi] whose inclusive prefix sum exceeds `target` (the caller knows it lies there; target < total = cdf[last]).
// 8-ary: the 7 pivots of a round are independent loads, so a search over 2^22 entries is 8 memory round trips instead of 22.
__device__ __forceinline__ long long ray_table_pick(const unsigned long long* __restrict__ cdf, long long lo, long long hi, unsigned long long target) {
    // the answer lies in [lo, hi]
    while (lo < hi) {
        const long long width = hi - lo;
        long long pivot[7];
        unsigned long long value[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) { pivot[j] = lo + ((width * (j + 1)) >> 3); value[j] = cdf[pivot[j]]; }      // lo <= pivot <= hi - 1
        int below = 0;                                             // pivots whose prefix sum is <= target (a prefix of the seven)
#pragma unroll
        for (int j = 0; j < 7; ++j) below += (value[j] <= target) ? 1 : 0;
        long long new_lo = lo, new_hi = hi;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (j + 1 == below) new_lo = pivot[j] + 1;
            if (j == below) new_hi = pivot[j];
        }
        lo = new_lo; hi = new_hi;
    }
    return lo;
}

// Pass 4: the guide (one thread per bucket boundary).
__global__ __launch_bounds__(256) void ray_table_guide_kernel(long long count, RayTableHeader* table) {
    const int bits = ray_table_guide_bits(count);
    const long long g = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g > (1ll << bits)) return;
    unsigned* guide = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(table) + ray_table_guide_offset(count));
    const long long last = static_cast<long long>(table->last_positive);
    long long pick = last;
    if (g < (1ll << bits) && table->total != 0ull)
        pick = ray_table_pick(ray_table_sums(table), 0, last, __umul64hi(static_cast<unsigned long long>(g) << (64 - bits), table->total));
    guide[g] = static_cast<unsigned>(pick);
}

// LDS: keys [kTableSlots] (the index that owns a slot) | first [kTableSlots] (its earliest candidate number) | wave totals [16] | total
constexpr size_t kTableLdsBytes = sizeof(unsigned) * (2 * kTableSlots + kTableThreads / kWave + 1);

__device__ __forceinline__ void sample_table_body(RayTableHeader* table, long long count, int k, unsigned long long seed,
                                                  unsigned long long step, const unsigned long long* __restrict__ device_step,
                                                  const long long* __restrict__ remap, long long* __restrict__ indices) {
    extern __shared__ __attribute__((aligned(16))) unsigned table_lds[];
    unsigned* keys = table_lds;
    unsigned* first = table_lds + kTableSlots;
    unsigned* wave_totals = first + kTableSlots;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
#ifdef VSRD_TABLE_TIMERS
    const unsigned long long t_start = wall_clock64();
#endif
    if (device_step != nullptr) step = *device_step;
    const unsigned long long total = table->total;
    const unsigned long long* cdf = ray_table_sums(table);
    const unsigned* guide = reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(table) + ray_table_guide_offset(count));
    const int guide_shift = 64 - ray_table_guide_bits(count);
    const int need = static_cast<int>(min(static_cast<unsigned long long>(k), table->num_positive));
    for (int i = tid; i < kTableSlots; i += kTableThreads) { keys[i] = kTableEmpty; first[i] = kTableEmpty; }
    for (int i = tid; i < k; i += kTableThreads) indices[i] = -1ll;                        // the tail when fewer than k weights are positive
    __syncthreads();
#ifdef VSRD_TABLE_TIMERS
    if (tid == 0) table->pad[0] = wall_clock64() - t_start;
#endif
    int accepted = 0;
    for (int round = 0; round < kTableRounds && accepted < need; ++round) {
        unsigned number[2], slot[2], lo[2], hi[2];
        unsigned long long target[2];
        long long pick[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            number[c] = static_cast<unsigned>(round) * (2 * kTableThreads) + 2 * tid + c;
            const Philox4 rnd = philox4x32_10(number[c], 0u, static_cast<uint32_t>(step), static_cast<uint32_t>(step >> 32) ^ 0x52415954u /* "RAYT" */,
                                              static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
            const unsigned long long bits = (static_cast<unsigned long long>(rnd.x) << 32) | rnd.y;
            target[c] = __umul64hi(bits, total);                                           // uniform on [0, total)
            lo[c] = guide[bits >> guide_shift];
            hi[c] = guide[(bits >> guide_shift) + 1];
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) pick[c] = ray_table_pick(cdf, lo[c], hi[c], target[c]);
#ifdef VSRD_TABLE_TIMERS
        if (tid == 0 && round == 0) table->pad[1] = wall_clock64() - t_start;
#endif
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const unsigned key = static_cast<unsigned>(pick[c]);
            unsigned s = (key * 2654435761u) >> 19;                                        // 13 bits
            while (true) {
                const unsigned owner = atomicCAS(&keys[s], kTableEmpty, key);
                if (owner == kTableEmpty || owner == key) break;
                s = (s + 1) & (kTableSlots - 1);
            }
            atomicMin(&first[s], number[c]);
            slot[c] = s;
        }
        __syncthreads();
#ifdef VSRD_TABLE_TIMERS
        if (tid == 0 && round == 0) table->pad[2] = wall_clock64() - t_start;
#endif
        const int keep0 = first[slot[0]] == number[0], keep1 = first[slot[1]] == number[1];
        int inclusive = keep0 + keep1;
#pragma unroll
        for (int offset = 1; offset < kWave; offset <<= 1) {
            const int other = __shfl_up(inclusive, offset, kWave);
            if (lane >= offset) inclusive += other;
        }
        if (lane == kWave - 1) wave_totals[wave] = static_cast<unsigned>(inclusive);
        __syncthreads();
        int before = 0, all = 0;
        for (int w = 0; w < kTableThreads / kWave; ++w) {
            const int t = static_cast<int>(wave_totals[w]);
            if (w < wave) before += t;
            all += t;
        }
        int rank = accepted + before + inclusive - keep0 - keep1;
        if (keep0) { if (rank < k) indices[rank] = remap ? remap[pick[0]] : pick[0]; ++rank; }
        if (keep1 && rank < k) indices[rank] = remap ? remap[pick[1]] : pick[1];
        accepted += all;
        __syncthreads();
    }
#ifdef VSRD_TABLE_TIMERS
    if (tid == 0) table->pad[3] = wall_clock64() - t_start;
#endif
    if (accepted < need) {                      // out of candidates: the rest repeats the first picks (valid indices, not distinct) and the table says so
        for (int i = accepted + tid; i < need; i += kTableThreads) indices[i] = indices[(i - accepted) % accepted];
        if (tid == 0) table->incomplete = 1u;
    }
}

__global__ __launch_bounds__(kTableThreads) void sample_table_kernel(RayTableHeader* table, long long count, int k, unsigned long long seed,
                                                                     unsigned long long step, const unsigned long long* __restrict__ device_step,
                                                                     const long long* __restrict__ remap, long long* __restrict__ indices) {
    sample_table_body(table, count, k, seed, step, device_step, remap, indices);
}

}  // namespace vsrd
